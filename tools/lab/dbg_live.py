import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
R = importlib.import_module("3d_sln_amd.host.refine"); M = importlib.import_module("3d_sln_amd.host.Sg2ScVAE_model"); syn = importlib.import_module("3d_sln_amd.host.synthetic")
names = ["bed", "chair", "table", "sofa", "desk", "cabinet", "lamp", "television", "bookshelf", "dresser", "night_stand", "shelves", "__room__"]
n = len(names)
NR = int(os.environ.get('NR', '16'))
def mk(seed0, trainmode):
    torch.manual_seed(1)
    model = M.Sg2ScVAEModel(vocab=syn.default_vocab(), batch_size=1, train_3d=True, decoder_cat=True, embedding_dim=64, gconv_mode='feedforward',
                            gconv_num_layers=5, mlp_normalization='batch', vec_noise_dim=0, layout_noise_dim=32, use_AE=False).cuda()
    model.train(trainmode)
    objs = torch.arange(1, n + 1).cuda(); objs[-1] = 0
    triples = torch.tensor([[i, 1 + i % 10, (i + 1) % (n - 1)] for i in range(n - 1)] + [[i, 0, n - 1] for i in range(n - 1)]).cuda()
    attrs = torch.zeros(n, dtype=torch.int64).cuda()
    rooms = []
    for r in range(NR):
        gr = torch.Generator().manual_seed(seed0 + r)
        lo = torch.rand(n, 3, generator=gr) * 0.45 + 0.05; lo[:, 1] = 0.0; lo[:, 2] *= 0.6
        hi = lo + torch.rand(n, 3, generator=gr) * 0.2 + 0.12
        bx = torch.cat([lo, hi], 1); bx[-1] = torch.tensor([0, 0, 0, 4.0, 2.7, 5.0])
        rooms.append(dict(objs=objs, triples=triples, boxes=bx.cuda(), angles=torch.randint(0, 24, (n,), generator=gr).cuda(), attributes=attrs, class_names=names))
    return model, rooms
st = torch.cuda.Stream()
for seed0, tm in ((0, False), (100, False), (100, True)):
    with torch.cuda.stream(st):
        model, rooms = mk(seed0, tm)
        l = syn.overfit_to_rooms(model, rooms, steps=400)
        bank = R.MeshBank([x for x in names if x != "__room__"], "cuda", seed=3)
        rb = R.RefineBatch(model, rooms[:16], bank=bank, iters=60)
        out = []
        for k in (1, 9, 50):
            rb.run(k); torch.cuda.synchronize()
            lv = rb.live.cpu(); out.append(float((lv == 3).sum()) / 16)
        err = float((rb.boxes.view(16, n, 6)[:, :-1] - torch.stack([r["boxes"][:-1] for r in rooms[:16]])).abs().mean())
        print("seed0", seed0, "train-mode model", tm, "fit", [round(float(x), 4) for x in l], "live after 1/10/60 iterations", out, "box err", round(err, 4), flush=True)
        rb.close()
# the bench's order: over-fit (65 rooms), snapshot, one-room loop (eager + capture) on the model itself, restore, batch
with torch.cuda.stream(st):
    model, rooms = mk(100, True)
    g = torch.Generator().manual_seed(0)
    lo = torch.rand(n, 3, generator=g) * 0.45 + 0.05; lo[:, 1] = 0.0; lo[:, 2] *= 0.6
    hi = lo + torch.rand(n, 3, generator=g) * 0.2 + 0.12
    boxes = torch.cat([lo, hi], 1); boxes[-1] = torch.tensor([0, 0, 0, 4.0, 2.7, 5.0]); boxes = boxes.cuda()
    angles = torch.randint(0, 24, (n,), generator=g).cuda()
    one = dict(objs=rooms[0]["objs"], triples=rooms[0]["triples"], boxes=boxes, angles=angles, attributes=rooms[0]["attributes"])
    l = syn.overfit_to_rooms(model, rooms + [one], steps=400)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    bank = R.MeshBank([x for x in names if x != "__room__"], "cuda", seed=3)
    def live_now(tag):
        rb = R.RefineBatch(model, rooms[:16], bank=bank, iters=60); rb.run(3); torch.cuda.synchronize()
        lv = rb.live.cpu(); print(tag, "live", float((lv == 3).sum()) / 16, flush=True); rb.close()
    live_now("after fit")
    R.finetune_vae_fast(model, one["objs"], one["triples"], boxes, angles, one["attributes"], names, iters=3, bank=bank)
    model.load_state_dict(sd0); live_now("after eager one-room loop + restore")
    R.finetune_vae_fast(model, one["objs"], one["triples"], boxes, angles, one["attributes"], names, iters=10, bank=bank, capture=True)
    model.load_state_dict(sd0); live_now("after captured one-room loop + restore")
    print("training flag", model.training)
