import importlib, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
M = importlib.import_module("3d_sln_amd.host.Sg2ScVAE_model"); syn = importlib.import_module("3d_sln_amd.host.synthetic")
n = 13; NR = int(os.environ.get("NR", "64"))
torch.manual_seed(1)
model = M.Sg2ScVAEModel(vocab=syn.default_vocab(), batch_size=1, train_3d=True, decoder_cat=True, embedding_dim=64, gconv_mode='feedforward',
                        gconv_num_layers=5, mlp_normalization='batch', vec_noise_dim=0, layout_noise_dim=32, use_AE=False).cuda().train()
objs1 = torch.arange(1, n + 1).cuda(); objs1[-1] = 0
tri1 = torch.tensor([[i, 1 + i % 10, (i + 1) % (n - 1)] for i in range(n - 1)] + [[i, 0, n - 1] for i in range(n - 1)]).cuda()
rooms = []
for r in range(NR):
    gr = torch.Generator().manual_seed(100 + r)
    lo = torch.rand(n, 3, generator=gr) * 0.45 + 0.05; lo[:, 1] = 0.0; lo[:, 2] *= 0.6
    hi = lo + torch.rand(n, 3, generator=gr) * 0.2 + 0.12
    bx = torch.cat([lo, hi], 1); bx[-1] = torch.tensor([0, 0, 0, 4.0, 2.7, 5.0])
    rooms.append(dict(objs=objs1, triples=tri1, boxes=bx.cuda(), angles=torch.randint(0, 24, (n,), generator=gr).cuda(), attributes=torch.zeros(n, dtype=torch.int64).cuda()))
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    l = syn.overfit_to_rooms(model, rooms, steps=int(os.environ.get("STEPS", "400")))
    print("fit", [round(float(x), 4) for x in l])
    objs = torch.cat([r["objs"] for r in rooms]); tri = torch.cat([r["triples"] + torch.tensor([k * n, 0, k * n]).cuda() for k, r in enumerate(rooms)])
    boxes = torch.cat([r["boxes"] for r in rooms]).clone(); boxes[n - 1::n] = torch.tensor([0, 0, 0, 1.0, 1, 1]).cuda()
    angles = torch.cat([r["angles"] for r in rooms]); attrs = torch.cat([r["attributes"] for r in rooms])
    for mode in (True, False):
        model.train(mode)
        with torch.no_grad():
            mu, logvar = model.encoder(objs, tri, boxes, angles, attrs)
            bp, ap = model.decoder(mu, objs, tri, attrs)
        print("train" if mode else "eval ", "batch of %d rooms: |boxes_pred - target| mean %.4f max %.2f, |mu| %.3f" % (NR, float((bp - boxes).abs().mean()), float((bp - boxes).abs().max()), float(mu.abs().mean())))
    model.eval()
    with torch.no_grad():
        r = rooms[0]; b1 = boxes[:n]
        mu, logvar = model.encoder(r["objs"], r["triples"], b1, r["angles"], r["attributes"])
        bp, ap = model.decoder(mu, r["objs"], r["triples"], r["attributes"])
    print("eval  one room: |boxes_pred - target| mean %.4f, |mu| %.3f" % (float((bp - b1).abs().mean()), float(mu.abs().mean())))
    sd = model.state_dict()
    rv = [(k, float(v.min()), float(v.max())) for k, v in sd.items() if k.endswith("running_var")]
    print("running_var min over layers %.3e, max %.3e" % (min(x[1] for x in rv), max(x[2] for x in rv)))
    print([x for x in rv if x[1] < 1e-6][:5])
