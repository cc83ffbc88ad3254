"""Soak test of the refinement loop on the GPU box: 30 rooms, one after the other (a RefineScene + RefineLoss + hipGraph per room, 20
iterations each, a fresh copy of the model per room as test_render_refine.py:250-263 does); no memory may pile up."""
import sys, time, importlib, copy
sys.path.insert(0, "/root/repo")
import torch
R = importlib.import_module("3d_sln_amd.host.refine"); M = importlib.import_module("3d_sln_amd.host.Sg2ScVAE_model")
syn = importlib.import_module("3d_sln_amd.host.synthetic")
NAMES = ["bed", "chair", "table", "sofa", "desk", "cabinet", "lamp", "television", "bookshelf", "dresser", "night_stand", "shelves", "__room__"]
torch.manual_seed(1)
base = M.Sg2ScVAEModel(vocab=syn.default_vocab(), batch_size=1, train_3d=True, decoder_cat=True, embedding_dim=64, gconv_mode='feedforward',
                       gconv_num_layers=5, mlp_normalization='batch', vec_noise_dim=0, layout_noise_dim=32, use_AE=False).cuda().train()
sd0 = {k: v.detach().clone() for k, v in base.state_dict().items()}
bank = R.MeshBank([n for n in NAMES if n != "__room__"], "cuda", seed=3)
st = torch.cuda.Stream()
mem = []
t0 = time.perf_counter()
with torch.cuda.stream(st):
    for room in range(30):
        n = 6 + room % 7
        names = NAMES[:n] + ["__room__"]
        g = torch.Generator().manual_seed(room)
        lo = torch.rand(n + 1, 3, generator=g) * 0.45 + 0.05; lo[:, 1] = 0.0; lo[:, 2] *= 0.6
        hi = lo + torch.rand(n + 1, 3, generator=g) * 0.2 + 0.12
        boxes = torch.cat([lo, hi], 1); boxes[-1] = torch.tensor([0, 0, 0, 4.0, 2.7, 5.0]); boxes = boxes.cuda()
        angles = torch.randint(0, 24, (n + 1,), generator=g).cuda()
        objs = torch.arange(1, n + 2).cuda(); objs[-1] = 0
        triples = torch.tensor([[i, 1 + i % 10, (i + 1) % n] for i in range(n)] + [[i, 0, n] for i in range(n)]).cuda()
        attrs = torch.zeros(n + 1, dtype=torch.int64).cuda()
        base.load_state_dict(sd0)
        losses, _ = R.finetune_vae_fast(base, objs, triples, boxes, angles, attrs, names, iters=20, bank=bank, capture=(room % 2 == 0))
        assert torch.isfinite(losses).all(), room
        torch.cuda.synchronize()
        mem.append(torch.cuda.memory_allocated())
print("rooms 30, %.1f s, memory after rooms 5 / 15 / 30: %.1f / %.1f / %.1f MB" % (time.perf_counter() - t0, mem[4] / 1e6, mem[14] / 1e6, mem[29] / 1e6))
assert mem[29] <= mem[9] * 1.05 + 8e6, mem
print("refine soak ok")
