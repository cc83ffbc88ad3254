"""Time the refinement loop in its three forms on the GPU box:  python tools/finetune_time.py"""
import sys, time, importlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from conftest import pkg
R = pkg("host.refine"); M = pkg("host.Sg2ScVAE_model"); DR = pkg("host.diff_render")
NAMES = ["bed", "chair", "table", "sofa", "desk", "cabinet", "lamp", "television", "bookshelf", "dresser", "night_stand", "shelves", "__room__"]
n = len(NAMES)
g = torch.Generator().manual_seed(0)
lo = torch.rand(n, 3, generator=g) * 0.45 + 0.05; lo[:, 1] = 0.0; lo[:, 2] *= 0.6
hi = lo + torch.rand(n, 3, generator=g) * 0.2 + 0.12
boxes = torch.cat([lo, hi], 1); boxes[-1] = torch.tensor([0, 0, 0, 4.0, 2.7, 5.0]); boxes = boxes.cuda()
angles = torch.randint(0, 24, (n,), generator=g).cuda()
syn = pkg("host.synthetic")
torch.manual_seed(1)
model = M.Sg2ScVAEModel(vocab=syn.default_vocab(), batch_size=1, train_3d=True, decoder_cat=True, embedding_dim=64, gconv_mode='feedforward',
                        gconv_num_layers=5, mlp_normalization='batch', vec_noise_dim=0, layout_noise_dim=32, use_AE=False).cuda().train()
objs = torch.arange(1, n + 1).cuda(); objs[-1] = 0
triples = torch.tensor([[i, 1 + i % 10, (i + 1) % (n - 1)] for i in range(n - 1)] + [[i, 0, n - 1] for i in range(n - 1)]).cuda()
attrs = torch.zeros(n, dtype=torch.int64).cuda()
bank = R.MeshBank([n for n in NAMES if n != "__room__"], "cuda", seed=3)
sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
st = torch.cuda.Stream()
for mode in ("reference-shaped loop", "batched placement", "batched + hipGraph replay"):
    model.load_state_dict(sd0)
    with torch.cuda.stream(st):
        def run(iters):
            if mode.startswith("reference"):
                return R.finetune_vae(model, objs, triples, boxes, angles, attrs, NAMES, iters=iters, bank=bank)[0]
            return R.finetune_vae_fast(model, objs, triples, boxes, angles, attrs, NAMES, iters=iters, bank=bank, capture="Graph" in mode)[0]
        run(3); torch.cuda.synchronize()
        model.load_state_dict(sd0)
        t0 = time.perf_counter(); l = run(60); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    l = l.tolist() if torch.is_tensor(l) else l
    print("%-28s %7.2f ms/iteration (60 iterations %.3f s)  loss %.4g -> %.4g" % (mode, dt / 60 * 1e3, dt, l[0], l[-1]))
