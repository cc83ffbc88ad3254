"""Refinement loop, one room: ms per iteration (slope between 40 and 120 iterations) eager or as a replayed hipGraph.
   python tools/refine_profile.py [eager|graph]          (under rocprofv3 --kernel-trace --stats: the kernels of the iterations)"""
import sys, time, importlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from conftest import pkg
R = pkg("host.refine"); M = pkg("host.Sg2ScVAE_model")
mode = sys.argv[1] if len(sys.argv) > 1 else "eager"
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
with torch.cuda.stream(st):
    def run(iters):
        model.load_state_dict(sd0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        R.finetune_vae_fast(model, objs, triples, boxes, angles, attrs, NAMES, iters=iters, bank=bank, capture=mode == "graph")
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    run(5)
    a = min(run(40) for _ in range(2)); b = min(run(120) for _ in range(2))
print("%s: %.3f ms per iteration (slope), %.2f ms set-up" % (mode, (b - a) / 80 * 1e3, (a - 40 * (b - a) / 80) * 1e3))
