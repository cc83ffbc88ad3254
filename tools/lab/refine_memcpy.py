"""Which ops of a refinement iteration end in device-to-device memcpys / tiny ATen kernels?  (GPU box)  python tools/lab/refine_memcpy.py"""
import importlib, sys, collections
sys.path.insert(0, "/root/repo")
import torch
from torch.profiler import profile, ProfilerActivity
R = importlib.import_module("3d_sln_amd.host.refine"); M = importlib.import_module("3d_sln_amd.host.Sg2ScVAE_model")
syn = importlib.import_module("3d_sln_amd.host.synthetic")
names = ["bed", "chair", "table", "sofa", "desk", "cabinet", "lamp", "television", "bookshelf", "dresser", "night_stand", "shelves", "__room__"]
n = len(names); g = torch.Generator().manual_seed(0)
lo = torch.rand(n, 3, generator=g) * 0.45 + 0.05; lo[:, 1] = 0.0; lo[:, 2] *= 0.6
hi = lo + torch.rand(n, 3, generator=g) * 0.2 + 0.12
boxes = torch.cat([lo, hi], 1); boxes[-1] = torch.tensor([0, 0, 0, 4.0, 2.7, 5.0]); boxes = boxes.cuda()
angles = torch.randint(0, 24, (n,), generator=g).cuda()
torch.manual_seed(1)
model = M.Sg2ScVAEModel(vocab=syn.default_vocab(), batch_size=1, train_3d=True, decoder_cat=True, embedding_dim=64, gconv_mode='feedforward',
                        gconv_num_layers=5, mlp_normalization='batch', vec_noise_dim=0, layout_noise_dim=32, use_AE=False).cuda().train()
objs = torch.arange(1, n + 1).cuda(); objs[-1] = 0
triples = torch.tensor([[i, 1 + i % 10, (i + 1) % (n - 1)] for i in range(n - 1)] + [[i, 0, n - 1] for i in range(n - 1)]).cuda()
attrs = torch.zeros(n, dtype=torch.int64).cuda()
bank = R.MeshBank([x for x in names if x != "__room__"], "cuda", seed=3)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    R.finetune_vae_fast(model, objs, triples, boxes, angles, attrs, names, iters=3, bank=bank)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
        R.finetune_vae_fast(model, objs, triples, boxes, angles, attrs, names, iters=10, bank=bank)
        torch.cuda.synchronize()
ev = prof.key_averages()
rows = sorted(ev, key=lambda e: -e.count)
print("%-60s %6s %10s %10s" % ("op", "count", "cpu_us", "cuda_us"))
for e in rows[:70]:
    print("%-60s %6d %10.1f %10.1f" % (e.key[:60], e.count, e.cpu_time_total, getattr(e, "device_time_total", getattr(e, "cuda_time_total", 0.0))))
