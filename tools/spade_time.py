"""Batch-32 forward of the full-size SPADE generator (bench.py's `spade` leg without the checks): ms per batch.
   python tools/spade_time.py [iters]         environment switches of csrc/spade.hip apply (A/B runs on one box)"""
import importlib, os, sys, time
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
S = importlib.import_module("3d_sln_amd.host.SPADE_related")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
torch.manual_seed(0)
G = S.SPADEGenerator4(41, 3, 256, 64, 'spectralspadelayer3x3', 256, 'normal').cuda().eval()
B = 32
g = torch.Generator(device="cuda").manual_seed(0)
low = torch.rand(B, 1, 16, 16, device="cuda", generator=g) * 2 - 1
depth = F.interpolate(low, size=(256, 256), mode="bilinear", align_corners=False)
lab = F.interpolate(torch.randn(B, 40, 16, 16, device="cuda", generator=g), size=(256, 256), mode="bilinear", align_corners=False).argmax(1)
seg = torch.cat([depth, F.one_hot(lab, 40).permute(0, 3, 1, 2).float()], 1).contiguous()
z = torch.randn(B, 256, device="cuda", generator=g)
for _ in range(3):
    out = G(seg, z)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    out = G(seg, z)
torch.cuda.synchronize()
print("%.3f ms per batch of %d   (%s)" % ((time.perf_counter() - t0) / iters * 1e3, B,
      " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("SLN_")) or "defaults"))
