"""The unchanged colorize loop (one map, 50 z, one call per z at batch 1) - per-kernel time of the calls that use the kept planes.
   rocprofv3 --kernel-trace --stats -- python tools/lab/spade_b1_profile.py        (GPU box)"""
import importlib, os, sys, time
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
S = importlib.import_module("3d_sln_amd.host.SPADE_related")
torch.manual_seed(0)
G = S.SPADEGenerator4(41, 3, 256, 64, 'spectralspadelayer3x3', 256, 'normal').cuda().eval()
g = torch.Generator(device="cuda").manual_seed(0)
low = torch.rand(1, 1, 16, 16, device="cuda", generator=g) * 2 - 1
depth = F.interpolate(low, size=(256, 256), mode="bilinear", align_corners=False)
lab = F.interpolate(torch.randn(1, 40, 16, 16, device="cuda", generator=g), size=(256, 256), mode="bilinear", align_corners=False).argmax(1)
seg = torch.cat([depth, F.one_hot(lab, 40).permute(0, 3, 1, 2).float()], 1).contiguous()
zs = [torch.randn(1, 256, device="cuda", generator=g) for _ in range(50)]
with torch.no_grad():
    for z in zs[:3]:
        G(seg, z)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for z in zs:
        img = G(seg, z)
    torch.cuda.synchronize()
print("%.3f ms per call (batch 1, kept planes)" % ((time.perf_counter() - t0) / 50 * 1e3))
with torch.no_grad():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for z in zs:
        img = G(seg, z)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
print("host enqueue %.3f ms per call, GPU tail after the last enqueue %.3f ms (of 50 calls)" % ((t1 - t0) / 50 * 1e3, (t2 - t1) * 1e3))
