"""batch-1 SPADE calls on one map (testing/test_SPADE_shade.py:77-79): wall time per call; under rocprofv3 the per-kernel totals"""
import importlib, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
S = importlib.import_module("3d_sln_amd.host.SPADE_related"); syn = importlib.import_module("3d_sln_amd.host.synthetic")
torch.manual_seed(0)
G = S.SPADEGenerator4(41, 3, 256, 64, 'spectralspadelayer3x3', 256, 'normal').cuda().eval()
seg, z = syn.spade_input(2, seed=0); seg = seg.cuda()
g = torch.Generator(device="cuda").manual_seed(1)
zs = [torch.randn(1, 256, device="cuda", generator=g) for _ in range(50)]
with torch.no_grad():
    for zz in zs[:3]:
        G(seg[1:2].contiguous(), zz)
    for rep in range(3):
        total = seg[:1].clone()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for zz in zs:
            img = G(total, zz)
        t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print("50 calls: %.1f ms (%.3f ms per call), host enqueue %.1f ms" % ((t2 - t0) * 1e3, (t2 - t0) * 20, (t1 - t0) * 1e3), flush=True)
