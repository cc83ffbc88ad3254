import cProfile, pstats, io, importlib, os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, "/root/repo")
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
    for z in zs[:3]: G(seg, z)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for z in zs: G(seg, z)
    pr.disable(); torch.cuda.synchronize()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(22); print(st.getvalue()[:5000])
