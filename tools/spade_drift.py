"""HIP SPADE generator vs the oracle in fp32 and fp64 on the same weights / inputs: whose fp32 is closer to the exact result?
   python tools/spade_drift.py [unit|default|oracle]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import spade_ref
S = importlib.import_module("3d_sln_amd.host.SPADE_related")
mode = sys.argv[1] if len(sys.argv) > 1 else "unit"
torch.manual_seed(0)
cfg = spade_ref.SpadeConfig()
G = S.SPADEGenerator4(41, 3, 256, 64, 'spectralspadelayer3x3', 256, 'normal')
if mode == "unit":
    with torch.no_grad():
        for name, p in G.named_parameters():
            if p.dim() > 1:
                p.normal_(0.0, 1.0 / float(p[0].numel()) ** 0.5)
                if name.startswith("conv_img"):
                    p.mul_(0.15)
            else:
                p.normal_(0.0, 0.05)
elif mode == "oracle":
    G.load_state_dict(spade_ref.init_state(cfg, seed=7))
G = G.cuda().eval()
seg, z = spade_ref.synth_input(cfg, 1, seed=3)
taps = {}
with torch.no_grad():
    out = G(seg.cuda(), z.cuda(), taps=taps).cpu()
sd = {k: v.detach().cpu() for k, v in G.state_dict().items()}
torch.set_num_threads(16)
t32, t64 = {}, {}
with torch.no_grad():
    r32 = spade_ref.generator(sd, cfg, seg, z, t32)
    sd64 = {k: v.double() for k, v in sd.items()}
    r64 = spade_ref.generator(sd64, cfg, seg.double(), z.double(), t64)
def e(a, b): return float((a.double() - b.double()).abs().max() / b.double().abs().max())
print("image: hip-vs-64 %.2e  cpu32-vs-64 %.2e  hip-vs-cpu32 %.2e" % (e(out, r64), e(r32, r64), e(out, r32)))
for k in taps:
    print("%-12s hip-vs-64 %.2e  cpu32-vs-64 %.2e   scale %.2e" % (k, e(taps[k].cpu(), t64[k]), e(t32[k], t64[k]), float(t64[k].abs().max())))
