"""Lab: gradient image of the refinement loss with the LDS-staged backward kernel vs the strip-per-workgroup one (SLN_RBWD_OLD=1),
bit for bit (two processes: the switch is read once).  python tools/lab/rbwd_compare.py  (GPU box)"""
import importlib, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import torch
    R = importlib.import_module("3d_sln_amd.host.refine")
    g = torch.Generator().manual_seed(0)
    for B, S in ((3, 256), (2, 96)):
        tgt = torch.rand(B, 70, S, S, generator=g).cuda(); tgt[:, 1:41] = (tgt[:, 1:41] > 0.97).float()
        img = torch.rand(B, 70, S, S, generator=g).cuda().requires_grad_(True)
        rl = R.RefineLoss(tgt, per_room=True)
        out = rl(img)
        out[:, 0].sum().backward() if out.dim() > 1 else out[0].backward()
        np.save("/tmp/rbwd_%s_%d.npy" % (sys.argv[1], S), img.grad.cpu().numpy())
    sys.exit(0)
for tag, env in (("new", {}), ("old", {"SLN_RBWD_OLD": "1"})):
    subprocess.check_call([sys.executable, os.path.abspath(__file__), tag], env=dict(os.environ, **env))
for S in (256, 96):
    a, b = np.load("/tmp/rbwd_new_%d.npy" % S), np.load("/tmp/rbwd_old_%d.npy" % S)
    print("S=%d: identical %s, max |grad| %.3e, nonzero %.2f" % (S, np.array_equal(a, b), np.abs(a).max(), (a != 0).mean()))
