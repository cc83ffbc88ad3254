"""Time the reference-shaped 33-pass render (one depth pass + one rgb pass per class through nr.Renderer, diff_render.py:359-434)
against the fused scene pass, one room, forward + backward:  python tools/passes_time.py"""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from conftest import pkg
from oracle import raster_ref as rr
DR = pkg("host.diff_render")
V, F, ranges, box = rr.synth_room(1, n_objects=12, target_faces=2000)
f = torch.from_numpy(F)[None].cuda(); room = torch.from_numpy(box).cuda()
NR = pkg("host.neural_renderer")
for name, fn, reuse in (("33 passes, rasterising each", DR.scene_render_passes, False), ("33 passes, maps reused", DR.scene_render_passes, True),
                        ("fused scene pass", DR.scene_render, True)):
    NR.Renderer.reuse_rasterisation = reuse
    def run():
        v = torch.from_numpy(V)[None].cuda().requires_grad_(True)
        out = fn(v, f, ranges, room)
        out.sum().backward()
        return out
    for _ in range(3):
        run()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        out = run()
    torch.cuda.synchronize()
    print("%-32s %7.2f ms per render (forward + backward)" % (name, (time.perf_counter() - t0) / 20 * 1e3))
