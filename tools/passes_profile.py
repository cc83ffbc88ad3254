"""Where the 33-pass render spends its time: forward and backward wall time (host + device, synchronised) per variant.
   python tools/passes_profile.py [reuse|noreuse|both]      (run under rocprofv3 --kernel-trace --stats for the device side)"""
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from conftest import pkg
DR = pkg("host.diff_render"); NR = pkg("host.neural_renderer"); syn = pkg("host.synthetic")
V, F, ranges, box = syn.synthetic_room(1, n_objects=12, target_faces=2000)
f = torch.from_numpy(F)[None].cuda(); room = torch.from_numpy(box).cuda()
which = sys.argv[1] if len(sys.argv) > 1 else "both"
for name, reuse in (("maps reused", True), ("rasterising each pass", False)):
    if which != "both" and (which == "reuse") != reuse:
        continue
    NR.Renderer.reuse_rasterisation = reuse
    tf = tb = 0.0
    n = 10
    for it in range(3 + n):
        v = torch.from_numpy(V)[None].cuda().requires_grad_(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = DR.scene_render_passes(v, f, ranges, room)
        s = out.sum()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        s.backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        if it >= 3:
            tf += t1 - t0; tb += t2 - t1
    print("%-24s forward %6.2f ms   backward %6.2f ms" % (name, tf / n * 1e3, tb / n * 1e3))

# floor: the caller's own algebra (diff_render.py:381-431 as restated in scene_render_passes) with the Renderer replaced by a stub
# that returns precomputed images (attached to the vertices so that backward still runs through the caller's graph)
if which in ("both", "floor"):
    NR.Renderer.reuse_rasterisation = True
    v0 = torch.from_numpy(V)[None].cuda().requires_grad_(True)
    cache = {}
    real = NR.Renderer.render

    def rec(self, vertices, faces, textures=None, mode=None, *a, **k):
        out = real(self, vertices, faces, textures, mode, *a, **k)
        cache.setdefault(mode, []).append(out.detach())
        return out
    NR.Renderer.render = rec
    DR.scene_render_passes(v0, f, ranges, room)
    counters = {}

    def stub(self, vertices, faces, textures=None, mode=None, *a, **k):
        i = counters.get(mode, 0); counters[mode] = i + 1
        return cache[mode][i % len(cache[mode])] + 0.0 * vertices.sum()
    NR.Renderer.render = stub
    tf = tb = 0.0
    n = 10
    for it in range(3 + n):
        counters.clear()
        v = torch.from_numpy(V)[None].cuda().requires_grad_(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = DR.scene_render_passes(v, f, ranges, room)
        s = out.sum()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        s.backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        if it >= 3:
            tf += t1 - t0; tb += t2 - t1
    NR.Renderer.render = real
    print("%-24s forward %6.2f ms   backward %6.2f ms" % ("caller's algebra only", tf / n * 1e3, tb / n * 1e3))
