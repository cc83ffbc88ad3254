"""Lab: where a working wavefront of pixel_map_backward_kernel<PixClass> spends its clocks (16 bench rooms).  Needs the -DPMB_STAMP build
of raster.hip as tools/lab/lib_stamp.so (see LAB_NOTES 9a-2); run on the GPU box:
    cp tools/lab/lib_stamp.so 3d_sln_amd/libsln_hip.so && SLN_SCENE_NO_SIDE=1 python tools/lab/pmb_stamps.py"""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
L = importlib.import_module("3d_sln_amd._lib")
DR = importlib.import_module("3d_sln_amd.host.diff_render")
syn = importlib.import_module("3d_sln_amd.host.synthetic")

R = int(os.environ.get("ROOMS", 16))
rooms = [syn.synthetic_room(100 + i, n_objects=12, target_faces=2000) for i in range(R)]
pk = syn.pack_rooms(rooms, "cuda")
Vb = pk["V"].requires_grad_(True)
gout = torch.randn(R, 70, 256, 256, device="cuda")


def it():
    Vb.grad = None
    out = DR.scene_render_batch(Vb, pk["F"], pk["C"], pk["chan"], pk["dch"], pk["K"], pk["R"], pk["t"], 256, 0.001)
    out.backward(gout)


for _ in range(5):
    it()
torch.cuda.synchronize()
lab = L.lib().sln_lab_pmb_stamps
lab.restype = C.c_int
lab.argtypes = [C.c_void_p, C.c_int]
assert lab(None, 1) == 0
it()
torch.cuda.synchronize()
buf = np.zeros((1 << 19, 8), dtype=np.uint64)
assert lab(buf.ctypes.data, 0) == 0
w = buf[buf[:, 0] > 0].astype(np.float64)          # workgroups that reached the end (front-facing owners with a d0 range or not)
print("workgroups that wrote a slot: %d" % len(w))
names = ["total", "prologue", "phase 1", "2a wait", "2a eval", "2a passes", "2b", "2a rows"]
tot = w.sum(0)
for k, n in enumerate(names):
    print("  %-10s sum %14.0f  share of total %.3f  mean per workgroup %9.1f" % (n, tot[k], tot[k] / tot[0], w[:, k].mean()))
print("clocks per 2a pass: wait %.0f, eval %.0f; passes per row %.2f; other 2a clocks per row (total - listed) %.0f" %
      (tot[3] / tot[5], tot[4] / tot[5], tot[5] / tot[7], (tot[0] - tot[1] - tot[2] - tot[3] - tot[4] - tot[6]) / max(tot[7], 1)))
for q in (10, 50, 90, 99):
    print("  p%d total %.0f clocks" % (q, np.percentile(w[:, 0], q)))
