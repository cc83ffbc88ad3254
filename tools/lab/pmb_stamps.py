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
buf = np.zeros((1 << 19, 10), dtype=np.uint64)
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

# where in the launch the long units sit: block id -> (XCD, image, edge x axis, face) as the kernel maps it (B >= 8)
F = int(round(len(buf[:, 0].nonzero()[0]) and (pk["F"].shape[1] * 2)))      # faces after fill_back (the kernel's F) - printed for checking
lin = np.nonzero(buf[:, 0] > 0)[0]
Fk = int(os.environ.get("PMB_F", 0)) or F
xcd, j = lin & 7, lin >> 3
img_local, rem = j // (6 * Fk), j % (6 * Fk)
fn, ea = rem // 6, rem % 6            # face-major (round 6; PMB_EA_MAJOR builds: ea, fn = rem // Fk, rem % Fk)
tot_c = buf[lin, 0].astype(np.float64)
w0, w1 = buf[lin, 8].astype(np.float64), buf[lin, 9].astype(np.float64)
t0 = w0.min()
print("F = %d; launch spans %.1f us by the 100 MHz clock (first start to last end)" % (Fk, (w1.max() - t0) / 100.0))
edges = np.linspace(0, (w1.max() - t0), 21)
alive = [int(((w0 - t0 <= e) & (w1 - t0 > e)).sum()) for e in edges]
print("working wavefronts alive at 0, 5, ... 100 %% of the launch: %s" % alive)
late = (w1 - t0) > 0.8 * (w1.max() - t0)
print("units still running in the last 20 %%: %d; their start (%% of launch) p10/p50/p90: %s; their length us p50/p90/max: %s" % (
    late.sum(), np.percentile((w0[late] - t0) / (w1.max() - t0) * 100, [10, 50, 90]).round(1), np.percentile((w1[late] - w0[late]) / 100.0, [50, 90, 100]).round(1)))
dec = (fn * 10 // Fk)
print("share of the working clocks by face decile (face order of the batch):", [round(float(tot_c[dec == d].sum() / tot_c.sum()), 3) for d in range(10)])
print("share by edge x axis:", [round(float(tot_c[ea == k].sum() / tot_c.sum()), 3) for k in range(6)])
big = np.argsort(-tot_c)[:12]
print("longest units (image slot, edge x axis, face, us, start %%):", [(int(xcd[i] + 8 * img_local[i]), int(ea[i]), int(fn[i]), round((w1[i] - w0[i]) / 100.0, 1), round(float((w0[i] - t0) / (w1.max() - t0) * 100), 1)) for i in big])
