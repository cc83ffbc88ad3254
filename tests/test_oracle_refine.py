"""Known-answer tests of oracle/refine_ref.py (CPU): the restated object loop of mesh_render_func and the refinement loss on cases
whose answer follows from the cited statements by hand.  (tests/test_oracle_refine_golden.py holds the same functions to fixtures made
by executing the reference's own source.)"""
import math

import numpy as np
import torch

from oracle import refine_ref as rf


def _cube():
    v = torch.tensor([[x, y, z] for x in (0.0, 2.0) for y in (0.0, 1.0) for z in (0.0, 4.0)], dtype=torch.float64)
    return v, v.min(0).values, v.max(0).values


def test_identity_placement_and_isotropic_min_scale():
    v, lo, hi = _cube()
    ext = torch.tensor([10.0, 10.0, 10.0], dtype=torch.float64)
    box = torch.cat([lo, hi]) / 10.0                                   # the model's own bounding box, room-normalised
    out, size = rf.place_object(box, torch.tensor(0.0, dtype=torch.float64), ext, v, lo, hi)
    assert torch.allclose(out, v, atol=1e-12) and torch.allclose(size, hi - lo)
    # a box twice as wide in x only: the scale stays min(ratios) = 1, the model is centred in the box (diff_render.py:117,126)
    box2 = box.clone(); box2[3] = 0.4
    out2, _ = rf.place_object(box2, torch.tensor(0.0, dtype=torch.float64), ext, v, lo, hi)
    assert torch.allclose(out2[:, 0], v[:, 0] + 1.0) and torch.allclose(out2[:, 1:], v[:, 1:])
    # half the size in every direction: scale 0.5 about the box centre
    box3 = torch.cat([lo, lo + (hi - lo) / 2]) / 10.0
    out3, _ = rf.place_object(box3, torch.tensor(0.0, dtype=torch.float64), ext, v, lo, hi)
    assert torch.allclose(out3, v / 2, atol=1e-12)


def test_rotation_by_six_bins_is_minus_ninety_degrees_about_y():
    v, lo, hi = _cube()
    ext = torch.ones(3, dtype=torch.float64)
    box = torch.cat([lo, hi])
    out, _ = rf.place_object(box, torch.tensor(6.0, dtype=torch.float64), ext, v, lo, hi)       # theta = -6 * 2 pi / 24 = -pi / 2
    c = (lo + hi) / 2
    d = v - c
    # R_y(theta) = [[cos, 0, sin], [0, 1, 0], [-sin, 0, cos]] with theta = -pi/2:  x' = -z, z' = x
    exp = torch.stack([-d[:, 2], d[:, 1], d[:, 0]], 1) + c
    assert torch.allclose(out, exp, atol=1e-12)


def test_scene_loop_skips_non_furniture_and_sums_the_size_loss():
    v, lo, hi = _cube()
    models = {"bed": dict(v=v, bbox_min=lo, bbox_max=hi), "door": dict(v=v, bbox_min=lo, bbox_max=hi)}
    boxes = torch.tensor([[0.0, 0, 0, 0.2, 0.1, 0.4], [0.1, 0, 0, 0.3, 0.1, 0.4], [0, 0, 0, 10.0, 10.0, 10.0]], dtype=torch.float64)
    angles = torch.zeros(3, dtype=torch.float64)
    verts, sizes, sl = rf.place_scene(boxes, angles, ["bed", "door", "__room__"], models, [torch.tensor([2.0, 1.0, 3.0], dtype=torch.float64)])
    assert verts.shape == (8, 3) and len(sizes) == 1                   # 'door' is skipped (diff_render.py:93-97)
    assert abs(float(sl) - (0 + 0 + 1.0) / 3) < 1e-12                  # mse over the three size components


def test_pooling_of_constant_maps_and_loss_of_identical_images():
    x = torch.zeros(1, 70, 64, 64, dtype=torch.float64)
    x[:, 3] = 1.0                                                      # one class everywhere
    x[:, 41:] = 0.25
    pooled = rf.psp_pool(x[:, 41:])
    assert pooled.shape == (1, 4 * 29, 96, 96) and torch.allclose(pooled, torch.full_like(pooled, 0.25))
    labels = rf.target_labels(x)
    assert all((l == 2).all() for l in labels)                         # channel 3 of the image = class 2 of the 40 one-hot planes
    loss, depth, sem = rf.refinement_loss(x.clone(), x, labels, torch.zeros((), dtype=torch.float64))
    assert float(depth) == 0.0                                         # sum of the depth planes is 7.25 >= 0.5: no null fill
    # 40 logits, one of them 1, the others 0: CE = log(e + 39) - 1, four scales, / 800
    assert abs(float(sem) - 4 * (math.log(math.e + 39) - 1) / 800) < 1e-12
    assert abs(float(loss) - 100 * float(sem)) < 1e-12


def test_null_fill_and_ignored_labels():
    t = torch.zeros(1, 70, 64, 64, dtype=torch.float64)                # empty target: every label is ignored
    labels = rf.target_labels(t)
    assert all((l == -100).all() for l in labels)
    x = torch.zeros(1, 70, 64, 64, dtype=torch.float64)
    x[:, 1] = 1.0
    # iterate: depth planes all 0 -> the last one is set to 1 everywhere (test_render_refine.py:329); target stays 0
    _, depth, sem = rf.refinement_loss(x, t, labels, torch.zeros((), dtype=torch.float64))
    assert abs(float(depth) - 0.5 * (1.0 / 29)) < 1e-12                # L1 mean: one of 29 planes differs by 1, * 0.5
    assert np.isnan(float(sem))                                        # CrossEntropyLoss over zero valid targets is nan in torch
