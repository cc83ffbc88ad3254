"""Known-answer tests that pin the CPU restatement of the rasterizer (oracle/raster_ref.cpp).

The third-party package it restates is absent and un-pinned ("parity unpinned"): these
self-generated cases fix the semantics the HIP kernels are then held to, bit for bit.
"""
import numpy as np
import pytest
import torch

from oracle import raster_ref as rr

IS = 32


def _tri(z=(2.0, 2.0, 2.0)):
    # counter-clockwise in NDC (front-facing by the package's test)
    return np.array([[[-0.6, -0.5, z[0]], [0.7, -0.4, z[1]], [0.1, 0.8, z[2]]]], np.float32)[None]   # [1,1,3,3]


def _inside(tri, xp, yp):
    (x0, y0), (x1, y1), (x2, y2) = tri[0][:2], tri[1][:2], tri[2][:2]
    e = lambda xa, ya, xb, yb: (yp - ya) * (xb - xa) >= (xp - xa) * (yb - ya)
    return e(x0, y0, x1, y1) and e(x1, y1, x2, y2) and e(x2, y2, x0, y0)


def test_single_triangle_coverage_barycentrics_depth():
    f = _tri((2.0, 3.0, 4.0))
    fi, w, d = rr.nmr_forward(f, IS, 0.1, 100.0)
    t = f[0, 0].astype(np.float64)
    n_in = 0
    for yi in range(IS):
        for xi in range(IS):
            xp, yp = (2 * xi + 1 - IS) / IS, (2 * yi + 1 - IS) / IS
            if _inside(t, xp, yp):
                n_in += 1
                assert fi[0, yi, xi] == 0
                ww = w[0, yi, xi].astype(np.float64)
                assert abs(ww.sum() - 1) < 1e-6 and (ww >= 0).all()
                # barycentrics reproduce the pixel centre
                assert abs((ww * t[:, 0]).sum() - xp) < 1e-5 and abs((ww * t[:, 1]).sum() - yp) < 1e-5
                assert abs(d[0, yi, xi] - 1.0 / (ww / t[:, 2]).sum()) < 1e-5
            else:
                assert fi[0, yi, xi] == -1 and d[0, yi, xi] == 100.0
    assert n_in > 100


def test_equal_depth_lowest_index_wins_and_strict_less():
    a = _tri()[0, 0]
    f = np.stack([a, a, a])[None]                      # three identical faces
    fi, _, _ = rr.nmr_forward(f, IS, 0.1, 100.0)
    assert set(np.unique(fi)) == {-1, 0}
    g = f.copy(); g[0, 1, :, 2] = 1.5                   # face 1 strictly nearer
    fi, _, d = rr.nmr_forward(g, IS, 0.1, 100.0)
    assert set(np.unique(fi)) == {-1, 1} and np.isclose(d[fi == 1], 1.5).all()


def test_backface_and_fill_back():
    f = _tri()[:, :, ::-1].copy()                       # clockwise: back-facing
    fi, _, _ = rr.nmr_forward(f, IS, 0.1, 100.0)
    assert (fi == -1).all()
    both = np.concatenate([f, f[:, :, ::-1]], 1)        # what fill_back appends
    fi, _, _ = rr.nmr_forward(both, IS, 0.1, 100.0)
    assert set(np.unique(fi)) == {-1, 1}


def test_near_far_rejection():
    for z, vis in ((0.05, False), (0.2, True), (99.0, True), (100.5, False), (150.0, False)):
        fi, _, _ = rr.nmr_forward(_tri((z, z, z)), IS, 0.1, 100.0)
        assert (fi >= 0).any() == vis, z


def test_renderer_flips_rows_and_depth_mode_uses_library_near():
    K = torch.tensor([[[16.0, 0, 16.0], [0, 16.0, 16.0], [0, 0, 1.0]]])
    R = torch.eye(3)[None]; t = torch.zeros(1, 1, 3)
    # a small triangle above the optical axis (camera y up after v = orig - v)
    v = torch.tensor([[[-0.2, 0.3, 2.0], [0.2, 0.3, 2.0], [0.0, 0.7, 2.0]]])
    faces = torch.tensor([[[0, 1, 2]]], dtype=torch.int32)
    r = rr.RefRenderer(image_size=IS, K=K, R=R, t=t, orig_size=32, near=0.001)
    d = r(v, faces, None, mode='depth')[0].numpy()
    rows = np.where((d < 50).any(1))[0]
    assert len(rows) > 0
    tex = torch.ones(1, 1, 2, 2, 2, 3)
    img = r(v, faces, tex, mode='rgb')[0].numpy()
    assert img.shape == (3, IS, IS)
    rows_rgb = np.where((img[0] > 0.5).any(1))[0]
    assert (rows == rows_rgb).all()
    # image row 0 is the TOP: the triangle (raster rows > centre before the flip) shows in the upper/lower half
    # consistently with flipping the raw map
    fxyz = rr.vertices_to_faces(rr.project(v, K, R, t, 32), torch.cat((faces, faces[:, :, [2, 1, 0]]), 1))
    fi, _, _ = rr.nmr_forward(fxyz.numpy(), IS, 0.1, 100.0)
    raw_rows = np.where((fi[0] >= 0).any(1))[0]
    assert (np.sort(IS - 1 - raw_rows) == rows).all()
    # depth mode ignores the constructor's near (0.001) and uses 0.1: a triangle at z = 0.05 is invisible in depth, visible in rgb
    v2 = v.clone(); v2[..., 2] = 0.05; v2[..., :2] *= 0.025
    assert not (r(v2, faces, None, mode='depth') < 50).any()
    assert (r(v2, faces, tex, mode='rgb') > 0.5).any()


def test_uniform_texture_gives_class_bit():
    f = _tri((2.0, 3.0, 4.0))
    fi, w, d = rr.nmr_forward(f, IS, 0.001, 100.0)
    rgb = rr.nmr_texture_sample(f, np.ones((1, 1, 2, 2, 2, 3), np.float32), fi, w, d)
    assert np.allclose(rgb[fi >= 0], 1.0, atol=1e-6) and (rgb[fi < 0] == 0).all()


def test_depth_backward_matches_finite_differences():
    rng = np.random.default_rng(0)
    f0 = _tri((2.0, 3.0, 4.0)).astype(np.float64)
    gd = rng.standard_normal((1, IS, IS)).astype(np.float32)
    fi, w, d = rr.nmr_forward(f0.astype(np.float32), IS, 0.1, 100.0)
    # interior pixels only (coverage must not change under the perturbation)
    inner = np.zeros_like(fi, bool)
    for yi in range(1, IS - 1):
        for xi in range(1, IS - 1):
            inner[0, yi, xi] = (fi[0, yi - 1:yi + 2, xi - 1:xi + 2] == 0).all()
    gd = gd * inner
    g = rr.nmr_backward_depth(f0.astype(np.float32), fi, w, d, gd)[0, 0]
    h = 1e-3
    for k in range(3):
        for c in range(3):
            fp, fm = f0.copy(), f0.copy()
            fp[0, 0, k, c] += h; fm[0, 0, k, c] -= h
            dp = rr.nmr_forward(fp.astype(np.float32), IS, 0.1, 100.0)[2]
            dm = rr.nmr_forward(fm.astype(np.float32), IS, 0.1, 100.0)[2]
            num = ((dp.astype(np.float64) - dm) * gd).sum() / (2 * h)
            assert abs(num - g[k, c]) <= 2e-2 * max(1.0, abs(num)), (k, c, num, g[k, c])


def test_pixel_map_backward_moves_triangle_towards_bright_target():
    """White triangle on black; the loss wants a pixel just right of the triangle to be white: following the
    negative gradient must move the right-most vertex to the right (+x)."""
    f = _tri()
    fi, w, d = rr.nmr_forward(f, IS, 0.001, 100.0)
    rgb = rr.nmr_texture_sample(f, np.ones((1, 1, 2, 2, 2, 3), np.float32), fi, w, d)
    yi = IS // 2 - 4
    xs = np.where(fi[0, yi] == 0)[0]
    xt = xs.max() + 2
    target = rgb.copy(); target[0, yi, xt] = 1.0
    grad = 2 * (rgb - target)                                     # d/d rgb of sum (rgb - target)^2
    g = rr.nmr_backward_pixel_map(f, fi, rgb, grad)[0, 0]
    assert g[1, 0] < 0                                            # vertex 1 (right-most): loss decreases when x grows
    assert np.abs(g[:, 2]).max() == 0                             # no z gradient from the rgb path


def test_scene_render_layout_and_gradients_flow():
    V, F, ranges, box = rr.synth_room(3, n_objects=4, target_faces=200)
    v = torch.from_numpy(V)[None].requires_grad_(True)
    out = rr.scene_render(v, torch.from_numpy(F)[None], ranges, torch.from_numpy(box), image_size=64)
    assert out.shape == (1, 70, 64, 64)
    wall = out[0, 1 + rr.NYU_CLASS.index('wall')]
    assert (wall > 0.5).any()
    out[:, 41:].sum().backward()
    assert torch.isfinite(v.grad).all() and v.grad.abs().sum() > 0
