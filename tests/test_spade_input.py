"""SPADE input builder (host/spade_input.py) against the scipy.ndimage restatement of the reference's preprocessing
(oracle/spade_input_ref.py; skimage itself is not in this image - parity unpinned for the resize, see the oracle header)."""
import numpy as np
import torch

from conftest import pkg
from oracle import spade_input_ref as R


def _scene(n=256, seed=0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:n, 0:n] / n
    depth = (2.0 + 3.0 * yy + np.sin(6 * xx) + 0.05 * rng.standard_normal((n, n))).astype(np.float32)
    depth[:8, :8] = 65504.0                                              # background hits in the .exr: the reference clips at max(d[d<20])
    masks = {}
    for name, (y0, x0, h, w) in {"bed": (40, 30, 90, 120), "night_stand": (150, 170, 40, 50), "wall": (0, 0, 256, 40),
                                 "floor_mat": (200, 60, 30, 100)}.items():
        m = np.zeros((n, n), np.float32)
        m[y0 * n // 256:(y0 + h) * n // 256, x0 * n // 256:(x0 + w) * n // 256] = 255
        m[(y0 + 3) * n // 256, x0 * n // 256:(x0 + w) * n // 256] = 120      # exactly 120 stays 120 (neither < nor > 120)
        m += rng.integers(0, 100, size=(n, n)) * (m == 0)                     # anti-aliasing greys below the threshold
        masks[name] = m
    return depth, masks


def test_file_name_rule_and_class_order():
    S = pkg("host.spade_input")
    assert S.NYU40 == R.NYU40 and len(S.NYU40) == 40
    for nm in ("room_000_mask_bed.png", "room_000_mask_night_stand.png", "x_y_z_floor_mat.png"):
        assert S.class_of(nm) == R.class_of(nm)
    assert S.class_of("room_000_mask_night_stand.png") == "night_stand"


def test_resize_matrix_equals_scipy_pipeline():
    S = pkg("host.spade_input")
    rng = np.random.default_rng(1)
    for n_in, n_out in ((64, 16), (96, 32), (60, 20), (40, 40)):
        img = rng.standard_normal((n_in, n_in, 2))
        want = R.resize_skimage(img, [n_out, n_out])
        M = S.resize_matrix(n_in, n_out)
        got = np.einsum("oi,ijc,pj->opc", M, img, M)
        assert np.allclose(got, want, atol=1e-9), (n_in, n_out, np.abs(got - want).max())


def test_build_input_equals_reference_preprocessing():
    S = pkg("host.spade_input")
    depth, masks = _scene(256)
    want = R.build_input(depth, masks, size=64)
    got = S.build_input(torch.from_numpy(depth), {k: torch.from_numpy(v) for k, v in masks.items()}, size=64).numpy()
    assert got.shape == (1, 41, 64, 64) and got.dtype == np.float32
    assert np.allclose(got, want, atol=2e-6), np.abs(got - want).max()
    assert got[0, 0].min() >= -1.05 and got[0, 0].max() <= 1.05 and got[0, 1 + R.NYU40.index("bed")].max() > 0.9


def test_to_uint8_is_save_color():
    S = pkg("host.spade_input")
    img = torch.from_numpy(np.random.default_rng(3).uniform(-1, 1, size=(2, 3, 8, 8)).astype(np.float32))
    got = S.to_uint8(img).numpy()
    for i in range(2):
        assert np.array_equal(got[i], R.save_color_array(img[i].numpy()))
