"""SPADE input builder (host/spade_input.py) against the scipy.ndimage restatement of the reference's preprocessing
(oracle/spade_input_ref.py; skimage itself is not in this image - parity unpinned for the resize, see the oracle header) and,
for everything but the resize, against what the reference's own `colorize_with_spade` / `save_color` produced when executed from
their source (oracle/gen_golden_sampling.py::gen_spade_input -> tests/golden/spade_input.npz)."""
import numpy as np
import torch

from conftest import load_golden, pkg
from oracle import spade_input_ref as R


_scene = R.synth_scene


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


def test_oracle_and_product_reproduce_the_reference_function():
    """`colorize_with_spade` (testing/test_SPADE_shade.py:30-79) executed from its source on the 1 024^2 scene: the [1,41,256,256] tensor
    it hands the generator (every second pixel of the live channels + per-channel sums) and `save_color`'s uint8 image.  The oracle
    must reproduce it exactly (same numpy lines, same scipy resize); the product's matrix form of the resize to 2e-6."""
    S = pkg("host.spade_input")
    g = load_golden("spade_input")
    depth, masks = R.synth_scene(1024, seed=2)
    want_half, chans = g["total_half"], g["total_channels"]
    ora = R.build_input(depth, masks, size=256)[0]
    assert np.array_equal(ora[chans][:, ::2, ::2], want_half)
    assert np.allclose(ora.astype(np.float64).reshape(41, -1).sum(1), g["total_sums"], rtol=0, atol=1e-9)
    got = S.build_input(torch.from_numpy(depth), {k: torch.from_numpy(v) for k, v in masks.items()}, size=256)[0].numpy()
    assert np.abs(got[chans][:, ::2, ::2] - want_half).max() <= 2e-6
    assert np.abs(got.astype(np.float64).reshape(41, -1).sum(1) - g["total_sums"]).max() <= 2e-6 * 256 * 256
    assert sorted(chans.tolist()) == sorted([0] + [1 + R.NYU40.index(k) for k in masks])
    img = torch.from_numpy(g["save_color_in"])
    assert np.array_equal(S.to_uint8(img).numpy()[0], g["save_color_out"]) and np.array_equal(R.save_color_array(img[0].numpy()), g["save_color_out"])
