"""CPU oracle for the SPADE input builder.

TEST INFRASTRUCTURE ONLY.  numpy / scipy restatement of the array work in ``colorize_with_spade``
(/root/reference/testing/test_SPADE_shade.py:50-76): depth normalisation, class-mask thresholding, channel stacking and
``skimage.transform.resize(total, [256, 256], preserve_range=True, order=3, anti_aliasing=True)``.

Parity status: the numpy steps are PINNED (round 6): oracle/gen_golden_sampling.py::gen_spade_input executes the reference's
``colorize_with_spade`` and ``save_color`` from their source text (imageio replaced by arrays, the generator by a recorder) and
tests/test_spade_input.py holds ``build_input`` / ``save_color_array`` to what they produced; the resize is PARITY UNPINNED - scikit-image is not installed
in this image (and the reference pins no version), so its documented algorithm (scikit-image >= 0.19, transform/_warps.py
``resize``: ``ndi.gaussian_filter(image, sigma=(factor-1)/2 per resized axis, mode='mirror')`` followed by
``ndi.zoom(..., order=3, mode='mirror', grid_mode=True)``; ``mode='reflect'`` of skimage maps to scipy's 'mirror') is
restated with scipy.ndimage, which IS what scikit-image calls.
"""
import numpy as np
from scipy import ndimage as ndi

NYU40 = ['wall', 'floor', 'cabinet', 'bed', 'chair', 'sofa', 'table', 'door', 'window', 'bookshelf', 'picture',
         'counter', 'blinds', 'desk', 'shelves', 'curtain', 'dresser', 'pillow', 'mirror', 'floor_mat',
         'clothes', 'ceiling', 'books', 'refridgerator', 'television', 'paper', 'towel', 'shower_curtain',
         'box', 'whiteboard', 'person', 'night_stand', 'toilet', 'sink', 'lamp', 'bathtub', 'bag',
         'otherstructure', 'otherfurniture', 'otherprop']


def class_of(basename):
    """test_SPADE_shade.py:60-66: '<a>_<b>_<c>_<class>[_<class2>].png'"""
    parts = basename.split(".")[0].split("_")
    return parts[3] + "_" + parts[4] if len(parts) == 5 else parts[3]


def resize_skimage(total_hwc, out_hw):
    """skimage.transform.resize(total, out_hw, preserve_range=True, order=3, anti_aliasing=True) for an [H,W,C] array."""
    img = np.asarray(total_hwc, dtype=np.float64)
    factors = np.array([img.shape[0] / out_hw[0], img.shape[1] / out_hw[1], 1.0])
    sigma = np.maximum(0, (factors - 1) / 2)
    filtered = ndi.gaussian_filter(img, sigma, cval=0, mode="mirror")
    return ndi.zoom(filtered, 1.0 / factors, order=3, mode="mirror", cval=0, grid_mode=True)


def build_input(depth, masks, size=256):
    """depth [H,W] float (first channel of the .exr), masks {class name: [H,W] 0..255} -> [1,41,size,size] float32."""
    d = np.asarray(depth, dtype=np.float32)
    d = d - np.min(d)
    dmax = np.max(d[d < 20])
    d = np.clip(d, 0, dmax) / dmax
    d = ((d - 0.5) * 2).astype("float32")[None, :]
    buf = np.zeros((40,) + d.shape[1:])
    for name, m in masks.items():
        buf[NYU40.index(name)] = m
    buf = buf.astype("float32")
    buf[buf < 120] = 0.0
    buf[buf > 120] = 1.0
    total = np.vstack([d, buf])
    total = np.moveaxis(total, 0, 2)
    total = resize_skimage(total, [size, size])
    return np.moveaxis(total, 2, 0)[None, :].astype(np.float32)


def save_color_array(img):
    """save_color (test_SPADE_shade.py:16-27): [-1,1] CHW -> uint8 HWC"""
    a = (np.asarray(img, dtype=np.float32) + 1.0) / 2.0
    return (a.transpose((1, 2, 0)) * 255.0).astype(np.uint8)


def synth_scene(n=256, seed=0):
    """a procedural (depth [n,n] float32, {class: mask [n,n] 0..255}) pair: a tilted floor with background hits (the .exr's 65504), four
    class masks with anti-aliasing greys below the threshold and one row exactly AT it - shared by tests/test_spade_input.py and
    oracle/gen_golden_sampling.py::gen_spade_input"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:n, 0:n] / n
    depth = (2.0 + 3.0 * yy + np.sin(6 * xx) + 0.05 * rng.standard_normal((n, n))).astype(np.float32)
    depth[:8 * n // 256, :8 * n // 256] = 65504.0                       # background hits: the reference clips at max(d[d < 20])
    masks = {}
    for name, (y0, x0, h, w) in {"bed": (40, 30, 90, 120), "night_stand": (150, 170, 40, 50), "wall": (0, 0, 256, 40),
                                 "floor_mat": (200, 60, 30, 100)}.items():
        m = np.zeros((n, n), np.float32)
        m[y0 * n // 256:(y0 + h) * n // 256, x0 * n // 256:(x0 + w) * n // 256] = 255
        m[(y0 + 3) * n // 256, x0 * n // 256:(x0 + w) * n // 256] = 120      # exactly 120 stays 120 (neither < nor > 120)
        m += rng.integers(0, 100, size=(n, n)) * (m == 0)                     # anti-aliasing greys below the threshold
        masks[name] = m
    return depth, masks
