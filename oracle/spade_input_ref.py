"""CPU oracle for the SPADE input builder.

TEST INFRASTRUCTURE ONLY.  numpy / scipy restatement of the array work in ``colorize_with_spade``
(/root/reference/testing/test_SPADE_shade.py:50-76): depth normalisation, class-mask thresholding, channel stacking and
``skimage.transform.resize(total, [256, 256], preserve_range=True, order=3, anti_aliasing=True)``.

Parity status: the numpy steps are the reference's own lines; the resize is PARITY UNPINNED - scikit-image is not installed
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
