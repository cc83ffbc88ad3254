"""Input / output side of ``colorize_with_spade`` (reference testing/test_SPADE_shade.py:16-79): the 41-channel tensor the
generator consumes and the many-z colourisation of one room.

  * ``build_input``  - depth normalisation (:50-55), class-mask stacking / thresholding at 120 (:56-70) and
    ``skimage.transform.resize(total, [256,256], preserve_range=True, order=3, anti_aliasing=True)`` (:73) as device tensor
    ops.  The resize is a fixed linear operator per axis - Gaussian anti-aliasing (sigma (f-1)/2, mirror boundary, 4 sigma
    support), cubic B-spline prefilter (mirror boundary), evaluation at the pixel centres of the coarse grid - so it is built
    once as an [out, in] matrix R in float64 and applied as ``R @ X @ R^T`` to all 41 channels at once.  scikit-image is not
    part of this image: the matrix follows its documented algorithm (scipy.ndimage gaussian_filter + zoom(grid_mode=True))
    and is tested against scipy.ndimage itself.
  * ``colorize``     - ``num_z`` images of one map in ONE generator call (gamma/beta shared, SPADEGenerator4.forward with a
    single-row ``seg``) instead of ``num_z`` batch-1 calls (:74-79); ``to_uint8`` is ``save_color``'s conversion (:16-27).
File reading (.exr / .png through imageio, :45-58) stays with the caller.
"""
import functools

import numpy as np
import torch

NYU40 = ['wall', 'floor', 'cabinet', 'bed', 'chair', 'sofa', 'table', 'door', 'window', 'bookshelf', 'picture',
         'counter', 'blinds', 'desk', 'shelves', 'curtain', 'dresser', 'pillow', 'mirror', 'floor_mat',
         'clothes', 'ceiling', 'books', 'refridgerator', 'television', 'paper', 'towel', 'shower_curtain',
         'box', 'whiteboard', 'person', 'night_stand', 'toilet', 'sink', 'lamp', 'bathtub', 'bag',
         'otherstructure', 'otherfurniture', 'otherprop']


def class_of(basename):
    """class name encoded in a mask file name '<a>_<b>_<c>_<class>[_<class2>].png' (:60-66)"""
    parts = basename.split(".")[0].split("_")
    return parts[3] + "_" + parts[4] if len(parts) == 5 else parts[3]


def _mirror(i, n):
    if n == 1:
        return 0
    p = 2 * (n - 1)
    i = i % p
    return i if i < n else p - i


@functools.lru_cache(maxsize=8)
def resize_matrix(n_in, n_out):
    """[n_out, n_in] float64: anti-aliased cubic-spline resize of one axis (see the module docstring)."""
    f = n_in / n_out
    if n_in == n_out:
        return np.eye(n_in)
    sigma = max(0.0, (f - 1) / 2)
    G = np.eye(n_in)
    if sigma > 0:
        lw = int(4.0 * sigma + 0.5)
        w = np.exp(-0.5 * (np.arange(-lw, lw + 1) / sigma) ** 2); w /= w.sum()
        G = np.zeros((n_in, n_in))
        for i in range(n_in):
            for k, wk in zip(range(-lw, lw + 1), w):
                G[i, _mirror(i + k, n_in)] += wk
    C = np.zeros((n_in, n_in))                                  # cubic B-spline collocation, whole-sample mirror boundary
    for i in range(n_in):
        C[i, i] += 4.0 / 6.0
        C[i, _mirror(i - 1, n_in)] += 1.0 / 6.0
        C[i, _mirror(i + 1, n_in)] += 1.0 / 6.0
    P = np.linalg.inv(C)
    S = np.zeros((n_out, n_in))
    for o in range(n_out):
        x = (o + 0.5) * f - 0.5                                  # grid_mode: pixel centres of the coarse grid
        k = int(np.floor(x)); t = x - k
        wts = [(1 - t) ** 3 / 6, (3 * t ** 3 - 6 * t ** 2 + 4) / 6, (-3 * t ** 3 + 3 * t ** 2 + 3 * t + 1) / 6, t ** 3 / 6]
        for j, wj in zip(range(k - 1, k + 3), wts):
            S[o, _mirror(j, n_in)] += wj
    return S @ P @ G


def normalise_depth(depth):
    d = depth.float()
    d = d - d.min()
    dmax = d[d < 20].max()
    d = d.clamp(0, float(dmax)) / dmax
    return (d - 0.5) * 2


def build_input(depth, masks, size=256, device=None):
    """depth [H,W] (first channel of the .exr), masks {class name: [H,W] tensor 0..255} -> [1,41,size,size] float32."""
    dev = device or depth.device
    depth = depth.to(dev)
    H, W = depth.shape
    total = torch.zeros(41, H, W, dtype=torch.float32, device=dev)
    total[0] = normalise_depth(depth)
    for name, m in masks.items():
        m = m.to(dev).float()
        total[1 + NYU40.index(name)] = torch.where(m < 120, torch.zeros_like(m), torch.where(m > 120, torch.ones_like(m), m))
    Rh = torch.from_numpy(resize_matrix(H, size)).to(dev)
    Rw = torch.from_numpy(resize_matrix(W, size)).to(dev)
    out = torch.matmul(torch.matmul(Rh, total.double()), Rw.t())   # float64 like skimage's internal image
    return out.float()[None].contiguous()


def colorize(model, total, num_z, generator=None):
    """-> [num_z, 3, S, S] in (-1, 1): ``num_z`` z ~ N(0,1) for the ONE map ``total`` [1,41,S,S] (:36-38, :74-79)."""
    z = torch.randn(num_z, model.nz, device=total.device, generator=generator)
    return model(total, z)


def to_uint8(images):
    """save_color's conversion (:16-27): [N,3,S,S] in [-1,1] -> uint8 [N,S,S,3]"""
    a = (images.detach().float() + 1.0) / 2.0
    return (a.permute(0, 2, 3, 1) * 255.0).to(torch.uint8)
