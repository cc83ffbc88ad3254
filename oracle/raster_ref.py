"""CPU oracle for the differentiable renderer path.

TEST INFRASTRUCTURE ONLY (see oracle/raster_ref.cpp for the arithmetic and for the statement
"PARITY UNPINNED": the third-party ``neural_renderer`` package is absent and un-pinned, the
reference's call sites are models/diff_render.py:359-361,366,398).

This module wraps the C++ restatement with torch autograd Functions and restates, in plain
torch on the CPU:
  * ``RefRenderer``        - what ``nr.Renderer(camera_mode='projection', ...)(v, f, t, mode=)`` does for
                             mode='depth' / 'rgb' (backward 9-10);
  * ``get_cam_mat``        - models/diff_render.py:13-46;
  * ``scene_render``       - the tensor algebra of ``mesh_render_func`` after the mesh buffers are assembled
                             (models/diff_render.py:344-434): near-plane cull, 1 depth + one rgb pass per class,
                             per-class masks / mean depths / wall_max normalisation, the 70-channel layout.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
F32P, I32P = C.POINTER(C.c_float), C.POINTER(C.c_int32)

NYU_CLASS = ['wall', 'floor', 'cabinet', 'bed', 'chair', 'sofa', 'table', 'door', 'window', 'bookshelf', 'picture',
             'counter', 'blinds', 'desk', 'shelves', 'curtain', 'dresser', 'pillow', 'mirror', 'floor mat', 'clothes',
             'ceiling', 'books', 'refridgerator', 'television', 'paper', 'towel', 'shower curtain', 'box', 'whiteboard',
             'person', 'night stand', 'toilet', 'sink', 'lamp', 'bathtub', 'bag', 'otherstructure', 'otherfurniture',
             'otherprop']            # models/diff_render.py:3 (a data table: the NYU-40 label order)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "build", "libraster_ref.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/build/libraster_ref.so missing: run `make -C oracle`")
        _LIB = C.CDLL(path)
    return _LIB


def _p(a, t=F32P):
    return a.ctypes.data_as(t)


def nmr_forward(faces: np.ndarray, image_size: int, near: float, far: float):
    B, F = faces.shape[0], faces.shape[1]
    faces = np.ascontiguousarray(faces.reshape(B, F, 9), np.float32)
    fi = np.empty((B, image_size, image_size), np.int32)
    w = np.empty((B, image_size, image_size, 3), np.float32)
    d = np.empty((B, image_size, image_size), np.float32)
    lib().nmr_forward(_p(faces), B, F, image_size, C.c_float(near), C.c_float(far), _p(fi, I32P), _p(w), _p(d))
    return fi, w, d


def nmr_texture_sample(faces, textures, fi, w, d, eps=1e-3):
    B, F = faces.shape[0], faces.shape[1]
    is_, ts = fi.shape[1], textures.shape[2]
    faces = np.ascontiguousarray(faces.reshape(B, F, 9), np.float32)
    textures = np.ascontiguousarray(textures, np.float32)
    rgb = np.empty((B, is_, is_, 3), np.float32)
    lib().nmr_texture_sample(_p(faces), _p(textures), _p(fi, I32P), _p(w), _p(d), B, F, is_, ts, C.c_float(eps), _p(rgb))
    return rgb


def nmr_backward_depth(faces, fi, w, d, grad_depth):
    B, F, is_ = faces.shape[0], faces.shape[1], fi.shape[1]
    faces = np.ascontiguousarray(faces.reshape(B, F, 9), np.float32)
    gf = np.zeros((B, F, 9), np.float32)
    lib().nmr_backward_depth(_p(faces), _p(fi, I32P), _p(w), _p(d), _p(np.ascontiguousarray(grad_depth, np.float32)),
                             B, F, is_, _p(gf))
    return gf.reshape(B, F, 3, 3)


def nmr_backward_pixel_map(faces, fi, rgb, grad_rgb, eps=1e-3):
    B, F, is_, Cn = faces.shape[0], faces.shape[1], fi.shape[1], rgb.shape[-1]
    faces = np.ascontiguousarray(faces.reshape(B, F, 9), np.float32)
    gf = np.zeros((B, F, 9), np.float32)
    lib().nmr_backward_pixel_map(_p(faces), _p(fi, I32P), _p(np.ascontiguousarray(rgb, np.float32)),
                                 _p(np.ascontiguousarray(grad_rgb, np.float32)), B, F, is_, Cn, C.c_float(eps), _p(gf))
    return gf.reshape(B, F, 3, 3)


# ----------------------------------------------------------------------------------------------
# torch wrappers
# ----------------------------------------------------------------------------------------------
class _RasterDepth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, faces, image_size, near, far):
        f = faces.detach().numpy()
        fi, w, d = nmr_forward(f, image_size, near, far)
        ctx.save_for_backward(faces)
        ctx.maps = (fi, w, d)
        return torch.from_numpy(d.copy())

    @staticmethod
    def backward(ctx, gd):
        (faces,) = ctx.saved_tensors
        fi, w, d = ctx.maps
        return torch.from_numpy(nmr_backward_depth(faces.detach().numpy(), fi, w, d, gd.contiguous().numpy())), None, None, None


class _RasterRgb(torch.autograd.Function):
    @staticmethod
    def forward(ctx, faces, textures, image_size, near, far, eps):
        f = faces.detach().numpy()
        fi, w, d = nmr_forward(f, image_size, near, far)
        rgb = nmr_texture_sample(f, textures.detach().numpy(), fi, w, d, eps)
        ctx.save_for_backward(faces)
        ctx.maps, ctx.eps = (fi, rgb), eps
        return torch.from_numpy(rgb.copy())

    @staticmethod
    def backward(ctx, grgb):
        (faces,) = ctx.saved_tensors
        fi, rgb = ctx.maps
        g = nmr_backward_pixel_map(faces.detach().numpy(), fi, rgb, grgb.contiguous().numpy(), ctx.eps)
        return torch.from_numpy(g), None, None, None, None, None


def project(vertices, K, R, t, orig_size, eps=1e-9):
    """Camera projection with the README patch applied (no lens distortion), SURVEY.md App. B step 2."""
    v = torch.matmul(vertices, R.transpose(2, 1)) + t
    x, y, z = v[:, :, 0], v[:, :, 1], v[:, :, 2]
    x_, y_ = x / (z + eps), y / (z + eps)
    h = torch.stack([x_, y_, torch.ones_like(z)], dim=-1)
    h = torch.matmul(h, K.transpose(1, 2))
    u, vv = h[:, :, 0], h[:, :, 1]
    vv = orig_size - vv
    u = 2 * (u - orig_size / 2.) / orig_size
    vv = 2 * (vv - orig_size / 2.) / orig_size
    return torch.stack([u, vv, z], dim=-1)


def vertices_to_faces(vertices, faces):
    B, V = vertices.shape[:2]
    idx = faces.long() + (torch.arange(B, dtype=torch.int64) * V)[:, None, None]
    return vertices.reshape(B * V, 3)[idx]                     # [B, F, 3, 3]


class RefRenderer:
    """The subset of nr.Renderer the reference uses (models/diff_render.py:359-361)."""

    def __init__(self, camera_mode='projection', image_size=256, K=None, R=None, t=None, anti_aliasing=False,
                 orig_size=512, near=0.001, far=100.0, light_intensity_ambient=1.0, light_intensity_directional=0.0,
                 fill_back=True, rasterizer_eps=1e-3):
        assert camera_mode == 'projection' and not anti_aliasing
        self.image_size, self.K, self.R, self.t = image_size, K, R, t
        self.orig_size, self.near, self.far, self.fill_back, self.eps = orig_size, near, far, fill_back, rasterizer_eps
        self.ambient, self.directional = light_intensity_ambient, light_intensity_directional

    def __call__(self, vertices, faces, textures=None, mode=None):
        if self.fill_back:
            faces = torch.cat((faces, faces[:, :, [2, 1, 0]]), dim=1)
        v = project(vertices, self.K, self.R, self.t, self.orig_size)
        fxyz = vertices_to_faces(v, faces)
        if mode == 'depth':
            # the package's render_depth does not forward near/far: library defaults 0.1 / 100 apply
            d = _RasterDepth.apply(fxyz, self.image_size, 0.1, 100.0)
            return torch.flip(d, dims=[1])
        if mode == 'rgb':
            if self.fill_back:
                textures = torch.cat((textures, textures.permute((0, 1, 4, 3, 2, 5))), dim=1)
            textures = textures * (self.ambient + 0.0 * self.directional)   # ambient-only lighting (= identity here)
            rgb = _RasterRgb.apply(fxyz, textures, self.image_size, self.near, self.far, self.eps)
            return torch.flip(rgb.permute(0, 3, 1, 2), dims=[2])
        raise ValueError(mode)


def get_cam_mat(room_box):
    """models/diff_render.py:13-46; ``room_box`` = boxes[-1] (6 floats)."""
    theta, fl, inter = -0.4, 400, 512
    K = torch.tensor([[fl * inter / 1024, 0, inter / 2.0], [0, fl * inter / 1024, inter / 2.0], [0, 0, 1.0]],
                     dtype=torch.float32)[None]
    w2c = torch.tensor([[1, 0, 0], [0, np.cos(theta), np.sin(theta)], [0, -np.sin(theta), np.cos(theta)]], dtype=torch.float32)
    cam = torch.zeros(3, 1)
    cam[0, 0] = room_box[3] / 2.0
    cam[1, 0] = room_box[4] / 2.0 + min(0.1, abs(float(room_box[4]) / 2.0))
    cam[2, 0] = room_box[5]
    t_w2c = torch.matmul(w2c, -cam)
    c2cv = torch.tensor([[1, 0, 0], [0, -1, 0], [0, 0, -1]], dtype=torch.float32)
    return K, torch.matmul(c2cv, w2c).reshape(1, 3, 3), torch.matmul(c2cv, t_w2c).reshape(1, 1, 3)


def scene_render(vertices_buf, face_buf, class_ranges, room_box, image_size=256):
    """models/diff_render.py:344-434 on already assembled buffers.

    vertices_buf [1,V,3] (may require grad), face_buf [1,F,3] int32, class_ranges {class name: [[a,b],...]} of
    face index ranges BEFORE culling (``model_idx_buffer``).  Returns final [1,70,is,is].
    """
    K, R, t = get_cam_mat(room_box)
    eps = 0.06
    cam = torch.matmul(vertices_buf, R.transpose(1, 2)) + t
    zc = cam[:, :, 2]
    fz = zc[:, face_buf.long()][0]                        # [1,F,3]
    F_old = face_buf.shape[1]
    valid = ~torch.any(fz < eps, dim=2)
    face_buf = face_buf[:, valid[0], :].detach()
    rend = RefRenderer(image_size=image_size, K=K, R=R, t=t, orig_size=512, near=0.001)
    tex = torch.ones(1, face_buf.shape[1], 2, 2, 2, 3)
    depth = rend(vertices_buf, face_buf, tex, mode='depth')
    depth[depth > 15] = -1
    one_hot = torch.zeros(41, image_size, image_size)
    classes = sorted(set(class_ranges.keys()))
    classes.remove("wall"); classes.insert(0, "wall")
    depth_hot = torch.zeros(len(classes) - 3, image_size, image_size)
    ci, wall_max = 0, None
    for name in classes:
        t_un = torch.zeros(1, F_old, 2, 2, 2, 3)
        for a, b in class_ranges[name]:
            t_un[:, a:b] = 1.0
        tex = t_un[:, valid[0]]
        images = rend(vertices_buf, face_buf, tex, mode='rgb')
        image = torch.sum(images, dim=1, keepdim=True)[0] / 3.0
        mask = image.detach() > 0.1
        csd = torch.zeros_like(depth)
        mean = torch.mean(depth[mask])
        if name == "wall":
            wall_max = torch.max(depth[mask]).detach() if mask.any() else torch.tensor(float('nan'))
            if torch.isnan(wall_max):
                wall_max = 10.0
        if torch.isnan(mean):
            mean = wall_max
        csd[~mask] = mean / wall_max
        csd[mask] = depth[mask] / wall_max
        if name not in ("wall", "floor", "ceiling"):
            depth_hot[ci] = csd
            ci += 1
        one_hot[NYU_CLASS.index(name.replace("_", " ")) + 1] = image
    return torch.cat((depth, one_hot[1:], depth_hot), dim=0)[None]


# ----------------------------------------------------------------------------------------------
# synthetic rooms: cuboids + room shell, every quad split into a grid of triangles
# ----------------------------------------------------------------------------------------------
def _quad(p0, du, dv, n):
    """n x n grid of the parallelogram p0 + a*du + b*dv -> (vertices [(n+1)^2,3], faces [2n^2,3])."""
    a = np.linspace(0, 1, n + 1)
    g = p0[None, None] + a[:, None, None] * du[None, None] + a[None, :, None] * dv[None, None]
    v = g.reshape(-1, 3)
    idx = np.arange((n + 1) * (n + 1)).reshape(n + 1, n + 1)
    q = np.stack([idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]], -1).reshape(-1, 4)
    f = np.concatenate([q[:, [0, 1, 2]], q[:, [0, 2, 3]]], 0)
    return v, f


def _cuboid(lo, hi, n):
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    d = hi - lo
    ex, ey, ez = np.array([d[0], 0, 0]), np.array([0, d[1], 0]), np.array([0, 0, d[2]])
    quads = [(lo, ey, ex), (lo + ez, ex, ey), (lo, ex, ez), (lo + ey, ez, ex), (lo, ez, ey), (lo + ex, ey, ez)]
    vs, fs, off = [], [], 0
    for p0, du, dv in quads:
        v, f = _quad(p0, du, dv, n)
        vs.append(v); fs.append(f + off); off += v.shape[0]
    return np.concatenate(vs), np.concatenate(fs)


FURNITURE = ['cabinet', 'bed', 'chair', 'sofa', 'table', 'bookshelf', 'desk', 'shelves', 'dresser', 'night_stand',
             'television', 'lamp', 'toilet', 'sink', 'bathtub', 'counter', 'refridgerator', 'mirror', 'picture', 'box',
             'bag', 'books', 'clothes', 'pillow', 'towel', 'paper', 'whiteboard', 'otherprop', 'otherfurniture']


def synth_room(seed, n_objects=12, target_faces=2000, room=(4.0, 2.7, 5.0)):
    """Procedural room: `n_objects` cuboids on the floor + floor / ceiling / 3 walls, ~target_faces triangles.
    Returns vertices [V,3] float32, faces [F,3] int32, class_ranges (all 29 furniture classes present as keys,
    possibly empty, plus wall/floor/ceiling), room_box (6,)."""
    rng = np.random.default_rng(seed)
    room = np.asarray(room, np.float64)
    ranges = {c: [] for c in FURNITURE}
    ranges.update(wall=[], floor=[], ceiling=[])
    vs, fs, voff, foff = [], [], 0, 0

    def add(v, f, name):
        nonlocal voff, foff
        vs.append(v); fs.append(f + voff)
        ranges[name].append([foff, foff + f.shape[0]])
        voff += v.shape[0]; foff += f.shape[0]
    n_obj = 2 if target_faces >= 1500 else 1
    names = rng.choice(FURNITURE, size=n_objects, replace=False)
    for nm in names:
        size = rng.uniform([0.4, 0.3, 0.4], [1.2, 1.6, 1.2])
        pos = rng.uniform([0.1, 0.0, 0.3], [room[0] - size[0] - 0.1, 0.0, room[2] - size[2] - 1.2])
        v, f = _cuboid(pos, pos + size, n_obj)
        add(v, f, str(nm))
    used = foff
    shell = [("floor", np.zeros(3), np.array([0, 0, room[2]]), np.array([room[0], 0, 0])),
             ("ceiling", np.array([0, room[1], 0]), np.array([room[0], 0, 0]), np.array([0, 0, room[2]])),
             ("wall", np.zeros(3), np.array([room[0], 0, 0]), np.array([0, room[1], 0])),
             ("wall", np.zeros(3), np.array([0, room[1], 0]), np.array([0, 0, room[2]])),
             ("wall", np.array([room[0], 0, 0]), np.array([0, 0, room[2]]), np.array([0, room[1], 0]))]
    per = max(1, int(round(np.sqrt(max(target_faces - used, 10) / (2.0 * len(shell))))))
    for nm, p0, du, dv in shell:
        v, f = _quad(p0, du, dv, per)
        add(v, f, nm)
    V = np.concatenate(vs).astype(np.float32)
    Fc = np.concatenate(fs).astype(np.int32)
    return V, Fc, ranges, np.array([0, 0, 0, room[0], room[1], room[2]], np.float32)
