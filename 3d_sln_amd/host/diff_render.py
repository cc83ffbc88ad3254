"""HIP-backed counterpart of the reference's ``models/diff_render.py`` (layout-refinement renderer).

``get_cam_mat`` (diff_render.py:13-46) and the tensor algebra of ``mesh_render_func`` after the mesh buffers
exist (:344-434) are reproduced here; mesh retrieval / OBJ loading (models/misc.py, licensed SUNCG data,
pywavefront, pymesh) is OUT OF SCOPE, so the scene arrives as ``vertices_buf`` / ``face_buf`` plus the
per-class face ranges (``model_idx_buffer``) - exactly the seam at diff_render.py:344.

``scene_render`` is the MI355X-first formulation: the reference's 33 raster passes over identical geometry
become ONE fused pass (csrc/raster.hip, sln_scene_forward/backward); ``scene_render_passes`` keeps the
reference's pass structure on top of ``neural_renderer.Renderer`` (used to cross-check the fused path).
"""
import ctypes as C

import numpy as np
import torch

from .. import _lib
from . import neural_renderer as nr

nyu_class = ['wall', 'floor', 'cabinet', 'bed', 'chair', 'sofa', 'table', 'door', 'window', 'bookshelf', 'picture', 'counter',
             'blinds', 'desk', 'shelves', 'curtain', 'dresser', 'pillow', 'mirror', 'floor mat', 'clothes', 'ceiling', 'books',
             'refridgerator', 'television', 'paper', 'towel', 'shower curtain', 'box', 'whiteboard', 'person', 'night stand',
             'toilet', 'sink', 'lamp', 'bathtub', 'bag', 'otherstructure', 'otherfurniture', 'otherprop']
inter_out = 512
final_out = 256
CULL_EPS = 0.06


def get_cam_mat(boxes, device="cuda"):
    """K, R, t of the refinement camera from the room box (last entry of ``boxes``)."""
    room = boxes[-1] if isinstance(boxes, (list, tuple)) else boxes
    room = [float(x) for x in room]
    theta, fl = -0.4, 400
    K = torch.tensor([[fl * inter_out / 1024, 0, inter_out / 2.0], [0, fl * inter_out / 1024, inter_out / 2.0], [0, 0, 1.0]],
                     dtype=torch.float32)[None]
    w2c = torch.tensor([[1, 0, 0], [0, np.cos(theta), np.sin(theta)], [0, -np.sin(theta), np.cos(theta)]], dtype=torch.float32)
    cam = torch.tensor([[room[3] / 2.0], [room[4] / 2.0 + min(0.1, abs(room[4] / 2.0))], [room[5]]], dtype=torch.float32)
    c2cv = torch.tensor([[1, 0, 0], [0, -1, 0], [0, 0, -1]], dtype=torch.float32)
    R = torch.matmul(c2cv, w2c).reshape(1, 3, 3)
    t = torch.matmul(c2cv, torch.matmul(w2c, -cam)).reshape(1, 1, 3)
    return K.to(device), R.to(device), t.to(device)


def class_tables(class_names):
    """Reference ordering (diff_render.py:372-376): sorted class list with 'wall' first; NYU channel per class
    (:429-431) and depth-channel index for everything but wall / floor / ceiling (:422-425)."""
    classes = sorted(set(class_names))
    classes.remove("wall"); classes.insert(0, "wall")
    chan = [nyu_class.index(c.replace("_", " ")) for c in classes]
    dch, k = [], 0
    for c in classes:
        if c in ("wall", "floor", "ceiling"):
            dch.append(-1)
        else:
            dch.append(k); k += 1
    # the fused pass writes the reference's 70 channels: depth + 40 NYU one-hot + (len(classes) - 3) depth-hot planes, 29 with
    # the shipped valid_types.json (diff_render.py:377-379); a longer class list would not fit the output tensor
    if k > N_DEPTH_HOT:
        raise ValueError("%d object classes besides wall / floor / ceiling: the fused scene pass holds %d depth-hot channels "
                         "(the reference's valid_types.json has 29)" % (k, N_DEPTH_HOT))
    return classes, chan, dch


N_DEPTH_HOT = 29
N_SCENE_CHANNELS = 41 + N_DEPTH_HOT


class _SceneFn(torch.autograd.Function):
    """final[B,70,is,is] = fused scene pass over faces[B,F,3,3]; backward -> d faces."""

    @staticmethod
    def forward(ctx, faces, face_class, chan, dch, image_size, near_rgb):
        L = _lib.lib()
        faces = faces.contiguous()
        B, F = faces.shape[0], faces.shape[1]
        dev = faces.device
        ws = torch.empty(int(L.sln_scene_workspace_bytes(B, F, image_size)), dtype=torch.uint8, device=dev)
        out = torch.empty(B, N_SCENE_CHANNELS, image_size, image_size, device=dev)
        nc = chan.numel()
        _lib.check(L.sln_scene_forward(_lib.ptr(faces), _lib.ptr(face_class), B, F, image_size, nc, _lib.ptr(chan), _lib.ptr(dch),
                                       0.1, float(near_rgb), 100.0, 1e-3, _lib.ptr(ws), _lib.ptr(out), _lib.current_stream_ptr()),
                   "sln_scene_forward")
        ctx.save_for_backward(faces, face_class, chan, dch, ws)
        ctx.image_size = image_size
        return out

    @staticmethod
    def backward(ctx, gout):
        faces, face_class, chan, dch, ws = ctx.saved_tensors
        B, F = faces.shape[0], faces.shape[1]
        g = torch.empty_like(faces)
        _lib.check(_lib.lib().sln_scene_backward(_lib.ptr(faces), _lib.ptr(face_class), B, F, ctx.image_size, chan.numel(),
                                                 _lib.ptr(chan), _lib.ptr(dch), 1e-3, _lib.ptr(ws), _lib.ptr(gout.contiguous()),
                                                 _lib.ptr(g), _lib.current_stream_ptr()), "sln_scene_backward")
        return g, None, None, None, None, None


def cull_and_classify(vertices_buf, face_buf, class_ranges, R, t):
    """Near-plane cull (diff_render.py:346-356) and the per-face class id in reference class order."""
    classes, chan, dch = class_tables(class_ranges.keys())
    F_old = face_buf.shape[1]
    cls = torch.full((F_old,), -1, dtype=torch.int32)
    for ci, name in enumerate(classes):
        for a, b in class_ranges[name]:
            cls[a:b] = ci
    dev = vertices_buf.device
    cam_z = (torch.matmul(vertices_buf, R.transpose(1, 2)) + t)[:, :, 2]
    fz = cam_z[:, face_buf.long()][0]
    valid = ~torch.any(fz < CULL_EPS, dim=2)[0]
    return face_buf[:, valid, :].detach(), cls.to(dev)[valid], classes, chan, dch


def scene_render(vertices_buf, face_buf, class_ranges, room_box, image_size=final_out, near=0.001):
    """diff_render.py:344-434 in one fused HIP pass.  vertices_buf [1,V,3] (grad flows), face_buf [1,F,3] int32,
    class_ranges {class: [[a,b],...]}, room_box = boxes[-1].  Returns final [1,70,is,is]."""
    dev = vertices_buf.device
    K, R, t = get_cam_mat(room_box, dev)
    faces, cls, classes, chan, dch = cull_and_classify(vertices_buf, face_buf, class_ranges, R, t)
    faces = torch.cat((faces, faces[:, :, [2, 1, 0]]), dim=1)                 # fill_back
    cls = torch.cat((cls, cls))[None].contiguous()
    faces, cls = _never_empty(faces, cls)
    fxyz = nr.project_faces(vertices_buf, faces, K, R, t, inter_out)
    chan_t = torch.tensor(chan, dtype=torch.int32, device=dev)
    dch_t = torch.tensor(dch, dtype=torch.int32, device=dev)
    return _SceneFn.apply(fxyz, cls, chan_t, dch_t, image_size, near)


def _never_empty(faces, face_class):
    """Every face culled (the reference then renders an empty image): keep one degenerate triangle of no class so that the
    kernels have a face list to walk; it covers no pixel."""
    if faces.shape[1] > 0:
        return faces, face_class
    B = faces.shape[0]
    return (torch.zeros(B, 1, 3, dtype=faces.dtype, device=faces.device),
            torch.full((B, 1), -1, dtype=torch.int32, device=faces.device))


def scene_render_batch(vertices, faces, face_class, chan, dch, K, R, t, image_size=final_out, near=0.001):
    """Batched fused pass for B rooms with equal (padded) V and F: vertices [B,V,3], faces [B,F,3] int32 (already
    culled, fill_back applied by the caller or not at all), face_class [B,F] int32, K/R/t [B,...]."""
    faces, face_class = _never_empty(faces, face_class)
    fxyz = nr.project_faces(vertices, faces, K, R, t, inter_out)
    return _SceneFn.apply(fxyz, face_class.contiguous(), chan, dch, image_size, near)


class SceneRenderGraph:
    """``scene_render_batch`` forward + backward to the vertices as ONE hipGraph for fixed shapes.  A refinement loop renders the
    same topology every iteration (testing/test_render_refine.py:279-359: only boxes / angles move); replaying the ~19 launches
    of one fused pass takes the host out of the loop.  (On the bench box the eager pass is already GPU-bound: 0.85 ms per 16
    rooms either way - what a graph cannot remove is the ~5 us boundary between dependent launches.)
    ``g = SceneRenderGraph(V, F, C, chan, dch, K, R, t); image, dV = g(V_new, grad_out)`` - the returned tensors are the graph's
    static buffers (overwritten by the next call); ``grad_out=None`` reuses the gradient already in ``g.grad_out``."""

    def __init__(self, vertices, faces, face_class, chan, dch, K, R, t, image_size=final_out, near=0.001):
        if vertices.device.type != 'cuda':
            raise _lib.SlnError("SceneRenderGraph runs on the MI355X only (no CPU fallback)")
        self.vertices = vertices.detach().clone().requires_grad_(True)
        args = (faces, face_class, chan, dch, K, R, t, image_size, near)

        def run():
            out = scene_render_batch(self.vertices, *args)
            return out, torch.autograd.grad(out, self.vertices, self.grad_out)[0]
        with torch.no_grad():
            probe = scene_render_batch(self.vertices.detach(), *args)
        self.grad_out = torch.zeros_like(probe)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                              # warm-up outside the capture (allocator, lazy init)
            run()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.image, self.grad_vertices = run()

    def __call__(self, vertices=None, grad_out=None):
        with torch.no_grad():
            if vertices is not None:
                self.vertices.copy_(vertices)
            if grad_out is not None:
                self.grad_out.copy_(grad_out)
        self.graph.replay()
        return self.image, self.grad_vertices


def scene_render_passes(vertices_buf, face_buf, class_ranges, room_box, image_size=final_out):
    """Same result through the reference's own pass structure (1 depth + one rgb pass per class) on top of the
    HIP ``Renderer`` - 33 rasterisations; kept as a cross-check of the fused path."""
    dev = vertices_buf.device
    K, R, t = get_cam_mat(room_box, dev)
    cam_z = (torch.matmul(vertices_buf, R.transpose(1, 2)) + t)[:, :, 2]
    F_old = face_buf.shape[1]
    valid = ~torch.any(cam_z[:, face_buf.long()][0] < CULL_EPS, dim=2)
    face_buf = face_buf[:, valid[0], :].detach()
    renderer = nr.Renderer(camera_mode='projection', image_size=image_size, K=K, R=R, t=t, anti_aliasing=False,
                           orig_size=inter_out, near=0.001, light_intensity_ambient=1.0, light_intensity_directional=0.0)
    tex = torch.ones(1, face_buf.shape[1], 2, 2, 2, 3, device=dev)
    depth = renderer(vertices_buf, face_buf, tex, mode='depth')
    depth = torch.where(depth > 15, torch.full_like(depth, -1.0), depth)
    classes, chan, dch = class_tables(class_ranges.keys())
    one_hot = torch.zeros(41, image_size, image_size, device=dev)
    depth_hot = torch.zeros(len(classes) - 3, image_size, image_size, device=dev)
    planes_oh, planes_dh, wall_max = {}, {}, None
    for ci, name in enumerate(classes):
        t_un = torch.zeros(1, F_old, 2, 2, 2, 3, device=dev)
        for a, b in class_ranges[name]:
            t_un[:, a:b] = 1.0
        images = renderer(vertices_buf, face_buf, t_un[:, valid[0]], mode="rgb")
        image = torch.sum(images, dim=1, keepdim=True)[0] / 3.0
        mask = image.detach() > 0.1
        mean = torch.mean(depth[mask]) if mask.any() else torch.tensor(float('nan'), device=dev)
        if name == "wall":
            wall_max = torch.max(depth[mask]).detach() if mask.any() else torch.tensor(10.0, device=dev)
        if torch.isnan(mean):
            mean = wall_max
        csd = torch.where(mask, depth, mean.expand_as(depth)) / wall_max
        if dch[ci] >= 0:
            planes_dh[dch[ci]] = csd[0]
        planes_oh[chan[ci] + 1] = image[0]
    oh = torch.stack([planes_oh.get(i, one_hot[i]) for i in range(41)])
    dh = torch.stack([planes_dh.get(i, depth_hot[i]) for i in range(depth_hot.shape[0])])
    return torch.cat((depth, oh[1:], dh), dim=0)[None]


def mesh_render_func(boxes, angles, objs, model_ids_old=None, obj_size_target=None):
    """The reference's entry point name and signature (diff_render.py:48); implementation in ``host/refine.py``."""
    from . import refine
    return refine.mesh_render_func(boxes, angles, objs, model_ids_old, obj_size_target)
