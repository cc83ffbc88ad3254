"""Layout refinement on the MI355X: counterpart of ``testing/test_render_refine.py::finetune_VAE`` (:243-359)
and of the scene assembly inside ``models/diff_render.py::mesh_render_func`` (:48-342).

What is reproduced from the reference
  * object placement (diff_render.py:76-159): box -> centre/size in room units, ``theta = -angle * 2pi/24``,
    isotropic scale ``min(size / model_size)``, rotation about y, translation ``centre - scale * R @ model_centre``;
    non-furniture classes are skipped (:93-97); the size loss (:98-100,160-165);
  * the room box of the last row is frozen to the first iteration's value (:55-60);
  * ``softargmax`` (test_render_refine.py:20-25), the ``fix_grad`` / ``quad_grad`` hooks (:217-228), the null-fill,
    ``PSP_pool_new`` multi-scale pooling (:192-215), ``loss = 100*0.5*L1(depth) + 100*sum CE/800 + 2*size_loss``
    and the per-iteration ``SGD(lr=2e-4, momentum=0.1, nesterov)`` over ``z`` (model parameters at lr/10) (:286-359).
What is replaced
  * mesh retrieval (models/misc.py: SUNCG meshes + pywavefront + pymesh, out of scope): every object class gets a
    procedural cuboid "model" with a fixed aspect ratio; walls / floor / ceiling are quads built from the room box;
    meshes stay resident on the GPU instead of being re-read from disk every iteration (models/misc.py:111-121);
  * the 33 raster passes per call: one fused HIP pass (diff_render.scene_render semantics).
``render_fn`` is injectable so that the tests can run the very same loss graph on the CPU oracle.
"""
import ctypes as C
import math
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import diff_render as DR
from . import synthetic
from .. import _lib

DO_NOT_VIS = ["wall", "ceiling", "floor", "person", "door", "window", "curtain", "blinds"]


def softargmax(input_vec, sum_dim, beta=2.0):
    idx = torch.cumsum(torch.ones_like(input_vec), dim=sum_dim)
    return torch.sum(F.softmax(input_vec * beta, dim=sum_dim) * idx, dim=sum_dim) - 1.0


def fix_grad(g):
    avg = g[:, 3:] / 2.0 + g[:, :3] / 2.0
    return torch.cat([avg, avg], dim=1)


def quad_grad(g):
    return g * 4.0


def psp_pool(feats, sizes=(32, 48, 64, 96), as_list=False):
    """PSP_pool_new: bilinear(align_corners=True) to each size, then bilinear (default) to the largest."""
    outs = [F.interpolate(F.interpolate(feats, size=(s, s), mode='bilinear', align_corners=True), size=(sizes[-1], sizes[-1]),
                          mode='bilinear', align_corners=False) for s in sizes]
    return outs if as_list else torch.cat(outs, 1)


class MeshBank:
    """Procedural stand-in for the SUNCG model table: one cuboid model per class, resident on the device."""

    def __init__(self, class_names, device, subdiv=2, seed=0):
        rng = np.random.default_rng(seed)
        self.models = {}
        for name in class_names:
            size = rng.uniform(0.5, 1.5, size=3)
            v, f = synthetic._grid_cuboid(-size / 2 + rng.uniform(-0.1, 0.1, size=3), size / 2, subdiv)
            v = torch.from_numpy(v.astype(np.float32)).to(device)
            self.models[name] = dict(v=v, f=torch.from_numpy(f.astype(np.int32)).to(device),
                                     bbox_min=v.min(0).values, bbox_max=v.max(0).values)


    vocab = None        # class list of the scene tensor's planes (object_idx_to_name[1:], diff_render.py:65); None: synthetic.FURNITURE
    shell = None        # wall / floor / ceiling meshes + their table boxes (see from_arrays); None: quads built from the room box

    @classmethod
    def from_arrays(cls, meshes, device, vocab=None, shell=None):
        """Caller-supplied meshes - the seam of models/diff_render.py:62,131, where the reference hands the retrieved SUNCG model's
        vertices / faces to the placement: ``meshes`` = {class name: (V [n,3] float, F [m,3] int)} or (V, F, bbox_min, bbox_max)
        (one model per class, any topology; faces index V).  The bounding box the placement scales by is the model table's
        (diff_render.py:106-115 read ``bbox_min`` / ``bbox_max`` of suncg_data) - the vertices' own when none is given.
        ``vocab``: the class list that lays out the 70 planes (``object_idx_to_name[1:]``, diff_render.py:65-69,372-379).
        ``shell``: the room's retrieved wall / floor / ceiling (diff_render.py:166-342) as arrays - dict(wall_v [n,3], wall_f = list of
        [m,3] sub-meshes over wall_v (models/misc.py:84-104), wall_bbox [2,3], floor_v, floor_f, floor_bbox [2,3], ceil_v, ceil_f);
        without it the shell is five quads on the room box."""
        bank = cls.__new__(cls)
        bank.models = {}
        for name, mesh in meshes.items():
            V, F = mesh[0], mesh[1]
            v = torch.as_tensor(np.asarray(V, dtype=np.float32)).reshape(-1, 3).to(device)
            f = torch.as_tensor(np.asarray(F).astype(np.int32)).reshape(-1, 3).to(device)
            if v.shape[0] == 0 or f.shape[0] == 0:
                raise ValueError("mesh of class %r is empty" % (name,))
            if int(f.min()) < 0 or int(f.max()) >= v.shape[0]:
                raise IndexError("faces of class %r index vertices outside [0, %d)" % (name, v.shape[0]))
            lo = torch.as_tensor(np.asarray(mesh[2], dtype=np.float32)).to(device) if len(mesh) > 2 else v.min(0).values
            hi = torch.as_tensor(np.asarray(mesh[3], dtype=np.float32)).to(device) if len(mesh) > 3 else v.max(0).values
            bank.models[name] = dict(v=v.contiguous(), f=f.contiguous(), bbox_min=lo, bbox_max=hi)
        bank.vocab = list(vocab) if vocab is not None else None
        if shell is not None:
            sh = {k: (np.asarray(shell[k], dtype=np.float32) if not k.endswith("_f") else None) for k in shell}
            sh["wall_f"] = [np.asarray(f).astype(np.int64).reshape(-1, 3) for f in shell["wall_f"]]
            sh["floor_f"], sh["ceil_f"] = (np.asarray(shell[k]).astype(np.int64).reshape(-1, 3) for k in ("floor_f", "ceil_f"))
            for f, nv, what in [(f, sh["wall_v"].shape[0], "wall") for f in sh["wall_f"]] + [(sh["floor_f"], sh["floor_v"].shape[0], "floor"),
                                                                                                (sh["ceil_f"], sh["ceil_v"].shape[0], "ceiling")]:
                if f.size and (int(f.min()) < 0 or int(f.max()) >= nv):
                    raise IndexError("faces of the %s index vertices outside [0, %d)" % (what, nv))
            bank.shell = sh
        return bank


def _classes_of(bank):
    return list(bank.vocab) if getattr(bank, "vocab", None) is not None else list(synthetic.FURNITURE)


_PROCEDURAL_SHELL = ("floor", "ceiling", "wall", "wall", "wall")
_SHELL_DIV = 6


def shell_topology(bank):
    """[(class, vertex count, faces [m,3] int64 numpy)] of the room shell in buffer order - does not depend on the room."""
    sh = getattr(bank, "shell", None)
    if sh is None:
        unit = np.zeros(3), np.array([1.0, 0, 0]), np.array([0, 1.0, 0])
        v, f = synthetic._grid_quad(unit[0], unit[1], unit[2], _SHELL_DIV)
        return [(nm, v.shape[0], f.astype(np.int64)) for nm in _PROCEDURAL_SHELL]
    # reference order (diff_render.py:166-342): every wall sub-mesh with its own copy of the wall's vertices, the floor, the ceiling
    return ([("wall", sh["wall_v"].shape[0], f) for f in sh["wall_f"]] +
            [("floor", sh["floor_v"].shape[0], sh["floor_f"]), ("ceiling", sh["ceil_v"].shape[0], sh["ceil_f"])])


def _scaled_into_room(v, scale, model_center, center):
    """diff_render.py:188-200: [I | center - scale * model_center] x diag(scale) as two 4x4 matrices applied to the homogeneous
    vertices - in fp32 on the host with the reference's operation order, so that the shell's vertices are the reference's bit for bit"""
    move, grow = torch.eye(4), torch.eye(4)
    move[:3, 3] = center - scale * model_center
    grow[:3, :3] = grow[:3, :3] * scale
    hom = torch.cat((v.t(), torch.ones(1, v.shape[0])), 0)
    return torch.matmul(torch.matmul(move, grow)[:3], hom).t().contiguous()


def place_shell(bank, room_ext):
    """Vertices of the room shell in room coordinates, [sum of shell_topology's vertex counts, 3] fp32 on the host.
    Procedural bank: floor, ceiling, back / left / right wall on the room box.  Bank with shell tables: diff_render.py:166-342 - the
    walls are scaled isotropically by the LARGEST ratio of room extent to the table's wall box and centred in the room, a wall
    sub-mesh that comes closer to the camera than 0.9 of the depth while its mean x lies in the middle 80 % of the width is dropped
    (:203-213; here: its vertices collapse to one point, so that the face list keeps its shape and order); floor: x / z ratios, y = 0;
    ceiling: x / z ratios of its own bounding box, resting on the room's height.  One-off per room: the room row is frozen (:55-60)."""
    sh = getattr(bank, "shell", None)
    room = [float(x) for x in room_ext]
    if sh is None:
        quads = [((0, 0, 0), (0, 0, room[2]), (room[0], 0, 0)), ((0, room[1], 0), (room[0], 0, 0), (0, 0, room[2])),
                 ((0, 0, 0), (room[0], 0, 0), (0, room[1], 0)), ((0, 0, 0), (0, room[1], 0), (0, 0, room[2])),
                 ((room[0], 0, 0), (0, 0, room[2]), (0, room[1], 0))]               # floor, ceiling, three walls (the order of shell_topology)
        sv = [synthetic._grid_quad(np.array(p0, np.float64), np.array(du, np.float64), np.array(dv, np.float64), _SHELL_DIV)[0] for p0, du, dv in quads]
        return torch.from_numpy(np.concatenate(sv).astype(np.float32))
    ext = torch.tensor(room, dtype=torch.float32)
    T = torch.from_numpy
    lo, hi = T(sh["wall_bbox"][0]), T(sh["wall_bbox"][1])
    wall = _scaled_into_room(T(sh["wall_v"]), torch.max(ext / (hi - lo)), (lo + hi) / 2.0, ext / 2.0)
    parts = []
    for f in sh["wall_f"]:
        fz, fx = wall[:, 2][T(f)], wall[:, 0][T(f)]
        drop = f.shape[0] > 0 and bool(fz.max() > 0.9 * ext[2]) and bool(fx.mean() > 0.1 * ext[0]) and bool(fx.mean() < 0.9 * ext[0])
        parts.append(torch.zeros_like(wall) if drop else wall)
    lo, hi = T(sh["floor_bbox"][0]), T(sh["floor_bbox"][1])
    span = hi - lo
    parts.append(_scaled_into_room(T(sh["floor_v"]), torch.max(ext[0] / span[0], ext[2] / span[2]), (lo + hi) / 2.0,
                                   torch.stack((ext[0] / 2.0, ext.new_zeros(()), ext[2] / 2.0))))
    cv = T(sh["ceil_v"])
    hi, lo = cv.max(0).values, cv.min(0).values
    span = hi - lo
    scale = torch.max(ext[0] / span[0], ext[2] / span[2])
    parts.append(_scaled_into_room(cv, scale, (lo + hi) / 2.0, torch.stack((ext[0] / 2.0, 0.5 * (scale * span)[1] + ext[1], ext[2] / 2.0))))
    return torch.cat(parts).contiguous()


def assemble_scene(boxes, angles, class_names, bank, room_box, obj_size_target=None):
    """diff_render.py:76-165 for one room: returns vertices_buf [1,V,3] (differentiable w.r.t. boxes / angles),
    face_buf [1,F,3] int32, class_ranges, obj sizes, size_loss.  ``boxes`` [n,6] in room-normalised units with the
    room row last, ``angles`` [n] in bins, ``room_box`` the frozen room row (6,)."""
    dev = boxes.device
    ranges = {c: [] for c in _classes_of(bank)}
    ranges.update(wall=[], floor=[], ceiling=[])
    verts, faces, sizes, voff, foff = [], [], [], 0, 0
    size_loss = boxes.new_zeros(())
    k = 0
    for i, name in enumerate(class_names[:-1]):
        if name in DO_NOT_VIS or name not in bank.models:
            continue
        m = bank.models[name]
        bmin, bmax = boxes[i][:3] * room_box[3:], boxes[i][3:] * room_box[3:]
        center, size = (bmax + bmin) / 2, bmax - bmin
        if obj_size_target is not None:
            size_loss = size_loss + F.mse_loss(size, obj_size_target[k])
        sizes.append(size.detach())
        k += 1
        theta = -angles[i] * (2 * math.pi / 24)
        msize, mcenter = m["bbox_max"] - m["bbox_min"], (m["bbox_min"] + m["bbox_max"]) / 2.0
        scale = torch.min(size / msize)
        c, s = torch.cos(theta), torch.sin(theta)
        zero, one = torch.zeros_like(c), torch.ones_like(c)
        rot = torch.stack([torch.stack([c, zero, s]), torch.stack([zero, one, zero]), torch.stack([-s, zero, c])])
        trans = center - scale * torch.matmul(rot, mcenter)
        v = torch.matmul(m["v"], (rot * scale).t()) + trans
        verts.append(v); faces.append(m["f"] + voff)
        ranges.setdefault(name, []).append([foff, foff + m["f"].shape[0]])
        voff += v.shape[0]; foff += m["f"].shape[0]
    # room shell for the frozen room box (diff_render.py:166-342; see place_shell)
    shell_v = place_shell(bank, room_box[3:].detach().cpu().tolist()).to(dev)
    at = 0
    for nm, nv, f in shell_topology(bank):
        verts.append(shell_v[at:at + nv]); faces.append(torch.from_numpy(f.astype(np.int32)).to(dev) + voff)
        ranges[nm].append([foff, foff + f.shape[0]])
        voff += nv; foff += f.shape[0]; at += nv
    return torch.cat(verts)[None], torch.cat(faces)[None], ranges, sizes, size_loss


_MESH_SOURCE = {"object_idx_to_name": None, "bank": None}


def configure_meshes(object_idx_to_name, bank=None, device="cuda"):
    """Stands in for the module globals of models/misc.py (:17-31: vocabulary + SUNCG model tables) that
    ``mesh_render_func`` reads.  ``bank`` defaults to one procedural model per furniture class."""
    _MESH_SOURCE["object_idx_to_name"] = list(object_idx_to_name)
    _MESH_SOURCE["bank"] = bank or MeshBank([n for n in set(object_idx_to_name) if n not in DO_NOT_VIS and n != "__room__"], device)


def mesh_render_func(boxes, angles, objs, model_ids_old=None, obj_size_target=None):
    """Same call contract as the reference (models/diff_render.py:48-435):
    ``boxes``: list of b tensors [6] (room-normalised, room row last), ``angles``: list of b scalar tensors (bins),
    ``objs``: list of b class indices -> ``(final[1,70,256,256], model_ids_return, obj_size_return, size_loss)``.
    First call (``model_ids_old is None``) caches the room box ("box_info"), the retrieved model id per object and the
    object sizes; later calls reuse them, overload the room box (:55-57) and add the size / wall-drift penalties
    (:98-100,160-165).  Mesh retrieval is the procedural ``MeshBank`` (see the module docstring)."""
    src = _MESH_SOURCE
    if src["bank"] is None:
        raise RuntimeError("call refine.configure_meshes(object_idx_to_name) first (stands in for models/misc.py globals)")
    names, bank = src["object_idx_to_name"], src["bank"]
    boxes = list(boxes)
    dev = boxes[0].device
    model_ids_return, obj_size_return = {}, []
    old_wall = boxes[-1].clone()
    if model_ids_old is not None:
        boxes[-1] = torch.from_numpy(np.asarray(model_ids_old["box_info"])).float().to(dev)
    else:
        model_ids_return["box_info"] = boxes[-1].detach().cpu().numpy()
    class_names = [names[int(o)] for o in objs]
    for i, n in enumerate(class_names[:-1]):
        if model_ids_old is None:
            model_ids_return[i] = n + "#0"                       # one procedural model per class
    target = None
    if obj_size_target is not None:
        target = [torch.from_numpy(np.asarray(t)).float().to(dev) for t in obj_size_target[:-1]]
    v, f, ranges, sizes, size_loss = assemble_scene(torch.stack(boxes), torch.stack([a.reshape(()) for a in angles]).float(),
                                                    class_names, bank, boxes[-1].detach(), target)
    if obj_size_target is not None:
        size_loss = size_loss + F.mse_loss(old_wall, torch.from_numpy(np.asarray(obj_size_target[-1])).float().to(dev))
    else:
        obj_size_return = [x.cpu().numpy() for x in sizes] + [boxes[-1].detach().cpu().numpy()]
        model_ids_return["wall"] = {"wall_bbox_min": [0.0, 0.0, 0.0], "wall_bbox_max": [float(x) for x in boxes[-1][3:]]}
    final = DR.scene_render(v, f, ranges, boxes[-1].detach())
    return final, model_ids_return, obj_size_return, size_loss


def refinement_loss(iter_image, target, target_container, size_loss):
    """test_render_refine.py:332-356 (target_container = per-scale argmax labels of the target, -100 where empty)."""
    iter_image = iter_image.clone()
    null = torch.sum(iter_image[:, 41:], dim=1) < 0.5
    last = iter_image[:, -1]
    iter_image[:, -1] = torch.where(null, torch.ones_like(last), last)
    depth_loss = F.l1_loss(psp_pool(iter_image[:, 41:]), psp_pool(target[:, 41:])) * 0.5
    sem = iter_image.new_zeros(())
    for pooled, tgt in zip(psp_pool(iter_image[:, 1:41], as_list=True), target_container):
        sem = sem + F.cross_entropy(pooled, tgt[:, 0]) / 800.0
    return depth_loss * 100 + sem * 100 + size_loss * 2.0, depth_loss, sem


def target_labels(target, sizes=(32, 48, 64, 96)):
    out = []
    for pooled in psp_pool(target[:, 1:41], sizes, as_list=True):
        flat = torch.argmax(pooled, dim=1, keepdim=True)
        flat[torch.sum(pooled, dim=1, keepdim=True) < 0.5] = -100
        out.append(flat.detach())
    return out


def _bilinear_taps(in_size, out_size, align_corners):
    """Source indices and weight of the upper neighbour exactly as upsample_bilinear2d derives them (fp32 arithmetic)."""
    dst = np.arange(out_size, dtype=np.float32)
    if align_corners:
        scale = np.float32(in_size - 1) / np.float32(out_size - 1) if out_size > 1 else np.float32(0)
        src = scale * dst
    else:
        src = np.maximum(np.float32(in_size) / np.float32(out_size) * (dst + np.float32(0.5)) - np.float32(0.5), np.float32(0))
    i0 = src.astype(np.int32)
    i1 = i0 + (i0 < in_size - 1)
    return i0, i1.astype(np.int32), (src - i0.astype(np.float32)).astype(np.float32)


class _RefineLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, rl):
        image = image.contiguous()
        out = torch.empty((image.shape[0], 3) if rl.per_room else (3,), device=image.device)     # per_room: one (loss, depth, sem) triple per image
        _lib.check(_lib.lib().sln_refine_loss_forward(rl.desc, _lib.ptr(image), _lib.ptr(rl.target_depth), _lib.ptr(rl.labels),
                                                      _lib.ptr(rl.inv_count), _lib.ptr(rl.ws), _lib.ptr(out), _lib.current_stream_ptr()),
                   "sln_refine_loss_forward")
        ctx.rl, ctx.shape = rl, image.shape
        return out

    @staticmethod
    def backward(ctx, gout):
        rl = ctx.rl
        g = torch.empty(ctx.shape, device=gout.device)
        # d/d(out[0]); out[1:] (the two parts) are reporting values.  per_room: ONE scale for every room (the rooms' losses are
        # independent objectives, each back-propagated with the same weight: element [0, 0])
        scale = gout.reshape(-1)[0:1].contiguous()
        _lib.check(_lib.lib().sln_refine_loss_backward(rl.desc, _lib.ptr(rl.ws), _lib.ptr(scale), _lib.ptr(g), _lib.current_stream_ptr()),
                   "sln_refine_loss_backward")
        return g, None


_REFINE_TABLES = {}          # (image size, scales, device) -> (device tables, longest CSR row): see RefineLoss.__init__


class RefineLoss:
    """``refinement_loss`` for a fixed target as two C calls (csrc/refine_loss.hip) instead of ~200 torch launches.
    ``rl(image)`` -> tensor [100*depth + 100*sem, depth, sem]; gradients flow to ``image`` through element 0.  The
    workspace holds d loss / d pooled between forward and backward: one outstanding forward per instance."""

    def __init__(self, target, sizes=(32, 48, 64, 96), per_room=False):
        self.per_room = bool(per_room)
        dev = target.device
        B, C, S, _ = target.shape
        P, ns, pmax = sizes[-1], len(sizes), max(sizes)
        # the resampling tables depend on the geometry only (image size, scales), not on the room: built once per geometry and device
        # (python / numpy loops over 4 x 256 image rows: ~5 ms of the ~7 ms per-room set-up of the refinement loop)
        key = (S, tuple(sizes), str(dev))
        cached = _REFINE_TABLES.get(key)
        if cached is None:
            cached = _REFINE_TABLES[key] = self._build_tables(S, sizes, dev)
        self._keep, max_col = cached
        d = self._describe(B, S, P, C, ns, pmax, max_col)
        self._finish(d, target, B, S, P, C, ns, dev)

    @staticmethod
    def _build_tables(S, sizes, dev):
        P, ns, pmax = sizes[-1], len(sizes), max(sizes)
        s2 = [np.zeros((ns, P), np.int32), np.zeros((ns, P), np.int32), np.zeros((ns, P), np.float32)]
        s1 = [np.zeros((ns, pmax), np.int32), np.zeros((ns, pmax), np.int32), np.zeros((ns, pmax), np.float32)]
        ptr, outs, ws_ = [], [], []
        for k, sz in enumerate(sizes):
            a = _bilinear_taps(S, sz, True)
            b = _bilinear_taps(sz, P, False)
            for dst, src in zip(s1, a):
                dst[k, :sz] = src
            for dst, src in zip(s2, b):
                dst[k] = src
            R1 = np.zeros((sz, S)); R2 = np.zeros((P, sz))
            np.add.at(R1, (np.arange(sz), a[0]), 1.0 - a[2].astype(np.float64)); np.add.at(R1, (np.arange(sz), a[1]), a[2].astype(np.float64))
            np.add.at(R2, (np.arange(P), b[0]), 1.0 - b[2].astype(np.float64)); np.add.at(R2, (np.arange(P), b[1]), b[2].astype(np.float64))
            comp = R2 @ R1                                             # [P, S]: pooled index <- image index
            base = sum(len(o) for o in outs)
            cp = [base]
            for y in range(S):
                nz = np.nonzero(comp[:, y])[0]
                outs.append(nz.astype(np.int32)); ws_.append(comp[nz, y].astype(np.float32))
                cp.append(cp[-1] + len(nz))
            ptr.append(np.asarray(cp, np.int32))
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        keep = [t(x) for x in s2 + s1] + [t(np.stack(ptr)), t(np.concatenate(outs)), t(np.concatenate(ws_))]
        return keep, int(max(len(o) for o in outs))

    def _describe(self, B, S, P, C, ns, pmax, max_col):
        d = _lib.SlnRefineLoss()
        d.B, d.image_size, d.pooled_size, d.channels = B, S, P, C
        d.sem0, d.n_sem, d.dep0, d.n_dep, d.n_scales, d.stage1_stride = 1, 40, 41, C - 41, ns, pmax
        for name, buf in zip(("s2_k0", "s2_k1", "s2_l1", "s1_i0", "s1_i1", "s1_l1", "col_ptr", "col_out", "col_w"), self._keep):
            setattr(d, name, buf.data_ptr())
        d.max_col_entries = max_col
        d.per_room = int(getattr(self, "per_room", False))
        return d

    def _finish(self, d, target, B, S, P, C, ns, dev):
        self.desc = d
        L = _lib.lib()
        self.ws = torch.empty(int(L.sln_refine_loss_workspace_bytes(B, S, P, ns, 40, C - 41)), dtype=torch.uint8, device=dev)
        _lib.check(L.sln_refine_loss_init(d, _lib.ptr(self.ws), _lib.current_stream_ptr()), "sln_refine_loss_init")
        # the target goes through the SAME resampling kernel as the iterates (no null-fill, test_render_refine.py:328-331):
        # where both images agree the pooled difference is exactly 0, as in the reference
        pooled = torch.empty(B, ns, C - 1, P, P, device=dev)
        _lib.check(L.sln_refine_pool(d, _lib.ptr(target.contiguous()), 0, _lib.ptr(self.ws), _lib.ptr(pooled), _lib.current_stream_ptr()),
                   "sln_refine_pool")
        self.target_depth = pooled[:, :, 40:].contiguous()                                              # [B, ns, 29, P, P]
        sem = pooled[:, :, :40]
        lab = torch.argmax(sem, dim=2)
        lab[sem.sum(dim=2) < 0.5] = -100                                                                # :341-343
        self.labels = lab.to(torch.int32).contiguous()                                                  # [B, ns, P, P]
        # per_room: every image is its own room - its cross-entropy means run over its own labels ([B, ns] counts)
        cnt = (lab >= 0).sum(dim=(2, 3) if self.per_room else (0, 2, 3)).to(torch.float32)
        self.inv_count = (1.0 / cnt).contiguous()                                                       # 1/0: nan loss, as torch's empty mean

    def pooled_ones(self):
        """[n_scales, P, P]: what the resampling kernel makes of an all-ones plane (one per geometry and device; see
        SlnRefineLoss::pooled_ones)"""
        d = self.desc
        key = ("ones", d.image_size, d.pooled_size, d.n_scales, d.stage1_stride, str(self.ws.device), self._keep[0].data_ptr())
        t = _REFINE_TABLES.get(key)
        if t is None:
            L, dev = _lib.lib(), self.ws.device
            d1 = type(d).from_buffer_copy(d)
            d1.B, d1.per_room, d1.live_planes, d1.pooled_ones = 1, 0, None, None
            C_, S, P, ns = d.channels, d.image_size, d.pooled_size, d.n_scales
            ws1 = torch.empty(int(L.sln_refine_loss_workspace_bytes(1, S, P, ns, 40, C_ - 41)), dtype=torch.uint8, device=dev)
            pooled = torch.empty(1, ns, C_ - 1, P, P, device=dev)
            _lib.check(L.sln_refine_pool(d1, _lib.ptr(torch.ones(1, C_, S, S, device=dev)), 0, _lib.ptr(ws1), _lib.ptr(pooled),
                                         _lib.current_stream_ptr()), "sln_refine_pool(ones)")
            t = _REFINE_TABLES[key] = pooled[0, :, 40].contiguous()
        return t

    def __call__(self, image):
        return _RefineLossFn.apply(image, self)


class _PlaceFn(torch.autograd.Function):
    """boxes [n,6], angles [n] -> (faces [2F,3,3] projected / culled / fill_back'ed, size_loss, sizes): csrc/placement.hip"""

    @staticmethod
    def forward(ctx, boxes, angles, scene, size_target):
        boxes, angles = boxes.contiguous().float(), angles.contiguous().float()
        dev = boxes.device
        fxyz = torch.empty(2 * scene.desc.F, 3, 3, device=dev)
        sizes = torch.empty(max(scene.n_vis, 1), 3, device=dev)
        sl = torch.empty(1, device=dev)
        tgt = size_target.contiguous().float() if size_target is not None else None
        _lib.check(_lib.lib().sln_place_forward(scene.desc, _lib.ptr(boxes), _lib.ptr(angles), _lib.ptr(tgt), _lib.ptr(fxyz), _lib.ptr(sizes),
                                                _lib.ptr(sl), _lib.current_stream_ptr()), "sln_place_forward")
        ctx.scene, ctx.tgt = scene, tgt
        ctx.save_for_backward(boxes, angles)
        sizes = sizes[:scene.n_vis]
        ctx.mark_non_differentiable(sizes)             # the reference detaches the cached sizes (diff_render.py:102)
        return fxyz, sl.reshape(()), sizes

    @staticmethod
    def backward(ctx, g_fxyz, g_sl, _g_sizes):
        boxes, angles = ctx.saved_tensors
        gb, ga = torch.empty_like(boxes), torch.empty_like(angles)
        _lib.check(_lib.lib().sln_place_backward(ctx.scene.desc, _lib.ptr(boxes), _lib.ptr(angles), _lib.ptr(ctx.tgt), _lib.ptr(g_fxyz.contiguous()),
                                                 _lib.ptr(g_sl.reshape(1).contiguous()), _lib.ptr(gb), _lib.ptr(ga), _lib.current_stream_ptr()),
                   "sln_place_backward")
        return gb, ga, None, None


class _HeadFn(torch.autograd.Function):
    """(boxes_pred [n,6], angles_pred [n,24], noise [n], box_last [6], angle_last [1]) -> (boxes_full [n,6], idx [n]): the
    soft-argmax, the noise, the two ``torch.cat`` with the frozen room row and - in backward - the ``fix_grad`` / ``quad_grad``
    hooks (testing/test_render_refine.py:20-25, 217-228, 296-306) as one launch each way (csrc/placement.hip) instead of ~25."""

    @staticmethod
    def forward(ctx, boxes_pred, angles_pred, noise, box_last, angle_last, beta):
        boxes_pred, angles_pred = boxes_pred.contiguous().float(), angles_pred.contiguous().float()
        n, na = angles_pred.shape
        boxes_full, idx = torch.empty_like(boxes_pred), torch.empty(n, device=boxes_pred.device)
        _lib.check(_lib.lib().sln_refine_head_forward(n, na, _lib.ptr(boxes_pred), _lib.ptr(angles_pred), _lib.ptr(noise), _lib.ptr(box_last),
                                                      _lib.ptr(angle_last), float(beta), _lib.ptr(boxes_full), _lib.ptr(idx),
                                                      _lib.current_stream_ptr()), "sln_refine_head_forward")
        ctx.save_for_backward(angles_pred)
        ctx.beta = float(beta)
        return boxes_full, idx

    @staticmethod
    def backward(ctx, g_boxes, g_idx):
        angles_pred, = ctx.saved_tensors
        n, na = angles_pred.shape
        gb, ga = torch.empty(n, 6, device=angles_pred.device), torch.empty_like(angles_pred)
        if g_boxes is None:
            g_boxes = torch.zeros(n, 6, device=angles_pred.device)
        if g_idx is None:
            g_idx = torch.zeros(n, device=angles_pred.device)
        _lib.check(_lib.lib().sln_refine_head_backward(n, na, _lib.ptr(angles_pred), _lib.ptr(g_boxes.contiguous()), _lib.ptr(g_idx.contiguous()),
                                                       ctx.beta, _lib.ptr(gb), _lib.ptr(ga), _lib.current_stream_ptr()),
                   "sln_refine_head_backward")
        return gb, ga, None, None, None, None


class RefineScene:
    """The same placement + render as ``assemble_scene`` + ``DR.scene_render`` with every per-object python loop of
    diff_render.py:76-159 turned into ONE batched tensor expression over the visible objects, and fixed tensor shapes:
    the near-plane cull (:346-356) degenerates the culled faces (all three corners collapse to a point, so they never
    cover a pixel) instead of compacting the face list.  Nothing in ``render`` synchronises with the host, which is what
    lets a whole refinement iteration be captured into one hipGraph (``finetune_vae(..., capture=True)``).
    Built once per room: the room box is frozen (:55-60) and so are K, R, t."""

    def __init__(self, class_names, bank, room_box, image_size=DR.final_out):
        dev = room_box.device
        self.image_size = image_size
        self.room = room_box.detach().clone()
        # everything that depends on the class list only (object meshes, face topology incl. the shell's, class tables) is built once
        # per (bank, class list) and shared by the rooms that use it - the refinement of a test set builds thousands of scenes from a
        # few dozen class lists; per room: the shell's vertices and the camera (round 5: 1.4 -> 0.3 ms per scene)
        cache = bank.__dict__.setdefault("_scene_cache", {})
        key = (tuple(class_names), str(dev))
        st = cache.get(key)
        if st is None:
            st = cache[key] = self._static_part(class_names, bank, dev)
        for k_, v_ in st.items():
            setattr(self, k_, v_)
        Vm = self._Vm
        room_host = [float(x) for x in room_box.detach().cpu().tolist()]           # one device -> host copy per scene
        room = room_host[3:]
        self.shell_v = place_shell(bank, room).to(dev)                             # (the order of shell_topology / _static_part)
        Kc, Rc, tc = DR.get_cam_mat([room_host], "cpu")
        self.K, self.R, self.t = Kc.to(dev), Rc.to(dev), tc.to(dev)
        # descriptor of the fused placement kernels (csrc/placement.hip)
        self._keep = list(self._keep_static) + [self.shell_v]
        d = _lib.SlnPlacement()
        d.n, d.n_vis, d.Vm, d.Vs, d.F = len(class_names), self.n_vis, Vm, int(self.shell_v.shape[0]), int(self.faces.shape[0])
        for name, buf in zip(("vis", "model_v", "msize", "mcenter", "faces", "obj_face_ptr", "shell_v"), self._keep):
            setattr(d, name, buf.data_ptr())
        for name, vals in (("ext", room), ("K", Kc.reshape(-1).tolist()), ("R", Rc.reshape(-1).tolist()), ("t", tc.reshape(-1).tolist())):
            setattr(d, name, (type(getattr(d, name)))(*[float(x) for x in vals]))
        d.orig_size, d.proj_eps, d.cull_eps = float(DR.inter_out), 1e-9, float(DR.CULL_EPS)
        self.desc = d

    @staticmethod
    def _static_part(class_names, bank, dev):
        vis = [i for i, nm in enumerate(class_names[:-1]) if nm not in DO_NOT_VIS and nm in bank.models]
        models = [bank.models[class_names[i]] for i in vis]
        n_vis = len(vis)
        Vm = max([m["v"].shape[0] for m in models] + [1])
        mv = torch.zeros(max(n_vis, 1), Vm, 3, device=dev)
        for k, m in enumerate(models):
            mv[k, :m["v"].shape[0]] = m["v"]
        msize = torch.stack([m["bbox_max"] - m["bbox_min"] for m in models]) if models else torch.ones(1, 3, device=dev)
        mcenter = torch.stack([(m["bbox_min"] + m["bbox_max"]) / 2.0 for m in models]) if models else torch.zeros(1, 3, device=dev)
        ranges = {c: [] for c in _classes_of(bank)}
        ranges.update(wall=[], floor=[], ceiling=[])
        faces, foff = [], 0
        for k, (i, m) in enumerate(zip(vis, models)):
            faces.append(m["f"].long() + k * Vm)
            ranges.setdefault(class_names[i], []).append([foff, foff + m["f"].shape[0]]); foff += m["f"].shape[0]
        voff = n_vis * Vm
        for nm, nv, f in shell_topology(bank):                                 # topology only: the corners are the room's (place_shell)
            faces.append(torch.from_numpy(f.astype(np.int64)).to(dev) + voff)
            ranges[nm].append([foff, foff + f.shape[0]]); voff += nv; foff += f.shape[0]
        faces = torch.cat(faces)                                             # [F,3] into the flattened vertex list
        faces32 = faces.to(torch.int32)[None].contiguous()
        classes, chan, dch = DR.class_tables(ranges.keys())
        cls = torch.full((foff,), -1, dtype=torch.int32)
        for ci, name in enumerate(classes):
            for a, b in ranges[name]:
                cls[a:b] = ci
        vis_t = torch.tensor(vis, dtype=torch.int64, device=dev)
        counts = [m["f"].shape[0] for m in models]
        keep = [vis_t.to(torch.int32), mv.reshape(-1, 3).contiguous(), msize.contiguous(), mcenter.contiguous(), faces32[0].contiguous(),
                torch.tensor(np.concatenate([[0], np.cumsum(counts)]).astype(np.int32), device=dev)]
        return dict(vis=vis_t, n_vis=n_vis, _Vm=Vm, model_v=mv, msize=msize, mcenter=mcenter, faces=faces, faces32=faces32,
                    cls2=torch.cat((cls, cls))[None].contiguous().to(dev),          # fill_back doubles the faces
                    chan=torch.tensor(chan, dtype=torch.int32, device=dev), dch=torch.tensor(dch, dtype=torch.int32, device=dev),
                    _keep_static=keep)

    def place(self, boxes, angles):
        """-> vertices [1,V,3] (differentiable), object sizes [n_vis,3]"""
        ext = self.room[3:]
        b = boxes[self.vis]
        bmin, bmax = b[:, :3] * ext, b[:, 3:] * ext
        center, size = (bmax + bmin) / 2, bmax - bmin
        theta = -angles[self.vis] * (2 * math.pi / 24)
        scale = (size / self.msize).min(dim=1).values
        c, s_ = torch.cos(theta), torch.sin(theta)
        z, o = torch.zeros_like(c), torch.ones_like(c)
        rot = torch.stack([torch.stack([c, z, s_], 1), torch.stack([z, o, z], 1), torch.stack([-s_, z, c], 1)], 1)     # [n,3,3]
        trans = center - scale[:, None] * torch.matmul(rot, self.mcenter[:, :, None])[:, :, 0]
        v = torch.matmul(self.model_v, (rot * scale[:, None, None]).transpose(1, 2)) + trans[:, None, :]
        return torch.cat([v.reshape(-1, 3), self.shell_v])[None], size

    def render(self, boxes, angles, obj_size_target=None, fused=True):
        """-> final [1,70,is,is], size_loss, sizes.  ``fused``: placement, projection, cull and fill_back as ONE kernel each way
        (csrc/placement.hip) instead of the torch expression below (~100 launches per iteration with its autograd nodes)."""
        if fused:
            fxyz, size_loss, size = _PlaceFn.apply(boxes, angles, self, obj_size_target if self.n_vis else None)
            img = DR._SceneFn.apply(fxyz[None], self.cls2, self.chan, self.dch, self.image_size, 0.001)
            return img, size_loss, size
        verts, size = self.place(boxes, angles)
        size_loss = boxes.new_zeros(())
        if obj_size_target is not None and self.n_vis:
            size_loss = ((size - obj_size_target) ** 2).mean(1).sum()
        cam_z = (torch.matmul(verts, self.R.transpose(1, 2)) + self.t)[0, :, 2]
        culled = (cam_z[self.faces] < DR.CULL_EPS).any(1).detach()
        fxyz = DR.nr.project_faces(verts, self.faces32, self.K, self.R, self.t, DR.inter_out)[0]      # [F,3,3]
        fxyz = torch.where(culled[:, None, None], torch.zeros_like(fxyz), fxyz)
        fxyz = torch.cat((fxyz, torch.flip(fxyz, [1])), 0)[None]              # fill_back: corners (2, 1, 0)
        img = DR._SceneFn.apply(fxyz, self.cls2, self.chan, self.dch, self.image_size, 0.001)
        return img, size_loss, size


def finetune_vae_fast(model, objs, triples, boxes_gt, angles_gt, attributes, class_names, iters=60, bank=None, learning_rate=1e-4,
                      noise_seed=13, image_size=256, capture=False, log=None, fused_loss=True, fused_head=True):
    """``finetune_vae`` with the batched ``RefineScene`` and no per-iteration python optimiser objects: the reference builds a
    NEW SGD(momentum=0.1, nesterov) every iteration (test_render_refine.py:286-292), so its step is exactly
    ``p -= lr * (1 + momentum) * grad``; that closed form is applied to ``z`` (lr 2e-4) and to the flat parameter buffer
    (lr ``learning_rate``/10).  ``fused_loss``: the PSP-pool / L1 / cross-entropy block runs as ``RefineLoss`` (two C calls)
    instead of torch ops; ``fused_head``: the soft-argmax / noise / concatenation glue and the two gradient hooks as one launch
    each way (``_HeadFn``) and both parameter updates plus the gradient zero-fill as one launch (``sln_refine_sgd``).
    ``capture=True`` records one iteration (decoder, placement, fused render, PSP losses,
    backward, both updates) into a hipGraph and replays it; the noise of the angle soft-argmax is then drawn on the device.
    Returns (losses [iters] tensor on the device, (boxes_pred, angle_idx))."""
    dev = boxes_gt.device
    bank = bank or MeshBank([n for n in set(class_names) if n not in DO_NOT_VIS], dev)
    model.eval()
    with torch.no_grad():
        mu, logvar = model.encoder(objs, triples, boxes_gt, angles_gt, attributes)
    gen = torch.Generator(device="cpu").manual_seed(noise_seed)
    z = (mu + torch.randn(mu.shape, generator=gen).to(dev) * torch.exp(0.5 * logvar)).detach().clone().requires_grad_(True)
    room_box = boxes_gt[-1].detach().clone()
    scene = RefineScene(class_names, bank, room_box, image_size)
    with torch.no_grad():
        target, _, sizes = scene.render(boxes_gt, angles_gt.float())
    labels = target_labels(target)
    fused = RefineLoss(target) if fused_loss else None             # the PSP / L1 / cross-entropy block as two C calls
    size_target = sizes.detach().clone()                           # (a buffer: filled with the FIRST iterate's sizes below)
    n = boxes_gt.shape[0]
    noise = torch.zeros(n, device=dev)
    # the soft-argmax noise of every iteration, drawn in the reference's order (one randn(n) per iteration) but uploaded once:
    # a per-iteration host-to-device copy would block the host and leave the GPU idle between two iterations
    noise_all = torch.stack([torch.randn(n, generator=gen) for _ in range(iters)]).to(dev) if iters > 0 else torch.zeros(0, n, device=dev)
    losses = torch.zeros(iters, device=dev)
    flat, flat_grad = model.flat_params, model.flat_grads
    state = {}

    box_last = boxes_gt[-1].detach().float().contiguous()
    angle_last = angles_gt[-1:].detach().float().contiguous()
    if fused_head:
        flat_grad.zero_()                                  # from here on the update kernel leaves the gradient buffer zeroed
    if iters > 0 and scene.n_vis:
        # the size penalty holds the objects to the sizes of the FIRST iterate (test_render_refine.py:319-327: size_infos is what the
        # first mesh_render_func call on boxes_pred returns; the target render's sizes are discarded): one decoder + placement pass
        # with iteration 0's z, parameters and noise row.  Iteration 0 then measures its own sizes against themselves - a size loss
        # of exactly 0 with zero gradient, which is the reference's ``size_loss = 0.0`` of the first call.
        with torch.no_grad():
            bp0, ap0 = model.decoder(z, objs, triples, attributes)
            b0 = torch.cat([bp0[:-1], boxes_gt[-1:]], 0)
            i0 = torch.cat([(softargmax(ap0, sum_dim=1) + noise_all[0] / 10.0)[:-1], angles_gt[-1:].float()], 0)
            size_target.copy_(_PlaceFn.apply(b0, i0, scene, None)[2])

    def iteration():
        boxes_pred, angles_pred = model.decoder(z, objs, triples, attributes)
        if fused_head:
            boxes_full, idx = _HeadFn.apply(boxes_pred, angles_pred, noise, box_last, angle_last, 2.0)
        else:
            boxes_pred.register_hook(fix_grad)
            boxes_full = torch.cat([boxes_pred[:-1], boxes_gt[-1:]], 0)
            idx = softargmax(angles_pred, sum_dim=1) + noise / 10.0
            idx.register_hook(quad_grad)
            idx = torch.cat([idx[:-1], angles_gt[-1:].float()], 0)
        image, size_loss, _ = scene.render(boxes_full, idx, size_target)
        if fused is not None:
            loss = torch.add(fused(image)[0], size_loss, alpha=2.0)
        else:
            loss, _, _ = refinement_loss(image, target, labels, size_loss)
        z.grad = None
        if not fused_head:
            flat_grad.zero_()
        loss.backward()
        with torch.no_grad():
            if fused_head:
                _lib.check(_lib.lib().sln_refine_sgd(_lib.ptr(flat), _lib.ptr(flat_grad), flat.numel(), (learning_rate / 10.0) * 1.1,
                                                     _lib.ptr(z), _lib.ptr(z.grad), z.numel(), 2e-4 * 1.1, _lib.current_stream_ptr()),
                           "sln_refine_sgd")
            else:
                z.add_(z.grad, alpha=-2e-4 * 1.1)
                flat.add_(flat_grad, alpha=-(learning_rate / 10.0) * 1.1)
        model.params_changed()
        state["boxes"], state["idx"] = boxes_full.detach(), idx.detach()
        return loss.detach()

    graph = None
    for k in range(iters):
        if capture:
            noise.copy_(noise_all[k])
            if graph is None:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):                      # warm-up outside the capture (allocator, lazy init)
                    state["loss"] = iteration()
                torch.cuda.current_stream().wait_stream(side)
                losses[k] = state["loss"]
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    state["loss"] = iteration()
                continue
            graph.replay()
            losses[k] = state["loss"]
        else:
            noise.copy_(noise_all[k])
            losses[k] = iteration()
        if log:
            log("iter %d: loss %.4f" % (k, float(losses[k])))
    return losses, (state["boxes"], state["idx"])


class RefineBatch:
    """R rooms in flight: one refinement iteration of R independent rooms (testing/test_render_refine.py:250-263 runs its trials
    one after the other; each reloads the checkpoint, ``model.eval()``, and :279-359 steps ``z`` AND its own copy of the parameters)
    as ONE sequence of launches - no autograd graph, no per-room python:

        decoder of all rooms (sln_vae_group_decoder: R parameter copies, one launch per step) -> soft-argmax / noise / frozen room
        row (sln_refine_head_forward_rooms) -> placement + projection + cull + fill_back (sln_place_forward_rooms) -> fused scene
        pass over the R padded face lists (sln_scene_forward) -> PSP / L1 / cross-entropy per room (SlnRefineLoss.per_room) ->
        the same chain backwards -> decoder backward of all rooms -> SGD on every room's parameter copy and on z (one launch).

    ``rooms``: list of dicts with ``objs, triples, boxes, angles, attributes`` (one room's collated graph, room row last) and
    ``class_names``.  Every room starts from ``model``'s parameters (the checkpoint), encoded with ``model`` in eval mode.
    A room's numbers do not depend on the other rooms of the batch: the kernels are the single-room ones over a (room) grid axis.
    After ``run(iters)``: ``losses`` [iters, R], ``boxes`` / ``idx`` (row-concatenated, ``row0`` / ``rows`` per room), ``params``
    [R, n_flat] the fine-tuned copies, ``z`` the refined latents."""

    def __init__(self, model, rooms, bank=None, learning_rate=1e-4, noise_seed=13, image_size=256, iters=60):
        L = _lib.lib()
        self.model, self.R, self.iters, self.lr = model, len(rooms), int(iters), float(learning_rate)
        R = self.R
        _log = os.environ.get("SLN_REFINE_SETUP_LOG")
        _t = [time.perf_counter()]

        def tick(what):                              # lab: where the set-up time of a batch goes (synchronises: only with the switch)
            if _log:
                torch.cuda.synchronize()
                now = time.perf_counter()
                print("RefineBatch set-up: %-28s %7.2f ms" % (what, (now - _t[0]) * 1e3), flush=True)
                _t[0] = now
        if R < 1:
            raise ValueError("RefineBatch needs at least one room")
        dev = model.flat_params.device
        names_all = set(n for rm in rooms for n in rm["class_names"])
        bank = bank or MeshBank([n for n in names_all if n not in DO_NOT_VIS and n != "__room__"], dev)
        model.eval()
        E, na, S = model.embedding_dim, model.Nangle, int(image_size)
        self.rows = [int(rm["objs"].shape[0]) for rm in rooms]
        self.row0 = [0]
        for n in self.rows[:-1]:
            self.row0.append(self.row0[-1] + n)
        N = sum(self.rows)
        self.N = N
        f32 = dict(dtype=torch.float32, device=dev)
        # ---- per room: encoder (the checkpoint model), z draw, noise of every iteration, scene, target render ----
        z = torch.empty(N, E, **f32)
        noise = torch.zeros(max(self.iters, 1), N)
        scenes, targets, size_targets = [], [], []
        box_last, angle_last = torch.empty(R, 6, **f32), torch.empty(R, **f32)
        for r, rm in enumerate(rooms):
            with torch.no_grad():
                mu, logvar = model.encoder(rm["objs"], rm["triples"], rm["boxes"], rm["angles"], rm["attributes"])
            gen = torch.Generator(device="cpu").manual_seed(noise_seed)          # torch.manual_seed(13) in front of every trial (:274-275)
            a, n = self.row0[r], self.rows[r]
            z[a:a + n] = mu + torch.randn(mu.shape, generator=gen).to(dev) * torch.exp(0.5 * logvar)
            if self.iters > 0:           # one randn(n) per iteration, as the reference draws them (:304); one strided copy into the table
                noise[:self.iters, a:a + n] = torch.stack([torch.randn(n, generator=gen) for _ in range(self.iters)])
            sc = RefineScene(rm["class_names"], bank, rm["boxes"][-1].detach().clone(), S)
            with torch.no_grad():
                tgt, _, sizes = sc.render(rm["boxes"], rm["angles"].float())
            scenes.append(sc); targets.append(tgt); size_targets.append(sizes.detach().clone().contiguous())
            box_last[r] = rm["boxes"][-1].detach().float(); angle_last[r] = rm["angles"][-1].detach().float()
        self.z, self.scenes = z, scenes
        tick("encoder, z, scenes, targets")
        self.noise_all = noise.to(dev)
        self.box_last, self.angle_last = box_last, angle_last
        # ---- the loss of all rooms: one descriptor, per-room normalisation ----
        self.loss = RefineLoss(torch.cat(targets, 0), per_room=True)
        del targets
        tick("RefineLoss")
        # ---- R parameter copies, R engines, one launch program ----
        nflat = model.flat_params.numel()
        self.params = model.flat_params.detach().unsqueeze(0).repeat(R, 1).contiguous()
        self.grads = torch.zeros(R, nflat, **f32)
        tick("parameter copies")
        self._engines = model.room_engines(self.params, self.grads, max(self.rows), max(int(rm["triples"].shape[0]) for rm in rooms))
        tick("room engines (create + bind)")
        st = _lib.current_stream_ptr()
        self._keep = []
        for (h, _ws, _arr), rm in zip(self._engines, rooms):
            n = int(rm["objs"].shape[0])
            objs, tri, attrs = (rm[k].to(torch.int64).contiguous() for k in ("objs", "triples", "attributes"))
            zb, za = torch.zeros(n, model.box_dim, **f32), torch.zeros(n, dtype=torch.int64, device=dev)
            b = _lib.SlnVaeBatch()
            b.objs, b.triples, b.boxes, b.angles, b.attributes = objs.data_ptr(), tri.data_ptr(), zb.data_ptr(), za.data_ptr(), attrs.data_ptr()
            b.O, b.T = n, int(tri.shape[0])
            _lib.check(L.sln_vae_set_batch(h, C.byref(b), st), "sln_vae_set_batch")
            self._keep.append((objs, tri, attrs, zb, za))
        self.boxes_pred, self.angles_pred = torch.empty(N, model.box_dim, **f32), torch.empty(N, na, **f32)
        self.d_boxes_pred, self.d_angles_pred = torch.zeros(N, 8, **f32), torch.empty(N, na, **f32)
        self.dz = torch.empty(N, E, **f32)
        io = _lib.SlnVaeGroupIO()
        io.rows_total = N
        self._row0_c = (C.c_int * R)(*self.row0)
        io.row0_host = self._row0_c
        io.z, io.boxes_pred, io.angles_pred = self.z.data_ptr(), self.boxes_pred.data_ptr(), self.angles_pred.data_ptr()
        io.d_boxes_pred, io.d_angles_pred, io.dz = self.d_boxes_pred.data_ptr(), self.d_angles_pred.data_ptr(), self.dz.data_ptr()
        # SGD of the Linear weights / biases in the epilogue of their wgrads (one pass over the R parameter copies instead of the
        # gradient += and the optimizer's read of both and write; SLN_REFINE_SEPARATE_SGD=1: the stand-alone step over everything)
        self._step = torch.full((1,), (self.lr / 10.0) * 1.1, **f32)
        if not os.environ.get("SLN_REFINE_SEPARATE_SGD"):
            io.sgd_step = self._step.data_ptr()
        harr = (C.c_void_p * R)(*[e[0] for e in self._engines])
        g = C.c_void_p()
        torch.cuda.current_stream(dev).synchronize()          # the tables of the program are uploaded with blocking copies
        _lib.check(L.sln_vae_group_create(harr, R, C.byref(io), C.byref(g)), "sln_vae_group_create")
        self._group = g
        tick("set_batch + group create")
        # ---- head / placement tables ----
        self.room_of_row = torch.cat([torch.full((n,), r, dtype=torch.int32) for r, n in enumerate(self.rows)]).to(dev)
        self.last_row = torch.tensor([a + n - 1 for a, n in zip(self.row0, self.rows)], dtype=torch.int32, device=dev)
        self.boxes, self.idx = torch.empty(N, 6, **f32), torch.empty(N, **f32)
        self.g_boxes, self.g_idx = torch.empty(N, 6, **f32), torch.empty(N, **f32)
        self.F2 = 2 * max(sc.desc.F for sc in scenes)
        self.n_max = max(self.rows)
        self.faces = torch.zeros(R, self.F2, 3, 3, **f32)             # rows beyond a room's 2 F stay degenerate triangles of no class
        self.g_faces = torch.empty(R, self.F2, 3, 3, **f32)
        cls = torch.full((R, self.F2), -1, dtype=torch.int32, device=dev)
        self.sizes = torch.zeros(R, max(max(sc.n_vis for sc in scenes), 1), 3, **f32)
        self.size_loss = torch.zeros(R, **f32)
        self.g_size_loss = torch.full((1,), 2.0, **f32)               # loss = ... + 2 * size_loss (test_render_refine.py:350-352)
        self._size_targets = size_targets
        # the soft-argmax / noise / frozen-row glue inside the two placement launches (SlnPlacementRoom's head fields; the noise of
        # iteration k is row k of noise_all, k a device counter the backward launch advances)
        self._fused_head = max(self.rows) <= 128 and not os.environ.get("SLN_REFINE_SEPARATE_HEAD")
        self._noise_step = torch.zeros(1, dtype=torch.int32, device=dev)
        tab = (_lib.SlnPlacementRoom * R)()
        for r, sc in enumerate(scenes):
            cls[r, :2 * sc.desc.F] = sc.cls2[0]
            e, a = tab[r], self.row0[r]
            e.P = sc.desc
            e.boxes, e.angles = self.boxes.data_ptr() + 24 * a, self.idx.data_ptr() + 4 * a
            e.size_target = size_targets[r].data_ptr() if sc.n_vis else None
            e.faces_out, e.sizes, e.size_loss = self.faces[r].data_ptr(), self.sizes[r].data_ptr(), self.size_loss.data_ptr() + 4 * r
            e.grad_faces, e.grad_size_loss = self.g_faces[r].data_ptr(), self.g_size_loss.data_ptr()
            e.grad_boxes, e.grad_angles = self.g_boxes.data_ptr() + 24 * a, self.g_idx.data_ptr() + 4 * a
            if self._fused_head:
                e.boxes_pred, e.angles_pred = self.boxes_pred.data_ptr() + 24 * a, self.angles_pred.data_ptr() + 4 * na * a
                e.noise, e.noise_step, e.noise_stride = self.noise_all.data_ptr() + 4 * a, self._noise_step.data_ptr(), N
                e.box_last, e.angle_last = self.box_last.data_ptr() + 24 * r, self.angle_last.data_ptr() + 4 * r
                e.grad_boxes_pred, e.grad_angles_pred = self.d_boxes_pred.data_ptr() + 32 * a, self.d_angles_pred.data_ptr() + 4 * na * a
                e.n_angle, e.ld_gb, e.beta = na, 8, 2.0
        self._place_tab = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(dev)
        self.cls = cls
        self.chan, self.dch = scenes[0].chan, scenes[0].dch
        for sc in scenes[1:]:
            if not (torch.equal(sc.chan, self.chan) and torch.equal(sc.dch, self.dch)):
                raise _lib.SlnError("rooms of one batch must share the class tables")
        self.S = S
        self.scene_ws = torch.empty(int(L.sln_scene_workspace_bytes(R, self.F2, S)), dtype=torch.uint8, device=dev)
        self.image = torch.zeros(R, DR.N_SCENE_CHANNELS, S, S, **f32)         # (planes flagged dead are never written, nor read)
        self.g_image = torch.zeros(R, DR.N_SCENE_CHANNELS, S, S, **f32)
        self.loss_out = torch.empty(R, 3, **f32)
        # the semantic planes of classes without a visible pixel are zeros, and the scene pass never reads their gradients nor
        # those of such classes' depth-hot planes: the loss skips them (SlnRefineLoss::live_planes, refreshed after every scene pass)
        self.live = torch.full((R, DR.N_SCENE_CHANNELS), 3, dtype=torch.uint8, device=dev)
        self.null_mask = None
        # (only when the loss takes the flags for this geometry - image sizes below the pooled size do not: it would read planes
        #  the sparse scene pass leaves unwritten)
        if not os.environ.get("SLN_REFINE_ALL_PLANES") and L.sln_refine_loss_live_ok(C.byref(self.loss.desc)):
            self.loss.desc.live_planes = self.live.data_ptr()
            if not os.environ.get("SLN_REFINE_NULL_MASK_APART"):       # (lab switch: the loss computes the null mask itself)
                self.null_mask = torch.zeros(R, S, S, dtype=torch.uint8, device=dev)
                self.loss.desc.null_mask = self.null_mask.data_ptr()
            if not os.environ.get("SLN_REFINE_POOL_ONES"):             # (lab switch: pool the constant planes per room)
                self._pooled_ones = self.loss.pooled_ones()
                self.loss.desc.pooled_ones = self._pooled_ones.data_ptr()
        self.one = torch.ones(1, **f32)
        self.losses = torch.zeros(max(self.iters, 1), R, **f32)
        rg = model.decoder_param_ranges()
        nf = int(L.sln_vae_group_fused_params(self._group, None, None, 0))
        if nf > 0:                                                   # ... minus the tensors the wgrad launches step themselves
            ptrs, lens = (C.c_void_p * nf)(), (C.c_int64 * nf)()
            L.sln_vae_group_fused_params(self._group, ptrs, lens, nf)
            base = self.params.data_ptr()
            offs = [(int(ptrs[i]) - base) // 4 for i in range(nf)]
            al = 64 if all(o % 64 == 0 for o in offs) else 4         # the flat layout pads every tensor to 64 floats (nothing lives in the pad)
            cut = sorted((o, -(-int(lens[i]) // al) * al) for i, o in enumerate(offs))
            out = []
            for a, n in rg:
                pos, end = a, a + n
                for c0, cn in cut:
                    if c0 + cn <= pos or c0 >= end:
                        continue
                    if c0 % 4 or c0 < pos:
                        raise _lib.SlnError("a fused parameter tensor does not start on a 16-byte boundary of the decoder run")
                    if c0 > pos:
                        out.append((pos, c0 - pos))
                    pos = c0 + cn
                if pos < end:
                    out.append((pos, end - pos))
            rg = out
        self._sgd_off = (C.c_int64 * len(rg))(*[a for a, _ in rg])
        self._sgd_len = (C.c_int64 * len(rg))(*[b for _, b in rg])
        self._n_rg = len(rg)
        self.noise = torch.zeros(N, **f32)
        self._graph = None
        self.k = 0
        # the side stream of the wgrads / the scene backward's depth chain is PROBED for real overlap with the caller's stream, which
        # synchronises that stream: here, at set-up, not inside the first iteration's asynchronous calls (csrc/streams.hip)
        if not torch.cuda.is_current_stream_capturing():
            L.sln_side_stream_prepare(st)
        tick("tables, buffers")
        if self.iters > 0:
            self._first_iterate_sizes()
            tick("first iterate's sizes")

    def _first_iterate_sizes(self):
        """The size penalty holds every object to the size of the FIRST iterate (testing/test_render_refine.py:319-327: ``size_infos`` is
        what the first ``mesh_render_func`` call on ``boxes_pred`` returns; the sizes of the target render are ``unused_sizes``): one
        decoder + head + placement forward of all rooms with iteration 0's z, parameters and noise row, whose ``sizes`` become the
        rooms' targets.  Iteration 0 then measures its own sizes against themselves: a size loss of exactly 0 with zero gradient - the
        reference's ``size_loss = 0.0`` of a first call."""
        L, st, P = _lib.lib(), _lib.current_stream_ptr(), _lib.ptr
        _lib.check(L.sln_vae_group_decoder(self._group, st), "sln_vae_group_decoder")
        if not self._fused_head:
            _lib.check(L.sln_refine_head_forward_rooms(self.N, self.model.Nangle, P(self.room_of_row), P(self.last_row), P(self.boxes_pred),
                                                       P(self.angles_pred), P(self.noise_all[0]), P(self.box_last), P(self.angle_last), 2.0,
                                                       P(self.boxes), P(self.idx), st), "sln_refine_head_forward_rooms")
        _lib.check(L.sln_place_forward_rooms(P(self._place_tab), self.R, self.F2 // 2, st), "sln_place_forward_rooms")
        for r, sc in enumerate(self.scenes):
            if sc.n_vis:
                self._size_targets[r].copy_(self.sizes[r, :sc.n_vis])

    def launches(self):
        f, b, s1 = C.c_int(0), C.c_int(0), C.c_int(0)
        _lib.check(_lib.lib().sln_vae_group_launches(self._group, C.byref(f), C.byref(b), C.byref(s1)), "sln_vae_group_launches")
        return dict(decoder_forward=f.value, decoder_backward=b.value, single_room_fallbacks=s1.value)

    def _iteration(self, noise, out):
        """one iteration of every room on the current stream; ``noise`` [N], ``out`` [R] receives the rooms' losses"""
        L, st, P = _lib.lib(), _lib.current_stream_ptr(), _lib.ptr
        N, R, na, S = self.N, self.R, self.model.Nangle, self.S
        _lib.check(L.sln_vae_group_decoder(self._group, st), "sln_vae_group_decoder")
        if not self._fused_head:
            _lib.check(L.sln_refine_head_forward_rooms(N, na, P(self.room_of_row), P(self.last_row), P(self.boxes_pred), P(self.angles_pred), P(noise),
                                                       P(self.box_last), P(self.angle_last), 2.0, P(self.boxes), P(self.idx), st),
                       "sln_refine_head_forward_rooms")
        _lib.check(L.sln_place_forward_rooms(P(self._place_tab), R, self.F2 // 2, st), "sln_place_forward_rooms")
        rl = self.loss
        if rl.desc.live_planes:
            _lib.check(L.sln_scene_forward_live(P(self.faces), P(self.cls), R, self.F2, S, self.chan.numel(), P(self.chan), P(self.dch), 0.1, 0.001, 100.0,
                                                1e-3, P(self.scene_ws), P(self.image), P(self.live), P(self.null_mask), st), "sln_scene_forward_live")
        else:
            _lib.check(L.sln_scene_forward(P(self.faces), P(self.cls), R, self.F2, S, self.chan.numel(), P(self.chan), P(self.dch), 0.1, 0.001, 100.0,
                                           1e-3, P(self.scene_ws), P(self.image), st), "sln_scene_forward")
        _lib.check(L.sln_refine_loss_forward(rl.desc, P(self.image), P(rl.target_depth), P(rl.labels), P(rl.inv_count), P(rl.ws), P(self.loss_out), st),
                   "sln_refine_loss_forward")
        torch.add(self.loss_out[:, 0], self.size_loss, alpha=2.0, out=out)
        _lib.check(L.sln_refine_loss_backward(rl.desc, P(rl.ws), P(self.one), P(self.g_image), st), "sln_refine_loss_backward")
        _lib.check(L.sln_scene_backward(P(self.faces), P(self.cls), R, self.F2, S, self.chan.numel(), P(self.chan), P(self.dch), 1e-3, P(self.scene_ws),
                                        P(self.g_image), P(self.g_faces), st), "sln_scene_backward")
        _lib.check(L.sln_place_backward_rooms(P(self._place_tab), R, self.n_max, st), "sln_place_backward_rooms")
        if not self._fused_head:
            _lib.check(L.sln_refine_head_backward_rooms(N, na, P(self.room_of_row), P(self.last_row), P(self.angles_pred), P(self.g_boxes), P(self.g_idx),
                                                        2.0, P(self.d_boxes_pred), 8, P(self.d_angles_pred), st), "sln_refine_head_backward_rooms")
        _lib.check(L.sln_vae_group_decoder_backward(self._group, st), "sln_vae_group_decoder_backward")
        _lib.check(L.sln_refine_sgd_rooms(P(self.params), P(self.grads), R, self.params.shape[1], self._sgd_off, self._sgd_len, self._n_rg,
                                          (self.lr / 10.0) * 1.1, P(self.z), P(self.dz), self.z.numel(), 2e-4 * 1.1, st), "sln_refine_sgd_rooms")

    def run(self, iters=None, capture=False):
        """``iters`` more iterations (default: all that remain).  ``capture``: one iteration recorded into a hipGraph and replayed -
        a LINEAR graph: inside a capture the library keeps its side work (wgrads, the depth chain of the scene backward) on the
        captured stream, because this runtime replays forked graphs node by node from the host without overlapping the branches.
        What a replay buys is the host (0.09 ms of enqueue per iteration instead of 0.3-0.4 ms); the GPU time is that of the one-stream
        order, ~5 % above the eager loop with its side streams (16 rooms: 1.66 against 1.56-1.59 ms) - eager is the default."""
        n = (self.iters - self.k) if iters is None else int(iters)
        if self.k + n > self.iters:
            raise ValueError("RefineBatch was built for %d iterations (the noise of every iteration is drawn at construction)" % self.iters)
        scratch = self.loss_out.new_empty(self.R)
        for _ in range(n):
            k = self.k
            if capture:
                if not self._fused_head:
                    self.noise.copy_(self.noise_all[k])
                if self._graph is None:
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):                  # warm-up outside the capture (lazy kernel attributes)
                        self._iteration(self.noise, self.losses[k])
                    torch.cuda.current_stream().wait_stream(side)
                    self._graph = torch.cuda.CUDAGraph()
                    self._graph_out = scratch
                    with torch.cuda.graph(self._graph):
                        self._iteration(self.noise, self._graph_out)
                else:
                    self._graph.replay()
                    self.losses[k].copy_(self._graph_out)
            else:
                self._iteration(self.noise_all[k], self.losses[k])
            self.k += 1
        return self.losses[:self.k]

    def results(self):
        """[(boxes_full [n,6], angle idx [n]) per room] of the last iteration"""
        return [(self.boxes[a:a + n], self.idx[a:a + n]) for a, n in zip(self.row0, self.rows)]

    def close(self):
        L = _lib.lib()
        if getattr(self, "_group", None) is not None:
            torch.cuda.synchronize()
            L.sln_vae_group_destroy(self._group)
            self._group = None
        for e in getattr(self, "_engines", []):
            L.sln_vae_destroy(e[0])
        self._engines = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def finetune_vae_fast_batch(model, rooms, iters=60, bank=None, learning_rate=1e-4, noise_seed=13, image_size=256, capture=False):
    """``finetune_vae_fast`` for R rooms at once (see ``RefineBatch``): every room from ``model``'s parameters, its own z, its own
    noise stream (seeded like a single-room call).  -> (losses [iters, R] on the device, [(boxes, angle idx) per room])."""
    rb = RefineBatch(model, rooms, bank=bank, learning_rate=learning_rate, noise_seed=noise_seed, image_size=image_size, iters=iters)
    try:
        losses = rb.run(capture=capture).clone()
        res = [(b.clone(), i.clone()) for b, i in rb.results()]
    finally:
        rb.close()
    return losses, res


def finetune_vae(model, objs, triples, boxes_gt, angles_gt, attributes, class_names, iters=60, render_fn=None, bank=None,
                 learning_rate=1e-4, noise_seed=13, image_size=256, log=None):
    """finetune_VAE's inner loop for ONE room (test_render_refine.py:265-359) on synthetic meshes.
    Returns the list of per-iteration losses and the final (boxes_pred, angles_idx)."""
    render_fn = render_fn or DR.scene_render
    dev = boxes_gt.device
    bank = bank or MeshBank([n for n in set(class_names) if n not in DO_NOT_VIS], dev)
    model.eval()
    with torch.no_grad():
        mu, logvar = model.encoder(objs, triples, boxes_gt, angles_gt, attributes)
    gen = torch.Generator(device="cpu").manual_seed(noise_seed)
    z = (mu + torch.randn(mu.shape, generator=gen).to(dev) * torch.exp(0.5 * logvar)).detach().requires_grad_(True)
    room_box = boxes_gt[-1].detach().clone()
    v, f, ranges, sizes, _ = assemble_scene(boxes_gt, angles_gt.float(), class_names, bank, room_box)
    with torch.no_grad():
        target = render_fn(v, f, ranges, room_box, image_size=image_size)
    labels = target_labels(target)
    size_target = None                   # the sizes of the FIRST iterate, not the target's (test_render_refine.py:319-327: the target render's
    #                                      sizes are "unused_sizes"; size_infos is what the first render of boxes_pred returns)
    losses = []
    for k in range(iters):
        opt = torch.optim.SGD([{'params': [z]}, {'params': list(model.parameters()), 'lr': learning_rate / 10.0}], lr=2e-4,
                              nesterov=True, momentum=0.1)
        boxes_pred, angles_pred = model.decoder(z, objs, triples, attributes)
        boxes_pred.register_hook(fix_grad)
        boxes_pred = torch.cat([boxes_pred[:-1], boxes_gt[-1:]], 0)
        idx = softargmax(angles_pred, sum_dim=1) + torch.randn(angles_pred.shape[0], generator=gen).to(dev) / 10.0
        idx.register_hook(quad_grad)
        idx = torch.cat([idx[:-1], angles_gt[-1:].float()], 0)
        v, f, ranges, sizes_k, size_loss = assemble_scene(boxes_pred, idx, class_names, bank, room_box, size_target)
        if size_target is None:
            size_target = [s.clone() for s in sizes_k]
        image = render_fn(v, f, ranges, room_box, image_size=image_size)
        loss, dl, sl = refinement_loss(image, target, labels, size_loss)
        opt.zero_grad()
        loss.backward()
        opt.step()
        if hasattr(model, "params_changed"):
            model.params_changed()
        losses.append(float(loss.detach()))
        if log:
            log("iter %d: loss %.4f (depth %.4f, semantic %.4f)" % (k, losses[-1], float(dl), float(sl)))
    return losses, (boxes_pred.detach(), idx.detach())
