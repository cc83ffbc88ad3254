"""Synthetic scene-graph batches with the tuple layout of the reference's ``suncg_collate_fn``
(data/suncg_dataset.py:295-337): objects of one room are contiguous, the room node (class 0) is
last, every object has an ``__in_room__`` (predicate 0) triple to it (:207-212), and triples carry
global row ids.  Used by bench.py / train.py because the SUNCG metadata is not distributable."""
import numpy as np
import torch


def scene_graph_batch(n_graphs, objs_per_graph=32, triples_per_graph=64, seed=0, num_objs=32, num_preds=16,
                      num_attrs=5, n_angle=24, box_dim=6, device="cpu"):
    rng = np.random.default_rng(seed)
    n, tt = objs_per_graph, triples_per_graph
    n_rand = tt - (n - 1)
    if n < 2 or n_rand < 0:
        raise ValueError("need >= 2 objects and >= objs-1 triples per graph")
    objs = rng.integers(1, num_objs, size=(n_graphs, n)); objs[:, -1] = 0
    s = rng.integers(0, n - 1, size=(n_graphs, n_rand))
    o = (s + rng.integers(1, max(n - 1, 2), size=(n_graphs, n_rand))) % (n - 1) if n > 2 else s.copy()
    p = rng.integers(1, num_preds, size=(n_graphs, n_rand))
    off = (np.arange(n_graphs) * n)[:, None]
    rand = np.stack([s + off, p, o + off], -1)
    room = np.stack([np.arange(n - 1)[None] + off, np.zeros((n_graphs, n - 1), np.int64),
                     np.full((n_graphs, n - 1), n - 1) + off], -1)
    triples = np.concatenate([rand, room], 1).reshape(-1, 3)
    lo = rng.uniform(0.0, 0.7, size=(n_graphs, n, 3)); hi = lo + rng.uniform(0.05, 0.3, size=(n_graphs, n, 3))
    boxes = np.concatenate([lo, hi], -1); boxes[:, -1] = [0, 0, 0, 1, 1, 1]
    if box_dim == 4:
        boxes = boxes[..., [0, 2, 3, 5]]
    angles = rng.integers(0, n_angle, size=(n_graphs, n))
    attrs = rng.integers(0, num_attrs, size=(n_graphs, n))
    o2i = np.repeat(np.arange(n_graphs), n)
    t2i = np.repeat(np.arange(n_graphs), tt)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt)).to(device)
    return dict(objs=t(objs.reshape(-1), np.int64), triples=t(triples, np.int64),
                boxes=t(boxes.reshape(-1, box_dim), np.float32), angles=t(angles.reshape(-1), np.int64),
                attributes=t(attrs.reshape(-1), np.int64), obj_to_img=t(o2i, np.int64), triple_to_img=t(t2i, np.int64))


def default_vocab(num_objs=32, num_preds=16, num_attrs=5):
    return {"object_idx_to_name": ["__room__"] + ["type%02d" % i for i in range(1, num_objs)],
            "pred_idx_to_name": ["pred%02d" % i for i in range(num_preds)],
            "attrib_idx_to_name": ["attr%d" % i for i in range(num_attrs)]}


# ----------------------------------------------------------------------------------------------
# synthetic rooms for the renderer path (BASELINE configs[2]: 16 rooms x ~2k triangles): cuboid furniture
# on the floor + floor / ceiling / three walls, every quad split into a grid of triangles.  Stand-in for the
# SUNCG meshes that models/misc.py retrieves (licensed data, out of scope); same buffers as diff_render.py:344.
# ----------------------------------------------------------------------------------------------
FURNITURE = ['cabinet', 'bed', 'chair', 'sofa', 'table', 'bookshelf', 'desk', 'shelves', 'dresser', 'night_stand',
             'television', 'lamp', 'toilet', 'sink', 'bathtub', 'counter', 'refridgerator', 'mirror', 'picture', 'box',
             'bag', 'books', 'clothes', 'pillow', 'towel', 'paper', 'whiteboard', 'otherprop', 'otherfurniture']


def _grid_quad(p0, du, dv, n):
    a = np.linspace(0, 1, n + 1)
    g = p0[None, None] + a[:, None, None] * du[None, None] + a[None, :, None] * dv[None, None]
    idx = np.arange((n + 1) * (n + 1)).reshape(n + 1, n + 1)
    q = np.stack([idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]], -1).reshape(-1, 4)
    return g.reshape(-1, 3), np.concatenate([q[:, [0, 1, 2]], q[:, [0, 2, 3]]], 0)


def _grid_cuboid(lo, hi, n):
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    d = hi - lo
    ex, ey, ez = np.array([d[0], 0, 0]), np.array([0, d[1], 0]), np.array([0, 0, d[2]])
    vs, fs, off = [], [], 0
    for p0, du, dv in [(lo, ey, ex), (lo + ez, ex, ey), (lo, ex, ez), (lo + ey, ez, ex), (lo, ez, ey), (lo + ex, ey, ez)]:
        v, f = _grid_quad(p0, du, dv, n)
        vs.append(v); fs.append(f + off); off += v.shape[0]
    return np.concatenate(vs), np.concatenate(fs)


def synthetic_room(seed, n_objects=12, target_faces=2000, room=(4.0, 2.7, 5.0)):
    """-> vertices [V,3] f32, faces [F,3] i32, class_ranges {class: [[a,b],..]} (all 32 classes as keys), room_box [6]."""
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
    sub = 2 if target_faces >= 1500 else 1
    for nm in rng.choice(FURNITURE, size=n_objects, replace=False):
        size = rng.uniform([0.4, 0.3, 0.4], [1.2, 1.6, 1.2])
        pos = rng.uniform([0.1, 0.0, 0.3], [room[0] - size[0] - 0.1, 0.0, room[2] - size[2] - 1.2])
        add(*_grid_cuboid(pos, pos + size, sub), str(nm))
    shell = [("floor", np.zeros(3), np.array([0, 0, room[2]]), np.array([room[0], 0, 0])),
             ("ceiling", np.array([0, room[1], 0]), np.array([room[0], 0, 0]), np.array([0, 0, room[2]])),
             ("wall", np.zeros(3), np.array([room[0], 0, 0]), np.array([0, room[1], 0])),
             ("wall", np.zeros(3), np.array([0, room[1], 0]), np.array([0, 0, room[2]])),
             ("wall", np.array([room[0], 0, 0]), np.array([0, 0, room[2]]), np.array([0, room[1], 0]))]
    per = max(1, int(round(np.sqrt(max(target_faces - foff, 10) / (2.0 * len(shell))))))
    for nm, p0, du, dv in shell:
        add(*_grid_quad(p0, du, dv, per), nm)
    return (np.concatenate(vs).astype(np.float32), np.concatenate(fs).astype(np.int32), ranges,
            np.array([0, 0, 0, room[0], room[1], room[2]], np.float32))


def scene_rooms(n_rooms, objs_per_room=31, seed=0, n_classes=31):
    """Synthetic room table in the structure of data_rot_*.json (reference data/suncg_dataset.py:84-90): raw boxes of
    floor-standing / stacked / nested objects in a room, one class and rotation bin each.
    -> (rooms, object_idx_to_name, size_data, size_data_30) for ``SuncgDataset.from_tables``."""
    rng = np.random.default_rng(seed)
    names = ["__room__"] + ["type%02d" % i for i in range(1, n_classes + 1)]
    rooms = []
    for _ in range(n_rooms):
        room = rng.uniform([3, 2.5, 3], [7, 3.2, 8])
        n = objs_per_room
        size = rng.uniform([0.2, 0.2, 0.2], [1.6, 1.4, 1.6], size=(n, 3))
        lo = rng.uniform(0, 1, size=(n, 3)) * np.maximum(room - size, 0.1)
        lo[:, 1] = 0.0
        for i in range(1, n):                                   # every fifth object stands exactly on an earlier one
            if i % 5 == 0:
                j = int(rng.integers(0, i))
                size[i, 0] = min(size[i, 0], size[j, 0]); size[i, 2] = min(size[i, 2], size[j, 2])
                lo[i] = [lo[j, 0] + (size[j, 0] - size[i, 0]) / 2, lo[j, 1] + size[j, 1], lo[j, 2] + (size[j, 2] - size[i, 2]) / 2]
        rooms.append(dict(objs=rng.integers(1, n_classes + 1, size=n).tolist(), boxes=np.concatenate([lo, lo + size], 1).astype(np.float32),
                          rot=rng.integers(0, 24, size=n).tolist(), bbox=room.astype(np.float32)))
    size_data = {nm: [[0.1, float(rng.uniform(0.1, 0.4))], float(rng.uniform(0.001, 0.02))] for nm in names[1::2]}
    size_data_30 = {nm: dict(height_7=float(rng.uniform(0.25, 0.5)), height_3=float(rng.uniform(0.0, 0.25)),
                             volume_7=float(rng.uniform(0.01, 0.03)), volume_3=float(rng.uniform(0.0005, 0.01))) for nm in names[1::2]}
    return rooms, names, size_data, size_data_30


def pack_rooms(rooms, device="cuda"):
    """Pad a list of ``synthetic_room`` tuples (different V / F) into one render batch for ``diff_render.scene_render_batch``:
    near-plane cull + class lookup per room (diff_render.py:346-356,372-376), fill_back duplicates, faces padded with
    degenerate triangles of class -1.  -> dict(V [B,Vmax,3], F [B,Fmax,3] i32, C [B,Fmax] i32, chan, dch, K, R, t, tris)."""
    import importlib
    import torch
    DR = importlib.import_module("3d_sln_amd.host.diff_render")
    Vmax = max(r[0].shape[0] for r in rooms)
    prepared, Fmax, tris = [], 0, 0
    for V, F, ranges, box in rooms:
        K, R, t = DR.get_cam_mat(torch.from_numpy(box), "cpu")
        faces, cls, classes, chan, dch = DR.cull_and_classify(torch.from_numpy(V)[None], torch.from_numpy(F)[None], ranges, R, t)
        tris += faces.shape[1]
        faces = torch.cat((faces, faces[:, :, [2, 1, 0]]), 1)[0]; cls = torch.cat((cls, cls))
        vp = torch.zeros(Vmax, 3); vp[:V.shape[0]] = torch.from_numpy(V)
        prepared.append((vp, faces, cls, K[0], R[0], t[0])); Fmax = max(Fmax, faces.shape[0])
    B = len(rooms)
    Fb = torch.zeros(B, Fmax, 3, dtype=torch.int32); Cb = torch.full((B, Fmax), -1, dtype=torch.int32)
    for i, p in enumerate(prepared):
        Fb[i, :p[1].shape[0]] = p[1]; Cb[i, :p[2].shape[0]] = p[2]
    return dict(V=torch.stack([p[0] for p in prepared]).to(device), F=Fb.to(device), C=Cb.to(device),
                chan=torch.tensor(chan, dtype=torch.int32, device=device), dch=torch.tensor(dch, dtype=torch.int32, device=device),
                K=torch.stack([p[3] for p in prepared]).to(device), R=torch.stack([p[4] for p in prepared]).to(device),
                t=torch.stack([p[5] for p in prepared]).to(device), tris=tris)


def spade_input(batch, crop=256, semantic_nc=41, nz=256, seed=0):
    """Synthetic input of SPADEGenerator4 (the tensor contract of testing/test_SPADE_shade.py:50-76): channel 0 = smooth depth in
    [-1, 1], channels 1.. = one-hot of the argmax of upsampled noise maps; z ~ N(0, 1).  CPU tensors from numpy generators, one
    per image (image i is the same whatever the batch size): bench.py's SPADE leg and the reference-generated fixture
    tests/golden/spade_bench.npz (oracle/gen_golden_spade.py) use the same call."""
    import torch.nn.functional as F
    segs, zs = [], []
    for i in range(batch):
        rng = np.random.default_rng(1000003 * seed + i)
        low = torch.from_numpy(rng.uniform(-1, 1, size=(1, 1, 16, 16)).astype(np.float32))
        depth = F.interpolate(low, size=(crop, crop), mode="bilinear", align_corners=False).clamp(-1, 1)
        noise = torch.from_numpy(rng.standard_normal((1, semantic_nc - 1, 16, 16)).astype(np.float32))
        lab = F.interpolate(noise, size=(crop, crop), mode="bilinear", align_corners=False).argmax(1)
        onehot = F.one_hot(lab, semantic_nc - 1).permute(0, 3, 1, 2).float()
        segs.append(torch.cat([depth, onehot], 1))
        zs.append(torch.from_numpy(rng.standard_normal((1, nz)).astype(np.float32)))
    return torch.cat(segs).contiguous(), torch.cat(zs).contiguous()


def overfit_to_rooms(model, rooms, steps=400, lr=2e-3, kl_weight=1e-3, settle=80):
    """``steps`` fused Adam steps of the VAE on the collated batch of ``rooms`` (dicts as ``refine.RefineBatch`` takes them: object
    rows room-normalised, last row the room's metric box - the encoder sees that row in training and in refinement alike).  Afterwards the decoder places every room's objects near their targets -
    what a trained checkpoint does (testing/test_render_refine.py:250-263 reloads one per trial) - so that a refinement benchmark
    renders the furniture: a randomly initialised decoder predicts near-degenerate boxes and the iterate shows the empty room.
    The refinement evaluates BatchNorm on its RUNNING statistics (model.eval()): ``settle`` more steps with lr = 0 let them converge
    to the final weights' batch statistics (after 400 steps at 2e-3 the 10-step moving averages lag the weights, and five stacked
    gconv layers amplify the mismatch: boxes of magnitude 1e2..1e5 in eval mode with a train-mode loss of 0.03).
    Returns the last step's losses [bbox, angle, KL, total] (device tensor)."""
    import torch
    objs, triples, boxes, angles, attrs, off = [], [], [], [], [], 0
    for rm in rooms:
        n = int(rm["objs"].shape[0])
        nb = rm["boxes"].detach().clone().float()        # as the dataset hands them out (data/suncg_dataset.py:113-143): object rows
        #                                                   room-normalised, the room row [0, 0, 0, W, H, D] in metres - the refinement
        #                                                   feeds the encoder the same rows (test_render_refine.py:273)
        t = rm["triples"].clone()
        t[:, 0] += off; t[:, 2] += off
        objs.append(rm["objs"]); triples.append(t); boxes.append(nb); angles.append(rm["angles"].long()); attrs.append(rm["attributes"])
        off += n
    objs, triples, boxes, angles, attrs = (torch.cat(x) for x in (objs, triples, boxes, angles, attrs))
    was_training = model.training
    model.train()
    losses = None
    for _ in range(int(steps)):
        losses = model.train_step(objs, triples, boxes, angles, attrs, kl_weight=kl_weight, lr=lr, use_graph=False)
    for _ in range(int(settle) if steps > 0 else 0):
        losses = model.train_step(objs, triples, boxes, angles, attrs, kl_weight=kl_weight, lr=0.0, use_graph=False)
    model.train(was_training)
    return losses
