"""CPU restatement of the two torch-side pieces of the refinement loop that the product implements as HIP kernels.
TEST INFRASTRUCTURE ONLY (tests/ import it as the checker; the product never does).

  place_object / place_scene   models/diff_render.py:76-165 - box -> centre / size, theta = -angle * 2 pi / 24, isotropic
                               scale = min(size / model_size), R_y, translation, transformed vertices, size loss
  psp_pool / target_labels /   testing/test_render_refine.py:192-215 (PSP_pool_new), :328-356 (null fill, L1 * 0.5,
  refinement_loss              cross-entropy / 800 per scale, 100 * depth + 100 * semantic + 2 * size)

  place_shell                  models/diff_render.py:166-342 - wall sub-meshes (max-ratio scale, the bad-wall rule :203-213), floor, ceiling
  render_room                  mesh_render_func as a whole on table arrays (:48-435: frozen room box, size / drift penalties, buffers,
                               cull, 33 passes through oracle/raster_ref.py, channel layout)
  softargmax / fix_grad /      testing/test_render_refine.py:20-25, 217-228
  quad_grad
  refine_loop                  the ``for k in range(Niter_train)`` statement of finetune_VAE (:284-359)

Parity status: the CALLERS ARE PINNED, GIVEN THE RASTERIZER.  testing/test_render_refine.py and models/diff_render.py cannot be
imported (neural_renderer / pymesh / pywavefront / SUNCG metadata at module level, models/misc.py:7-31), but
oracle/gen_golden_refine.py executes their function / class / loop nodes from the source text (ast) with the mesh I/O replaced by
tables and ``nr`` replaced by oracle/raster_ref.py::RefRenderer, and writes tests/golden/refine_{helpers,scene,loop}.npz;
tests/test_oracle_refine_golden.py holds every function below to those fixtures.  The rasterizer itself stays PARITY UNPINNED
(third-party source absent): what is pinned is everything around it.
"""
import math

import torch
import torch.nn.functional as F

DO_NOT_VIS = ["wall", "ceiling", "floor", "person", "door", "window", "curtain", "blinds"]     # diff_render.py:93


def place_object(box, angle, room_ext, model_v, model_bbox_min, model_bbox_max):
    """diff_render.py:78-81,104,117-137: one object's vertices in room coordinates, and its size."""
    bbox_min, bbox_max = box[:3] * room_ext, box[3:] * room_ext                           # :78-79
    obj_center, obj_size = (bbox_max + bbox_min) / 2, bbox_max - bbox_min                 # :80-81
    theta = -angle * (2 * float(math.pi) / 24)                                            # :104
    model_size = model_bbox_max - model_bbox_min                                          # :110
    model_center = (model_bbox_min + model_bbox_max) / 2.0                                # :113
    scale = min([obj_size[0] / model_size[0], obj_size[1] / model_size[1], obj_size[2] / model_size[2]])     # :117
    rot = torch.eye(3, dtype=box.dtype)                                                   # :118-124
    cos_theta, sin_theta = torch.cos(theta), torch.sin(theta)
    rot = rot.clone()
    rot[0, 0] = cos_theta; rot[0, 2] = sin_theta; rot[2, 0] = -sin_theta; rot[2, 2] = cos_theta
    trans = obj_center - scale * torch.matmul(rot, model_center)                          # :126
    trans_4x4 = torch.eye(4, dtype=box.dtype); trans_4x4[:3, -1] = trans                  # :127-128
    rot_4x4 = torch.eye(4, dtype=box.dtype); rot_4x4[:3, :3] = rot * scale                # :129-130
    final = torch.matmul(trans_4x4, rot_4x4)[:3]                                          # :131
    v = torch.cat((model_v.t(), torch.ones(1, model_v.shape[0], dtype=box.dtype)), dim=0)            # :133-137
    return torch.t(torch.matmul(final, v)), obj_size


def place_scene(boxes, angles, class_names, models, obj_size_target=None):
    """The object loop of mesh_render_func (:76-159) over ``models`` = {class: dict(v, bbox_min, bbox_max)}: the concatenated
    object vertices (placement order), the sizes, and the size loss (:98-100).  The room row (last) is not placed."""
    verts, sizes = [], []
    size_loss = boxes.new_zeros(())
    k = 0
    for i, name in enumerate(class_names[:-1]):
        if name in DO_NOT_VIS or name not in models:
            continue
        m = models[name]
        v, size = place_object(boxes[i], angles[i], boxes[-1][3:], m["v"], m["bbox_min"], m["bbox_max"])
        if obj_size_target is not None:
            size_loss = size_loss + F.mse_loss(size, obj_size_target[k])
        verts.append(v); sizes.append(size)
        k += 1
    return (torch.cat(verts) if verts else boxes.new_zeros(0, 3)), sizes, size_loss


def psp_pool(feats, sizes=(32, 48, 64, 96), as_list=False):
    """PSP_pool_new (test_render_refine.py:192-215): nn.Upsample(size, 'bilinear', align_corners=True) per stage, then
    F.upsample(..., size=max, mode='bilinear') (align_corners defaults to False)."""
    outs = [F.interpolate(F.interpolate(feats, size=(s, s), mode='bilinear', align_corners=True), size=(sizes[-1], sizes[-1]),
                          mode='bilinear', align_corners=False) for s in sizes]
    return outs if as_list else torch.cat(outs, 1)


def target_labels(target, sizes=(32, 48, 64, 96)):
    """:336-343: per scale argmax of the pooled one-hot block, -100 where nothing is there."""
    out = []
    for pooled in psp_pool(target[:, 1:41], sizes, as_list=True):
        flat = torch.argmax(pooled, dim=1, keepdim=True)
        flat[torch.sum(pooled, dim=1, keepdim=True) < 0.5] = -100
        out.append(flat.detach())
    return out


def refinement_loss(iter_image, target, target_container, size_loss, sizes=(32, 48, 64, 96)):
    """:328-356.  Returns (loss_val, depth_loss, semantic_loss)."""
    iter_image = iter_image.clone()
    iter_image[:, -1][torch.sum(iter_image[:, 41:], dim=1) < 0.5] = 1.0                   # :329 fill in null regions
    scaled_target_depth = psp_pool(target[:, 41:], sizes)                                 # :331
    scaled_input_depth = psp_pool(iter_image[:, 41:], sizes)                              # :332
    train_labels_pooled = psp_pool(iter_image[:, 1:41], sizes, as_list=True)              # :335
    semantic_loss = iter_image.new_zeros(())
    for scale_idx in range(len(train_labels_pooled)):                                     # :346-347
        semantic_loss = semantic_loss + F.cross_entropy(train_labels_pooled[scale_idx], target_container[scale_idx][:, 0, :, :].long()) / 800.0
    depth_loss = F.l1_loss(scaled_input_depth, scaled_target_depth) * 0.5                 # :348 (orig_scaler 0.5)
    loss_val = depth_loss * 100 + semantic_loss * 100 + size_loss * 2.0                   # :350-352
    return loss_val, depth_loss, semantic_loss


# ----------------------------------------------------------------------------------------------------------------------------
# the fixtures' vocabulary and table layout (tests/golden/refine_scene.npz, refine_loop.npz; written by oracle/gen_golden_refine.py)
# ----------------------------------------------------------------------------------------------------------------------------
FIXTURE_VOCAB = ['cabinet', 'bed', 'chair', 'sofa', 'table', 'bookshelf', 'desk', 'shelves', 'dresser', 'night_stand', 'television', 'lamp',
                 'toilet', 'sink', 'bathtub', 'counter', 'refridgerator', 'mirror', 'picture', 'box', 'bag', 'books', 'clothes', 'pillow',
                 'towel', 'paper', 'whiteboard', 'door', 'window']     # 27 furniture classes + two that mesh_render_func skips (:93-97)


def load_tables(npz, prefix="tab:"):
    """-> dict(vocab, models {class: v, f, bbox_min, bbox_max (numpy)}, shell {wall_v, wall_f [list], wall_bbox, floor_*, ceil_*})"""
    import numpy as np
    models = {}
    for name in FIXTURE_VOCAB:
        bb = np.asarray(npz["%smodel:%s:bbox" % (prefix, name)])
        models[name] = dict(v=np.asarray(npz["%smodel:%s:v" % (prefix, name)]), f=np.asarray(npz["%smodel:%s:f" % (prefix, name)]), bbox_min=bb[0], bbox_max=bb[1])
    wall_f, i = [], 0
    while "%swall_f:%d" % (prefix, i) in npz:
        wall_f.append(np.asarray(npz["%swall_f:%d" % (prefix, i)])); i += 1
    shell = dict(wall_v=np.asarray(npz[prefix + "wall_v"]), wall_f=wall_f, wall_bbox=np.asarray(npz[prefix + "wall_bbox"]),
                 floor_v=np.asarray(npz[prefix + "floor_v"]), floor_f=np.asarray(npz[prefix + "floor_f"]), floor_bbox=np.asarray(npz[prefix + "floor_bbox"]),
                 ceil_v=np.asarray(npz[prefix + "ceil_v"]), ceil_f=np.asarray(npz[prefix + "ceil_f"]))
    return dict(vocab=list(FIXTURE_VOCAB), models=models, shell=shell)


def softargmax(logits, beta=2.0):
    """test_render_refine.py:20-25 over dim 1: expectation of the 1-based index under softmax(beta * x), minus one"""
    pos = torch.arange(1, logits.shape[1] + 1, dtype=logits.dtype)
    return (F.softmax(logits * beta, dim=1) * pos).sum(dim=1) - 1.0


def fix_grad(g):
    """:217-222 - both halves of a box gradient replaced by their mean"""
    half = g[:, 3:] / 2.0 + g[:, :3] / 2.0
    return torch.cat((half, half), dim=1)


def quad_grad(g):
    """:224-227"""
    return g.detach() * 4.0


def _fit(v, scale, model_center, center):
    """:188-200 - translation(center - scale * model_center) x scale, as the reference's product of two 4x4 matrices applied to [v; 1]"""
    move, grow = torch.eye(4, dtype=v.dtype), torch.eye(4, dtype=v.dtype)
    move[:3, -1] = center - scale * torch.matmul(torch.eye(3, dtype=v.dtype), model_center)
    grow[:3, :3] = torch.eye(3, dtype=v.dtype) * scale
    rows = torch.cat((v.t(), torch.ones(1, v.shape[0], dtype=v.dtype)), dim=0)
    return torch.t(torch.matmul(torch.matmul(move, grow)[:3], rows))


def place_shell(room_ext, shell, dtype=torch.float32):
    """diff_render.py:166-342 -> [(class, vertices [n,3], faces [m,3] int64), ...] in buffer order: the kept wall sub-meshes, the floor,
    the ceiling.  ``room_ext`` = boxes[-1][3:]."""
    t = lambda a: torch.as_tensor(a, dtype=dtype)
    ext = t(room_ext)
    out = []
    # walls (:166-229): centred in the room, isotropic scale = the LARGEST of the three ratios
    lo, hi = t(shell["wall_bbox"][0]), t(shell["wall_bbox"][1])
    scale = torch.max(ext / (hi - lo))
    wv = _fit(t(shell["wall_v"]), scale, (lo + hi) / 2.0, ext / 2.0)
    for f in shell["wall_f"]:
        f = torch.as_tensor(f).long()
        too_close = wv[:, 2][f].max() > 0.9 * ext[2]                                           # :203-204
        mid = wv[:, 0][f].mean()                                                                # :205-209 (a mean over face corners)
        if bool(too_close) and bool(mid > 0.1 * ext[0]) and bool(mid < 0.9 * ext[0]):           # :211-213
            continue
        if f.shape[0] > 0:
            out.append(("wall", wv, f))
    # floor (:236-283): x / z ratios only, y centre 0
    lo, hi = t(shell["floor_bbox"][0]), t(shell["floor_bbox"][1])
    msize = hi - lo
    scale = torch.max(ext[0] / msize[0], ext[2] / msize[2])
    center = ext / 2.0 * t([1.0, 0.0, 1.0])
    out.append(("floor", _fit(t(shell["floor_v"]), scale, (lo + hi) / 2.0, center), torch.as_tensor(shell["floor_f"]).long()))
    # ceiling (:285-336): bounding box of its own vertices, lifted so that its lower side sits on the room's height
    cv = t(shell["ceil_v"])
    hi, lo = cv.max(0).values, cv.min(0).values
    msize = hi - lo
    scale = torch.max(ext[0] / msize[0], ext[2] / msize[2])
    center = ext / 2.0
    center = torch.stack([center[0], 0.5 * (scale * msize)[1] + ext[1], center[2]])
    out.append(("ceiling", _fit(cv, scale, (lo + hi) / 2.0, center), torch.as_tensor(shell["ceil_f"]).long()))
    return out


def render_room(boxes, angles, class_names, tables, image_size=256, room_box=None, size_target=None):
    """mesh_render_func (:48-435) for one room on table arrays.  ``boxes`` [n,6] (room row last), ``angles`` [n], ``class_names`` the
    rows' classes; ``room_box`` = the cached "box_info" of a later call (:55-57), ``size_target`` = the cached sizes [n_vis,3] + the
    cached room row, as (sizes, room_row).  -> (final [1,70,S,S], sizes [n_vis,3] detached, size_loss)"""
    from oracle import raster_ref
    dtype = boxes.dtype
    t = lambda a: torch.as_tensor(a, dtype=dtype)
    old_wall = boxes[-1]
    room = boxes[-1].detach() if room_box is None else t(room_box)
    ext = room[3:]
    names = list(tables["vocab"]) + ["ceiling", "floor", "wall"]                                 # :65-69
    ranges = {c: [] for c in names}
    verts, faces, sizes, voff, foff = [], [], [], 0, 0
    size_loss = boxes.new_zeros(())
    for i, name in enumerate(class_names[:-1]):
        if name in DO_NOT_VIS:                                                                   # :93-97
            continue
        m = tables["models"][name]
        v, size = place_object(boxes[i], angles[i], ext, t(m["v"]), t(m["bbox_min"]), t(m["bbox_max"]))
        if size_target is not None:
            size_loss = size_loss + F.mse_loss(size, t(size_target[0][len(sizes)]))              # :98-100
        sizes.append(size.detach())
        f = torch.as_tensor(m["f"]).long()
        ranges[name].append([foff, foff + f.shape[0]])
        verts.append(v); faces.append(f + voff); voff += v.shape[0]; foff += f.shape[0]
    if size_target is not None:
        size_loss = size_loss + F.mse_loss(old_wall, t(size_target[1]))                          # :160-162
    for name, v, f in place_shell(ext, tables["shell"], dtype):
        ranges[name].append([foff, foff + f.shape[0]])
        verts.append(v); faces.append(f + voff); voff += v.shape[0]; foff += f.shape[0]
    vbuf, fbuf = torch.cat(verts)[None], torch.cat(faces).to(torch.int32)[None]
    final = raster_ref.scene_render(vbuf.float(), fbuf, ranges, room.float(), image_size=image_size)
    return final, (torch.stack(sizes) if sizes else boxes.new_zeros(0, 3)), size_loss


def refine_loop(decoder, params, z, room, tables, noise, image_size=256, learning_rate=1e-4, record=None):
    """finetune_VAE's k loop (:284-359) for one room.  ``decoder(z) -> (boxes_pred [n,6], angles_pred [n,24])`` differentiable w.r.t.
    ``z`` and the tensors of ``params`` (leaf tensors, stepped in place); ``room`` = dict(boxes [n,6], angles [n], class_names);
    ``noise`` [iters, n] the N(0,1) rows of :304.  Every iteration builds a NEW SGD(momentum 0.1, nesterov) (:286), whose first
    step is p -= lr * (1 + 0.1) * grad.  -> list of per-iteration dicts (loss, depth, sem, size, z, boxes, idx, dz)."""
    boxes_gt, angles_gt, names = room["boxes"], room["angles"].float(), room["class_names"]
    with torch.no_grad():
        target, _, _ = render_room(boxes_gt, angles_gt, names, tables, image_size)               # :318-321
    labels = target_labels(target)
    cached, out = None, []
    for k in range(noise.shape[0]):
        boxes_pred, angles_pred = decoder(z)
        boxes_pred.register_hook(fix_grad)                                                       # :294
        boxes_full = torch.cat((boxes_pred[:-1], boxes_gt[-1:]), 0)                              # :297
        idx = softargmax(angles_pred) + noise[k] / 10.0                                          # :299
        idx.register_hook(quad_grad)                                                             # :303
        idx_full = torch.cat((idx[:-1], angles_gt[-1:]), 0)                                      # :304
        if cached is None:                              # first render of the iterate: caches the room row and ITS sizes (:323-327)
            image, sizes, size_loss = render_room(boxes_full, idx_full, names, tables, image_size)
            cached = (boxes_full[-1].detach().clone(), (sizes.clone(), boxes_full[-1].detach().clone()))
        else:
            image, sizes, size_loss = render_room(boxes_full, idx_full, names, tables, image_size, cached[0], cached[1])
        loss, depth, sem = refinement_loss(image, target, labels, size_loss)
        for p in [z] + list(params):
            p.grad = None
        loss.backward()
        dz = z.grad.detach().clone()
        with torch.no_grad():
            z -= 2e-4 * 1.1 * z.grad
            for p in params:
                if p.grad is not None:
                    p -= (learning_rate / 10.0) * 1.1 * p.grad
        rec = dict(loss=float(loss.detach()), depth=float(depth.detach()), sem=float(sem.detach()), size=float(size_loss.detach()),
                   z=z.detach().clone(), boxes=boxes_full.detach().clone(), idx=idx_full.detach().clone(), dz=dz, image=image.detach())
        if record is not None:
            record(k, rec)
        out.append(rec)
    return out
