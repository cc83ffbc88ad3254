"""Golden fixtures for path B's callers, produced by EXECUTING THE REFERENCE'S OWN SOURCE TEXT (build container only).

``models/diff_render.py`` and ``testing/test_render_refine.py`` cannot be imported: at module level they pull in
``neural_renderer`` (un-vendored), ``pymesh``, ``pywavefront``, ``imageio`` and the SUNCG metadata.  Their function / class /
statement nodes can still be run: this script reads the two files (and ``models/misc.py``) as text, takes the nodes below with
``ast`` and ``exec``s them, unmodified, in a namespace that supplies what the module level would have:

    models/diff_render.py       get_cam_mat (:13-46), mesh_render_func (:48-435), nyu_class / inter_out / final_out
    models/misc.py              suncg_retrieve (:34-63), wall_retrieve (:135-149), floor_retrieve (:151-165), get_bbox (:216-217)
    testing/test_render_refine.py   softargmax (:20-25), PSP_pool_new (:192-215), depth_pooler / semantic_pooler_novel (:214-215),
                                fix_grad / quad_grad (:217-228), matching_loss_func / ce_loss_func (:17-18), and the whole
                                ``for k in range(Niter_train)`` statement of finetune_VAE (:284-380) - decoder, hooks, soft-argmax +
                                noise, both renders, null fill, PSP pooling, L1 / cross-entropy, a fresh SGD-Nesterov step per iteration

What is injected (and therefore NOT pinned by these fixtures):
  * ``nr`` - the third-party rasterizer: ``oracle/raster_ref.py::RefRenderer`` (the restatement of SURVEY Appendix B; PARITY
    UNPINNED, the package's source is not on disk).  Everything AROUND it - placement, room shell, cull, texture ranges, per-class
    normalisation, channel layout, loss, hooks, optimiser - is the reference's code running;
  * the mesh / metadata I/O of models/misc.py (:66-133,167-214: OBJ files, pymesh remeshing, the SUNCG json tables - licensed data,
    out of scope): synthetic tables built below, stored in the fixture so that the tests feed the product the same arrays;
  * ``Tensor.cuda()`` is the identity in this process, ``np.float`` is restored (the reference predates numpy 1.24), ``save_images``
    (gif writer) is a no-op, ``print`` is silenced.  One statement is ADDED to the loop body, after ``optimizer.step()``:
    ``_record()`` copies the iteration's tensors out of the namespace.  Nothing else is edited.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_refine.py [helpers|scene|loop|all]

Writes tests/golden/refine_helpers.npz, refine_scene.npz, refine_loop.npz, refine_loop_recurrent.npz (numeric arrays only).
"""
from __future__ import annotations

import ast
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


# ------------------------------------------------------------------------------------------------------------------------
# source extraction
# ------------------------------------------------------------------------------------------------------------------------
def _top_level(path, names):
    """{name: ast node} of the top-level FunctionDef / ClassDef / single-target Assign statements called ``names``"""
    src = open(path).read()
    tree = ast.parse(src, filename=path)
    out = {}
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            out[node.name] = node
        elif isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name) and node.targets[0].id in names:
            out[node.targets[0].id] = node
    missing = [n for n in names if n not in out]
    if missing:
        raise SystemExit("%s: not found: %s" % (path, missing))
    return out


def _run(nodes, ns, path):
    for node in nodes:
        mod = ast.Module(body=[node], type_ignores=[])
        exec(compile(mod, path, "exec"), ns)


def _quiet(*a, **k):
    pass


def _neutralise():
    """this process has no GPU: .cuda() is the identity and .cpu() a copy (as a device -> host transfer is: models/misc.py:35-42 scales
    the numpy view of ``box.cpu()`` in place, which must not reach the caller's tensor); numpy >= 1.24 dropped the np.float alias"""
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.Tensor.cpu = lambda self, *a, **k: self.clone()
    nn.Module.cuda = lambda self, *a, **k: self
    if not hasattr(np, "float"):
        np.float = float


def reference_namespaces(tables, image_size=256):
    """-> (diff_render namespace holding get_cam_mat / mesh_render_func, test_render_refine namespace holding the helpers)"""
    from oracle import raster_ref
    _neutralise()
    dr_path, misc_path, tr_path = (os.path.join(REF, p) for p in ("models/diff_render.py", "models/misc.py", "testing/test_render_refine.py"))
    T = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).astype(dt))
    by_id = {m["id"]: m for ms in tables["suncg_data"].values() for m in ms}
    shell = tables["shell"]
    ns = dict(torch=torch, np=np, nn=nn, print=_quiet,
              nr=types.SimpleNamespace(Renderer=raster_ref.RefRenderer),
              object_idx_to_name=list(tables["object_idx_to_name"]), suncg_data=tables["suncg_data"], wall_data_json=tables["wall_data_json"],
              # --- the I/O of models/misc.py (:106-214), replaced by table look-ups -------------------------------------------
              load_suncg_obj=lambda model_id: (T(by_id[model_id]["v"], np.float32), T(by_id[model_id]["f"], np.int32)),
              load_wall_obj_new=lambda wall_data: ([T(shell["wall_v"], np.float32) for _ in shell["wall_f"]], [T(f, np.int32) for f in shell["wall_f"]]),
              load_floor_obj=lambda floor_data: (T(shell["floor_v"], np.float32), T(shell["floor_f"], np.int32)),
              load_ceil_obj=lambda wall_data: (T(shell["ceil_v"], np.float32), T(shell["ceil_f"], np.int32)))
    misc = _top_level(misc_path, ["suncg_retrieve", "wall_retrieve", "floor_retrieve", "get_bbox"])
    _run(misc.values(), ns, misc_path)
    dr = _top_level(dr_path, ["nyu_class", "inter_out", "final_out", "get_cam_mat", "mesh_render_func"])
    _run(dr.values(), ns, dr_path)
    ns["final_out"] = int(image_size)                         # (a module global of the reference: 256; smaller for the multi-iteration fixture)
    tr = _top_level(tr_path, ["matching_loss_func", "ce_loss_func", "softargmax", "PSP_pool_new", "depth_pooler", "semantic_pooler_novel",
                              "fix_grad", "quad_grad", "finetune_VAE"])
    ns2 = dict(torch=torch, np=np, nn=nn, F=F, os=os, print=_quiet)
    _run([v for k, v in tr.items() if k != "finetune_VAE"], ns2, tr_path)
    ns2["_finetune_node"] = tr["finetune_VAE"]
    return ns, ns2


# ------------------------------------------------------------------------------------------------------------------------
# synthetic stand-ins for the SUNCG tables / meshes (data, not code under test)
# ------------------------------------------------------------------------------------------------------------------------
from oracle.refine_ref import FIXTURE_VOCAB as VOCAB          # 27 furniture classes + two of the classes mesh_render_func skips (:93-97)


def synth_tables(seed=0, subdiv=2, shell_div=5):
    """one model per class (retrieval - models/misc.py:34-63 - is out of scope; a second, badly proportioned wall / floor entry makes
    wall_retrieve / floor_retrieve choose), meshes in 'model' coordinates with the table's bounding boxes NOT equal to the vertices' own
    (the reference scales by the table's, :106-115)"""
    from oracle import raster_ref
    rng = np.random.default_rng(seed)
    suncg = {}
    for name in VOCAB:
        size = rng.uniform(0.5, 1.5, size=3)
        lo = -size / 2 + rng.uniform(-0.2, 0.2, size=3)
        v, f = raster_ref._cuboid(lo, lo + size, subdiv)
        v = v.copy()
        v[:, 0] += 0.12 * (v[:, 1] - lo[1]) * rng.uniform(-1, 1)        # sheared, tapered: not a box
        v[:, 2] *= 1.0 - 0.15 * (v[:, 1] - lo[1]) / size[1]
        pad = rng.uniform(0.0, 0.03, size=3)
        suncg[name] = [dict(id="%s_0" % name, bbox_min=(v.min(0) - pad).tolist(), bbox_max=(v.max(0) + pad).tolist(),
                            v=v.astype(np.float32), f=f.astype(np.int32))]
    # the room shell in 'house' coordinates: four walls sharing ONE vertex array (models/misc.py:84-104: every sub-mesh is handed all
    # the vertices and its own faces), a floor, a ceiling
    org, ext = np.array([37.2, 0.05, 41.5]), np.array([5.2, 2.9, 6.1])
    quads = [(org, [ext[0], 0, 0], [0, ext[1], 0]),                                   # back wall  (z = 0)
             (org, [0, ext[1], 0], [0, 0, ext[2]]),                                   # left wall  (x = 0)
             (org + [ext[0], 0, 0], [0, 0, ext[2]], [0, ext[1], 0]),                  # right wall (x = X)
             (org + [0, 0, ext[2]], [0, ext[1], 0], [ext[0], 0, 0])]                  # front wall (z = Z: the camera's; skipped by :203-213)
    wv, wf, off = [], [], 0
    for p0, du, dv in quads:
        v, f = raster_ref._quad(np.asarray(p0, np.float64), np.asarray(du, np.float64), np.asarray(dv, np.float64), shell_div)
        wv.append(v); wf.append((f + off).astype(np.int32)); off += v.shape[0]
    fv, ff = raster_ref._quad(org + [-0.1, 0, -0.1], np.array([0, 0, ext[2] + 0.2]), np.array([ext[0] + 0.2, 0, 0]), shell_div)
    cv, cf = raster_ref._quad(org + [0, ext[1], 0], np.array([ext[0], 0, 0]), np.array([0, 0, ext[2]]), shell_div)
    shell = dict(wall_v=np.concatenate(wv).astype(np.float32), wall_f=wf, floor_v=fv.astype(np.float32), floor_f=ff.astype(np.int32),
                 ceil_v=cv.astype(np.float32), ceil_f=cf.astype(np.int32))
    good = dict(house_id="h0", model_id="m0", wall_bbox_min=org.tolist(), wall_bbox_max=(org + ext).tolist(),
                floor_bbox_min=(org + [-0.1, 0, -0.1]).tolist(), floor_bbox_max=(org + [ext[0] + 0.1, 0, ext[2] + 0.1]).tolist())
    bad = dict(house_id="h1", model_id="m1", wall_bbox_min=[0, 0, 0], wall_bbox_max=[9.0, 2.0, 2.0], floor_bbox_min=[0, 0, 0], floor_bbox_max=[9.0, 0, 2.0])
    return dict(object_idx_to_name=["__room__"] + VOCAB, suncg_data=suncg, wall_data_json=[bad, good], shell=shell)


def synth_room(seed, n_obj, room=(4.0, 2.7, 5.0)):
    """(objs [n+1], boxes [n+1, 6] room-normalised with the room row last, angles [n+1]) - objects inside the camera's view"""
    rng = np.random.default_rng(seed)
    room = np.asarray(room, np.float64)
    cls = rng.choice(np.arange(1, len(VOCAB) + 1), size=n_obj, replace=False)
    if n_obj >= 4:
        cls[1] = 1 + VOCAB.index("door")                    # one skipped object per room
    size = rng.uniform([0.7, 0.5, 0.7], [1.6, 1.7, 1.6], size=(n_obj, 3))
    lo = rng.uniform([0.2, 0.0, 0.3], np.maximum(room - size - [0.2, 0.0, 0.9], [0.3, 0.01, 0.4]), size=(n_obj, 3))
    lo[:, 1] = rng.uniform(0.0, 0.3, size=n_obj)
    boxes = np.concatenate([lo / room, (lo + size) / room], 1)
    boxes = np.concatenate([boxes, [[0, 0, 0, room[0], room[1], room[2]]]], 0).astype(np.float32)
    objs = np.concatenate([cls, [0]]).astype(np.int64)
    angles = np.concatenate([rng.integers(0, 24, size=n_obj), [0]]).astype(np.int64)
    return objs, boxes, angles


def _tables_to_arrays(tables, prefix="tab:"):
    out = {}
    for name, (m,) in tables["suncg_data"].items():
        out["%smodel:%s:v" % (prefix, name)] = m["v"]; out["%smodel:%s:f" % (prefix, name)] = m["f"]
        out["%smodel:%s:bbox" % (prefix, name)] = np.asarray([m["bbox_min"], m["bbox_max"]], np.float32)
    sh, good = tables["shell"], tables["wall_data_json"][1]
    out[prefix + "wall_v"] = sh["wall_v"]
    for i, f in enumerate(sh["wall_f"]):
        out["%swall_f:%d" % (prefix, i)] = f
    out[prefix + "floor_v"], out[prefix + "floor_f"], out[prefix + "ceil_v"], out[prefix + "ceil_f"] = sh["floor_v"], sh["floor_f"], sh["ceil_v"], sh["ceil_f"]
    out[prefix + "wall_bbox"] = np.asarray([good["wall_bbox_min"], good["wall_bbox_max"]], np.float32)
    out[prefix + "floor_bbox"] = np.asarray([good["floor_bbox_min"], good["floor_bbox_max"]], np.float32)
    out[prefix + "vocab_ids"] = np.arange(len(VOCAB))            # (the names are oracle.gen_golden_refine.VOCAB's, in this order)
    return out


# ------------------------------------------------------------------------------------------------------------------------
# 1. the self-contained helpers
# ------------------------------------------------------------------------------------------------------------------------
def gen_helpers():
    ns, ns2 = reference_namespaces(synth_tables(), 256)
    out = {}
    rng = np.random.default_rng(11)
    # get_cam_mat: ten room boxes incl. a very low room (the min(0.1, |h/2|) branch, :26)
    rooms = rng.uniform([0, 0, 0, 2.5, 2.2, 2.5], [0.2, 0.1, 0.2, 8.0, 3.4, 9.0], size=(10, 6)).astype(np.float32)
    rooms[3, 4] = 0.15; rooms[7, 4] = -0.1
    cams = [ns["get_cam_mat"]([torch.zeros(6), torch.from_numpy(r)]) for r in rooms]
    out["cam:rooms"] = rooms
    out["cam:K"], out["cam:R"], out["cam:t"] = (torch.cat([c[i] for c in cams]).numpy() for i in range(3))
    # softargmax forward / backward
    logits = torch.from_numpy(rng.normal(0, 2.0, size=(13, 24)).astype(np.float32)).requires_grad_(True)
    w = torch.from_numpy(rng.normal(size=13).astype(np.float32))
    idx = ns2["softargmax"](logits, sum_dim=1)
    (idx * w).sum().backward()
    out["sam:logits"], out["sam:w"], out["sam:idx"], out["sam:grad"] = logits.detach().numpy(), w.numpy(), idx.detach().numpy(), logits.grad.numpy()
    # the two gradient hooks
    g6 = torch.from_numpy(rng.normal(size=(13, 6)).astype(np.float32)); g1 = torch.from_numpy(rng.normal(size=13).astype(np.float32))
    out["hook:g6"], out["hook:fix"], out["hook:g1"], out["hook:quad"] = g6.numpy(), ns2["fix_grad"](g6).numpy(), g1.numpy(), ns2["quad_grad"](g1).numpy()
    # PSP_pool_new: the two module-level instances (cat / list mode), forward + backward, three input sizes
    for S, C in ((256, 1), (96, 2), (64, 2)):
        x = torch.from_numpy(rng.uniform(0, 1, size=(1, C, S, S)).astype(np.float32)).requires_grad_(True)
        y = ns2["depth_pooler"](x)
        ys = ns2["semantic_pooler_novel"](x)
        assert isinstance(ys, list) and torch.equal(torch.cat(ys, 1), y)
        wy = torch.from_numpy(rng.normal(size=tuple(y.shape)).astype(np.float32))
        (y * wy).sum().backward()
        out["psp%d:x" % S], out["psp%d:y" % S], out["psp%d:w" % S], out["psp%d:gx" % S] = x.detach().numpy(), y.detach().numpy(), wy.numpy(), x.grad.numpy()
    np.savez_compressed(os.path.join(GOLD, "refine_helpers.npz"), **out)
    print("wrote refine_helpers.npz: %d arrays" % len(out))


# ------------------------------------------------------------------------------------------------------------------------
# 2. mesh_render_func as a whole (first call / later call), values and gradients
# ------------------------------------------------------------------------------------------------------------------------
def _image_summary(img):
    """per-channel float64 (sum, sum of squares, count of > 0.1) of a [1, C, S, S] image"""
    d = img.detach().double()[0]
    return torch.stack([d.sum((1, 2)), (d * d).sum((1, 2)), (d > 0.1).double().sum((1, 2))], 1).numpy()


def gen_scene():
    tables = synth_tables()
    out = _tables_to_arrays(tables)
    for tag, S, seed, n_obj in (("s64", 64, 3, 5), ("s256", 256, 4, 8)):
        ns, ns2 = reference_namespaces(tables, S)
        objs, boxes, angles = synth_room(seed, n_obj)
        B = [torch.from_numpy(b.copy()) for b in boxes]
        A = [torch.tensor(float(a)) for a in angles]
        tgt, ids, sizes, sl0 = ns["mesh_render_func"](B, A, objs.tolist())
        assert sl0 == 0.0 and tgt.shape == (1, 70, S, S)
        # a later call: perturbed boxes / fractional angles that require grad, the cached ids and sizes (:55-60,98-100,160-165)
        rng = np.random.default_rng(100 + seed)
        b2 = torch.from_numpy(boxes + np.concatenate([rng.normal(0, 0.01, size=(n_obj, 6)), np.zeros((1, 6))]).astype(np.float32)).requires_grad_(True)
        a2 = torch.from_numpy((angles + np.concatenate([rng.normal(0, 0.3, size=n_obj), [0]])).astype(np.float32)).requires_grad_(True)
        B2 = [b2[i] for i in range(n_obj)] + [b2[n_obj] * 1.01]                # a drifted room row: overloaded by box_info, penalised by :161-162
        img, ids2, sizes2, sl = ns["mesh_render_func"](B2, [a2[i] for i in range(n_obj + 1)], objs.tolist(), ids, sizes)
        wc, wp = torch.from_numpy(rng.normal(size=(1, 70, 1, 1)).astype(np.float32)), torch.from_numpy(rng.uniform(0.5, 1.5, size=(1, 1, S, S)).astype(np.float32))
        w = wc * wp                                                            # d/d image of the scalar that is back-propagated
        (img * w).sum().backward(retain_graph=True)
        gb_img, ga_img = b2.grad.clone(), a2.grad.clone()
        b2.grad = None; a2.grad = None
        sl.backward()
        p = tag + ":"
        out.update({p + "objs": objs, p + "boxes": boxes, p + "angles": angles, p + "boxes2": b2.detach().numpy(), p + "angles2": a2.detach().numpy(),
                    p + "sizes": np.stack(sizes[:-1]), p + "box_info": ids["box_info"], p + "size_loss2": np.float64(float(sl)),
                    p + "w_chan": wc.numpy(), p + "w_pix": wp.numpy(),
                    p + "grad_boxes_img": gb_img.numpy(), p + "grad_angles_img": ga_img.numpy(),
                    p + "grad_boxes_size": b2.grad.numpy(), p + "grad_angles_size": (a2.grad if a2.grad is not None else torch.zeros_like(a2)).numpy(),
                    p + "target_summary": _image_summary(tgt), p + "image_summary": _image_summary(img)})
        assert ids2 == {} and sizes2 == []
        # the loss statements of the loop (:328-350) on (iterate, target), and the gradient they send back into the image
        lv, ld, ls, gimg = reference_loss(ns2, img.detach(), tgt.detach())
        out.update({p + "loss": np.float64(lv), p + "loss_depth": np.float64(ld), p + "loss_sem": np.float64(ls)})
        gd = gimg.double()[0]
        out[p + "grad_image_summary"] = torch.stack([gd.sum((1, 2)), gd.abs().sum((1, 2)), (gd * gd).sum((1, 2))], 1).numpy()
        if S <= 64:
            out[p + "grad_image"] = gimg.numpy()
        else:
            out[p + "grad_image_sub"] = gimg.numpy()[:, :, 1::4, 2::4]
        if S <= 64:
            out[p + "target"], out[p + "image"] = tgt.detach().numpy(), img.detach().numpy()
        else:                                      # 256^2: every 4th row / column of every plane + the summaries above
            out[p + "target_sub"], out[p + "image_sub"] = tgt.detach().numpy()[:, :, ::4, ::4], img.detach().numpy()[:, :, ::4, ::4]
        print("scene %s: %d objects, size loss %.6f, live planes %d" % (tag, n_obj, float(sl), int((_image_summary(img)[1:41, 0] > 0).sum())))
    np.savez_compressed(os.path.join(GOLD, "refine_scene.npz"), **out)
    print("wrote refine_scene.npz: %d arrays" % len(out))


# ------------------------------------------------------------------------------------------------------------------------
# 3. the refinement loop: the reference's own ``for k in range(Niter_train)`` statement
# ------------------------------------------------------------------------------------------------------------------------
def _k_loop(finetune_node):
    """the ``for k in range(Niter_train):`` statement inside finetune_VAE's trial loop, with ``_record()`` after optimizer.step()"""
    loops = [n for n in ast.walk(finetune_node) if isinstance(n, ast.For) and isinstance(n.target, ast.Name) and n.target.id == "k"]
    assert len(loops) == 1
    loop = loops[0]
    done = 0
    for parent in ast.walk(loop):
        body = getattr(parent, "body", None)
        if not isinstance(body, list):
            continue
        for i, st in enumerate(body):
            if isinstance(st, ast.Expr) and ast.unparse(st) == "optimizer.step()":
                body.insert(i + 1, ast.parse("_record()").body[0]); done += 1
                break
    assert done == 1
    return ast.fix_missing_locations(loop)


def _loss_block(finetune_node):
    """the statements of the loop body from the null fill (:329) to ``loss_val = depth_loss*100 + semantic_loss*100`` (:350), as a module"""
    loop = _k_loop(finetune_node)
    for parent in ast.walk(loop):
        body = getattr(parent, "body", None)
        if not isinstance(body, list):
            continue
        src = [ast.unparse(st) for st in body]
        a = [i for i, t in enumerate(src) if t.startswith("iter_image[:, -1][")]
        b = [i for i, t in enumerate(src) if t.startswith("loss_val = depth_loss * 100")]
        if a and b:
            return ast.fix_missing_locations(ast.Module(body=body[a[0]:b[0] + 1], type_ignores=[]))
    raise SystemExit("loss block not found")


def reference_loss(ns2, image, target):
    """runs the reference's loss statements on (iterate, target) -> (loss_val, depth_loss, semantic_loss, d loss_val / d image)"""
    leaf = image.clone().requires_grad_(True)
    env = dict(ns2)
    node = env.pop("_finetune_node")
    env.update(iter_image=leaf * 1.0, target=target, target_mesh=None, long_dtype=torch.LongTensor, orig_scaler=0.5)
    import copy
    exec(compile(_loss_block(copy.deepcopy(node)), "testing/test_render_refine.py", "exec"), env)
    env["loss_val"].backward()
    return float(env["loss_val"].detach()), float(env["depth_loss"].detach()), float(env["semantic_loss"].detach()), leaf.grad


LOOP_CFG = dict(embedding_dim=32, gconv_num_layers=2, num_objs=len(VOCAB) + 1)
LOOP_ITERS = 4
LOOP_ROOMS = ((21, 5), (22, 7))           # (seed, objects): R = 2 rooms
LOOP_IMAGE = 96
# second fixture: a 'recurrent' decoder (models/graph.py:129-143: ONE GraphTripleConv applied gconv_num_layers times - every wgrad of the
# refinement step adds into the same weights), one room, three iterations
LOOP_CASES = {
    "refine_loop": (LOOP_CFG, LOOP_ROOMS, LOOP_ITERS),
    "refine_loop_recurrent": (dict(embedding_dim=32, gconv_num_layers=3, gconv_mode="recurrent", num_objs=len(VOCAB) + 1), ((23, 6),), 3),
}


def _room_graph(cfg, seed, n_obj):
    """one room's collated graph (data/suncg_dataset.py:207-212,295-337: random triples + one __in_room__ triple per object)"""
    objs_np, boxes_np, angles_np = synth_room(seed, n_obj)
    rng = np.random.default_rng(seed)
    s = rng.integers(0, n_obj, size=n_obj); o = (s + rng.integers(1, n_obj, size=n_obj)) % n_obj
    tri = np.concatenate([np.stack([s, rng.integers(1, cfg.num_preds, size=n_obj), o], 1),
                          np.stack([np.arange(n_obj), np.zeros(n_obj, np.int64), np.full(n_obj, n_obj)], 1)]).astype(np.int64)
    attrs_np = rng.integers(0, cfg.num_attrs, size=n_obj + 1).astype(np.int64)
    return objs_np, tri, boxes_np, angles_np, attrs_np


def loop_state(cfg, ref_vae, ref_utils, rooms=LOOP_ROOMS, steps=400, settle=80):
    """A 'checkpoint' for the loop: the reference model over-fitted to the fixture's rooms with the reference's own loss (a randomly
    initialised decoder predicts degenerate boxes, the render is an empty room and nothing reaches z).  ``settle`` steps at lr 0 let the
    BatchNorm running statistics - what model.eval() reads - catch up with the final weights.  How the state was made is immaterial to
    the fixture: it is stored, and every consumer loads it."""
    from oracle import vae_ref
    sd = vae_ref.init_state(cfg, seed=5)
    model = ref_vae.Sg2ScVAEModel(**cfg.model_kwargs())
    model.load_state_dict({k_: v_.clone() for k_, v_ in sd.items()})
    parts, off = [], 0
    for seed, n_obj in rooms:
        ob, tr, bx, an, at = _room_graph(cfg, seed, n_obj)
        tr = tr.copy(); tr[:, 0] += off; tr[:, 2] += off
        parts.append((ob, tr, bx, an, at)); off += n_obj + 1
    objs, triples, boxes, angles, attrs = (torch.from_numpy(np.concatenate(x)) for x in zip(*parts))
    torch.manual_seed(7)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    args = types.SimpleNamespace(use_AE=cfg.use_AE)
    for it in range(steps + settle):
        if it == steps:
            for g in opt.param_groups:
                g["lr"] = 0.0
        mu, lv, bp, ap = model(objs, triples, boxes, angles, attrs, None)
        total, _ = ref_utils.calculate_model_losses(args, model, boxes, bp, angles, ap, mu=mu, logvar=lv, KL_weight=1e-3)
        opt.zero_grad(); total.backward(); opt.step()
    print("loop_state: over-fitted, train-mode loss %.4f" % float(total.detach()))
    return {k_: v_.detach().clone() for k_, v_ in model.state_dict().items()}


def gen_loop(only=None):
    for name, (cfg_kw, rooms, iters) in LOOP_CASES.items():
        if only is None or name in only:
            _gen_loop_case(name, cfg_kw, rooms, iters)


def _gen_loop_case(case, cfg_kw, LOOP_ROOMS, LOOP_ITERS):
    from oracle import gen_golden, vae_ref
    tables = synth_tables()
    out = _tables_to_arrays(tables)
    ref_graph, ref_vae, ref_utils = gen_golden._import_reference()
    cfg = vae_ref.VaeConfig(**cfg_kw)
    sd0 = loop_state(cfg, ref_vae, ref_utils, LOOP_ROOMS)
    for k_, v_ in sd0.items():
        out["state:" + k_] = v_.numpy()
    for r, (seed, n_obj) in enumerate(LOOP_ROOMS):
        ns, ns2 = reference_namespaces(tables, LOOP_IMAGE)
        objs_np, tri, boxes_np, angles_np, attrs_np = _room_graph(cfg, seed, n_obj)
        n = n_obj + 1
        model = ref_vae.Sg2ScVAEModel(**cfg.model_kwargs())
        model.load_state_dict({k_: v_.clone() for k_, v_ in sd0.items()})
        model.eval()
        objs, triples, boxes_gt, angles, attributes = (torch.from_numpy(a) for a in (objs_np, tri, boxes_np, angles_np, attrs_np))
        obj_to_img = torch.zeros(n, dtype=torch.int64)
        # test_render_refine.py:273-278: encoder, manual_seed(13), reparameterize
        mu, logvar = model.encoder(objs, triples, boxes_gt, angles, attributes)
        torch.manual_seed(13)
        z_np = (mu + torch.randn_like(mu) * torch.exp(0.5 * logvar)).detach().numpy()
        st = torch.get_rng_state()
        noise = torch.stack([torch.randn(n) for _ in range(LOOP_ITERS)]).numpy()              # what the loop is about to draw (:304)
        torch.set_rng_state(st)
        rec = []
        tmp = tempfile.mkdtemp(prefix="sln_refine_fixture_")
        env = dict(ns2)
        del env["_finetune_node"]

        def _record():
            e = env
            rec.append(dict(loss=float(e["loss_val"].detach()), depth=float(e["depth_loss"].detach()), sem=float(e["semantic_loss"].detach()),
                            size=float(e["size_loss"]) if not isinstance(e["size_loss"], float) else e["size_loss"],
                            z=e["z"].detach().clone().numpy(), boxes=e["boxes_pred"].detach().clone().numpy(),
                            idx=e["angles_pred_idx2"].detach().clone().numpy(), image=_image_summary(e["iter_image"]),
                            dz=e["z"].grad.detach().clone().numpy(),
                            params={k_: p.detach().clone().numpy() for k_, p in e["model"].named_parameters()
                                    if k_.startswith(("box_net", "angle_net.0", "gconv_net_dc.gconvs.1.net2.0", "gconv_net_dc.gconvs.0.net1.0"))}))
        env.update(model=model, z=None, z_np=z_np.copy(), save_name=tmp,        # (a copy: on the CPU .type(FloatTensor) aliases z_np, which SGD then steps)
                   float_dtype=torch.FloatTensor, long_dtype=torch.LongTensor,
                   args=types.SimpleNamespace(learning_rate=1e-4), objs=objs, triples=triples, attributes=attributes, obj_to_img=obj_to_img,
                   boxes_gt=boxes_gt, angles=angles, mesh_render_func=ns["mesh_render_func"], save_images=_quiet, pickle=__import__("pickle"),
                   target_mesh=None, model_infos=None, size_infos=None, Niter_train=LOOP_ITERS, used_ids=[r], trial=0, orig_bbox=None, _record=_record)
        loop = _k_loop(ns2["_finetune_node"])
        exec(compile(ast.Module(body=[loop], type_ignores=[]), "testing/test_render_refine.py", "exec"), env)
        assert len(rec) == LOOP_ITERS
        p = "room%d:" % r
        out.update({p + "objs": objs_np, p + "triples": tri, p + "in_boxes": boxes_np, p + "in_angles": angles_np, p + "attributes": attrs_np,
                    p + "z0": z_np, p + "noise": noise, p + "mu": mu.detach().numpy(), p + "logvar": logvar.detach().numpy(),
                    p + "target_summary": _image_summary(torch.from_numpy(env["target_image_np"])),
                    p + "size_target": np.stack(env["size_infos"][:-1]) if len(env["size_infos"]) > 1 else np.zeros((0, 3), np.float32)})
        for k_ in ("loss", "depth", "sem", "size"):
            out[p + k_] = np.asarray([x[k_] for x in rec], np.float64)
        for k_ in ("z", "boxes", "idx", "image", "dz"):
            out[p + k_] = np.stack([x[k_] for x in rec])
        for name in rec[-1]["params"]:
            out[p + "param:" + name] = np.stack([x["params"][name] for x in rec])
        print("loop room %d (%d objects): losses %s" % (r, n_obj, " ".join("%.5f" % x["loss"] for x in rec)))
        print("    depth %s  sem %s  size %s" % (["%.5f" % x["depth"] for x in rec], ["%.5f" % x["sem"] for x in rec], ["%.2e" % x["size"] for x in rec]))
    np.savez_compressed(os.path.join(GOLD, case + ".npz"), **out)
    print("wrote %s.npz: %d arrays" % (case, len(out)))


if __name__ == "__main__":
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present; fixtures can only be regenerated in the build container")
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    os.makedirs(GOLD, exist_ok=True)
    if what in ("helpers", "all"):
        gen_helpers()
    if what in ("scene", "all"):
        gen_scene()
    if what in ("loop", "all"):
        gen_loop()
