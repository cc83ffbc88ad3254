"""The oracle's restatement of path B's callers against fixtures produced by EXECUTING THE REFERENCE'S SOURCE
(oracle/gen_golden_refine.py: get_cam_mat, softargmax, PSP_pool_new, fix_grad / quad_grad, mesh_render_func and the k loop of
finetune_VAE, taken from the reference files with ``ast``; the rasterizer under them is oracle/raster_ref.py - unpinned)."""
import os

import numpy as np
import pytest
import torch

from oracle import raster_ref, refine_ref as rf, vae_ref

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(GOLD, name))


def _close(a, b, tol, what=""):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b).max() if a.size else 0.0
    assert err <= tol * max(np.abs(b).max() if b.size else 0.0, 1e-30), "%s: max err %.3e (scale %.3e)" % (what, err, np.abs(b).max())


def test_camera_matches_the_reference_function():
    g = _load("refine_helpers.npz")
    for i, room in enumerate(g["cam:rooms"]):
        K, R, t = raster_ref.get_cam_mat(torch.from_numpy(room))
        _close(K[0], g["cam:K"][i], 1e-7, "K"); _close(R[0], g["cam:R"][i], 1e-7, "R"); _close(t[0], g["cam:t"][i], 1e-6, "t")


def test_softargmax_and_gradient_hooks_match_the_reference_functions():
    g = _load("refine_helpers.npz")
    x = torch.from_numpy(g["sam:logits"]).requires_grad_(True)
    idx = rf.softargmax(x)
    (idx * torch.from_numpy(g["sam:w"])).sum().backward()
    _close(idx.detach(), g["sam:idx"], 1e-6, "softargmax"); _close(x.grad, g["sam:grad"], 1e-5, "d softargmax")
    assert np.array_equal(rf.fix_grad(torch.from_numpy(g["hook:g6"])).numpy(), g["hook:fix"])
    assert np.array_equal(rf.quad_grad(torch.from_numpy(g["hook:g1"])).numpy(), g["hook:quad"])


@pytest.mark.parametrize("S", [256, 96, 64])
def test_psp_pooling_matches_the_reference_class(S):
    g = _load("refine_helpers.npz")
    x = torch.from_numpy(g["psp%d:x" % S]).requires_grad_(True)
    y = rf.psp_pool(x)
    assert torch.equal(torch.cat(rf.psp_pool(x, as_list=True), 1), y)
    (y * torch.from_numpy(g["psp%d:w" % S])).sum().backward()
    _close(y.detach(), g["psp%d:y" % S], 1e-6, "pooled"); _close(x.grad, g["psp%d:gx" % S], 1e-5, "d pooled")


def _names(objs):
    return [(["__room__"] + rf.FIXTURE_VOCAB)[int(o)] for o in objs]


@pytest.mark.parametrize("tag,S", [("s64", 64), ("s256", 256)])
def test_render_room_matches_the_reference_mesh_render_func(tag, S):
    g = _load("refine_scene.npz")
    tables = rf.load_tables(g)
    p = tag + ":"
    names = _names(g[p + "objs"])
    boxes, angles = torch.from_numpy(g[p + "boxes"]), torch.from_numpy(g[p + "angles"]).float()
    tgt, sizes, sl0 = rf.render_room(boxes, angles, names, tables, S)
    assert float(sl0) == 0.0
    _close(sizes, g[p + "sizes"], 1e-6, "cached sizes")
    # later call: cached room row + sizes, gradients of an image functional and of the size loss
    b2 = torch.from_numpy(g[p + "boxes2"]).requires_grad_(True)
    a2 = torch.from_numpy(g[p + "angles2"]).requires_grad_(True)
    bb = torch.cat((b2[:-1], b2[-1:] * 1.01), 0)
    img, _, sl = rf.render_room(bb, a2, names, tables, S, room_box=g[p + "box_info"], size_target=(g[p + "sizes"], g[p + "box_info"]))
    _close(float(sl), float(g[p + "size_loss2"]), 1e-5, "size loss")
    w = torch.from_numpy(g[p + "w_chan"]) * torch.from_numpy(g[p + "w_pix"])
    (img * w).sum().backward(retain_graph=True)
    gb, ga = b2.grad.clone(), a2.grad.clone()
    b2.grad = None; a2.grad = None
    sl.backward()
    if S <= 64:
        assert np.array_equal(tgt.numpy(), g[p + "target"]), "target image differs from the reference's"
        # the iterate goes through the placement arithmetic twice (reference: 4x4 matmuls; here: scale * R v + t): silhouettes may move
        diff = np.abs(img.detach().numpy() - g[p + "image"])
        assert (diff > 1e-5).sum() <= 40, "pixels that differ: %d" % int((diff > 1e-5).sum())
    else:
        sub = np.abs(tgt.numpy()[:, :, ::4, ::4] - g[p + "target_sub"])
        assert (sub > 1e-6).sum() == 0
    for got, name in ((tgt, "target_summary"), (img.detach(), "image_summary")):
        s = torch.stack([got.double()[0].sum((1, 2)), (got.double()[0] ** 2).sum((1, 2)), (got[0] > 0.1).double().sum((1, 2))], 1).numpy()
        assert np.abs(s[:, 2] - g[p + name][:, 2]).max() <= 3, "covered pixel counts per channel"
        _close(s[:, 0], g[p + name][:, 0], 2e-4, name)
    _close(gb, g[p + "grad_boxes_img"], 2e-3, "d image / d boxes"); _close(ga, g[p + "grad_angles_img"], 2e-3, "d image / d angles")
    _close(b2.grad, g[p + "grad_boxes_size"], 1e-5, "d size loss / d boxes")


@pytest.mark.parametrize("case,r", [("refine_loop", 0), ("refine_loop", 1), ("refine_loop_recurrent", 0)])
def test_refine_loop_matches_the_reference_loop(case, r):
    g = _load(case + ".npz")
    tables = rf.load_tables(g)
    from oracle.gen_golden_refine import LOOP_CASES, LOOP_IMAGE
    cfg = vae_ref.VaeConfig(**LOOP_CASES[case][0])
    sd = {k[6:]: torch.from_numpy(g[k]).clone() for k in g.files if k.startswith("state:")}
    train = [k for k in vae_ref.trainable_keys(cfg)]
    for k in train:
        sd[k].requires_grad_(True)
    p = "room%d:" % r
    objs, tri, attrs = (torch.from_numpy(g[p + k]) for k in ("objs", "triples", "attributes"))
    room = dict(boxes=torch.from_numpy(g[p + "in_boxes"]), angles=torch.from_numpy(g[p + "in_angles"]), class_names=_names(g[p + "objs"]))
    with torch.no_grad():
        mu, lv = vae_ref.encoder(sd, cfg, objs, tri, room["boxes"], room["angles"], attrs, training=False)
    _close(mu, g[p + "mu"], 1e-5, "mu"); _close(lv, g[p + "logvar"], 1e-5, "logvar")
    z = torch.from_numpy(g[p + "z0"]).clone().requires_grad_(True)
    out = rf.refine_loop(lambda zz: vae_ref.decoder(sd, cfg, zz, objs, tri, attrs, training=False), [sd[k] for k in train], z, room, tables,
                         torch.from_numpy(g[p + "noise"]), image_size=LOOP_IMAGE)
    # measured in the build container: every quantity below reproduces the reference's run bit for bit (one weight step differs by
    # 1.3e-6 of its size); the bounds leave room for another CPU's matmul kernels / thread count
    for k, rec in enumerate(out):
        _close(rec["loss"], g[p + "loss"][k], 1e-5, "loss[%d]" % k)
        _close(rec["depth"], g[p + "depth"][k], 1e-5, "depth[%d]" % k); _close(rec["sem"], g[p + "sem"][k], 1e-5, "sem[%d]" % k)
        assert abs(rec["size"] - g[p + "size"][k]) <= 1e-5 * g[p + "size"][k] + 1e-12
        _close(rec["boxes"], g[p + "boxes"][k], 1e-5, "boxes[%d]" % k)
        _close(rec["idx"], g[p + "idx"][k], 1e-5, "idx[%d]" % k)
        _close(rec["dz"], g[p + "dz"][k], 1e-4, "dz[%d]" % k)
        z_prev = g[p + "z"][k - 1] if k else g[p + "z0"]
        _close(rec["z"].numpy() - z_prev, g[p + "z"][k] - z_prev, 1e-3, "step of z[%d]" % k)
        _close(rec["z"], g[p + "z"][k], 1e-6, "z[%d]" % k)
    for key in [k for k in g.files if k.startswith(p + "param:")]:
        name = key[len(p) + 6:]
        p0 = g["state:" + name]
        _close(sd[name].detach().numpy() - p0, g[key][-1] - p0, 1e-3, "four steps of " + name)


def test_refinement_loss_matches_the_reference_statements():
    """the loop's loss statements (:328-350) run by the generator on (iterate, target) of the 64^2 scene: value, parts, d / d image"""
    g = _load("refine_scene.npz")
    img = torch.from_numpy(g["s64:image"]).requires_grad_(True)
    tgt = torch.from_numpy(g["s64:target"])
    loss, depth, sem = rf.refinement_loss(img * 1.0, tgt, rf.target_labels(tgt), torch.zeros(()))
    loss.backward()
    _close(float(loss.detach()), float(g["s64:loss"]), 1e-6, "loss"); _close(float(depth.detach()), float(g["s64:loss_depth"]), 1e-6, "depth")
    _close(float(sem.detach()), float(g["s64:loss_sem"]), 1e-6, "sem")
    _close(img.grad, g["s64:grad_image"], 1e-5, "d loss / d image")
