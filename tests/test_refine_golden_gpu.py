"""Path B's callers on the GPU against fixtures made by EXECUTING THE REFERENCE'S SOURCE (oracle/gen_golden_refine.py: get_cam_mat,
softargmax, PSP_pool_new, fix_grad / quad_grad, mesh_render_func, the loss statements and the k loop of finetune_VAE, all taken from
the reference files with ``ast`` and run over the restated rasterizer oracle/raster_ref.py).  What these tests pin to the reference's
text: camera, head glue, PSP pooling, placement (objects AND the wall / floor / ceiling rules of diff_render.py:166-342), cull,
per-class normalisation, the 70-plane layout, the loss, the size penalty's target, the hooks, the per-iteration SGD-Nesterov step.
What they cannot pin: the rasterizer's own arithmetic (third-party source absent; see tests/test_raster_gpu.py)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, pkg
from parity import assert_close

pytestmark = pytest.mark.gpu

from oracle import refine_ref, vae_ref     # noqa: E402   (the checker: table loader + configs of the fixtures)
from oracle.refine_ref import FIXTURE_VOCAB     # noqa: E402

LOOP_CFG = dict(embedding_dim=32, gconv_num_layers=2, num_objs=len(FIXTURE_VOCAB) + 1)       # oracle/gen_golden_refine.py::LOOP_CASES
LOOP_CFGS = {"refine_loop": LOOP_CFG,
             "refine_loop_recurrent": dict(embedding_dim=32, gconv_num_layers=3, gconv_mode="recurrent", num_objs=len(FIXTURE_VOCAB) + 1)}
LOOP_IMAGE = 96


def _names(objs):
    return [(["__room__"] + FIXTURE_VOCAB)[int(o)] for o in objs]


def _bank(g, dev="cuda"):
    R = pkg("host.refine")
    t = refine_ref.load_tables(g)
    meshes = {k: (m["v"], m["f"], m["bbox_min"], m["bbox_max"]) for k, m in t["models"].items()}
    return R.MeshBank.from_arrays(meshes, dev, vocab=t["vocab"], shell=t["shell"])


def test_camera_matches_the_reference_function():
    DR = pkg("host.diff_render")
    g = load_golden("refine_helpers")
    for i, room in enumerate(g["cam:rooms"]):
        K, R_, t = DR.get_cam_mat([torch.zeros(6), torch.from_numpy(room)], "cuda")
        assert_close(K[0].cpu(), g["cam:K"][i], "K", rtol=1e-7, atol=0); assert_close(R_[0].cpu(), g["cam:R"][i], "R", rtol=1e-7, atol=0)
        assert_close(t[0].cpu(), g["cam:t"][i], "t", rtol=1e-6, atol=0)


def test_head_kernels_match_the_reference_softargmax_and_hooks():
    """sln_refine_head_forward / _backward: rows 0..n-2 are softargmax (+ noise / 10) and, backwards, quad_grad / fix_grad applied to
    what arrives; the last row is the frozen room row (zero gradient)."""
    R = pkg("host.refine")
    g = load_golden("refine_helpers")
    logits = torch.from_numpy(g["sam:logits"]).cuda().requires_grad_(True)
    n = logits.shape[0]
    boxes_pred = torch.rand(n, 6, device="cuda").requires_grad_(True)
    noise = torch.randn(n, device="cuda")
    box_last, angle_last = torch.tensor([0, 0, 0, 4.0, 2.7, 5.0], device="cuda"), torch.tensor([3.0], device="cuda")
    boxes_full, idx = R._HeadFn.apply(boxes_pred, logits, noise, box_last, angle_last, 2.0)
    assert_close((idx - noise / 10.0)[:-1].detach().cpu(), g["sam:idx"][:-1], "softargmax", rtol=1e-5, atol=1e-6)
    assert float(idx[-1]) == 3.0 and torch.equal(boxes_full[-1], box_last) and torch.equal(boxes_full[:-1], boxes_pred[:-1])
    w, g6 = torch.from_numpy(g["sam:w"]).cuda(), torch.from_numpy(g["hook:g6"]).cuda()
    ((idx * w).sum() + (boxes_full * g6).sum()).backward()
    want = 4.0 * g["sam:grad"]; want[-1] = 0                               # quad_grad (:226-228) on the soft-argmax gradient
    assert_close(logits.grad.cpu(), want, "d softargmax through quad_grad", rtol=1e-4, atol=1e-7)
    fix = g["hook:fix"].copy(); fix[-1] = 0                                  # fix_grad (:217-224); frozen row
    assert_close(boxes_pred.grad.cpu(), fix, "fix_grad", rtol=1e-6, atol=0)


@pytest.mark.parametrize("S", [256, 96, 64])
def test_pooling_kernel_matches_the_reference_class(S):
    """sln_refine_pool (pool_lds_kernel / pool_kernel) on the planes PSP_pool_new saw"""
    R = pkg("host.refine"); _lib = pkg("_lib")
    g = load_golden("refine_helpers")
    x = torch.from_numpy(g["psp%d:x" % S]).cuda()
    C = x.shape[1]
    img = torch.zeros(1, 70, S, S, device="cuda")
    img[:, 41:41 + C] = x                                                    # depth-hot planes ...
    img[:, 1:1 + C] = x                                                      # ... and semantic planes take the same resampling
    rl = R.RefineLoss(img)
    d = rl.desc
    pooled = torch.empty(1, 4, 69, 96, 96, device="cuda")
    _lib.check(_lib.lib().sln_refine_pool(d, _lib.ptr(img), 0, _lib.ptr(rl.ws), _lib.ptr(pooled), _lib.current_stream_ptr()), "sln_refine_pool")
    want = torch.from_numpy(g["psp%d:y" % S]).reshape(4, C, 96, 96)          # PSP_pool_new concatenates scale-major
    assert_close(pooled[0, :, 40:40 + C].cpu(), want, "pooled depth planes", rtol=1e-5, atol=1e-6)
    assert_close(pooled[0, :, 0:C].cpu(), want, "pooled semantic planes", rtol=1e-5, atol=1e-6)


def test_fused_loss_matches_the_reference_statements():
    """RefineLoss (csrc/refine_loss.hip) on the 64^2 (iterate, target) pair the reference's loss statements (:328-350) were run on:
    value, both parts, and d loss / d image"""
    R = pkg("host.refine")
    g = load_golden("refine_scene")
    img = torch.from_numpy(g["s64:image"]).cuda().requires_grad_(True)
    tgt = torch.from_numpy(g["s64:target"]).cuda()
    out = R.RefineLoss(tgt)(img)
    out[0].backward()
    o = out.detach().cpu().numpy()
    assert abs(o[0] - g["s64:loss"]) <= 1e-4 * g["s64:loss"] and abs(o[1] - g["s64:loss_depth"]) <= 1e-4 * g["s64:loss_depth"]
    assert abs(o[2] - g["s64:loss_sem"]) <= 1e-4 * g["s64:loss_sem"]
    assert_close(img.grad.cpu(), g["s64:grad_image"], "d loss / d image", rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize("tag,S", [("s64", 64), ("s256", 256)])
def test_scene_matches_the_reference_mesh_render_func(tag, S):
    """RefineScene (fused placement kernels + fused scene pass) and the drop-in mesh_render_func against the reference's
    mesh_render_func: first call (target) and a later call with cached room row / sizes, values and gradients.  A silhouette pixel may
    land on the other side (the placement arithmetic runs in a different order on the device): counted, bounded, and the gradient
    bounds below are what that leaves."""
    R = pkg("host.refine"); DR = pkg("host.diff_render")
    g = load_golden("refine_scene")
    bank = _bank(g)
    p = tag + ":"
    names = _names(g[p + "objs"])
    boxes, angles = torch.from_numpy(g[p + "boxes"]).cuda(), torch.from_numpy(g[p + "angles"]).float().cuda()
    sc = R.RefineScene(names, bank, boxes[-1].clone(), S)
    with torch.no_grad():
        tgt, sl0, sizes = sc.render(boxes, angles)
    assert float(sl0) == 0.0
    assert_close(sizes.cpu(), g[p + "sizes"], "cached sizes", rtol=1e-6, atol=0)

    def summary(im):
        d = im.detach().double()[0]
        return torch.stack([d.sum((1, 2)), (d * d).sum((1, 2)), (d > 0.1).double().sum((1, 2))], 1).cpu().numpy()

    def same_image(got, full_key, sub_key, sum_key, flips):
        if full_key in g.files:
            bad = int((np.abs(got.detach().cpu().numpy() - g[full_key]) > 1e-4).any(1).sum())
        else:
            bad = int((np.abs(got.detach().cpu().numpy()[:, :, ::4, ::4] - g[sub_key]) > 1e-4).any(1).sum())
        assert bad <= flips, "%d pixels differ from the reference's image" % bad
        s = summary(got)
        assert np.abs(s[:, 2] - g[sum_key][:, 2]).max() <= flips, "covered pixels per plane"
        assert_close(s[:, 0], g[sum_key][:, 0], "plane sums", rtol=3e-4 if S >= 256 else 2e-3, atol=0)
    same_image(tgt, p + "target", p + "target_sub", p + "target_summary", 4)
    # later call: perturbed layout, drifted room row, cached sizes
    b2 = torch.from_numpy(g[p + "boxes2"]).cuda().requires_grad_(True)
    a2 = torch.from_numpy(g[p + "angles2"]).cuda().requires_grad_(True)
    size_t = torch.from_numpy(g[p + "sizes"]).cuda()
    img, sl, _ = sc.render(b2, a2, size_t)
    same_image(img, p + "image", p + "image_sub", p + "image_summary", 6)
    w = (torch.from_numpy(g[p + "w_chan"]) * torch.from_numpy(g[p + "w_pix"])).cuda()
    (img * w).sum().backward(retain_graph=True)
    gb, ga = b2.grad.clone(), a2.grad.clone()
    b2.grad = None; a2.grad = None
    sl.backward()
    # the drifted room row's own penalty (:160-162) is added by mesh_render_func, not by the scene: compare the objects' part
    drift = float(torch.nn.functional.mse_loss(torch.from_numpy(g[p + "boxes2"][-1] * 1.01), torch.from_numpy(g[p + "box_info"])))
    assert abs(float(sl.detach()) + drift - float(g[p + "size_loss2"])) <= 1e-5 * float(g[p + "size_loss2"])
    assert_close(b2.grad.cpu()[:-1], g[p + "grad_boxes_size"][:-1], "d size loss / d boxes", rtol=1e-5, atol=0)
    # (pixel-map gradients concentrate on silhouettes: a flipped pixel moves them; 1e-4 where no pixel flips - test_raster_gpu.py)
    assert_close(gb.cpu()[:-1], g[p + "grad_boxes_img"][:-1], "d image functional / d boxes", rtol=5e-3, atol=0)
    assert_close(ga.cpu()[:-1], g[p + "grad_angles_img"][:-1], "d image functional / d angles", rtol=5e-3, atol=0)
    # the drop-in entry point, both calls
    R.configure_meshes(["__room__"] + FIXTURE_VOCAB, bank)
    if S == DR.final_out:
        final, ids, sz, l0 = DR.mesh_render_func([b for b in boxes], [a for a in angles], g[p + "objs"].tolist())
        assert float(l0) == 0.0            # (the drop-in places through torch ops, the scene through the fused kernel: same image up to silhouettes)
        same_image(final, p + "target", p + "target_sub", p + "target_summary", 4)
        bb = [b2.detach()[i] for i in range(len(names) - 1)] + [b2.detach()[-1] * 1.01]
        final2, _, _, l2 = DR.mesh_render_func(bb, [a for a in a2.detach()], g[p + "objs"].tolist(), ids, sz)
        assert abs(float(l2) - float(g[p + "size_loss2"])) <= 1e-5 * float(g[p + "size_loss2"])
        same_image(final2, p + "image", p + "image_sub", p + "image_summary", 6)


def _loop_model(g, case="refine_loop"):
    M = pkg("host.Sg2ScVAE_model")
    cfg = vae_ref.VaeConfig(**LOOP_CFGS[case])
    model = M.Sg2ScVAEModel(**cfg.model_kwargs())
    model.load_state_dict({k[6:]: torch.from_numpy(g[k]).clone() for k in g.files if k.startswith("state:")})
    return model.cuda().eval(), cfg


def _loop_rooms(g, rooms):
    out = []
    for r in rooms:
        p = "room%d:" % r
        out.append(dict(objs=torch.from_numpy(g[p + "objs"]).cuda(), triples=torch.from_numpy(g[p + "triples"]).cuda(),
                        boxes=torch.from_numpy(g[p + "in_boxes"]).cuda(), angles=torch.from_numpy(g[p + "in_angles"]).cuda(),
                        attributes=torch.from_numpy(g[p + "attributes"]).cuda(), class_names=_names(g[p + "objs"])))
    return out


def _check_loop(g, r, losses, z_hist, boxes_hist, idx_hist, params_after, model_names):
    p = "room%d:" % r
    it = len(losses)
    for k in range(it):
        # north_star's 1e-4, relative to the quantity, on everything that is not a gradient through a silhouette
        assert abs(losses[k] - g[p + "loss"][k]) <= 1e-4 * g[p + "loss"][k], ("loss", k, losses[k], g[p + "loss"][k])
        assert_close(boxes_hist[k], g[p + "boxes"][k], "boxes[%d]" % k, rtol=1e-4, atol=0)
        assert_close(idx_hist[k], g[p + "idx"][k], "angle idx[%d]" % k, rtol=1e-4, atol=0)
        assert_close(z_hist[k], g[p + "z"][k], "z[%d]" % k, rtol=1e-5, atol=0)
        z_prev = g[p + "z"][k - 1] if k else g[p + "z0"]
        # the step of z IS the gradient through the render (x 2.2e-4).  Measured (tools/lab/refine_loop_errs.py): z equals the
        # reference's bit for bit in room 0 and to one ulp of z (3e-7 absolute, 1.4e-2 of a step of 7e-6) in room 1
        ulp = float(np.spacing(np.float32(np.abs(g[p + "z"][k]).max())))          # z itself is fp32: its step is known to an ulp of z, not of the step
        assert_close(z_hist[k] - z_prev, g[p + "z"][k] - z_prev, "step of z[%d]" % k, rtol=1e-3, atol=2 * ulp)
    for name, got in params_after.items():
        p0 = g["state:" + name]
        want = g[p + "param:" + name][it - 1]
        ulp = float(np.spacing(np.float32(np.abs(want).max())))          # the parameter is fp32: every one of its steps is rounded to an ulp of the PARAMETER
        assert_close(got - p0, want - p0, "%d steps of %s" % (it, name), rtol=6e-3, atol=it * ulp)        # measured <= 2.1e-3 of the steps' size


@pytest.mark.parametrize("case,rooms", [("refine_loop", [0, 1]), ("refine_loop", [1]), ("refine_loop_recurrent", [0])])
def test_refine_batch_matches_the_reference_loop(case, rooms):
    """RefineBatch (R rooms in flight, every kernel of the device loop) against the reference's own k loop, four iterations at 96^2:
    per iteration the loss, boxes, angle indices and z; after the last one the stepped decoder parameters.  Second fixture: a
    'recurrent' decoder (one GraphTripleConv applied three times: every application's wgrad steps the same weights)."""
    R = pkg("host.refine")
    g = load_golden(case)
    model, cfg = _loop_model(g, case)
    bank = _bank(g)
    rm = _loop_rooms(g, rooms)
    it = g["room0:noise"].shape[0]
    rb = R.RefineBatch(model, rm, bank=bank, image_size=LOOP_IMAGE, iters=it)
    try:
        for i, r in enumerate(rooms):                   # the reference's draws: z (reparameterisation under manual_seed(13)) and the noise rows
            a, n = rb.row0[i], rb.rows[i]
            assert_close(rb.z[a:a + n].cpu(), g["room%d:z0" % r], "z0 (encoder + seed-13 draw)", rtol=1e-5, atol=0)
            assert_close(rb.noise_all[:it, a:a + n].cpu(), g["room%d:noise" % r], "noise rows", rtol=0, atol=0)
            rb.z[a:a + n] = torch.from_numpy(g["room%d:z0" % r]).cuda()
        zs, bs, ids = [], [], []
        for k in range(it):
            rb.run(1)
            zs.append(rb.z.detach().cpu().numpy().copy()); bs.append(rb.boxes.detach().cpu().numpy().copy()); ids.append(rb.idx.detach().cpu().numpy().copy())
        losses = rb.losses[:it].cpu().numpy()
        for i, r in enumerate(rooms):
            a, n = rb.row0[i], rb.rows[i]
            names = [k[len("room%d:param:" % r):] for k in g.files if k.startswith("room%d:param:" % r)]
            after = {}
            for name in names:
                t = dict(model.named_parameters())[name]
                off = (t.data_ptr() - model.flat_params.data_ptr()) // 4
                after[name] = rb.params[i, off:off + t.numel()].reshape(t.shape).cpu().numpy()
            _check_loop(g, r, losses[:, i], [z[a:a + n] for z in zs], [b[a:a + n] for b in bs], [x[a:a + n] for x in ids], after, names)
    finally:
        rb.close()


def test_one_room_loops_match_the_reference_loop():
    """finetune_vae_fast (autograd around the fused kernels) on room 0: same comparison"""
    R = pkg("host.refine")
    g = load_golden("refine_loop")
    model, cfg = _loop_model(g)
    bank = _bank(g)
    rm = _loop_rooms(g, [0])[0]
    it = g["room0:noise"].shape[0]
    p0 = {k: v.detach().cpu().numpy().copy() for k, v in model.named_parameters()}
    losses, (bp, idx) = R.finetune_vae_fast(model, rm["objs"], rm["triples"], rm["boxes"], rm["angles"], rm["attributes"], rm["class_names"],
                                            iters=it, bank=bank, image_size=LOOP_IMAGE)
    losses = losses.cpu().numpy()
    for k in range(it):
        assert abs(losses[k] - g["room0:loss"][k]) <= 1e-4 * g["room0:loss"][k], ("loss", k, losses[k], g["room0:loss"][k])
    assert_close(bp.cpu(), g["room0:boxes"][it - 1], "boxes of the last iteration", rtol=1e-4, atol=0)
    assert_close(idx.cpu(), g["room0:idx"][it - 1], "angle idx of the last iteration", rtol=1e-4, atol=0)
    for name in [k[len("room0:param:"):] for k in g.files if k.startswith("room0:param:")]:
        got = dict(model.named_parameters())[name].detach().cpu().numpy()
        assert_close(got - p0[name], g["room0:param:" + name][it - 1] - g["state:" + name], "%d steps of %s" % (it, name), rtol=1e-2, atol=1e-9)
