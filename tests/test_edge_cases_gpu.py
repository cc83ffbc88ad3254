"""Edge cases of the three paths on the GPU (run with -m gpu): inputs the reference accepts but the bulk parity
tests never produce.  Checker: the CPU oracle, as everywhere else.

  * scene graphs with isolated objects (degree 0 -> clamp(count, 1), models/graph.py:102-108), self loops and
    repeated triples (scatter_add adds every row, :96-100), one-object graphs, graphs without triples (eval mode);
  * rasterizer: nothing visible, zero-area and collinear triangles, triangles far larger than the image,
    coincident triangles (lowest index wins), near / far rejection.
"""
import numpy as np
import pytest
import torch

from conftest import pkg
from parity import assert_close

pytestmark = pytest.mark.gpu

from oracle import raster_ref as rr       # noqa: E402
from oracle import vae_ref                # noqa: E402


def _model(cfg, sd):
    M = pkg("host.Sg2ScVAE_model")
    m = M.Sg2ScVAEModel(**cfg.model_kwargs())
    m.load_state_dict({k: v.clone() for k, v in sd.items()})
    return m.cuda()


def _odd_batch(cfg, seed):
    """3 graphs; objects 2 and 9 appear in no triple, triple rows are repeated and two are self loops."""
    objs, triples, boxes, angles, attrs, _ = vae_ref.synth_batch(3, 6, 8, seed=seed, cfg=cfg)
    t = triples.clone()
    for iso, repl in ((2, 3), (9, 10)):
        t[:, 0][t[:, 0] == iso] = repl
        t[:, 2][t[:, 2] == iso] = repl
    t = torch.cat([t, t[:4], torch.tensor([[4, 3, 4], [13, 1, 13]])], 0)          # duplicates + self loops
    return objs, t, boxes, angles, attrs


def _run_both(cfg, batch, training, seed=3):
    sd = vae_ref.init_state(cfg, seed=seed)
    O = batch[0].shape[0]
    eps = torch.from_numpy(np.random.default_rng(seed).standard_normal((O, cfg.embedding_dim)).astype(np.float32))
    with torch.no_grad():
        ref = vae_ref.forward({k: v.clone() for k, v in sd.items()}, cfg, *batch, eps, training=training)
    model = _model(cfg, sd)
    model.train(training)
    with torch.no_grad():
        got = model(*[t.cuda() for t in batch], None, eps=eps.cuda())
    return got, ref, sd, eps


@pytest.mark.parametrize("norm,training", [("batch", True), ("batch", False), ("none", True)])
def test_isolated_objects_self_loops_and_repeated_triples(norm, training):
    cfg = vae_ref.VaeConfig(embedding_dim=16, gconv_num_layers=2, mlp_normalization=norm)
    batch = _odd_batch(cfg, seed=11)
    deg = np.bincount(batch[1][:, [0, 2]].reshape(-1).numpy(), minlength=batch[0].shape[0])
    assert deg[2] == 0 and deg[9] == 0
    got, ref, sd, eps = _run_both(cfg, batch, training)
    for g, r, nm in zip(got, ref, ("mu", "logvar", "boxes_pred", "angles_pred")):
        assert_close(g.cpu().numpy(), r.numpy(), "odd:%s:%s" % (norm, nm), rtol=2e-4)


def test_isolated_objects_gradients_and_adam():
    cfg = vae_ref.VaeConfig(embedding_dim=16, gconv_num_layers=2, mlp_normalization="batch")
    batch = _odd_batch(cfg, seed=12)
    sd = vae_ref.init_state(cfg, seed=5)
    O = batch[0].shape[0]
    eps = torch.from_numpy(np.random.default_rng(5).standard_normal((O, cfg.embedding_dim)).astype(np.float32))
    sdr = {k: v.clone() for k, v in sd.items()}
    keys = vae_ref.trainable_keys(cfg)
    m = {k: torch.zeros_like(sdr[k]) for k in keys}; v = {k: torch.zeros_like(sdr[k]) for k in keys}
    total, parts, grads = vae_ref.train_step(sdr, cfg, batch, eps, 0.1, m, v, step=1)
    model = _model(cfg, sd).train()
    losses = model.train_step(*[t.cuda() for t in batch], kl_weight=0.1, lr=1e-4, eps=eps.cuda(), use_graph=False).cpu().numpy()
    assert_close(losses[3], total.numpy(), "odd:total", rtol=2e-4)
    gscale = max(float(g.abs().max()) for g in grads.values())
    named = dict(model.named_parameters())
    for k, g in grads.items():
        assert_close(named[k].grad.cpu().numpy(), g.numpy(), "odd:grad:" + k, rtol=5e-4, atol=2e-5 * gscale)


@pytest.mark.parametrize("norm", ["batch", "none"])
def test_graph_without_triples_and_single_object_eval(norm):
    """Eval mode (train-mode BatchNorm1d refuses an empty / one-row batch in the reference)."""
    cfg = vae_ref.VaeConfig(embedding_dim=16, gconv_num_layers=2, mlp_normalization=norm)
    objs, triples, boxes, angles, attrs, _ = vae_ref.synth_batch(1, 5, 6, seed=2, cfg=cfg)
    for sub, tr in ((slice(0, 5), triples[:0]), (slice(4, 5), triples[:0]), (slice(0, 1), torch.tensor([[0, 2, 0]]))):
        batch = (objs[sub], tr, boxes[sub], angles[sub], attrs[sub])
        got, ref, _, _ = _run_both(cfg, batch, training=False)
        for g, r, nm in zip(got, ref, ("mu", "logvar", "boxes_pred", "angles_pred")):
            assert g.shape == r.shape
            assert_close(g.cpu().numpy(), r.numpy(), "tiny:%s:%s" % (norm, nm), rtol=2e-4)


# ----------------------------------------------------------------------------- rasterizer
def _hip_forward(f, image_size, near, far):
    L = pkg("_lib")
    fd = torch.from_numpy(f).cuda().contiguous()
    B, F = fd.shape[:2]
    fi = torch.empty(B, image_size, image_size, dtype=torch.int32, device="cuda")
    w = torch.empty(B, image_size, image_size, 3, device="cuda"); d = torch.empty(B, image_size, image_size, device="cuda")
    ws = torch.empty(max(int(L.lib().sln_raster_workspace_bytes(B, F)), 16), dtype=torch.uint8, device="cuda")
    L.check(L.lib().sln_raster_forward(L.ptr(fd), B, F, image_size, near, far, L.ptr(ws), L.ptr(fi), L.ptr(w), L.ptr(d),
                                       L.current_stream_ptr()), "fwd")
    return fi.cpu().numpy(), w.cpu().numpy(), d.cpu().numpy()


def _tri(x0, y0, x1, y1, x2, y2, z):
    return [[x0, y0, z], [x1, y1, z], [x2, y2, z]]


def _check(f, size, near, far, expect_visible=None):
    f = np.asarray(f, np.float32)
    rfi, rw, rd = rr.nmr_forward(f, size, near, far)
    fi, w, d = _hip_forward(f, size, near, far)
    assert (fi == rfi).all(), "face index differs at %d pixels" % int((fi != rfi).sum())
    assert (d == rd).all() and (w == rw).all()
    if expect_visible is not None:
        assert bool((rfi >= 0).any()) == expect_visible
    return rfi


def test_raster_nothing_visible():
    off = [_tri(2.0, 2.0, 3.0, 2.0, 2.0, 3.0, 1.0), _tri(-5.0, -5.0, -4.0, -5.0, -5.0, -4.0, 1.0)]     # outside the view
    behind = [_tri(-1, -1, 1, -1, 0, 1, 0.0005), _tri(-1, -1, 1, -1, 0, 1, 500.0)]                    # closer than near / beyond far
    rfi = _check([off + behind], 64, 0.001, 100.0, expect_visible=False)
    assert (rfi == -1).all()


def test_raster_degenerate_and_giant_triangles():
    rng = np.random.default_rng(4)
    faces = [_tri(0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 1.0),                         # a point
             _tri(-0.5, -0.5, 0.0, 0.0, 0.5, 0.5, 1.0),                       # collinear
             _tri(-0.3, 0.2, 0.4, 0.2, 0.4, 0.2, 1.0),                        # two equal vertices
             _tri(-40.0, -30.0, 50.0, -35.0, 3.0, 60.0, 2.0),                 # covers the image, vertices far outside
             _tri(-0.2, -0.2, 0.3, -0.1, 0.0, 0.4, 1.5),
             _tri(-0.2, -0.2, 0.3, -0.1, 0.0, 0.4, 1.5)]                      # coincident with the previous one: index 4 wins
    for _ in range(40):                                                        # thin slivers through pixel centres
        x = float(rng.integers(-30, 30)) / 32.0 + 1.0 / 64.0
        faces.append(_tri(x, -1.0, x + 1e-4, -1.0, x, 1.0, float(rng.uniform(0.5, 3.0))))
    rfi = _check([faces], 64, 0.001, 100.0, expect_visible=True)
    assert not np.isin(rfi, (0, 1, 2, 5)).any() and (rfi == 4).any() and (rfi == 3).any()


def test_raster_mixed_depth_face_and_batch_of_unequal_content():
    # a triangle whose vertices straddle the far plane is tested per pixel; the second image is empty
    img0 = [[[-0.8, -0.8, 50.0], [0.8, -0.8, 150.0], [0.0, 0.9, 90.0]], _tri(-0.1, -0.1, 0.1, -0.1, 0.0, 0.1, 0.0009)]
    img1 = [_tri(3, 3, 4, 3, 3, 4, 1.0), _tri(3, 3, 4, 3, 3, 4, 1.0)]
    rfi = _check([img0, img1], 100, 0.001, 100.0, expect_visible=True)
    assert (rfi[1] == -1).all() and (rfi[0] == 0).any() and not (rfi[0] == 1).any()


# ----------------------------------------------------------------------------- C ABI contract violations
def test_c_abi_rejects_null_pointers_and_bad_sizes_without_launching():
    """include/sln_hip.h: 'negative SLN_E_* code for a contract violation (nothing is launched in that case)'."""
    import ctypes as C
    Lm = pkg("_lib")
    L = Lm.lib()
    st = Lm.current_stream_ptr()
    x = torch.zeros(64, device="cuda")
    p = Lm.ptr(x)
    assert L.sln_device_ok() == 0
    assert L.sln_raster_forward(None, 1, 4, 64, 0.1, 100.0, p, p, p, p, st) < 0
    assert L.sln_scene_forward(None, None, 1, 4, 64, 3, None, None, 0.1, 0.001, 100.0, 1e-3, None, None, st) < 0
    assert L.sln_spade_conv(None, 1, 32, 8, 8, p, p, 64, 64, 3, 0, 0.0, p, st) < 0
    assert L.sln_layernorm_stats(None, 1, 64, 1e-5, None, None, st) < 0
    # round-2 SPADE entry points: null operands, an odd image under a read-through upsampling, a row that is no whole float4
    assert L.sln_spade_conv_sums(None, 1, 32, 8, 8, p, p, 64, 64, 3, 0, 0.0, p, None, None, st) < 0
    assert L.sln_spade_modulate_up(p, 1, 32, 7, 8, p, p, 32, 64, p, 1, p, 0, 0.2, p, st) < 0
    assert L.sln_spade_modulate_up(p, 1, 32, 8, 8, p, None, 32, 64, p, 0, p, 0, 0.2, p, st) < 0
    assert L.sln_layernorm_finalize(None, 1, 64, 1, 1e-5, p, st) < 0 and L.sln_layernorm_finalize(p, 1, 1, 1, 1e-5, p, st) < 0
    assert L.sln_block_tail(p, 0, p, 1, 8, 4, 6, None, p, p, p, -1, p, None, 1, 1e-5, None, st) < 0          # W % 4
    assert L.sln_block_tail(p, 0, p, 1, 12, 4, 4, None, p, p, p, -1, p, None, 1, 1e-5, None, st) < 0         # C % 8
    assert L.sln_block_tail(p, 0, p, 1, 8, 4, 4, None, p, p, p, 0, p, None, 1, 1e-5, p, st) < 0              # stats without accumulators
    assert L.sln_spade_apply_up(p, 1, p, 1, 32, 3, 4, 64, p, 0, 0.2, p, st) < 0
    assert L.sln_refine_loss_forward(None, p, p, None, p, p, p, st) < 0
    d = Lm.SlnRefineLoss()                                                       # all-zero descriptor
    assert L.sln_refine_loss_forward(d, p, p, p, p, p, p, st) < 0 and L.sln_refine_loss_backward(d, p, p, p, st) < 0
    assert L.sln_refine_loss_workspace_bytes(0, 256, 96, 4, 40, 29) < 0 and L.sln_refine_loss_workspace_bytes(1, 256, 96, 9, 40, 29) < 0
    assert L.sln_place_forward(None, p, p, None, p, p, p, st) < 0
    pl = Lm.SlnPlacement()
    assert L.sln_place_forward(pl, p, p, None, p, p, p, st) < 0 and L.sln_place_backward(pl, p, p, None, p, None, p, p, st) < 0
    assert L.sln_graph_plan(None, None, 4, None, None, st) < 0
    assert L.sln_vae_set_batch(None, None, st) < 0 and L.sln_vae_train_step(None, None, 0.1, 1e-4, None, 0, 1, st) < 0
    assert L.sln_vae_set_training(None, 1) < 0 and L.sln_vae_adam_reset(None, 3, st) < 0
    cfg = Lm.SlnVaeConfig()                                                      # zeroed config: unsupported, no engine is created
    h = C.c_void_p()
    assert L.sln_vae_create(C.byref(cfg), C.byref(h)) < 0 and not h.value
    torch.cuda.synchronize()                                                     # nothing faulted asynchronously
