"""GPU parity tests of the scene-graph VAE path (run with -m gpu on an MI355X).

Everything goes through the C ABI (libsln_hip.so via ctypes); the checker is the CPU oracle
(oracle/vae_ref.py) and the fixtures the reference itself produced (tests/golden/).
"""
import ctypes as C

import os

import numpy as np
import pytest
import torch

from conftest import load_golden, pkg
from parity import assert_adam_close, assert_close, assert_close_conditioned, max_err

pytestmark = pytest.mark.gpu

from oracle import vae_ref                                   # noqa: E402
from oracle.gen_golden import KL_WEIGHT, VAE_CASES           # noqa: E402

HIP_CASES = list(VAE_CASES)          # every reference-generated fixture, decoder_cat / use_attr off included


def _lib():
    return pkg("_lib")


def _model(cfg, sd):
    M = pkg("host.Sg2ScVAE_model")
    m = M.Sg2ScVAEModel(**cfg.model_kwargs())
    m.load_state_dict({k: v.clone() for k, v in sd.items()})
    return m.cuda()


def _dev(*ts):
    return [t.cuda() for t in ts]


# ----------------------------------------------------------------------------- GEMM family
@pytest.mark.parametrize("M,N,K", [(4096, 256, 384), (4096, 640, 256), (2048, 128, 256), (12, 640, 256),
                                   (8, 6, 256), (100, 48, 128), (333, 24, 36)])
@pytest.mark.parametrize("tile", [-1, 0, 1, 2])
def test_linear_forward(M, N, K, tile):
    L = _lib()
    g = torch.Generator().manual_seed(M * 7 + N)
    x = torch.randn(M, K, generator=g); W = torch.randn(N, K, generator=g) / K ** 0.5; b = torch.randn(N, generator=g)
    xd, Wd, bd = _dev(x, W, b)
    y = torch.empty(M, N, device="cuda"); sums = torch.zeros(2, N, dtype=torch.float64, device="cuda")
    L.check(L.lib().sln_linear_forward(L.ptr(xd), M, K, L.ptr(Wd), L.ptr(bd), L.ptr(y), N, L.ptr(sums), tile,
                                       L.current_stream_ptr()), "sln_linear_forward")
    ref = (x.double() @ W.double().t() + b.double())
    assert_close(y.cpu().numpy(), ref.numpy(), "y", rtol=2e-6, atol=1e-6)
    assert_close(sums[0].cpu().numpy(), ref.sum(0).numpy(), "colsum", rtol=1e-5, atol=1e-5)
    assert_close(sums[1].cpu().numpy(), (ref * ref).sum(0).numpy(), "colsumsq", rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("R,N,K", [(4096, 640, 256), (4096, 256, 384), (2048, 128, 256), (13, 8, 36), (100, 24, 256)])
def test_linear_wgrad(R, N, K):
    L = _lib()
    g = torch.Generator().manual_seed(R + N)
    gq = torch.randn(R, N, generator=g); x = torch.randn(R, K, generator=g)
    gd, xd = _dev(gq, x)
    dW = torch.zeros(N, K, device="cuda"); db = torch.zeros(N, device="cuda")
    L.check(L.lib().sln_linear_wgrad(L.ptr(gd), L.ptr(xd), R, N, K, L.ptr(dW), L.ptr(db), L.current_stream_ptr()), "wgrad")
    assert_close(dW.cpu().numpy(), (gq.double().t() @ x.double()).numpy(), "dW", rtol=1e-5, atol=1e-5)
    assert_close(db.cpu().numpy(), gq.double().sum(0).numpy(), "db", rtol=1e-5, atol=1e-5)


# ----------------------------------------------------------------------------- golden fixtures
def _fp64_grads(cfg, sd64, b64, eps64, perturb_seed=None, kl_weight=KL_WEIGHT, l1_sign=None):
    """Gradients of one oracle iteration.  ``perturb_seed``: every floating parameter is first moved by a relative 2^-23
    (one fp32 rounding) in a random direction - the spread of the exact gradient under such perturbations is the
    conditioning of the problem itself (train-mode BatchNorm over 8 rows: ReLU masks and L1 signs flip).
    ``l1_sign``: the L1 term is taken with THIS sign pattern of (boxes_pred - boxes) instead of its own (d|x|/dx = sign(x) is
    discontinuous: a residual smaller than the forward error of a path flips it) - the gradient an exact backward pass gives
    for the forward that path actually computed."""
    s = {k: v.clone() for k, v in sd64.items()}
    if perturb_seed is not None:
        gen = torch.Generator().manual_seed(perturb_seed)
        for k in vae_ref.trainable_keys(cfg):
            s[k] = s[k] * (1.0 + (torch.rand(s[k].shape, generator=gen, dtype=s[k].dtype) * 2 - 1) * 2.0 ** -23)
    keys = vae_ref.trainable_keys(cfg)
    if l1_sign is None:
        m = {k: torch.zeros_like(s[k]) for k in keys}
        v = {k: torch.zeros_like(s[k]) for k in keys}
        _, _, grads = vae_ref.train_step(s, cfg, b64, eps64, kl_weight, m, v, 1)
        return {k: (grads[k].numpy() if k in grads else np.zeros(tuple(s[k].shape))) for k in keys}
    for k in keys:
        s[k].requires_grad_(True)
    mu, lv, bp, ap = vae_ref.forward(s, cfg, *b64, eps64, True)
    total, parts = vae_ref.losses(cfg, b64[2], bp, b64[3], ap, mu, lv, kl_weight)
    total = total - parts["bbox_pred"] + (l1_sign.to(bp.dtype) * (bp - b64[2])).sum() / bp.numel()
    gr = torch.autograd.grad(total, [s[k] for k in keys], allow_unused=True)
    return {k: (g.detach().numpy() if g is not None else np.zeros(tuple(s[k].shape))) for k, g in zip(keys, gr)}


def _trace_oracle(sd, cfg, batch, eps, training):
    vae_ref.TRACE = {}
    try:
        out = vae_ref.forward(sd, cfg, *batch, eps, training=training)
        tr = vae_ref.TRACE
    finally:
        vae_ref.TRACE = None
    return out, tr


def _tap_report(model, cfg, trace):
    """max-abs error of every stored pre-activation against the oracle's trace (debug aid)."""
    rows = []
    L = cfg.gconv_num_layers
    for net, tag in enumerate(("ec", "dc")):
        seen = {}
        for l in range(L):
            j = 0 if cfg.gconv_mode == "recurrent" else l
            pre = "gconv_net_%s.gconvs.%d" % (tag, j)
            k = seen.get(pre, 0); seen[pre] = k + 1
            per = 3 if cfg.mlp_normalization == "batch" else 2
            names = [pre + ".net1.0", pre + ".net1.%d" % per, pre + ".pooled", pre + ".net2.0", pre + ".net2.%d" % per]
            for what, nm in enumerate(names):
                ref = trace[nm][k].numpy()
                got = model.tap(net * L + l, what).cpu().numpy()
                e, s = max_err(got, ref)
                rows.append("%s[%d] %-8s err %.2e scale %.2e" % (tag, l, ["A1", "A2", "M", "A3", "A4"][what], e, s))
    return "\n".join(rows)


@pytest.mark.parametrize("name", HIP_CASES)
def test_golden_eval_and_train(name):
    g = load_golden(name)
    cfg = vae_ref.VaeConfig(**VAE_CASES[name][0])
    sd0 = vae_ref.init_state(cfg, seed=42)
    model = _model(cfg, sd0)
    ins = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("in:")}
    objs, triples, boxes, angles, attrs, eps = _dev(ins["objs"], ins["triples"], ins["boxes"], ins["angles"], ins["attrs"], ins["eps"])
    ill = name == "vae_c1_full"       # BatchNorm over 8 rows: see tests/parity.py
    batch_cpu = (ins["objs"], ins["triples"], ins["boxes"], ins["angles"], ins["attrs"])

    def cmp(got, key, ref64=None, ref32=None, **kw):
        if ill and ref64 is not None:
            assert_close_conditioned(got.detach().cpu().numpy(), ref64, ref32, name + ":" + key, **kw)
        else:
            assert_close(got.detach().cpu().numpy(), g[key], name + ":" + key, **kw)

    # fp64 evaluation of the oracle for the conditioned comparison
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    b64 = (ins["objs"], ins["triples"], ins["boxes"].double(), ins["angles"], ins["attrs"])

    # ---- eval mode
    model.eval()
    with torch.no_grad():
        mu, lv, bp, ap = model(objs, triples, boxes, angles, attrs, None, eps=eps)
        r64 = vae_ref.forward(sd64, cfg, *b64, ins["eps"].double(), training=False)
    for got, key, r in ((mu, "eval_mu", r64[0]), (lv, "eval_logvar", r64[1]), (bp, "eval_boxes_pred", r64[2]),
                        (ap, "eval_angles_pred", r64[3])):
        cmp(got, key, r.numpy(), g[key])

    # ---- train mode forward + loss + backward
    model.train()
    mu, lv, bp, ap = model(objs, triples, boxes, angles, attrs, None, eps=eps)
    (_, trace) = _trace_oracle({k: v.clone() for k, v in sd0.items()}, cfg, batch_cpu, ins["eps"], True)
    report = _tap_report(model, cfg, trace)
    sd64t = {k: v.clone() for k, v in sd64.items()}
    r64 = vae_ref.forward(sd64t, cfg, *b64, ins["eps"].double(), training=True)
    try:
        for got, key, r in ((mu, "mu", r64[0]), (lv, "logvar", r64[1]), (bp, "boxes_pred", r64[2]), (ap, "angles_pred", r64[3])):
            cmp(got, key, r.detach().numpy(), g[key])
    except AssertionError as e:
        raise AssertionError(str(e) + "\n" + report)
    U = pkg("host.utils")
    import types
    total, parts = U.calculate_model_losses(types.SimpleNamespace(use_AE=cfg.use_AE), model, boxes, bp, angles, ap,
                                            mu=mu, logvar=lv, KL_weight=KL_WEIGHT)
    # losses of the fp64 oracle (BASELINE config c1 - BatchNorm over 8 rows - is held to them within the reference's own fp32
    # distance, like its forward outputs; the well-conditioned fixtures to the reference's numbers directly)
    t64, p64 = vae_ref.losses(cfg, b64[2], r64[2], b64[3], r64[3], r64[0], r64[1], KL_WEIGHT)
    if ill:
        assert_close_conditioned(total.item(), float(t64), g["total_loss"], name + ":total")
        for k, v in parts.items():
            assert_close_conditioned(v, float(p64[k]), g["loss_" + k], name + ":loss_" + k)
    else:
        assert_close(total.item(), g["total_loss"], name + ":total")
        for k, v in parts.items():
            assert_close(v, g["loss_" + k], name + ":loss_" + k)
    model.zero_grad()
    total.backward()
    torch.cuda.synchronize()
    # gradients of EVERY parameter against the fp64 oracle, with the reference's own fp32 distance from it as slack (the
    # reference's fp32 gradient can be 1e-2 off in a fixture with a near-constant BatchNorm column, vae_small_2d).  The
    # full-width fixture stores the reference's gradients only for tensors <= 4096 elements (+ checksums of all of them):
    # for the others the fp32 oracle - proven equal to the reference on every fixture by tests/test_oracle_vae.py - stands in.
    # c1: the L1 signs of the HIP forward (a residual below the forward's 1e-3 conditioning noise may come out with the other
    # sign, and ONE flipped sign moves box_net's bias gradient by 2 / 48): the oracle gradients are taken for that sign pattern
    l1s = torch.sign(bp.detach().cpu().double() - b64[2]) if ill else None
    sign_as_fixture = (not ill) or bool((l1s == torch.sign(torch.from_numpy(g["boxes_pred"]).double() - b64[2])).all())
    g64 = _fp64_grads(cfg, sd64, b64, ins["eps"].double(), l1_sign=l1s)
    g32 = _fp64_grads(cfg, {k: v.clone() for k, v in sd0.items()}, batch_cpu, ins["eps"], l1_sign=l1s)
    gscale = max(float(np.abs(v).max()) for v in g64.values())
    named = dict(model.named_parameters())
    bad = []
    # c1 (BatchNorm over 8 rows): any fp32 evaluation of the forward is 1e-3..1e-2 away from the exact one (the reference's own
    # boxes_pred moves by 1e-2 between 1 and 8 CPU threads), ReLU masks and L1 signs flip, and the gradient moves by O(1) of its
    # scale.  The yardstick is the reference path's own fp32 scatter: its stored gradient, the fp32 oracle on 1 thread and on all
    # of them, and the fp32 oracle with every parameter moved by one ulp (five rounding trajectories), all measured against the
    # fp64 gradient.  The TIGHT check of the backward kernels at this shape is test_tight_gradients_when_no_threshold_is_near.
    spread = {}
    if ill:
        sd32 = {k: v.clone() for k, v in sd0.items()}
        samples = [_fp64_grads(cfg, sd32, batch_cpu, ins["eps"], perturb_seed=ps, l1_sign=l1s) for ps in (1, 2, 3, 4, 5)]
        nthr = torch.get_num_threads()
        torch.set_num_threads(1)
        samples.append(_fp64_grads(cfg, sd32, batch_cpu, ins["eps"], l1_sign=l1s))
        torch.set_num_threads(nthr)
        for gp in samples + [g32]:
            for k in g64:
                spread[k] = max(spread.get(k, 0.0), float(np.abs(gp[k] - g64[k]).max()))
    for k, r64g in g64.items():
        ref32 = g["grad:" + k] if (("grad:" + k) in g.files and sign_as_fixture) else g32[k]
        try:
            got = named[k].grad.cpu().numpy()
            if ill:
                err, scale = max_err(got, r64g)
                noise = max(max_err(ref32, r64g)[0], spread[k])
                assert np.isfinite(err) and err <= 5e-6 * gscale + 1e-4 * scale + 4.0 * noise, \
                    "%s:grad:%s: err %.3e, scale %.3e, conditioning spread %.3e" % (name, k, err, scale, noise)
            else:
                assert_close_conditioned(got, r64g, ref32, name + ":grad:" + k, atol=5e-6 * gscale, k=4.0)
        except AssertionError as e:
            bad.append(str(e))
    assert not bad, "\n".join(bad[:40]) + "\n" + report
    for k in g.files:
        if k.startswith("gsum:") and sign_as_fixture:  # the reference's own checksums of every gradient tensor (full-width fixture)
            gg = named[k[5:]].grad.double().cpu()
            got = np.array([float(gg.sum()), float(gg.abs().sum()), float((gg * gg).sum())])
            r64g = torch.from_numpy(g64[k[5:]])
            ref64 = np.array([float(r64g.sum()), float(r64g.abs().sum()), float((r64g * r64g).sum())])
            # the reference's own checksums: sum |g| of ours within the reference's distance from fp64 plus the spread above
            slack = 4.0 * (abs(g[k][1] - ref64[1]) + spread.get(k[5:], 0.0) * gg.numel())
            assert abs(got[1] - ref64[1]) <= 1e-3 * ref64[1] + slack, "%s:%s sum|g| %.4e vs %.4e" % (name, k, got[1], ref64[1])
    for k in g.files:
        if k.startswith("buf:"):
            if ill:
                assert_close_conditioned(model.state_dict()[k[4:]].double().cpu().numpy(), sd64t[k[4:]].numpy(), g[k], name + ":" + k)
            else:
                assert_close(model.state_dict()[k[4:]].cpu().numpy(), g[k], name + ":" + k)


@pytest.mark.parametrize("name", HIP_CASES)
def test_golden_fused_train_step(name):
    """sln_vae_train_step (zero_grad + fwd + loss + bwd + Adam in one call) against the reference's
    losses, BatchNorm buffers and Adam-updated parameters."""
    g = load_golden(name)
    cfg = vae_ref.VaeConfig(**VAE_CASES[name][0])
    model = _model(cfg, vae_ref.init_state(cfg, seed=42))
    ins = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("in:")}
    objs, triples, boxes, angles, attrs, eps = _dev(ins["objs"], ins["triples"], ins["boxes"], ins["angles"], ins["attrs"], ins["eps"])
    model.train()
    for use_graph in (False,):
        losses = model.train_step(objs, triples, boxes, angles, attrs, kl_weight=KL_WEIGHT, lr=1e-4, eps=eps,
                                  use_graph=use_graph).cpu().numpy()
    ill = name == "vae_c1_full"       # BatchNorm over 8 rows: conditioned comparison against the fp64 oracle (tests/parity.py)
    sd = model.state_dict()
    gscale = max(float(np.abs(g[k]).max()) for k in g.files if k.startswith("grad:"))
    sd0 = vae_ref.init_state(cfg, seed=42)
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
    b64 = (ins["objs"], ins["triples"], ins["boxes"].double(), ins["angles"], ins["attrs"])
    g64 = _fp64_grads(cfg, sd64, b64, ins["eps"].double())
    sd64t = {k: v.clone() for k, v in sd64.items()}
    r64 = vae_ref.forward(sd64t, cfg, *b64, ins["eps"].double(), training=True)       # also leaves the fp64 BatchNorm buffers in sd64t
    t64, p64 = vae_ref.losses(cfg, b64[2], r64[2], b64[3], r64[3], r64[0], r64[1], KL_WEIGHT)
    refs = [("total", losses[3], float(t64), g["total_loss"]), ("bbox", losses[0], float(p64["bbox_pred"]), g["loss_bbox_pred"]),
            ("angle", losses[1], float(p64["angle_pred"]), g["loss_angle_pred"])]
    if not cfg.use_AE:
        refs.append(("kld", losses[2], float(p64["KLD_Gauss"]), g["loss_KLD_Gauss"]))
    for nm, got, r6, r32 in refs:
        if ill:
            assert_close_conditioned(got, r6, r32, name + ":" + nm)
        else:
            assert_close(got, r32, name + ":" + nm)
    for k in g.files:
        if k.startswith("buf:"):
            if ill:
                assert_close_conditioned(sd[k[4:]].double().cpu().numpy(), sd64t[k[4:]].numpy(), g[k], name + ":" + k)
            else:
                assert_close(sd[k[4:]].cpu().numpy(), g[k], name + ":" + k)
        elif k.startswith("adam:") and ("grad:" + k[5:]) in g.files:
            gnoise = float(np.abs(g["grad:" + k[5:]] - g64[k[5:]]).max())
            assert_adam_close(sd[k[5:]].cpu().numpy(), g[k], g["grad:" + k[5:]], name + ":" + k, gscale=gscale, gnoise=gnoise)


def test_graph_replay_matches_eager():
    """hipGraph replay of the training iteration == eager launches (same inputs, same state).  Three steps at lr 1e-3: the
    first two losses must agree tightly.  Adam's first step moves every parameter by +-lr whatever |g| is, so parameters whose
    gradient is rounding noise take random-sign steps in BOTH paths (tools/replay_stress.py: 40 runs of either path give the
    same three or four discrete third-step losses, 3e-4 apart) - the third loss and the parameters are bounded accordingly."""
    cfg = vae_ref.VaeConfig(embedding_dim=32, gconv_num_layers=2)
    b = vae_ref.synth_batch(8, 12, 20, seed=3, cfg=cfg)
    eps = torch.randn(b[0].shape[0], cfg.embedding_dim, generator=torch.Generator().manual_seed(0))
    outs = []
    for use_graph in (False, True):
        model = _model(cfg, vae_ref.init_state(cfg, seed=1)).train()
        dev = _dev(*b[:5], eps)
        s = torch.cuda.Stream()
        ls = []
        with torch.cuda.stream(s):
            for _ in range(3):
                ls.append(model.train_step(*dev[:5], kl_weight=0.1, lr=1e-3, eps=dev[5], use_graph=use_graph).clone())
        torch.cuda.synchronize()
        outs.append((np.stack([l.cpu().numpy() for l in ls]), model.flat_params.cpu().numpy().copy()))
    assert_close(outs[1][0][:2], outs[0][0][:2], "losses of steps 1-2, graph vs eager", rtol=1e-6)
    assert_close(outs[1][0][2], outs[0][0][2], "losses of step 3, graph vs eager", rtol=5e-4)
    assert outs[0][0][2, 3] < outs[0][0][1, 3] < outs[0][0][0, 3]
    d = np.abs(outs[1][1] - outs[0][1])
    assert d.max() <= 2.05 * 1e-3 * 3, d.max()


@pytest.mark.parametrize("use_graph", [False, True])
def test_two_half_iteration_equals_the_whole_one(use_graph):
    """SLN_TRAIN_UPTO_DECODER + SLN_TRAIN_ENCODER_BWD == SLN_TRAIN_BACKWARD, and after the first half the decoder-side
    gradients (the upper part of flat_grads, what the overlapped all-reduce ships first) are already final."""
    cfg = vae_ref.VaeConfig(embedding_dim=32, gconv_num_layers=2)
    b = vae_ref.synth_batch(8, 12, 20, seed=4, cfg=cfg)
    eps = torch.randn(b[0].shape[0], cfg.embedding_dim, generator=torch.Generator().manual_seed(1))
    dev = _dev(*b[:5], eps)
    model = _model(cfg, vae_ref.init_state(cfg, seed=2)).train()
    split = model.decoder_grad_offset
    names = [n for n, _ in model.named_parameters()]
    assert 0 < split < model.flat_grads.numel() and any(n.startswith("gconv_net_dc.") for n in names)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for rep in range(2):                                          # second round replays the captured graphs
            l0 = model.train_step(*dev[:5], kl_weight=0.1, lr=1e-3, eps=dev[5], use_graph=use_graph, with_adam=False)
            whole = model.flat_grads.clone()
            model.flat_grads.fill_(7.0)                               # the first half must zero and rewrite everything it owns
            l1 = model.train_step_begin(*dev[:5], kl_weight=0.1, lr=1e-3, eps=dev[5], use_graph=use_graph)
            upper = model.flat_grads[split:].clone()
            model.train_step_finish(use_graph=use_graph)
            halves = model.flat_grads.clone()
            torch.cuda.synchronize()
            gs = float(whole.abs().max())
            assert_close(l1.cpu().numpy(), l0.cpu().numpy(), "losses", rtol=1e-6)
            assert_close(upper.cpu().numpy(), whole[split:].cpu().numpy(), "decoder half after the first half", rtol=1e-5, atol=1e-6 * gs)
            assert_close(halves.cpu().numpy(), whole.cpu().numpy(), "all gradients after both halves", rtol=1e-5, atol=1e-6 * gs)
            assert float(whole[:split].abs().max()) > 0
    # the second half alone is refused
    fresh = _model(cfg, vae_ref.init_state(cfg, seed=2)).train()
    fresh._set_batch(*dev[:5])
    with pytest.raises(Exception):
        fresh.train_step_finish(use_graph=False)


# ----------------------------------------------------------------------------- large batches: a size-independent property
def test_replicated_batch_gives_the_same_losses_and_gradients():
    """bench.py's large-batch points (1 024 / 4 096 graphs per step) have no oracle run of their size.  The property that carries
    parity over: a batch made of c copies of a 64-graph batch (graphs stay disjoint, eps repeated) has the same train-mode BatchNorm
    statistics, hence the same per-row activations, the same mean-reduced losses and the same parameter gradients, whatever c is -
    on launches of c times the rows (row chunks of the wgrads, 4 096-row-tile grids).  4 copies = 256 graphs is the size
    test_c2_full_size_train_step_vs_oracle holds against the oracle; 32 copies = 2 048 graphs must reproduce it to fp32 rounding of
    the SUMS (every row is computed by the same kernels).  Against the single 64-graph batch only the losses are tight: that size
    takes other kernel bodies (16 x 16 MFMA tiles, paired head launches) whose K sums round differently, and one ReLU mask or L1 sign
    decided the other way moves a gradient entry by a whole term (the 1e-2-level sensitivity documented in the c2 test)."""
    cfg = vae_ref.VaeConfig()
    sd = vae_ref.init_state(cfg, seed=42, scale=0.25)       # quarter-scale weights: see test_c2_full_size_train_step_vs_oracle
    objs, triples, boxes, angles, attrs = vae_ref.synth_batch(64, 32, 64, seed=0, cfg=cfg)[:5]
    O = objs.shape[0]
    eps = torch.from_numpy(np.random.default_rng(1).standard_normal((O, cfg.embedding_dim)).astype(np.float32))
    off = torch.zeros_like(triples); off[:, 0] = 1; off[:, 2] = 1

    def run(c):
        b = (objs.repeat(c), torch.cat([triples + r * O * off for r in range(c)]), boxes.repeat(c, 1), angles.repeat(c), attrs.repeat(c),
             eps.repeat(c, 1))
        model = _model(cfg, sd).train()
        d = _dev(*b)
        losses = model.train_step(*d[:5], kl_weight=0.1, lr=1e-4, eps=d[5], use_graph=False).cpu().numpy()
        return (losses, {k: p.grad.detach().cpu().numpy().copy() for k, p in model.named_parameters()},
                {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()})

    (l1, g1, _), (l4, g4, s4), (l32, g32, s32) = run(1), run(4), run(32)
    np.testing.assert_allclose(l4, l1, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(l32, l4, rtol=2e-6, atol=1e-7)
    gs = max(float(np.abs(g).max()) for g in g4.values())

    def rel_l2(a, b):
        return np.sqrt(sum(float(((a[k] - b[k]).astype(np.float64) ** 2).sum()) for k in b) / sum(float((b[k].astype(np.float64) ** 2).sum()) for k in b))

    import os
    errs = []
    for k in g4:
        # relative to the tensor's own scale, with a floor of 1e-4 of the largest gradient.  A bias whose Linear feeds a BatchNorm
        # (every hidden Linear, and box_embeddings through the first net1) has an exactly-zero gradient - what is computed for it
        # is the rounding noise of a sum over all rows, which grows with the rows: biases are held to the largest gradient instead
        # of their own size
        floor = gs if k.endswith(".bias") else 1e-4 * gs
        errs.append((float(np.abs(g32[k] - g4[k]).max()) / max(float(np.abs(g4[k]).max()), floor), k))
    errs.sort(reverse=True)
    worst = (errs[0][1], errs[0][0])
    if os.environ.get("SLN_TEST_SPREAD_REPORT"):
        print("2048 vs 256 graphs: worst", errs[:6], "rel L2", rel_l2(g32, g4), "| 256 vs 64 graphs: rel L2", rel_l2(g4, g1))
    assert worst[1] < 2e-5, "gradient of %s differs by %.2e of its scale between 4 and 32 copies of the batch" % worst
    assert rel_l2(g32, g4) < 4e-6      # (2.3e-6 since round 6: both passes' wgrads share two launches, whose row chunks are planned over more problems)
    assert rel_l2(g4, g1) < 2e-2       # other kernel bodies at 64 graphs, see above
    # the BatchNorm running statistics (momentum update of the same batch statistics; the unbiased variance's n / (n - 1) differs
    # by 1 / rows between the two sizes).  Not the updated parameters: the first Adam step moves an entry by lr * g / |g|, so the
    # zero-gradient biases above move by +-1e-4 on the sign of their rounding noise - in any implementation
    for k in s4:
        if "num_batches" in k:
            assert int(s4[k]) == int(s32[k])
        elif "running_var" in k:
            np.testing.assert_allclose(s32[k], s4[k], rtol=2e-4, atol=1e-7, err_msg=k)
        elif "running_mean" in k:
            np.testing.assert_allclose(s32[k], s4[k], rtol=1e-5, atol=1e-7, err_msg=k)


# ----------------------------------------------------------------------------- BASELINE config c2
@pytest.mark.parametrize("n_graphs,n_obj,n_tri", [(64, 32, 64), (256, 32, 64), (7, 23, 37)])
def test_c2_full_size_train_step_vs_oracle(n_graphs, n_obj, n_tri):
    """Batch=64 x (32 objects, 64 triples) at train.py defaults: O=2048, T=4096 (BASELINE.json configs[1]); 256 graphs: the
    large-tile GEMM variants and the separate dgrad / wgrad launches the engine switches to above the batch-64 sizes."""
    cfg = vae_ref.VaeConfig()
    # other than the BASELINE batch: with unit-scale weights the posterior heads give |logvar| ~ 16 and z = eps * exp(logvar / 2) + mu reaches
    # 2e4 - the decoder input then amplifies the encoder's fp32 drift a hundredfold in ANY fp32 evaluation; quarter-scale
    # weights (BatchNorm renormalises every hidden layer, only the head outputs shrink) keep the comparison meaningful
    sd = vae_ref.init_state(cfg, seed=42, scale=1.0 if n_graphs == 64 else 0.25)
    batch = vae_ref.synth_batch(n_graphs, n_obj, n_tri, seed=0, cfg=cfg)        # (7, 23, 37): O = 161, T = 259 - no dimension a multiple of 4
    O = batch[0].shape[0]
    eps = torch.from_numpy(np.random.default_rng(1).standard_normal((O, cfg.embedding_dim)).astype(np.float32))
    if n_graphs != 64:
        # d L1 / d boxes_pred = sign(boxes_pred - boxes) / numel: a residual within fp32 rounding of 0 gets its sign from rounding,
        # and ONE flipped sign moves box_net's bias gradient by 2 / numel (seen: 2e-4 of its scale at 256 graphs, 1.5 % at 7
        # graphs) and every gradient behind it.  Move the targets of the few residuals close to 0 away from the prediction
        # (fp64 evaluation of the oracle decides) so that no sign is ambiguous.
        sdx = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        boxes = batch[2].clone()
        for _ in range(4):
            with torch.no_grad():
                bp64 = vae_ref.forward(sdx, cfg, batch[0], batch[1], boxes.double(), batch[3], batch[4], eps.double(), training=True)[2]
            res = (bp64 - boxes.double()).abs()
            band = 2e-4 + 2e-5 * float(bp64.abs().max())
            if not (res < band).any():               # moving targets moves every prediction a little (train-mode BatchNorm): the
                break                                # shifted band is 10x wider than the one that must end up empty
            boxes = torch.where(res < 10 * band, boxes + 100 * band, boxes)
        assert not (res < band).any()
        batch = (batch[0], batch[1], boxes) + tuple(batch[3:])
    model = _model(cfg, sd).train()
    dev = _dev(*batch[:5], eps)
    mu, lv, bp, ap = model(*dev[:5], None, eps=dev[5])
    sdr = {k: v.clone() for k, v in sd.items()}
    (rmu, rlv, rbp, rap), trace = _trace_oracle(sdr, cfg, batch[:5], eps, True)
    report = _tap_report(model, cfg, trace)
    # 20 BatchNorm'ed stages deep, two fp32 evaluations (the reference's CPU path and ours) drift apart by
    # accumulated rounding; measure both against an fp64 evaluation of the oracle and require ours to be
    # no further from it than a small multiple of the reference path's own fp32 distance.
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    b64 = (batch[0], batch[1], batch[2].double(), batch[3], batch[4])
    with torch.no_grad():
        r64 = vae_ref.forward(sd64, cfg, *b64, eps.double(), training=True)
    try:
        for got, ref, r6, nm in ((mu, rmu, r64[0], "mu"), (lv, rlv, r64[1], "logvar"), (bp, rbp, r64[2], "boxes_pred"),
                                 (ap, rap, r64[3], "angles_pred")):
            assert_close_conditioned(got.detach().cpu().numpy(), r6.numpy(), ref.detach().numpy(), "c2:" + nm, k=4.0)
    except AssertionError as e:
        raise AssertionError(str(e) + "\n" + report)
    # gradients of one full step against CPU autograd (fp32 = the reference's path, fp64 = exact)
    sdg = {k: v.clone() for k, v in sd.items()}
    m = {k: torch.zeros_like(sdg[k]) for k in vae_ref.trainable_keys(cfg)}
    v = {k: torch.zeros_like(sdg[k]) for k in vae_ref.trainable_keys(cfg)}
    total, parts, grads = vae_ref.train_step(sdg, cfg, batch[:5], eps, 0.1, m, v, step=1)
    sdg64 = {k: v.clone() for k, v in sd64.items()}
    m64 = {k: torch.zeros_like(sdg64[k]) for k in vae_ref.trainable_keys(cfg)}
    v64 = {k: torch.zeros_like(sdg64[k]) for k in vae_ref.trainable_keys(cfg)}
    total64, _, grads64 = vae_ref.train_step(sdg64, cfg, b64, eps.double(), 0.1, m64, v64, step=1)
    model2 = _model(cfg, sd).train()
    losses = model2.train_step(*dev[:5], kl_weight=0.1, lr=1e-4, eps=dev[5], use_graph=False).cpu().numpy()
    assert_close_conditioned(losses[3], total64.numpy(), total.numpy(), "c2:total", k=4.0)
    gscale = max(float(gr.abs().max()) for gr in grads.values())
    named = dict(model2.named_parameters())
    bad = []
    # 256 graphs: a gradient entry is a sum over 8 192 / 16 384 rows of terms gated by ReLU masks; a pre-activation within fp32
    # rounding of 0 flips its mask in one fp32 evaluation and not in another and moves the entry by one term.  Measured on
    # gconv_net_dc.gconvs.1.net2.1.bias (entries = sums of largely cancelling terms): 47 of 256 entries of the reference's own
    # fp32 gradient and 78 of ours are > 1e-5 (0.4 % of the scale) from the fp64 one, the worst of ours by two flips = 8 %.
    # With 161 rows (7 graphs) one flipped term is 0.6 % of a sum and the reference's own fp32 gradients are 1 % from fp64.
    # The tight comparison is the 64-graph case; the other sizes check that the large-tile / separately launched kernels and the
    # odd shapes (no dimension a multiple of 4) produce the same tensors: max error within 2 % of the tensor's scale, or relative
    # L2 error within 10 %.
    # the yardstick at 64 graphs: the fp32 scatter of the reference path itself - its gradient and the fp32 oracle with every
    # parameter moved by one ulp (three rounding trajectories) - measured against the fp64 gradient; ONE fp32 sample
    # under-estimates the spread of a tensor by chance (the problem is chaotic at the 1e-2 level, see above)
    # (round 4) the other sizes get the same kind of yardstick, from more trajectories (their bound is a larger multiple of it, see
    # below): every size is now held to the reference's own fp32 scatter, tensor by tensor
    spread = {}
    if True:
        samples = [{k: g.numpy() for k, g in grads.items()}]
        for ps in ((1, 2, 3) if n_graphs == 64 else (1, 2, 3, 4, 5, 6)):
            samples.append(_fp64_grads(cfg, {k: v.clone() for k, v in sd.items()}, batch[:5], eps, perturb_seed=ps))
        for smp in samples:
            for k in grads64:
                if k in smp:
                    spread[k] = max(spread.get(k, 0.0), float(np.abs(smp[k] - grads64[k].numpy()).max()))
    for k, gr in grads.items():
        try:
            got, r64 = named[k].grad.cpu().numpy(), grads64[k].numpy()
            if n_graphs == 64:
                err, scale = max_err(got, r64)
                assert np.isfinite(err) and err <= 5e-6 * gscale + 1e-4 * scale + 4.0 * spread[k], \
                    "c2:grad:%s: err %.3e, scale %.3e, fp32-reference spread %.3e" % (k, err, scale, spread[k])
            else:
                # parity, conditioned: within 4x (256 graphs) / 10x (7 graphs, 161 rows) the largest distance of seven fp32 evaluations
                # of the reference path (its own gradient + six one-ulp perturbations) from the fp64 gradient - measured 1.9x and
                # 6.7x, identical over three runs; AND the size-independent sanity bound below
                err, scale = max_err(got, r64)
                _RATIOS.append(((err - 5e-6 * gscale - 1e-4 * scale) / max(spread[k], 1e-30), k, err, scale, spread[k]))
                kk = 4.0 if n_graphs >= 64 else 10.0          # 256 graphs: the batch-64 multiple (measured 1.9x); 7 graphs: 6.7x
                assert np.isfinite(err) and err <= 5e-6 * gscale + 1e-4 * scale + kk * spread[k], \
                    "c2:grad:%s: err %.3e, scale %.3e, fp32-reference spread %.3e" % (k, err, scale, spread[k])
                try:
                    assert_close_conditioned(got, r64, gr.numpy(), "c2:grad:" + k, rtol=2e-2, atol=5e-6 * gscale, k=4.0)
                except AssertionError:
                    l2 = float(np.linalg.norm(got.astype(np.float64) - r64) / max(np.linalg.norm(r64), 1e-30))
                    assert l2 <= 0.1, "c2:grad:%s: relative L2 error %.3f" % (k, l2)
        except AssertionError as e:
            bad.append(str(e))
    if _RATIOS and os.environ.get("SLN_TEST_SPREAD_REPORT"):
        _RATIOS.sort(reverse=True)
        print("SPREAD REPORT n_graphs=%d: worst (err - floor) / spread:" % n_graphs, [(round(r, 2), k, "%.2e" % e, "%.2e" % sc, "%.2e" % sp) for r, k, e, sc, sp in _RATIOS[:5]])
    del _RATIOS[:]
    assert not bad, "\n".join(bad[:40])
    for k in sdg:
        if "running" in k:
            assert_close(model2.state_dict()[k].cpu().numpy(), sdg[k].numpy(), "c2:" + k)


_RATIOS = []


def _threshold_free_state(cfg, seed, train_bn=False):
    """A state in which no ReLU pre-activation and no L1 residual sits near its threshold: weights at a tenth of their initial
    scale, every Linear bias in front of a ReLU at +8 (|W x| stays below ~1.5), posterior-head biases 0.  The network is then
    a smooth function of its parameters and fp32 evaluations agree with fp64 to ~1e-6 - unlike the random-init state, where
    ~1e2 of the ~4e8 pre-activations of a 256-graph step lie within fp32 rounding of 0 and every flipped mask moves the
    gradients by 1e-3 of their scale in ANY fp32 evaluation (the reference's included).

    `train_bn`: the variant for train-mode BatchNorm, which removes whatever the Linear in front of it adds: the +8 sits on the
    BatchNorm beta (`*.1.bias`, `*.4.bias`) instead, gamma is squeezed into [0.5, 1], the Linear biases stay free.  A normalised
    pre-activation is a z-score of its column (|.| <= sqrt(rows - 1), ~4.5 for Gaussian columns of 16 k rows), so every ReLU
    input is >= 8 - gamma |z| > 0 by a wide margin in train mode too (the test asserts that on the fp64 oracle's trace)."""
    sd = vae_ref.init_state(cfg, seed=seed, scale=0.1)
    if train_bn:
        assert cfg.mlp_normalization == "batch"
        for k, v in sd.items():
            mod = k.rsplit(".", 2)[-2] if k.count(".") >= 2 else ""
            if mod in ("1", "4") and v.dim() == 1 and k.endswith(".bias"):
                v.fill_(8.0)
            elif mod in ("1", "4") and v.dim() == 1 and k.endswith(".weight"):
                v.copy_(0.5 + (v - 0.5) * 0.5)                 # init_state draws gamma ~ U(0.5, 1.5)
        return sd
    relu_free = ("box_mean.", "box_var.", "angle_mean.", "angle_var.", "box_net.%d." % (3 if cfg.mlp_normalization == "batch" else 2),
                 "angle_net.%d." % (3 if cfg.mlp_normalization == "batch" else 2))
    for k, v in sd.items():
        if k.endswith(".bias") and v.dim() == 1 and not k.startswith(relu_free) and "embeddings" not in k:
            if cfg.mlp_normalization == "batch" and k.rsplit(".", 2)[-2] in ("1", "4"):
                continue                                   # BatchNorm beta stays 0: the Linear bias in front of it carries the +8
            v.fill_(8.0)
    return sd


@pytest.mark.parametrize("norm,training,n_graphs", [("none", True, 256), ("none", True, 1), ("batch", False, 256), ("batch", False, 1),
                                                    ("batch", True, 256), ("batch", True, 64), ("batch", True, 1)])
def test_tight_gradients_when_no_threshold_is_near(norm, training, n_graphs):
    """1e-4 on EVERY gradient of a full-width step, (a) at 256 graphs (O = 8192, T = 16384): the >= 128-row tiles and the
    separately launched dgrad / wgrad kernels the engine switches to above the batch-64 sizes, (b) at BASELINE config c1's shape
    (1 graph, 8 objects, 12 triples), without BatchNorm and with eval-mode BatchNorm (train.py:63-65 keeps training after
    model.eval()), (c) with TRAIN-mode BatchNorm (models/graph.py:14-15, the default of train.py) at 1 / 64 / 256 graphs: the +8
    sits on the BatchNorm beta there (`_threshold_free_state(train_bn=True)`), so the two-source BatchNorm-backward operand, the
    statistics epilogues, `bn_param_grads_kernel` and the running statistics get the same 1e-4 as the other modes."""
    cfg = vae_ref.VaeConfig(mlp_normalization=norm)
    train_bn = norm == "batch" and training
    sd = _threshold_free_state(cfg, seed=3, train_bn=train_bn)
    batch = list(vae_ref.synth_batch(n_graphs, 32 if n_graphs > 1 else 8, 64 if n_graphs > 1 else 12, seed=0, cfg=cfg)[:5])
    if train_bn:
        # every L1 residual ~ +-20, the sign drawn per element: with ONE sign the loss gradient is the same row vector on every row
        # and the first train-mode BatchNorm backward (N dY - sum dY - ...) annihilates it - box_net would get no gradient at all
        sg = torch.from_numpy(np.random.default_rng(7).integers(0, 2, tuple(batch[2].shape)).astype(np.float32)) * 2 - 1
        batch[2] = batch[2] + 20.0 * sg
    else:
        batch[2] = batch[2] - 100.0                         # every L1 residual positive, far from 0
    O = batch[0].shape[0]
    eps = torch.from_numpy(np.random.default_rng(1).standard_normal((O, cfg.embedding_dim)).astype(np.float32))
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    b64 = (batch[0], batch[1], batch[2].double(), batch[3], batch[4])
    keys = vae_ref.trainable_keys(cfg)
    m64 = {k: torch.zeros_like(sd64[k]) for k in keys}; v64 = {k: torch.zeros_like(sd64[k]) for k in keys}
    if train_bn:
        # the premise, checked on the fp64 oracle: every BatchNorm'ed pre-activation ends up >= 2 (no ReLU mask near its threshold)
        (_, trace) = _trace_oracle({k: v.clone() for k, v in sd64.items()}, cfg, b64, eps.double(), True)     # _ = (mu, logvar, boxes_pred, angles_pred)
        lo = 1e30
        for base, taps in trace.items():
            pre, idx = base.rsplit(".", 1)
            bn = "%s.%d" % (pre, int(idx) + 1) if idx.isdigit() else ""
            if bn + ".running_mean" not in sd64:
                continue
            for a in taps:
                z = (a - a.mean(0)) / torch.sqrt(a.var(0, unbiased=False) + 1e-5)
                lo = min(lo, float((z * sd64[bn + ".weight"] + sd64[bn + ".bias"]).min()))
        assert lo >= 1.0, "threshold-free premise: smallest BatchNorm output %.3f" % lo
        assert float((_[2] - b64[2]).abs().min()) >= 10.0
    total64, parts64, g64 = vae_ref.train_step(sd64, cfg, b64, eps.double(), 0.1, m64, v64, step=1, training=training)
    model = _model(cfg, sd).train(training)
    dev = _dev(*batch, eps)
    losses = model.train_step(*dev[:5], kl_weight=0.1, lr=1e-4, eps=dev[5], use_graph=False, with_adam=False).cpu().numpy()
    assert_close(losses[3], float(total64), "total")
    assert_close(losses[0], parts64["bbox_pred"], "bbox"); assert_close(losses[1], parts64["angle_pred"], "angle")
    assert_close(losses[2], parts64["KLD_Gauss"], "kld")
    named = dict(model.named_parameters())
    bad = []
    for k in keys:
        ref = g64[k].numpy() if k in g64 else np.zeros(tuple(sd[k].shape))
        atol = 1e-7 * max(float(np.abs(ref).max()), 1e-30) + 1e-9
        if train_bn and k.endswith(".bias") and (k[:-len("bias")] + "weight") in g64 and float(np.abs(ref).max()) < 1e-12:
            # biases whose exact gradient is 0 (fp64 returns 1e-17): a Linear bias in front of a train-mode BatchNorm (the sum over
            # rows of the BatchNorm backward vanishes identically) and, with every ReLU open, a BatchNorm beta in front of another
            # Linear -> BatchNorm (a constant shift of that Linear's output).  Any fp32 evaluation returns the rounding residue of
            # the cancelling sum: held to 1e-4 of the scale of the same module's weight gradient (same rows, same dY).
            atol = 1e-4 * float(np.abs(g64[k[:-len("bias")] + "weight"].numpy()).max())
        # 8 rows under train-mode BatchNorm (1 graph): the problem itself is that ill-conditioned - the reference's own fp32 evaluation
        # is up to 1.7e-4 from the fp64 gradient there (median 5.7e-5; ours: worst 1.3e-4, median 4.5e-5, tools/lab/bn_rows_probe.py)
        rtol = 3e-4 if (train_bn and n_graphs == 1) else 1e-4
        try:
            assert_close(named[k].grad.cpu().numpy(), ref, "grad:" + k, rtol=rtol, atol=atol)
        except AssertionError as e:
            bad.append(str(e))
    assert not bad, "\n".join(bad[:40])
    if train_bn:                                            # running statistics of the step (momentum 0.1, unbiased variance)
        got = model.state_dict()
        for k in sd64:
            if "running" in k:
                assert_close(got[k].cpu().numpy(), sd64[k].numpy(), k, rtol=1e-4, atol=1e-6)
            elif "num_batches" in k:
                assert int(got[k]) == int(sd64[k]), k


@pytest.mark.parametrize("deterministic", [False, True])
def test_wide_model_whose_wgrads_do_not_fit_one_launch_table(deterministic):
    """embedding_dim 256 (hidden 1024, the reference sets hidden = 4 x gconv_dim, Sg2ScVAE_model.py:19-20): net1's second
    Linear alone has 40 x 16 = 640 output tiles - more than one XCD list of the per-pass wgrad launch holds - and a pass has
    ~2 900.  The pass is cut into several launches by planned tile count and the 640-tile problem runs on its own; every gradient
    within 1e-4 of the fp64 oracle (threshold-free state), in the default and in the deterministic mode."""
    cfg = vae_ref.VaeConfig(embedding_dim=256, gconv_num_layers=2, mlp_normalization="none")
    sd = _threshold_free_state(cfg, seed=4)
    batch = list(vae_ref.synth_batch(3, 8, 12, seed=2, cfg=cfg)[:5])
    batch[2] = batch[2] - 100.0
    O = batch[0].shape[0]
    eps = torch.from_numpy(np.random.default_rng(1).standard_normal((O, cfg.embedding_dim)).astype(np.float32))
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    b64 = (batch[0], batch[1], batch[2].double(), batch[3], batch[4])
    keys = vae_ref.trainable_keys(cfg)
    m64 = {k: torch.zeros_like(sd64[k]) for k in keys}; v64 = {k: torch.zeros_like(sd64[k]) for k in keys}
    total64, parts64, g64 = vae_ref.train_step(sd64, cfg, b64, eps.double(), 0.1, m64, v64, step=1, training=True)
    L = _lib().lib()
    try:
        L.sln_set_deterministic(int(deterministic))
        model = _model(cfg, sd).train()
        dev = _dev(*batch, eps)
        runs = []
        for _ in range(2 if deterministic else 1):
            losses = model.train_step(*dev[:5], kl_weight=0.1, lr=1e-4, eps=dev[5], use_graph=False, with_adam=False).cpu().numpy()
            runs.append(model.flat_grads.clone())
    finally:
        L.sln_set_deterministic(0)
    if deterministic:
        assert torch.equal(runs[0], runs[1]), "deterministic mode: repeated steps must agree bit for bit"
    assert_close(losses[3], float(total64), "total")
    named = dict(model.named_parameters())
    bad = []
    for k in keys:
        ref = g64[k].numpy() if k in g64 else np.zeros(tuple(sd[k].shape))
        try:
            assert_close(named[k].grad.cpu().numpy(), ref, "grad:" + k, rtol=1e-4, atol=1e-7 * max(float(np.abs(ref).max()), 1e-30) + 1e-9)
        except AssertionError as e:
            bad.append(str(e))
    assert not bad, "\n".join(bad[:40])


def test_deep_recurrent_stack_in_deterministic_mode_has_enough_launch_slots():
    """'recurrent' weights in deterministic mode flush the wgrads once per layer (two launches each): 12 layers need 24 table
    slots per pass (the table is sized from the layer count; the reference places no limit on gconv_num_layers)."""
    cfg = vae_ref.VaeConfig(embedding_dim=16, gconv_num_layers=12, gconv_mode="recurrent", mlp_normalization="none")
    sd = _threshold_free_state(cfg, seed=4)
    batch = list(vae_ref.synth_batch(4, 8, 12, seed=2, cfg=cfg)[:5])
    batch[2] = batch[2] - 100.0
    O = batch[0].shape[0]
    eps = torch.from_numpy(np.random.default_rng(1).standard_normal((O, cfg.embedding_dim)).astype(np.float32))
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    b64 = (batch[0], batch[1], batch[2].double(), batch[3], batch[4])
    keys = vae_ref.trainable_keys(cfg)
    m64 = {k: torch.zeros_like(sd64[k]) for k in keys}; v64 = {k: torch.zeros_like(sd64[k]) for k in keys}
    total64, parts64, g64 = vae_ref.train_step(sd64, cfg, b64, eps.double(), 0.1, m64, v64, step=1, training=True)
    L = _lib().lib()
    try:
        L.sln_set_deterministic(1)
        model = _model(cfg, sd).train()
        dev = _dev(*batch, eps)
        losses = model.train_step(*dev[:5], kl_weight=0.1, lr=1e-4, eps=dev[5], use_graph=False, with_adam=False).cpu().numpy()
    finally:
        L.sln_set_deterministic(0)
    assert_close(losses[3], float(total64), "total")
    named = dict(model.named_parameters())
    for k in keys:
        ref = g64[k].numpy() if k in g64 else np.zeros(tuple(sd[k].shape))
        assert_close(named[k].grad.cpu().numpy(), ref, "grad:" + k, rtol=1e-4, atol=1e-7 * max(float(np.abs(ref).max()), 1e-30) + 1e-9)


def test_out_of_range_ids_raise_like_the_reference():
    """The reference's embedding / index ops raise IndexError on bad ids; the HIP path must not read out of bounds."""
    cfg = vae_ref.VaeConfig(embedding_dim=16, gconv_num_layers=1)
    model = _model(cfg, vae_ref.init_state(cfg, seed=0)).eval()
    b = [t.clone() for t in vae_ref.synth_batch(2, 4, 5, seed=1, cfg=cfg)[:5]]
    for which, val in ((0, 33), (1, None), (3, 24), (4, 5)):
        bad = [t.clone() for t in b]
        if which == 1:
            bad[1][0, 1] = 16                      # predicate id out of range
        else:
            bad[which][0] = val
        with pytest.raises(IndexError):
            model(*_dev(*bad), None, eps=torch.zeros(8, 16, device="cuda"))
    out = model(*_dev(*b), None, eps=torch.zeros(8, 16, device="cuda"))      # the engine still works afterwards
    assert torch.isfinite(out[2]).all()


def test_batched_sampling_matches_per_sample_decoding():
    """host/sampling.py (tensor work of testing/test_VAE.py:83-84): n decodes in one engine call == n separate ones
    == the oracle's eval-mode decoder."""
    S = pkg("host.sampling")
    cfg = vae_ref.VaeConfig(embedding_dim=16, gconv_num_layers=2)
    sd = vae_ref.init_state(cfg, seed=2)
    model = _model(cfg, sd).eval()
    objs, triples, boxes, angles, attrs, _ = vae_ref.synth_batch(3, 5, 7, seed=4, cfg=cfg)
    bp, ang, z = S.sample_layouts(model, objs.cuda(), triples.cuda(), attrs.cuda(), n_samples=5,
                                  generator=torch.Generator().manual_seed(0))
    assert bp.shape == (5, 15, 6) and ang.shape == (5, 15)
    for k in range(5):
        with torch.no_grad():
            rb, ra = vae_ref.decoder({k_: v.clone() for k_, v in sd.items()}, cfg, z[k].cpu(), objs, triples, attrs, training=False)
        assert_close(bp[k].cpu().numpy(), rb.numpy(), "boxes sample %d" % k)
        assert (ang[k].cpu() == ra.argmax(1)).float().mean() > 0.9
    mean, cov = S.posterior_stats(model, [(objs.cuda(), triples.cuda(), boxes.cuda(), angles.cuda(), attrs.cuda())])
    with torch.no_grad():
        mu, _ = vae_ref.encoder({k_: v.clone() for k_, v in sd.items()}, cfg, objs, triples, boxes, angles, attrs, training=False)
    assert_close(mean.numpy(), mu.double().mean(0).numpy(), "posterior mean", rtol=1e-4, atol=1e-5)
    assert_close(cov.numpy(), np.cov(mu.double().numpy().T), "posterior cov", rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("norm,mode,layers", [("batch", "feedforward", 3), ("none", "recurrent", 2), ("batch", "recurrent", 2)])
def test_standalone_graph_triple_conv_net(norm, mode, layers):
    """models/graph.py boundary row: GraphTripleConvNet(...).forward(obj_vecs, pred_vecs, edges) on its own, train-mode
    BatchNorm (running statistics updated) and eval mode, against the oracle's gconv_net_apply."""
    G = pkg("host.graph")
    cfg = vae_ref.VaeConfig(embedding_dim=16, gconv_num_layers=layers, gconv_mode=mode, mlp_normalization=norm)
    sd = vae_ref.init_state(cfg, seed=9)
    net = G.GraphTripleConvNet(32, num_layers=layers, hidden_dim=64, mode=mode, mlp_normalization=norm)
    sub = {k[len("gconv_net_ec."):]: v.clone() for k, v in sd.items() if k.startswith("gconv_net_ec.")}
    net.load_state_dict(sub)
    net = net.cuda()
    g = torch.Generator().manual_seed(0)
    O, T = 150, 260
    x = torch.randn(O, 32, generator=g); p = torch.randn(T, 32, generator=g)
    edges = torch.randint(0, O, (T, 2), generator=g)
    for training in (True, False):
        net.train(training)
        sdr = {k: v.clone() for k, v in sd.items()}
        with torch.no_grad():
            ro, rp = vae_ref.gconv_net_apply(sdr, cfg, "ec", x, p, edges, training)
            ho, hp = net(x.cuda(), p.cuda(), edges.cuda())
        assert_close(ho.cpu().numpy(), ro.numpy(), "new_obj training=%s" % training)
        assert_close(hp.cpu().numpy(), rp.numpy(), "new_pred training=%s" % training)
        if training and norm == "batch":
            for k, v in net.state_dict().items():
                if "running" in k or "num_batches" in k:
                    assert_close(v.cpu().numpy(), sdr["gconv_net_ec." + k].numpy(), k)
            sd = sdr                       # carry the updated running statistics into the eval comparison
    # autograd (models/graph.py:57-111,136-143 are differentiable): gradients w.r.t. both inputs and every parameter against
    # CPU autograd through the oracle, train-mode and eval-mode BatchNorm; a second backward accumulates into .grad
    keys = [k for k in sd if k.startswith("gconv_net_ec.") and sd[k].is_floating_point() and "running" not in k]
    wo = torch.randn(O, 32, generator=g); wp = torch.randn(T, 32, generator=g)
    for training in (True, False):
        net.train(training)
        net.zero_grad(set_to_none=True)
        sdr = {k: (v.clone().double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        for k in keys:
            sdr[k].requires_grad_(True)
        x1 = x.double().requires_grad_(True); p1 = p.double().requires_grad_(True)
        ro, rp = vae_ref.gconv_net_apply(sdr, cfg, "ec", x1, p1, edges, training)
        ((ro * wo.double()).sum() + (rp * wp.double()).sum()).backward()
        before = {k: v.clone() for k, v in net.state_dict().items()}
        x2 = x.cuda().requires_grad_(True); p2 = p.cuda().requires_grad_(True)
        ho, hp = net(x2, p2, edges.cuda())
        ((ho * wo.cuda()).sum() + (hp * wp.cuda()).sum()).backward()
        assert_close(ho.detach().cpu().numpy(), ro.detach().numpy(), "autograd forward obj training=%s" % training)
        gs = max(float(sdr[k].grad.abs().max()) for k in keys if sdr[k].grad is not None)
        assert_close(x2.grad.cpu().numpy(), x1.grad.numpy(), "d obj_vecs training=%s" % training, rtol=2e-4, atol=1e-5 * float(x1.grad.abs().max()))
        assert_close(p2.grad.cpu().numpy(), p1.grad.numpy(), "d pred_vecs training=%s" % training, rtol=2e-4, atol=1e-5 * float(p1.grad.abs().max()))
        named = dict(net.named_parameters())
        for k in keys:
            ref = sdr[k].grad
            if ref is None:
                continue
            got = named[k[len("gconv_net_ec."):]].grad
            assert_close(got.cpu().numpy(), ref.numpy(), "d %s training=%s" % (k, training), rtol=2e-4, atol=2e-6 * gs)
        if training:
            net.load_state_dict(before)           # the autograd forward moved the running statistics once more
    g1 = {k: v.grad.clone() for k, v in net.named_parameters()}
    ho, hp = net(x.cuda().requires_grad_(True), p.cuda(), edges.cuda())
    ((ho * wo.cuda()).sum() + (hp * wp.cuda()).sum()).backward()
    for k, v in net.named_parameters():
        assert_close(v.grad.cpu().numpy(), 2 * g1[k].cpu().numpy(), "accumulated " + k, rtol=1e-4, atol=1e-6 * float(g1[k].abs().max()) + 1e-9)


@pytest.mark.parametrize("norm", ["batch", "none"])
def test_bare_graph_triple_conv_with_other_output_dim(norm):
    """models/graph.py:36-56 allows GraphTripleConv(input_dim, output_dim != input_dim) (net1 -> 2H + Dout, net2 -> Dout).
    Forward and autograd - through torch.autograd.grad with the parameters as targets, which needs them to be inputs of the
    autograd node - against fp64 CPU autograd through the oracle's gconv_apply."""
    G = pkg("host.graph")
    torch.manual_seed(4)
    conv = G.GraphTripleConv(32, output_dim=48, hidden_dim=64, mlp_normalization=norm)
    sd = {"g." + k: v.clone() for k, v in conv.state_dict().items()}
    conv = conv.cuda().train()
    g = torch.Generator().manual_seed(1)
    O, T = 90, 170
    x = torch.randn(O, 32, generator=g); p = torch.randn(T, 32, generator=g)
    edges = torch.randint(0, O, (T, 2), generator=g)
    wo = torch.randn(O, 48, generator=g); wp = torch.randn(T, 48, generator=g)
    keys = [k for k in sd if sd[k].is_floating_point() and "running" not in k]
    sdr = {k: (v.clone().double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    for k in keys:
        sdr[k].requires_grad_(True)
    x1 = x.double().requires_grad_(True); p1 = p.double().requires_grad_(True)
    ro, rp = vae_ref.gconv_apply(sdr, "g", x1, p1, edges, 64, norm, True)
    assert ro.shape == (O, 48) and rp.shape == (T, 48)
    ref = torch.autograd.grad((ro * wo.double()).sum() + (rp * wp.double()).sum(), [x1, p1] + [sdr[k] for k in keys])
    x2 = x.cuda().requires_grad_(True); p2 = p.cuda().requires_grad_(True)
    ho, hp = conv(x2, p2, edges.cuda())
    assert_close(ho.detach().cpu().numpy(), ro.detach().numpy(), "new_obj")
    assert_close(hp.detach().cpu().numpy(), rp.detach().numpy(), "new_pred")
    named = dict(conv.named_parameters())
    prm = [named[k[2:]] for k in keys]
    got = torch.autograd.grad((ho * wo.cuda()).sum() + (hp * wp.cuda()).sum(), [x2, p2] + prm)
    assert all(q.grad is None for q in prm), "autograd.grad must not write .grad"
    gs = max(float(r.abs().max()) for r in ref[2:])
    for name, a, b in zip(["d obj_vecs", "d pred_vecs"] + keys, got, ref):
        scale = float(b.abs().max()) if name.startswith("d ") else gs
        assert_close(a.cpu().numpy(), b.numpy(), name, rtol=2e-4, atol=2e-6 * scale + 1e-9)
    with pytest.raises(ValueError):
        bad = G.GraphTripleConvNet(32, num_layers=2, hidden_dim=64)
        bad.gconvs[0].output_dim = 48
        G._gconv_autograd(bad, list(bad.gconvs), 2, x2, p2, edges.cuda(), True)


@pytest.mark.parametrize("norm", ["batch", "none"])
def test_bare_graph_triple_conv_with_widths_that_are_not_multiples_of_four(norm):
    """models/graph.py:36-56 places no constraint on input_dim / hidden_dim / output_dim.  Widths like 30 / 50 / 22 run on a
    zero-padded shadow of the layer (host/graph.py::_PaddedShadow); forward, the gradients of both inputs and of every parameter
    and the BatchNorm running statistics against fp64 CPU autograd through the oracle's gconv_apply."""
    G = pkg("host.graph")
    torch.manual_seed(5)
    D, H, Do = 30, 50, 22
    conv = G.GraphTripleConv(D, output_dim=Do, hidden_dim=H, mlp_normalization=norm)
    sd = {"g." + k: v.clone() for k, v in conv.state_dict().items()}
    conv = conv.cuda().train()
    g = torch.Generator().manual_seed(1)
    O, T = 37, 61
    x = torch.randn(O, D, generator=g); p = torch.randn(T, D, generator=g)
    edges = torch.randint(0, O, (T, 2), generator=g)
    wo = torch.randn(O, Do, generator=g); wp = torch.randn(T, Do, generator=g)
    keys = [k for k in sd if sd[k].is_floating_point() and "running" not in k]
    sdr = {k: (v.clone().double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    for k in keys:
        sdr[k].requires_grad_(True)
    x1 = x.double().requires_grad_(True); p1 = p.double().requires_grad_(True)
    ro, rp = vae_ref.gconv_apply(sdr, "g", x1, p1, edges, H, norm, True)
    ref = torch.autograd.grad((ro * wo.double()).sum() + (rp * wp.double()).sum(), [x1, p1] + [sdr[k] for k in keys])
    x2 = x.cuda().requires_grad_(True); p2 = p.cuda().requires_grad_(True)
    ho, hp = conv(x2, p2, edges.cuda())
    assert ho.shape == (O, Do) and hp.shape == (T, Do)
    assert_close(ho.detach().cpu().numpy(), ro.detach().numpy(), "new_obj")
    assert_close(hp.detach().cpu().numpy(), rp.detach().numpy(), "new_pred")
    named = dict(conv.named_parameters())
    prm = [named[k[2:]] for k in keys]
    got = torch.autograd.grad((ho * wo.cuda()).sum() + (hp * wp.cuda()).sum(), [x2, p2] + prm)
    gs = max(float(r.abs().max()) for r in ref[2:])
    for name, a, b in zip(["d obj_vecs", "d pred_vecs"] + keys, got, ref):
        assert a.shape == b.shape, name
        scale = float(b.abs().max()) if name.startswith("d ") else gs
        assert_close(a.cpu().numpy(), b.numpy(), name, rtol=2e-4, atol=2e-6 * scale + 1e-9)
    if norm == "batch":
        got_sd = conv.state_dict()
        for k in sdr:
            if "running" in k:
                assert_close(got_sd[k[2:]].cpu().numpy(), sdr[k].numpy(), k, rtol=1e-4, atol=1e-6)
    with torch.no_grad():                                  # eval / no-grad calls of the same odd-width layer
        conv.eval()
        eo, ep = conv(x.cuda(), p.cuda(), edges.cuda())
    sde = {k: (v.clone().double() if v.is_floating_point() else v.clone()) for k, v in {"g." + k: v.cpu() for k, v in conv.state_dict().items()}.items()}
    reo, rep = vae_ref.gconv_apply(sde, "g", x.double(), p.double(), edges, H, norm, False)
    assert_close(eo.cpu().numpy(), reo.numpy(), "eval new_obj"); assert_close(ep.cpu().numpy(), rep.numpy(), "eval new_pred")


def test_high_degree_room_node_beyond_the_lds_entry_cache():
    """One graph with 150 objects: the room node is incident to >= 149 triples, past the 64-entry LDS cache of the CSR
    edge kernels (vae_kernels.hip: ECACHE), so the tail of its entry list is read from global memory."""
    cfg = vae_ref.VaeConfig(embedding_dim=32, gconv_num_layers=2, mlp_normalization="none")
    sd = vae_ref.init_state(cfg, seed=5)
    parts = [vae_ref.synth_batch(1, 150, 420, seed=3, cfg=cfg), vae_ref.synth_batch(1, 9, 14, seed=4, cfg=cfg)]
    off = parts[0][0].shape[0]
    tr1 = parts[1][1].clone(); tr1[:, 0] += off; tr1[:, 2] += off
    batch = (torch.cat([parts[0][0], parts[1][0]]), torch.cat([parts[0][1], tr1]), torch.cat([parts[0][2], parts[1][2]]),
             torch.cat([parts[0][3], parts[1][3]]), torch.cat([parts[0][4], parts[1][4]]))
    deg = torch.bincount(torch.cat([batch[1][:, 0], batch[1][:, 2]]))
    assert int(deg.max()) > 128
    O = batch[0].shape[0]
    eps = torch.from_numpy(np.random.default_rng(2).standard_normal((O, cfg.embedding_dim)).astype(np.float32))
    sdg = {k: v.clone() for k, v in sd.items()}
    m = {k: torch.zeros_like(sdg[k]) for k in vae_ref.trainable_keys(cfg)}
    v = {k: torch.zeros_like(sdg[k]) for k in vae_ref.trainable_keys(cfg)}
    total, parts_l, grads = vae_ref.train_step(sdg, cfg, batch, eps, 0.1, m, v, step=1)
    model = _model(cfg, sd).train()
    dev = _dev(*batch, eps)
    losses = model.train_step(*dev[:5], kl_weight=0.1, lr=1e-4, eps=dev[5], use_graph=False).cpu().numpy()
    assert_close(losses[3], total.detach().numpy(), "total loss", rtol=1e-4)
    named = dict(model.named_parameters())
    gscale = max(float(g.abs().max()) for g in grads.values())
    for k, g in grads.items():
        assert_close(named[k].grad.cpu().numpy(), g.numpy(), "grad:" + k, rtol=1e-4, atol=1e-5 * gscale)


def test_heatmap_from_words_runs_chunked_decodes():
    """testing/test_heatmap.py:52-99: posterior samples of a worded scene graph, decoded in chunks, accumulated on the device"""
    S = pkg("host.sampling")
    cfg = vae_ref.VaeConfig(embedding_dim=16, gconv_num_layers=2)
    model = _model(cfg, vae_ref.init_state(cfg, seed=2)).eval()
    E = cfg.embedding_dim
    mean = torch.zeros(E, dtype=torch.float64); cov = torch.eye(E, dtype=torch.float64) * 0.25
    objs5 = ["bed", "desk", "cabinet", "chair", "lamp"]
    rels5 = [("bed", "behind", "desk"), ("cabinet", "left of", "bed"), ("chair", "left of", "desk"), ("lamp", "on", "desk")]
    h = S.heatmap_from_words(model, objs5, rels5, mean, cov, num_iter=700, chunk=256, container_size=40, generator=torch.Generator().manual_seed(0))
    assert h.shape == (5, 40, 40) and torch.isfinite(h).all()
    assert_close(h.sum((1, 2)).cpu().numpy(), np.ones(5), "every object's histogram sums to one", rtol=1e-5)
    # one chunk == the plain pipeline
    objs, triples, attrs = S.scene_graph_from_words(objs5, rels5, device="cuda")
    bp, _, _ = S.sample_layouts(model, objs, triples, attrs, n_samples=64, mean=mean, cov=cov, generator=torch.Generator().manual_seed(3))
    h1 = S.heatmap_from_words(model, objs5, rels5, mean, cov, num_iter=64, chunk=64, container_size=40, generator=torch.Generator().manual_seed(3))
    assert_close(h1.cpu().numpy(), S.layout_heatmap(bp, 40).cpu().numpy(), "single chunk", rtol=1e-6, atol=1e-7)


def test_device_drawn_posterior_samples_follow_the_requested_distribution():
    """testing/test_heatmap.py:52-64 draws z with np.random.multivariate_normal on the host; host/sampling.py draws eps on the device
    (the engine's Philox stream) and forms z = mean + eps L^T there.  20 000 draws: mean / covariance of z within sampling noise of
    the requested ones, the per-object centre histograms within sampling noise of the host-drawn path's (total-variation distance
    against two host-drawn runs with different seeds), the histogram kernel == the torch accumulation, and two runs differ
    (the stream advances) unless the model is re-seeded."""
    S = pkg("host.sampling")
    cfg = vae_ref.VaeConfig(embedding_dim=16, gconv_num_layers=2)
    model = _model(cfg, vae_ref.init_state(cfg, seed=2)).eval()
    E = cfg.embedding_dim
    rng = np.random.default_rng(5)
    A = rng.standard_normal((E, E)) * 0.3
    mean = torch.from_numpy(rng.standard_normal(E) * 0.2); cov = torch.from_numpy(A @ A.T + 0.05 * np.eye(E))
    objs5 = ["bed", "desk", "cabinet", "chair", "lamp"]
    rels5 = [("bed", "behind", "desk"), ("cabinet", "left of", "bed"), ("chair", "left of", "desk"), ("lamp", "on", "desk")]
    objs, triples, attrs = S.scene_graph_from_words(objs5, rels5, device="cuda")
    n = 20000
    model.manual_seed(11)
    bp, _, z = S.sample_layouts(model, objs, triples, attrs, n_samples=n, mean=mean, cov=cov)
    zf = z.reshape(-1, E).double().cpu().numpy()
    sd_mean = np.sqrt(np.diag(cov.numpy()) / zf.shape[0])
    assert np.all(np.abs(zf.mean(0) - mean.numpy()) <= 6 * sd_mean), "mean of the device draws"
    emp = np.cov(zf.T)
    scale = np.sqrt(np.outer(np.diag(cov.numpy()), np.diag(cov.numpy())))
    assert np.abs(emp - cov.numpy()).max() <= 8 * scale.max() / np.sqrt(zf.shape[0]), "covariance of the device draws"
    # the histogram kernel against the torch accumulation on the same boxes
    counts = S.layout_counts(bp, 40)
    ref = S.layout_heatmap(bp, 40)
    assert_close((counts / counts.sum((1, 2), keepdim=True).clamp(min=1.0)).cpu().numpy(), ref.cpu().numpy(), "histogram kernel", rtol=1e-6, atol=1e-7)
    assert float(counts.sum()) == n * 5
    # distribution: device-drawn histograms against host-drawn ones
    h_dev = S.heatmap_from_words(model, objs5, rels5, mean, cov, num_iter=n, container_size=40).cpu().numpy()
    h_a = S.heatmap_from_words(model, objs5, rels5, mean, cov, num_iter=n, container_size=40, generator=torch.Generator().manual_seed(1)).cpu().numpy()
    h_b = S.heatmap_from_words(model, objs5, rels5, mean, cov, num_iter=n, container_size=40, generator=torch.Generator().manual_seed(2)).cpu().numpy()
    for o in range(5):
        tv_noise = 0.5 * np.abs(h_a[o] - h_b[o]).sum()
        tv = 0.5 * np.abs(h_dev[o] - h_a[o]).sum()
        assert tv <= 1.5 * tv_noise + 0.01, "object %d: TV(device, host) %.4f vs TV(host, host) %.4f" % (o, tv, tv_noise)
    # the stream advances; re-seeding replays it
    b1, _, z1 = S.sample_layouts(model, objs, triples, attrs, n_samples=64, mean=mean, cov=cov)
    b2, _, z2 = S.sample_layouts(model, objs, triples, attrs, n_samples=64, mean=mean, cov=cov)
    assert not torch.equal(z1, z2)
    model.manual_seed(11)
    _, _, z3 = S.sample_layouts(model, objs, triples, attrs, n_samples=n, mean=mean, cov=cov)
    assert torch.equal(z3, z)


def test_large_batch_equals_per_graph_evaluation():
    """256 graphs (O = 8192, T = 16384: the 128x64 / 128x128 GEMM tiles, long CSR lists) in eval mode: graphs never share rows
    (suncg_collate_fn offsets), so every graph of the batch must come out as it does alone - a size-independent check of the
    batched kernels at 4x the BASELINE batch."""
    cfg = vae_ref.VaeConfig()
    sd = vae_ref.init_state(cfg, seed=42)
    n_obj, n_tri, B = 32, 64, 256
    batch = vae_ref.synth_batch(B, n_obj, n_tri, seed=3, cfg=cfg)
    eps = torch.from_numpy(np.random.default_rng(4).standard_normal((B * n_obj, cfg.embedding_dim)).astype(np.float32))
    model = _model(cfg, sd).eval()
    with torch.no_grad():
        full = [t.cpu().numpy() for t in model(*_dev(*batch[:5]), None, eps=eps.cuda())]
        for g in (0, 101, 255):
            o0, o1, t0, t1 = g * n_obj, (g + 1) * n_obj, g * n_tri, (g + 1) * n_tri
            tr = batch[1][t0:t1].clone(); tr[:, 0] -= o0; tr[:, 2] -= o0
            one = model(*_dev(batch[0][o0:o1], tr, batch[2][o0:o1], batch[3][o0:o1], batch[4][o0:o1]), None, eps=eps[o0:o1].cuda())
            for a, b, nm in zip(full, one, ("mu", "logvar", "boxes_pred", "angles_pred")):
                assert_close(a[o0:o1], b.cpu().numpy(), "graph %d %s" % (g, nm), rtol=2e-5, atol=2e-5)
    # and against the oracle on one of them
    g = 101
    o0, o1, t0, t1 = g * n_obj, (g + 1) * n_obj, g * n_tri, (g + 1) * n_tri
    tr = batch[1][t0:t1].clone(); tr[:, 0] -= o0; tr[:, 2] -= o0
    with torch.no_grad():
        ref = vae_ref.forward(sd, cfg, batch[0][o0:o1], tr, batch[2][o0:o1], batch[3][o0:o1], batch[4][o0:o1], eps[o0:o1], training=False)
    for a, r, nm in zip(full, ref, ("mu", "logvar", "boxes_pred", "angles_pred")):
        assert_close(a[o0:o1], r.numpy(), "oracle graph %d %s" % (g, nm), rtol=1e-4, atol=1e-5)


def test_graph_is_recaptured_when_the_batch_shape_changes():
    """Real rooms differ in size from batch to batch: a captured iteration is tied to one (O, T); alternating shapes must
    re-capture (also the two half-iteration graphs of the overlapped data-parallel step) and give what eager launches give."""
    cfg = vae_ref.VaeConfig(embedding_dim=32, gconv_num_layers=2)
    # (16, 20, 30) outgrows the bound workspace (the engine is re-created, Adam's step counter must survive); its repetition is a
    # NEW batch of the SAME shape: the replayed graph must read the new tensors, not the ones it was captured with
    shapes = [(8, 12, 20), (5, 9, 14), (8, 12, 20), (3, 30, 41), (16, 20, 30), (16, 20, 30), (16, 20, 30)]
    batches = []
    for i, (g, o, t) in enumerate(shapes):
        b = vae_ref.synth_batch(g, o, t, seed=20 + i, cfg=cfg)
        e = torch.randn(b[0].shape[0], cfg.embedding_dim, generator=torch.Generator().manual_seed(i))
        batches.append(_dev(*b[:5], e))
    outs = {}
    for mode in ("eager", "graph", "halves"):
        model = _model(cfg, vae_ref.init_state(cfg, seed=1)).train()
        s = torch.cuda.Stream()
        losses = []
        with torch.cuda.stream(s):
            for d in batches:
                if mode == "halves":
                    l = model.train_step_begin(*d[:5], kl_weight=0.1, lr=1e-3, eps=d[5], use_graph=True)
                    model.train_step_finish(use_graph=True)
                    model.adam_step(lr=1e-3)
                else:
                    l = model.train_step(*d[:5], kl_weight=0.1, lr=1e-3, eps=d[5], use_graph=(mode == "graph"))
                losses.append(l)
        torch.cuda.synchronize()
        outs[mode] = (torch.stack(losses).cpu().numpy(), model.flat_params.cpu().numpy().copy())
    for mode in ("graph", "halves"):
        # Adam turns the rounding noise of near-zero gradients into +-lr steps, so two runs of this sequence drift apart (seen:
        # 1e-3 of the loss at the 7th step); reading a stale batch, restarting Adam's bias correction or losing its constants
        # moves the later losses by 1.5 % and more (or to NaN)
        assert_close(outs[mode][0], outs["eager"][0], mode + ": losses", rtol=3e-3)
        d = np.abs(outs[mode][1] - outs["eager"][1])
        assert d.max() <= 2.05 * 1e-3 * len(shapes), (mode, d.max())


@pytest.mark.parametrize("norm", ["batch", "none"])
def test_eval_mode_gradients_match_the_oracle(norm):
    """train.py:63-65 switches the model to eval mode after --eval_mode_after iterations and KEEPS training: BatchNorm then is a
    fixed affine map (running statistics), its gamma / beta still receive gradients.  Autograd path, every parameter."""
    cfg = vae_ref.VaeConfig(embedding_dim=16, gconv_num_layers=2, mlp_normalization=norm)
    sd = vae_ref.init_state(cfg, seed=9)
    batch = vae_ref.synth_batch(6, 7, 11, seed=5, cfg=cfg)
    O = batch[0].shape[0]
    eps = torch.from_numpy(np.random.default_rng(2).standard_normal((O, cfg.embedding_dim)).astype(np.float32))
    keys = vae_ref.trainable_keys(cfg)
    sdr = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    for k in keys:
        sdr[k].requires_grad_(True)
    mu, lv, bp, ap = vae_ref.forward(sdr, cfg, batch[0], batch[1], batch[2].double(), batch[3], batch[4], eps.double(), training=False)
    total, _ = vae_ref.losses(cfg, batch[2].double(), bp, batch[3], ap, mu, lv, 0.1)
    total.backward()
    model = _model(cfg, sd).eval()
    dev = _dev(*batch[:5], eps)
    out = model(*dev[:5], None, eps=dev[5])
    U = pkg("host.utils")
    import types
    t2, _ = U.calculate_model_losses(types.SimpleNamespace(use_AE=cfg.use_AE), model, dev[2], out[2], dev[3], out[3], mu=out[0], logvar=out[1],
                                     KL_weight=0.1)
    model.zero_grad()
    t2.backward()
    assert_close(float(t2.detach()), float(total.detach()), "eval total", rtol=1e-5)
    gscale = max(float(sdr[k].grad.abs().max()) for k in keys if sdr[k].grad is not None)
    named = dict(model.named_parameters())
    for k in keys:
        ref = sdr[k].grad
        got = named[k].grad
        if ref is None:
            assert got is None or float(got.abs().max()) == 0.0, k
            continue
        assert_close(got.cpu().numpy(), ref.numpy(), "eval grad " + k, rtol=2e-4, atol=2e-6 * gscale)
    for k, v in model.state_dict().items():          # eval mode leaves the BatchNorm buffers alone
        if "running" in k or "num_batches" in k:
            assert torch.equal(v.cpu(), sd[k]), k


@pytest.mark.parametrize("use_graph", [False, True])
def test_fused_step_in_eval_mode_matches_the_oracle(use_graph):
    """The fused iteration after ``model.eval()`` (train.py:63-65): running statistics in forward and backward, buffers untouched,
    gradients of every parameter as the oracle's eval-mode autograd gives them; back in train mode the graph is re-captured."""
    cfg = vae_ref.VaeConfig(embedding_dim=16, gconv_num_layers=2, mlp_normalization="batch")
    sd = vae_ref.init_state(cfg, seed=9)
    batch = vae_ref.synth_batch(6, 7, 11, seed=5, cfg=cfg)
    O = batch[0].shape[0]
    eps = torch.from_numpy(np.random.default_rng(2).standard_normal((O, cfg.embedding_dim)).astype(np.float32))
    keys = vae_ref.trainable_keys(cfg)
    refs = {}
    for training in (False, True):
        sdr = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        for k in keys:
            sdr[k].requires_grad_(True)
        mu, lv, bp, ap = vae_ref.forward(sdr, cfg, batch[0], batch[1], batch[2].double(), batch[3], batch[4], eps.double(), training=training)
        total, _ = vae_ref.losses(cfg, batch[2].double(), bp, batch[3], ap, mu, lv, 0.1)
        total.backward()
        refs[training] = (float(total.detach()), {k: sdr[k].grad for k in keys})
    model = _model(cfg, sd)
    dev = _dev(*batch[:5], eps)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for training in (True, False, False, True):                      # mode switches in both directions, one replay in between
            model.train(training)
            before = {k: v.clone() for k, v in model.state_dict().items() if "running" in k}
            losses = model.train_step(*dev[:5], kl_weight=0.1, lr=1e-4, eps=dev[5], use_graph=use_graph, with_adam=False)
            torch.cuda.synchronize()
            total, grads = refs[training]
            assert_close(float(losses[3]), total, "total (training=%s)" % training, rtol=2e-5)
            gscale = max(float(g.abs().max()) for g in grads.values() if g is not None)
            named = dict(model.named_parameters())
            for k, g in grads.items():
                if g is not None:
                    assert_close(named[k].grad.cpu().numpy(), g.numpy(), "grad %s (training=%s)" % (k, training), rtol=3e-4, atol=3e-6 * gscale)
            changed = any(not torch.equal(before[k], v) for k, v in model.state_dict().items() if "running" in k)
            assert changed == training
            if training:                                                 # undo the running-stat update: the oracle starts from sd
                model.load_state_dict({k: v.clone() for k, v in sd.items()})


@pytest.mark.parametrize("use_graph", [False, True])
def test_non_finite_loss_skips_the_update_like_the_reference(use_graph):
    """train.py:79-81: 'WARNING: Got loss = NaN, not backpropping' - the iteration is skipped.  The fused step decides on the
    device: parameters, Adam moments and the step count stay as they were, and the next good batch trains normally."""
    cfg = vae_ref.VaeConfig(embedding_dim=16, gconv_num_layers=2)
    sd = vae_ref.init_state(cfg, seed=3)
    good = _dev(*vae_ref.synth_batch(4, 6, 9, seed=1, cfg=cfg)[:5])
    bad = [t.clone() for t in good]
    bad[2][3, 1] = float("nan")                                   # one box coordinate
    eps = torch.randn(good[0].shape[0], cfg.embedding_dim, generator=torch.Generator().manual_seed(0)).cuda()
    ref = _model(cfg, sd).train()
    model = _model(cfg, sd).train()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        l_ref = [ref.train_step(*good, kl_weight=0.1, lr=1e-3, eps=eps, use_graph=use_graph) for _ in range(2)]
        p0 = model.flat_params.clone()
        l_bad = model.train_step(*bad, kl_weight=0.1, lr=1e-3, eps=eps, use_graph=use_graph)
        p1 = model.flat_params.clone()
        l_good = [model.train_step(*good, kl_weight=0.1, lr=1e-3, eps=eps, use_graph=use_graph) for _ in range(2)]
    torch.cuda.synchronize()
    assert not np.isfinite(float(l_bad[3]))
    assert torch.equal(p0, p1), "a non-finite loss must not move the parameters"
    assert torch.isfinite(model.flat_params).all()
    # (BatchNorm's running statistics saw the NaN batch, as in the reference where the forward pass ran; train mode ignores them)
    a, b = float(l_good[0][3]), float(l_ref[0][3])
    assert abs(a - b) <= 1e-5 * abs(b), (a, b)                    # first good step: same parameters as an untouched model


def test_optimizer_state_is_interchangeable_with_torch_adam():
    """train.py:94 saves ``optimizer.state_dict()`` and :25 restores it.  The fused Adam exports / imports torch.optim.Adam's
    own layout: three fused steps, then the 4th step taken (a) fused, (b) by a real torch.optim.Adam that loaded the exported
    state, (c) fused by a fresh model that imported that torch optimizer's state - all three must agree."""
    cfg = vae_ref.VaeConfig(embedding_dim=16, gconv_num_layers=2)
    sd = vae_ref.init_state(cfg, seed=4)
    dev = _dev(*vae_ref.synth_batch(5, 6, 9, seed=2, cfg=cfg)[:5])
    eps = torch.randn(dev[0].shape[0], cfg.embedding_dim, generator=torch.Generator().manual_seed(0)).cuda()
    a = _model(cfg, sd).train()
    for _ in range(3):
        a.train_step(*dev, kl_weight=0.1, lr=1e-3, eps=eps, use_graph=False)
    osd = a.optim_state_dict(lr=1e-3)
    assert len(osd['state']) == len(list(a.parameters())) and float(osd['state'][0]['step']) == 3.0
    msd = {k: v.clone() for k, v in a.state_dict().items()}
    # (b) torch.optim.Adam on a copy of the model, gradients through the autograd path
    b = _model(cfg, msd).train()
    import copy
    opt = torch.optim.Adam(b.parameters(), lr=1e-3)
    opt.load_state_dict(copy.deepcopy(osd))        # torch keeps tensors of matching dtype / device by reference and steps them in place
    U = pkg("host.utils")
    import types
    out = b(*dev, None, eps=eps)
    total, _ = U.calculate_model_losses(types.SimpleNamespace(use_AE=cfg.use_AE), b, dev[2], out[2], dev[3], out[3], mu=out[0], logvar=out[1],
                                        KL_weight=0.1)
    opt.zero_grad(); total.backward(); opt.step(); b.params_changed()
    # (c) a fresh model that imports a torch optimizer's state (the one of (b) BEFORE its step = the exported one, via torch)
    c = _model(cfg, msd).train()
    opt_c = torch.optim.Adam(c.parameters(), lr=1e-3)
    opt_c.load_state_dict(copy.deepcopy(osd))
    c.load_optim_state_dict(opt_c.state_dict())
    c.train_step(*dev, kl_weight=0.1, lr=1e-3, eps=eps, use_graph=False)
    a.train_step(*dev, kl_weight=0.1, lr=1e-3, eps=eps, use_graph=False)
    pa, pb, pc = (m.flat_params.detach().cpu().numpy() for m in (a, b, c))
    p0 = np.concatenate([np.pad(msd[k].cpu().numpy().reshape(-1), (0, (-msd[k].numel()) % 64)) for k, _ in a.named_parameters()])
    assert np.abs(pa - p0).max() > 1e-4, "the 4th step moved nothing"
    for other, nm in ((pb, "torch.optim.Adam with the exported state"), (pc, "fused Adam with the imported state")):
        d = np.abs(other - pa)
        # +-lr noise on parameters whose true gradient is 0 (biases in front of BatchNorm): bound it, require the bulk to agree
        assert d.max() <= 2.05e-3 and np.mean(d > 2e-6) < 0.03, (nm, d.max(), np.mean(d > 2e-6))
    assert float(c.optim_state_dict()['state'][0]['step']) == 4.0


def test_validation_forward_between_training_steps_leaves_training_untouched():
    """A validation pass (model.eval(); model(val_batch) on a same-shaped batch; model.train()) between two fused steps: the
    second step's replayed graph must see its own batch and train-mode BatchNorm again, and give what an uninterrupted run gives."""
    cfg = vae_ref.VaeConfig(embedding_dim=32, gconv_num_layers=2)
    sd = vae_ref.init_state(cfg, seed=6)
    tr = [_dev(*vae_ref.synth_batch(6, 8, 13, seed=s, cfg=cfg)[:5]) for s in (1, 2)]
    val = _dev(*vae_ref.synth_batch(6, 8, 13, seed=9, cfg=cfg)[:5])
    eps = torch.randn(tr[0][0].shape[0], cfg.embedding_dim, generator=torch.Generator().manual_seed(0)).cuda()
    res = {}
    for interrupted in (False, True):
        m = _model(cfg, sd).train()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            l0 = m.train_step(*tr[0], kl_weight=0.1, lr=1e-3, eps=eps, use_graph=True)
            if interrupted:
                m.eval()
                with torch.no_grad():
                    v = m(*val, None, eps=eps)
                assert all(torch.isfinite(t).all() for t in v)
                m.train()
            l1 = m.train_step(*tr[1], kl_weight=0.1, lr=1e-3, eps=eps, use_graph=True)
        torch.cuda.synchronize()
        res[interrupted] = (l0.cpu().numpy(), l1.cpu().numpy(), m.flat_params.cpu().numpy().copy(),
                            {k: v.cpu().numpy() for k, v in m.state_dict().items() if "running" in k})
    assert_close(res[True][0], res[False][0], "first step", rtol=1e-6)
    assert_close(res[True][1], res[False][1], "second step after the validation pass", rtol=1e-5)
    d = np.abs(res[True][2] - res[False][2])
    assert d.max() <= 2.05e-3 * 2 and np.mean(d > 1e-5) < 0.02
    for k in res[True][3]:
        assert_close(res[True][3][k], res[False][3][k], k, rtol=3e-3, atol=1e-6)     # second-step statistics see the +-lr Adam noise of the first (up to 3e-4 seen)


@pytest.mark.parametrize("env", [{"SLN_NO_GROUP": "1"}, {"SLN_NO_DUAL": "1"}, {"SLN_NO_DUAL": "1", "SLN_NO_SIDE_STREAM": "1"},
                                 {"SLN_NO_DEFER": "1"}, {"SLN_NO_DEFER": "1", "SLN_NO_GROUP": "1"}, {"SLN_NO_DEFER": "1", "SLN_NO_DUAL": "1"},
                                 {"SLN_NO_DEFER": "1", "SLN_NO_DUAL": "1", "SLN_NO_SIDE_STREAM": "1"}, {"SLN_NO_MERGE": "1"},
                                 {"SLN_TN_SIDE": "1"}, {"SLN_TN_PER_LAYER": "1"}, {"SLN_TN_SIDE": "1", "SLN_TN_PER_LAYER": "1"}])
def test_unmerged_launch_paths_give_the_same_step(env):
    """The default step runs every wgrad of a pass in one launch and merges its bookkeeping launches (round 3); before that the dgrad
    and the wgrad of a Linear shared a grid and the twin branches were grouped (SLN_NO_DEFER=1), with fall-backs to separate launches
    - wgrads on a side stream, or everything on one stream.  Every switch that selects another launch structure for a whole engine
    must reproduce the default step (loss, every gradient, BatchNorm buffers)."""
    import os
    cfg = vae_ref.VaeConfig(embedding_dim=32, gconv_num_layers=3)
    sd = vae_ref.init_state(cfg, seed=8)
    dev = _dev(*vae_ref.synth_batch(12, 10, 17, seed=3, cfg=cfg)[:5])
    eps = torch.randn(dev[0].shape[0], cfg.embedding_dim, generator=torch.Generator().manual_seed(0)).cuda()

    def run():
        m = _model(cfg, sd).train()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            l = m.train_step(*dev, kl_weight=0.1, lr=1e-3, eps=eps, use_graph=False, with_adam=False)
        torch.cuda.synchronize()
        return l.cpu().numpy(), m.flat_grads.cpu().numpy().copy(), {k: v.cpu().numpy() for k, v in m.state_dict().items() if "running" in k}
    base = run()
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)                                   # read by sln_vae_create
    try:
        alt = run()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    # not bit-comparable: the unmerged paths run the under-filled GEMMs on the 32 x 32 split-K body, the merged ones on the
    # 64 x 64 body (other fp32 summation order; a ReLU mask near 0 may flip) - a wrong operand or a missed launch is O(1)
    assert_close(alt[0], base[0], "losses", rtol=1e-5)
    assert_close(alt[1], base[1], "gradients", rtol=2e-5, atol=1e-3 * float(np.abs(base[1]).max()))
    for k in base[2]:
        assert_close(alt[2][k], base[2][k], k, rtol=1e-5)


def test_graph_replay_sees_new_kl_weight_and_learning_rate():
    """--KL_linear_decay changes the KL weight while training (train.py:73-76), a scheduler may change the learning rate: both
    live in device scalars, not in captured kernel arguments, so a replayed graph must use the values of the current call."""
    cfg = vae_ref.VaeConfig(embedding_dim=32, gconv_num_layers=2)
    sd = vae_ref.init_state(cfg, seed=2)
    dev = _dev(*vae_ref.synth_batch(6, 8, 12, seed=4, cfg=cfg)[:5])
    eps = torch.randn(dev[0].shape[0], cfg.embedding_dim, generator=torch.Generator().manual_seed(0)).cuda()
    sched = [(0.1, 1e-3), (0.1, 1e-3), (1e-4, 1e-3), (1.0, 3e-4), (0.1, 1e-3)]
    res = {}
    for use_graph in (False, True):
        m = _model(cfg, sd).train()
        s = torch.cuda.Stream(); ls = []; moves = []
        with torch.cuda.stream(s):
            for w, lr in sched:
                p0 = m.flat_params.clone()
                ls.append(m.train_step(*dev, kl_weight=w, lr=lr, eps=eps, use_graph=use_graph))
                moves.append((m.flat_params - p0).abs().median())
        torch.cuda.synchronize()
        res[use_graph] = (torch.stack(ls).cpu().numpy(), torch.stack(moves).cpu().numpy())
    assert_close(res[True][0], res[False][0], "losses under a changing KL weight", rtol=2e-3)
    kld = res[True][0][:, 2]
    assert kld[2] < 0.01 * kld[1] and kld[3] > 5 * kld[1]            # the weighted KL term follows the weight of its own call
    assert_close(res[True][1], res[False][1], "step sizes under a changing learning rate", rtol=5e-2)
    assert res[True][1][3] < 0.6 * res[True][1][2]                    # lr 3e-4 after 1e-3
