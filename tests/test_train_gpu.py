"""GPU tests of the training-loop plumbing around the fused iteration: the data-parallel step over RCCL (world of one on the
1-GPU box: the collective path, the guard element, the two-half overlap), the on-device N(0,1) draw, the optimizer-state
bookkeeping and the reference-shaped loop with an unmodified torch optimizer (run with -m gpu on an MI355X)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from conftest import pkg
from parity import assert_close

pytestmark = pytest.mark.gpu

from oracle import vae_ref                                   # noqa: E402


def _model(cfg, sd):
    M = pkg("host.Sg2ScVAE_model")
    m = M.Sg2ScVAEModel(**cfg.model_kwargs())
    m.load_state_dict({k: v.clone() for k, v in sd.items()})
    return m.cuda()


def _dev(*ts):
    return [t.cuda() for t in ts]


@pytest.fixture(scope="module")
def rccl_world_of_one():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1)
    yield
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
@pytest.mark.parametrize("use_graph", [False, True])
def test_data_parallel_step_over_rccl_equals_the_plain_step(rccl_world_of_one, overlap, use_graph):
    """DataParallelStep(force=True): backward, all-reduce(AVG) of [gradients | guard] through RCCL, fused Adam - with a world of
    one the result must be the plain fused step's (the collective is an identity), in both the one-bucket and the two-half
    (decoder half reduced under the encoder's backward) forms, eager and replayed."""
    T = pkg("host.train")
    cfg = vae_ref.VaeConfig(embedding_dim=32, gconv_num_layers=2)
    sd = vae_ref.init_state(cfg, seed=7)
    batches = [_dev(*vae_ref.synth_batch(8, 12, 20, seed=s, cfg=cfg)[:5]) for s in (3, 4, 5)]
    eps = torch.randn(batches[0][0].shape[0], cfg.embedding_dim, generator=torch.Generator().manual_seed(0)).cuda()
    res = []
    for dp in (False, True):
        model = _model(cfg, sd).train()
        step = T.DataParallelStep(model, 1, overlap=overlap, force=dp)
        assert step.dp == dp and step.overlap == (overlap and dp) and step.guarded
        st = torch.cuda.Stream()
        ls = []
        with torch.cuda.stream(st):
            for b in batches:
                ls.append(step(dict(objs=b[0], triples=b[1], boxes=b[2], angles=b[3], attributes=b[4]), 0.1, 1e-3, use_graph=use_graph, eps=eps).clone())
        torch.cuda.synchronize()
        res.append((np.stack([l.cpu().numpy() for l in ls]), model.flat_params.cpu().numpy().copy(), model._sync_adam_steps(),
                    float(model.grad_bucket[-1].cpu())))
    assert res[0][2] == res[1][2] == 3
    # steps 1-2 tightly; the third loss follows the random signs Adam's first steps give to parameters whose gradient is
    # rounding noise (tools/replay_stress.py: 3e-4 apart between two runs of the SAME path)
    assert_close(res[1][0][:2], res[0][0][:2], "losses of steps 1-2, dp vs plain", rtol=2e-6)
    assert_close(res[1][0][2], res[0][0][2], "losses of step 3, dp vs plain", rtol=5e-4)
    assert_close(res[1][3], res[1][0][2][3], "guard element = the rank's total loss", rtol=1e-6)
    d = np.abs(res[1][1] - res[0][1])
    # +-lr noise on parameters whose true gradient is 0 (atomic order differs between runs): bounded; the bulk agrees
    assert d.max() <= 2.05e-3 * 3 and np.mean(d > 1e-5) < 0.02, (d.max(), np.mean(d > 1e-5))


def test_non_finite_loss_skips_the_update_and_the_step_count(rccl_world_of_one):
    """train.py:79-81 ('not backpropping'): a NaN batch leaves parameters, moments and the step counter alone - on the fused
    step and on the data-parallel one (guard element), and the optimizer state exported afterwards carries the device's count."""
    T = pkg("host.train")
    cfg = vae_ref.VaeConfig(embedding_dim=16, gconv_num_layers=1)
    sd = vae_ref.init_state(cfg, seed=1)
    good = _dev(*vae_ref.synth_batch(4, 6, 9, seed=2, cfg=cfg)[:5])
    bad = [t.clone() for t in good]; bad[2][0, 0] = float("nan")
    eps = torch.zeros(good[0].shape[0], cfg.embedding_dim, device="cuda")
    for dp in (False, True):
        model = _model(cfg, sd).train()
        step = T.DataParallelStep(model, 1, force=dp)
        mk = lambda b: dict(objs=b[0], triples=b[1], boxes=b[2], angles=b[3], attributes=b[4])
        step(mk(good), 0.1, 1e-3, use_graph=False, eps=eps)
        p1 = model.flat_params.clone()
        l = step(mk(bad), 0.1, 1e-3, use_graph=False, eps=eps)
        assert not np.isfinite(l.cpu().numpy()[3])
        assert bool((model.flat_params == p1).all()), "a non-finite iteration moved the parameters"
        step(mk(good), 0.1, 1e-3, use_graph=False, eps=eps)
        assert bool(torch.isfinite(model.flat_params).all()) and not bool((model.flat_params == p1).all())
        assert model._sync_adam_steps() == 2
        assert float(model.optim_state_dict(1e-3)['state'][0]['step']) == 2.0


def test_device_side_normal_draw():
    """train_step / forward without eps draw N(0,1) on the device (Sg2ScVAE_model.py:182): moments and tail of the draw, a new
    draw per iteration also under hipGraph replay, reproducible from the seed, and the iteration computed with the drawn eps is
    the one an injected copy of that eps gives."""
    cfg = vae_ref.VaeConfig(embedding_dim=64, gconv_num_layers=1)
    sd = vae_ref.init_state(cfg, seed=3)
    b = _dev(*vae_ref.synth_batch(64, 16, 24, seed=1, cfg=cfg)[:5])            # O = 1024 -> 65 536 normals per draw
    draws = {}
    for use_graph in (False, True):
        model = _model(cfg, sd).train()
        model.manual_seed(1234)
        st = torch.cuda.Stream()
        seq = []
        with torch.cuda.stream(st):
            for _ in range(3):
                model.train_step(*b, kl_weight=0.1, lr=1e-4, use_graph=use_graph)
                seq.append(model.last_eps().cpu().numpy().copy())
        torch.cuda.synchronize()
        draws[use_graph] = seq
    for k in range(3):
        assert (draws[False][k] == draws[True][k]).all(), "graph replay and eager launches draw different streams"
    e0, e1 = draws[True][0].ravel(), draws[True][1].ravel()
    assert not (e0 == e1).any() or np.mean(e0 == e1) < 1e-4, "the replayed graph re-used its first draw"
    allv = np.concatenate([d.ravel() for d in draws[True]]).astype(np.float64)
    n = allv.size
    assert abs(allv.mean()) < 5 / np.sqrt(n) and abs(allv.var() - 1) < 5 * np.sqrt(2.0 / n)
    assert abs(np.mean(allv ** 3)) < 5 * np.sqrt(15.0 / n) and abs(np.mean(allv ** 4) - 3) < 5 * np.sqrt(96.0 / n)
    assert abs(np.corrcoef(allv[:-1], allv[1:])[0, 1]) < 5 / np.sqrt(n)
    from scipy import stats
    assert stats.kstest(allv[:50000], "norm").pvalue > 1e-4
    assert np.abs(allv).max() > 3.5 and np.isfinite(allv).all()
    # the step computed from the drawn eps == the step with that eps injected
    a = _model(cfg, sd).train(); a.manual_seed(99)
    la = a.train_step(*b, kl_weight=0.1, lr=1e-4, use_graph=False).cpu().numpy()
    eps = a.last_eps()
    c = _model(cfg, sd).train()
    lc = c.train_step(*b, kl_weight=0.1, lr=1e-4, eps=eps, use_graph=False).cpu().numpy()
    assert_close(la, lc, "losses drawn vs injected", rtol=1e-6)
    # autograd path: forward() without eps keeps the drawn eps for backward
    d = _model(cfg, sd).train(); d.manual_seed(99)
    mu, lv, bp, ap = d(*b, None)
    (bp.sum() + ap.sum() + mu.sum()).backward()
    e = _model(cfg, sd).train()
    mu2, lv2, bp2, ap2 = e(*b, None, eps=eps)
    (bp2.sum() + ap2.sum() + mu2.sum()).backward()
    assert_close(bp.detach().cpu().numpy(), bp2.detach().cpu().numpy(), "boxes drawn vs injected", rtol=1e-6)
    gs = float(e.flat_grads.abs().max())
    assert_close(d.flat_grads.cpu().numpy(), e.flat_grads.cpu().numpy(), "grads drawn vs injected", rtol=1e-5, atol=1e-6 * gs)


def test_reference_loop_with_an_unmodified_torch_optimizer():
    """train.py:70-84 as written: model(...) -> calculate_model_losses -> zero_grad -> backward -> torch.optim.Adam.step(), with
    NO call into the engine between the steps: the dgrad GEMMs must not run on the previous step's transposed weights
    (two steps against the oracle doing the same with CPU autograd; a stale W^T shows up in the second step's gradients)."""
    import types
    U = pkg("host.utils")
    cfg = vae_ref.VaeConfig(embedding_dim=32, gconv_num_layers=2, mlp_normalization="none")
    sd = vae_ref.init_state(cfg, seed=11)
    batch = vae_ref.synth_batch(6, 9, 14, seed=2, cfg=cfg)
    eps = torch.randn(batch[0].shape[0], cfg.embedding_dim, generator=torch.Generator().manual_seed(5))
    model = _model(cfg, sd).train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)           # a big step: stale weights would be far off
    dev = _dev(*batch[:5], eps)
    ref = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    keys = vae_ref.trainable_keys(cfg)
    ropt = torch.optim.Adam([ref[k] for k in keys], lr=1e-2)
    for it in range(2):
        out = model(*dev[:5], None, eps=dev[5])
        total, _ = U.calculate_model_losses(types.SimpleNamespace(use_AE=False), model, dev[2], out[2], dev[3], out[3], mu=out[0],
                                            logvar=out[1], KL_weight=0.1)
        opt.zero_grad(); total.backward()
        mu, lv, bp, ap = vae_ref.forward(ref, cfg, *batch[:5], eps, True)
        rt, _ = vae_ref.losses(cfg, batch[2], bp, batch[3], ap, mu, lv, 0.1)
        ropt.zero_grad(); rt.backward()
        assert_close(total.item(), rt.item(), "loss step %d" % it, rtol=2e-4)
        named = dict(model.named_parameters())
        gscale = max(float(ref[k].grad.abs().max()) for k in keys if ref[k].grad is not None)
        for k in keys:
            if ref[k].grad is not None:
                # atol: one ReLU input within an atomics-order rounding of zero flips in ~1 run out of 10 and moves a layer's gradient
                # by 5e-4 of the largest one (the same discrete 1.503e-3 every time); stale transposed weights are 10-20 % off
                assert_close(named[k].grad.cpu().numpy(), ref[k].grad.numpy(), "step %d grad %s" % (it, k), rtol=2e-3, atol=2e-3 * gscale)
        opt.step(); ropt.step()


def test_fused_adam_is_torch_adam_in_the_reference_loop():
    """`optimizer = model.fused_adam(lr)` (the one-line swap for train.py:15) against torch.optim.Adam in the literal train.py:70-84
    loop: three iterations from the same state and inputs; parameters agree to fp32 rounding, the Adam state round-trips through
    torch's own state_dict layout, and a stale guard value (left by a data-parallel iteration) does not block the update."""
    import types
    U = pkg("host.utils")
    cfg = vae_ref.VaeConfig(embedding_dim=32, gconv_num_layers=2, mlp_normalization="none")
    sd = vae_ref.init_state(cfg, seed=11)
    batch = vae_ref.synth_batch(6, 9, 14, seed=2, cfg=cfg)
    eps = torch.randn(batch[0].shape[0], cfg.embedding_dim, generator=torch.Generator().manual_seed(5))
    dev = _dev(*batch[:5], eps)
    outs = {}
    for name in ("torch", "fused"):
        model = _model(cfg, sd).train()
        model.route_torch_adam = name != "torch"                 # torch's OWN update on the 230 tensors (not routed to the fused kernel)
        opt = torch.optim.Adam(model.parameters(), lr=1e-3) if name == "torch" else model.fused_adam(lr=1e-3)
        if name == "fused":
            model(*dev[:5], None, eps=dev[5])                    # engine exists; poison the guard slot as a NaN rank would
            model.grad_bucket[-1] = float("nan")
        for it in range(3):
            out = model(*dev[:5], None, eps=dev[5])
            total, _ = U.calculate_model_losses(types.SimpleNamespace(use_AE=False), model, dev[2], out[2], dev[3], out[3], mu=out[0],
                                                logvar=out[1], KL_weight=0.1)
            opt.zero_grad(); total.backward(); opt.step()
        outs[name] = (model.flat_params.clone(), opt.state_dict())
    pt, pf = outs["torch"][0].cpu().numpy(), outs["fused"][0].cpu().numpy()
    moved = np.abs(pt - vae_ref_flat(cfg, sd, _model(cfg, sd))).max()
    assert moved > 1e-3, "the loop did not train"
    # Adam turns the rounding noise of near-zero gradients into +-lr steps of either sign: bound those, require the bulk to agree
    d = np.abs(pt - pf)
    assert d.max() <= 3 * 2.05e-3 and np.mean(d > 1e-5) < 0.05, (d.max(), np.mean(d > 1e-5))
    st, sf = outs["torch"][1], outs["fused"][1]
    assert sorted(st["state"].keys()) == sorted(sf["state"].keys())
    for i in st["state"]:
        a, b2 = st["state"][i]["exp_avg"].cpu().numpy(), sf["state"][i]["exp_avg"].cpu().numpy()
        assert_close(b2, a, "exp_avg %d" % i, rtol=1e-3, atol=1e-5 * max(float(np.abs(a).max()), 1e-30) + 1e-9)
        assert int(float(sf["state"][i]["step"])) == 3


def test_unmodified_torch_adam_is_routed_to_the_fused_kernel_and_keeps_torchs_state_layout():
    """train.py:15's ``torch.optim.Adam(model.parameters(), lr)`` with NO change: its step() runs the fused kernel (two optimizer
    hooks, host/Sg2ScVAE_model.py), optimizer.state holds views of the fused moments.  Against the same optimizer with the routing
    off (torch's multi-tensor update): parameters agree to Adam's rounding noise, state_dict() has torch's layout and values, a
    state_dict loaded into a fresh optimizer continues identically, and a step torch must take itself (one gradient missing)
    continues from the shared moments with the right step counters."""
    import types
    U = pkg("host.utils")
    cfg = vae_ref.VaeConfig(embedding_dim=32, gconv_num_layers=2, mlp_normalization="none")
    sd = vae_ref.init_state(cfg, seed=11)
    batch = vae_ref.synth_batch(6, 9, 14, seed=2, cfg=cfg)
    eps = torch.randn(batch[0].shape[0], cfg.embedding_dim, generator=torch.Generator().manual_seed(5))
    dev = _dev(*batch[:5], eps)

    def loop(model, opt, n, drop_grad_at=None):
        for it in range(n):
            out = model(*dev[:5], None, eps=dev[5])
            total, _ = U.calculate_model_losses(types.SimpleNamespace(use_AE=False), model, dev[2], out[2], dev[3], out[3], mu=out[0],
                                                logvar=out[1], KL_weight=0.1)
            opt.zero_grad(); total.backward()
            if drop_grad_at == it:
                model.box_net[0].bias.grad = None                 # torch skips this parameter: the step cannot be routed
            opt.step()
    outs = {}
    for routed in (False, True):
        model = _model(cfg, sd).train()
        model.route_torch_adam = routed
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        calls = []
        if routed:
            orig = model.adam_step
            model.adam_step = lambda lr=1e-4: (calls.append(lr), orig(lr=lr))[1]
        loop(model, opt, 3)
        assert len(calls) == (3 if routed else 0)
        assert len(opt.param_groups[0]['params']) == len(list(model.parameters()))       # the list is back after every step
        outs[routed] = (model.flat_params.clone(), opt.state_dict(), model, opt)
    pt, pf = outs[False][0].cpu().numpy(), outs[True][0].cpu().numpy()
    d = np.abs(pt - pf)
    assert d.max() <= 3 * 2.05e-3 and np.mean(d > 1e-5) < 0.05, (d.max(), np.mean(d > 1e-5))
    st, sf = outs[False][1], outs[True][1]
    assert sorted(st["state"].keys()) == sorted(sf["state"].keys()) == list(range(len(st["state"])))
    assert sf["param_groups"][0]["params"] == st["param_groups"][0]["params"]
    for i in st["state"]:
        for k in ("exp_avg", "exp_avg_sq"):
            a, b2 = st["state"][i][k].cpu().numpy(), sf["state"][i][k].cpu().numpy()
            assert a.shape == b2.shape
            assert_close(b2, a, "%s %d" % (k, i), rtol=1e-3, atol=1e-5 * max(float(np.abs(a).max()), 1e-30) + 1e-12)
        assert int(float(sf["state"][i]["step"])) == int(float(st["state"][i]["step"])) == 3
    # train.py:25: the saved state into a FRESH optimizer of a fresh model, two more steps on either path
    import copy
    ends = {}
    for routed in (False, True):
        model = _model(cfg, sd).train()
        model.load_state_dict(outs[True][2].state_dict())
        model.route_torch_adam = routed
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        opt.load_state_dict(copy.deepcopy(sf))
        loop(model, opt, 2, drop_grad_at=1 if routed else None)   # routed arm: second step falls back to torch's own update
        ends[routed] = (model.flat_params.clone().cpu().numpy(), opt.state_dict())
    d = np.abs(ends[False][0] - ends[True][0])
    assert d.max() <= 2 * 2.05e-3 and np.mean(d > 1e-5) < 0.05, (d.max(), np.mean(d > 1e-5))
    steps = sorted({int(float(v["step"])) for v in ends[True][1]["state"].values()})
    assert steps == [4, 5], steps                                  # the parameter without a gradient was skipped once, as torch does


def vae_ref_flat(cfg, sd, model):
    return model.flat_params.detach().cpu().numpy()


def test_optimizer_state_survives_a_device_or_dtype_re_flatten():
    """model.float() / .cuda() re-flatten the parameters: the Adam moments and the step count move along (they used to be
    dropped silently: load_optim_state_dict followed by .cuda() lost the state)."""
    cfg = vae_ref.VaeConfig(embedding_dim=16, gconv_num_layers=1)
    sd = vae_ref.init_state(cfg, seed=1)
    b = _dev(*vae_ref.synth_batch(4, 6, 9, seed=2, cfg=cfg)[:5])
    eps = torch.zeros(b[0].shape[0], cfg.embedding_dim, device="cuda")
    a = _model(cfg, sd).train()
    for _ in range(2):
        a.train_step(*b, kl_weight=0.1, lr=1e-3, eps=eps, use_graph=False)
    m_before = a._adam_m.clone()
    a = a.float().cuda()
    assert a._sync_adam_steps() == 2 and bool((a._adam_m == m_before).all())
    a.train_step(*b, kl_weight=0.1, lr=1e-3, eps=eps, use_graph=False)
    c = _model(cfg, sd).train()
    for _ in range(3):
        c.train_step(*b, kl_weight=0.1, lr=1e-3, eps=eps, use_graph=False)
    d = np.abs(a.flat_params.cpu().numpy() - c.flat_params.cpu().numpy())
    assert d.max() <= 2.05e-3 * 3 and np.mean(d > 1e-5) < 0.02, (d.max(), np.mean(d > 1e-5))
    assert a._sync_adam_steps() == 3


@pytest.mark.parametrize("mode", ["feedforward", "recurrent"])
def test_deterministic_mode_makes_fused_steps_bit_identical(mode):
    """sln_set_deterministic(1) (or SLN_DETERMINISTIC=1): the reference's CPU path is run-to-run deterministic; with the switch on so
    is the fused training step - three steps (eager, then replayed as a hipGraph) from the same state, batch and eps give
    bit-identical losses, gradients and parameters, also with shared (recurrent) GraphTripleConv weights, whose layers add into
    the same dW.  The deterministic result agrees with the default path's within the usual tolerance."""
    lib = pkg("_lib")
    cfg = vae_ref.VaeConfig(embedding_dim=32, gconv_num_layers=3, gconv_mode=mode)
    sd = vae_ref.init_state(cfg, seed=13)
    b = _dev(*vae_ref.synth_batch(24, 9, 15, seed=4, cfg=cfg)[:5])
    eps = torch.randn(b[0].shape[0], cfg.embedding_dim, generator=torch.Generator().manual_seed(3)).cuda()

    def run(use_graph):
        model = _model(cfg, sd).train()
        st = torch.cuda.Stream()
        out = []
        with torch.cuda.stream(st):
            for _ in range(3):
                l = model.train_step(*b, kl_weight=0.1, lr=1e-3, eps=eps, use_graph=use_graph)
                out.append((l.clone(), model.flat_grads.clone()))
        torch.cuda.synchronize()
        return out, model.flat_params.clone()
    try:
        lib.check(lib.lib().sln_set_deterministic(1), "sln_set_deterministic")
        assert lib.lib().sln_get_deterministic() == 1
        runs = [run(False), run(False), run(True), run(True)]
    finally:
        lib.lib().sln_set_deterministic(0)
    ref_steps, ref_params = runs[0]
    for steps, params in runs[1:]:
        for (l0, g0), (l1, g1) in zip(ref_steps, steps):
            assert torch.equal(l0, l1), (l0, l1)
            assert torch.equal(g0, g1), float((g0 - g1).abs().max())
        assert torch.equal(ref_params, params)
    plain, _ = run(False)
    assert_close(plain[0][0].cpu().numpy(), ref_steps[0][0].cpu().numpy(), "first-step losses, default vs deterministic", rtol=1e-5)
    gs = float(ref_steps[0][1].abs().max())
    assert_close(plain[0][1].cpu().numpy(), ref_steps[0][1].cpu().numpy(), "first-step gradients, default vs deterministic", rtol=2e-3,
                 atol=2e-5 * gs)


@pytest.mark.gpu
def test_eager_steps_on_batches_of_changing_shape_need_no_host_sync():
    """train.py on real rooms: every batch has another (O, T).  The wgrad problem tables and the BatchNorm row-count table then
    change every step; they are uploaded in stream order (pinned ring, csrc/vae_engine.hip::stage_upload) while earlier steps
    are still in flight.  In deterministic mode (fixed-order sums: two runs of the same sequence give the same bits) the run that
    never drains the stream equals the run that drains it after every step BIT FOR BIT, for more steps than the ring has slots.
    (In the default mode the arrival order of fp32 atomics differs from run to run and 30 Adam steps on 40-row BatchNorms amplify
    that to percents - with or without the drain.)"""
    lib = pkg("_lib").lib()
    cfg = vae_ref.VaeConfig(embedding_dim=32, gconv_num_layers=2)
    sd = vae_ref.init_state(cfg, seed=5)
    sizes = [(9, 8, 12), (7, 9, 14), (12, 6, 9), (5, 8, 12), (11, 7, 13)]
    batches = [_dev(*vae_ref.synth_batch(g, o, t, seed=20 + i, cfg=cfg)[:5]) for i, (g, o, t) in enumerate(sizes)]
    n_steps = 30
    eps = [torch.from_numpy(np.random.default_rng(100 + k).standard_normal((batches[k % 5][0].shape[0], cfg.embedding_dim)).astype(np.float32)).cuda()
           for k in range(n_steps)]
    runs = {}
    try:
        lib.sln_set_deterministic(1)
        for drain in (True, False):
            model = _model(cfg, sd).train()
            model.validate_inputs = False
            st = torch.cuda.Stream()
            losses = []
            with torch.cuda.stream(st):
                for k in range(n_steps):
                    losses.append(model.train_step(*batches[k % 5], kl_weight=0.1, lr=1e-3, eps=eps[k], use_graph=False))
                    if drain:
                        torch.cuda.synchronize()
            torch.cuda.synchronize()
            runs[drain] = (torch.stack(losses).cpu().numpy(), model.flat_params.detach().cpu().numpy().copy())
    finally:
        lib.sln_set_deterministic(0)
    assert np.isfinite(runs[False][0]).all()
    assert (runs[False][0] == runs[True][0]).all(), "losses without / with a drain per step"
    assert (runs[False][1] == runs[True][1]).all(), "parameters without / with a drain per step"


def test_optimizer_step_before_any_forward_creates_the_engine():
    """An empty-shard rank of the data-parallel trainer may reach adam_step() before it has ever run a forward (short first batch,
    batch_size < world): the update only needs the flat buffers - it used to raise on that rank alone, behind the
    collectives, and the others hung in the next all-reduce."""
    cfg = vae_ref.VaeConfig(embedding_dim=16, gconv_num_layers=1)
    sd = vae_ref.init_state(cfg, seed=1)
    m = _model(cfg, sd).train()
    assert m._eng is None
    before = m.flat_params.clone()
    m.flat_grads.fill_(1e-3)
    m.adam_step(lr=1e-3)
    torch.cuda.synchronize()
    assert m._eng is not None
    moved = (m.flat_params - before).abs()
    assert float(moved.max()) <= 1.01e-3 and float(moved.max()) > 0.9e-3       # Adam's first step: lr * g / (|g| + eps)
    # ... and a normal step still works on the engine created that way
    b = _dev(*vae_ref.synth_batch(4, 6, 9, seed=2, cfg=cfg)[:5])
    losses = m.train_step(*b, kl_weight=0.1, lr=1e-3, eps=torch.zeros(b[0].shape[0], cfg.embedding_dim, device="cuda"), use_graph=False)
    assert torch.isfinite(losses).all()
