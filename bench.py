#!/usr/bin/env python
"""bench.py - headline benchmark of the 3D_SLN hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without WORLD_SIZE: re-launches itself, one rank per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]; per GPU, weak scaling = configs[4] at N=8):
  one "step" = one full training iteration of the scene-graph VAE (train.py:62-84: zero_grad, forward with train-mode
  BatchNorm incl. the N(0,1) draw of the reparameterisation, the three losses, backward, Adam) on a synthetic batch of 64 scene
  graphs x (32 objects, 64 triples) => O=2048 object rows, T=4096 triples, at train.py's default widths (embedding_dim=64:
  GraphTripleConv 128/256/128, 5+5 layers).  Inputs are resident in HBM before the timed region.  For N>1 every rank trains its
  own 64 graphs and the flat 15.5 MB fp32 gradient buffer (+ the guard element) is all-reduced (RCCL) between backward and Adam.

Prints ONE JSON line (rank 0).  `value` = graphs/s over all GPUs.  `roofline` describes the dominant kernel family of that step,
timed live with HIP events on the launch stream in a separate eager pass; `cpu_baseline` is the CPU oracle (a PyTorch-CPU port
proven equal to the reference on the golden fixtures) timed on this box's host cores on the same batch.  At N=1 the line also
carries the other BASELINE configs as sub-objects, each with its own `roofline` and `cpu_baseline`: `c1` (configs[0]),
`render` (configs[2]), `spade` (configs[3]), plus `vae_large_batch`, `graph_build`, `refine`.

Every leg first checks the HIP path against the oracle on the leg's own inputs (`parity` keys); a failed check aborts the run
before anything is printed.  The oracle is only ever the checker and the timed CPU baseline, never part of a GPU number.
"""
import argparse
import ctypes as C
import gc
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The CPU-oracle legs (parity checks, cpu_baseline) run OpenMP teams whose idle threads spin between parallel regions; unbounded, they
# starve the thread that issues the next leg's GPU launches (round 5: the render leg's wall-clock mean was 1.9x its median behind the
# parity check).  A bounded spin (set before torch / libgomp load) keeps both: the render mean equals its median, and the CPU baselines
# stay at their round-5 level - OMP_WAIT_POLICY=PASSIVE (sleep at once) cost the oracle 20 % (299 vs 375 graphs/s, tools/lab/omp_ab.sh).
os.environ.setdefault("GOMP_SPINCOUNT", "30000")

MFMA_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32 matrix peak
VALU_F32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32 vector peak (2 flop x 64 lanes x 4 SIMD x 256 CU x 2.4 GHz)
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FAMILIES = ["gemm_nt", "gemm_tn", "edge", "other", "raster_fwd", "raster_bwd", "conv", "gemm_dual"]
# rocprofv3 summaries, ONE PER LEG (tools/profile_round.sh r03): kernel-trace statistics of a run that executes that leg's
# workload only, joined with the HBM traffic of two PMC passes of the same command (profiles/README.md)
PROFILE_TAG = "r06"


def profile_csv(leg):
    return os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (PROFILE_TAG, leg))


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--graphs", type=int, default=64, help="scene graphs per GPU per step")
    ap.add_argument("--objs", type=int, default=32)
    ap.add_argument("--triples", type=int, default=64)
    ap.add_argument("--batch-ring", type=int, default=4, help="distinct pre-generated batches cycled through by the timed loop (1 = one reused batch)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--inject-eps", action="store_true", help="feed one constant eps instead of drawing N(0,1) inside the step")
    ap.add_argument("--no-cpu", action="store_true", help="skip every CPU baseline leg")
    ap.add_argument("--no-check", action="store_true", help="skip the in-run parity checks against the oracle")
    ap.add_argument("--cpu-steps", type=int, default=20)
    ap.add_argument("--prof-steps", type=int, default=5)
    ap.add_argument("--rooms", type=int, default=16, help="rooms per render batch (BASELINE configs[2])")
    ap.add_argument("--tris", type=int, default=2000)
    ap.add_argument("--render-iters", type=int, default=100)
    ap.add_argument("--render-warmup", type=int, default=20)
    ap.add_argument("--no-render", action="store_true")
    ap.add_argument("--spade-batch", type=int, default=32, help="images per SPADE call (BASELINE configs[3])")
    ap.add_argument("--spade-iters", type=int, default=20)
    ap.add_argument("--spade-warmup", type=int, default=5)
    ap.add_argument("--no-spade", action="store_true")
    ap.add_argument("--graph-batch", type=int, default=512, help="rooms per scene-graph builder call")
    ap.add_argument("--graph-iters", type=int, default=100)
    ap.add_argument("--no-graph-build", action="store_true")
    ap.add_argument("--refine-iters", type=int, default=60, help="iterations of the layout-refinement leg (one room)")
    ap.add_argument("--no-refine", action="store_true")
    ap.add_argument("--refine-rooms", type=int, default=16, help="rooms in flight of the batched refinement leg")
    ap.add_argument("--refine-fit-steps", type=int, default=400,
                    help="Adam steps that over-fit the VAE to the refinement legs' rooms first (0: refine from the random decoder, the empty room)")
    ap.add_argument("--refine-rooms-large", type=int, default=64, help="a second, larger batch of rooms in flight (0 = skip)")
    ap.add_argument("--no-sampling", action="store_true")
    ap.add_argument("--legs-only", action="store_true",
                    help="run ONLY the side legs that are not switched off (no VAE loop at all) and print their objects: the command of "
                         "tools/profile_round.sh's per-leg traces, whose kernel tables must not contain another leg's launches")
    ap.add_argument("--sampling-draws", type=int, default=20000, help="posterior draws of the heat-map leg (testing/test_heatmap.py:39: num_iter)")
    ap.add_argument("--large-batches", type=str, default="128,256,512,1024,4096", help="extra VAE points (graphs per step; 128 = options/options.py:34, 512 = configs[4]'s global batch on one GPU), '' = none")
    ap.add_argument("--no-colorize", action="store_true", help="skip the one-map / 50-z SPADE leg (per-leg profiles: batch-32 launches only)")
    ap.add_argument("--no-dropin", action="store_true", help="skip the unchanged-call-sequence legs (vae_dropin, render_33pass, spade_50x1)")
    ap.add_argument("--dropin-steps", type=int, default=40)
    return ap.parse_args()


class ParityError(SystemExit):
    pass


def rel_err(got, ref):
    import numpy as np
    got = np.asarray(got, np.float64); ref = np.asarray(ref, np.float64)
    return float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30))


def require(ok, what):
    if not ok:
        raise ParityError("bench.py: parity check failed, nothing reported: " + what)


def prof_read(lib):
    n = len(FAMILIES)
    ms = (C.c_double * n)(); work = (C.c_double * n)(); cnt = (C.c_int64 * n)()
    lib.check(lib.lib().sln_prof_read(ms, work, cnt, n), "sln_prof_read")
    return {FAMILIES[i]: dict(ms=ms[i], work=work[i], launches=int(cnt[i])) for i in range(n) if cnt[i]}


PROFILE_STATUS = {}        # leg -> "ok" | "missing: <reason>" (why a `traffic` entry is null; printed next to it, never silent)


def profile_rows(prefixes, leg):
    """Rows of the committed rocprofv3 summary of `leg` ("vae" / "render" / "spade": kernel-trace stats joined with the PMC traffic
    passes) whose kernel name starts with one of `prefixes`: -> (launch-weighted HBM bytes per launch or None, launch-weighted
    avg us or None).  A missing file / row / traffic column is recorded in PROFILE_STATUS[leg] and logged."""
    import csv
    path = profile_csv(leg)
    why = None
    tot_b, tot_us, n_b, n_us = 0.0, 0.0, 0, 0
    try:
        rows = list(csv.DictReader(open(path)))
    except OSError as ex:
        rows, why = [], "missing: %s not readable (%s)" % (os.path.relpath(path, ROOT), ex.__class__.__name__)
    for r in rows:
        if not r["kernel"].startswith(tuple(prefixes)):
            continue
        calls = int(r["calls"])
        tot_us += float(r["avg_us"]) * calls; n_us += calls
        if r.get("hbm_MB_per_launch_corrected"):
            tot_b += float(r["hbm_MB_per_launch_corrected"]) * 1e6 * calls; n_b += calls
    if why is None and not n_us:
        why = "missing: no row of %s starts with %s" % (os.path.relpath(path, ROOT), "/".join(prefixes))
    elif why is None and not n_b:
        why = ("missing: %s has no hbm_MB_per_launch_corrected value for %s (the FETCH_SIZE / WRITE_SIZE passes of "
               "tools/profile_round.sh gave no rows)" % (os.path.relpath(path, ROOT), "/".join(prefixes)))
    if why is not None:
        log("profile_rows(%s): %s" % (leg, why))
        PROFILE_STATUS.setdefault(leg, why)
    elif not PROFILE_STATUS.get(leg, "").startswith("missing"):
        PROFILE_STATUS[leg] = "ok"
    return (round(tot_b / n_b) if n_b else None), (round(tot_us / n_us, 2) if n_us else None)


def profile_step_traffic(leg, per_step_kernel="adam_kernel"):
    """Counter HBM bytes of ONE step of `leg`'s profiled command: sum over every kernel row of bytes per launch x launches, divided by
    the number of steps the trace saw (= the calls of a kernel that runs once per step).  -> (bytes per step, steps) or (None, None)"""
    import csv
    try:
        rows = list(csv.DictReader(open(profile_csv(leg))))
    except OSError:
        return None, None
    steps = sum(int(r["calls"]) for r in rows if r["kernel"].startswith(per_step_kernel))
    have = [r for r in rows if r.get("hbm_MB_per_launch_corrected")]
    if not steps or not have:
        return None, None
    return sum(float(r["hbm_MB_per_launch_corrected"]) * 1e6 * int(r["calls"]) for r in have) / steps, steps


def traffic_source(leg, traffic):
    return (os.path.relpath(profile_csv(leg), ROOT) + " (bytes per launch)") if traffic else PROFILE_STATUS.get(leg, "missing: not looked up")


def cpu_threads(torch=None):
    """Host cores this process may actually use: the scheduler affinity mask, capped by the cgroup CPU quota (a container that
    sees 256 CPUs but owns 16 of them makes a 256-thread OpenMP team spin against itself)."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0]); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return max(1, n)


_T0 = time.perf_counter()


def log(msg):
    """progress on stderr (stdout carries the one JSON line)"""
    sys.stderr.write("[bench %7.1fs] %s\n" % (time.perf_counter() - _T0, msg))
    sys.stderr.flush()


# ------------------------------------------------------------------------------------------------------------- checks
def leg(name):
    """A leg starts with the cyclic collector run by hand: main() switches the automatic one off, because a generation-2 pass over the
    objects the CPU oracles leave behind takes ~50 ms on the host, and when it lands inside a 20-step timed loop the GPU idles for that
    long behind it (round 6: `--steps 20` showed the render leg's mean at 2x its median - one 53 ms iteration out of 100)."""
    log(name + " leg")
    gc.collect()


def check_vae(lib, torch, M, model, batch):
    """Smoke-style check of the fused step on a small configuration (1e-4 on the loss and on a gradient tensor) and a gross-error
    guard on the bench's own model and batch (forward + loss of the 64-graph batch against the oracle)."""
    import numpy as np
    from oracle import vae_ref
    cfg = vae_ref.VaeConfig(embedding_dim=32, gconv_num_layers=2)
    sd = vae_ref.init_state(cfg, seed=3)
    b = vae_ref.synth_batch(16, 8, 12, seed=5, cfg=cfg)
    eps = torch.from_numpy(np.random.default_rng(2).standard_normal((b[0].shape[0], cfg.embedding_dim)).astype(np.float32))
    small = M.Sg2ScVAEModel(**cfg.model_kwargs())
    small.load_state_dict({k: v.clone() for k, v in sd.items()})
    small = small.cuda().train()
    losses = small.train_step(*[t.cuda() for t in b[:5]], kl_weight=0.1, lr=1e-4, eps=eps.cuda(), use_graph=False, with_adam=False)
    keys = vae_ref.trainable_keys(cfg)
    m = {k: torch.zeros_like(sd[k]) for k in keys}; v = {k: torch.zeros_like(sd[k]) for k in keys}
    total, _, grads = vae_ref.train_step({k: t.clone() for k, t in sd.items()}, cfg, b[:5], eps, 0.1, m, v, step=1)
    e_loss = rel_err(losses[3].cpu().numpy(), float(total))
    k = "gconv_net_dc.gconvs.0.net1.0.weight"
    e_grad = rel_err(dict(small.named_parameters())[k].grad.cpu().numpy(), grads[k].numpy())
    require(e_loss <= 1e-4 and e_grad <= 1e-4, "VAE small step: loss %.2e, grad %.2e" % (e_loss, e_grad))
    # the bench model on the bench batch: same parameters through the oracle (state_dict keys are the reference's)
    sdb = {k: t.detach().cpu().clone() for k, t in model.state_dict().items()}
    cfgb = vae_ref.VaeConfig()
    cb = [t.cpu() for t in batch]
    epsb = torch.randn(cb[0].shape[0], cfgb.embedding_dim, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        model.train()
        mu, lv, bp, ap = model(*batch, None, eps=epsb.cuda())
        lg = model.loss(kl_weight=0.1).cpu().numpy()
        rmu, rlv, rbp, rap = vae_ref.forward({k: t.clone() for k, t in sdb.items()}, cfgb, *cb, epsb, True)
        rt, _ = vae_ref.losses(cfgb, cb[2], rbp, cb[3], rap, rmu, rlv, 0.1)
        # ... and in fp64: north_star's 1e-4 is held against this one; the fp32 oracle's own distance from it is printed beside it
        sd64 = {k: (t.double() if t.is_floating_point() else t.clone()) for k, t in sdb.items()}
        cb64 = [cb[0], cb[1], cb[2].double(), cb[3], cb[4]]
        dmu, dlv, dbp, dap = vae_ref.forward(sd64, cfgb, *cb64, epsb.double(), True)
        dt_, _ = vae_ref.losses(cfgb, cb64[2], dbp, cb64[3], dap, dmu, dlv, 0.1)
    model.load_state_dict(sdb)                       # the forward above moved the BatchNorm running statistics: restore
    e_full = rel_err(lg[3], float(rt))
    e_64 = rel_err(lg[3], float(dt_))
    e_ref = rel_err(float(rt), float(dt_))
    e_bp = rel_err(bp.cpu().numpy(), dbp.numpy())
    require(e_64 <= 1e-4 and e_bp <= 1e-4 + 4 * rel_err(rbp.numpy(), dbp.numpy()),
            "VAE bench batch: total loss %.6f vs fp64 oracle %.6f (%.2e), boxes_pred %.2e" % (lg[3], float(dt_), e_64, e_bp))
    return {"small_step_loss_rel_err": e_loss, "small_step_grad_rel_err": e_grad, "bench_batch_loss_rel_err": e_full,
            "bench_batch_loss_rel_err_vs_fp64_oracle": e_64, "fp32_oracle_vs_fp64_oracle": e_ref, "bench_batch_boxes_pred_rel_err_vs_fp64": e_bp}


def check_render(torch, pk, rooms_idx):
    """face-index / weight / depth maps of some rooms of the bench batch: bit-identical to the CPU restatement on identical faces"""
    from oracle import raster_ref as rr
    NR = importlib.import_module("3d_sln_amd.host.neural_renderer")
    with torch.no_grad():
        fxyz = NR.project_faces(pk["V"], pk["F"], pk["K"], pk["R"], pk["t"], 512)[rooms_idx].contiguous()
        fi, w, d = NR._rasterize(fxyz, 256, 0.001, 100.0)
    rfi, rw, rd = rr.nmr_forward(fxyz.cpu().numpy(), 256, 0.001, 100.0)
    diff = int((fi.cpu().numpy() != rfi).sum())
    same = bool((w.cpu().numpy() == rw).all() and (d.cpu().numpy() == rd).all())
    require(diff == 0 and same, "render: face index map differs at %d pixels, weights/depth identical: %s" % (diff, same))
    return {"rooms_checked": list(rooms_idx), "face_index_pixels_differing": diff, "weights_depth_bit_identical": same}


def check_spade(torch, S):
    """the reduced generator (ngf 8, 64 x 64) against the oracle: 1e-4 on the image"""
    from oracle import spade_ref
    cfg = spade_ref.SpadeConfig(nz=16, ngf=8, crop_size=64)
    sd = spade_ref.init_state(cfg, seed=7)
    G = S.SPADEGenerator4(cfg.semantic_nc, cfg.target_nc, cfg.nz, cfg.ngf, 'spectralspadelayer3x3', cfg.crop_size, 'normal')
    G.load_state_dict(sd); G = G.cuda().eval()
    seg, z = spade_ref.synth_input(cfg, 2, seed=3)
    with torch.no_grad():
        out = G(seg.cuda(), z.cuda()).cpu().numpy()
        ref = spade_ref.generator(sd, cfg, seg, z).numpy()
    e = rel_err(out, ref)
    require(e <= 1e-4, "SPADE small generator: image rel err %.2e" % e)
    return {"small_generator_image_rel_err": e}


def check_refine(torch):
    """The device refinement loop (RefineBatch: 2 rooms in flight, every kernel of the loop, 4 iterations at 96 x 96) against the
    REFERENCE'S OWN k loop of finetune_VAE, executed from its source text by oracle/gen_golden_refine.py (tests/golden/refine_loop.npz;
    the rasterizer under it is the restated one): per iteration the loss, the boxes, the soft-argmax angles and z at 1e-4."""
    import numpy as np
    from oracle import refine_ref, vae_ref
    R = importlib.import_module("3d_sln_amd.host.refine")
    M = importlib.import_module("3d_sln_amd.host.Sg2ScVAE_model")
    g = np.load(os.path.join(ROOT, "tests", "golden", "refine_loop.npz"))
    t = refine_ref.load_tables(g)
    bank = R.MeshBank.from_arrays({k: (m["v"], m["f"], m["bbox_min"], m["bbox_max"]) for k, m in t["models"].items()}, "cuda",
                                  vocab=t["vocab"], shell=t["shell"])
    cfg = vae_ref.VaeConfig(embedding_dim=32, gconv_num_layers=2, num_objs=len(t["vocab"]) + 1)
    model = M.Sg2ScVAEModel(**cfg.model_kwargs())
    model.load_state_dict({k[6:]: torch.from_numpy(g[k]).clone() for k in g.files if k.startswith("state:")})
    model = model.cuda().eval()
    names = ["__room__"] + t["vocab"]
    rooms = [dict(objs=torch.from_numpy(g["room%d:objs" % r]).cuda(), triples=torch.from_numpy(g["room%d:triples" % r]).cuda(),
                  boxes=torch.from_numpy(g["room%d:in_boxes" % r]).cuda(), angles=torch.from_numpy(g["room%d:in_angles" % r]).cuda(),
                  attributes=torch.from_numpy(g["room%d:attributes" % r]).cuda(), class_names=[names[int(o)] for o in g["room%d:objs" % r]])
             for r in (0, 1)]
    it = int(g["room0:noise"].shape[0])
    rb = R.RefineBatch(model, rooms, bank=bank, image_size=96, iters=it)
    errs = {"loss": 0.0, "boxes": 0.0, "angle_idx": 0.0, "z": 0.0, "z0_from_encoder": 0.0}
    try:
        for i in (0, 1):
            a, n = rb.row0[i], rb.rows[i]
            errs["z0_from_encoder"] = max(errs["z0_from_encoder"], rel_err(rb.z[a:a + n].cpu().numpy(), g["room%d:z0" % i]))
            rb.z[a:a + n] = torch.from_numpy(g["room%d:z0" % i]).cuda()
        for k in range(it):
            rb.run(1)
            for i in (0, 1):
                a, n = rb.row0[i], rb.rows[i]
                p = "room%d:" % i
                errs["loss"] = max(errs["loss"], rel_err(float(rb.losses[k, i]), g[p + "loss"][k]))
                errs["boxes"] = max(errs["boxes"], rel_err(rb.boxes[a:a + n].cpu().numpy(), g[p + "boxes"][k]))
                errs["angle_idx"] = max(errs["angle_idx"], rel_err(rb.idx[a:a + n].cpu().numpy(), g[p + "idx"][k]))
                errs["z"] = max(errs["z"], rel_err(rb.z[a:a + n].cpu().numpy(), g[p + "z"][k]))
    finally:
        rb.close()
    require(max(errs["loss"], errs["boxes"], errs["angle_idx"], errs["z"]) <= 1e-4 and errs["z0_from_encoder"] <= 1e-4,
            "refinement loop vs the reference's loop: " + ", ".join("%s %.2e" % kv for kv in errs.items()))
    errs["against"] = "the reference's own finetune_VAE k loop run from its source (tests/golden/refine_loop.npz), 2 rooms x %d iterations, 96 x 96" % it
    return errs


def vae_dropin_leg(args, torch, M, syn, fused_ms):
    """The literal train.py:70-84 iteration on the aliased classes - what a user gets WITHOUT touching train.py:
    model(...) -> utils.calculate_model_losses (three .item() syncs, utils.py:139-146) -> losses['total_loss'] = total_loss.item()
    -> isfinite test -> optimizer.zero_grad() -> total_loss.backward() -> optimizer.step(), eager launches, a new batch per step.
    Three optimizers: torch.optim.Adam as train.py:15 builds it; the documented one-line swap `optimizer = model.fused_adam(lr)`
    (same update, one kernel over the flat parameter buffer); and, as the reference point, the fused step of the headline."""
    import math
    U = importlib.import_module("3d_sln_amd.host.utils")

    class A:                                            # what calculate_model_losses reads from `args`
        use_AE = False
    ring = [syn.scene_graph_batch(args.graphs, args.objs, args.triples, seed=5000 + 7919 * k, device="cuda") for k in range(4)]
    res = {"workload": "train.py:70-84 unchanged on the aliased model: %d graphs x (%d objects, %d triples), eager, a new batch every step"
                       % (args.graphs, args.objs, args.triples), "fused_step_ms": round(fused_ms, 4)}
    for name in ("torch_adam", "fused_adam"):
        torch.manual_seed(42)
        model = M.Sg2ScVAEModel(vocab=syn.default_vocab(), batch_size=args.graphs, train_3d=True, decoder_cat=True, embedding_dim=64,
                                gconv_mode='feedforward', gconv_num_layers=5, mlp_normalization='batch', vec_noise_dim=0,
                                layout_noise_dim=32, use_AE=False).cuda().train()
        model.validate_inputs = False
        optimizer = torch.optim.Adam(model.parameters(), lr=1e-4) if name == "torch_adam" else model.fused_adam(lr=1e-4)
        st = torch.cuda.Stream()
        skipped = 0

        def one(t):
            b = ring[t % len(ring)]
            mu, logvar, boxes_pred, angles_pred = model(b["objs"], b["triples"], b["boxes"], b["angles"], b["attributes"], None)
            total_loss, losses = U.calculate_model_losses(A, model, b["boxes"], boxes_pred, b["angles"], angles_pred, mu=mu, logvar=logvar,
                                                          KL_weight=0.1)
            losses['total_loss'] = total_loss.item()
            if not math.isfinite(losses['total_loss']):
                return None
            optimizer.zero_grad()
            total_loss.backward()
            optimizer.step()
            return losses
        with torch.cuda.stream(st):
            for t in range(5):
                one(t)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in range(args.dropin_steps):
                l = one(t)
                skipped += l is None
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / args.dropin_steps
        res[name] = {"ms_per_step": round(dt * 1e3, 4), "graphs_per_s": round(args.graphs / dt, 1), "ratio_to_fused_step": round(dt * 1e3 / fused_ms, 2),
                     "steps": args.dropin_steps, "skipped_non_finite": int(skipped), "final_total_loss": None if l is None else round(l['total_loss'], 5)}
        del model, optimizer
        torch.cuda.empty_cache()
    return res


def render_33pass_leg(args, torch, fused_ms_per_room):
    """mesh_render_func's own call pattern (diff_render.py:366,381-398): ONE room, 1 depth pass + 32 class passes through the
    aliased nr.Renderer, forward + backward - what an unchanged diff_render.py gets.  The Renderer shares the projection node and
    the maps of identical geometry between the passes and back-propagates all rgb passes in one edge walk (host/neural_renderer.py);
    `rasterising_each_pass` switches that off.  `callers_torch_algebra_only_ms` is the same loop with a stub Renderer: the floor
    the reference's own per-class torch code sets, whatever the Renderer does."""
    DR = importlib.import_module("3d_sln_amd.host.diff_render")
    NR = importlib.import_module("3d_sln_amd.host.neural_renderer")
    syn = importlib.import_module("3d_sln_amd.host.synthetic")
    V, F, ranges, box = syn.synthetic_room(1, n_objects=12, target_faces=args.tris)
    f = torch.from_numpy(F)[None].cuda(); room = torch.from_numpy(box).cuda()
    res = {"workload": "one room, %d triangles, 256x256: 1 depth + 32 class passes through nr.Renderer, forward + backward" % int(F.shape[0]),
           "fused_scene_pass_ms_per_room_in_batch_of_%d" % args.rooms: round(fused_ms_per_room, 4)}
    keep = NR.Renderer.reuse_rasterisation
    for key, fn, reuse in (("maps_reused", DR.scene_render_passes, True), ("rasterising_each_pass", DR.scene_render_passes, False),
                           ("fused_one_room", DR.scene_render, True)):
        NR.Renderer.reuse_rasterisation = reuse

        def run():
            v = torch.from_numpy(V)[None].cuda().requires_grad_(True)
            out = fn(v, f, ranges, room)
            out.sum().backward()
            return out
        for _ in range(3):
            run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            run()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        res[key] = {"ms_per_render": round(ms, 3), "renders_per_s": round(1e3 / ms, 1)}
    # What the 33 Renderer calls cost, and what the caller's own torch algebra around them costs (diff_render.py:381-431: per class a
    # texture tensor, boolean-mask indexing, mean / max / isnan with their host syncs): the same loop with the Renderer replaced by a
    # stub that hands back recorded images (attached to the vertices, so backward still walks the caller's graph).
    NR.Renderer.reuse_rasterisation = True
    recorded, real = {}, NR.Renderer.render

    def rec(self, vertices, faces, textures=None, mode=None, *a, **k):
        out = real(self, vertices, faces, textures, mode, *a, **k)
        recorded.setdefault(mode, []).append(out.detach())
        return out
    counters = {}

    def stub(self, vertices, faces, textures=None, mode=None, *a, **k):
        i = counters.get(mode, 0); counters[mode] = i + 1
        return recorded[mode][i % len(recorded[mode])] + 0.0 * vertices.sum()
    try:
        NR.Renderer.render = rec
        DR.scene_render_passes(torch.from_numpy(V)[None].cuda().requires_grad_(True), f, ranges, room)
        NR.Renderer.render = stub

        def run_stub():
            counters.clear()
            v = torch.from_numpy(V)[None].cuda().requires_grad_(True)
            DR.scene_render_passes(v, f, ranges, room).sum().backward()
        for _ in range(3):
            run_stub()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            run_stub()
        torch.cuda.synchronize()
        floor = (time.perf_counter() - t0) / 10 * 1e3
    finally:
        NR.Renderer.render = real
        NR.Renderer.reuse_rasterisation = keep
    res["callers_torch_algebra_only_ms"] = round(floor, 3)
    res["renderer_share_ms"] = round(res["maps_reused"]["ms_per_render"] - floor, 3)
    res["ratio_33pass_to_fused_one_room"] = round(res["maps_reused"]["ms_per_render"] / res["fused_one_room"]["ms_per_render"], 1)
    return res


def spade_50x1_leg(args, torch, G, seg, batched_img_per_s):
    """testing/test_SPADE_shade.py:77-79 unchanged: for each of 50 z vectors, colorization_model(total, z) at batch 1, on the
    same tensor `total` - from the second call the generator keeps the map's gamma|beta planes (host/SPADE_related.py).  Timed as
    the reference runs it (a NEW map, then its 50 calls); `without_kept_planes` is the same loop with the reuse switched off."""
    g = torch.Generator(device="cuda").manual_seed(1)
    zs = [torch.randn(1, 256, device="cuda", generator=g) for _ in range(50)]
    res = {"workload": "50 x colorization_model(total[1,41,256,256], z[1,256]), one call per z, a new map per room"}
    for key, reuse in (("kept_planes", True), ("without_kept_planes", False)):
        G.reuse_map_planes = reuse
        with torch.no_grad():
            for z in zs[:3]:
                G(seg[1:2].contiguous(), z)
            total = seg[:1].clone()                                  # a new room: a tensor the generator has not seen
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for z in zs:
                img = G(total, z)
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        res[key] = {"ms_per_room_of_50": round(dt * 1e3, 2), "images_per_s": round(50 / dt, 1),
                    "ratio_to_batch_path": round((50 / dt) / batched_img_per_s, 3), "finite": bool(torch.isfinite(img).all().item())}
    G.reuse_map_planes = True
    G.clear_map_cache()
    res["images_per_s"] = res["kept_planes"]["images_per_s"]
    return res


# --------------------------------------------------------------------------------------------------------------- legs
def c1_leg(args, torch, M):
    """BASELINE configs[0]: Sg2ScVAE forward + loss on ONE synthetic 8-object / 12-triple scene graph, CPU, 1 iteration after 3
    warm-ups, all cores.  The same graph through the HIP path is timed next to it."""
    from oracle import vae_ref
    cfg = vae_ref.VaeConfig()
    sd = vae_ref.init_state(cfg, seed=42)
    b = vae_ref.synth_batch(1, 8, 12, seed=0, cfg=cfg)
    eps = torch.randn(8, cfg.embedding_dim, generator=torch.Generator().manual_seed(0))
    n = cpu_threads(torch)
    torch.set_num_threads(n)

    def cpu_iter():
        with torch.no_grad():
            mu, lv, bp, ap = vae_ref.forward({k: v.clone() for k, v in sd.items()}, cfg, *b[:5], eps, True)
            return vae_ref.losses(cfg, b[2], bp, b[3], ap, mu, lv, 0.1)[0]
    for _ in range(3):
        cpu_iter()
    t0 = time.perf_counter(); tot = cpu_iter(); cdt = time.perf_counter() - t0
    res = {"workload": "BASELINE configs[0]: 1 graph, 8 objects / 12 triples, forward + loss, train-mode BatchNorm, train.py defaults",
           "cpu_ms_per_iter": round(cdt * 1e3, 3), "cpu_graphs_per_s": round(1.0 / cdt, 2), "cores": n, "kind": "port",
           "sample": "1 iteration after 3 warm-ups (SURVEY.md 8d protocol), oracle/vae_ref.py, torch CPU fp32", "cpu_total_loss": round(float(tot), 6)}
    model = M.Sg2ScVAEModel(**cfg.model_kwargs())
    model.load_state_dict({k: v.clone() for k, v in sd.items()})
    model = model.cuda().train()
    dev = [t.cuda() for t in b[:5]]; epsd = eps.cuda()
    sd_dev = {k: v.clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        model(*dev, None, eps=epsd); l0 = model.loss(kl_weight=0.1)      # first call: same BatchNorm buffers as the CPU iteration
        res["hip_total_loss"] = round(float(l0[3].item()), 6)
        for _ in range(3):
            model(*dev, None, eps=epsd); l = model.loss(kl_weight=0.1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            model(*dev, None, eps=epsd); l = model.loss(kl_weight=0.1)
        torch.cuda.synchronize()
        gdt = (time.perf_counter() - t0) / 20
    del sd_dev
    res["hip_ms_per_iter"] = round(gdt * 1e3, 3)
    res["note"] = "8-row train-mode BatchNorm: any two fp32 evaluations differ by ~1e-3 (tests/parity.py); latency-bound on the GPU (one graph)"
    return res


def large_batch_leg(args, torch, M, syn, sizes):
    """The same fused step at larger per-GPU batches, where kernels fill the chip ('report larger-B points')."""
    out = {}
    for B in sizes:
        torch.manual_seed(42)
        model = M.Sg2ScVAEModel(vocab=syn.default_vocab(), batch_size=B, train_3d=True, decoder_cat=True, embedding_dim=64,
                                gconv_mode='feedforward', gconv_num_layers=5, mlp_normalization='batch', vec_noise_dim=0,
                                layout_noise_dim=32, use_AE=False).cuda().train()
        model.validate_inputs = False
        b = syn.scene_graph_batch(B, args.objs, args.triples, seed=77, device="cuda")
        batch = (b["objs"], b["triples"], b["boxes"], b["angles"], b["attributes"])
        st = torch.cuda.Stream()
        n_w, n_t = 3, (20 if B <= 1024 else 8)
        with torch.cuda.stream(st):
            for _ in range(n_w):
                l = model.train_step(*batch, kl_weight=0.1, lr=1e-4, use_graph=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_t):
                l = model.train_step(*batch, kl_weight=0.1, lr=1e-4, use_graph=True)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n_t
        flop = 1.24e9 * B                              # 1.24 GFLOP per graph (fwd + bwd)
        out[str(B)] = {"graphs_per_s": round(B / dt, 1), "ms_per_step": round(dt * 1e3, 3), "tflops_whole_step": round(flop / dt / 1e12, 2),
                       "frac_mfma_whole_step": round(flop / dt / 1e12 / MFMA_F32_PEAK_TFLOPS, 4), "finite": bool(torch.isfinite(l).all().item())}
        del model
        torch.cuda.empty_cache()
    return out


def render_leg(args, lib, torch, rank):
    """BASELINE configs[2]: differentiable render of `rooms` synthetic rooms x ~`tris` triangles at 256x256,
    forward (70-channel scene tensor of mesh_render_func) + backward to the vertices, fused HIP pass."""
    DR = importlib.import_module("3d_sln_amd.host.diff_render")
    syn = importlib.import_module("3d_sln_amd.host.synthetic")
    dev = "cuda"
    rooms = [syn.synthetic_room(100 + rank * 64 + i, n_objects=12, target_faces=args.tris) for i in range(args.rooms)]
    pk = syn.pack_rooms(rooms, dev)
    res = {}
    if not args.no_check:
        res["parity"] = check_render(torch, pk, sorted({0, args.rooms - 1}))
    Vb = pk["V"].requires_grad_(True)
    Fb, Cb, Kb, Rb, tb, chan_t, dch_t, tri_count = pk["F"], pk["C"], pk["K"], pk["R"], pk["t"], pk["chan"], pk["dch"], pk["tris"]
    gout = torch.randn(args.rooms, 70, 256, 256, device=dev)

    def it():
        Vb.grad = None
        out = DR.scene_render_batch(Vb, Fb, Cb, chan_t, dch_t, Kb, Rb, tb, 256, 0.001)
        out.backward(gout)
        return out
    for _ in range(args.render_warmup):
        out = it()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.render_iters + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for k in range(args.render_iters):
        it()
        marks[k + 1].record()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per_raw = [marks[k].elapsed_time(marks[k + 1]) for k in range(args.render_iters)]
    per = sorted(per_raw)
    if os.environ.get("SLN_BENCH_DEBUG"):
        worst = sorted(range(len(per_raw)), key=lambda k: -per_raw[k])[:6]
        log("render debug: slowest iterations (index, ms): %s; host enqueue of the loop %.1f ms for %.1f ms of GPU time" %
            ([(k, round(per_raw[k], 3)) for k in worst], t_enq * 1e3, sum(per_raw)))
    lib.check(lib.lib().sln_prof_enable(1), "prof")
    for _ in range(5):
        it()
    torch.cuda.synchronize()
    fam = prof_read(lib)
    lib.check(lib.lib().sln_prof_enable(0), "prof")
    # renders_per_s = rooms / WALL-CLOCK MEAN batch time of the timed loop (as rounds 1-4 reported it; round 5 printed the median
    # because the mean was 1.9x the median behind the parity check - the oracle's OpenMP threads were still spinning; they sleep now,
    # see OMP_WAIT_POLICY at the top).  The median (HIP events around every iteration) is printed next to it with their ratio.
    p50_ms = per[int(0.5 * (len(per) - 1))]
    mean_ms = dt / args.render_iters * 1e3
    per_render = mean_ms * 1e-3 / args.rooms
    tris = tri_count / args.rooms
    res.update({"renders_per_s": round(1.0 / per_render, 1), "renders_per_s_note": "rooms / wall-clock mean batch time of the timed loop",
                "renders_per_s_median": round(args.rooms / (p50_ms * 1e-3), 1), "mean_over_median": round(mean_ms / p50_ms, 3),
                "wall_mean_ms_per_batch": round(mean_ms, 4), "ms_per_room_fwd_bwd": round(per_render * 1e3, 4),
                "ms_per_batch_p10_p50_p90": [round(per[int(q * (len(per) - 1))], 4) for q in (0.1, 0.5, 0.9)],
                "warmup": args.render_warmup, "iters": args.render_iters,
                "nmr_equivalent_raster_passes_per_s": round(33.0 / per_render, 1),
                "workload": "BASELINE configs[2]: %d rooms x %d triangles (%.0f after near-plane cull, x2 fill_back), 256x256, "
                            "70-channel scene tensor fwd + bwd to vertices" % (args.rooms, args.tris, tris),
                "covered_pixels": round(float((out[:, 0] > 0).float().mean().item()), 3), "finite": bool(torch.isfinite(Vb.grad).all().item())})
    bytes_per_render = 2 * 70 * 256 * 256 * 4 + 2 * tris * 36 * 2     # out + grad + faces
    algo = (70 * 256 * 256 * 4 + tris * 72) * args.rooms
    for k, name in (("raster_fwd", "scene_forward"), ("raster_bwd", "scene_backward")):
        if k in fam:
            ms = fam[k]["ms"] / fam[k]["launches"]
            res[name] = {"avg_ms_per_batch": round(ms, 4), "gbs_algorithmic": round(algo / (ms * 1e-3) / 1e9, 1),
                         "frac_hbm": round(algo / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    res["algorithmic_bytes_per_render"] = int(bytes_per_render)
    # rooflines of the leg.  (i) SURVEY.md 8d's definition: the 70-channel output, its gradient and the faces over HBM.
    # (ii) the kernels that actually carry the time, named as in profiles/: the forward tile kernel is bound by per-pixel edge
    # tests on the vector ALUs (3 edge functions x 2 fma + sign tests per (pixel, face) pair: counted from the brute-force
    # 256^2 x 2F tests the package performs = the work replaced, and priced at the fp32 vector peak).
    whole_ms = mean_ms
    res["roofline"] = {"kernel": "scene_forward + scene_backward (all launches of one batch)", "bound": "hbm",
                       "achieved": round(bytes_per_render * args.rooms / (whole_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": round(bytes_per_render * args.rooms / (whole_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                       "traffic": None, "algorithmic_bytes_per_launch": int(bytes_per_render * args.rooms)}
    # counter bytes of one batch (every kernel of the leg-isolated profile, per batch = per call of the scan kernel)
    tr_b, tr_n = profile_step_traffic("render", "pixel_map_backward_kernel")
    if tr_b:
        res["roofline"].update(traffic=int(tr_b), traffic_over_algorithmic=round(tr_b / (bytes_per_render * args.rooms), 3),
                               traffic_source=os.path.relpath(profile_csv("render"), ROOT) + " (all kernels of one 16-room batch, %d batches traced)" % tr_n)
    tr_pm, us_pm = profile_rows(["pixel_map_backward", "class_scan_backward"], "render")
    tr_rt, us_rt = profile_rows(["raster_tile_kernel"], "render")
    brute_tests = 256.0 * 256.0 * 2.0 * tris * args.rooms          # (pixel, face) pairs of ONE brute-force pass over the batch
    flop_per_test = 3 * 2 * 2 + 3                                    # three edge functions (2 fma each) + sign tests
    # edge tests the tile kernel can at most execute: (front-facing face, 16x16 tile its pixel bounding box touches) pairs x 256
    # pixels (the tile-corner rejection drops some of these pairs - a triangle's bbox is twice its area - so this is an upper
    # bound of the executed work, from the projected faces of the batch)
    NRm = importlib.import_module("3d_sln_amd.host.neural_renderer")
    with torch.no_grad():
        fx = NRm.project_faces(Vb.detach(), Fb, Kb, Rb, tb, 512)
        px = 0.5 * (fx[..., :2] * 256 + 255)                                            # pixel coordinates of the corners
        front = ((px[:, :, 1, 0] - px[:, :, 0, 0]) * (px[:, :, 2, 1] - px[:, :, 0, 1]) -
                 (px[:, :, 1, 1] - px[:, :, 0, 1]) * (px[:, :, 2, 0] - px[:, :, 0, 0])) != 0
        lo = px.amin(2).floor().clamp(0, 255); hi = px.amax(2).ceil().clamp(0, 255)
        inside = (px.amax(2) >= 0).all(-1) & (px.amin(2) <= 255).all(-1) & (Cb >= 0)
        ntile = ((hi[..., 0] // 16 - lo[..., 0] // 16 + 1) * (hi[..., 1] // 16 - lo[..., 1] // 16 + 1))
        bbox_tests = float((ntile * (front & inside)).sum().item()) * 256.0 / 2.0          # fill_back: one of a face and its reverse
    res["roofline_kernels"] = {
        "raster_tile_kernel": {"bound": "valu", "unit": "TFLOP/s", "peak": VALU_F32_PEAK_TFLOPS, "avg_launch_us": us_rt,
                               "edge_tests_per_launch_upper_bound": bbox_tests, "flop_per_test": flop_per_test,
                               "achieved": round(bbox_tests * flop_per_test / (us_rt * 1e-6) / 1e12, 2) if us_rt else None,
                               "frac": round(bbox_tests * flop_per_test / (us_rt * 1e-6) / 1e12 / VALU_F32_PEAK_TFLOPS, 4) if us_rt else None,
                               "traffic": tr_rt,
                               "brute_force_tests_replaced_per_launch": brute_tests,
                               "brute_force_tests_per_executed_test": round(brute_tests / max(bbox_tests, 1.0), 1),
                               "note": "achieved = (face, tile) pairs the bounding boxes admit x 256 pixels x 15 flop / duration: an UPPER bound of "
                                       "the executed edge tests (tile-corner rejection drops part of them)"},
        "pixel_map_backward_kernel": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "avg_launch_us": us_pm, "traffic": tr_pm,
                                      "achieved": round(tr_pm / (us_pm * 1e-6) / 1e9, 1) if (tr_pm and us_pm) else None,
                                      "frac": round(tr_pm / (us_pm * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if (tr_pm and us_pm) else None,
                                      "note": "achieved = measured HBM traffic / duration (L2-resident scans: the kernel is latency / occupancy bound)"}}
    # the streaming passes of the scene algebra against the HBM roofline: algorithmic bytes of one launch / its duration in the
    # ONE-STREAM trace (profiles/<tag>_render_noside_kernel_stats.csv: with the side stream the durations of the two chains overlap)
    plane_b = 256.0 * 256.0 * 4.0
    live_planes = float((out[:, 1:41] > 0).flatten(2).any(2).sum().item())     # class planes with a visible pixel (the others are skipped)
    live_depth = float(((out[:, 41:] != 1.0) & (out[:, 41:] != 0.0)).flatten(2).any(2).sum().item())   # depth-hot planes that are not a constant
    stream_bytes = {"scene_compose_kernel": args.rooms * (70 * plane_b + 3 * plane_b),                       # 70 planes written, 3 maps read
                    "scene_bwd_plane_sums_kernel": live_depth * plane_b,                                     # the depth-hot gradient planes of visible classes, read
                    # one launch since round 5: the live class planes read once, g and g^T written; the three per-pixel maps + the
                    # own-class gradient read, two 16-byte records per pixel written
                    "scene_bwd_tables_kernel": live_planes * plane_b * 3 + args.rooms * (plane_b * 3 + 2 * 4 * plane_b)}
    for kname, nbytes in stream_bytes.items():
        _, us = profile_rows([kname], "render_noside")
        res["roofline_kernels"][kname] = {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "avg_launch_us": us,
                                          "algorithmic_bytes_per_launch": int(nbytes),
                                          "achieved": round(nbytes / (us * 1e-6) / 1e9, 1) if us else None,
                                          "frac": round(nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if us else None, "traffic": None}
    res["roofline_kernels"]["scene_bwd_tables_kernel"]["note"] = "%d live class planes in the batch (planes of classes without a visible pixel are skipped)" % int(live_planes)
    res["roofline_kernels"]["scene_bwd_plane_sums_kernel"]["note"] = "%d depth-hot planes of visible classes (the others are skipped since round 5)" % int(live_depth)
    if not args.no_dropin:
        res["render_33pass"] = render_33pass_leg(args, torch, per_render * 1e3)
    if not args.no_cpu:
        # CPU baseline: the 33-pass restatement (oracle/raster_ref.py + the OpenMP C++ rasterizer) forward + backward on rooms of
        # the same batch, all cores; bounded sample
        from oracle import raster_ref as rr
        n = cpu_threads(torch)
        torch.set_num_threads(n)
        try:                                   # the C++ restatement's own OpenMP team (same libgomp as torch's where both link it)
            C.CDLL("libgomp.so.1").omp_set_num_threads(n)
        except OSError:
            pass
        log('render CPU baseline on %d threads' % n)
        n_cpu, t_cpu = 0, 0.0
        gcpu = gout[:2].cpu()
        for V, F, ranges, box in rooms[:2]:
            v1 = torch.from_numpy(V)[None].requires_grad_(True)
            t0 = time.perf_counter()
            ref = rr.scene_render(v1, torch.from_numpy(F)[None], ranges, torch.from_numpy(box), image_size=256)
            (ref * gcpu[n_cpu:n_cpu + 1]).sum().backward()
            t_cpu += time.perf_counter() - t0
            n_cpu += 1
            if t_cpu > 20.0:
                break
        res["cpu_baseline"] = {"value": round(n_cpu / t_cpu, 3), "unit": "renders/s", "cores": n, "kind": "port",
                               "sample": "%d room(s) of the same batch, forward + backward through oracle/raster_ref.py (33 brute-force "
                                         "passes per render, OpenMP C++ rasterizer), %.2f s per render" % (n_cpu, t_cpu / n_cpu)}
    return res


def graph_build_leg(args, lib, torch):
    """SuncgDataset.__getitem__ + suncg_collate_fn for a batch of rooms on the device
    (csrc/graph_build.hip): 512 rooms x 31 objects (+ room row = the 32 objects/graph of configs[1]) per call,
    random decisions drawn on the device; the timed region includes the 3-int read-back that sizes the outputs."""
    D = importlib.import_module("3d_sln_amd.host.suncg_dataset")
    syn = importlib.import_module("3d_sln_amd.host.synthetic")
    rooms, names, sd, sd30 = syn.scene_rooms(2048, objs_per_room=args.objs - 1, seed=0)
    ds = D.SuncgDataset.from_tables(rooms, names, sd, sd30)
    B = args.graph_batch
    gen = torch.Generator(device="cuda").manual_seed(0)
    idx = torch.randint(0, len(rooms), (B,), generator=torch.Generator().manual_seed(0)).cuda()
    for _ in range(20):
        out = ds.build_batch(idx, generator=gen)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.graph_iters):
        out = ds.build_batch(idx, generator=gen)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.graph_iters
    O, T = int(out[1].shape[0]), int(out[3].shape[0])
    # the trainer's form: indices on the HOST (a sampler's permutation) - sizes come from per-room counts taken once, nothing is
    # read back per batch (round 3)
    idx_h = idx.cpu()
    for _ in range(20):
        ds.build_batch(idx_h, generator=gen)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.graph_iters):
        ds.build_batch(idx_h, generator=gen)
    torch.cuda.synchronize()
    dth = (time.perf_counter() - t0) / args.graph_iters
    nbytes = sum(int(t.numel()) * t.element_size() for t in out) + O * 6 * 4          # outputs + the raw boxes read
    res = {"roofline": {"kernel": "sln_graph_plan + sln_graph_draw + sln_graph_emit of one batch (host-indexed form)", "bound": "hbm", "unit": "GB/s",
                        "peak": HBM_PEAK_GBS, "algorithmic_bytes_per_launch": int(nbytes), "achieved": round(nbytes / dth / 1e9, 2),
                        "frac": round(nbytes / dth / 1e9 / HBM_PEAK_GBS, 5), "traffic": None,
                        "note": "a batch is ~1.5 MB of outputs: three launches at the launch floor, not a bandwidth problem"},
           "graphs_per_s": round(B / dt, 1), "us_per_batch": round(dt * 1e6, 1), "us_per_batch_host_indices": round(dth * 1e6, 1),
           "graphs_per_s_host_indices": round(B / dth, 1), "batch": B, "objects": O, "triples": T,
           "algorithmic_bytes_per_batch": nbytes,
           "workload": "%d rooms x %d objects: 'on' pairs over all ordered pairs, one drawn relation per object, in-room rows, "
                       "normalised boxes, size attributes, collate offsets" % (B, args.objs - 1)}
    if not args.no_cpu:
        from oracle import graph_build_ref as G                  # CPU baseline = the oracle (python restatement of the reference loop)
        import random as _r
        table = G.RoomTable(rooms, names, sd, sd30)
        _r.seed(0)
        n_cpu = 48
        t0 = time.perf_counter()
        batch = []
        for i in range(n_cpu):
            room = rooms[i]
            batch.append((i,) + G.build_room(room, table, G.draw_room(len(room["objs"]), room["objs"], table)))
        G.collate(batch)
        cdt = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": round(n_cpu / cdt, 1), "unit": "graphs/s", "cores": 1, "kind": "port",
                               "sample": "%d rooms through oracle/graph_build_ref.py (python loop per object pair, as the reference)" % n_cpu}
    return res


def refine_leg(args, lib, torch):
    """the inner loop of finetune_VAE (testing/test_render_refine.py:279-359) for ONE room of 12 objects on
    procedural meshes: decoder -> soft-argmax -> fused placement -> fused 70-channel render -> fused PSP / L1 / CE loss ->
    backward -> Nesterov SGD on z and the model copy, 60 iterations, eager launches, nothing synchronises inside the loop."""
    R = importlib.import_module("3d_sln_amd.host.refine")
    M = importlib.import_module("3d_sln_amd.host.Sg2ScVAE_model")
    syn = importlib.import_module("3d_sln_amd.host.synthetic")
    parity = None
    if not args.no_check:
        parity = check_refine(torch)
    names = ["bed", "chair", "table", "sofa", "desk", "cabinet", "lamp", "television", "bookshelf", "dresser", "night_stand", "shelves",
             "__room__"]
    n = len(names)
    g = torch.Generator().manual_seed(0)
    lo = torch.rand(n, 3, generator=g) * 0.45 + 0.05; lo[:, 1] = 0.0; lo[:, 2] *= 0.6
    hi = lo + torch.rand(n, 3, generator=g) * 0.2 + 0.12
    boxes = torch.cat([lo, hi], 1); boxes[-1] = torch.tensor([0, 0, 0, 4.0, 2.7, 5.0]); boxes = boxes.cuda()
    angles = torch.randint(0, 24, (n,), generator=g).cuda()
    torch.manual_seed(1)
    model = M.Sg2ScVAEModel(vocab=syn.default_vocab(), batch_size=1, train_3d=True, decoder_cat=True, embedding_dim=64,
                            gconv_mode='feedforward', gconv_num_layers=5, mlp_normalization='batch', vec_noise_dim=0,
                            layout_noise_dim=32, use_AE=False).cuda().train()
    objs = torch.arange(1, n + 1).cuda(); objs[-1] = 0
    triples = torch.tensor([[i, 1 + i % 10, (i + 1) % (n - 1)] for i in range(n - 1)] + [[i, 0, n - 1] for i in range(n - 1)]).cuda()
    attrs = torch.zeros(n, dtype=torch.int64).cuda()
    bank = R.MeshBank([x for x in names if x != "__room__"], "cuda", seed=3)
    st = torch.cuda.Stream()
    # the rooms of the R-rooms-in-flight legs (below), and the "checkpoint": the VAE over-fitted to all of them and to the one-room
    # leg's room, so that its decoder places the furniture near the targets and the iterate SHOWS it.  (Rounds 3-4 refined from a
    # randomly initialised decoder: near-degenerate boxes, the render was the empty room - 6 of the 70 scene planes non-constant
    # instead of ~22 - and the scan kernels had less to do than in the reference's use.)
    batch_rooms = []
    for r in range(max(args.refine_rooms, args.refine_rooms_large)):
        gr = torch.Generator().manual_seed(100 + r)
        lo_r = torch.rand(n, 3, generator=gr) * 0.45 + 0.05; lo_r[:, 1] = 0.0; lo_r[:, 2] *= 0.6
        hi_r = lo_r + torch.rand(n, 3, generator=gr) * 0.2 + 0.12
        bx = torch.cat([lo_r, hi_r], 1); bx[-1] = torch.tensor([0, 0, 0, 4.0, 2.7, 5.0])
        batch_rooms.append(dict(objs=objs, triples=triples, boxes=bx.cuda(), angles=torch.randint(0, 24, (n,), generator=gr).cuda(), attributes=attrs,
                                class_names=names))
    with torch.cuda.stream(st):
        fit = syn.overfit_to_rooms(model, batch_rooms + [dict(objs=objs, triples=triples, boxes=boxes, angles=angles, attributes=attrs)],
                                   steps=args.refine_fit_steps)
        fit = [round(float(x), 4) for x in fit] if fit is not None else None
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    iters = args.refine_iters
    def timed(n, capture=False):
        model.load_state_dict(sd0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        losses, _ = R.finetune_vae_fast(model, objs, triples, boxes, angles, attrs, names, iters=n, bank=bank, capture=capture)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, losses
    with torch.cuda.stream(st):
        R.finetune_vae_fast(model, objs, triples, boxes, angles, attrs, names, iters=3, bank=bank)
        # the per-room set-up (encoder, target render, the loss tables built on the HOST) is timed apart from the iterations: rooms of
        # `iters` and of 2 x `iters` iterations, five times each; an iteration costs the slope, the set-up the intercept.  (One timing
        # of set-up + iterations moved by +-8 % with the host's share.)
        t1s, t2s = [], []
        for _ in range(5):
            dt1, losses = timed(iters)
            dt2, _ = timed(2 * iters)
            t1s.append(dt1); t2s.append(dt2)
        slopes = sorted((b - a) / iters for a, b in zip(t1s, t2s))
        per_iter = slopes[2]
        setup = sorted(a - iters * per_iter for a in t1s)[2]
        dt = sorted(t1s)[2]
        slopes = [slopes[0], slopes[2], slopes[4]]
        # the same iteration captured once per room and replayed (capture=True): the launches of an iteration leave the host's hands
        g1s, g2s = [], []
        for _ in range(3):
            a_, _ = timed(iters, True)
            b_, _ = timed(2 * iters, True)
            g1s.append(a_); g2s.append(b_)
        gslopes = sorted((b_ - a_) / iters for a_, b_ in zip(g1s, g2s))
    # ---- R rooms in flight (round 5): testing/test_render_refine.py:250-263 runs its rooms one after the other; RefineBatch runs
    # `--refine-rooms` of them (each on its own copy of the parameters) as one launch sequence per iteration
    nr = args.refine_rooms
    rooms = batch_rooms
    model.load_state_dict(sd0)
    batch = {}
    all_rooms = rooms
    larger = None
    if args.refine_rooms_large > nr:
        # the same loop with more rooms in flight: the ~1 ms chain of dependent decoder launches is paid once per iteration whatever R
        with torch.cuda.stream(st):
            tl = []
            for n_it in (iters, 2 * iters, iters, 2 * iters):
                rb = R.RefineBatch(model, all_rooms[:args.refine_rooms_large], bank=bank, iters=n_it)
                torch.cuda.synchronize(); t1 = time.perf_counter()
                rb.run()
                torch.cuda.synchronize(); tl.append((n_it, time.perf_counter() - t1))
                finl = bool(torch.isfinite(rb.losses).all().item()); rb.close()
            a_ = min(x[1] for x in tl if x[0] == iters); b_ = min(x[1] for x in tl if x[0] == 2 * iters)
            larger = {"rooms": args.refine_rooms_large, "ms_per_iteration": round((b_ - a_) / iters * 1e3, 3),
                      "ms_per_room_iteration": round((b_ - a_) / iters * 1e3 / args.refine_rooms_large, 4), "finite": finl}
    rooms = all_rooms[:nr]
    with torch.cuda.stream(st):
        runs = []
        for n_it in (iters, 2 * iters, iters, 2 * iters, iters, 2 * iters):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            rb = R.RefineBatch(model, rooms, bank=bank, iters=n_it)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            rb.run()
            torch.cuda.synchronize(); t2 = time.perf_counter()
            runs.append((n_it, t1 - t0, t2 - t1))
            info, fin = rb.launches(), bool(torch.isfinite(rb.losses).all().item())
            lv = rb.live.cpu()
            n3, n1 = float((lv == 3).sum()) / nr, float((lv == 1).sum()) / nr     # planes per room: written and read / known constant
            if os.environ.get("SLN_BENCH_DEBUG"):
                berr = float((rb.boxes.view(nr, n, 6)[:, :-1] - torch.stack([r_["boxes"][:-1] for r_ in rooms])).abs().mean())
                log("refine debug: run of %d iterations, live planes %.1f, constant %.1f, |boxes - target| %.4f, first/last loss %.3f %.3f"
                    % (n_it, n3, n1, berr, float(rb.losses[0].mean()), float(rb.losses[n_it - 1].mean())))
            rb.close()
        a_ = sorted(x[2] for x in runs if x[0] == iters)[1]; b_ = sorted(x[2] for x in runs if x[0] == 2 * iters)[1]
        it_ms = (b_ - a_) / iters * 1e3
        plane = 256.0 * 256.0 * 4.0
        dec_bytes = 4.0 * sum(ln for _, ln in model.decoder_param_ranges())
        # algorithmic HBM bytes of one room-iteration: the LIVE planes of the 70-plane scene tensor (channel 0 and the planes of the
        # classes visible in the room: n3 of 70, measured on the last iteration) written, read by the pooling, their gradient
        # written and read by the scene backward; the pooled tensor (4 scales x (n3 - 1) live planes of 96 x 96; the n1 constant
        # planes share one pooled plane per scale) written, read, its gradient written, read; the decoder's parameters read by the
        # forward Linears, transposed (read + write), read by the dgrads, and stepped in the wgrads' epilogue (read + write)
        algo = 4 * n3 * plane + 4 * (4 * (n3 - 1) * 96 * 96 * 4.0) + 6 * dec_bytes
        batch = {"rooms": nr, "ms_per_iteration": round(it_ms, 3), "ms_per_room_iteration": round(it_ms / nr, 4),
                 "ms_setup_per_room": round(sorted(x[1] for x in runs)[len(runs) // 2] * 1e3 / nr, 2), "iterations": iters, "finite": fin,
                 "launches": info, "live_planes_per_room": round(n3, 1), "constant_planes_per_room": round(n1, 1), "speedup_vs_one_room_at_a_time": round(per_iter * 1e3 / (it_ms / nr), 2),
                 "roofline": {"kernel": "one refinement iteration of %d rooms (all launches)" % nr, "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                              "algorithmic_bytes_per_room_iteration": int(algo), "achieved": round(algo * nr / (it_ms * 1e-3) / 1e9, 1),
                              "frac": round(algo * nr / (it_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None}}
    cpu_base = None
    if not args.no_cpu:
        # CPU baseline: ONE refinement iteration of the same room through the oracle - decoder (oracle/vae_ref.py, autograd),
        # placement + PSP / L1 / CE loss (oracle/refine_ref.py), 33-pass renderer (oracle/raster_ref.py), all cores; bounded sample
        from oracle import vae_ref, refine_ref, raster_ref as rr
        ncores = cpu_threads(torch)
        torch.set_num_threads(ncores)
        cfg = vae_ref.VaeConfig()
        sdc = {k: (v.detach().cpu().clone().requires_grad_(v.is_floating_point() and "running" not in k)) for k, v in sd0.items()}
        cb, ca = boxes.cpu(), angles.cpu()
        cbank = R.MeshBank([x for x in names if x != "__room__"], "cpu", seed=3)
        v0, f0, ranges, sizes0, _ = R.assemble_scene(cb, ca.float(), names, cbank, cb[-1].clone())      # topology + shell vertices (constants)
        n_objv = sum(cbank.models[x]["v"].shape[0] for x in names[:-1] if x in cbank.models and x not in R.DO_NOT_VIS)
        shell_v = v0[0, n_objv:].detach()
        with torch.no_grad():
            tgt = rr.scene_render(v0.detach(), f0, ranges, cb[-1], image_size=256)
        lab = refine_ref.target_labels(tgt)
        zc = torch.randn(n, 64, generator=torch.Generator().manual_seed(13)).requires_grad_(True)
        co, ct, cat_ = objs.cpu(), triples.cpu(), attrs.cpu()
        n_cpu, t_cpu = 0, 0.0
        while n_cpu < 3 and t_cpu < 25.0:
            t0 = time.perf_counter()
            bp, ap = vae_ref.decoder(sdc, cfg, zc, co, ct, cat_, training=False)
            bp.register_hook(R.fix_grad)
            bfull = torch.cat([bp[:-1], cb[-1:]], 0)
            idx = R.softargmax(ap, sum_dim=1) + torch.randn(n) / 10.0
            idx.register_hook(R.quad_grad)
            idx = torch.cat([idx[:-1], ca[-1:].float()], 0)
            vo, _, sl = refine_ref.place_scene(bfull, idx, names, cbank.models, [s_.clone() for s_ in sizes0])
            img = rr.scene_render(torch.cat([vo, shell_v])[None], f0, ranges, cb[-1], image_size=256)
            loss_c, _, _ = refine_ref.refinement_loss(img, tgt, lab, sl)
            loss_c.backward()
            with torch.no_grad():
                zc -= 2e-4 * 1.1 * zc.grad; zc.grad = None
                for k_, v_ in sdc.items():
                    if v_.grad is not None:
                        v_ -= 1e-5 * 1.1 * v_.grad; v_.grad = None
            t_cpu += time.perf_counter() - t0; n_cpu += 1
        cpu_base = {"value": round(n_cpu / t_cpu, 3), "unit": "room-iterations/s", "cores": ncores, "kind": "port",
                    "sample": "%d iteration(s) of the same 12-object room through oracle/vae_ref.py + refine_ref.py + raster_ref.py (33 brute-force raster "
                              "passes per render), %.2f s per iteration" % (n_cpu, t_cpu / n_cpu)}
    one_room_bytes = 4 * 70 * 256.0 * 256.0 * 4.0 + 4 * (4 * 69 * 96 * 96 * 4.0) + 9 * 4.0 * sum(ln for _, ln in model.decoder_param_ranges())
    return {"parity": parity, "rooms_%d" % nr: batch, "rooms_%d" % args.refine_rooms_large: larger, "cpu_baseline": cpu_base,
            "model": "VAE over-fitted to the legs' rooms (%d fused Adam steps; last losses [bbox, angle, KL, total] = %s): its decoder places "
                     "the furniture in view, as a trained checkpoint does" % (args.refine_fit_steps, fit),
            "roofline": {"kernel": "one refinement iteration of ONE room (all launches; latency-bound: ~110 dependent launches)", "bound": "hbm", "unit": "GB/s",
                         "peak": HBM_PEAK_GBS, "algorithmic_bytes_per_room_iteration": int(one_room_bytes),
                         "achieved": round(one_room_bytes / (per_iter) / 1e9, 1), "frac": round(one_room_bytes / per_iter / 1e9 / HBM_PEAK_GBS, 4), "traffic": None},
            "ms_per_iteration": round(per_iter * 1e3, 3), "ms_per_iteration_min_median_max": [round(x * 1e3, 3) for x in slopes],
            "ms_per_iteration_hipgraph_replay_min_median_max": [round(x * 1e3, 3) for x in gslopes],
            "ms_setup_per_room_hipgraph": round(sorted(a_ - iters * gslopes[1] for a_ in g1s)[1] * 1e3, 2),
            "ms_setup_per_room": round(setup * 1e3, 2),
            "ms_per_iteration_incl_setup": round(dt / iters * 1e3, 3), "iterations": iters, "finite": bool(torch.isfinite(losses).all()),
            "includes": "ms_per_iteration: slope between rooms of 60 and 120 iterations (median of 5); the per-room set-up (encoder, target render, "
                        "loss tables built on the host) is ms_setup_per_room, ms_per_iteration_incl_setup amortises it over the iterations",
            "workload": "one room, 12 objects + shell (%d triangles x2 fill_back), 256x256, VAE at train.py defaults" %
                        int(R.RefineScene(names, bank, boxes[-1]).faces.shape[0])}


def sampling_leg(args, lib, torch):
    """posterior sampling / heat map of a worded scene (testing/test_heatmap.py:39-64: the default 5-object
    room, 20 000 draws of z ~ N(mean, cov) per object, one decoder call per draw in the reference).  Here: eps drawn on the device,
    z = mean + eps L^T as one GEMM, ONE decoder call over the 20 000 replicated graphs, one histogram launch."""
    S = importlib.import_module("3d_sln_amd.host.sampling")
    M = importlib.import_module("3d_sln_amd.host.Sg2ScVAE_model")
    syn = importlib.import_module("3d_sln_amd.host.synthetic")
    import numpy as np
    torch.manual_seed(3)
    vocab = syn.default_vocab()
    model = M.Sg2ScVAEModel(vocab=vocab, batch_size=1, train_3d=True, decoder_cat=True, embedding_dim=64, gconv_mode='feedforward',
                            gconv_num_layers=5, mlp_normalization='batch', vec_noise_dim=0, layout_noise_dim=32, use_AE=False).cuda().eval()
    model.validate_inputs = False
    E = 64
    rng = np.random.default_rng(0)
    A = rng.standard_normal((E, E)) * 0.2
    mean = torch.from_numpy(rng.standard_normal(E) * 0.1); cov = torch.from_numpy(A @ A.T + 0.05 * np.eye(E))
    objs5 = ["bed", "desk", "cabinet", "chair", "lamp"]
    rels5 = [("bed", "behind", "desk"), ("cabinet", "left of", "bed"), ("chair", "left of", "desk"), ("lamp", "on", "desk")]
    n = args.sampling_draws
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(2):
            h = S.heatmap_from_words(model, objs5, rels5, mean, cov, num_iter=n)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            h = S.heatmap_from_words(model, objs5, rels5, mean, cov, num_iter=n)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[2]
    O, T = 6, 9
    shp = vae_gemm_shapes(O, T)                      # flops of a training step (forward + dgrad + wgrad) of the 6-object graph
    dec_fwd = 0.0                                    # forward Linears of the DECODER: gconv_net_dc + box_net + angle_net
    Eh, H, D = 64, 256, 128
    for _l in range(5):
        dec_fwd += 2.0 * (T * H * 3 * D + T * (2 * H + D) * H + O * H * H + O * D * H)
    dec_fwd += 2.0 * (O * H * (D + Eh // 4) + O * 6 * H + O * H * D + O * 24 * H)
    tf = dec_fwd * n / dt / 1e12
    res = {"layouts_per_s": round(n / dt, 1), "ms_per_heatmap": round(dt * 1e3, 3), "draws": n,
           "ms_min_median_max": [round(sorted(ts)[0] * 1e3, 3), round(dt * 1e3, 3), round(sorted(ts)[-1] * 1e3, 3)],
           "workload": "testing/test_heatmap.py:39-64: 5 objects + room row, 9 triples, %d posterior draws -> 5 centre histograms of 100 x 100; "
                       "eps drawn on the device, one decoder call over %d rows / %d triples, train.py-default model in eval mode" % (n, n * O, n * T),
           "finite": bool(torch.isfinite(h).all().item()), "histograms_sum_to_one": bool(((h.sum((1, 2)) - 1).abs() < 1e-4).all().item()),
           "roofline": {"kernel": "decoder forward of the replicated graphs (gemm_nt family, eval-mode BatchNorm folded into the operand loads)",
                        "bound": "mfma", "unit": "TFLOP/s", "peak": MFMA_F32_PEAK_TFLOPS, "achieved": round(tf, 2), "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4),
                        "flop_per_layout": round(dec_fwd, 1), "traffic": None,
                        "note": "whole heat map (draw, z GEMM, decoder, histogram) over the decoder's forward flops; no per-kernel rocprof split for this leg"}}
    if not args.no_cpu:
        # CPU baseline: the reference's own loop shape - one multivariate-normal draw and ONE single-graph decoder call per layout
        # (test_heatmap.py:52-60) - through the oracle, bounded sample
        from oracle import vae_ref
        ncores = cpu_threads(torch)
        torch.set_num_threads(ncores)
        cfg = vae_ref.VaeConfig()
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        objs, triples, attrs = S.scene_graph_from_words(objs5, rels5)
        m_np, c_np = mean.numpy(), cov.numpy()
        n_cpu = 200
        with torch.no_grad():
            vae_ref.decoder(sd, cfg, torch.zeros(O, E), objs, triples, attrs, training=False)
            t0 = time.perf_counter()
            for _ in range(n_cpu):
                zc = torch.from_numpy(np.random.multivariate_normal(m_np, c_np, O)).float()
                vae_ref.decoder(sd, cfg, zc, objs, triples, attrs, training=False)
            cdt = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": round(n_cpu / cdt, 1), "unit": "layouts/s", "cores": ncores, "kind": "port",
                               "sample": "%d draws, one numpy multivariate-normal draw + one single-graph eval-mode decoder call each "
                                         "(oracle/vae_ref.py), %.2f ms per layout" % (n_cpu, cdt / n_cpu * 1e3)}
    del model
    torch.cuda.empty_cache()
    return res


def spade_leg(args, lib, torch):
    """BASELINE configs[3]: SPADEGenerator4(41,3,256,64,'spectralspadelayer3x3',256,'normal') forward, batch 32,
    256x256 semantic+depth -> RGB, seeded random weights (the authors' checkpoint is not distributable), eval."""
    S = importlib.import_module("3d_sln_amd.host.SPADE_related")
    res = {}
    if not args.no_check:
        res["parity"] = check_spade(torch, S)
    syn = importlib.import_module("3d_sln_amd.host.synthetic")
    SPADE_SEED, IMG_GAIN = 0, 0.04          # = oracle/gen_golden_spade.py BENCH_SEED / BENCH_IMG_GAIN (tests/golden/spade_bench.npz)
    torch.manual_seed(SPADE_SEED)
    G = S.SPADEGenerator4(41, 3, 256, 64, 'spectralspadelayer3x3', 256, 'normal')        # torch's default init, as the reference's constructor
    with torch.no_grad():                   # ... except conv_img: at the default gain tanh sits at |0.97| and hides every error in front of it
        G.conv_img.weight.mul_(IMG_GAIN); G.conv_img.bias.mul_(IMG_GAIN)
    G = G.cuda().eval()
    B = args.spade_batch
    seg, z = syn.spade_input(B, seed=SPADE_SEED)
    seg, z = seg.cuda(), z.cuda()
    g = torch.Generator(device="cuda").manual_seed(0)
    for _ in range(args.spade_warmup):
        out = G(seg, z)
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.spade_iters + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for k in range(args.spade_iters):
        out = G(seg, z)
        marks[k + 1].record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.spade_iters
    per = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(args.spade_iters))
    lib.check(lib.lib().sln_prof_enable(1), "prof")
    G(seg, z)
    torch.cuda.synchronize()
    fam = prof_read(lib)
    lib.check(lib.lib().sln_prof_enable(0), "prof")
    flop_img = 2 * 152.61e9                                       # 152.6 GMAC per 256x256 image
    res.update({"images_per_s": round(B / dt, 2), "ms_per_batch": round(dt * 1e3, 2), "batch": B, "warmup": args.spade_warmup, "iters": args.spade_iters,
                "ms_per_batch_p10_p50_p90": [round(per[int(q * (len(per) - 1))], 2) for q in (0.1, 0.5, 0.9)],
                "tflops_end_to_end": round(flop_img * B / dt / 1e12, 2), "frac_mfma_end_to_end": round(flop_img * B / dt / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                "workload": "BASELINE configs[3]: SPADEGenerator4 256x256 semantic+depth -> RGB, batch %d, fp32, seeded random weights" % B,
                "finite": bool(torch.isfinite(out).all().item())})
    if "conv" in fam:
        c = fam["conv"]
        tf = c["work"] / (c["ms"] * 1e-3) / 1e12
        res["conv_kernels"] = {"launches": c["launches"], "ms": round(c["ms"], 2), "tflops": round(tf, 2),
                               "frac_mfma": round(tf / MFMA_F32_PEAK_TFLOPS, 4)}
        tr, us = profile_rows(["conv_glds_kernel", "conv_mfma_kernel"], "spade")
        res["roofline"] = {"kernel": "conv_glds_kernel + conv_mfma_kernel (reflect-padded implicit GEMM: direct-to-LDS and register-staged "
                                     "variants, %d launches per batch)" % c["launches"], "bound": "mfma",
                           "achieved": round(tf, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4),
                           "traffic": tr, "flop_per_launch": round(c["work"] / c["launches"], 1), "avg_launch_us": round(c["ms"] / c["launches"] * 1e3, 2),
                           "rocprof_avg_launch_us": us, "traffic_source": traffic_source("spade", tr)}
    # colorize_with_spade's own shape (testing/test_SPADE_shade.py:30-79): ONE semantic map, 50 z vectors.  gamma/beta depend
    # on the map only, so they are computed once (sln_spade_apply does the per-sample part).
    nz = 50
    if not args.no_colorize:
        z50 = torch.randn(nz, 256, device="cuda", generator=g)
        G(seg[:1], z50)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            o50 = G(seg[:1], z50)
        torch.cuda.synchronize()
        dt50 = (time.perf_counter() - t0) / 5
        res["colorize_one_map_50z"] = {"images_per_s": round(nz / dt50, 1), "ms_per_room": round(dt50 * 1e3, 2),
                                       "speedup_vs_per_sample_path": round((nz / dt50) / (B / dt), 2),
                                       "finite": bool(torch.isfinite(o50).all().item())}
    if not args.no_dropin and not args.no_colorize:
        res["spade_50x1"] = spade_50x1_leg(args, torch, G, seg, B / dt)
    if not args.no_cpu:
        # CPU baseline: the oracle (PyTorch-CPU restatement = what the reference module computes) on images of the same batch, all cores
        from oracle import spade_ref
        n = cpu_threads(torch)
        torch.set_num_threads(n)
        log('spade CPU baseline on %d threads' % n)
        cfg = spade_ref.SpadeConfig()
        sd = {k: v.detach().cpu() for k, v in G.state_dict().items()}
        bc = 2
        segc, zc = seg[:bc].cpu(), z[:bc].cpu()
        with torch.no_grad():
            ref = spade_ref.generator(sd, cfg, segc, zc)                       # warm-up; doubles as a full-size parity check
            t0 = time.perf_counter()
            n_it = 0
            while n_it < 3 and time.perf_counter() - t0 < 20.0:
                spade_ref.generator(sd, cfg, segc, zc); n_it += 1
            cdt = (time.perf_counter() - t0) / n_it
        if not args.no_check:
            # full-size parity (round 4): |hip - fp64 oracle| <= 1e-4 of the image scale + the fp32 oracle's own distance from the
            # fp64 one (the CPU path itself is ~1.1e-4 away at these weights: tanh of a 1 600-term cancelling sum), and hip within
            # 2e-4 of the CPU fp32 path - the quantity north_star names.  Rounds 1-3 needed 4x the CPU distance here (3.0e-4):
            # the squeeze-excite FCs ran as fp32 chains and the MFMA accumulators as serial chains over K = 9 Cin up to 9 216
            # (tools/spade_error_budget.py locates both; csrc/spade.hip: se_fc_kernel in fp64, blocked accumulation).
            log('spade full-size parity: fp64 oracle, 1 image')
            with torch.no_grad():
                ref64 = spade_ref.generator({k: v.double() for k, v in sd.items()}, cfg, segc[:1].double(), zc[:1].double())
            e_hip, e_cpu = rel_err(out[:1].cpu().numpy(), ref64.numpy()), rel_err(ref[:1].numpy(), ref64.numpy())
            e_hip_cpu = rel_err(out[:1].cpu().numpy(), ref[:1].numpy())
            require(e_hip <= 1e-4 + e_cpu, "SPADE full-size generator (110 M parameters, 256x256): image rel err vs fp64 oracle %.2e "
                                           "(fp32 oracle: %.2e)" % (e_hip, e_cpu))
            require(e_hip_cpu <= 2e-4, "SPADE full-size generator: image rel err vs the CPU fp32 oracle %.2e" % e_hip_cpu)
            res["parity"].update({"full_size_image_rel_err_vs_fp64_oracle": e_hip, "fp32_oracle_rel_err_vs_fp64_oracle": e_cpu,
                                  "full_size_image_rel_err_vs_fp32_oracle": e_hip_cpu})
            # ... and against the REFERENCE's own output on these weights and this image (tests/golden/spade_bench.npz, generated by
            # oracle/gen_golden_spade.py from the reference class in the build container): a crop and seven full rows
            import numpy as np
            fx = np.load(os.path.join(ROOT, "tests", "golden", "spade_bench.npz"))
            sc_f = float(np.abs(fx["out_rows"]).max())
            o0 = out[:1].cpu().numpy()
            e_fix = max(float(np.abs(o0[:, :, 100:132, 60:92] - fx["out_crop"]).max()), float(np.abs(o0[:, :, ::37, :] - fx["out_rows"]).max())) / sc_f
            require(e_fix <= 2e-4, "SPADE full-size generator: image rel err vs the reference-generated fixture %.2e" % e_fix)
            res["parity"].update({"full_size_image_rel_err_vs_reference_fixture": e_fix, "reference_fixture_out_abs_mean": float(fx["out_abs_mean"][0])})
        res["cpu_baseline"] = {"value": round(bc / cdt, 3), "unit": "images/s", "cores": n, "kind": "port",
                               "sample": "%d x batch of %d images of the same input through oracle/spade_ref.py (torch CPU fp32), %.2f s per image"
                                         % (n_it, bc, cdt / bc)}
    return res


def vae_gemm_shapes(O, T, E=64, L=5, box_dim=6, n_angle=24):
    """The Linears of Sg2ScVAEModel at train.py's defaults (decoder_cat, attributes, BatchNorm): (rows, out, in, gathered, bn).
    -> flops of one training step (forward + dgrad + wgrad = 3 x 2MNK) and the algorithmic fp32 operand bytes of its NT launches
    (forward + dgrad) and of its TN problems (wgrad)."""
    D, H, W = 2 * E, 4 * E, 2 * E
    lin = []
    for _net in range(2):
        for _l in range(L):
            lin += [(T, H, 3 * D, True, True), (T, 2 * H + D, H, False, True), (O, H, H, False, True), (O, D, H, False, True)]
    lin += [(O, H, W, False, True), (O, W, H, False, True), (O, E * 3 // 4, W, False, False), (O, E * 3 // 4, W, False, False)]
    lin += [(O, H, W, False, True), (O, W, H, False, True), (O, E // 4, W, False, False), (O, E // 4, W, False, False)]
    lin += [(O, H, W + E // 4, False, True), (O, box_dim, H, False, False), (O, H, W, False, True), (O, n_angle, H, False, False)]
    flop = nt = tn = 0.0
    for M, N, K, _g, bn in lin:
        flop += 3 * 2.0 * M * N * K
        nt += 4.0 * (M * K + N * K + M * N)                                   # forward: x, W, y
        nt += 4.0 * ((2 if bn else 1) * M * N + N * K + M * K + (M * K))        # dgrad: g (+ pre-activation), W^T, dx, mask's xprev (upper bound)
        tn += 4.0 * ((2 if bn else 1) * M * N + M * K + N * K)                  # wgrad: g (+ pre-activation), x, dW
    return {"flop_step": flop, "nt_bytes_step": nt, "tn_bytes_step": tn, "linears": len(lin)}


def main():
    args = parse()
    gc.disable()                                 # collected by hand at the leg boundaries instead, see leg()
    import torch
    import torch.distributed as dist

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # launched plainly (`python bench.py --gpus N`): become the launcher - one rank per GPU under torch.distributed.run,
        # exactly the command line the driver uses; the ranks' output (rank 0 prints the JSON line) passes through
        import socket
        import subprocess
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit("bench.py: --gpus %d but only %d GPU(s) visible; refusing to report a smaller job as n_gpus=%d"
                             % (args.gpus, have, args.gpus))
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(cmd, env=env))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU)" % (args.gpus, world))
    if torch.cuda.device_count() <= local:
        raise SystemExit("bench.py: rank %d has no GPU (LOCAL_RANK=%d, %d visible)" % (rank, local, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    # SLN_BENCH_FORCE_DP=1 exercises the data-parallel code path (graph without Adam + RCCL all-reduce + fused Adam)
    # on a single GPU; used to test that path on a 1-GPU box
    force_dp = os.environ.get("SLN_BENCH_FORCE_DP", "0") == "1"
    if world > 1 or force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":
            del os.environ["NCCL_DEBUG"]               # RCCL's version banner goes to STDOUT, which carries the one JSON line
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    lib = importlib.import_module("3d_sln_amd._lib")
    lib.check(lib.lib().sln_device_ok(), "sln_device_ok")
    M = importlib.import_module("3d_sln_amd.host.Sg2ScVAE_model")
    syn = importlib.import_module("3d_sln_amd.host.synthetic")

    if args.legs_only:
        if world != 1:
            raise SystemExit("bench.py: --legs-only is a single-GPU mode")
        out = {"metric": "scene-graph VAE steps/sec + 256² diff-render fps, 1/2/4/8 MI355X", "value": None, "legs_only": True, "n_gpus": 1}
        if not args.no_render:
            leg('render'); out["render"] = render_leg(args, lib, torch, rank)
        if not args.no_spade:
            leg('spade'); out["spade"] = spade_leg(args, lib, torch)
        if not args.no_graph_build:
            leg('graph-build'); out["graph_build"] = graph_build_leg(args, lib, torch)
        if not args.no_refine:
            leg('refine'); out["refine"] = refine_leg(args, lib, torch)
        if not args.no_sampling:
            leg('sampling'); out["sampling"] = sampling_leg(args, lib, torch)
        print(json.dumps(out))
        return

    torch.manual_seed(42)
    kwargs = dict(vocab=syn.default_vocab(), batch_size=args.graphs, train_3d=True, decoder_cat=True, embedding_dim=64,
                  gconv_mode='feedforward', gconv_num_layers=5, mlp_normalization='batch', vec_noise_dim=0,
                  layout_noise_dim=32, use_AE=False)      # build_dataset_model.py:40-52 at options.py defaults
    model = M.Sg2ScVAEModel(**kwargs).cuda().train()
    model.manual_seed(42 + 7919 * rank)         # every replica draws its own eps
    dp = world > 1 or force_dp
    if dp:
        dist.broadcast(model.flat_params, 0)
        torch.cuda.synchronize()               # the timed loop runs on a side stream
        model.params_changed()
    # a ring of pre-generated batches (same shape, different graphs, different tensors): every step binds a NEW batch, so the
    # per-batch work of a real training loop - staging the inputs, int32 ids, degrees, the CSR of incident triples - is inside
    # the timed region (with ONE reused batch the host mirror skips it); inputs are resident in HBM before the clock starts
    ring = [syn.scene_graph_batch(args.graphs, args.objs, args.triples, seed=1000 + rank + 7919 * k, device="cuda") for k in range(args.batch_ring)]
    b = ring[0]
    batch = (b["objs"], b["triples"], b["boxes"], b["angles"], b["attributes"])
    O = b["objs"].shape[0]
    solo = world == 1          # the side legs, the checks and the CPU baselines are single-GPU measurements (rank 0 at N = 1 only)
    parity = None
    if rank == 0 and solo and not args.no_check:
        log('parity check: VAE')
        parity = check_vae(lib, torch, M, model, batch)
    if rank == 0:
        log('timed VAE loop')
    # eps: by default the step draws its own N(0,1) on the device (Philox, inside the captured iteration), as the reference's
    # forward does with randn_like every call (Sg2ScVAE_model.py:182); --inject-eps feeds one constant tensor instead
    eps = torch.randn(O, 64, device="cuda") if args.inject_eps else None
    stream = torch.cuda.Stream()
    use_graph = not args.no_graph
    T = importlib.import_module("3d_sln_amd.host.train")
    # N > 1: backward, ONE all-reduce of the 15.5 MB flat gradient buffer (averaging inside RCCL), fused Adam - the trainer's
    # own step (host/train.py::DataParallelStep; SLN_DP_OVERLAP=1 ships the decoder half under the encoder's backward)
    dp_step = T.DataParallelStep(model, world, force=force_dp)
    bdicts = [dict(objs=r["objs"], triples=r["triples"], boxes=r["boxes"], angles=r["angles"], attributes=r["attributes"]) for r in ring]
    counter = [0]
    model.validate_inputs = False               # synthetic ids, checked above: no host sync per new batch (as host/train.py does)

    def step():
        counter[0] += 1
        return dp_step(bdicts[counter[0] % len(bdicts)], 0.1, 1e-4, use_graph=use_graph, eps=eps)

    def barrier():
        torch.cuda.synchronize()
        if dp:
            dist.barrier()
            torch.cuda.synchronize()

    gc.collect()
    with torch.cuda.stream(stream):
        for _ in range(args.warmup):
            losses = step()
        barrier()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        for k in range(args.steps):
            losses = step()
            marks[k + 1].record()                # per-step GPU time for the percentiles (SURVEY.md 8d protocol); ~1 us each
        torch.cuda.synchronize()                 # (barrier() starts with the same call)
        dt_local = time.perf_counter() - t0      # this rank's own time, before it waits for the others
        barrier()
        dt = time.perf_counter() - t0
    per_step = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps))
    pct = lambda q: round(per_step[min(len(per_step) - 1, int(q * len(per_step)))], 4)
    if dp:
        tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    coll_us = None
    if dp:                                     # the same collective on its own, after the timed region
        with torch.cuda.stream(stream):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for i in range(25):
                if i == 5:
                    e0.record()
                dp_step._reduce(model.grad_bucket, False)
            e1.record()
        torch.cuda.synchronize()
        coll_us = round(e0.elapsed_time(e1) / 20 * 1e3, 1)
    # SURVEY.md 8(d) asks for >= 20 warm-up + >= 100 timed iterations with median / p10 / p90.  `value` is the EXACT --steps K the
    # caller asked for; when K < 100 (the driver's short run) the same loop runs once more, behind the timed region, at the
    # protocol's length - reported next to it, never as `value`.  Single GPU only (no extra collectives in a multi-rank run).
    survey = None
    if not dp and args.steps < 100:
        with torch.cuda.stream(stream):
            for _ in range(20):
                losses = step()
            m2 = [torch.cuda.Event(enable_timing=True) for _ in range(101)]
            m2[0].record()
            for k in range(100):
                losses = step()
                m2[k + 1].record()
            torch.cuda.synchronize()
        ps = sorted(m2[k].elapsed_time(m2[k + 1]) for k in range(100))
        survey = {"warmup": 20, "steps": 100, "ms_per_step_p10_p50_p90": [round(ps[10], 4), round(ps[50], 4), round(ps[90], 4)],
                  "graphs_per_s_at_p50": round(args.graphs / ps[50] * 1e3, 1)}
    final_loss = float(losses[3].item())
    ms_per_step = dt / args.steps * 1e3
    value = args.graphs * world * args.steps / dt
    dp_diag = None
    if dp:
        # Diagnostics BEHIND the timed region (they do not enter `value`), so that the first SCALE line that exists answers what
        # DESIGN.md section 5 leaves open: (i) every rank's own wall time per step, (ii) the stand-alone all-reduce of the gradient
        # bucket on communicators created with NCCL_ALGO=Ring and =Tree (RCCL reads the variable when a communicator is
        # initialised: a new group per setting), (iii) plain vs overlapped (two-half) iteration, same box, same run.
        dp_diag = {}
        mine = torch.tensor([dt_local / args.steps * 1e3], device="cuda", dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        ranks_ms = [float(x.item()) for x in every]
        dp_diag["ms_per_step_by_rank_min_max"] = [round(min(ranks_ms), 4), round(max(ranks_ms), 4)]
        algo_us = {}
        keep_algo = os.environ.get("NCCL_ALGO")
        for algo in ("Ring", "Tree"):
            try:
                os.environ["NCCL_ALGO"] = algo
                grp = dist.new_group(ranks=list(range(world)), backend="nccl")
                with torch.cuda.stream(stream):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    for i in range(25):
                        if i == 5:
                            e0.record()
                        dist.all_reduce(model.grad_bucket, op=dist.ReduceOp.AVG, group=grp)
                    e1.record()
                torch.cuda.synchronize()
                algo_us[algo] = round(e0.elapsed_time(e1) / 20 * 1e3, 1)
                dist.destroy_process_group(grp)
            except Exception as ex:                      # a diagnostic must not take the measurement down
                algo_us[algo] = "failed: %s" % (str(ex)[:80],)
        if keep_algo is None:
            os.environ.pop("NCCL_ALGO", None)
        else:
            os.environ["NCCL_ALGO"] = keep_algo
        dp_diag["allreduce_us_by_NCCL_ALGO"] = algo_us
        ab = {}
        for name, ov in (("plain", False), ("overlap", True)):
            st2 = T.DataParallelStep(model, world, overlap=ov, force=force_dp)
            if st2.overlap != ov:
                ab[name] = None
                continue
            with torch.cuda.stream(stream):
                for k in range(5):
                    st2(bdicts[k % len(bdicts)], 0.1, 1e-4, use_graph=use_graph, eps=eps)
                barrier()
                t0 = time.perf_counter()
                for k in range(20):
                    st2(bdicts[k % len(bdicts)], 0.1, 1e-4, use_graph=use_graph, eps=eps)
                barrier()
                d2 = torch.tensor([time.perf_counter() - t0], device="cuda", dtype=torch.float64)
            dist.all_reduce(d2, op=dist.ReduceOp.MAX)
            ab[name] = round(float(d2.item()) / 20 * 1e3, 4)
        dp_diag["ms_per_step_plain_vs_overlap"] = ab

    out = {
        "metric": "scene-graph VAE steps/sec + 256² diff-render fps, 1/2/4/8 MI355X",
        "value": round(value, 1), "unit": "graphs/s (scene-graph VAE fwd+loss+bwd+Adam)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "ms_per_step_p10_p50_p90": [pct(0.10), pct(0.50), pct(0.90)],
        "survey_protocol_20_100": survey,
        "steps_per_s": round(args.steps / dt, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: batch=%d scene graphs x (%d objects, %d triples) per GPU, "
                               "Sg2ScVAE train step at train.py defaults (embedding_dim=64, 5+5 GraphTripleConv, BatchNorm)"
                               % (args.graphs, args.objs, args.triples),
                   "O": int(O), "T": int(b["triples"].shape[0]), "hipgraph": bool(use_graph), "distinct_batches_cycled": len(ring),
                   "eps": "constant tensor" if args.inject_eps else "N(0,1) drawn on the device inside the step",
                   "parallelism": "dp%d" % world,
                   "collective": ("all-reduce(avg) of %d fp32 grads + 1 guard element per step%s"
                                  % (model.flat_grads.numel(), " in 2 buckets, decoder half overlapped with the encoder backward" if dp_step.overlap else "")) if dp else None,
                   "allreduce_us_standalone": coll_us, "final_total_loss": round(final_loss, 5)},
    }
    if dp_diag is not None:
        out["data_parallel"] = dp_diag
    if not (final_loss == final_loss and abs(final_loss) < 1e30):
        raise SystemExit("bench.py: the training loss is not finite (%r): nothing reported" % final_loss)
    if parity is not None:
        out["parity"] = parity

    if rank == 0:
        log('VAE loop done: %.3f ms/step' % ms_per_step)
    if rank == 0 and args.prof_steps > 0:
        # ---- per-kernel-family timing, eager launches + HIP events on the launch stream -----------
        with torch.cuda.stream(stream):
            lib.check(lib.lib().sln_prof_enable(1), "prof")
            for _ in range(args.prof_steps):
                model.train_step(*batch, kl_weight=0.1, lr=1e-4, eps=eps, use_graph=False, with_adam=True)
            torch.cuda.synchronize()
            fam = prof_read(lib)
            lib.check(lib.lib().sln_prof_enable(0), "prof")
        kern = {}
        for k, v in fam.items():
            per = v["ms"] / max(v["launches"], 1)
            rate = v["work"] / (v["ms"] * 1e-3) if v["ms"] > 0 else 0.0
            kern[k] = {"launches_per_step": v["launches"] // args.prof_steps, "ms_per_step": round(v["ms"] / args.prof_steps, 4),
                       "avg_us": round(per * 1e3, 2),
                       ("tflops" if k.startswith("gemm") else "gbs"): round(rate / (1e12 if k.startswith("gemm") else 1e9), 2)}
        out["kernels"] = kern
        dom = max((k for k in fam if k.startswith("gemm")), key=lambda k: fam[k]["ms"])
        ach = fam[dom]["work"] / (fam[dom]["ms"] * 1e-3) / 1e12
        # algorithmic operand bytes of the step's GEMMs, from the model's Linear shapes (vae_gemm_shapes below; its flops are
        # checked against what the launchers themselves summed): NT = forward Linears + dgrads, TN = the wgrads
        shp = vae_gemm_shapes(int(O), int(b["triples"].shape[0]))
        flop_prof = sum(v["work"] for k, v in fam.items() if k.startswith("gemm")) / args.prof_steps
        require(abs(shp["flop_step"] - flop_prof) <= 0.01 * flop_prof, "GEMM flops of the step: shapes say %.4g, launchers summed %.4g"
                % (shp["flop_step"], flop_prof))
        prefixes = {"gemm_nt": ["gemm_nt_kernel", "gemm_nt_small_kernel", "gemm_nt16_kernel"], "gemm_tn": ["gemm_tn_multi_kernel", "gemm_tn_kernel"],
                    "gemm_dual": ["gemm_group_kernel", "gemm_dual_kernel"]}
        per_family = {}
        for k in fam:
            if not k.startswith("gemm"):
                continue
            tr_k, us_k = profile_rows(prefixes.get(k, [k + "_kernel"]), "vae")
            a_k = fam[k]["work"] / (fam[k]["ms"] * 1e-3) / 1e12
            per_family[k] = {"bound": "mfma", "achieved": round(a_k, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(a_k / MFMA_F32_PEAK_TFLOPS, 4), "traffic": tr_k,
                             "launches_per_step": fam[k]["launches"] // args.prof_steps,
                             "flop_per_launch": round(fam[k]["work"] / fam[k]["launches"], 1),
                             "avg_launch_us": round(fam[k]["ms"] / fam[k]["launches"] * 1e3, 2), "rocprof_avg_launch_us": us_k}
            # the same fraction from the two clocks: `frac_event` = HIP events around every eager launch of this run (includes the
            # gap to the previous launch), `frac_rocprof` = the kernel durations of the committed trace of the same command
            per_family[k]["frac_event"] = per_family[k]["frac"]
            per_family[k]["frac_rocprof"] = round(fam[k]["work"] / fam[k]["launches"] / (us_k * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4) if us_k else None
        per_family["gemm_nt"]["what"] = "forward Linears and dgrads, one launch each (64x64 tiles of 32x32x2 MFMAs, 64x96 / 64x160 tiles of 16x16x4 MFMAs for N = 384 / 640, 32x32 split-K tiles for the object side)"
        if "gemm_tn" in per_family:
            per_family["gemm_tn"]["what"] = "every wgrad of the iteration in two multi-problem launches (gathered / plain rows; the decoder pass's problems ride with the encoder pass's since round 6)"
            per_family["gemm_tn"]["algorithmic_bytes_per_launch"] = int(shp["tn_bytes_step"] / max(per_family["gemm_tn"]["launches_per_step"], 1))
        if "gemm_dual" in per_family:
            per_family["gemm_dual"]["what"] = "the twin head branches (box / angle), two Linears per launch"
        nt_launches = sum(per_family[k]["launches_per_step"] for k in ("gemm_nt", "gemm_dual") if k in per_family)
        # counter bytes against algorithmic bytes (the tier's wasted-re-read check), per launch: NT = forward Linears + dgrads (the
        # gemm_nt and gemm_dual rows together, as nt_bytes_step counts them), TN = the per-pass wgrad launches
        tr_ntd, _ = profile_rows(prefixes["gemm_nt"] + prefixes["gemm_dual"], "vae")
        for fam_k, tr_k, algo_k in (("gemm_nt", tr_ntd, shp["nt_bytes_step"] / max(nt_launches, 1)),
                                    ("gemm_tn", per_family.get("gemm_tn", {}).get("traffic"),
                                     shp["tn_bytes_step"] / max(per_family.get("gemm_tn", {}).get("launches_per_step", 1), 1))):
            if fam_k in per_family:
                per_family[fam_k]["algorithmic_bytes_per_launch"] = int(algo_k)
                per_family[fam_k]["traffic_over_algorithmic"] = round(tr_k / algo_k, 3) if tr_k else None
                per_family[fam_k]["traffic_source"] = traffic_source("vae", tr_k)
        if "gemm_nt" in per_family:
            per_family["gemm_nt"]["traffic_nt_and_dual_rows"] = tr_ntd
        traffic, prof_us = per_family[dom]["traffic"], per_family[dom]["rocprof_avg_launch_us"]
        if dom == "gemm_nt":
            # `traffic`, `algorithmic_bytes_per_launch` and their quotient over ONE launch set: the forward Linears + dgrads, i.e. the
            # gemm_nt rows and the twin-head gemm_group rows of the profile together (round 5 printed the gemm_nt rows' traffic next
            # to the NT + head launches' algorithmic bytes)
            traffic = tr_ntd
        out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_F32_PEAK_TFLOPS,
                           "unit": "TFLOP/s", "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4), "traffic": traffic,
                           "flop_per_launch": round(fam[dom]["work"] / fam[dom]["launches"], 1),
                           "avg_launch_us": round(fam[dom]["ms"] / fam[dom]["launches"] * 1e3, 2), "rocprof_avg_launch_us": prof_us,
                           "frac_event": per_family[dom]["frac_event"], "frac_rocprof": per_family[dom]["frac_rocprof"],
                           "frac_note": "frac = frac_event (HIP events around the eager launches of this run, gaps included); frac_rocprof "
                                        "divides the same flops by the kernel durations of profiles/%s_vae_kernel_stats.csv" % PROFILE_TAG,
                           "traffic_source": traffic_source("vae", traffic),
                           "traffic_launch_set": ("%d launches per step: gemm_nt + gemm_group rows (forward Linears, dgrads, twin heads)" % nt_launches)
                                                 if dom == "gemm_nt" else "the %s rows" % dom,
                           "algorithmic_bytes_per_launch": per_family[dom if dom in ("gemm_nt", "gemm_tn") else "gemm_nt"]["algorithmic_bytes_per_launch"],
                           "traffic_over_algorithmic": per_family[dom if dom in ("gemm_nt", "gemm_tn") else "gemm_nt"]["traffic_over_algorithmic"],
                           "algorithmic_bytes_note": "launch-weighted mean over the step: fp32 operand rows (x2 for the two-source BatchNorm-"
                                                     "backward operand) + weights + output (+ the pre-activation the mask reads)"}
        out["roofline_kernels"] = per_family
        gemm_flop = sum(v["work"] for k, v in fam.items() if k.startswith("gemm")) / args.prof_steps
        out["roofline_step"] = {"kernel": "whole training step (all launches)", "bound": "mfma", "unit": "TFLOP/s", "peak": MFMA_F32_PEAK_TFLOPS,
                                "achieved": round(gemm_flop / (ms_per_step * 1e-3) / 1e12, 2),
                                "frac": round(gemm_flop / (ms_per_step * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                                "gemm_flop_per_step": round(gemm_flop, 1)}
        step_bytes, step_n = profile_step_traffic("vae")
        survey_bytes = 19.0e6 * args.graphs                      # 1.2 GB per 64-graph step (BatchNorm-materialised model), 19 MB per graph
        out["roofline_step"].update({"traffic": round(step_bytes) if step_bytes else None,
                                     "traffic_source": ("%s: sum over all kernels of bytes per launch x launches, / %d traced steps"
                                                        % (os.path.relpath(profile_csv("vae"), ROOT), step_n)) if step_bytes else PROFILE_STATUS.get("vae", "missing"),
                                     "survey_bytes_per_step": int(survey_bytes),
                                     "traffic_over_survey_bytes": round(step_bytes / survey_bytes, 3) if step_bytes else None,
                                     "hbm_gbs_at_this_step_time": round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1) if step_bytes else None})
        if "edge" in fam:
            e = fam["edge"]
            gbs = e["work"] / (e["ms"] * 1e-3) / 1e9
            out["roofline_edge"] = {"kernel": "edge scatter/gather", "bound": "hbm", "achieved": round(gbs, 1),
                                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None}

    if rank == 0 and not args.no_cpu:
        # ---- CPU baseline: the oracle (PyTorch-CPU port of the reference path) on the same batch, all host cores ----
        # (at N > 1 too, with a shorter sample: a SCALE line then carries its own baseline; the other ranks wait at the barrier)
        from oracle import vae_ref
        cfg = vae_ref.VaeConfig()
        sd = vae_ref.init_state(cfg, seed=42)
        cb = tuple(t.cpu() for t in batch)
        ceps = torch.randn(O, 64, generator=torch.Generator().manual_seed(1))
        keys = vae_ref.trainable_keys(cfg)

        def cpu_rate(nthreads, steps):
            torch.set_num_threads(nthreads)
            s = {k: v.clone() for k, v in sd.items()}
            m = {k: torch.zeros_like(s[k]) for k in keys}; v = {k: torch.zeros_like(s[k]) for k in keys}
            for i in range(2):
                vae_ref.train_step(s, cfg, cb, ceps, 0.1, m, v, step=i + 1)
            t0 = time.perf_counter()
            for i in range(steps):
                vae_ref.train_step(s, cfg, cb, ceps, 0.1, m, v, step=i + 3)
            return (time.perf_counter() - t0) / steps
        ncores = cpu_threads(torch)
        log('CPU baseline: VAE step on %d threads (os.cpu_count() = %s)' % (ncores, os.cpu_count()))
        cpu_steps = args.cpu_steps if solo else max(3, args.cpu_steps // 4)
        cdt = cpu_rate(ncores, cpu_steps)
        out["cpu_baseline"] = {"value": round(args.graphs / cdt, 1), "unit": "graphs/s", "cores": ncores,
                               "kind": "port", "sample": "%d train steps of the same batch (%d graphs), oracle/vae_ref.py, "
                               "torch CPU fp32, %.1f ms/step" % (cpu_steps, args.graphs, cdt * 1e3)}
        if ncores > 16 and solo:                            # the small GEMMs of this step stop scaling beyond ~16 threads: report that point too
            cdt16 = cpu_rate(16, max(5, args.cpu_steps // 2))
            out["cpu_baseline"]["value_at_16_threads"] = round(args.graphs / cdt16, 1)

    if rank == 0 and solo:
        # the BASELINE configs first (render, SPADE), on a memory pool that has only seen the headline loop: behind the
        # large-batch points (11 GB workspaces allocated and released) the scene backward measured 0.55 instead of 0.43 ms
        if not args.no_render:
            leg('render'); out["render"] = render_leg(args, lib, torch, rank)
        if not args.no_spade:
            leg('spade'); out["spade"] = spade_leg(args, lib, torch)
        if not args.no_cpu:
            leg('c1'); out["c1"] = c1_leg(args, torch, M)
        if not args.no_dropin:
            leg('drop-in VAE'); out["vae_dropin"] = vae_dropin_leg(args, torch, M, syn, ms_per_step)
        sizes = [int(x) for x in args.large_batches.split(",") if x.strip()]
        if sizes:
            leg('large-batch'); out["vae_large_batch"] = large_batch_leg(args, torch, M, syn, sizes)
        if not args.no_graph_build:
            leg('graph-build'); out["graph_build"] = graph_build_leg(args, lib, torch)
        if not args.no_refine:
            leg('refine'); out["refine"] = refine_leg(args, lib, torch)
        if not args.no_sampling:
            leg('sampling'); out["sampling"] = sampling_leg(args, lib, torch)
        log('done')
    if rank == 0:
        # LAST key of the line: every config of the metric in one compact object (a log tail then carries both halves of
        # "VAE steps/sec + 256^2 diff-render fps", each leg's roofline fraction and its in-run parity figure)
        def dig(d, *path):
            for k in path:
                if not isinstance(d, dict) or d.get(k) is None:
                    return None
                d = d[k]
            return d
        sg = lambda x: None if x is None else float("%.3g" % x)
        lb = out.get("vae_large_batch") or {}
        big = max(lb, key=lambda k: int(k)) if lb else None
        out["summary"] = {
            "c2_graphs_per_s": out["value"], "c2_ms_step": round(ms_per_step, 4), "c2_frac_mfma_step": dig(out, "roofline_step", "frac"),
            "c2_frac_mfma_gemm_nt": dig(out, "roofline", "frac"), "c2_err": sg(dig(out, "parity", "bench_batch_loss_rel_err_vs_fp64_oracle")),
            "c3_renders_per_s": dig(out, "render", "renders_per_s"), "c3_renders_per_s_median": dig(out, "render", "renders_per_s_median"),
            "c3_frac_hbm": dig(out, "render", "roofline", "frac"), "c3_px_diff": dig(out, "render", "parity", "face_index_pixels_differing"),
            "c4_images_per_s": dig(out, "spade", "images_per_s"), "c4_frac_mfma": dig(out, "spade", "frac_mfma_end_to_end"),
            "c4_err": sg(dig(out, "spade", "parity", "full_size_image_rel_err_vs_reference_fixture")),
            "refine%d_ms" % args.refine_rooms: dig(out, "refine", "rooms_%d" % args.refine_rooms, "ms_per_iteration"),
            "refine%d_ms" % args.refine_rooms_large: dig(out, "refine", "rooms_%d" % args.refine_rooms_large, "ms_per_iteration"),
            "refine_err": sg(max([dig(out, "refine", "parity", k) or 0.0 for k in ("loss", "boxes", "angle_idx", "z")]) if dig(out, "refine", "parity") else None),
            "sampling_layouts_per_s": dig(out, "sampling", "layouts_per_s"),
            "large_batch_frac_mfma": dig(lb, big, "frac_mfma_whole_step") if big else None,
            "cpu_graphs_per_s": dig(out, "cpu_baseline", "value"), "n_gpus": world}
        print(json.dumps(out))
    if dp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
