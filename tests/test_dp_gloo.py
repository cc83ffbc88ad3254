"""World-size-2 test of the data-parallel step on CPU (gloo): sharding of whole scene graphs across ranks,
ONE all-reduce of the flat gradient buffer, identical replicas after the update.  The compute on each rank is
the CPU oracle (the HIP model cannot run without a GPU); what is under test is host/train.py's plumbing."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, pkg

from oracle import vae_ref


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


class CpuStandInPlain:
    """Same surface as Sg2ScVAEModel for train(): flat_params / flat_grads / train_step(with_adam) / adam_step.  Without the
    two-half iteration: DataParallelStep must fall back to one all-reduce after the whole backward."""

    def __init__(self, cfg, seed):
        self.cfg, self.sd = cfg, vae_ref.init_state(cfg, seed)
        self.keys = vae_ref.trainable_keys(cfg)
        self.sizes = [self.sd[k].numel() for k in self.keys]
        self.flat_params = torch.cat([self.sd[k].reshape(-1) for k in self.keys]).clone()
        # [gradients | guard element]: the trainer all-reduces the bucket, adam_step looks at the (then averaged) guard
        self.grad_bucket = torch.zeros(self.flat_params.numel() + 1)
        self.flat_grads = self.grad_bucket[:-1]
        self.m = torch.zeros_like(self.flat_params); self.v = torch.zeros_like(self.flat_params); self.t = 0
        self._views()

    def _views(self):
        o = 0
        for k, n in zip(self.keys, self.sizes):
            self.sd[k] = self.flat_params[o:o + n].view(self.sd[k].shape); o += n

    def params_changed(self):
        self._views()

    def train(self):
        return self

    def state_dict(self):
        return self.sd

    def train_step(self, objs, triples, boxes, angles, attributes, kl_weight=0.1, lr=1e-4, with_adam=True, eps=None, use_graph=True):
        for k in self.keys:
            self.sd[k].requires_grad_(True); self.sd[k].grad = None
        eps = torch.zeros(objs.shape[0], self.cfg.embedding_dim)
        mu, lv, bp, ap = vae_ref.forward(self.sd, self.cfg, objs, triples, boxes, angles, attributes, eps, True)
        total, parts = vae_ref.losses(self.cfg, boxes, bp, angles, ap, mu, lv, kl_weight)
        grads = torch.autograd.grad(total, [self.sd[k] for k in self.keys], allow_unused=True)
        for k in self.keys:
            self.sd[k].requires_grad_(False)
        self.flat_grads.copy_(torch.cat([(g if g is not None else torch.zeros(n)).reshape(-1) for g, n in zip(grads, self.sizes)]))
        self.grad_bucket[-1] = total.detach()
        if with_adam:
            self.adam_step(lr)
        return torch.stack([parts["bbox_pred"], parts["angle_pred"], parts.get("KLD_Gauss", torch.zeros(())), total]).detach()

    def adam_step(self, lr=1e-4):
        if not torch.isfinite(self.grad_bucket[-1]):        # collective guard: the averaged total loss (train.py:79-81)
            return
        self.t += 1
        g = self.flat_grads
        self.m.mul_(0.9).add_(g, alpha=0.1); self.v.mul_(0.999).addcmul_(g, g, value=0.001)
        denom = (self.v.sqrt() / (1 - 0.999 ** self.t) ** 0.5).add_(1e-8)
        self.flat_params.addcdiv_(self.m, denom, value=-lr / (1 - 0.9 ** self.t))


class CpuStandIn(CpuStandInPlain):
    """... plus the two-half iteration (train_step_begin / train_step_finish / decoder_grad_offset) of the overlapped all-reduce."""

    @property
    def decoder_grad_offset(self):
        split, o, offs = sum(self.sizes), 0, []
        for k, n in zip(self.keys, self.sizes):
            offs.append((k, o)); o += n
        for k, start in reversed(offs):
            if not k.startswith(("gconv_net_dc.", "box_net.", "angle_net.")):
                break
            split = start
        return split

    def train_step_begin(self, objs, triples, boxes, angles, attributes, kl_weight=0.1, lr=1e-4, eps=None, use_graph=True):
        """everything is computed here, but only the decoder half of the gradients is handed out (as on the GPU, where
        the encoder half does not exist yet); a trainer that reduces the lower half too early averages zeros"""
        losses = self.train_step(objs, triples, boxes, angles, attributes, kl_weight=kl_weight, lr=lr, with_adam=False)
        self._late = self.flat_grads.clone()
        s = self.decoder_grad_offset
        self.flat_grads[:s] = 0.0
        return losses

    def train_step_finish(self, use_graph=True):
        s = self.decoder_grad_offset
        self.flat_grads[:s] = self._late[:s]


def _worker_nan(rank, world, port, out_dir, two_halves):
    """Two iterations; on the first one rank 1's batch produces a NaN loss.  Every rank must skip that update (same step
    count, parameters untouched by NaN gradients), and the second iteration must leave identical finite replicas."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    T = pkg("host.train")
    cfg = vae_ref.VaeConfig(embedding_dim=16, gconv_num_layers=2)
    model = (CpuStandIn if two_halves else CpuStandInPlain)(cfg, seed=5)
    os.environ["SLN_DP_OVERLAP"] = "1" if two_halves else "0"
    args = T.build_parser().parse_args(["--batch_size", "24", "--num_iterations", "2", "--print_every", "1000"])
    full = vae_ref.synth_batch(24, 5, 8, seed=11, cfg=cfg)
    p_start = model.flat_params.clone()
    seen = []

    def batch_fn(t, lo, hi):
        o0, o1, t0, t1 = lo * 5, hi * 5, lo * 8, hi * 8
        tr = full[1][t0:t1].clone(); tr[:, 0] -= o0; tr[:, 2] -= o0
        boxes = full[2][o0:o1].clone()
        if t == 1 and rank == 1:
            boxes[0, 0] = float("nan")
        if t == 2:
            seen.append((model.t, bool((model.flat_params == p_start).all())))     # state after the guarded first iteration
        return dict(objs=full[0][o0:o1], triples=tr, boxes=boxes, angles=full[3][o0:o1], attributes=full[4][o0:o1])
    T.train(args, model, batch_fn, rank, world, log=lambda *_: None)
    assert seen == [(0, True)], seen                  # iteration 1 skipped on THIS rank too (also on the rank whose loss was finite)
    assert model.t == 1 and bool(torch.isfinite(model.flat_params).all())
    np.save(os.path.join(out_dir, "p%d.npy" % rank), model.flat_params.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("two_halves", [True, False])
def test_rank_local_nan_skips_the_update_on_every_rank(tmp_path, two_halves):
    port = _free_port()
    mp.spawn(_worker_nan, args=(2, port, str(tmp_path), two_halves), nprocs=2, join=True)
    p0, p1 = np.load(tmp_path / "p0.npy"), np.load(tmp_path / "p1.npy")
    assert (p0 == p1).all() and np.isfinite(p0).all()


def _worker(rank, world, port, out_dir, two_halves):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    T = pkg("host.train")
    cfg = vae_ref.VaeConfig(embedding_dim=16, gconv_num_layers=2)
    model = (CpuStandIn if two_halves else CpuStandInPlain)(cfg, seed=5 + rank)      # different init: the broadcast must fix it
    os.environ["SLN_DP_OVERLAP"] = "1"                          # opt in; a model without the two-half API must still fall back
    assert T.DataParallelStep(model, world).overlap == two_halves
    args = T.build_parser().parse_args(["--batch_size", "24", "--num_iterations", "1", "--print_every", "1000"])
    full = vae_ref.synth_batch(24, 5, 8, seed=11, cfg=cfg)

    def batch_fn(t, lo, hi):
        o0, o1, t0, t1 = lo * 5, hi * 5, lo * 8, hi * 8
        tr = full[1][t0:t1].clone(); tr[:, 0] -= o0; tr[:, 2] -= o0
        return dict(objs=full[0][o0:o1], triples=tr, boxes=full[2][o0:o1], angles=full[3][o0:o1], attributes=full[4][o0:o1])
    T.train(args, model, batch_fn, rank, world, log=lambda *_: None)
    np.save(os.path.join(out_dir, "p%d.npy" % rank), model.flat_params.numpy())
    np.save(os.path.join(out_dir, "g%d.npy" % rank), model.flat_grads.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("two_halves", [True, False])
def test_two_rank_step_matches_gradient_average(tmp_path, two_halves):
    """two_halves: all-reduce of the decoder half between the two halves of backward, then the rest (the default on the
    GPU); otherwise one all-reduce after the whole backward."""
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), two_halves), nprocs=2, join=True)
    p0, p1 = np.load(tmp_path / "p0.npy"), np.load(tmp_path / "p1.npy")
    g0, g1 = np.load(tmp_path / "g0.npy"), np.load(tmp_path / "g1.npy")
    assert (p0 == p1).all() and (g0 == g1).all(), "replicas diverged"
    # single-process recomputation: average of the two shard gradients, two steps
    T = pkg("host.train")
    cfg = vae_ref.VaeConfig(embedding_dim=16, gconv_num_layers=2)
    ref = CpuStandIn(cfg, seed=5)
    full = vae_ref.synth_batch(24, 5, 8, seed=11, cfg=cfg)
    for _ in range(1):
        gs = []
        for rank in range(2):
            lo, hi = T.shard_range(24, rank, 2)
            o0, o1, t0, t1 = lo * 5, hi * 5, lo * 8, hi * 8
            tr = full[1][t0:t1].clone(); tr[:, 0] -= o0; tr[:, 2] -= o0
            ref.train_step(full[0][o0:o1], tr, full[2][o0:o1], full[3][o0:o1], full[4][o0:o1], with_adam=False)
            gs.append(ref.flat_grads.clone())
        ref.flat_grads.copy_((gs[0] + gs[1]) / 2)
        ref.adam_step(1e-4)
    # averaged gradient of the last step (tight); parameters: Adam turns the rounding noise of exactly-zero
    # gradients (biases in front of BatchNorm) into +-lr steps, so bound those and require the bulk to agree
    gr = ref.flat_grads.numpy()
    assert np.abs(g0 - gr).max() <= 1e-4 * np.abs(gr).max() + 1e-7
    d = np.abs(p0 - ref.flat_params.numpy())
    assert d.max() <= 2.05 * 1e-4 and np.mean(d > 1e-6) < 0.08


def test_shard_ranges_cover_batch():
    T = pkg("host.train")
    for n, w in ((512, 8), (10, 3), (7, 8), (64, 1)):
        r = [T.shard_range(n, k, w) for k in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))


def _slice_graphs(full, lo, hi, n_obj=5, n_tri=8):
    o0, o1, t0, t1 = lo * n_obj, hi * n_obj, lo * n_tri, hi * n_tri
    tr = full[1][t0:t1].clone(); tr[:, 0] -= o0; tr[:, 2] -= o0
    return dict(objs=full[0][o0:o1], triples=tr, boxes=full[2][o0:o1], angles=full[3][o0:o1], attributes=full[4][o0:o1])


def _worker_ragged(rank, world, port, out_dir):
    """The short last batch of an epoch (build_dataset_model.py:28-34: drop_last=False): iteration 1 has 3 graphs for 2 ranks
    (shards of 2 and 1: unequal weights), iteration 2 has ONE graph (rank 1 gets nothing).  No rank may hang or raise, the
    replicas stay identical, and the averaged gradient is the gradient of the WHOLE batch (row-weighted, not mean of means)."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    T = pkg("host.train")
    cfg = vae_ref.VaeConfig(embedding_dim=16, gconv_num_layers=2, mlp_normalization="none")
    model = CpuStandIn(cfg, seed=5)
    os.environ["SLN_DP_OVERLAP"] = "1"                     # asked for, but the weighted step must not take the two-half route
    args = T.build_parser().parse_args(["--batch_size", "3", "--num_iterations", "2", "--print_every", "1000"])
    full = vae_ref.synth_batch(4, 5, 8, seed=11, cfg=cfg)
    sizes = {1: 3, 2: 1}
    grads = {}

    def batch_fn(t, lo, hi):
        if t == 2:
            grads[1] = model.flat_grads.clone()
        base = 0 if t == 1 else 3
        lo2, hi2 = T.shard_range(sizes[t], rank, world)
        return None if hi2 == lo2 else _slice_graphs(full, base + lo2, base + hi2)
    batch_fn.ragged = True
    T.train(args, model, batch_fn, rank, world, log=lambda *_: None)
    assert model.t == 2
    np.save(os.path.join(out_dir, "p%d.npy" % rank), model.flat_params.numpy())
    np.save(os.path.join(out_dir, "g1_%d.npy" % rank), grads[1].numpy())
    np.save(os.path.join(out_dir, "g2_%d.npy" % rank), model.flat_grads.numpy())
    dist.destroy_process_group()


def test_ragged_last_batch_empty_shard_and_row_weighted_average(tmp_path):
    port = _free_port()
    mp.spawn(_worker_ragged, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    p0, p1 = np.load(tmp_path / "p0.npy"), np.load(tmp_path / "p1.npy")
    assert (p0 == p1).all() and np.isfinite(p0).all(), "replicas diverged"
    cfg = vae_ref.VaeConfig(embedding_dim=16, gconv_num_layers=2, mlp_normalization="none")
    full = vae_ref.synth_batch(4, 5, 8, seed=11, cfg=cfg)
    ref = CpuStandIn(cfg, seed=5)
    # iteration 1: without BatchNorm the graphs are independent, so the gradient of the 3-graph batch IS the row-weighted mean
    ref.train_step(**_slice_graphs(full, 0, 3), with_adam=False)
    whole = ref.flat_grads.numpy().copy()
    g1 = np.load(tmp_path / "g1_0.npy")
    assert (g1 == np.load(tmp_path / "g1_1.npy")).all()
    assert np.abs(g1 - whole).max() <= 1e-5 * np.abs(whole).max() + 1e-8
    # ... and an unweighted mean of the two shard means would have been visibly different
    a = CpuStandIn(cfg, seed=5); a.train_step(**_slice_graphs(full, 0, 2), with_adam=False)
    b = CpuStandIn(cfg, seed=5); b.train_step(**_slice_graphs(full, 2, 3), with_adam=False)
    naive = 0.5 * (a.flat_grads + b.flat_grads).numpy()
    assert np.abs(naive - whole).max() > 1e-3 * np.abs(whole).max()
    # iteration 2: one graph, rank 1 empty - the reduced gradient is rank 0's
    ref.adam_step(1e-4)
    ref.train_step(**_slice_graphs(full, 3, 4), with_adam=False)
    g2 = np.load(tmp_path / "g2_0.npy")
    assert (g2 == np.load(tmp_path / "g2_1.npy")).all()
    assert np.abs(g2 - ref.flat_grads.numpy()).max() <= 1e-4 * np.abs(g2).max() + 1e-7


# ---------------------------------------------------------------------------------------------- world 8 (BASELINE config 5's shape)
def _worker8(rank, world, port, out_dir, two_halves):
    """Config 5 in miniature: 8 ranks, equal shards of whole graphs (24 graphs -> 3 per rank), two iterations."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    T = pkg("host.train")
    cfg = vae_ref.VaeConfig(embedding_dim=16, gconv_num_layers=2)
    model = (CpuStandIn if two_halves else CpuStandInPlain)(cfg, seed=5 + rank)      # different init: the broadcast must fix it
    os.environ["SLN_DP_OVERLAP"] = "1" if two_halves else "0"
    order = []
    if two_halves:                      # the decoder half must be reduced BEFORE the encoder half exists (train_step_finish)
        fin = model.train_step_finish
        model.train_step_finish = lambda **k: (order.append("finish"), fin(**k))[1]
        red = T.DataParallelStep._reduce

        def spy(self, buf, async_op):
            order.append("reduce%d" % buf.numel())
            return red(self, buf, async_op)
        T.DataParallelStep._reduce = spy
    args = T.build_parser().parse_args(["--batch_size", "24", "--num_iterations", "2", "--print_every", "1000"])
    full = vae_ref.synth_batch(24, 5, 8, seed=11, cfg=cfg)
    T.train(args, model, lambda t, lo, hi: _slice_graphs(full, lo, hi), rank, world, log=lambda *_: None)
    if two_halves:
        n, s = model.grad_bucket.numel(), model.decoder_grad_offset
        assert order == ["reduce%d" % (n - s), "finish", "reduce%d" % s] * 2, order
    assert model.t == 2
    np.save(os.path.join(out_dir, "p%d.npy" % rank), model.flat_params.numpy())
    np.save(os.path.join(out_dir, "g%d.npy" % rank), model.flat_grads.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("two_halves", [True, False])
def test_eight_rank_step_matches_gradient_average(tmp_path, two_halves):
    port = _free_port()
    mp.spawn(_worker8, args=(8, port, str(tmp_path), two_halves), nprocs=8, join=True)
    ps = [np.load(tmp_path / ("p%d.npy" % r)) for r in range(8)]
    gs = [np.load(tmp_path / ("g%d.npy" % r)) for r in range(8)]
    assert all((p == ps[0]).all() for p in ps) and all((g == gs[0]).all() for g in gs), "replicas diverged"
    T = pkg("host.train")
    cfg = vae_ref.VaeConfig(embedding_dim=16, gconv_num_layers=2)
    ref = CpuStandIn(cfg, seed=5)
    full = vae_ref.synth_batch(24, 5, 8, seed=11, cfg=cfg)
    for _ in range(2):
        acc = torch.zeros_like(ref.flat_grads)
        for rank in range(8):
            lo, hi = T.shard_range(24, rank, 8)
            assert hi - lo == 3
            ref.train_step(**_slice_graphs(full, lo, hi), with_adam=False)
            acc += ref.flat_grads
        ref.flat_grads.copy_(acc / 8)
        ref.adam_step(1e-4)
    gr = ref.flat_grads.numpy()
    assert np.abs(gs[0] - gr).max() <= 2e-4 * np.abs(gr).max() + 1e-7       # second step: parameters already moved by Adam's +-lr noise
    d = np.abs(ps[0] - ref.flat_params.numpy())
    assert d.max() <= 4.1 * 1e-4 and np.mean(d > 1e-6) < 0.12


def _worker8_ragged(rank, world, port, out_dir):
    """Real rooms on 8 ranks: iteration 1 has 11 graphs (shard_range remainders: 2,2,2,1,1,1,1,1), iteration 2 the short last
    batch of an epoch with 3 graphs (ranks 3..7 WITHOUT a graph), iteration 3 a single graph (7 empty ranks)."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    T = pkg("host.train")
    cfg = vae_ref.VaeConfig(embedding_dim=16, gconv_num_layers=2, mlp_normalization="none")
    model = CpuStandIn(cfg, seed=5)
    args = T.build_parser().parse_args(["--batch_size", "11", "--num_iterations", "3", "--print_every", "1"])
    full = vae_ref.synth_batch(15, 5, 8, seed=11, cfg=cfg)
    base, sizes = {1: 0, 2: 11, 3: 14}, {1: 11, 2: 3, 3: 1}
    grads = {}

    def batch_fn(t, lo, hi):
        if t > 1:
            grads[t - 1] = model.flat_grads.clone()
        lo2, hi2 = T.shard_range(sizes[t], rank, world)
        return None if hi2 == lo2 else _slice_graphs(full, base[t] + lo2, base[t] + hi2)
    batch_fn.ragged = True
    ck = T.train(args, model, batch_fn, rank, world, log=lambda *_: None)
    grads[3] = model.flat_grads.clone()
    assert model.t == 3
    np.save(os.path.join(out_dir, "p%d.npy" % rank), model.flat_params.numpy())
    for t in (1, 2, 3):
        np.save(os.path.join(out_dir, "g%d_%d.npy" % (t, rank)), grads[t].numpy())
    if rank == 0:
        np.save(os.path.join(out_dir, "losses.npy"), np.array([ck['losses'][k] for k in ('bbox_pred', 'angle_pred', 'KLD_Gauss', 'total_loss')]))
    dist.destroy_process_group()


def test_eight_ranks_ragged_shards_with_several_empty_ranks(tmp_path):
    port = _free_port()
    mp.spawn(_worker8_ragged, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    ps = [np.load(tmp_path / ("p%d.npy" % r)) for r in range(8)]
    assert all((p == ps[0]).all() for p in ps) and np.isfinite(ps[0]).all(), "replicas diverged"
    cfg = vae_ref.VaeConfig(embedding_dim=16, gconv_num_layers=2, mlp_normalization="none")
    full = vae_ref.synth_batch(15, 5, 8, seed=11, cfg=cfg)
    ref = CpuStandIn(cfg, seed=5)
    logged = np.load(tmp_path / "losses.npy")                  # [4 names, 3 iterations], rank 0's log
    for t, (lo, hi) in enumerate(((0, 11), (11, 14), (14, 15)), start=1):
        # without BatchNorm the graphs are independent: the gradient / the losses of the whole batch ARE the row-weighted means
        whole_losses = ref.train_step(**_slice_graphs(full, lo, hi), with_adam=False).numpy()
        whole = ref.flat_grads.numpy().copy()
        g = [np.load(tmp_path / ("g%d_%d.npy" % (t, r))) for r in range(8)]
        assert all((x == g[0]).all() for x in g)
        assert np.abs(g[0] - whole).max() <= 1e-4 * np.abs(whole).max() + 1e-7, t
        # the logged losses are the global (row-weighted) means, not rank 0's shard means
        assert np.abs(logged[:, t - 1] - whole_losses).max() <= 1e-4 * np.abs(whole_losses).max() + 1e-6, (t, logged[:, t - 1], whole_losses)
        ref.adam_step(1e-4)
