"""Training loop of the scene-graph VAE on MI355X: counterpart of the reference's ``train.py`` (:10-122).

Single GPU:   python -m 3d_sln_amd.host.train ...        (module name is not an identifier: use
              ``python 3d_sln_amd/host/train.py`` or ``runpy``)
One node, N GPUs (the reference asserts ``Multi-GPU not supported``, build_dataset_model.py:54-55):
              python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
                     3d_sln_amd/host/train.py --batch_size 512

Data parallelism: one process per GPU; every rank owns a contiguous block of whole scene graphs
(graphs never share rows, data/suncg_dataset.py:318-325), runs the fused forward/loss/backward, then
ONE all-reduce (RCCL over xGMI; ``backend='nccl'``) of the flat 15.5 MB fp32 gradient buffer, divides by
the world size and applies the fused Adam.  BatchNorm statistics stay per replica (standard DDP
semantics; a shard of 64 graphs behaves exactly like a single-GPU run at batch 64).

Flag names follow options/options.py:20-57.  The SUNCG metadata is not distributable, so batches come
from ``synthetic.scene_graph_batch`` unless a ``batch_fn`` is supplied.
"""
import argparse
import math
import os
import sys
from collections import defaultdict

import torch
import torch.distributed as dist

if __package__ in (None, ""):                      # executed as a script
    import importlib
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    synthetic = importlib.import_module("3d_sln_amd.host.synthetic")
else:
    from . import synthetic


def bool_flag(s):
    if s in ('1', '0'):
        return s == '1'
    raise ValueError('Invalid value "%s" for bool flag (should be 0 or 1)' % s)


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--embedding_dim', default=64, type=int)
    p.add_argument('--gconv_mode', default='feedforward')
    p.add_argument('--gconv_num_layers', default=5, type=int)
    p.add_argument('--mlp_normalization', default='batch', type=str)
    p.add_argument('--vec_noise_dim', default=0, type=int)
    p.add_argument('--layout_noise_dim', default=32, type=int)
    p.add_argument('--batch_size', default=128, type=int, help="GLOBAL batch (scene graphs per step over all GPUs)")
    p.add_argument('--num_iterations', default=600000, type=int)
    p.add_argument('--eval_mode_after', default=-1, type=int)
    p.add_argument('--learning_rate', default=1e-4, type=float)
    p.add_argument('--print_every', default=100, type=int)
    p.add_argument('--checkpoint_every', default=1000, type=int)
    p.add_argument('--snapshot_every', default=10000, type=int)
    p.add_argument('--output_dir', default='./checkpoints')
    p.add_argument('--checkpoint_name', default='latest_checkpoint')
    p.add_argument('--restore_from_checkpoint', default=False, type=bool_flag)
    p.add_argument('--KL_loss_weight', default=0.1, type=float)
    p.add_argument('--use_AE', default=False, type=bool_flag)
    p.add_argument('--decoder_cat', default=True, type=bool_flag)
    p.add_argument('--train_3d', default=True, type=bool_flag)
    p.add_argument('--KL_linear_decay', default=False, type=bool_flag)
    p.add_argument('--manual_seed', default=42, type=int)
    p.add_argument('--suncg_train_dir', default=None,
                   help="data_rot_train.json (options.py:19); when given, batches come from the device scene-graph builder "
                        "(host/suncg_dataset.py) instead of the synthetic generator")
    p.add_argument('--metadata_dir', default='metadata', help="valid_types.json / size_info_many.json / 30_size_info_many.json")
    p.add_argument('--use_attr_30', default=True, type=bool_flag)
    p.add_argument('--objs_per_graph', default=32, type=int)
    p.add_argument('--triples_per_graph', default=64, type=int)
    return p


def shard_range(n_items, rank, world):
    """Contiguous block of whole graphs owned by ``rank`` (remainder spread over the first ranks)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class DataParallelStep:
    """One data-parallel iteration: every rank runs the step on its own graphs, the flat fp32 gradient buffer is averaged
    over the ranks (RCCL over xGMI, the only exchange of the path), then the fused Adam update runs on identical gradients.

    ``overlap`` (or SLN_DP_OVERLAP=1): the backward pass is issued in two halves (``train_step_begin`` /
    ``train_step_finish``): when the decoder's backward is done the gradients of ``gconv_net_dc`` / ``box_net`` /
    ``angle_net`` - the contiguous upper half of the flat buffer - are final, so their all-reduce runs on the collective's
    stream while the encoder's backward still computes; the lower half follows.  Off by default: on one GPU (world of one,
    where the collective itself is almost free) the two-half sequence costs 0.10 ms per step more than the plain one
    (second graph launch, second collective, RCCL's kernel sharing the CUs with the encoder's GEMMs), about what hiding
    half of a 15.5 MB all-reduce can win back - it has to be measured on an 8-GPU node before it becomes the default."""

    def __init__(self, model, world, overlap=None, force=False, weighted=False):
        self.model, self.world = model, world
        self.dp = world > 1 or force                # force: the collective path with a world of one (single-GPU test of it)
        if overlap is None:
            overlap = os.environ.get("SLN_DP_OVERLAP", "0") == "1"
        # ``weighted``: shards of unequal size (real rooms; the short last batch of an epoch, which may leave ranks WITHOUT any
        # graph).  The reference's three loss terms are means over the object rows of the whole batch (utils.py:16-27), so the
        # gradient of the global batch is sum_r O_r g_r / sum_r O_r: every rank scales its bucket by its own row count O_r, the
        # all-reduce sums, one more one-element all-reduce sums the O_r, and the bucket is divided on the device.  A rank with
        # an empty shard skips the compute and contributes zeros with weight 0 - it still joins both collectives, so nobody
        # waits for it forever.  Equal shards (the synthetic generator, the bench) keep the plain average: no extra passes.
        self.weighted = bool(weighted) and self.dp
        self.split = int(getattr(model, "decoder_grad_offset", 0)) if hasattr(model, "train_step_begin") else 0
        self.overlap = bool(overlap) and self.dp and not self.weighted and 0 < self.split < model.flat_grads.numel()
        # Collective non-finite guard (the reference's 'not backpropping', train.py:79-81, is single-GPU): a model with a
        # ``grad_bucket`` leaves its total loss in the element behind the gradients; the all-reduce averages it with them, and
        # ``adam_step`` skips on EVERY rank when that average is not finite - one NaN rank can neither poison the other replicas
        # through the averaged gradients nor let their step counters drift apart.
        self.guarded = hasattr(model, "grad_bucket")
        # RCCL divides inside the collective; gloo (CPU tests) has no AVG
        self.avg_in_collective = self.dp and dist.get_backend() == "nccl" and not self.weighted
        self._w = None

    def _reduce(self, buf, async_op):
        op = dist.ReduceOp.AVG if self.avg_in_collective else dist.ReduceOp.SUM
        return dist.all_reduce(buf, op=op, async_op=async_op)

    def __call__(self, b, kl_weight, lr, use_graph=True, eps=None):
        m = self.model
        g = m.grad_bucket if self.guarded else m.flat_grads       # [gradients | total loss of this rank]
        kw = dict(kl_weight=kl_weight, lr=lr, use_graph=use_graph)
        if eps is not None:
            kw["eps"] = eps
        rows = 0 if b is None else int(b["objs"].shape[0])
        if rows == 0 and not self.weighted:
            raise ValueError("empty shard: build DataParallelStep(weighted=True) for batches that may leave a rank without graphs")
        args = None if rows == 0 else (b["objs"], b["triples"], b["boxes"], b["angles"], b["attributes"])
        if not self.dp:
            return m.train_step(*args, with_adam=True, **kw)
        if self.weighted:
            # _w = [rows | rows x the four loss values]: one small all-reduce carries the weight total AND the row-weighted global
            # loss means (what the reduced gradients belong to; a rank's own shard means would be logged and checkpointed otherwise)
            if self._w is None:
                self._w = torch.zeros(5, dtype=g.dtype, device=g.device)
            if rows:
                losses = m.train_step(*args, with_adam=False, **kw)
                g.mul_(float(rows))
                self._w[1:] = losses.detach().to(g.dtype) * float(rows)
            else:
                self._w[1:] = 0
                g.zero_()
            self._w[:1] = float(rows)
            self._reduce(g, False)
            self._reduce(self._w, False)
            g.div_(self._w[:1])                                 # on the device: no host read-back of the total
            m.adam_step(lr=lr)
            return self._w[1:] / self._w[:1]
        if self.overlap:
            losses = m.train_step_begin(*args, **kw)
            w_dec = self._reduce(g[self.split:], True)          # waits for the first half, runs beside the second
            m.train_step_finish(use_graph=use_graph)
            w_enc = self._reduce(g[:self.split], True)
            w_dec.wait(); w_enc.wait()                          # stream-side waits on the GPU; blocking on gloo
        else:
            losses = m.train_step(*args, with_adam=False, **kw)
            self._reduce(g, False)
        if not self.avg_in_collective:
            g.mul_(1.0 / self.world)
        m.adam_step(lr=lr)
        return losses


class EpochSampler:
    """DataLoader(shuffle=True, drop_last=False) of build_dataset_model.py:28-34 as a pure function of the step: every
    epoch is one seeded permutation of the rooms (no replacement), cut into batches; the last batch of an epoch may be
    short.  The permutation depends on (seed, epoch) only, so every rank computes the same one and takes its own block."""

    def __init__(self, n_items, batch_size, seed):
        self.n, self.bs, self.seed = int(n_items), int(batch_size), int(seed)
        self.per_epoch = max(1, (self.n + self.bs - 1) // self.bs)
        self._cache = (None, None)

    def epoch(self, t):
        """1-based epoch of training step t (t counts from 1, as train.py:56-66)."""
        return (t - 1) // self.per_epoch + 1

    def batch(self, t):
        e, k = (t - 1) // self.per_epoch, (t - 1) % self.per_epoch
        if self._cache[0] != e:
            self._cache = (e, torch.randperm(self.n, generator=torch.Generator().manual_seed(self.seed + 1000003 * e)))
        return self._cache[1][k * self.bs:(k + 1) * self.bs]


def kl_weight_at(args, t):
    return 10 ** (t // 1e5 - 6) if args.KL_linear_decay else args.KL_loss_weight        # train.py:73-76


def train(args, model, batch_fn, rank=0, world=1, log=print, use_graph=True):
    """train.py:56-114.  ``model`` exposes train_step(..., with_adam=False) / adam_step / flat_params / flat_grads and,
    optionally, train_step_begin / train_step_finish / decoder_grad_offset (Sg2ScVAEModel on the GPU; tests plug a CPU
    stand-in).  ``batch_fn(t, lo, hi)`` returns the rank's graphs; a ``batch_fn.ragged = True`` attribute says that shards may
    differ in size or be empty (it then returns None for an empty shard) and switches the step to the row-weighted average."""
    if world > 1:
        dist.broadcast(model.flat_params, 0)                      # identical replicas
        if hasattr(model, "params_changed"):
            model.params_changed()
    step = DataParallelStep(model, world, weighted=bool(getattr(batch_fn, "ragged", False)))
    if hasattr(model, "validate_inputs"):
        model.validate_inputs = False             # no per-batch host sync inside the loop
    lo, hi = shard_range(args.batch_size, rank, world)
    checkpoint = {'args': dict(vars(args)), 'losses_ts': [], 'losses': defaultdict(list), 'checkpoint_ts': [],
                  'counters': {'t': None, 'epoch': None}, 'model_state': None, 'optim_state': None}
    t = 0
    # train.py:16-31: resume from '<checkpoint_name>_with_model.pt' (model, optimizer - torch.optim.Adam's own state layout -, t)
    # (the reference restores '<name>_with_model.pt' but writes 'latest_<name>_with_model.pt', :18 vs :103 - only its default
    # name hides the mismatch; here the file this loop wrote is found under either spelling)
    restore_path = None
    if getattr(args, "restore_from_checkpoint", False):
        for cand in ('%s_with_model.pt', 'latest_%s_with_model.pt'):
            restore_path = os.path.join(args.output_dir, cand % args.checkpoint_name)
            if os.path.isfile(restore_path):
                break
    if restore_path is not None and os.path.isfile(restore_path) and hasattr(model, "load_optim_state_dict"):
        log('Restoring from checkpoint:')
        log(restore_path)
        ck = torch.load(restore_path, map_location="cpu", weights_only=False)
        model.load_state_dict(ck['model_state'])
        if ck.get('optim_state') is not None:
            model.load_optim_state_dict(ck['optim_state'])
        t = ck['counters']['t']
        checkpoint['counters']['epoch'] = ck['counters'].get('epoch')
        model.eval() if 0 <= args.eval_mode_after <= t else model.train()
        checkpoint.update({k: ck[k] for k in ('losses_ts', 'losses', 'checkpoint_ts') if k in ck})
    while t < args.num_iterations:
        if t == args.eval_mode_after:
            model.eval()
        t += 1
        losses = step(batch_fn(t, lo, hi), kl_weight_at(args, t), args.learning_rate, use_graph=use_graph)
        if t % args.print_every == 0 or t == args.num_iterations:
            vals = [float(x) for x in losses.detach().cpu()]          # the only host sync, every print_every steps
            if not math.isfinite(vals[3]):
                log('WARNING: Got loss = NaN, not backpropping')     # train.py:79-81; the fused step skipped the update on the device
            if rank == 0:
                log("On batch {} out of {}".format(t, args.num_iterations))
                for name, v in zip(('bbox_pred', 'angle_pred', 'KLD_Gauss', 'total_loss'), vals):
                    log(' [%s]: %.4f' % (name, v))
                    checkpoint['losses'][name].append(v)
                checkpoint['losses_ts'].append(t)
        if rank == 0 and t % args.checkpoint_every == 0:
            checkpoint['model_state'] = {k: v.detach().cpu() for k, v in model.state_dict().items()}
            if hasattr(model, "optim_state_dict"):
                osd = model.optim_state_dict(args.learning_rate)                                # train.py:94
                osd['state'] = {i: {k: v.detach().cpu() for k, v in st.items()} for i, st in osd['state'].items()}
                checkpoint['optim_state'] = osd
            checkpoint['counters']['t'] = t
            checkpoint['counters']['epoch'] = getattr(batch_fn, "epoch", lambda _t: None)(t)
            os.makedirs(args.output_dir, exist_ok=True)
            torch.save(checkpoint, os.path.join(args.output_dir, 'latest_%s_with_model.pt' % args.checkpoint_name))
    return checkpoint


def main(argv=None):
    import importlib
    args = build_parser().parse_args(argv)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.manual_seed(args.manual_seed)
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    M = importlib.import_module("3d_sln_amd.host.Sg2ScVAE_model")
    dataset = None
    if args.suncg_train_dir:
        suncg_dataset = importlib.import_module("3d_sln_amd.host.suncg_dataset")
        dataset = suncg_dataset.SuncgDataset(args.suncg_train_dir, args.train_3d, use_attr_30=args.use_attr_30,
                                             metadata_dir=args.metadata_dir)
        if rank == 0:
            print('Training dataset has %d scenes and %d objects' % (len(dataset), dataset.total_objects()))
    model = M.Sg2ScVAEModel(vocab=dataset.vocab if dataset is not None else synthetic.default_vocab(), batch_size=args.batch_size,
                            train_3d=args.train_3d,
                            decoder_cat=args.decoder_cat, embedding_dim=args.embedding_dim, gconv_mode=args.gconv_mode,
                            gconv_num_layers=args.gconv_num_layers, mlp_normalization=args.mlp_normalization,
                            vec_noise_dim=args.vec_noise_dim, layout_noise_dim=args.layout_noise_dim,
                            use_AE=args.use_AE).cuda().train()
    model.manual_seed(args.manual_seed + 7919 * rank)        # the on-device N(0,1) draws differ between the replicas (same parameters)

    sampler = EpochSampler(len(dataset), args.batch_size, args.manual_seed) if dataset is not None else None

    def batch_fn(t, lo, hi):
        if dataset is not None:                                  # DataLoader(shuffle=True) + collate (build_dataset_model.py:28-34), on the device
            idx = sampler.batch(t)
            lo2, hi2 = shard_range(len(idx), rank, world)        # the epoch's last batch may be short (drop_last=False) ...
            if hi2 == lo2:
                return None                                      # ... and leave this rank without a room: it contributes weight 0
            _, objs, boxes, triples, angles, attrs, _, _ = dataset.build_batch(idx[lo2:hi2])
            return dict(objs=objs, triples=triples, boxes=boxes, angles=angles, attributes=attrs)
        return synthetic.scene_graph_batch(hi - lo, args.objs_per_graph, args.triples_per_graph, seed=t * 100003 + lo,
                                           box_dim=6 if args.train_3d else 4, device="cuda")
    if sampler is not None:
        batch_fn.epoch = sampler.epoch
        batch_fn.ragged = True                                   # rooms differ in size: row-weighted gradient average
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        # real rooms differ in size from batch to batch: eager launches (a hipGraph is tied to one (O, T) pair).  Nothing in such a
        # step waits on the host: the wgrad problem tables that change with (O, T) go to the device in stream order (pinned ring,
        # csrc/vae_engine.hip::stage_upload) and the launches queue ahead of the GPU - 2.22 ms per step with a new shape every
        # step against 2.05 ms for a fixed shape (tools/varshape_time.py; 2.40 ms while the upload sat behind a stream drain)
        train(args, model, batch_fn, rank, world, use_graph=dataset is None)
    torch.cuda.synchronize()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
