"""HIP-backed drop-in for the reference's ``models/Sg2ScVAE_model.py::Sg2ScVAEModel``.

Same constructor keywords (build_dataset_model.py:40-52), same sub-module / parameter names
(so ``state_dict`` / ``load_state_dict`` interoperate with reference checkpoints), same
``encoder`` / ``decoder`` / ``forward`` call surfaces (models/Sg2ScVAE_model.py:115-188).
All arithmetic runs in libsln_hip.so through the C ABI (include/sln_hip.h); this file only
owns the parameters (packed into one flat fp32 buffer, which is also what the data-parallel
trainer all-reduces and what the fused Adam kernel walks) and wires autograd.

There is no CPU path: calling the model before ``.cuda()`` raises.
"""
from __future__ import annotations

import ctypes as C
import weakref

import torch
import torch.nn as nn

from .. import _lib
from .graph import GraphTripleConvNet, _init_weights, make_mlp, mlp_linears

_ALIGN = 64   # floats; keeps every tensor 256-byte aligned inside the flat buffers


class _ForwardFn(torch.autograd.Function):
    """mu, logvar, boxes_pred, angles_pred = model(...) with gradients routed to the engine."""

    @staticmethod
    def forward(ctx, anchor, model, eps, training):
        model.params_changed()            # a torch optimizer may have stepped since the last backward (dgrad uses cached W^T)
        mu, lv, z, bp, ap = model._engine_forward(eps, training)
        if eps is None and not model.use_AE:
            eps = model.last_eps()        # drawn on the device (Sg2ScVAE_model.py:182); backward needs it
        ctx.model, ctx.gen = model, model._generation
        ctx.save_for_backward(eps, lv)
        return mu, lv, bp, ap

    @staticmethod
    def backward(ctx, dmu, dlv, dbp, dap):
        model = ctx.model
        eps, lv = ctx.saved_tensors
        model._check_generation(ctx.gen)
        model._alias_grads()
        dz = model._engine_decoder_backward(dbp, dap)
        if model.use_AE:
            dmu_t = dz if dmu is None else dmu + dz
            dlv_t = dlv
        else:
            dmu_t = dz if dmu is None else dmu + dz
            dlv_z = dz * eps * (0.5 * torch.exp(0.5 * lv))
            dlv_t = dlv_z if dlv is None else dlv + dlv_z
        model._engine_encoder_backward(dmu_t, dlv_t)
        return None, None, None, None


class _EncoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, training):
        model.params_changed()
        mu, lv = model._engine_encoder(training)
        ctx.model, ctx.gen = model, model._generation
        return mu, lv

    @staticmethod
    def backward(ctx, dmu, dlv):
        model = ctx.model
        model._check_generation(ctx.gen)
        model._alias_grads()
        model._engine_encoder_backward(dmu, dlv)
        return None, None, None


class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, z, model, training):
        model.params_changed()
        bp, ap = model._engine_decoder(z, training)
        ctx.model, ctx.gen = model, model._generation
        return bp, ap

    @staticmethod
    def backward(ctx, dbp, dap):
        model = ctx.model
        model._check_generation(ctx.gen)
        model._alias_grads()
        dz = model._engine_decoder_backward(dbp, dap)
        return None, dz, None, None


class Sg2ScVAEModel(nn.Module):
    def __init__(self, vocab, embedding_dim=128, batch_size=32, train_3d=True, decoder_cat=False, Nangle=24,
                 gconv_mode='feedforward', gconv_pooling='avg', gconv_num_layers=5, mlp_normalization='none',
                 vec_noise_dim=0, layout_noise_dim=0, use_AE=False, use_attr=True):
        super().__init__()
        if gconv_num_layers < 1:
            raise NotImplementedError("gconv_num_layers must be >= 1 on the HIP path")
        if embedding_dim % 16:
            raise NotImplementedError("embedding_dim must be a multiple of 16 on the HIP path")
        E = embedding_dim
        hidden = E * 4
        box_e, angle_e = int(E * 3 / 4), int(E / 4)
        obj_e, attr_e = (int(E * 3 / 4), int(E / 4)) if use_attr else (E, 0)          # Sg2ScVAE_model.py:23-24,35-37
        dc_dim = E * 2 if decoder_cat else E                                          # :47,51-52,79-88
        self.use_attr, self.batch_size, self.train_3d, self.decoder_cat = use_attr, batch_size, train_3d, decoder_cat
        self.vocab, self.vec_noise_dim, self.layout_noise_dim, self.use_AE = vocab, vec_noise_dim, layout_noise_dim, use_AE
        self.embedding_dim, self.Nangle, self.gconv_mode = E, Nangle, gconv_mode
        self.gconv_num_layers, self.mlp_normalization = gconv_num_layers, mlp_normalization
        num_objs, num_preds = len(vocab['object_idx_to_name']), len(vocab['pred_idx_to_name'])
        num_attrs = len(vocab['attrib_idx_to_name'])
        self.box_dim = 6 if train_3d else 4

        # registration order == reference (Sg2ScVAE_model.py:44-103): it is the parameter order
        self.obj_embeddings_ec = nn.Embedding(num_objs + 1, obj_e)
        self.pred_embeddings_ec = nn.Embedding(num_preds, E * 2)
        self.obj_embeddings_dc = nn.Embedding(num_objs + 1, obj_e)
        self.pred_embeddings_dc = nn.Embedding(num_preds, dc_dim)
        if use_attr:
            self.attr_embedding_ec = nn.Embedding(num_attrs, attr_e)
            self.attr_embedding_dc = nn.Embedding(num_attrs, attr_e)
        self.box_embeddings = nn.Linear(self.box_dim, box_e)
        self.angle_embeddings = nn.Embedding(Nangle, angle_e)
        n = mlp_normalization
        self.box_mean_var = make_mlp([E * 2, hidden, E * 2], batch_norm=n)
        self.box_mean = make_mlp([E * 2, box_e], batch_norm=n, norelu=True)
        self.box_var = make_mlp([E * 2, box_e], batch_norm=n, norelu=True)
        self.angle_mean_var = make_mlp([E * 2, hidden, E * 2], batch_norm=n)
        self.angle_mean = make_mlp([E * 2, angle_e], batch_norm=n, norelu=True)
        self.angle_var = make_mlp([E * 2, angle_e], batch_norm=n, norelu=True)
        kw = dict(hidden_dim=hidden, pooling=gconv_pooling, num_layers=gconv_num_layers, mode=gconv_mode,
                  mlp_normalization=n)
        self.gconv_net_ec = GraphTripleConvNet(input_dim=E * 2, **kw)
        self.gconv_net_dc = GraphTripleConvNet(input_dim=dc_dim, **kw)
        self.box_net = make_mlp([E * 2 + attr_e, hidden, self.box_dim], batch_norm=n, norelu=True)
        self.angle_net = make_mlp([E * 2, hidden, Nangle], batch_norm=n, norelu=True)
        for m in (self.box_embeddings, self.box_mean_var, self.box_mean, self.box_var, self.angle_mean_var,
                  self.angle_mean, self.angle_var, self.box_net):
            m.apply(_init_weights)

        self.validate_inputs = True      # one host sync per NEW batch; set False in a tuned training loop
        self._eng = None
        self._generation = 0
        self._batch_key = None
        self._flatten()

    # ------------------------------------------------------------------ flat parameter storage
    def _flatten(self):
        params = list(self.parameters())
        for p in params:
            if p.dtype != torch.float32:
                raise NotImplementedError("the HIP path is fp32 only (bit-for-tolerance parity with the reference)")
        dev = params[0].device
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        flat = torch.zeros(n, dtype=torch.float32, device=dev)
        # gradients + one guard element (data-parallel NaN guard, see grad_bucket) + padding to the alignment
        gfull = torch.zeros(n + _ALIGN, dtype=torch.float32, device=dev)
        gflat = gfull[:n]
        views = []
        with torch.no_grad():
            for p, o in zip(params, offs):
                v = flat[o:o + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
                gv = gflat[o:o + p.numel()].view(p.shape)
                if p.grad is not None:
                    gv.copy_(p.grad)
                p.grad = gv
                views.append(gv)
        # .cuda() / .float() re-flatten: the optimizer state moves with the parameters (it used to be dropped silently)
        old_m, old_v = getattr(self, "_adam_m", None), getattr(self, "_adam_v", None)
        steps = self._sync_adam_steps() if getattr(self, "_eng", None) is not None else getattr(self, "_adam_steps", 0)
        self._flat, self._gflat, self._gfull, self._gviews, self._params = flat, gflat, gfull, views, params
        ref = weakref.ref(self)
        for p in params:
            p._sln_owner = ref               # lets an unmodified torch.optim.Adam find the fused update (see _adam_step_pre_hook)
        self._offs = offs
        if old_m is not None and old_m.numel() == n:
            self._adam_m, self._adam_v, self._adam_steps = old_m.to(dev), old_v.to(dev), steps
        else:
            self._adam_m = self._adam_v = None
            self._adam_steps = 0                   # host copy of Adam's step counter (the device's lives in the engine's workspace)
        self._anchor = torch.zeros(1, dtype=torch.float32, device=dev, requires_grad=True)
        self._drop_engine()

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._flatten()
        return out

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)     # copies in place: the flat views stay valid
        if self._eng is not None:
            _lib.check(_lib.lib().sln_vae_params_changed(self._eng), "sln_vae_params_changed")
        return out

    @property
    def flat_params(self):
        return self._flat

    @property
    def flat_grads(self):
        return self._gflat

    @property
    def grad_bucket(self):
        """What the data-parallel trainer all-reduces: ``flat_grads`` plus ONE trailing guard element.  ``train_step(with_adam=
        False)`` / ``train_step_begin`` leave the rank's total loss there; after the all-reduce it is the rank average, and
        ``adam_step`` skips the update on EVERY rank when it is not finite (train.py:79-81 on a single GPU)."""
        return self._gfull[:self._gflat.numel() + 1]

    def _sync_adam_steps(self):
        """The device's step counter (it does not advance on an iteration skipped for a non-finite loss)."""
        if self._eng is not None:
            out = C.c_int64(0)
            _lib.check(_lib.lib().sln_vae_adam_get_step(self._eng, C.byref(out), _lib.current_stream_ptr()), "sln_vae_adam_get_step")
            self._adam_steps = int(out.value)
        return self._adam_steps

    def manual_seed(self, seed):
        """Seed of the on-device N(0,1) draws taken when no ``eps`` is passed (default: torch.initial_seed())."""
        self._rng_seed, self._rng_epoch = int(seed) & (2 ** 64 - 1), 0
        if self._eng is not None:
            _lib.check(_lib.lib().sln_vae_seed(self._eng, self._rng_seed, 0, _lib.current_stream_ptr()), "sln_vae_seed")
            # the live engine now consumes the offsets of epoch 0: an engine re-created later (a larger batch arrives) must start
            # on the NEXT disjoint range, not replay them
            self._rng_epoch = 1

    def last_eps(self):
        """eps of the last forward / train_step (injected or drawn on the device)."""
        out = self._new(self._O, self.embedding_dim)
        _lib.check(_lib.lib().sln_vae_last_eps(self._eng, _lib.ptr(out), _lib.current_stream_ptr()), "sln_vae_last_eps")
        return out

    def device_randn(self, rows, cols):
        """[rows, cols] ~ N(0,1) drawn ON THE DEVICE from the engine's Philox stream (``manual_seed``): the z draws of posterior
        sampling (host/sampling.py) - no host generator, no host-to-device copy of the sample."""
        if self._eng is None:
            self._ensure_engine(1, 1)
        out = self._new(int(rows), int(cols))
        _lib.check(_lib.lib().sln_vae_randn(self._eng, _lib.ptr(out), out.numel(), _lib.current_stream_ptr()), "sln_vae_randn")
        return out

    def params_changed(self):
        """Call after modifying parameters outside the engine (e.g. a torch optimizer step)."""
        if self._eng is not None:
            _lib.check(_lib.lib().sln_vae_params_changed(self._eng), "sln_vae_params_changed")

    def _alias_grads(self):
        """optimizer.zero_grad(set_to_none=True) (torch's default) detaches every p.grad from the flat buffer: re-attach the views,
        zeroed.  When ALL of them are detached - every step of the unchanged train.py loop - that is ONE fill of the flat buffer
        (round 3 zeroed the 230 views one launch at a time: ~1 ms of host time per step, most of what the loop cost on top of the
        fused step)."""
        params, views = self._params, self._gviews
        stale = []
        for i in range(len(params)):
            g = params[i].grad
            if g is views[i]:
                continue
            if g is None or g.data_ptr() != views[i].data_ptr():
                stale.append(i)
        if not stale:
            return
        if len(stale) == len(params):
            self._gflat.zero_()
        else:
            for i in stale:
                views[i].zero_()
        for i in stale:
            params[i].grad = views[i]

    # ------------------------------------------------------------------ engine plumbing
    def _drop_engine(self):
        if getattr(self, "_eng", None) is not None:
            try:
                self._sync_adam_steps()            # the counter lives in the workspace that goes away with the engine
            except Exception:
                pass
            _lib.lib().sln_vae_destroy(self._eng)
        self._eng = None
        self._maxO = self._maxT = 0
        self._batch_key = None

    def __del__(self):
        try:
            self._drop_engine()
        except Exception:
            pass

    def _config(self):
        c = _lib.SlnVaeConfig()
        c.embedding_dim, c.gconv_num_layers = self.embedding_dim, self.gconv_num_layers
        c.recurrent = int(self.gconv_mode == 'recurrent')
        c.batch_norm = int(self.mlp_normalization == 'batch')
        c.decoder_cat, c.use_ae, c.box_dim, c.n_angle = int(self.decoder_cat), int(self.use_AE), self.box_dim, self.Nangle
        c.num_objs = self.obj_embeddings_ec.num_embeddings
        c.num_preds = self.pred_embeddings_ec.num_embeddings
        c.num_attrs = self.attr_embedding_ec.num_embeddings if self.use_attr else 0
        c.no_attr = int(not self.use_attr)
        return c

    def _unit_modules(self):
        seqs = [self.box_mean_var, self.box_mean, self.box_var, self.angle_mean_var, self.angle_mean, self.angle_var]
        for net in (self.gconv_net_ec, self.gconv_net_dc):
            for gc in net.gconvs:
                seqs += [gc.net1, gc.net2]
        seqs += [self.box_net, self.angle_net]
        out = []
        for s in seqs:
            out += mlp_linears(s)
        return out

    def _tensor_table(self, cfg, pbase, gbase):
        """SlnVaeTensors (+ the unit array it points to) for a parameter / gradient buffer pair laid out like this model's flat
        buffers and starting at device addresses ``pbase`` / ``gbase``: the model's own (``_ensure_engine``) or a per-room COPY
        (``room_engine``: layout refinement fine-tunes one copy of the checkpoint per room).  BatchNorm running statistics are
        buffers, not parameters: every table points at the model's own (eval-mode engines only read them)."""
        L = _lib.lib()
        p0, g0 = self._flat.data_ptr(), self._gflat.data_ptr()
        lay = getattr(self, "_table_layout", None)
        if lay is not None and lay["bufs"] != tuple(b.data_ptr() for bn in lay["bns"] for b in (bn.running_mean, bn.running_var, bn.num_batches_tracked)):
            lay = None                                # a BatchNorm buffer was re-registered (they live outside the flat buffers)
        if lay is None or lay["key"] != (p0, g0):
            # byte offsets of every tensor inside the flat buffers (and the addresses of the BatchNorm buffers): worked out once, the
            # tables of the R per-room engines of a refinement batch are then filled from them (0.56 -> ~0.1 ms per engine)
            units = self._unit_modules()
            n_units = L.sln_vae_num_units(C.byref(cfg))
            assert n_units == len(units), (n_units, len(units))
            goff = {id(p): gv.data_ptr() - g0 for p, gv in zip(self._params, self._gviews)}
            po = lambda p: p.data_ptr() - p0
            urows = []
            for lin, bn in units:
                row = dict(weight=po(lin.weight), bias=po(lin.bias), d_weight=goff[id(lin.weight)], d_bias=goff[id(lin.bias)])
                if bn is not None:
                    row.update(bn_weight=po(bn.weight), bn_bias=po(bn.bias), d_bn_weight=goff[id(bn.weight)], d_bn_bias=goff[id(bn.bias)],
                               _abs=dict(bn_running_mean=bn.running_mean.data_ptr(), bn_running_var=bn.running_var.data_ptr(),
                                         bn_num_batches_tracked=bn.num_batches_tracked.data_ptr()))
                urows.append(row)
            embs = dict(obj_emb_ec=self.obj_embeddings_ec.weight, pred_emb_ec=self.pred_embeddings_ec.weight,
                        obj_emb_dc=self.obj_embeddings_dc.weight, pred_emb_dc=self.pred_embeddings_dc.weight,
                        box_emb_w=self.box_embeddings.weight, box_emb_b=self.box_embeddings.bias,
                        angle_emb=self.angle_embeddings.weight)
            if self.use_attr:
                embs.update(attr_emb_ec=self.attr_embedding_ec.weight, attr_emb_dc=self.attr_embedding_dc.weight)
            bufs = tuple(b.data_ptr() for _, bn in units if bn is not None for b in (bn.running_mean, bn.running_var, bn.num_batches_tracked))
            lay = self._table_layout = dict(key=(p0, g0), units=urows, embs={k: (po(p), goff[id(p)]) for k, p in embs.items()}, bufs=bufs,
                                            bns=[bn for _, bn in units if bn is not None])
        arr = (_lib.SlnVaeUnit * len(lay["units"]))()
        for u, row in zip(arr, lay["units"]):
            for k, off in row.items():
                if k == "_abs":
                    for kk, ptr in off.items():
                        setattr(u, kk, ptr)
                elif k.startswith("d_"):
                    setattr(u, k, gbase + off)
                else:
                    setattr(u, k, pbase + off)
        t = _lib.SlnVaeTensors()
        for k, (po_, go_) in lay["embs"].items():
            setattr(t, k, pbase + po_)
            setattr(t, "d_" + k, gbase + go_)
        t.units_host = arr
        t.flat_params, t.flat_grads, t.n_flat = pbase, gbase, self._flat.numel()
        return t, arr

    def room_engines(self, params, grads, max_objs, max_triples):
        """One engine per row of ``params`` / ``grads`` ([R, n_flat] copies of ``flat_params`` / zeroed gradients, rows 16-byte
        aligned): -> list of (handle, workspace, keep-alive).  The caller owns the handles (``sln_vae_destroy``)."""
        if self._flat.device.type != 'cuda':
            raise _lib.SlnError("Sg2ScVAEModel runs on the MI355X only: call model.cuda() first (no CPU fallback)")
        L = _lib.lib()
        cfg = self._config()
        maxO, maxT = max(int(max_objs), 64), max(int(max_triples), 64)
        out = []
        for r in range(params.shape[0]):
            h = C.c_void_p()
            _lib.check(L.sln_vae_create(C.byref(cfg), C.byref(h)), "sln_vae_create")
            nbytes = L.sln_vae_workspace_bytes(h, maxO, maxT)
            if nbytes < 0:
                _lib.check(int(nbytes), "sln_vae_workspace_bytes")
            out.append([h, torch.zeros(int(nbytes), dtype=torch.uint8, device=self._flat.device), None, int(nbytes)])
        torch.cuda.current_stream(self._flat.device).synchronize()         # sln_vae_bind writes into the workspaces with blocking copies
        for r, e in enumerate(out):
            t, arr = self._tensor_table(cfg, params[r].data_ptr(), grads[r].data_ptr())
            e[2] = arr
            _lib.check(L.sln_vae_bind(e[0], C.byref(t), C.c_void_p(e[1].data_ptr()), e[3], maxO, maxT), "sln_vae_bind")
        return [(e[0], e[1], e[2]) for e in out]

    def decoder_param_ranges(self):
        """(offset, length) runs of ``flat_params`` that hold every parameter the DECODER reads: the *_dc embedding tables
        (registered in front, between the encoder's) and the trailing gconv_net_dc / box_net / angle_net run."""
        spans, o = [], 0
        for name, p in self.named_parameters():
            n = (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
            if name.startswith(self._DECODER_ONLY) or name.startswith(("obj_embeddings_dc.", "pred_embeddings_dc.", "attr_embedding_dc.")):
                if spans and spans[-1][0] + spans[-1][1] == o:
                    spans[-1][1] += n
                else:
                    spans.append([o, n])
            o += n
        return [(a, b) for a, b in spans]

    def _ensure_engine(self, O, T):
        if self._flat.device.type != 'cuda':
            raise _lib.SlnError("Sg2ScVAEModel runs on the MI355X only: call model.cuda() first (no CPU fallback)")
        L = _lib.lib()
        if self._eng is not None and O <= self._maxO and T <= self._maxT:
            return
        self._drop_engine()
        cfg = self._config()
        h = C.c_void_p()
        _lib.check(L.sln_vae_create(C.byref(cfg), C.byref(h)), "sln_vae_create")
        self._eng = h
        maxO, maxT = max(O, 64), max(T, 64)
        nbytes = L.sln_vae_workspace_bytes(h, maxO, maxT)
        if nbytes < 0:
            _lib.check(int(nbytes), "sln_vae_workspace_bytes")
        dev = self._flat.device
        self._ws = torch.zeros(int(nbytes), dtype=torch.uint8, device=dev)
        # sln_vae_bind writes Adam's constants into the workspace with a BLOCKING copy (legacy stream); the zero-fill above is
        # asynchronous on the current (non-blocking) stream and, with work queued in front of it, would land AFTER that copy and
        # wipe beta1 / beta2 / eps (seen when a larger batch re-created the engine in the middle of training: NaN two steps later)
        torch.cuda.current_stream(dev).synchronize()
        if self._adam_m is None:
            self._adam_m = torch.zeros_like(self._flat)
            self._adam_v = torch.zeros_like(self._flat)
        t, arr = self._tensor_table(cfg, self._flat.data_ptr(), self._gflat.data_ptr())
        t.adam_m, t.adam_v = self._adam_m.data_ptr(), self._adam_v.data_ptr()
        self._units_keepalive = arr
        _lib.check(L.sln_vae_bind(h, C.byref(t), C.c_void_p(self._ws.data_ptr()), int(nbytes), maxO, maxT), "sln_vae_bind")
        if self._adam_steps:
            # a larger batch re-creates the engine (bigger workspace); the moments live in this module, the bias-correction step
            # in the workspace - without this the update after a re-bind would be scaled as if it were the first one
            _lib.check(L.sln_vae_adam_reset(h, int(self._adam_steps), _lib.current_stream_ptr()), "sln_vae_adam_reset")
        _lib.check(L.sln_vae_set_grad_guard(h, C.c_void_p(self._gfull.data_ptr() + 4 * self._gflat.numel())), "sln_vae_set_grad_guard")
        if getattr(self, "_rng_seed", None) is None:
            self._rng_seed, self._rng_epoch = int(torch.initial_seed()) & (2 ** 64 - 1), 0
        # a re-created engine (larger batch) continues on a disjoint range of Philox offsets
        _lib.check(L.sln_vae_seed(h, self._rng_seed, self._rng_epoch << 40, _lib.current_stream_ptr()), "sln_vae_seed")
        self._rng_epoch += 1
        self._maxO, self._maxT = maxO, maxT
        self._batch_key = None

    def _set_batch(self, objs, triples, boxes, angles, attributes):
        O, T = int(objs.shape[0]), int(triples.shape[0])
        self._ensure_engine(O, T)
        dev = self._flat.device

        def prep(x, dt):
            if x.device != dev:
                raise _lib.SlnError("inputs must live on the model's GPU")
            return x.to(dt).contiguous()
        objs, triples, attributes = prep(objs, torch.int64), prep(triples, torch.int64), prep(attributes, torch.int64)
        if boxes is None or angles is None:
            # decoder-only calls (no ground truth): ONE pair of zero tensors per size, so that repeated calls on the same graph
            # (the refinement loop: 60 decoder calls per room) keep the same batch key and skip the re-binding
            zc = getattr(self, "_zero_inputs", None)
            if zc is None or zc[0] != (O, dev):
                zc = ((O, dev), torch.zeros(O, self.box_dim, device=dev), torch.zeros(O, dtype=torch.int64, device=dev))
                self._zero_inputs = zc
        boxes = prep(boxes, torch.float32) if boxes is not None else zc[1]
        angles = prep(angles, torch.int64) if angles is not None else zc[2]
        key = tuple((x.data_ptr(), tuple(x.shape), x._version) for x in (objs, triples, boxes, angles, attributes))
        if key == self._batch_key:
            return
        b = _lib.SlnVaeBatch()
        b.objs, b.triples, b.boxes = objs.data_ptr(), triples.data_ptr(), boxes.data_ptr()
        b.angles, b.attributes, b.O, b.T = angles.data_ptr(), attributes.data_ptr(), O, T
        _lib.check(_lib.lib().sln_vae_set_batch(self._eng, C.byref(b), _lib.current_stream_ptr()), "sln_vae_set_batch")
        self._batch_refs = (objs, triples, boxes, angles, attributes)
        self._batch_key, self._O, self._T = key, O, T
        if self.validate_inputs:
            rc = _lib.lib().sln_vae_check_batch(self._eng, _lib.current_stream_ptr())
            if rc == -1:
                self._batch_key = None
                raise IndexError("scene-graph batch holds an out-of-range object class / predicate / attribute / angle / row id")
            _lib.check(rc, "sln_vae_check_batch")

    def _new(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=self._flat.device)

    def _check_generation(self, gen):
        if gen != self._generation:
            raise _lib.SlnError("backward() through a stale forward: another forward ran on this model in between")

    def _engine_forward(self, eps, training):
        O, E = self._O, self.embedding_dim
        mu, lv, z = self._new(O, E), self._new(O, E), self._new(O, E)
        bp, ap = self._new(O, self.box_dim), self._new(O, self.Nangle)
        self._generation += 1
        _lib.check(_lib.lib().sln_vae_forward(self._eng, _lib.ptr(eps), _lib.ptr(mu), _lib.ptr(lv), _lib.ptr(z), _lib.ptr(bp),
                                              _lib.ptr(ap), int(training), _lib.current_stream_ptr()), "sln_vae_forward")
        return mu, lv, z, bp, ap

    def _engine_encoder(self, training):
        O, E = self._O, self.embedding_dim
        mu, lv = self._new(O, E), self._new(O, E)
        self._generation += 1
        _lib.check(_lib.lib().sln_vae_encoder(self._eng, _lib.ptr(mu), _lib.ptr(lv), int(training),
                                              _lib.current_stream_ptr()), "sln_vae_encoder")
        return mu, lv

    def _engine_decoder(self, z, training):
        O = self._O
        z = z.detach().to(torch.float32).contiguous()
        bp, ap = self._new(O, self.box_dim), self._new(O, self.Nangle)
        self._generation += 1
        _lib.check(_lib.lib().sln_vae_decoder(self._eng, _lib.ptr(z), _lib.ptr(bp), _lib.ptr(ap), int(training),
                                              _lib.current_stream_ptr()), "sln_vae_decoder")
        return bp, ap

    def _engine_decoder_backward(self, dbp, dap):
        dz = self._new(self._O, self.embedding_dim)
        dbp = None if dbp is None else dbp.to(torch.float32).contiguous()
        dap = None if dap is None else dap.to(torch.float32).contiguous()
        _lib.check(_lib.lib().sln_vae_decoder_backward(self._eng, _lib.ptr(dbp), _lib.ptr(dap), _lib.ptr(dz),
                                                       _lib.current_stream_ptr()), "sln_vae_decoder_backward")
        return dz

    def _engine_encoder_backward(self, dmu, dlv):
        dmu = None if dmu is None else dmu.to(torch.float32).contiguous()
        dlv = None if dlv is None else dlv.to(torch.float32).contiguous()
        _lib.check(_lib.lib().sln_vae_encoder_backward(self._eng, _lib.ptr(dmu), _lib.ptr(dlv), _lib.current_stream_ptr()),
                   "sln_vae_encoder_backward")

    # ------------------------------------------------------------------ reference call surfaces
    def encoder(self, objs, triples, boxes_gt, angles_gt, attributes):
        self._set_batch(objs, triples, boxes_gt, angles_gt, attributes)
        if torch.is_grad_enabled():
            return _EncoderFn.apply(self._anchor, self, self.training)
        return self._engine_encoder(self.training)

    def decoder(self, z, objs, triples, attributes):
        boxes, angles = (self._batch_refs[2], self._batch_refs[3]) if self._batch_key is not None and \
            self._batch_refs[0].shape[0] == objs.shape[0] else (None, None)
        self._set_batch(objs, triples, boxes, angles, attributes)
        if torch.is_grad_enabled():
            return _DecoderFn.apply(self._anchor, z, self, self.training)
        return self._engine_decoder(z, self.training)

    def forward(self, objs, triples, boxes_gt, angles_gt, attributes, obj_to_img=None, eps=None):
        """Returns (mu, logvar, boxes_pred, angles_pred).  ``eps`` (optional, [O, embedding_dim]) pins the
        N(0,1) draw that the reference takes with torch.randn_like (Sg2ScVAE_model.py:182)."""
        self._set_batch(objs, triples, boxes_gt, angles_gt, attributes)
        if eps is not None:
            eps = eps.to(torch.float32).contiguous()       # None: drawn on the device inside sln_vae_forward
        if torch.is_grad_enabled():
            return _ForwardFn.apply(self._anchor, self, eps, self.training)
        mu, lv, z, bp, ap = self._engine_forward(eps, self.training)
        return mu, lv, bp, ap

    # ------------------------------------------------------------------ fused training iteration
    def train_step(self, objs, triples, boxes, angles, attributes, kl_weight=0.1, lr=1e-4, eps=None,
                   use_graph=True, with_adam=True):
        """train.py:62-84 in one call: zero_grad, forward (BatchNorm in the module's mode: ``model.eval()`` keeps training on the
        running statistics, train.py:63-65), losses, backward, Adam.

        Returns a 4-element device tensor [bbox_pred, angle_pred, KLD_Gauss*w, total_loss] (no host sync).
        ``with_adam=False`` stops after backward (gradients in ``flat_grads``): the data-parallel
        trainer all-reduces them and then calls ``adam_step``.  hipGraph replay needs a non-default
        current stream.
        """
        self._set_batch(objs, triples, boxes, angles, attributes)
        if eps is not None:
            eps = eps.to(torch.float32).contiguous()       # None: the iteration draws N(0,1) on the device (also under graph replay)
        losses = self._new(4)
        self._generation += 1
        _lib.check(_lib.lib().sln_vae_set_training(self._eng, int(self.training)), "sln_vae_set_training")     # train.py:63-65
        _lib.check(_lib.lib().sln_vae_train_step(
            self._eng, _lib.ptr(eps), float(kl_weight), float(lr), _lib.ptr(losses), int(use_graph), int(with_adam),
            _lib.current_stream_ptr()), "sln_vae_train_step")
        self._adam_steps += int(bool(with_adam))
        self._alias_grads()
        return losses

    # -- the same iteration in two halves, for the data-parallel trainer (host/train.py::DataParallelStep) ------------
    _DECODER_ONLY = ("gconv_net_dc.", "box_net.", "angle_net.")

    @property
    def decoder_grad_offset(self):
        """First element of ``flat_grads`` that belongs to the trailing run of decoder-only parameters: everything from
        here on is final when ``train_step_begin`` has been executed."""
        off, o = [], 0
        for name, p in self.named_parameters():
            off.append((name, o))
            o += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        split = o
        for name, start in reversed(off):
            if not name.startswith(self._DECODER_ONLY):
                break
            split = start
        return split

    def train_step_begin(self, objs, triples, boxes, angles, attributes, kl_weight=0.1, lr=1e-4, eps=None, use_graph=True):
        """zero_grad, forward, losses and the decoder's half of backward (SLN_TRAIN_UPTO_DECODER); returns the losses."""
        self._set_batch(objs, triples, boxes, angles, attributes)
        if eps is not None:
            eps = eps.to(torch.float32).contiguous()
        losses = self._new(4)
        self._generation += 1
        _lib.check(_lib.lib().sln_vae_set_training(self._eng, int(self.training)), "sln_vae_set_training")
        _lib.check(_lib.lib().sln_vae_train_step(
            self._eng, _lib.ptr(eps), float(kl_weight), float(lr), _lib.ptr(losses), int(use_graph), 2,
            _lib.current_stream_ptr()), "sln_vae_train_step(begin)")
        return losses

    def train_step_finish(self, use_graph=True):
        """The encoder's half of backward (SLN_TRAIN_ENCODER_BWD) of the iteration ``train_step_begin`` started."""
        _lib.check(_lib.lib().sln_vae_train_step(
            self._eng, None, 0.0, 0.0, None, int(use_graph), 3, _lib.current_stream_ptr()), "sln_vae_train_step(finish)")
        self._alias_grads()

    def adam_step(self, lr=1e-4):
        """torch.optim.Adam(lr).step() over the flat parameter buffer (fused kernel)."""
        if self._eng is None:
            # a rank whose shard was empty on its very first step (short last batch, batch_size < world) updates with the reduced
            # gradients like every other rank: the optimizer only needs the flat buffers, not a bound batch
            self._ensure_engine(0, 0)
        _lib.check(_lib.lib().sln_vae_adam_step(self._eng, float(lr), _lib.current_stream_ptr()), "sln_vae_adam_step")
        self._adam_steps += 1

    def fused_adam(self, lr=1e-4):
        """The one-line swap for train.py:15 - ``optimizer = model.fused_adam(lr=args.learning_rate)`` instead of
        ``torch.optim.Adam(model.parameters(), lr=...)``: same update rule (default betas / eps, no weight decay), same
        ``zero_grad()`` / ``step()`` / ``state_dict()`` / ``load_state_dict()`` surface, but ``step()`` is ONE kernel over the flat
        parameter buffer instead of torch's multi-tensor passes over 230 tensors, and ``zero_grad()`` one fill of the flat gradient
        buffer (the parameters' ``.grad`` stay views of it)."""
        return FusedAdam(self, lr)

    # -- optimizer state in torch.optim.Adam's own format (train.py:94 saves optimizer.state_dict(), :25 restores it) ---------
    def optim_state_dict(self, lr=1e-4):
        """What ``torch.optim.Adam(model.parameters(), lr).state_dict()`` would hold after the same steps: per-parameter
        ``step`` / ``exp_avg`` / ``exp_avg_sq`` in ``model.parameters()`` order - a reference checkpoint's ``optim_state`` and this
        one are interchangeable."""
        state = {}
        if self._adam_m is not None and self._sync_adam_steps() > 0:
            for i, (p, o) in enumerate(zip(self._params, self._offs)):
                n = p.numel()
                state[i] = {'step': torch.tensor(float(self._adam_steps)),
                            'exp_avg': self._adam_m[o:o + n].view(p.shape).clone(), 'exp_avg_sq': self._adam_v[o:o + n].view(p.shape).clone()}
        group = {'lr': lr, 'betas': (0.9, 0.999), 'eps': 1e-08, 'weight_decay': 0, 'amsgrad': False, 'maximize': False, 'foreach': None,
                 'capturable': False, 'differentiable': False, 'fused': None, 'params': list(range(len(self._params)))}
        return {'state': state, 'param_groups': [group]}

    def load_optim_state_dict(self, sd):
        """Restore the fused Adam from ``optim_state_dict()`` output or from a ``torch.optim.Adam`` state_dict of this model."""
        g = sd['param_groups'][0]
        if tuple(g.get('betas', (0.9, 0.999))) != (0.9, 0.999) or g.get('weight_decay', 0) != 0 or g.get('amsgrad', False):
            raise NotImplementedError("the fused Adam implements the reference's configuration (train.py:15): default betas, no weight decay")
        if self._adam_m is None:
            self._adam_m = torch.zeros_like(self._flat)
            self._adam_v = torch.zeros_like(self._flat)
            self._drop_engine()                                    # the engine binds the moment buffers at creation
        self._adam_m.zero_(); self._adam_v.zero_()
        steps = 0
        with torch.no_grad():
            for i, (p, o) in enumerate(zip(self._params, self._offs)):
                st = sd['state'].get(i)
                if st is None:
                    continue
                n = p.numel()
                self._adam_m[o:o + n].copy_(st['exp_avg'].reshape(-1).to(self._flat.device, torch.float32))
                self._adam_v[o:o + n].copy_(st['exp_avg_sq'].reshape(-1).to(self._flat.device, torch.float32))
                steps = max(steps, int(float(st['step'])))
        self._adam_steps = steps
        if self._eng is not None:
            _lib.check(_lib.lib().sln_vae_adam_reset(self._eng, steps, _lib.current_stream_ptr()), "sln_vae_adam_reset")

    def loss(self, kl_weight=0.1, with_grads=False):
        """calculate_model_losses (utils.py:12-33) on the outputs of the last forward, on device."""
        losses = self._new(4)
        _lib.check(_lib.lib().sln_vae_loss(self._eng, None, None, None, None, float(kl_weight), _lib.ptr(losses),
                                           int(with_grads), _lib.current_stream_ptr()), "sln_vae_loss")
        return losses

    def tap(self, layer, what):
        """Debug: copy of an internal pre-activation (see sln_vae_tap)."""
        H, D = self.embedding_dim * 4, self.embedding_dim * 2
        if layer >= self.gconv_num_layers and not self.decoder_cat:
            D = self.embedding_dim
        shape = {0: (self._T, H), 1: (self._T, 2 * H + D), 2: (self._O, H), 3: (self._O, H), 4: (self._O, D)}[what]
        out = self._new(*shape)
        n = _lib.lib().sln_vae_tap(self._eng, layer, what, _lib.ptr(out), _lib.current_stream_ptr())
        if n < 0:
            _lib.check(int(n), "sln_vae_tap")
        return out


class FusedAdam:
    """``torch.optim.Adam(model.parameters(), lr)`` for a ``Sg2ScVAEModel``, on the engine's fused kernel (see ``fused_adam``)."""

    def __init__(self, model, lr=1e-4):
        self.model = model
        self.param_groups = [{'params': list(model.parameters()), 'lr': lr, 'betas': (0.9, 0.999), 'eps': 1e-08, 'weight_decay': 0,
                              'amsgrad': False}]

    def zero_grad(self, set_to_none=False):
        m = self.model
        m._alias_grads()                                   # .grad of every parameter is (again) its view of the flat buffer
        m._gfull.zero_()                                   # gradients + the guard slot: one fill

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        m = self.model
        if m._eng is None:
            raise _lib.SlnError("FusedAdam.step() before the first forward / backward of the model")
        m.adam_step(lr=float(self.param_groups[0]['lr']))
        return loss

    def state_dict(self):
        return self.model.optim_state_dict(self.param_groups[0]['lr'])

    def load_state_dict(self, sd):
        self.model.load_optim_state_dict(sd)
        self.param_groups[0]['lr'] = sd['param_groups'][0].get('lr', self.param_groups[0]['lr'])


# ---------------------------------------------------------------------------------------------------------------------------
# train.py:15 builds ``torch.optim.Adam(model.parameters(), lr=args.learning_rate)`` and calls ``optimizer.step()`` (:84).  On 230
# parameter tensors that step is ~1.7 ms of host work (state look-ups, a dozen multi-tensor launches with their descriptor
# tables) next to a 1.9 ms training iteration - although all 230 tensors are views of ONE flat buffer that the engine's fused
# Adam kernel updates in 20 us.  Two process-wide optimizer hooks route such a step: when the optimizer is a plain
# ``torch.optim.Adam`` over exactly ``model.parameters()`` with the reference's hyper-parameters (default betas / eps, no weight
# decay, no amsgrad / maximize / capturable / fused / differentiable, no closure) and every parameter holds its gradient view,
# the pre-hook runs ``sln_vae_adam_step`` and hands torch an empty parameter list for the duration of its own ``step``; the
# post-hook puts the list back.  ``optimizer.state`` holds VIEWS of the fused moment buffers, so ``optimizer.state_dict()`` /
# ``load_state_dict()`` keep torch's layout (train.py:25,94) without conversions; anything else - other hyper-parameters, a
# parameter subset, a missing gradient - takes torch's own path untouched.  ``Sg2ScVAEModel.route_torch_adam = False`` switches
# the routing off.
def _fast_adam_owner(opt):
    if type(opt) is not torch.optim.Adam or len(opt.param_groups) != 1:
        return None
    g = opt.param_groups[0]
    ps = g['params']
    cached = opt.__dict__.get('_sln_owner_ref')
    if cached is None:
        ref = getattr(ps[0], "_sln_owner", None) if ps else None
        m = ref() if ref is not None else None
        ok = m is not None and len(ps) == len(m._params) and all(a is b for a, b in zip(ps, m._params))
        opt.__dict__['_sln_owner_ref'] = (ref, len(ps)) if ok else (None, -1)
        cached = opt.__dict__['_sln_owner_ref']
    ref, n = cached
    m = ref() if ref is not None else None
    if m is None or n != len(ps) or not m.route_torch_adam or ps[0] is not m._params[0]:
        return None
    if tuple(g['betas']) != (0.9, 0.999) or g['eps'] != 1e-8 or g['weight_decay'] != 0 or g['amsgrad'] or g.get('maximize') or \
            g.get('capturable') or g.get('differentiable') or g.get('fused') or torch.is_tensor(g['lr']):
        return None
    return m


def _fast_adam_views(opt, m):
    """optimizer.state <-> the fused moment buffers.  A fresh optimizer (no state) starts from zero moments, like torch's; state
    that torch owns (its own earlier steps, ``load_state_dict``) is copied in once; afterwards the entries are views."""
    p0 = m._params[0]
    st0 = opt.state.get(p0)
    if st0 is not None and m._adam_m is not None and st0['exp_avg'].data_ptr() == m._adam_m.data_ptr() + 4 * m._offs[0] \
            and opt.__dict__.get('_sln_views_of') == m._adam_m.data_ptr():
        return
    g = opt.param_groups[0]
    # (re-flatten / .cuda() after routed steps: the per-parameter `step` tensors are only refreshed lazily - bring them up to the
    # device's count BEFORE they are read back as "foreign" state, or the rebuilt moments would restart their bias correction)
    if opt.__dict__.get('_sln_views_of') is not None:
        _fast_adam_materialize_steps(opt)
    foreign = {i: opt.state[p] for i, p in enumerate(m._params) if p in opt.state and 'exp_avg' in opt.state[p]}
    m.load_optim_state_dict({'state': foreign, 'param_groups': [dict(g, params=list(range(len(m._params))))]})   # empty: zero moments, step 0
    if m._eng is None:
        raise _lib.SlnError("optimizer.step() before the first forward / backward of the model")
    for p, o in zip(m._params, m._offs):
        n = p.numel()
        opt.state[p] = {'step': torch.tensor(float(m._adam_steps)), 'exp_avg': m._adam_m[o:o + n].view(p.shape),
                        'exp_avg_sq': m._adam_v[o:o + n].view(p.shape)}
    opt.__dict__['_sln_views_of'] = m._adam_m.data_ptr()
    if not opt.__dict__.get('_sln_sd_hook'):
        opt.register_state_dict_pre_hook(_fast_adam_materialize_steps)
        # optimizer.load_state_dict replaces the state with torch-owned tensors: a stale flag left from routed steps must not let
        # the next state_dict() overwrite the LOADED step counts with the device's count from before the load
        opt.register_load_state_dict_post_hook(_fast_adam_loaded)
        opt.__dict__['_sln_sd_hook'] = True


def _fast_adam_loaded(opt):
    opt.__dict__['_sln_steps_stale'] = False
    opt.__dict__['_sln_views_of'] = None                 # the entries are torch's again: the next routed step copies them in


def _fast_adam_materialize_steps(opt):
    """The per-parameter ``step`` tensors are refreshed when somebody is about to read them (state_dict, a step on torch's path)."""
    if not opt.__dict__.get('_sln_steps_stale'):
        return
    m = _fast_adam_owner(opt)
    ref = opt.__dict__.get('_sln_owner_ref', (None, 0))[0]
    m = m if m is not None else (ref() if ref is not None else None)
    if m is not None:
        n = float(m._sync_adam_steps())                 # the device's count: a step skipped for a non-finite loss does not advance it
        for p in m._params:
            st = opt.state.get(p)
            if st is not None:
                st['step'].fill_(n)
    opt.__dict__['_sln_steps_stale'] = False


def _adam_step_pre_hook(opt, args, kwargs):
    if type(opt) is not torch.optim.Adam:
        return None
    m = _fast_adam_owner(opt)
    # (torch hands the hook step()'s own argument tuple: args[0] is the optimizer, a closure would be args[1] or a keyword)
    closure = kwargs.get('closure') if len(args) < 2 else args[1]
    routed = m is not None and closure is None and m._flat.device.type == 'cuda' and m._eng is not None
    if routed:
        for p, gv in zip(m._params, m._gviews):          # torch skips parameters without a gradient: only the all-views case is routed
            gr = p.grad
            if gr is not gv and (gr is None or gr.data_ptr() != gv.data_ptr()):
                routed = False
                break
    if not routed:
        if opt.__dict__.get('_sln_views_of') is not None:
            _fast_adam_materialize_steps(opt)            # torch's own path continues on the shared moments with the right counters
        return None
    _fast_adam_views(opt, m)
    m.adam_step(lr=float(opt.param_groups[0]['lr']))
    opt.__dict__['_sln_saved_params'] = opt.param_groups[0]['params']
    opt.param_groups[0]['params'] = []                   # torch's step() now has nothing to do
    opt.__dict__['_sln_steps_stale'] = True
    return None


def _adam_step_post_hook(opt, args, kwargs):
    saved = opt.__dict__.pop('_sln_saved_params', None)
    if saved is not None:
        opt.param_groups[0]['params'] = saved
    return None


Sg2ScVAEModel.route_torch_adam = True
if not getattr(torch.optim.Optimizer, "_sln_hooks_installed", False):
    from torch.optim.optimizer import register_optimizer_step_post_hook, register_optimizer_step_pre_hook
    register_optimizer_step_pre_hook(_adam_step_pre_hook)
    register_optimizer_step_post_hook(_adam_step_post_hook)
    torch.optim.Optimizer._sln_hooks_installed = True
