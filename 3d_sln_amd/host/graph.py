"""Parameter containers with the reference's names for the graph-convolution blocks.

Mirrors models/graph.py of the reference: ``make_mlp`` (:10-27), ``_init_weights`` (:30-33),
``GraphTripleConv`` (:36-111), ``GraphTripleConvNet`` (:114-143).  The modules own the
parameters (so ``state_dict`` keys are the reference's: ``gconvs.{i}.net1.{0,1,3,4}.*``);
the arithmetic runs in the HIP engine (csrc/vae_engine.hip), which the owning
``Sg2ScVAEModel`` drives.  A GraphTripleConv(Net) used on its own also runs on the
engine's kernels through ``forward`` below, with autograd to both inputs and every parameter.
"""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib


def make_mlp(dim_list, activation='relu', batch_norm='none', dropout=0, norelu=False):
    """nn.Sequential of Linear -> [BatchNorm1d] -> ReLU blocks; ``norelu`` strips the tail
    activation (and its BatchNorm).  Index layout identical to the reference's."""
    if activation != 'relu' or dropout > 0:
        raise NotImplementedError("the HIP path implements activation='relu', dropout=0 "
                                  "(the only configuration the reference instantiates)")
    if batch_norm not in ('none', 'batch'):
        raise ValueError('unknown mlp normalization "%s"' % batch_norm)
    mods = []
    n = len(dim_list) - 1
    for i in range(n):
        mods.append(nn.Linear(dim_list[i], dim_list[i + 1]))
        tail = norelu and i == n - 1
        if batch_norm == 'batch' and not tail:
            mods.append(nn.BatchNorm1d(dim_list[i + 1]))
        if not tail:
            mods.append(nn.ReLU())
    return nn.Sequential(*mods)


def _init_weights(module):
    if isinstance(module, nn.Linear):
        nn.init.kaiming_normal_(module.weight)


def mlp_linears(seq):
    """[(Linear, BatchNorm1d | None), ...] of an MLP built by make_mlp."""
    out, mods = [], list(seq)
    for i, m in enumerate(mods):
        if isinstance(m, nn.Linear):
            bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm1d) else None
            out.append((m, bn))
    return out


class GraphTripleConv(nn.Module):
    """One triple convolution: parameters ``net1`` (3D -> H -> 2H+Dout) and ``net2`` (H -> H -> Dout)."""

    def __init__(self, input_dim, output_dim=None, hidden_dim=512, pooling='avg', mlp_normalization='none'):
        super().__init__()
        output_dim = input_dim if output_dim is None else output_dim
        if pooling != 'avg':
            raise AssertionError('Invalid pooling "%s"' % pooling)
        self.input_dim, self.output_dim, self.hidden_dim, self.pooling = input_dim, output_dim, hidden_dim, pooling
        self.net1 = make_mlp([3 * input_dim, hidden_dim, 2 * hidden_dim + output_dim], batch_norm=mlp_normalization)
        self.net2 = make_mlp([hidden_dim, hidden_dim, output_dim], batch_norm=mlp_normalization)
        self.net1.apply(_init_weights)
        self.net2.apply(_init_weights)

    def forward(self, obj_vecs, pred_vecs, edges):
        """(new_obj_vecs [O, Dout], new_pred_vecs [T, Dout]); differentiable (see _gconv_forward)."""
        return _gconv_forward([self], 1, obj_vecs, pred_vecs, edges, self.training, owner=self)


class GraphTripleConvNet(nn.Module):
    def __init__(self, input_dim, num_layers=5, hidden_dim=512, pooling='avg', mode='recurrent',
                 mlp_normalization='none'):
        super().__init__()
        if mode not in ('recurrent', 'feedforward'):
            raise ValueError('Invalid mode "%s"' % mode)
        self.num_layers, self.mode = num_layers, mode
        n_mod = 1 if mode == 'recurrent' else num_layers
        self.gconvs = nn.ModuleList([
            GraphTripleConv(input_dim=input_dim, hidden_dim=hidden_dim, pooling=pooling,
                            mlp_normalization=mlp_normalization) for _ in range(n_mod)])

    def forward(self, obj_vecs, pred_vecs, edges):
        return _gconv_forward(list(self.gconvs), self.num_layers, obj_vecs, pred_vecs, edges, self.training, owner=self)


class _GconvEngine:
    """Engine handle for a bare GraphTripleConv(Net) with autograd (sln_gconv_net_*): weights bound by pointer, parameter
    gradients accumulated by the kernels into buffers owned here (handed to ``p.grad`` after every backward)."""

    def __init__(self, modules, num_layers, device):
        m0 = modules[0]
        self.D, self.H, self.L, self.modules = m0.input_dim, m0.hidden_dim, num_layers, modules
        self.Dout = m0.output_dim
        self.recurrent = len(modules) == 1 and num_layers > 1
        pairs = [pr for m in modules for pr in (mlp_linears(m.net1) + mlp_linears(m.net2))]
        self.bn = any(bn is not None for _, bn in pairs)
        self.params = [p for m in modules for p in m.parameters()]
        self.key = tuple((p.data_ptr(), p._version) for p in self.params)
        self.gbuf = {id(p): torch.zeros_like(p) for p in self.params}
        L = _lib.lib()
        h = C.c_void_p()
        _lib.check(L.sln_gconv_net_create(self.D, self.H, self.Dout, num_layers, int(self.recurrent), int(self.bn), C.byref(h)), "sln_gconv_net_create")
        self.h = h
        self.units = (_lib.SlnVaeUnit * len(pairs))()
        for u, (lin, bn) in zip(self.units, pairs):
            u.weight, u.bias = lin.weight.data_ptr(), lin.bias.data_ptr()
            u.d_weight, u.d_bias = self.gbuf[id(lin.weight)].data_ptr(), self.gbuf[id(lin.bias)].data_ptr()
            if bn is not None:
                u.bn_weight, u.bn_bias = bn.weight.data_ptr(), bn.bias.data_ptr()
                u.bn_running_mean, u.bn_running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                u.bn_num_batches_tracked = bn.num_batches_tracked.data_ptr()
                u.d_bn_weight, u.d_bn_bias = self.gbuf[id(bn.weight)].data_ptr(), self.gbuf[id(bn.bias)].data_ptr()
        self.maxO = self.maxT = 0
        self.device = device

    def ensure(self, O, T):
        if O <= self.maxO and T <= self.maxT:
            return
        L = _lib.lib()
        maxO, maxT = max(O, 64, self.maxO), max(T, 64, self.maxT)
        nbytes = L.sln_vae_workspace_bytes(self.h, maxO, maxT)
        if nbytes < 0:
            _lib.check(int(nbytes), "sln_vae_workspace_bytes")
        self.ws = torch.zeros(int(nbytes), dtype=torch.uint8, device=self.device)
        torch.cuda.current_stream(self.device).synchronize()          # bind writes with blocking copies (see sln_vae_bind)
        t = _lib.SlnVaeTensors()
        t.units_host = self.units
        _lib.check(L.sln_vae_bind(self.h, C.byref(t), C.c_void_p(self.ws.data_ptr()), int(nbytes), maxO, maxT), "sln_vae_bind")
        self.maxO, self.maxT = maxO, maxT

    def __del__(self):
        try:
            _lib.lib().sln_vae_destroy(self.h)
        except Exception:
            pass


class _GconvNetFn(torch.autograd.Function):
    """(new_obj, new_pred) = net(obj_vecs, pred_vecs, edges) with gradients w.r.t. both inputs and every parameter.

    The parameters are INPUTS of the function (``*params``): autograd accumulates their gradients itself, so
    ``torch.autograd.grad(loss, net.parameters())``, gradient hooks (DDP) and a second backward through a retained graph behave
    as for any other module.  (Round 2 assigned ``p.grad`` inside backward: ``autograd.grad`` then failed with 'not used in
    the graph' and asking for the input gradients alone still wrote ``.grad``.)  Double backward is not supported
    (``once_differentiable``)."""

    @staticmethod
    def forward(ctx, obj_vecs, pred_vecs, edges, eng, training, *params):
        L = _lib.lib()
        x = obj_vecs.detach().float().contiguous(); p = pred_vecs.detach().float().contiguous()
        e = edges.to(torch.int64).contiguous()
        O, T = x.shape[0], p.shape[0]
        eng.ensure(O, T)
        st = _lib.current_stream_ptr()
        _lib.check(L.sln_gconv_net_set_edges(eng.h, _lib.ptr(e), O, T, st), "sln_gconv_net_set_edges")
        new_obj = torch.empty(O, eng.Dout, device=x.device); new_pred = torch.empty(T, eng.Dout, device=x.device)
        _lib.check(L.sln_gconv_net_forward(eng.h, _lib.ptr(x), _lib.ptr(p), _lib.ptr(new_obj), _lib.ptr(new_pred), int(training), st),
                   "sln_gconv_net_forward")
        eng.generation = getattr(eng, "generation", 0) + 1
        ctx.eng, ctx.gen, ctx.shapes = eng, eng.generation, (O, T)
        return new_obj, new_pred

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_obj, d_pred):
        eng = ctx.eng
        if ctx.gen != eng.generation:
            raise _lib.SlnError("backward() through a stale forward: another forward ran on this GraphTripleConv(Net) in between")
        O, T = ctx.shapes
        dev = eng.device
        d_obj = torch.zeros(O, eng.Dout, device=dev) if d_obj is None else d_obj.float().contiguous()
        d_pred = torch.zeros(T, eng.Dout, device=dev) if d_pred is None else d_pred.float().contiguous()
        dx = torch.empty(O, eng.D, device=dev); dp = torch.empty(T, eng.D, device=dev)
        for buf in eng.gbuf.values():           # the kernels accumulate (+=) into the engine's own buffers ...
            buf.zero_()
        _lib.check(_lib.lib().sln_gconv_net_backward(eng.h, _lib.ptr(d_obj), _lib.ptr(d_pred), _lib.ptr(dx), _lib.ptr(dp),
                                                     _lib.current_stream_ptr()), "sln_gconv_net_backward")
        # ... and autograd receives copies: a retained graph may run this backward again while the first result is still in use
        need = ctx.needs_input_grad
        pg = tuple(eng.gbuf[id(prm)].clone() if need[5 + i] else None for i, prm in enumerate(eng.params))
        return (dx if need[0] else None, dp if need[1] else None, None, None, None) + pg


def _rup4(n):
    return (n + 3) // 4 * 4


class _PaddedShadow:
    """Dimensions that are not multiples of 4 (the kernels' 16-byte rows): the layer runs on a SHADOW of itself whose widths are
    rounded up - extra weight rows / columns and BatchNorm gammas are zero, so the padded channels carry exact zeros through
    Linear, BatchNorm and ReLU and the real channels see the arithmetic of the unpadded layer.  The shadow's parameters are rebuilt
    from the real ones on every forward by a differentiable scatter (``index_put``): gradients reach the real parameters through
    it.  (The reference places no constraint on the three widths, models/graph.py:36-56.)"""

    def __init__(self, modules, device):
        m0 = modules[0]
        D, H, Do = m0.input_dim, m0.hidden_dim, m0.output_dim
        self.D, self.H, self.Do = D, H, Do
        Dp, Hp, Dop = _rup4(D), _rup4(H), _rup4(Do)
        norm = 'batch' if any(bn is not None for _, bn in mlp_linears(m0.net1)) else 'none'
        self.shadow = [GraphTripleConv(Dp, Dop, Hp, mlp_normalization=norm).to(device) for _ in modules]
        ar = lambda n, o=0: torch.arange(n, device=device) + o
        in1 = torch.cat([ar(D, k * Dp) for k in range(3)])
        out2 = torch.cat([ar(H), ar(Do, Hp), ar(H, Hp + Dop)])
        self.maps = [(ar(H), in1), (out2, ar(H)), (ar(H), ar(H)), (ar(Do), ar(H))]       # (rows, cols) of net1.0, net1.1, net2.0, net2.1
        for sm in self.shadow:
            for p in sm.parameters():
                p.requires_grad_(False)
            for _, bn in mlp_linears(sm.net1) + mlp_linears(sm.net2):
                if bn is not None:
                    bn.weight.zero_(); bn.bias.zero_()

    def sync(self, modules):
        """-> the padded parameter tensors (functions of the real parameters, in ``shadow.parameters()`` order); the shadow's own
        storage (what the engine is bound to) receives the same values."""
        padded = []
        for m, sm in zip(modules, self.shadow):
            pairs, spairs = mlp_linears(m.net1) + mlp_linears(m.net2), mlp_linears(sm.net1) + mlp_linears(sm.net2)
            per = {}
            for (lin, bn), (slin, sbn), (rows, cols) in zip(pairs, spairs, self.maps):
                per[id(slin.weight)] = torch.zeros_like(slin.weight).index_put((rows[:, None], cols[None, :]), lin.weight.float())
                per[id(slin.bias)] = torch.zeros_like(slin.bias).index_put((rows,), lin.bias.float())
                if bn is not None:
                    per[id(sbn.weight)] = torch.zeros_like(sbn.weight).index_put((rows,), bn.weight.float())
                    per[id(sbn.bias)] = torch.zeros_like(sbn.bias).index_put((rows,), bn.bias.float())
                    with torch.no_grad():
                        sbn.running_mean.zero_(); sbn.running_var.fill_(1.0)
                        sbn.running_mean[rows] = bn.running_mean; sbn.running_var[rows] = bn.running_var
                        sbn.num_batches_tracked.copy_(bn.num_batches_tracked)
            for sp in sm.parameters():
                pv = per[id(sp)]
                with torch.no_grad():
                    sp.copy_(pv)
                padded.append(pv)
        return padded

    def write_back_running_stats(self, modules):
        with torch.no_grad():
            for m, sm in zip(modules, self.shadow):
                for (lin, bn), (slin, sbn), (rows, _c) in zip(mlp_linears(m.net1) + mlp_linears(m.net2),
                                                               mlp_linears(sm.net1) + mlp_linears(sm.net2), self.maps):
                    if bn is not None:
                        bn.running_mean.copy_(sbn.running_mean[rows]); bn.running_var.copy_(sbn.running_var[rows])
                        bn.num_batches_tracked.copy_(sbn.num_batches_tracked)


def _gconv_autograd_padded(owner, modules, num_layers, obj_vecs, pred_vecs, edges, training):
    sh = getattr(owner, "_sln_shadow", None)
    if sh is None or sh.shadow[0].net1[0].weight.device != obj_vecs.device:
        sh = _PaddedShadow(modules, obj_vecs.device)
        object.__setattr__(owner, "_sln_shadow", sh)
    padded = sh.sync(modules)
    eng = getattr(owner, "_sln_engine", None)
    key = tuple(p.data_ptr() for m in sh.shadow for p in m.parameters())
    if eng is None or eng.key_ptrs != key or eng.device != obj_vecs.device:
        eng = _GconvEngine(sh.shadow, num_layers, obj_vecs.device)
        eng.key_ptrs = key
        object.__setattr__(owner, "_sln_engine", eng)
    F = torch.nn.functional
    Dp = sh.shadow[0].input_dim
    new_obj, new_pred = _GconvNetFn.apply(F.pad(obj_vecs.float(), (0, Dp - sh.D)), F.pad(pred_vecs.float(), (0, Dp - sh.D)), edges, eng, training,
                                          *padded)
    if training:
        sh.write_back_running_stats(modules)
    return new_obj[:, :sh.Do], new_pred[:, :sh.Do]


def _gconv_autograd(owner, modules, num_layers, obj_vecs, pred_vecs, edges, training):
    m0 = modules[0]
    if num_layers > 1 and m0.input_dim != m0.output_dim:
        raise ValueError("a stack of GraphTripleConv layers needs output_dim == input_dim (models/graph.py:121-131)")
    if m0.input_dim % 4 or m0.hidden_dim % 4 or m0.output_dim % 4:
        return _gconv_autograd_padded(owner, modules, num_layers, obj_vecs, pred_vecs, edges, training)
    if num_layers > 1 and m0.input_dim != m0.output_dim:
        raise ValueError("a stack of GraphTripleConv layers needs output_dim == input_dim (models/graph.py:121-131)")
    eng = getattr(owner, "_sln_engine", None)
    key = tuple(p.data_ptr() for m in modules for p in m.parameters())
    if eng is None or eng.key_ptrs != key or eng.device != obj_vecs.device:
        eng = _GconvEngine(modules, num_layers, obj_vecs.device)
        eng.key_ptrs = key
        object.__setattr__(owner, "_sln_engine", eng)
    return _GconvNetFn.apply(obj_vecs, pred_vecs, edges, eng, training, *eng.params)


def _gconv_forward(modules, num_layers, obj_vecs, pred_vecs, edges, training, owner=None):
    """Standalone GraphTripleConv(Net).forward on the HIP kernels.  With gradients enabled it runs on an engine handle that
    keeps the pre-activations for backward (sln_gconv_net_*: models/graph.py:57-111,136-143 are differentiable in the
    reference); under torch.no_grad() on the workspace-per-call inference entry point (sln_gconv_forward)."""
    if obj_vecs.device.type != 'cuda':
        raise _lib.SlnError("GraphTripleConv runs on the MI355X only (no CPU fallback)")
    if torch.is_grad_enabled() and (obj_vecs.requires_grad or pred_vecs.requires_grad or
                                    any(p.requires_grad for m in modules for p in m.parameters())):
        return _gconv_autograd(owner if owner is not None else modules[0], modules, num_layers, obj_vecs, pred_vecs, edges, training)
    m0 = modules[0]
    D, H, Do = m0.input_dim, m0.hidden_dim, m0.output_dim
    if D % 32 or H % 4 or Do % 4:
        # widths the workspace-per-call entry point does not take (it stages 32-column k-tiles of the input): the engine path runs
        # them - padded to multiples of 4 where needed - with or without gradients
        return _gconv_autograd(owner if owner is not None else modules[0], modules, num_layers, obj_vecs, pred_vecs, edges, training)
    units = (_lib.SlnVaeUnit * (4 * len(modules)))()
    bn_any = False
    for mi, m in enumerate(modules):
        for k, (lin, bn) in enumerate(mlp_linears(m.net1) + mlp_linears(m.net2)):
            u = units[4 * mi + k]
            u.weight, u.bias = lin.weight.data_ptr(), lin.bias.data_ptr()
            if bn is not None:
                bn_any = True
                u.bn_weight, u.bn_bias = bn.weight.data_ptr(), bn.bias.data_ptr()
                u.bn_running_mean, u.bn_running_var = bn.running_mean.data_ptr(), bn.running_var.data_ptr()
                u.bn_num_batches_tracked = bn.num_batches_tracked.data_ptr()
    L = _lib.lib()
    x = obj_vecs.detach().float().contiguous(); p = pred_vecs.detach().float().contiguous()
    e = edges.to(torch.int64).contiguous()
    O, T = x.shape[0], p.shape[0]
    nbytes = L.sln_gconv_workspace_bytes(D, H, Do, O, T, num_layers)
    if nbytes < 0:
        _lib.check(int(nbytes), "sln_gconv_workspace_bytes")
    ws = torch.empty(int(nbytes), dtype=torch.uint8, device=x.device)
    new_obj = torch.empty(O, Do, device=x.device); new_pred = torch.empty(T, Do, device=x.device)
    _lib.check(L.sln_gconv_forward(D, H, Do, num_layers, len(modules), int(bn_any), units, _lib.ptr(x), _lib.ptr(p), _lib.ptr(e), O, T,
                                   int(training), _lib.ptr(ws), int(nbytes), _lib.ptr(new_obj), _lib.ptr(new_pred),
                                   _lib.current_stream_ptr()), "sln_gconv_forward")
    return new_obj, new_pred
