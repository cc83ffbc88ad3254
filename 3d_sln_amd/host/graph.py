"""Parameter containers with the reference's names for the graph-convolution blocks.

Mirrors models/graph.py of the reference: ``make_mlp`` (:10-27), ``_init_weights`` (:30-33),
``GraphTripleConv`` (:36-111), ``GraphTripleConvNet`` (:114-143).  The modules own the
parameters (so ``state_dict`` keys are the reference's: ``gconvs.{i}.net1.{0,1,3,4}.*``);
the arithmetic runs in the HIP engine (csrc/vae_engine.hip), which the owning
``Sg2ScVAEModel`` drives.  A GraphTripleConv(Net) used on its own also runs on the
engine's kernels through ``forward`` below.
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
        """(new_obj_vecs [O, Dout], new_pred_vecs [T, Dout]) - inference only (see _gconv_forward)."""
        return _gconv_forward([self], 1, obj_vecs, pred_vecs, edges, self.training)


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
        return _gconv_forward(list(self.gconvs), self.num_layers, obj_vecs, pred_vecs, edges, self.training)


def _gconv_forward(modules, num_layers, obj_vecs, pred_vecs, edges, training):
    """Standalone GraphTripleConv(Net).forward on the HIP kernels (sln_gconv_forward).  No autograd: training runs
    through Sg2ScVAEModel (fused forward/backward in csrc/vae_engine.hip)."""
    if torch.is_grad_enabled() and (obj_vecs.requires_grad or pred_vecs.requires_grad or
                                    any(p.requires_grad for m in modules for p in m.parameters())):
        raise NotImplementedError("standalone GraphTripleConv(Net).forward is inference-only on the HIP path "
                                  "(wrap the call in torch.no_grad(); train through Sg2ScVAEModel)")
    if obj_vecs.device.type != 'cuda':
        raise _lib.SlnError("GraphTripleConv runs on the MI355X only (no CPU fallback)")
    m0 = modules[0]
    D, H, Do = m0.input_dim, m0.hidden_dim, m0.output_dim
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
