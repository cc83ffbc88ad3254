"""CPU oracle for the scene-graph VAE hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``3d_sln_amd/`` may import this file;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and only as the checker / the timed CPU baseline.

This is a *functional* restatement of the reference algorithm in plain
PyTorch fp32 on the CPU.  It operates directly on a ``state_dict`` whose keys
are the reference's (``gconv_net_ec.gconvs.0.net1.0.weight`` ...), so the very
same tensors can be loaded into the reference classes (in the build container,
see ``oracle/gen_golden.py``) and into the HIP-backed module.

Parity status: PINNED.  ``oracle/gen_golden.py`` imports the reference modules
from /root/reference on CPU, loads the same deterministic state and writes the
fixtures under ``tests/golden/``; ``tests/test_oracle_vae.py`` checks this
restatement against those fixtures.

Reference lines followed (all relative to /root/reference):
  make_mlp ................ models/graph.py:10-27
  GraphTripleConv.forward . models/graph.py:57-111
  GraphTripleConvNet ...... models/graph.py:136-143
  Sg2ScVAEModel.__init__ .. models/Sg2ScVAE_model.py:7-113
  encoder / decoder ....... models/Sg2ScVAE_model.py:115-172
  forward (reparam.) ...... models/Sg2ScVAE_model.py:174-188
  calculate_model_losses .. utils.py:12-33, add_loss utils.py:139-146
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

State = Dict[str, torch.Tensor]
TRACE = None     # set to a dict to record every Linear output ("pre-activation") by state_dict prefix
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


@dataclasses.dataclass
class VaeConfig:
    """Constructor arguments of the reference model that shape the tensors.

    Defaults are ``train.py``'s (options/options.py:24-34,55), not the class
    defaults.
    """
    embedding_dim: int = 64
    gconv_num_layers: int = 5
    gconv_mode: str = "feedforward"          # or "recurrent"
    mlp_normalization: str = "batch"         # or "none"
    decoder_cat: bool = True
    use_AE: bool = False
    use_attr: bool = True                    # the class default; build_model never passes it (Sg2ScVAE_model.py:17)
    train_3d: bool = True
    Nangle: int = 24
    num_objs: int = 32                       # len(vocab['object_idx_to_name'])
    num_preds: int = 16
    num_attrs: int = 5

    # derived sizes (models/Sg2ScVAE_model.py:19-37)
    @property
    def gconv_dim(self): return self.embedding_dim
    @property
    def hidden(self): return self.embedding_dim * 4
    @property
    def box_emb(self): return int(self.embedding_dim * 3 / 4)
    @property
    def angle_emb(self): return int(self.embedding_dim / 4)
    @property
    def obj_emb(self): return int(self.embedding_dim * 3 / 4) if self.use_attr else self.embedding_dim      # :24,35-37
    @property
    def attr_emb(self): return int(self.embedding_dim / 4) if self.use_attr else 0
    @property
    def box_dim(self): return 6 if self.train_3d else 4
    @property
    def d_ec(self): return self.embedding_dim * 2
    @property
    def d_dc(self): return self.embedding_dim * 2 if self.decoder_cat else self.embedding_dim
    @property
    def n_gconv_modules(self): return 1 if self.gconv_mode == "recurrent" else self.gconv_num_layers

    def vocab(self) -> dict:
        return {
            "object_idx_to_name": ["__room__"] + ["type%02d" % i for i in range(1, self.num_objs)],
            "pred_idx_to_name": ["pred%02d" % i for i in range(self.num_preds)],
            "attrib_idx_to_name": ["attr%d" % i for i in range(self.num_attrs)],
        }

    def model_kwargs(self) -> dict:
        """kwargs for the reference ctor, as build_dataset_model.py:40-52 passes them."""
        kw = dict(vocab=self.vocab(), batch_size=128, train_3d=self.train_3d,
                  decoder_cat=self.decoder_cat, embedding_dim=self.embedding_dim,
                  gconv_mode=self.gconv_mode, gconv_num_layers=self.gconv_num_layers,
                  mlp_normalization=self.mlp_normalization, vec_noise_dim=0,
                  layout_noise_dim=32, use_AE=self.use_AE)
        if not self.use_attr:
            kw["use_attr"] = False
        return kw


# --------------------------------------------------------------------------
# state layout
# --------------------------------------------------------------------------
def _mlp_keys(prefix: str, dims: List[int], norm: str, norelu: bool) -> List[Tuple[str, tuple, str]]:
    """(key, shape, kind) for an MLP built like models/graph.py:10-27."""
    out = []
    per = 3 if norm == "batch" else 2            # Linear,[BN],ReLU
    n_lin = len(dims) - 1
    for i in range(n_lin):
        base = "%s.%d" % (prefix, i * per)
        out.append((base + ".weight", (dims[i + 1], dims[i]), "w"))
        out.append((base + ".bias", (dims[i + 1],), "b"))
        last = i == n_lin - 1
        if norm == "batch" and not (norelu and last):
            bn = "%s.%d" % (prefix, i * per + 1)
            out.append((bn + ".weight", (dims[i + 1],), "bn_w"))
            out.append((bn + ".bias", (dims[i + 1],), "bn_b"))
            out.append((bn + ".running_mean", (dims[i + 1],), "bn_rm"))
            out.append((bn + ".running_var", (dims[i + 1],), "bn_rv"))
            out.append((bn + ".num_batches_tracked", (), "bn_nbt"))
    return out


def state_layout(cfg: VaeConfig) -> List[Tuple[str, tuple, str]]:
    E, H, n = cfg.embedding_dim, cfg.hidden, cfg.mlp_normalization
    L: List[Tuple[str, tuple, str]] = []
    L.append(("obj_embeddings_ec.weight", (cfg.num_objs + 1, cfg.obj_emb), "emb"))
    L.append(("pred_embeddings_ec.weight", (cfg.num_preds, 2 * E), "emb"))
    L.append(("obj_embeddings_dc.weight", (cfg.num_objs + 1, cfg.obj_emb), "emb"))
    L.append(("pred_embeddings_dc.weight", (cfg.num_preds, cfg.d_dc), "emb"))
    if cfg.use_attr:                                                            # :48-50
        L.append(("attr_embedding_ec.weight", (cfg.num_attrs, cfg.attr_emb), "emb"))
        L.append(("attr_embedding_dc.weight", (cfg.num_attrs, cfg.attr_emb), "emb"))
    L.append(("box_embeddings.weight", (cfg.box_emb, cfg.box_dim), "w"))
    L.append(("box_embeddings.bias", (cfg.box_emb,), "b"))
    L.append(("angle_embeddings.weight", (cfg.Nangle, cfg.angle_emb), "emb"))
    L += _mlp_keys("box_mean_var", [2 * E, H, 2 * E], n, False)
    L += _mlp_keys("box_mean", [2 * E, cfg.box_emb], n, True)
    L += _mlp_keys("box_var", [2 * E, cfg.box_emb], n, True)
    L += _mlp_keys("angle_mean_var", [2 * E, H, 2 * E], n, False)
    L += _mlp_keys("angle_mean", [2 * E, cfg.angle_emb], n, True)
    L += _mlp_keys("angle_var", [2 * E, cfg.angle_emb], n, True)
    for tag, D in (("ec", cfg.d_ec), ("dc", cfg.d_dc)):
        for i in range(cfg.n_gconv_modules):
            p = "gconv_net_%s.gconvs.%d" % (tag, i)
            L += _mlp_keys(p + ".net1", [3 * D, H, 2 * H + D], n, False)
            L += _mlp_keys(p + ".net2", [H, H, D], n, False)
    L += _mlp_keys("box_net", [2 * E + cfg.attr_emb, H, cfg.box_dim], n, True)
    L += _mlp_keys("angle_net", [2 * E, H, cfg.Nangle], n, True)
    return L


def init_state(cfg: VaeConfig, seed: int = 0, scale: float = 1.0) -> State:
    """Deterministic, torch-version-independent parameter fill.

    Keys are visited in sorted order and filled from one numpy Generator:
    Linear weights ~ N(0, 2/fan_in) (kaiming-like, so activations stay O(1)),
    biases ~ N(0, 0.1^2), embeddings ~ N(0,1), BN gamma ~ U(0.5,1.5),
    BN beta ~ N(0,0.1^2), running_mean ~ N(0,0.1^2), running_var ~ U(0.5,1.5).
    """
    rng = np.random.default_rng(seed)
    sd: State = {}
    for key, shape, kind in sorted(state_layout(cfg)):
        if kind == "w":
            v = rng.standard_normal(shape) * np.sqrt(2.0 / shape[1]) * scale
        elif kind in ("b", "bn_b", "bn_rm"):
            v = rng.standard_normal(shape) * 0.1
        elif kind == "emb":
            v = rng.standard_normal(shape)
        elif kind in ("bn_w", "bn_rv"):
            v = rng.uniform(0.5, 1.5, shape)
        elif kind == "bn_nbt":
            sd[key] = torch.zeros((), dtype=torch.int64)
            continue
        else:
            raise AssertionError(kind)
        sd[key] = torch.from_numpy(np.asarray(v, dtype=np.float32)).clone()
    return sd


def trainable_keys(cfg: VaeConfig) -> List[str]:
    return [k for k, _, kind in state_layout(cfg) if kind in ("w", "b", "emb", "bn_w", "bn_b")]


# --------------------------------------------------------------------------
# functional forward
# --------------------------------------------------------------------------
def mlp_apply(sd: State, prefix: str, n_lin: int, x: torch.Tensor, norm: str,
              training: bool, norelu: bool = False) -> torch.Tensor:
    """Linear -> [BatchNorm1d] -> ReLU chain (models/graph.py:10-27)."""
    per = 3 if norm == "batch" else 2
    for i in range(n_lin):
        base = "%s.%d" % (prefix, i * per)
        x = F.linear(x, sd[base + ".weight"], sd[base + ".bias"])
        if TRACE is not None:
            TRACE.setdefault(base, []).append(x.detach().clone())
        if norelu and i == n_lin - 1:
            break
        if norm == "batch":
            bn = "%s.%d" % (prefix, i * per + 1)
            if training:
                sd[bn + ".num_batches_tracked"] += 1
            x = F.batch_norm(x, sd[bn + ".running_mean"], sd[bn + ".running_var"],
                             sd[bn + ".weight"], sd[bn + ".bias"], training,
                             BN_MOMENTUM, BN_EPS)
        x = torch.relu(x)
    return x


def gconv_apply(sd: State, prefix: str, obj: torch.Tensor, pred: torch.Tensor,
                edges: torch.Tensor, H: int, norm: str, training: bool):
    """One GraphTripleConv (models/graph.py:57-111)."""
    O, D = obj.shape
    s_idx, o_idx = edges[:, 0], edges[:, 1]
    t_in = torch.cat([obj[s_idx], pred, obj[o_idx]], dim=1)
    t_out = mlp_apply(sd, prefix + ".net1", 2, t_in, norm, training)
    Do = t_out.shape[1] - 2 * H                      # output_dim (graph.py:84-86); = D inside a GraphTripleConvNet
    new_s, new_p, new_o = t_out[:, :H], t_out[:, H:H + Do], t_out[:, H + Do:2 * H + Do]
    pooled = torch.zeros(O, H, dtype=obj.dtype)
    pooled = pooled.index_add(0, s_idx, new_s)       # scatter_add along rows, s first
    pooled = pooled.index_add(0, o_idx, new_o)       # then o (graph.py:97-98)
    deg = torch.zeros(O, dtype=obj.dtype)
    ones = torch.ones(edges.shape[0], dtype=obj.dtype)
    deg = deg.index_add(0, s_idx, ones).index_add(0, o_idx, ones).clamp(min=1)
    pooled = pooled / deg[:, None]
    if TRACE is not None:
        TRACE.setdefault(prefix + ".pooled", []).append(pooled.detach().clone())
    new_obj = mlp_apply(sd, prefix + ".net2", 2, pooled, norm, training)
    return new_obj, new_p


def gconv_net_apply(sd: State, cfg: VaeConfig, tag: str, obj, pred, edges, training: bool):
    for i in range(cfg.gconv_num_layers):
        j = 0 if cfg.gconv_mode == "recurrent" else i
        obj, pred = gconv_apply(sd, "gconv_net_%s.gconvs.%d" % (tag, j), obj, pred, edges,
                                cfg.hidden, cfg.mlp_normalization, training)
    return obj, pred


def encoder(sd: State, cfg: VaeConfig, objs, triples, boxes, angles, attrs, training: bool):
    """models/Sg2ScVAE_model.py:115-143 -> (mu, logvar) each [O, embedding_dim]."""
    s, p, o = triples[:, 0], triples[:, 1], triples[:, 2]
    edges = torch.stack([s, o], dim=1)
    parts = [sd["obj_embeddings_ec.weight"][objs]]
    if cfg.use_attr:
        parts.append(sd["attr_embedding_ec.weight"][attrs])
    x = torch.cat(parts + [F.linear(boxes, sd["box_embeddings.weight"], sd["box_embeddings.bias"]),
                           sd["angle_embeddings.weight"][angles]], dim=1)
    pv = sd["pred_embeddings_ec.weight"][p]
    if cfg.gconv_num_layers > 0:
        x, pv = gconv_net_apply(sd, cfg, "ec", x, pv, edges, training)
    n = cfg.mlp_normalization
    hb = mlp_apply(sd, "box_mean_var", 2, x, n, training)
    mu_b = mlp_apply(sd, "box_mean", 1, hb, n, training, norelu=True)
    lv_b = mlp_apply(sd, "box_var", 1, hb, n, training, norelu=True)
    ha = mlp_apply(sd, "angle_mean_var", 2, x, n, training)
    mu_a = mlp_apply(sd, "angle_mean", 1, ha, n, training, norelu=True)
    lv_a = mlp_apply(sd, "angle_var", 1, ha, n, training, norelu=True)
    return torch.cat([mu_b, mu_a], 1), torch.cat([lv_b, lv_a], 1)


def decoder(sd: State, cfg: VaeConfig, z, objs, triples, attrs, training: bool):
    """models/Sg2ScVAE_model.py:145-172 -> (boxes_pred [O,6], angles_pred [O,24] log-probs)."""
    s, p, o = triples[:, 0], triples[:, 1], triples[:, 2]
    edges = torch.stack([s, o], dim=1)
    x = sd["obj_embeddings_dc.weight"][objs]
    if cfg.use_attr:
        attr_v = sd["attr_embedding_dc.weight"][attrs]
        x = torch.cat([x, attr_v], dim=1)
    pv = sd["pred_embeddings_dc.weight"][p]
    if cfg.decoder_cat:
        x = torch.cat([x, z], dim=1)
        x, pv = gconv_net_apply(sd, cfg, "dc", x, pv, edges, training)
    else:
        x, pv = gconv_net_apply(sd, cfg, "dc", x, pv, edges, training)
        x = torch.cat([x, z], dim=1)
    n = cfg.mlp_normalization
    boxes_pred = mlp_apply(sd, "box_net", 2, torch.cat([x, attr_v], 1) if cfg.use_attr else x, n, training, norelu=True)
    angles_pred = F.log_softmax(mlp_apply(sd, "angle_net", 2, x, n, training, norelu=True), dim=1)
    return boxes_pred, angles_pred


def forward(sd: State, cfg: VaeConfig, objs, triples, boxes, angles, attrs,
            eps: Optional[torch.Tensor], training: bool):
    """models/Sg2ScVAE_model.py:174-188 with the N(0,1) draw injected as ``eps``."""
    mu, logvar = encoder(sd, cfg, objs, triples, boxes, angles, attrs, training)
    if cfg.use_AE:
        z = mu
    else:
        z = eps * torch.exp(0.5 * logvar) + mu
    boxes_pred, angles_pred = decoder(sd, cfg, z, objs, triples, attrs, training)
    return mu, logvar, boxes_pred, angles_pred


def losses(cfg: VaeConfig, boxes, boxes_pred, angles, angles_pred, mu, logvar, kl_weight: float):
    """utils.py:12-33: L1(box) + NLL(angle) + w * KL; returns (total, dict of weighted parts)."""
    l_box = F.l1_loss(boxes_pred, boxes)
    l_ang = F.nll_loss(angles_pred, angles)
    total = l_box + l_ang
    parts = {"bbox_pred": l_box, "angle_pred": l_ang}
    if not cfg.use_AE:
        l_kl = -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp()) / mu.size(0)
        total = total + l_kl * kl_weight
        parts["KLD_Gauss"] = l_kl * kl_weight
    return total, parts


def adam_step(sd: State, grads: Dict[str, torch.Tensor], m: State, v: State, step: int,
              lr: float = 1e-4, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8) -> None:
    """torch.optim.Adam defaults as train.py:15 uses them (no weight decay, no amsgrad)."""
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    with torch.no_grad():
        for k, g in grads.items():
            m[k].mul_(b1).add_(g, alpha=1 - b1)
            v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
            denom = (v[k].sqrt() / (bc2 ** 0.5)).add_(eps)
            sd[k].addcdiv_(m[k], denom, value=-lr / bc1)


def train_step(sd: State, cfg: VaeConfig, batch, eps, kl_weight: float, m: State, v: State,
               step: int, lr: float = 1e-4, training: bool = True):
    """One iteration of train.py:62-84 on the CPU (forward, loss, backward, Adam).  ``training=False``: BatchNorm on its
    running statistics, as after ``--eval_mode_after`` (train.py:63-65 calls model.eval() and keeps optimising)."""
    keys = trainable_keys(cfg)
    for k in keys:
        sd[k].requires_grad_(True)
        sd[k].grad = None
    objs, triples, boxes, angles, attrs = batch
    mu, logvar, bp, ap = forward(sd, cfg, objs, triples, boxes, angles, attrs, eps, training)
    total, parts = losses(cfg, boxes, bp, angles, ap, mu, logvar, kl_weight)
    total.backward()
    grads = {k: sd[k].grad for k in keys if sd[k].grad is not None}
    for k in keys:
        sd[k].requires_grad_(False)
    adam_step(sd, grads, m, v, step, lr)
    return total.detach(), {k: float(x.detach()) for k, x in parts.items()}, grads


# --------------------------------------------------------------------------
# synthetic scene graphs honouring suncg_collate_fn's tuple
# (data/suncg_dataset.py:207-212 in-room rows, :295-337 offsets)
# --------------------------------------------------------------------------
def synth_batch(n_graphs: int, objs_per_graph: int, triples_per_graph: int, seed: int = 0,
                cfg: Optional[VaeConfig] = None):
    cfg = cfg or VaeConfig()
    rng = np.random.default_rng(seed)
    n, tt = objs_per_graph, triples_per_graph
    n_rand = tt - (n - 1)
    assert n >= 2 and n_rand >= 0
    objs, trip, boxes, angles, attrs, o2i = [], [], [], [], [], []
    for g in range(n_graphs):
        off = g * n
        ob = rng.integers(1, cfg.num_objs, size=n)
        ob[-1] = 0                                            # __room__ is last
        s = rng.integers(0, n - 1, size=n_rand)
        o = (s + rng.integers(1, n - 1, size=n_rand)) % (n - 1) if n > 2 else s
        p = rng.integers(1, cfg.num_preds, size=n_rand)
        rows = [np.stack([s + off, p, o + off], 1)]
        rows.append(np.stack([np.arange(n - 1) + off, np.zeros(n - 1, np.int64),
                              np.full(n - 1, n - 1 + off)], 1))
        lo = rng.uniform(0.0, 0.7, size=(n, 3))
        hi = lo + rng.uniform(0.05, 0.3, size=(n, 3))
        bx = np.concatenate([lo, hi], 1)
        bx[-1] = [0, 0, 0, 1, 1, 1]
        if not cfg.train_3d:
            bx = bx[:, [0, 2, 3, 5]]
        objs.append(ob); trip.append(np.concatenate(rows, 0)); boxes.append(bx)
        angles.append(rng.integers(0, cfg.Nangle, size=n))
        attrs.append(rng.integers(0, cfg.num_attrs, size=n))
        o2i.append(np.full(n, g))
    t = lambda a, dt: torch.from_numpy(np.concatenate(a, 0).astype(dt))
    return (t(objs, np.int64), t(trip, np.int64), t(boxes, np.float32),
            t(angles, np.int64), t(attrs, np.int64), t(o2i, np.int64))
