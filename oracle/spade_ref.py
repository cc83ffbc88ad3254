"""CPU oracle for the SPADE generator path.

TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's cpu_baseline leg).  Functional PyTorch-CPU restatement
of ``SPADEGenerator4.forward`` in eval mode over a ``state_dict`` with the reference's key names.

Parity status: PINNED - ``oracle/gen_golden_spade.py`` runs the reference's own
``models/SPADE_related.py::SPADEGenerator4`` on CPU with the same deterministic state and writes
``tests/golden/spade_*.npz``; ``tests/test_oracle_spade.py`` holds this file to them.

Reference lines followed (relative to /root/reference/models/SPADE_related.py):
  SEBlock2 ............ :70-85      LayerNorm2D ........ :128-149 (unbiased std, eps added to sigma)
  SPADE4 .............. :1404-1454  SPADEResnetBlock4 .. :1457-1505
  SPADEGenerator4 ..... :1507-1605  spectral_norm (eval): W = W_orig / (u . (W_mat v)), torch.nn.utils.spectral_norm
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

State = Dict[str, torch.Tensor]
NHIDDEN = 128          # SPADE_related.py:1426 ("Yes, hardcoded.")


@dataclasses.dataclass
class SpadeConfig:
    semantic_nc: int = 41
    target_nc: int = 3
    nz: int = 256
    ngf: int = 64
    crop_size: int = 256          # testing/test_SPADE_shade.py:9: (41, 3, 256, 64, 'spectralspadelayer3x3', 256, 'normal')

    @property
    def sw(self): return self.crop_size // 32          # 'normal' = 5 upsamplings (:1546-1561)

    def blocks(self) -> List[Tuple[str, int, int]]:
        nf = self.ngf
        return [("head_0", 16 * nf, 16 * nf), ("G_middle_0", 16 * nf, 16 * nf), ("G_middle_1", 16 * nf, 16 * nf),
                ("up_0", 16 * nf, 8 * nf), ("up_1", 8 * nf, 4 * nf), ("up_2", 4 * nf, 2 * nf), ("up_3", 2 * nf, nf)]


def state_layout(cfg: SpadeConfig):
    L = [("fc.weight", (16 * cfg.ngf * cfg.sw * cfg.sw, cfg.nz), "w"), ("fc.bias", (16 * cfg.ngf * cfg.sw * cfg.sw,), "b")]

    def spade(prefix, c):
        return [(prefix + ".mlp_preshared_depth.1.weight", (NHIDDEN // 8, 1, 3, 3), "w"), (prefix + ".mlp_preshared_depth.1.bias", (NHIDDEN // 8,), "b"),
                (prefix + ".mlp_shared.1.weight", (NHIDDEN, NHIDDEN // 8 + cfg.semantic_nc - 1, 3, 3), "w"), (prefix + ".mlp_shared.1.bias", (NHIDDEN,), "b"),
                (prefix + ".mlp_gamma.1.weight", (c, NHIDDEN, 3, 3), "w"), (prefix + ".mlp_gamma.1.bias", (c,), "b"),
                (prefix + ".mlp_beta.1.weight", (c, NHIDDEN, 3, 3), "w"), (prefix + ".mlp_beta.1.bias", (c,), "b")]
    for name, fin, fout in cfg.blocks():
        fmid = min(fin, fout)
        for cn, (co, ci, k) in (("conv_0.1", (fmid, fin, 3)), ("conv_1.1", (fout, fmid, 3))):
            L += [("%s.%s.weight_orig" % (name, cn), (co, ci, k, k), "w"), ("%s.%s.bias" % (name, cn), (co,), "b"),
                  ("%s.%s.weight_u" % (name, cn), (co,), "u"), ("%s.%s.weight_v" % (name, cn), (ci * k * k,), "u")]
        if fin != fout:
            L += [(name + ".conv_s.weight_orig", (fout, fin, 1, 1), "w"), (name + ".conv_s.weight_u", (fout,), "u"),
                  (name + ".conv_s.weight_v", (fin,), "u")]
        L += [(name + ".se.fc.0.weight", (fout // 8, fout), "w"), (name + ".se.fc.2.weight", (fout, fout // 8), "w")]
        L += spade(name + ".norm_0", fin) + spade(name + ".norm_1", fmid)
        if fin != fout:
            L += spade(name + ".norm_s", fin)
    L += [("conv_img.weight", (cfg.target_nc, cfg.ngf, 5, 5), "w"), ("conv_img.bias", (cfg.target_nc,), "b")]
    return L


def init_state(cfg: SpadeConfig, seed: int = 0, img_gain: float = 0.15) -> State:
    """Deterministic fill (numpy Generator, sorted key order): weights ~ N(0, 1/fan_in) (activations stay O(1)
    through the 14 residual stages), biases ~ N(0, 0.05^2), spectral u/v = random unit vectors."""
    rng = np.random.default_rng(seed)
    sd: State = {}
    for key, shape, kind in sorted(state_layout(cfg)):
        if kind == "w":
            fan_in = int(np.prod(shape[1:]))
            v = rng.standard_normal(shape, dtype=np.float32) * np.float32(1.0 / np.sqrt(fan_in))
            if key == "conv_img.weight":
                v *= np.float32(img_gain)             # keep the final tanh out of saturation (full width: see gen_golden_spade.CASES)
        elif kind == "b":
            v = rng.standard_normal(shape, dtype=np.float32) * np.float32(0.05)
        else:
            v = rng.standard_normal(shape, dtype=np.float32)
            v /= np.linalg.norm(v)
        sd[key] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    return sd


def synth_input(cfg: SpadeConfig, batch: int, seed: int = 0):
    """channel 0 = smooth depth in [-1,1], channels 1..40 = one-hot of the argmax of 40 upsampled
    noise maps (the tensor contract of testing/test_SPADE_shade.py:50-76); z ~ N(0,1)."""
    rng = np.random.default_rng(seed)
    s, lo = cfg.crop_size, 16
    low = torch.from_numpy(rng.uniform(-1, 1, size=(batch, 1, lo, lo)).astype(np.float32))
    depth = F.interpolate(low, size=(s, s), mode="bilinear", align_corners=False).clamp(-1, 1)
    noise = torch.from_numpy(rng.standard_normal((batch, cfg.semantic_nc - 1, lo, lo)).astype(np.float32))
    lab = F.interpolate(noise, size=(s, s), mode="bilinear", align_corners=False).argmax(1)
    onehot = F.one_hot(lab, cfg.semantic_nc - 1).permute(0, 3, 1, 2).float()
    z = torch.from_numpy(rng.standard_normal((batch, cfg.nz)).astype(np.float32))
    return torch.cat([depth, onehot], 1).contiguous(), z


# ----------------------------------------------------------------------------------------------
def sn_weight(sd: State, prefix: str) -> torch.Tensor:
    w = sd[prefix + ".weight_orig"]
    sigma = torch.dot(sd[prefix + ".weight_u"], torch.mv(w.reshape(w.shape[0], -1), sd[prefix + ".weight_v"]))
    return w / sigma


def layernorm2d(x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    flat = x.reshape(x.shape[0], -1)
    mean = flat.mean(1).view(-1, 1, 1, 1)
    std = flat.std(1).view(-1, 1, 1, 1)              # unbiased
    return (x - mean) / (std + eps)


def _conv_reflect(x, w, b):
    return F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w, b)


def spade4(sd: State, p: str, x: torch.Tensor, seg: torch.Tensor) -> torch.Tensor:
    normalized = layernorm2d(x)
    s = F.interpolate(seg, size=x.shape[2:], mode="bilinear", align_corners=False)
    d = F.leaky_relu(_conv_reflect(s[:, 0:1], sd[p + ".mlp_preshared_depth.1.weight"], sd[p + ".mlp_preshared_depth.1.bias"]), 0.01)
    a = F.relu(_conv_reflect(torch.cat((d, s[:, 1:]), 1), sd[p + ".mlp_shared.1.weight"], sd[p + ".mlp_shared.1.bias"]))
    gamma = _conv_reflect(a, sd[p + ".mlp_gamma.1.weight"], sd[p + ".mlp_gamma.1.bias"])
    beta = _conv_reflect(a, sd[p + ".mlp_beta.1.weight"], sd[p + ".mlp_beta.1.bias"])
    return normalized * (1 + gamma) + beta


def se_block(sd: State, p: str, x: torch.Tensor) -> torch.Tensor:
    y = x.mean((2, 3))
    y = torch.sigmoid(F.linear(F.relu(F.linear(y, sd[p + ".fc.0.weight"])), sd[p + ".fc.2.weight"]))
    return x * y[:, :, None, None]


def resblock(sd: State, p: str, x: torch.Tensor, seg: torch.Tensor, fin: int, fout: int) -> torch.Tensor:
    if fin != fout:
        x_s = F.conv2d(spade4(sd, p + ".norm_s", x, seg), sn_weight(sd, p + ".conv_s"))
    else:
        x_s = x
    dx = _conv_reflect(F.leaky_relu(spade4(sd, p + ".norm_0", x, seg), 0.2), sn_weight(sd, p + ".conv_0.1"), sd[p + ".conv_0.1.bias"])
    dx = _conv_reflect(F.leaky_relu(spade4(sd, p + ".norm_1", dx, seg), 0.2), sn_weight(sd, p + ".conv_1.1"), sd[p + ".conv_1.1.bias"])
    return x_s + se_block(sd, p + ".se", dx)


def generator(sd: State, cfg: SpadeConfig, seg: torch.Tensor, z: torch.Tensor, taps: dict = None) -> torch.Tensor:
    """SPADEGenerator4.forward (:1563-1605), n_up='normal', has_z."""
    nf, sw = cfg.ngf, cfg.sw
    x = F.linear(z, sd["fc.weight"], sd["fc.bias"]).view(-1, 16 * nf, sw, sw)
    up_n = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
    up_b = lambda t: F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False)
    seg_1 = F.interpolate(seg, size=[sw, sw])                       # nearest (:1579)
    blocks = dict((n, (a, b)) for n, a, b in cfg.blocks())

    def run(name, t, s):
        t = resblock(sd, name, t, s, *blocks[name])
        if taps is not None:
            taps[name] = t
        return t
    x = run("head_0", x, seg_1)
    x = run("G_middle_0", up_n(x), seg)
    x = run("G_middle_1", x, seg)
    x = run("up_0", up_n(x), seg)
    x = run("up_1", up_n(x), seg)
    x = run("up_2", up_n(x), seg)
    x = run("up_3", up_b(x), seg)
    x = F.conv2d(F.leaky_relu(x, 0.2), sd["conv_img.weight"], sd["conv_img.bias"], padding=2)
    return torch.tanh(x)
