"""The CPU oracle (oracle/vae_ref.py) against fixtures produced by the reference itself.

Pins SURVEY.md §8c row A: outputs, the three loss scalars, every parameter gradient,
BatchNorm running statistics and one Adam step, in train and eval mode, for
mlp_normalization in {batch, none}, feedforward / recurrent, decoder_cat on/off, AE.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import vae_ref
from oracle.gen_golden import VAE_CASES, KL_WEIGHT

from parity import assert_close as _close, assert_adam_close


@pytest.mark.parametrize("name", list(VAE_CASES))
def test_oracle_matches_reference_fixture(name):
    # the fixtures were generated single-threaded; c1 (BatchNorm over 8 rows) is so
    # ill-conditioned that the reference differs from itself at 8 threads (tests/parity.py)
    torch.set_num_threads(1)
    g = load_golden(name)
    cfg = vae_ref.VaeConfig(**VAE_CASES[name][0])
    sd = vae_ref.init_state(cfg, seed=42)
    ins = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("in:")}
    batch = (ins["objs"], ins["triples"], ins["boxes"], ins["angles"], ins["attrs"])

    # eval mode
    with torch.no_grad():
        mu, lv, bp, ap = vae_ref.forward(sd, cfg, *batch, ins["eps"], training=False)
    for k, v in (("eval_mu", mu), ("eval_logvar", lv), ("eval_boxes_pred", bp), ("eval_angles_pred", ap)):
        _close(v.numpy(), g[k], name + ":" + k)

    # train mode forward + loss + backward + Adam
    m = {k: torch.zeros_like(sd[k]) for k in vae_ref.trainable_keys(cfg)}
    v = {k: torch.zeros_like(sd[k]) for k in vae_ref.trainable_keys(cfg)}
    total, parts, grads = vae_ref.train_step(sd, cfg, batch, ins["eps"], KL_WEIGHT, m, v, step=1)
    _close(total.numpy(), g["total_loss"], name + ":total")
    for k, val in parts.items():
        _close(val, g["loss_" + k], name + ":loss_" + k)
    n_checked = 0
    gscale = max(float(np.abs(g[k]).max()) for k in g.files if k.startswith("grad:"))
    for k in g.files:
        if k.startswith("grad:"):
            got = grads.get(k[5:])
            got = got.numpy() if got is not None else np.zeros_like(g[k])
            _close(got, g[k], name + ":" + k, atol=5e-6 * gscale)
            n_checked += 1
        elif k.startswith("gsum:"):
            gg = grads[k[5:]].double()
            got = np.array([gg.sum(), gg.abs().sum(), (gg * gg).sum()])
            _close(got[1:], g[k][1:], name + ":" + k, rtol=1e-4)
        elif k.startswith("buf:"):
            _close(sd[k[4:]].numpy(), g[k], name + ":" + k)
        elif k.startswith("adam:"):
            if "grad:" + k[5:] in g.files:
                assert_adam_close(sd[k[5:]].numpy(), g[k], g["grad:" + k[5:]], name + ":" + k, gscale=gscale)
    assert n_checked > 10


def test_state_layout_is_reference_key_set():
    """230-ish keys with the reference's names; count pinned by the c1 fixture's buffers+grads."""
    cfg = vae_ref.VaeConfig()
    sd = vae_ref.init_state(cfg, 0)
    n_params = sum(sd[k].numel() for k in vae_ref.trainable_keys(cfg))
    assert n_params == 3879790
    assert "gconv_net_ec.gconvs.4.net1.4.running_var" in sd
    assert "box_net.3.weight" in sd and "box_net.4.weight" not in sd
