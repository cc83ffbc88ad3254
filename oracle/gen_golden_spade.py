"""Golden fixtures for the SPADE generator: runs the REFERENCE's SPADEGenerator4 on CPU (build container only).

Writes tests/golden/spade_small.npz (reduced width, every block's output checksum + the full image) and
tests/golden/spade_full.npz (the shipped configuration at 256x256, batch 1: a 3x32x32 crop of the image,
float64 checksums of the image and of every block output)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

CASES = {
    "spade_small": (dict(ngf=8, nz=16, crop_size=64), 2),
    "spade_full": (dict(), 1),
}


def _checks(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


def gen_spade():
    from oracle import spade_ref
    sys.path.insert(0, "/root/reference")
    import models.SPADE_related as ref
    torch.set_num_threads(8)
    for name, (over, B) in CASES.items():
        cfg = spade_ref.SpadeConfig(**over)
        sd = spade_ref.init_state(cfg, seed=7)
        G = ref.SPADEGenerator4(cfg.semantic_nc, cfg.target_nc, cfg.nz, cfg.ngf, 'spectralspadelayer3x3', cfg.crop_size, 'normal')
        keys = set(G.state_dict().keys())
        assert keys == set(sd.keys()), sorted(keys ^ set(sd.keys()))[:10]
        G.load_state_dict(sd)
        G.eval()
        seg, z = spade_ref.synth_input(cfg, B, seed=3)
        taps = {}
        hooks = [getattr(G, n).register_forward_hook(lambda m, i, o, n=n: taps.__setitem__(n, o.clone())) for n, _, _ in cfg.blocks()]
        with torch.no_grad():
            out = G(seg, z)
        for h in hooks:
            h.remove()
        data = {"out_check": _checks(out)}
        for n, t in taps.items():
            data["check:" + n] = _checks(t)
        if name == "spade_small":
            data["out"] = out.numpy()
            data["tap:head_0"] = taps["head_0"].numpy()
        else:
            data["out_crop"] = out[:, :, 100:132, 60:92].numpy()
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **data)
        print("wrote", name, tuple(out.shape), "keys", len(keys), "out abs mean %.4f" % out.abs().mean().item())


if __name__ == "__main__":
    gen_spade()
