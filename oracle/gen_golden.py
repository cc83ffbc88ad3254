"""Generate golden fixtures under tests/golden/ by running the REFERENCE on CPU.

Runs ONLY in the build container (needs /root/reference, read-only).  It imports
the reference's own ``models.graph`` / ``models.Sg2ScVAE_model`` / ``utils`` /
``models.SPADE_related`` modules, loads the deterministic state produced by the
oracle's ``init_state`` (keys must match exactly - that is itself a boundary
check), runs seeded synthetic inputs and stores inputs/outputs/gradients as
small ``.npz`` files.  Only data is written; no reference source travels.

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [vae|spade|all]
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


def _import_reference():
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present; fixtures can only be regenerated in the build container")
    sys.path.insert(0, REF)
    import models.graph as ref_graph                      # noqa: E402
    import models.Sg2ScVAE_model as ref_vae               # noqa: E402
    import utils as ref_utils                             # noqa: E402
    return ref_graph, ref_vae, ref_utils


# --------------------------------------------------------------------------
# VAE cases: name -> (config overrides, graphs, objs/graph, triples/graph)
# --------------------------------------------------------------------------
VAE_CASES = {
    # BASELINE.json configs[0]: one 8-object / 12-triple graph at train.py defaults
    "vae_c1_full": (dict(), 1, 8, 12),
    # reduced width, 2 layers: small files, every code path
    "vae_small_batch": (dict(embedding_dim=16, gconv_num_layers=2), 16, 6, 9),
    "vae_small_none": (dict(embedding_dim=16, gconv_num_layers=2, mlp_normalization="none"), 16, 6, 9),
    "vae_small_recurrent": (dict(embedding_dim=16, gconv_num_layers=3, gconv_mode="recurrent"), 16, 5, 8),
    "vae_small_nocat": (dict(embedding_dim=16, gconv_num_layers=2, decoder_cat=False), 16, 5, 8),
    # use_attr=False (Sg2ScVAE_model.py:17,35-37,48-50,97-98), with z in front of / behind the decoder's gconv net
    "vae_small_noattr": (dict(embedding_dim=16, gconv_num_layers=2, use_attr=False), 16, 5, 8),
    "vae_small_nocat_noattr": (dict(embedding_dim=16, gconv_num_layers=2, decoder_cat=False, use_attr=False), 16, 5, 8),
    # decoder_cat=False at a width whose segments are multiples of the GEMM's K tile (the aligned operand loader)
    "vae_e32_nocat": (dict(embedding_dim=32, gconv_num_layers=1, decoder_cat=False), 8, 5, 8),
    "vae_small_ae": (dict(embedding_dim=16, gconv_num_layers=2, use_AE=True), 16, 5, 8),
    "vae_small_2d": (dict(embedding_dim=16, gconv_num_layers=2, train_3d=False), 16, 5, 8),
    # ragged: graphs of different sizes concatenated like suncg_collate_fn does
    "vae_small_ragged": (dict(embedding_dim=16, gconv_num_layers=2), -1, 0, 0),
}
KL_WEIGHT = 0.1


def _ragged_batch(cfg, seed):
    from oracle import vae_ref
    parts = [vae_ref.synth_batch(1, n, t, seed + i, cfg) for i, (n, t) in enumerate([(3, 2), (9, 20), (2, 1), (6, 7), (12, 30), (5, 4), (20, 44), (7, 12), (4, 3), (15, 15)])]
    objs, trip, boxes, angles, attrs, o2i = [], [], [], [], [], []
    off = 0
    for gi, (ob, tr, bx, an, at, _) in enumerate(parts):
        tr = tr.clone(); tr[:, 0] += off; tr[:, 2] += off
        objs.append(ob); trip.append(tr); boxes.append(bx); angles.append(an); attrs.append(at)
        o2i.append(torch.full((ob.shape[0],), gi, dtype=torch.int64))
        off += ob.shape[0]
    return tuple(torch.cat(x) for x in (objs, trip, boxes, angles, attrs, o2i))


def gen_vae(only=None):
    from oracle import vae_ref
    ref_graph, ref_vae, ref_utils = _import_reference()
    for name, (over, B, n, tt) in VAE_CASES.items():
        if only and name not in only:
            continue
        cfg = vae_ref.VaeConfig(**over)
        sd0 = vae_ref.init_state(cfg, seed=42)
        batch = _ragged_batch(cfg, 7) if B < 0 else vae_ref.synth_batch(B, n, tt, seed=0, cfg=cfg)
        objs, triples, boxes, angles, attrs, o2i = batch
        O = objs.shape[0]
        eps = torch.from_numpy(np.random.default_rng(1).standard_normal((O, cfg.embedding_dim)).astype(np.float32))

        model = ref_vae.Sg2ScVAEModel(**cfg.model_kwargs())
        ref_keys = set(model.state_dict().keys())
        assert ref_keys == set(sd0.keys()), (sorted(ref_keys ^ set(sd0.keys())))
        model.load_state_dict({k: v.clone() for k, v in sd0.items()})
        out = {}
        # --- eval-mode pass (running statistics) -------------------------------
        model.eval()
        with torch.no_grad():
            mu_e, lv_e = model.encoder(objs, triples, boxes, angles, attrs)
            z_e = mu_e if cfg.use_AE else eps * torch.exp(0.5 * lv_e) + mu_e
            bp_e, ap_e = model.decoder(z_e, objs, triples, attrs)
        out.update(eval_mu=mu_e, eval_logvar=lv_e, eval_boxes_pred=bp_e, eval_angles_pred=ap_e)
        # --- train-mode pass with the N(0,1) draw pinned -------------------------
        model.train()
        mu, lv = model.encoder(objs, triples, boxes, angles, attrs)
        z = mu if cfg.use_AE else eps.mul(torch.exp(0.5 * lv)).add_(mu)
        bp, ap = model.decoder(z, objs, triples, attrs)
        args = types.SimpleNamespace(use_AE=cfg.use_AE)
        total, parts = ref_utils.calculate_model_losses(args, model, boxes, bp, angles, ap, mu=mu, logvar=lv,
                                                        KL_weight=KL_WEIGHT)
        model.zero_grad()
        total.backward()
        out.update(mu=mu, logvar=lv, boxes_pred=bp, angles_pred=ap, total_loss=total)
        for k, v in parts.items():
            out["loss_" + k] = torch.tensor(v)
        for k, p in model.named_parameters():
            out["grad:" + k] = p.grad if p.grad is not None else torch.zeros_like(p)
        for k, b in model.named_buffers():
            out["buf:" + k] = b
        # --- one Adam step exactly as train.py:15,82-84 --------------------------
        opt = torch.optim.Adam(model.parameters(), lr=1e-4)
        opt.step()
        big = name.endswith("_full")
        for k, p in model.named_parameters():
            if k.startswith(("box_net", "angle_net", "box_embeddings", "gconv_net_dc.gconvs.0.net2", "obj_embeddings_ec")):
                out["adam:" + k] = p
        if big:      # keep the full-width fixture small: drop the big gradient tensors, keep checksums
            for k in [k for k in out if k.startswith("grad:")]:
                g = out[k].detach().double()
                out["gsum:" + k[5:]] = torch.stack([g.sum(), g.abs().sum(), (g * g).sum()])
                if g.numel() > 4096:
                    del out[k]
        inputs = dict(objs=objs, triples=triples, boxes=boxes, angles=angles, attrs=attrs, eps=eps)
        np.savez_compressed(os.path.join(GOLD, name + ".npz"),
                            **{"in:" + k: v.numpy() for k, v in inputs.items()},
                            **{k: v.detach().numpy() for k, v in out.items()})
        print("wrote", name, "O=%d T=%d" % (O, triples.shape[0]), "loss=%.6f" % float(total.detach()))


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(1)
    if what in ("vae", "all"):
        gen_vae(set(sys.argv[2].split(",")) if len(sys.argv) > 2 else None)        # optional: only these cases
    if what in ("spade", "all"):
        from oracle import gen_golden_spade
        gen_golden_spade.gen_spade()
