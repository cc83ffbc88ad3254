"""Lab: gradient error of a train-mode-BatchNorm step against the fp64 oracle as a function of the number of rows (graphs of 8 objects /
12 triples), threshold-free state (tests/test_vae_gpu.py::_threshold_free_state(train_bn=True)).  GPU box."""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import vae_ref
t = importlib.import_module("test_vae_gpu")
for n in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4,16").split(",")]:
    cfg = vae_ref.VaeConfig(mlp_normalization="batch")
    sd = t._threshold_free_state(cfg, 3, train_bn=True)
    batch = list(vae_ref.synth_batch(n, 8, 12, seed=0, cfg=cfg)[:5])
    sg = torch.from_numpy(np.random.default_rng(7).integers(0, 2, tuple(batch[2].shape)).astype(np.float32)) * 2 - 1
    batch[2] = batch[2] + 20.0 * sg
    O = batch[0].shape[0]
    eps = torch.from_numpy(np.random.default_rng(1).standard_normal((O, cfg.embedding_dim)).astype(np.float32))
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    b64 = (batch[0], batch[1], batch[2].double(), batch[3], batch[4])
    keys = vae_ref.trainable_keys(cfg)
    m = {k: torch.zeros_like(sd64[k]) for k in keys}; v = {k: torch.zeros_like(sd64[k]) for k in keys}
    tot64, _, g64 = vae_ref.train_step({k: x.clone() for k, x in sd64.items()}, cfg, b64, eps.double(), 0.1, m, v, step=1, training=True)
    m = {k: torch.zeros_like(sd[k]) for k in keys}; v = {k: torch.zeros_like(sd[k]) for k in keys}
    tot32, _, g32 = vae_ref.train_step({k: x.clone() for k, x in sd.items()}, cfg, batch, eps, 0.1, m, v, step=1, training=True)
    model = t._model(cfg, sd).train()
    dev = t._dev(*batch, eps)
    # forward taps first: where does the forward drift start?
    (_, trace) = t._trace_oracle({k: x.clone() for k, x in sd64.items()}, cfg, b64, eps.double(), True)
    losses = model.train_step(*dev[:5], kl_weight=0.1, lr=1e-4, eps=dev[5], use_graph=False, with_adam=False).cpu().numpy()
    print("n=%d total: hip %.8f fp32-oracle %.8f fp64 %.8f" % (n, losses[3], float(tot32), float(tot64)))
    print(t._tap_report(model, cfg, trace)[:1800])
    named = dict(model.named_parameters())
    rows = []
    for k in keys:
        r = g64[k].numpy(); sc = np.abs(r).max()
        if sc < 1e-12:
            continue
        rows.append((np.abs(named[k].grad.cpu().numpy() - r).max() / sc, np.abs(g32[k].numpy() - r).max() / sc, k))
    rows.sort(reverse=True)
    print("n=%d worst hip rel err %.2e (fp32 oracle on that key %.2e) %s; median hip %.2e, median fp32-oracle %.2e" %
          (n, rows[0][0], rows[0][1], rows[0][2], np.median([r[0] for r in rows]), np.median([r[1] for r in rows])))
    for r in rows[:6] + rows[-3:]:
        print("   %.2e  %.2e  %s" % r)
