import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle import vae_ref
M = importlib.import_module("3d_sln_amd.host.Sg2ScVAE_model")
cfg = vae_ref.VaeConfig(embedding_dim=32, gconv_num_layers=2)
sd = vae_ref.init_state(cfg, seed=5)
sizes = [(9, 8, 12), (7, 9, 14), (12, 6, 9), (5, 8, 12), (11, 7, 13)]
if len(sys.argv) > 1 and sys.argv[1] == "bigfirst":
    sizes = [(11, 7, 13)] + sizes[:4]
batches = [[t.cuda() for t in vae_ref.synth_batch(g, o, t, seed=20 + i, cfg=cfg)[:5]] for i, (g, o, t) in enumerate(sizes)]
n_steps = 30
eps = [torch.from_numpy(np.random.default_rng(100 + k).standard_normal((batches[k % 5][0].shape[0], cfg.embedding_dim)).astype(np.float32)).cuda() for k in range(n_steps)]
runs = {}
for drain in (True, False, False):
    model = M.Sg2ScVAEModel(**cfg.model_kwargs()); model.load_state_dict({k: v.clone() for k, v in sd.items()}); model = model.cuda().train()
    model.validate_inputs = False
    st = torch.cuda.Stream(); losses = []
    with torch.cuda.stream(st):
        for k in range(n_steps):
            losses.append(model.train_step(*batches[k % 5], kl_weight=0.1, lr=1e-3, eps=eps[k], use_graph=False))
            if drain: torch.cuda.synchronize()
    torch.cuda.synchronize()
    L = torch.stack(losses).cpu().numpy()
    if drain: ref = L
    else:
        d = np.abs(L - ref).max(1) / np.abs(ref).max(1)
        print("first step with rel diff > 1e-5:", int(np.argmax(d > 1e-5)) if (d > 1e-5).any() else None, "max", d.max(), np.round(d[:12], 6))
