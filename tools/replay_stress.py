import sys, numpy as np, torch, importlib, collections
sys.path.insert(0,'/root/repo')
from oracle import vae_ref
M = importlib.import_module("3d_sln_amd.host.Sg2ScVAE_model")
cfg = vae_ref.VaeConfig(embedding_dim=32, gconv_num_layers=2)
b = vae_ref.synth_batch(8, 12, 20, seed=3, cfg=cfg)
eps = torch.randn(b[0].shape[0], cfg.embedding_dim, generator=torch.Generator().manual_seed(0))
sd0 = vae_ref.init_state(cfg, seed=1)
def run(use_graph, steps=3):
    m = M.Sg2ScVAEModel(**cfg.model_kwargs()); m.load_state_dict({k: v.clone() for k, v in sd0.items()}); m = m.cuda().train()
    dev = [t.cuda() for t in b[:5]] + [eps.cuda()]
    s = torch.cuda.Stream(); ls = []
    with torch.cuda.stream(s):
        for _ in range(steps):
            ls.append(m.train_step(*dev[:5], kl_weight=0.1, lr=1e-3, eps=dev[5], use_graph=use_graph).clone())
    torch.cuda.synchronize()
    return tuple(round(float(l[3]), 5) for l in ls)
for mode in (False, True):
    c = collections.Counter(run(mode) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30))
    print("graph" if mode else "eager", dict(c))
