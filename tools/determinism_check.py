import sys, numpy as np, torch, importlib
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from oracle import vae_ref
M = importlib.import_module("3d_sln_amd.host.Sg2ScVAE_model")
for cfg, nb in ((vae_ref.VaeConfig(embedding_dim=32, gconv_num_layers=2), (8,12,20)), (vae_ref.VaeConfig(), (64,32,64))):
    sd = vae_ref.init_state(cfg, seed=1)
    b = vae_ref.synth_batch(*nb, seed=3, cfg=cfg)
    eps = torch.randn(b[0].shape[0], cfg.embedding_dim, generator=torch.Generator().manual_seed(0)).cuda()
    m = M.Sg2ScVAEModel(**cfg.model_kwargs()); m.load_state_dict({k:v.clone() for k,v in sd.items()}); m=m.cuda().train()
    dev=[t.cuda() for t in b[:5]]
    outs=[]
    for rep in range(4):
        with torch.no_grad():
            o = m(*dev, None, eps=eps)
        l = m.train_step(*dev, kl_weight=0.1, lr=1e-3, eps=eps, use_graph=False, with_adam=False)
        outs.append(([x.cpu().numpy().copy() for x in o], l.cpu().numpy().copy(), m.flat_grads.cpu().numpy().copy()))
        m.load_state_dict({k:v.clone() for k,v in sd.items()})
    for rep in range(1,4):
        fo = max(float(np.abs(a-b_).max()) for a,b_ in zip(outs[0][0], outs[rep][0]))
        gs = float(np.abs(outs[0][2]).max())
        print(nb, "rep", rep, "fwd max diff %.3e" % fo, "loss diff %.3e" % float(np.abs(outs[0][1]-outs[rep][1]).max()), "grad diff rel %.3e" % (float(np.abs(outs[0][2]-outs[rep][2]).max())/gs))
