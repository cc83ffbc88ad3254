import importlib, sys, types, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from oracle import vae_ref
M=importlib.import_module('3d_sln_amd.host.Sg2ScVAE_model')
U=importlib.import_module('3d_sln_amd.host.utils')
cfg = vae_ref.VaeConfig(embedding_dim=32, gconv_num_layers=2, mlp_normalization="none")
sd = vae_ref.init_state(cfg, seed=11)
model = M.Sg2ScVAEModel(**cfg.model_kwargs()); model.load_state_dict({k: v.clone() for k, v in sd.items()}); model = model.cuda().train()
batch = vae_ref.synth_batch(6, 9, 14, seed=2, cfg=cfg)
dev = [t.cuda() for t in batch[:5]]
eps = torch.randn(batch[0].shape[0], cfg.embedding_dim).cuda()
opt = torch.optim.Adam(model.parameters(), lr=1e-3)
out = model(*dev, None, eps=eps)
total, _ = U.calculate_model_losses(types.SimpleNamespace(use_AE=False), model, dev[2], out[2], dev[3], out[3], mu=out[0], logvar=out[1], KL_weight=0.1)
opt.zero_grad(); total.backward()
m = M._fast_adam_owner(opt)
print("owner", m is model, "eng", model._eng is not None, model._flat.device)
bad = [(i, p.grad is None, None if p.grad is None else p.grad.data_ptr() == gv.data_ptr()) for i, (p, gv) in enumerate(zip(model._params, model._gviews)) if p.grad is None or p.grad.data_ptr() != gv.data_ptr()]
print("params not on their views:", bad[:10], len(bad))
import torch.optim.optimizer as O
print(len(O._global_optimizer_pre_hooks))
opt.step()
print("steps", model._adam_steps, "stale", opt.__dict__.get('_sln_steps_stale'))
