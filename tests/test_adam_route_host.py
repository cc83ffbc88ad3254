"""Host logic of the torch.optim.Adam routing (host/Sg2ScVAE_model.py::_fast_adam_owner) - runs without a GPU: which optimizers are
recognised as "train.py:15's Adam over exactly model.parameters()", and that the process-wide hooks leave every other optimizer
(and the CPU-resident model's own optimizer) on torch's path."""
import torch

from conftest import pkg
from oracle import vae_ref


def _model():
    M = pkg("host.Sg2ScVAE_model")
    cfg = vae_ref.VaeConfig(embedding_dim=16, gconv_num_layers=1)
    return M, M.Sg2ScVAEModel(**cfg.model_kwargs())


def test_only_the_reference_configuration_is_recognised():
    M, model = _model()
    ps = list(model.parameters())
    assert M._fast_adam_owner(torch.optim.Adam(ps, lr=1e-4)) is model
    assert M._fast_adam_owner(torch.optim.Adam(ps, lr=3e-4)) is model                     # any learning rate
    assert M._fast_adam_owner(torch.optim.Adam(ps, lr=1e-4, weight_decay=1e-2)) is None
    assert M._fast_adam_owner(torch.optim.Adam(ps, lr=1e-4, betas=(0.5, 0.999))) is None
    assert M._fast_adam_owner(torch.optim.Adam(ps, lr=1e-4, amsgrad=True)) is None
    assert M._fast_adam_owner(torch.optim.Adam(ps[:-1], lr=1e-4)) is None                 # a parameter subset
    assert M._fast_adam_owner(torch.optim.Adam(list(reversed(ps)), lr=1e-4)) is None      # another order
    assert M._fast_adam_owner(torch.optim.Adam([{"params": ps[:5]}, {"params": ps[5:]}], lr=1e-4)) is None
    assert M._fast_adam_owner(torch.optim.AdamW(ps, lr=1e-4, weight_decay=0.0)) is None
    assert M._fast_adam_owner(torch.optim.SGD(ps, lr=1e-4)) is None
    model.route_torch_adam = False
    assert M._fast_adam_owner(torch.optim.Adam(ps, lr=1e-4)) is None


def test_hooks_leave_other_optimizers_and_a_cpu_model_alone():
    M, model = _model()
    w = torch.nn.Parameter(torch.ones(3))
    opt = torch.optim.Adam([w], lr=0.1)
    w.grad = torch.ones(3)
    opt.step()
    assert torch.allclose(w.detach(), torch.full((3,), 0.9)) and len(opt.param_groups[0]["params"]) == 1
    # the model on the CPU (no engine, no GPU): torch's own Adam steps its parameters - and they stay views of the flat buffer
    ps = list(model.parameters())
    opt2 = torch.optim.Adam(ps, lr=1e-2)
    for p in ps:
        p.grad = torch.ones_like(p)
    before = model.flat_params.clone()
    opt2.step()
    assert len(opt2.param_groups[0]["params"]) == len(ps) and len(opt2.state) == len(ps)
    assert float((model.flat_params - before).abs().max()) > 5e-3                         # the views moved the flat buffer
