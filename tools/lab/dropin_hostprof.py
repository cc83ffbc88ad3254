"""cProfile of the unchanged train.py:70-84 iteration on the aliased model (GPU box): the Python side of forward / backward / step.
   python tools/lab/dropin_hostprof.py"""
import cProfile, importlib, io, os, pstats, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
M = importlib.import_module("3d_sln_amd.host.Sg2ScVAE_model"); syn = importlib.import_module("3d_sln_amd.host.synthetic")
U = importlib.import_module("3d_sln_amd.host.utils")


class A:
    use_AE = False


ring = [syn.scene_graph_batch(64, 32, 64, seed=5000 + 7919 * k, device="cuda") for k in range(4)]
torch.manual_seed(42)
model = M.Sg2ScVAEModel(vocab=syn.default_vocab(), batch_size=64, train_3d=True, decoder_cat=True, embedding_dim=64, gconv_mode='feedforward',
                        gconv_num_layers=5, mlp_normalization='batch', vec_noise_dim=0, layout_noise_dim=32, use_AE=False).cuda().train()
model.validate_inputs = False
optimizer = torch.optim.Adam(model.parameters(), lr=1e-4)


def one(t):
    b = ring[t % 4]
    mu, logvar, boxes_pred, angles_pred = model(b["objs"], b["triples"], b["boxes"], b["angles"], b["attributes"], None)
    total_loss, losses = U.calculate_model_losses(A, model, b["boxes"], boxes_pred, b["angles"], angles_pred, mu=mu, logvar=logvar, KL_weight=0.1)
    losses['total_loss'] = total_loss.item()
    optimizer.zero_grad()
    total_loss.backward()
    optimizer.step()


st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for t in range(10):
        one(t)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for t in range(100):
        one(t)
    pr.disable(); torch.cuda.synchronize()
out = io.StringIO(); pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(28); print(out.getvalue()[:6500])
