"""Host time per phase of the unchanged train.py:70-84 iteration on the aliased model (GPU box): where the 0.75 ms on top of the
fused step go.      python tools/lab/dropin_phases.py"""
import importlib, math, os, sys, time
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
names = ["forward", "losses (3 item)", "total.item", "zero_grad", "backward", "step"]
acc = [0.0] * len(names)


def one(t, rec):
    b = ring[t % 4]
    t0 = time.perf_counter()
    mu, logvar, boxes_pred, angles_pred = model(b["objs"], b["triples"], b["boxes"], b["angles"], b["attributes"], None)
    t1 = time.perf_counter()
    total_loss, losses = U.calculate_model_losses(A, model, b["boxes"], boxes_pred, b["angles"], angles_pred, mu=mu, logvar=logvar, KL_weight=0.1)
    t2 = time.perf_counter()
    losses['total_loss'] = total_loss.item()
    t3 = time.perf_counter()
    optimizer.zero_grad()
    t4 = time.perf_counter()
    total_loss.backward()
    t5 = time.perf_counter()
    optimizer.step()
    t6 = time.perf_counter()
    if rec:
        for i, (a, b_) in enumerate(((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5), (t5, t6))):
            acc[i] += b_ - a


st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for t in range(10):
        one(t, False)
    torch.cuda.synchronize()
    T0 = time.perf_counter()
    N = 100
    for t in range(N):
        one(t, True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - T0) / N
print("%.3f ms per step" % (dt * 1e3))
for n, a in zip(names, acc):
    print("  %-18s %.3f ms (host time until the call returns)" % (n, a / N * 1e3))
