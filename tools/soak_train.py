"""Soak test on the GPU box: 3000 fused training steps (hipGraph replay) on one batch; the loss must fall and no memory may leak."""
import sys, time, importlib
sys.path.insert(0, "/root/repo")
import torch
M = importlib.import_module("3d_sln_amd.host.Sg2ScVAE_model"); syn = importlib.import_module("3d_sln_amd.host.synthetic")
torch.manual_seed(0)
model = M.Sg2ScVAEModel(vocab=syn.default_vocab(), batch_size=64, train_3d=True, decoder_cat=True, embedding_dim=64, gconv_mode='feedforward',
                        gconv_num_layers=5, mlp_normalization='batch', vec_noise_dim=0, layout_noise_dim=32, use_AE=False).cuda().train()
b = syn.scene_graph_batch(64, 32, 64, seed=1, device="cuda")
st = torch.cuda.Stream()
hist = []
mem0 = None
with torch.cuda.stream(st):
    for it in range(3000):
        l = model.train_step(b["objs"], b["triples"], b["boxes"], b["angles"], b["attributes"], kl_weight=0.1, lr=1e-4, use_graph=True)
        if it % 500 == 0 or it == 2999:
            v = [float(x) for x in l.cpu()]
            hist.append((it, v[3]))
            if mem0 is None: mem0 = torch.cuda.memory_allocated()
torch.cuda.synchronize()
print(hist, "mem growth", torch.cuda.memory_allocated() - mem0)
assert all(x == x for _, x in hist) and hist[-1][1] < hist[0][1]
print("soak ok")

# ---- a NEW batch every step (new tensors, same shape): hipGraph replay must train on the current batch like eager launches do
def run(use_graph, steps=400):
    torch.manual_seed(0)
    m = M.Sg2ScVAEModel(vocab=syn.default_vocab(), batch_size=64, train_3d=True, decoder_cat=True, embedding_dim=64, gconv_mode='feedforward',
                        gconv_num_layers=5, mlp_normalization='batch', vec_noise_dim=0, layout_noise_dim=32, use_AE=False).cuda().train()
    s = torch.cuda.Stream(); ls = []
    with torch.cuda.stream(s):
        for it in range(steps):
            bb = syn.scene_graph_batch(64, 32, 64, seed=100 + it, device="cuda")
            eps = torch.randn(bb["objs"].shape[0], 64, generator=torch.Generator().manual_seed(it)).cuda()
            ls.append(m.train_step(bb["objs"], bb["triples"], bb["boxes"], bb["angles"], bb["attributes"], kl_weight=0.1, lr=1e-4, eps=eps,
                                   use_graph=use_graph))
            del bb
    torch.cuda.synchronize()
    return torch.stack(ls)[:, 3].cpu()
le, lg = run(False), run(True)
print("fresh batches: eager first/last-50 mean %.4f / %.4f   graph %.4f / %.4f" % (float(le[:50].mean()), float(le[-50:].mean()), float(lg[:50].mean()), float(lg[-50:].mean())))
print("first steps eager", [round(float(x), 4) for x in le[:8]])
print("first steps graph", [round(float(x), 4) for x in lg[:8]])
# From the third step on two runs of the SAME mode already differ by 1 % (random initialisation: |logvar| ~ 16, the decoder input
# exp(logvar / 2) amplifies fp32 atomics-order noise); the trajectories are compared where they are comparable: the first two
# steps tightly, the level reached after 400 steps loosely.  Replaying a graph on the batch it was captured with (the bug this
# guards against) trains on ONE batch: its running loss ends far below the fresh-batch level.
assert float((le[:2] - lg[:2]).abs().max()) < 1e-4 * float(le[:2].abs().max())
assert abs(float(le[-50:].mean()) - float(lg[-50:].mean())) < 0.06 * float(le[-50:].mean())
print("fresh-batch soak ok")
