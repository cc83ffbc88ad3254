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
