"""Time the posterior heat-map application (reference testing/test_heatmap.py: 20 000 single-graph decodes) on the GPU box."""
import sys, time, importlib
sys.path.insert(0, "/root/repo")
import torch
M = importlib.import_module("3d_sln_amd.host.Sg2ScVAE_model"); syn = importlib.import_module("3d_sln_amd.host.synthetic")
S = importlib.import_module("3d_sln_amd.host.sampling")
torch.manual_seed(0)
model = M.Sg2ScVAEModel(vocab=syn.default_vocab(), batch_size=1, train_3d=True, decoder_cat=True, embedding_dim=64, gconv_mode='feedforward',
                        gconv_num_layers=5, mlp_normalization='batch', vec_noise_dim=0, layout_noise_dim=32, use_AE=False).cuda().eval()
objs5 = ["bed", "desk", "cabinet", "chair", "lamp"]
rels5 = [("bed", "behind", "desk"), ("cabinet", "left of", "bed"), ("chair", "left of", "desk"), ("lamp", "on", "desk")]
mean = torch.zeros(64, dtype=torch.float64); cov = torch.eye(64, dtype=torch.float64)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    S.heatmap_from_words(model, objs5, rels5, mean, cov, num_iter=2000, chunk=2000); torch.cuda.synchronize()
    for chunk in (500, 2000, 10000):
        t0 = time.perf_counter()
        h = S.heatmap_from_words(model, objs5, rels5, mean, cov, num_iter=20000, chunk=chunk)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("20000 layouts, chunk %5d: %.3f s (%.0f layouts/s)" % (chunk, dt, 20000 / dt))
