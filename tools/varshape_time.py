"""Training steps on batches whose (O, T) changes every step, as real rooms do (train.py with a dataset): eager launches vs the
fixed-shape hipGraph step.  GPU box:  python tools/varshape_time.py"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
M = importlib.import_module("3d_sln_amd.host.Sg2ScVAE_model")
syn = importlib.import_module("3d_sln_amd.host.synthetic")
from oracle import vae_ref


def main():
    cfg = vae_ref.VaeConfig()
    torch.manual_seed(0)
    model = M.Sg2ScVAEModel(**cfg.model_kwargs()).cuda().train()
    model.validate_inputs = False            # as host/train.py does: no host sync per new batch
    sizes = [64, 61, 66, 59, 63, 67, 60, 65]
    batches = []
    for i, g in enumerate(sizes):
        b = vae_ref.synth_batch(g, 32, 64, seed=10 + i, cfg=cfg)
        batches.append([t.cuda() for t in b[:5]])
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for mode in ("fixed shape, graph", "fixed shape, eager", "varying shape, eager", "varying shape, graph"):
            graph = "graph" in mode
            pick = (lambda k: batches[0]) if "fixed" in mode else (lambda k: batches[k % len(batches)])
            for k in range(10):
                model.train_step(*pick(k), kl_weight=0.1, lr=1e-4, use_graph=graph)
            torch.cuda.synchronize()
            n = 100
            t0 = time.perf_counter()
            for k in range(n):
                model.train_step(*pick(k), kl_weight=0.1, lr=1e-4, use_graph=graph)
            torch.cuda.synchronize()
            print("%-24s %.3f ms/step" % (mode, (time.perf_counter() - t0) / n * 1e3))


if __name__ == "__main__":
    main()
