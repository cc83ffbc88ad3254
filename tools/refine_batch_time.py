"""Timing of the R-rooms-in-flight refinement loop (host/refine.py::RefineBatch) at train.py's defaults, 256 x 256: ms per iteration
(slope between runs of `iters` and 2 x `iters` iterations) and set-up per room, eager and under hipGraph replay.  GPU box."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
R = importlib.import_module("3d_sln_amd.host.refine")
M = importlib.import_module("3d_sln_amd.host.Sg2ScVAE_model")
syn = importlib.import_module("3d_sln_amd.host.synthetic")


def bench_rooms(n_rooms, n_obj=12, seed=0, dev="cuda"):
    names = ["bed", "chair", "table", "sofa", "desk", "cabinet", "lamp", "television", "bookshelf", "dresser", "night_stand", "shelves"]
    rooms = []
    for r in range(n_rooms):
        g = torch.Generator().manual_seed(seed + r)
        cn = names[:n_obj] + ["__room__"]; n = len(cn)
        lo = torch.rand(n, 3, generator=g) * 0.45 + 0.05; lo[:, 1] = 0.0; lo[:, 2] *= 0.6
        hi = lo + torch.rand(n, 3, generator=g) * 0.2 + 0.12
        boxes = torch.cat([lo, hi], 1); boxes[-1] = torch.tensor([0, 0, 0, 4.0, 2.7, 5.0])
        objs = torch.arange(1, n + 1); objs[-1] = 0
        tri = torch.tensor([[i, 1 + i % 10, (i + 1) % (n - 1)] for i in range(n - 1)] + [[i, 0, n - 1] for i in range(n - 1)])
        rooms.append(dict(objs=objs.to(dev), triples=tri.to(dev), boxes=boxes.to(dev), angles=torch.randint(0, 24, (n,), generator=g).to(dev),
                          attributes=torch.zeros(n, dtype=torch.int64, device=dev), class_names=cn))
    return rooms, names


def main():
    iters = int(os.environ.get("ITERS", "60"))
    torch.manual_seed(1)
    model = M.Sg2ScVAEModel(vocab=syn.default_vocab(), batch_size=1, train_3d=True, decoder_cat=True, embedding_dim=64, gconv_mode='feedforward',
                            gconv_num_layers=5, mlp_normalization='batch', vec_noise_dim=0, layout_noise_dim=32, use_AE=False).cuda().eval()
    st = torch.cuda.Stream()
    if not os.environ.get("NO_OVERFIT"):           # a trained checkpoint places the furniture in view; a random decoder renders the empty room
        with torch.cuda.stream(st):
            l = syn.overfit_to_rooms(model, bench_rooms(64)[0], steps=int(os.environ.get("OVERFIT_STEPS", "400")))
        print("over-fitted to 64 rooms: losses [bbox, angle, KL, total] =", [round(float(x), 4) for x in l], flush=True)
    for nr in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,4,16").split(",")]:
        rooms, names = bench_rooms(nr)
        bank = R.MeshBank(names, "cuda", seed=3)
        with torch.cuda.stream(st):
            for capture in (False, True):
                res = []; enq = []
                for n_it in (iters, 2 * iters, iters, 2 * iters, iters, 2 * iters):
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    G = int(os.environ.get("GROUPS", "1"))
                    rb = R.RefineBatch(model, rooms, bank=bank, iters=n_it)
                    torch.cuda.synchronize(); t1 = time.perf_counter()
                    rb.run(capture=capture)
                    t_enq = time.perf_counter() - t1                  # host time to enqueue everything (the GPU may still be running)
                    torch.cuda.synchronize(); t2 = time.perf_counter()
                    enq.append(t_enq / n_it)
                    lv = rb.live.cpu(); live = (float((lv == 3).sum()) / nr, float((lv == 1).sum()) / nr)
                    res.append((n_it, t1 - t0, t2 - t1)); info = rb.launches(); fin = bool(torch.isfinite(rb.losses).all()); rb.close()
                a = sorted(x[2] for x in res if x[0] == iters)[1]; b = sorted(x[2] for x in res if x[0] == 2 * iters)[1]
                setup = sorted(x[1] for x in res)[len(res) // 2]
                print("rooms %2d %s: %.3f ms / iteration (%.4f per room-iteration), run-intercept %.2f ms, set-up %.2f ms per room, %s, finite %s"
                      % (nr, "graph" if capture else "eager", (b - a) / iters * 1e3, (b - a) / iters * 1e3 / nr, (a - (b - a)) * 1e3, setup * 1e3 / nr, info, fin), flush=True)
                print("   host enqueue time per iteration: %.3f ms (min over runs); planes per room: %.1f live, %.1f constant" % (min(enq) * 1e3, live[0], live[1]), flush=True)


if __name__ == "__main__":
    main()
