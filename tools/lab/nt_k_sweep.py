"""How much of a big NT launch is its K loop?  y = x W^T + b with column statistics, M = 131 072, N = 256, K swept (GPU box).
   python tools/lab/nt_k_sweep.py"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
L = importlib.import_module("3d_sln_amd._lib")
lib = L.lib()


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    st = L.current_stream_ptr()
    M, N = 131072, 256
    for stats in (True, False):
        for K in (128, 256, 512, 1024, 2048):
            x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
            y = torch.empty(M, N, device="cuda"); sums = torch.zeros(2, N, dtype=torch.float64, device="cuda")
            row = []
            for tile in (0, 1, 2):
                us = timeit(lambda: lib.sln_linear_forward(L.ptr(x), M, K, L.ptr(W), L.ptr(b), L.ptr(y), N, L.ptr(sums) if stats else None, tile, st))
                row.append("tile %d %7.1f us %5.1f TF %.3f" % (tile, us, 2.0 * M * N * K / us / 1e6, 2.0 * M * N * K / us / 1e6 / 157.3))
            print("stats %d K=%4d | " % (stats, K) + " | ".join(row))


if __name__ == "__main__":
    main()
