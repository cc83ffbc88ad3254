"""Micro-benchmark of the fused fp32 GEMM family on the VAE's shapes (GPU box only).
   python tools/gemm_bench.py            -> forward (NT) per tile config, wgrad (TN)"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
L = importlib.import_module("3d_sln_amd._lib")
lib = L.lib()


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3   # us


def main():
    st = L.current_stream_ptr()
    print("== NT forward y = x W^T + b (+stats)")
    for (M, N, K) in [(4096, 256, 384), (4096, 640, 256), (4096, 256, 640), (4096, 384, 256), (2048, 256, 256),
                      (2048, 128, 256), (2048, 256, 128), (32768, 640, 256)]:
        x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
        y = torch.empty(M, N, device="cuda"); sums = torch.zeros(2, N, dtype=torch.float64, device="cuda")
        row = []
        for tile in (-1, 0):
            for stats in (True, False):
                f = lambda: lib.sln_linear_forward(L.ptr(x), M, K, L.ptr(W), L.ptr(b), L.ptr(y), N,
                                                   L.ptr(sums) if stats else None, tile, st)
                us = timeit(f)
                row.append("t%+d%s %6.1fus %5.1fTF" % (tile, "+stats" if stats else " plain", us, 2.0 * M * N * K / us / 1e6))
        print("M=%5d N=%4d K=%4d | " % (M, N, K) + " | ".join(row))
    print("== TN wgrad dW += g^T x")
    for (R, N, K) in [(4096, 640, 256), (4096, 256, 384), (2048, 256, 256), (2048, 128, 256), (32768, 640, 256)]:
        g = torch.randn(R, N, device="cuda"); x = torch.randn(R, K, device="cuda")
        dW = torch.zeros(N, K, device="cuda"); db = torch.zeros(N, device="cuda")
        f = lambda: lib.sln_linear_wgrad(L.ptr(g), L.ptr(x), R, N, K, L.ptr(dW), L.ptr(db), st)
        us = timeit(f)
        print("R=%5d N=%4d K=%4d | %6.1fus %5.1fTF" % (R, N, K, us, 2.0 * R * N * K / us / 1e6))


if __name__ == "__main__":
    main()
