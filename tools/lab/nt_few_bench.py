"""Launch-to-launch time of the split-K NT body (y = x W^T + b with column statistics) on graphs of a few rows, GPU box: 100 dependent
launches in a replayed hipGraph.  Round 4 used it for a variant with five register stages per wavefront (all tiles of a wave
requested at once, 246-256 registers): 3.78 / 4.75 / 5.63 / 7.33 us against 3.59 / 4.43 / 5.22 / 6.64 us for K = 128 / 256 / 384 /
640 - slower, dropped.      python tools/lab/nt_few_bench.py"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
L = importlib.import_module("3d_sln_amd._lib")
lib = L.lib()


def main():
    for (M, N, K) in [(24, 256, 384), (24, 640, 256), (24, 256, 640), (13, 256, 256), (13, 128, 256), (24, 256, 128)]:
        x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); b = torch.randn(N, device="cuda")
        y = torch.empty(M, N, device="cuda"); sums = torch.zeros(2, N, dtype=torch.float64, device="cuda")
        f = lambda: lib.sln_linear_forward(L.ptr(x), M, K, L.ptr(W), L.ptr(b), L.ptr(y), N, L.ptr(sums), -1, L.current_stream_ptr())
        for _ in range(20):
            f()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(100):
                f()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            s.record(); g.replay(); e.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) * 10.0)
        print("M=%3d N=%4d K=%4d  %.2f us per launch (100 dependent launches in a replayed graph)" % (M, N, K, best))


if __name__ == "__main__":
    main()
