import torch, time, os, sys, importlib
sys.path.insert(0, '/root/repo')
L = importlib.import_module('3d_sln_amd._lib'); S = importlib.import_module('3d_sln_amd.host.SPADE_related')
def run(B, Cin, C, H, W, mod):
    x = torch.randn(B, Cin, H, W, device='cuda')
    rows = 2 * C if mod else C
    w = torch.randn(rows, Cin, 3, 3, device='cuda') / (Cin * 9) ** 0.5
    wp, rp = S._pack(w)
    bp = torch.zeros(rp, device='cuda')
    y = torch.empty(B, C, H, W, device='cuda'); xin = torch.randn(B, C, H, W, device='cuda'); st = torch.ones(B, 2, device='cuda')
    def call():
        if mod:
            L.check(L.lib().sln_spade_modulate(L.ptr(x), B, Cin, H, W, L.ptr(wp), L.ptr(bp), C, rp, L.ptr(xin), L.ptr(st), 2, 0.2, L.ptr(y), L.current_stream_ptr()), 'm')
        else:
            L.check(L.lib().sln_spade_conv(L.ptr(x), B, Cin, H, W, L.ptr(wp), L.ptr(bp), C, rp, 3, 0, 0.0, L.ptr(y), L.current_stream_ptr()), 'c')
    for _ in range(3): call()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 2.0 * B * H * W * Cin * 9 * rows
    print("B%d Cin%d C%d %dx%d mod=%d rep=%s: %.3f ms  %.1f TF(1x)" % (B, Cin, C, H, W, mod, os.environ.get('SLN_REP', '0'), ms, fl / ms / 1e9))
for cfg in [(32, 128, 128, 256, 256, 1), (32, 128, 64, 256, 256, 0), (32, 128, 512, 64, 64, 1), (32, 1024, 512, 32, 32, 0)]:
    run(*cfg)
