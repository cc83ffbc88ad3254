"""Where does the fp32 rounding error of the full-size SPADE generator arise?  (GPU box)

    python tools/spade_error_budget.py [default|unit|oracle]       weights: torch's default init (= bench.py), N(0, 1/fan_in), oracle seed 7

Three tables, every error = max |a - b| / max |b| against an fp64 evaluation of oracle/spade_ref.py on the same weights / inputs:
  1. ACCUMULATED: the block outputs (taps) and the image of one forward - HIP, CPU fp32 (oracle/spade_ref.py, what the reference
     module computes), HIP against CPU fp32;
  2. LOCAL: every block (and the last conv + tanh) on its own, fed with the fp64 run's input of that block rounded to fp32 -
     the error a stage ADDS, free of what it inherits;
  3. ONE CONVOLUTION at the generator's K = 9 Cin (random operands): the MFMA accumulation chain against torch's CPU fp32 conv.
"""
import importlib, os, sys
import numpy as np, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import spade_ref
_lib = importlib.import_module("3d_sln_amd._lib")
S = importlib.import_module("3d_sln_amd.host.SPADE_related")

mode = sys.argv[1] if len(sys.argv) > 1 else "default"
if os.environ.get("SLN_BUDGET_SIGMA64") == "1":          # experiment: spectral sigma = u . (W v) evaluated in fp64 (an ill-conditioned sum)
    def _fold64(sd, prefix):
        w = sd[prefix + ".weight_orig"]
        sigma = torch.dot(sd[prefix + ".weight_u"].double(), torch.mv(w.reshape(w.shape[0], -1).double(), sd[prefix + ".weight_v"].double()))
        return (w.double() / sigma).float()
    S._fold_sn = _fold64
if os.environ.get("SLN_BUDGET_SIGMA64") == "0":          # experiment: the historical fp32 fold
    def _fold32(sd, prefix):
        w = sd[prefix + ".weight_orig"]
        sigma = torch.dot(sd[prefix + ".weight_u"], torch.mv(w.reshape(w.shape[0], -1), sd[prefix + ".weight_v"]))
        return w / sigma
    S._fold_sn = _fold32
torch.manual_seed(0)
cfg = spade_ref.SpadeConfig()
G = S.SPADEGenerator4(41, 3, 256, 64, 'spectralspadelayer3x3', 256, 'normal')
if mode == "unit":
    with torch.no_grad():
        for name, p in G.named_parameters():
            if p.dim() > 1:
                p.normal_(0.0, 1.0 / float(p[0].numel()) ** 0.5)
                if name.startswith("conv_img"):
                    p.mul_(0.15)
            else:
                p.normal_(0.0, 0.05)
elif mode == "oracle":
    G.load_state_dict(spade_ref.init_state(cfg, seed=7))
G = G.cuda().eval()
# bench.py's input (spade_leg), image 0
g = torch.Generator(device="cuda").manual_seed(0)
B = 32
low = torch.rand(B, 1, 16, 16, device="cuda", generator=g) * 2 - 1
depth = F.interpolate(low, size=(256, 256), mode="bilinear", align_corners=False)
lab = F.interpolate(torch.randn(B, 40, 16, 16, device="cuda", generator=g), size=(256, 256), mode="bilinear", align_corners=False).argmax(1)
seg = torch.cat([depth, F.one_hot(lab, 40).permute(0, 3, 1, 2).float()], 1).contiguous()[:1]
z = torch.randn(B, 256, device="cuda", generator=g)[:1]


def e(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max())


taps = {}
with torch.no_grad():
    out = G(seg, z, taps=taps).cpu()
sd = {k: v.detach().cpu() for k, v in G.state_dict().items()}
sd64 = {k: v.double() for k, v in sd.items()}
torch.set_num_threads(min(16, os.cpu_count() or 1))
t32, t64 = {}, {}
segc, zc = seg.cpu(), z.cpu()
with torch.no_grad():
    r32 = spade_ref.generator(sd, cfg, segc, zc, t32)
    r64 = spade_ref.generator(sd64, cfg, segc.double(), zc.double(), t64)
print("== 1. accumulated error (weights: %s)" % mode)
print("%-12s %12s %12s %12s %10s" % ("stage", "hip-vs-64", "cpu32-vs-64", "hip-vs-cpu32", "scale"))
for k in t64:
    print("%-12s %12.2e %12.2e %12.2e %10.2e" % (k, e(taps[k], t64[k]), e(t32[k], t64[k]), e(taps[k], t32[k]), float(t64[k].abs().max())))
print("%-12s %12.2e %12.2e %12.2e %10.2e" % ("image", e(out, r64), e(r32, r64), e(out, r32), float(r64.abs().max())))

# ---- 2. local error per block
print("== 2. local error: each stage on the fp64 run's input of that stage (rounded to fp32)")
print("%-12s %12s %12s %10s" % ("stage", "hip-vs-64", "cpu32-vs-64", "ratio"))
blocks = dict((n, (a, b)) for n, a, b in cfg.blocks())
up_n = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
up_b = lambda t: F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=False)
x0_64 = F.linear(zc.double(), sd64["fc.weight"], sd64["fc.bias"]).view(-1, 16 * cfg.ngf, cfg.sw, cfg.sw)
inputs64 = {"head_0": x0_64, "G_middle_0": up_n(t64["head_0"]), "G_middle_1": t64["G_middle_0"], "up_0": up_n(t64["G_middle_1"]),
            "up_1": up_n(t64["up_0"]), "up_2": up_n(t64["up_1"]), "up_3": up_b(t64["up_2"])}
G._packed = G._pack_all()
G._map_repeat = False
with torch.no_grad():
    for name in inputs64:
        xin64 = inputs64[name]
        xin32 = xin64.float()
        H = xin64.shape[2]
        s_level = F.interpolate(segc, size=(H, H)) if name == "head_0" else F.interpolate(segc, size=(H, H), mode="bilinear", align_corners=False)
        # the fp64 result of the block on the ROUNDED input (so that only the block's own arithmetic differs)
        ref = spade_ref.resblock(sd64, name, xin32.double(), segc.double() if name != "head_0" else F.interpolate(segc.double(), size=(H, H)),
                                 *blocks[name])
        c32 = spade_ref.resblock(sd, name, xin32, segc if name != "head_0" else F.interpolate(segc, size=(H, H)), *blocks[name])
        G._cat_cache = {}
        xg = xin32.cuda().contiguous()
        hip, _, _ = G._block(name, xg, False, G._ln_stats(xg), s_level.cuda().contiguous(), None, want_stats=False)
        eh, ec = e(hip, ref), e(c32, ref)
        print("%-12s %12.2e %12.2e %10.2f" % (name, eh, ec, eh / ec))
    G._cat_cache = None
    x7 = t64["up_3"].float()
    ref = torch.tanh(F.conv2d(F.leaky_relu(x7.double(), 0.2), sd64["conv_img.weight"], sd64["conv_img.bias"], padding=2))
    c32 = torch.tanh(F.conv2d(F.leaky_relu(x7, 0.2), sd["conv_img.weight"], sd["conv_img.bias"], padding=2))
    hip = torch.empty(1, 3, 256, 256, device="cuda")
    P = G._packed
    xg = x7.cuda().contiguous()
    _lib.check(_lib.lib().sln_conv_img_tanh(_lib.ptr(xg), 1, cfg.ngf, 256, 256, _lib.ptr(P["img_w"]), _lib.ptr(P["img_b"]), 3, _lib.ptr(hip),
                                            _lib.current_stream_ptr()), "conv_img")
    eh, ec = e(hip, ref), e(c32, ref)
    print("%-12s %12.2e %12.2e %10.2f" % ("conv_img", eh, ec, eh / ec))

# ---- 3. one convolution, random operands
print("== 3. one 3x3 reflect conv, N(0,1) input, N(0,1/K) weights: error of the accumulation chain (K = 9 Cin)")
print("%-22s %12s %12s %10s" % ("Cin -> Cout @ HxW", "hip-vs-64", "cpu32-vs-64", "ratio"))
rng = np.random.default_rng(5)
with torch.no_grad():
    for cin, cout, hw in ((56, 128, 64), (128, 256, 64), (256, 128, 32), (1024, 128, 16), (1024, 1024, 16)):
        x = torch.from_numpy(rng.standard_normal((1, cin, hw, hw)).astype(np.float32))
        w = torch.from_numpy((rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32))
        ref = F.conv2d(F.pad(x.double(), (1, 1, 1, 1), mode="reflect"), w.double())
        c32 = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w)
        wp, rp = S._pack(w.cuda())
        y = torch.empty(1, cout, hw, hw, device="cuda")
        xg = x.cuda()
        _lib.check(_lib.lib().sln_spade_conv(_lib.ptr(xg), 1, cin, hw, hw, _lib.ptr(wp), None, cout, rp, 3, 0, 0.0, _lib.ptr(y),
                                             _lib.current_stream_ptr()), "conv")
        eh, ec = e(y, ref), e(c32, ref)
        # rms as well: the max over 1e5 outputs is an extreme-value statistic
        rh = float(((y.cpu().double() - ref) ** 2).mean().sqrt() / ref.abs().max())
        rc = float(((c32.double() - ref) ** 2).mean().sqrt() / ref.abs().max())
        print("%-22s %12.2e %12.2e %10.2f   rms %.2e / %.2e = %.2f" % ("%d -> %d @ %d" % (cin, cout, hw), eh, ec, eh / ec, rh, rc, rh / rc))

# ---- 4. what a blocked accumulation would buy: the same kernels on channel slices, partial sums added in fp32 by torch
# (the MFMA chain is bit-for-bit an fmaf chain over K, tools/lab/mfma_round.hip; torch's CPU conv is blocked - its error does not
# grow with K, table 3)
if os.environ.get("SLN_BUDGET_BLOCKED", "1") != "0":
    print("== 4. image error with the 3x3 convs of Cin >= thr accumulated in slices of S channels (partial sums added in fp32)")
    print("%-28s %12s %12s" % ("policy", "image-vs-64", "up_3-vs-64"))
    L = _lib.lib()
    orig_conv, orig_spade = G._conv, G._spade

    def sliced(x, w, bias, rows, rp, S, act=0):
        Bx, Cin, H, W = x.shape
        y = None
        for c0 in range(0, Cin, S):
            xs, ws = x[:, c0:c0 + S].contiguous(), w[:, c0:c0 + S].contiguous()
            part = torch.empty(Bx, rows, H, W, device=x.device)
            _lib.check(L.sln_spade_conv(_lib.ptr(xs), Bx, xs.shape[1], H, W, _lib.ptr(ws), None, rows, rp, 3, 0, 0.0, _lib.ptr(part),
                                        _lib.current_stream_ptr()), "conv slice")
            y = part if y is None else y + part
        if bias is not None:
            y = y + bias[:rows].view(1, -1, 1, 1)
        return torch.relu(y) if act == 1 else y

    def run(policy):
        thr, S = policy

        def conv(x, wbr, cout, ks, ln_acc=None, gap_acc=None):
            if ks != 3 or x.shape[1] < thr:
                return orig_conv(x, wbr, cout, ks, ln_acc, gap_acc)
            return sliced(x, wbr[0], wbr[1], cout, wbr[2], S)

        def spade(e, x, stats, seg_l, leaky, x_up=False):
            if S_NH < thr:
                return orig_spade(e, x, stats, seg_l, leaky, x_up)
            Bx, Cc = x.shape[:2]
            H, W = seg_l.shape[2:]
            nd = S_NH // 8
            cat = G._cat_buffer(seg_l, nd)
            _lib.check(L.sln_spade_depth_concat(_lib.ptr(seg_l), 1, seg_l.shape[1], H, W, _lib.ptr(e["wpd"]), _lib.ptr(e["bpd"]), nd,
                                                _lib.ptr(cat), 0, _lib.current_stream_ptr()), "depth_concat")
            actv = torch.empty(1, S_NH, H, W, device=x.device)
            _lib.check(L.sln_spade_conv(_lib.ptr(cat), 1, cat.shape[1], H, W, _lib.ptr(e["wsh"]), _lib.ptr(e["bsh"]), S_NH, e["rps"], 3, 1, 0.0,
                                        _lib.ptr(actv), _lib.current_stream_ptr()), "shared")
            gb = sliced(actv, e["wgb"], e["bgb"], e["rpg"], e["rpg"], S).contiguous()
            out = torch.empty(Bx, Cc, H, W, device=x.device)
            _lib.check(L.sln_spade_apply_up(_lib.ptr(x), 1 if x_up else 0, _lib.ptr(gb), Bx, Cc, H, W, e["rpg"], _lib.ptr(stats),
                                            2 if leaky else 0, 0.2, _lib.ptr(out), _lib.current_stream_ptr()), "apply")
            return out
        G._conv, G._spade = conv, spade
        G.unfused = True
        tp = {}
        try:
            with torch.no_grad():
                o = G(seg, z, taps=tp).cpu()
        finally:
            G._conv, G._spade, G.unfused = orig_conv, orig_spade, False
        if os.environ.get("SLN_BUDGET_IMAGES"):
            return e(o, r64), e(tp["up_3"], t64["up_3"]), rms(o, r64), rms(tp["up_3"], t64["up_3"])
        return e(o, r64), e(tp["up_3"], t64["up_3"])

    def rms(a, b):
        return float(((a.double().cpu() - b.double().cpu()) ** 2).mean().sqrt() / b.double().abs().max())

    def se64(xs, dx, w0, w2):            # SEBlock2 with the pool and both FCs in fp64 (experiment)
        gp = dx.double().mean((2, 3))
        sc = torch.sigmoid(F.linear(torch.relu(F.linear(gp, w0.double())), w2.double())).float()
        return xs + dx * sc[:, :, None, None]

    def run_se64(pol):
        keep = G._block_unfused

        def blk(name, x, seg_l):
            b_, e_ = getattr(G, name), G._packed[name]
            st = G._ln_stats(x)
            x_s = G._conv(G._spade(e_["norm_s"], x, st, seg_l, leaky=False), e_["conv_s"], b_.fout, 1) if b_.learned_shortcut else x
            dx = G._conv(G._spade(e_["norm_0"], x, st, seg_l, leaky=True), e_["conv_0"], b_.fmiddle, 3)
            dx = G._conv(G._spade(e_["norm_1"], dx, G._ln_stats(dx), seg_l, leaky=True), e_["conv_1"], b_.fout, 3)
            return se64(x_s, dx, e_["se0"], e_["se2"])
        G._block_unfused = blk
        try:
            return run(pol)
        finally:
            G._block_unfused = keep

    S_NH = S.NHIDDEN
    pols = (("none (unfused schedule)", (1 << 30, 8)), ("all, S=8", (0, 8)), ("Cin>=512, S=8", (512, 8)), ("Cin>=256, S=8", (256, 8)),
            ("Cin>=128, S=8", (128, 8)), ("Cin>=128, S=32", (128, 32)), ("Cin>=128, S=64", (128, 64)), ("Cin>=256, S=64", (256, 64)),
            ("Cin>=256, S=128", (256, 128)), ("Cin>=512, S=128", (512, 128)), ("Cin>=512, S=256", (512, 256)))
    if os.environ.get("SLN_BUDGET_IMAGES"):
        # several images: max AND rms (the max over 2e5 pixels of a heavy-tailed error is a noisy statistic)
        n_img = int(os.environ["SLN_BUDGET_IMAGES"])
        gg = torch.Generator(device="cuda").manual_seed(0)
        low = torch.rand(B, 1, 16, 16, device="cuda", generator=gg) * 2 - 1
        depth = F.interpolate(low, size=(256, 256), mode="bilinear", align_corners=False)
        lab = F.interpolate(torch.randn(B, 40, 16, 16, device="cuda", generator=gg), size=(256, 256), mode="bilinear", align_corners=False).argmax(1)
        seg_all = torch.cat([depth, F.one_hot(lab, 40).permute(0, 3, 1, 2).float()], 1).contiguous()
        z_all = torch.randn(B, 256, device="cuda", generator=gg)

        print("%-5s %-26s %10s %10s %10s %10s" % ("image", "path", "img max", "img rms", "up_3 max", "up_3 rms"))
        for k in range(n_img):
            seg, z = seg_all[k:k + 1].contiguous(), z_all[k:k + 1].contiguous()
            segc, zc = seg.cpu(), z.cpu()
            t32, t64 = {}, {}
            with torch.no_grad():
                r32 = spade_ref.generator(sd, cfg, segc, zc, t32)
                r64 = spade_ref.generator(sd64, cfg, segc.double(), zc.double(), t64)
                tp = {}
                o = G(seg, z, taps=tp).cpu()
            print("%-5d %-26s %10.2e %10.2e %10.2e %10.2e" % (k, "cpu fp32", e(r32, r64), rms(r32, r64), e(t32["up_3"], t64["up_3"]), rms(t32["up_3"], t64["up_3"])))
            print("%-5d %-26s %10.2e %10.2e %10.2e %10.2e" % (k, "hip (fused)", e(o, r64), rms(o, r64), e(tp["up_3"], t64["up_3"]), rms(tp["up_3"], t64["up_3"])))
            for name, pol in pols[:2] + pols[5:6]:
                # run() closes over seg / z / r64 / t64 of this scope through the globals
                globals().update(seg=seg, z=z, r64=r64, t64=t64)
                G._conv, G._spade = orig_conv, orig_spade
                ei, eu, ri, ru = run(pol)
                print("%-5d %-26s %10.2e %10.2e %10.2e %10.2e" % (k, name, ei, ri, eu, ru))
            se_pols = pols[:2]
            if os.environ.get("SLN_BUDGET_SE_SWEEP"):
                se_pols = (("all, S=16", (0, 16)), ("all, S=32", (0, 32)), ("Cin>=256, S=16", (256, 16)), ("Cin>=512, S=16", (512, 16)),
                           ("Cin>=256, S=32", (256, 32)), ("Cin>=256, S=64", (256, 64)), ("Cin>=128, S=16", (128, 16)))
            for name, pol in se_pols:
                ei, eu, ri, ru = run_se64(pol)
                print("%-5d %-26s %10.2e %10.2e %10.2e %10.2e" % (k, name + " +SE fp64", ei, ri, eu, ru))
    else:
        for name, pol in pols:
            ei, eu = run(pol)
            print("%-28s %12.2e %12.2e" % (name, ei, eu))
