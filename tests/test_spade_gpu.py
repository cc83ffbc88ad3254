"""GPU parity tests of the SPADE generator path (run with -m gpu): csrc/spade.hip through the C ABI against the
CPU oracle (oracle/spade_ref.py) and the fixtures produced by the reference's own SPADEGenerator4."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, pkg
from parity import assert_close

pytestmark = pytest.mark.gpu

from oracle import spade_ref                       # noqa: E402
from oracle.gen_golden_spade import CASES          # noqa: E402


def _checks(t):
    t = t.detach().double().cpu()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


@pytest.mark.parametrize("B,Cin,Cout,H,W,ks", [(2, 56, 128, 16, 16, 3), (1, 128, 64, 40, 24, 3), (2, 16, 8, 8, 8, 3),
                                                (1, 64, 32, 2, 2, 3), (2, 128, 64, 16, 32, 1), (1, 8, 200, 9, 17, 3)])
def test_conv_against_torch_cpu(B, Cin, Cout, H, W, ks):
    L = pkg("_lib"); S = pkg("host.SPADE_related")
    g = torch.Generator().manual_seed(Cin * 31 + Cout)
    x = torch.randn(B, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect") if ks == 3 else x, w, b)
    for act, fn in ((0, lambda t: t), (1, F.relu), (2, lambda t: F.leaky_relu(t, 0.2))):
        wp, rp = S._pack(w.cuda())
        bp = torch.zeros(rp, device="cuda"); bp[:Cout] = b.cuda()
        y = torch.empty(B, Cout, H, W, device="cuda")
        L.check(L.lib().sln_spade_conv(L.ptr(x.cuda()), B, Cin, H, W, L.ptr(wp), L.ptr(bp), Cout, rp, ks, act, 0.2, L.ptr(y),
                                       L.current_stream_ptr()), "conv")
        assert_close(y.cpu().numpy(), fn(ref).numpy(), "conv act=%d" % act, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,Cin,H,W,Cout", [(1, 64, 40, 24, 3), (2, 20, 17, 9, 3), (1, 8, 8, 8, 1), (40, 16, 64, 64, 3), (3, 64, 128, 128, 4)])
def test_conv_img_tanh_against_torch(B, Cin, H, W, Cout):
    """tanh(conv5x5(leaky_relu(x, 0.2), padding 2)) (models/SPADE_related.py:1602-1603): the 16 x 16-tile kernel (many tiles) and
    the 8 x 8-tile / channels-over-wavefronts kernel of launches with fewer than 512 tiles (batch-1 calls), ragged sizes."""
    L = pkg("_lib")
    g = torch.Generator().manual_seed(Cin + 13 * H)
    x = torch.randn(B, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, 5, 5, generator=g) / (Cin * 25) ** 0.5; b = torch.randn(Cout, generator=g) * 0.1
    ref = torch.tanh(F.conv2d(F.leaky_relu(x.double(), 0.2), w.double(), b.double(), padding=2)).float()
    xd, wd, bd = x.cuda(), w.cuda().contiguous(), b.cuda()
    y = torch.empty(B, Cout, H, W, device="cuda")
    L.check(L.lib().sln_conv_img_tanh(L.ptr(xd), B, Cin, H, W, L.ptr(wd), L.ptr(bd), Cout, L.ptr(y), L.current_stream_ptr()), "conv_img")
    assert_close(y.cpu().numpy(), ref.numpy(), "conv_img", rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("C,H,W", [(64, 16, 16), (8, 8, 24), (100, 10, 10)])
def test_fused_spade_modulation_against_oracle(C, H, W):
    """SPADE4 (:1438-1454): LayerNorm2D stats + depth conv/concat + shared conv + fused gamma/beta modulation."""
    S = pkg("host.SPADE_related")
    g = torch.Generator().manual_seed(C)
    B = 2
    sd = {}
    p = "n"
    sd[p + ".mlp_preshared_depth.1.weight"] = torch.randn(16, 1, 3, 3, generator=g) / 3
    sd[p + ".mlp_preshared_depth.1.bias"] = torch.randn(16, generator=g) * 0.1
    sd[p + ".mlp_shared.1.weight"] = torch.randn(128, 56, 3, 3, generator=g) / (56 * 9) ** 0.5
    sd[p + ".mlp_shared.1.bias"] = torch.randn(128, generator=g) * 0.1
    for nm in ("gamma", "beta"):
        sd[p + ".mlp_%s.1.weight" % nm] = torch.randn(C, 128, 3, 3, generator=g) / (128 * 9) ** 0.5
        sd[p + ".mlp_%s.1.bias" % nm] = torch.randn(C, generator=g) * 0.1
    x = torch.randn(B, C, H, W, generator=g) * 3 + 1
    seg = torch.rand(B, 41, H, W, generator=g)
    ref = spade_ref.spade4(sd, p, x, seg)
    gen = S.SPADEGenerator4.__new__(S.SPADEGenerator4)
    wsh, rps = S._pack(sd[p + ".mlp_shared.1.weight"].cuda())
    bsh = torch.zeros(rps, device="cuda"); bsh[:128] = sd[p + ".mlp_shared.1.bias"].cuda()
    wgb, bgb, rpg = S._pack_gamma_beta(sd[p + ".mlp_gamma.1.weight"].cuda(), sd[p + ".mlp_gamma.1.bias"].cuda(),
                                       sd[p + ".mlp_beta.1.weight"].cuda(), sd[p + ".mlp_beta.1.bias"].cuda())
    e = dict(wpd=sd[p + ".mlp_preshared_depth.1.weight"].reshape(16, 9).cuda().contiguous(), bpd=sd[p + ".mlp_preshared_depth.1.bias"].cuda(),
             wsh=wsh, bsh=bsh, rps=rps, wgb=wgb, bgb=bgb, rpg=rpg)
    xd = x.cuda()
    stats = S.SPADEGenerator4._ln_stats(gen, xd)
    flat = x.reshape(B, -1)
    assert_close(stats[:, 0].cpu().numpy(), flat.mean(1).numpy(), "mean", rtol=1e-6)
    assert_close(stats[:, 1].cpu().numpy(), (1.0 / (flat.std(1) + 1e-5)).numpy(), "inv", rtol=1e-5)
    for leaky in (False, True):
        out = S.SPADEGenerator4._spade(gen, e, xd, stats, seg.cuda(), leaky)
        r = F.leaky_relu(ref, 0.2) if leaky else ref
        assert_close(out.cpu().numpy(), r.numpy(), "spade4 leaky=%s" % leaky, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", list(CASES))
def test_generator_against_reference_fixture(name):
    S = pkg("host.SPADE_related")
    g = load_golden(name)
    over, B, skw = CASES[name]
    cfg = spade_ref.SpadeConfig(**over)
    sd = spade_ref.init_state(cfg, seed=7, **skw)
    G = S.SPADEGenerator4(cfg.semantic_nc, cfg.target_nc, cfg.nz, cfg.ngf, 'spectralspadelayer3x3', cfg.crop_size, 'normal')
    assert set(G.state_dict().keys()) == set(sd.keys())
    G.load_state_dict(sd)
    G = G.cuda().eval()
    seg, z = spade_ref.synth_input(cfg, B, seed=3)
    taps = {}
    out = G(seg.cuda(), z.cuda(), taps=taps)
    report = []
    for n, t in taps.items():
        got, ref = _checks(t), g["check:" + n]
        report.append("%s: abs-sum rel err %.2e, sq-sum rel err %.2e" % (n, abs(got[1] - ref[1]) / ref[1], abs(got[2] - ref[2]) / ref[2]))
    try:
        for n, t in taps.items():
            assert_close(_checks(t)[1:], g["check:" + n][1:], name + ":" + n, rtol=1e-4)
        assert_close(_checks(out)[1:], g["out_check"][1:], name + ":out", rtol=1e-4)
        if "out" in g.files:
            assert_close(out.cpu().numpy(), g["out"], name + ":image")
            assert_close(taps["head_0"].cpu().numpy(), g["tap:head_0"], name + ":head_0")
        else:
            assert_close(out[:, :, 100:132, 60:92].cpu().numpy(), g["out_crop"], name + ":crop")
            assert_close(out[:, :, ::37, :].cpu().numpy(), g["out_rows"], name + ":rows")
    except AssertionError as e:
        raise AssertionError(str(e) + "\n" + "\n".join(report))


def test_one_semantic_map_many_z_equals_the_broadcast_of_the_reference():
    """colorize_with_spade (testing/test_SPADE_shade.py:30-79) draws many z for ONE map; the reference module
    broadcasts gamma/beta [1,C,H,W] over the batch.  The shared path (gamma/beta computed once, sln_spade_apply_up per sample)
    must equal the oracle's broadcast and the per-sample path on the expanded map."""
    S = pkg("host.SPADE_related")
    cfg = spade_ref.SpadeConfig(**CASES["spade_small"][0])
    sd = spade_ref.init_state(cfg, seed=7)
    G = S.SPADEGenerator4(cfg.semantic_nc, cfg.target_nc, cfg.nz, cfg.ngf, 'spectralspadelayer3x3', cfg.crop_size, 'normal')
    G.load_state_dict(sd); G = G.cuda().eval()
    seg, _ = spade_ref.synth_input(cfg, 1, seed=5)
    z = torch.from_numpy(np.random.default_rng(2).standard_normal((5, cfg.nz)).astype(np.float32))
    taps_ref = {}
    ref = spade_ref.generator(sd, cfg, seg, z, taps_ref)            # torch broadcasting, as the reference module does
    taps = {}
    out = G(seg.cuda(), z.cuda(), taps=taps)
    assert out.shape == (5, cfg.target_nc, cfg.crop_size, cfg.crop_size)
    for n in taps:
        assert_close(taps[n].cpu().numpy(), taps_ref[n].numpy(), "shared:" + n, rtol=1e-4, atol=1e-4 * float(taps_ref[n].abs().max()))
    assert_close(out.cpu().numpy(), ref.numpy(), "shared:image", rtol=1e-4, atol=1e-4)
    per_sample = G(seg.expand(5, -1, -1, -1).contiguous().cuda(), z.cuda())
    # the shared path's gamma/beta come out of the bias/activation conv (chunks of 8 input channels), the per-sample path's out of the
    # modulation conv (DMA kernel, chunks of 4): the same products summed in a different order
    assert_close(out.cpu().numpy(), per_sample.cpu().numpy(), "shared vs per-sample", rtol=1e-5, atol=1e-4)
    with pytest.raises(RuntimeError):
        G(seg.expand(2, -1, -1, -1).contiguous().cuda(), z.cuda())


def test_colorize_pipeline_from_raw_maps():
    """depth + class masks -> 41-channel tensor (host/spade_input.py) -> several z for the one map -> uint8 images"""
    S = pkg("host.SPADE_related"); I = pkg("host.spade_input")
    cfg = spade_ref.SpadeConfig(**CASES["spade_small"][0])
    G = S.SPADEGenerator4(cfg.semantic_nc, cfg.target_nc, cfg.nz, cfg.ngf, 'spectralspadelayer3x3', cfg.crop_size, 'normal')
    G.load_state_dict(spade_ref.init_state(cfg, seed=7)); G = G.cuda().eval()
    n = 4 * cfg.crop_size
    yy, xx = torch.meshgrid(torch.arange(n) / n, torch.arange(n) / n, indexing="ij")
    depth = 2.0 + 3.0 * yy + torch.sin(6 * xx)
    m = torch.zeros(n, n); m[n // 4:n // 2, n // 4:n // 2] = 255
    total = I.build_input(depth.cuda(), {"bed": m.cuda(), "wall": 255 - m.cuda()}, size=cfg.crop_size)
    assert total.shape == (1, 41, cfg.crop_size, cfg.crop_size) and total.is_cuda
    imgs = I.colorize(G, total, 4, generator=torch.Generator(device="cuda").manual_seed(0))
    ref = spade_ref.generator(spade_ref.init_state(cfg, seed=7), cfg, total.cpu(),
                              torch.randn(4, cfg.nz, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0)).cpu())
    assert_close(imgs.cpu().numpy(), ref.numpy(), "colorize", rtol=1e-4, atol=1e-4)
    u8 = I.to_uint8(imgs)
    assert u8.shape == (4, cfg.crop_size, cfg.crop_size, 3) and u8.dtype == torch.uint8


def test_full_size_batch_of_32_equals_per_sample_calls():
    """BASELINE configs[3] at its real size (batch 32, 256x256, ngf 64): the batched call must give, sample by sample, what a
    batch-1 call gives (no cross-sample leakage in the tiled convolutions, LayerNorm statistics per sample); a size-independent
    property on top of the fixture of the full-size reference output (spade_full)."""
    S = pkg("host.SPADE_related")
    cfg = spade_ref.SpadeConfig(**CASES["spade_full"][0])
    G = S.SPADEGenerator4(cfg.semantic_nc, cfg.target_nc, cfg.nz, cfg.ngf, 'spectralspadelayer3x3', cfg.crop_size, 'normal')
    G.load_state_dict(spade_ref.init_state(cfg, seed=7))
    G = G.cuda().eval()
    seg, z = spade_ref.synth_input(cfg, 32, seed=11)
    seg, z = seg.cuda(), z.cuda()
    with torch.no_grad():
        out = G(seg, z)
        assert out.shape == (32, 3, cfg.crop_size, cfg.crop_size) and torch.isfinite(out).all()
        assert float(out.abs().max()) <= 1.0                                        # tanh
        for b in (0, 13, 31):
            one = G(seg[b:b + 1].contiguous(), z[b:b + 1].contiguous())
            # (1e-4, not bit-equality: a batch-1 call splits the input channels of the 8 x 8 .. 64 x 64 layers over several
            #  workgroups - another summation order than the batched launch; leakage between samples would be an O(1) error)
            assert_close(out[b:b + 1].cpu().numpy(), one.cpu().numpy(), "sample %d" % b, rtol=1e-4, atol=1e-4)
        # different samples give different images (the batch is not broadcast from one sample)
        assert float((out[0] - out[1]).abs().max()) > 1e-3
        # one semantic map, several z (colorize_with_spade) at full size: the shared gamma/beta path against the per-sample path
        shared = G(seg[5:6].contiguous(), z[:3].contiguous())
        per = G(seg[5:6].expand(3, -1, -1, -1).contiguous(), z[:3].contiguous())
        # the two schedules sum SEBlock2's average pool in a different order (fp32 tree / fp64 epilogue sums): 1e-7 on a scale, which
        # the 110 M-parameter stack and the final 5x5 conv over 1 600 cancelling products carry to some 1e-5 of the image (the CPU
        # fp32 oracle sits 1e-4 from the fp64 one on the same weights, bench.py check_spade)
        assert_close(shared.cpu().numpy(), per.cpu().numpy(), "full-size shared vs per-sample", rtol=1e-5, atol=2e-4)


def test_repeated_calls_with_changing_inputs_keep_no_stale_state():
    """Weight packs are cached across calls, the gamma/beta of a shared map and the concat buffers only within one: alternating
    semantic maps, batch sizes and the shared / per-sample paths must reproduce the first answers, and reloading the weights must
    change them."""
    S = pkg("host.SPADE_related")
    cfg = spade_ref.SpadeConfig(**CASES["spade_small"][0])
    G = S.SPADEGenerator4(cfg.semantic_nc, cfg.target_nc, cfg.nz, cfg.ngf, 'spectralspadelayer3x3', cfg.crop_size, 'normal')
    G.load_state_dict(spade_ref.init_state(cfg, seed=7)); G = G.cuda().eval()
    segA, zA = [t.cuda() for t in spade_ref.synth_input(cfg, 3, seed=1)]
    segB, zB = [t.cuda() for t in spade_ref.synth_input(cfg, 2, seed=2)]
    with torch.no_grad():
        a1 = G(segA, zA); b1 = G(segB, zB)
        sh1 = G(segA[:1].contiguous(), zA)                           # one map, three z: the shared path
        a2 = G(segA, zA); sh2 = G(segA[:1].contiguous(), zA); b2 = G(segB, zB)
        segA.mul_(1.0)                                               # same values, new version counter
        a3 = G(segA, zA)
        assert torch.equal(a1, a2) and torch.equal(b1, b2) and torch.equal(sh1, sh2) and torch.equal(a1, a3)
        assert_close(sh1[0:1].cpu().numpy(), a1[0:1].cpu().numpy(), "shared path, first sample", rtol=1e-5, atol=1e-5)
        assert float((a1[:2] - b1).abs().max()) > 1e-3
        G.load_state_dict(spade_ref.init_state(cfg, seed=8))         # in-place copy: the packs must be rebuilt
        a4 = G(segA, zA)
        assert float((a4 - a1).abs().max()) > 1e-3
        G.load_state_dict(spade_ref.init_state(cfg, seed=7))
        assert torch.equal(G(segA, zA), a1)


@pytest.mark.parametrize("B,Cin,Cout,H,W,ks", [(3, 24, 128, 16, 16, 3), (2, 40, 72, 12, 20, 3), (2, 32, 64, 8, 8, 1), (1, 16, 200, 9, 17, 3),
                                                (1, 128, 64, 16, 16, 3), (2, 512, 128, 8, 8, 3), (1, 256, 72, 8, 8, 1), (1, 1024, 200, 9, 17, 3)])
def test_conv_epilogue_sums(B, Cin, Cout, H, W, ks):
    """sln_spade_conv_sums: the LayerNorm2D sums and SEBlock2's pixel sums of what the epilogue wrote (models/SPADE_related.py:70-85,
    128-149) against torch on the conv's own output.  The last four cases are launches of a few workgroups with many input
    channels: the input-channel split (conv_split_finish_kernel takes the sums; 3x3 plain, 3x3 with blocked accumulation, 1x1,
    ragged image and row count)."""
    L = pkg("_lib"); S = pkg("host.SPADE_related")
    g = torch.Generator().manual_seed(Cin + 7 * Cout)
    x = (torch.randn(B, Cin, H, W, generator=g) + 0.3).cuda()
    w = (torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5).cuda()
    wp, rp = S._pack(w)
    bp = torch.zeros(rp, device="cuda"); bp[:Cout] = torch.randn(Cout, generator=g).cuda()
    for act in (0, 2):
        y = torch.empty(B, Cout, H, W, device="cuda")
        ln = torch.zeros(16 * B, dtype=torch.float64, device="cuda"); gap = torch.zeros(B, Cout, dtype=torch.float64, device="cuda")
        L.check(L.lib().sln_spade_conv_sums(L.ptr(x), B, Cin, H, W, L.ptr(wp), L.ptr(bp), Cout, rp, ks, act, 0.2, L.ptr(y), L.ptr(ln), L.ptr(gap),
                                            L.current_stream_ptr()), "conv_sums")
        y2 = torch.empty_like(y)
        L.check(L.lib().sln_spade_conv(L.ptr(x), B, Cin, H, W, L.ptr(wp), L.ptr(bp), Cout, rp, ks, act, 0.2, L.ptr(y2), L.current_stream_ptr()), "conv")
        assert torch.equal(y, y2)
        ref = F.conv2d(F.pad(x.double(), (1, 1, 1, 1), mode="reflect") if ks == 3 else x.double(), w.double(), bp[:Cout].double())
        ref = F.leaky_relu(ref, 0.2) if act == 2 else ref
        assert_close(y.cpu().numpy(), ref.float().cpu().numpy(), "conv (act %d) against torch fp64" % act, rtol=2e-5, atol=2e-5)
        yd = y.double()
        ln = ln.view(B, 16)
        assert_close(ln[:, 0].cpu().numpy(), yd.sum((1, 2, 3)).cpu().numpy(), "sum", rtol=1e-6, atol=1e-4)
        assert_close(ln[:, 1].cpu().numpy(), (yd * yd).sum((1, 2, 3)).cpu().numpy(), "sumsq", rtol=1e-6)
        assert_close(gap.cpu().numpy(), yd.sum((2, 3)).cpu().numpy(), "pixel sums", rtol=1e-5, atol=1e-4)
        stats = torch.empty(B, 2, device="cuda")
        L.check(L.lib().sln_layernorm_finalize(L.ptr(ln), B, Cout * H * W, 1, 1e-5, L.ptr(stats), L.current_stream_ptr()), "finalize")
        flat = y.reshape(B, -1)
        assert_close(stats[:, 0].cpu().numpy(), flat.mean(1).cpu().numpy(), "mean", rtol=1e-5, atol=1e-6)
        assert_close(stats[:, 1].cpu().numpy(), (1.0 / (flat.std(1) + 1e-5)).cpu().numpy(), "inv", rtol=1e-5)


@pytest.mark.parametrize("up_mode,xs_up,with_sums", [(-1, 0, True), (-1, 1, False), (0, 0, True), (1, 0, False), (1, 1, True), (0, 1, False)])
def test_block_tail_against_torch(up_mode, xs_up, with_sums):
    """x_s + SEBlock2(dx) (:70-85, :1492-1493), nn.Upsample (:1585-1600) and the next LayerNorm2D's statistics in one launch."""
    L = pkg("_lib")
    g = torch.Generator().manual_seed(5 + up_mode)
    B, C, H, W = 3, 24, 12, 8
    dx = torch.randn(B, C, H, W, generator=g).cuda() + 0.2
    xs_small = torch.randn(B, C, H // 2, W // 2, generator=g).cuda()
    xs = F.interpolate(xs_small, scale_factor=2, mode="nearest") if xs_up else torch.randn(B, C, H, W, generator=g).cuda()
    w0 = torch.randn(C // 8, C, generator=g).cuda(); w2 = torch.randn(C, C // 8, generator=g).cuda()
    scale = torch.sigmoid(F.relu(dx.mean((2, 3)) @ w0.t()) @ w2.t())
    v = xs + dx * scale[:, :, None, None]
    ref = v if up_mode < 0 else F.interpolate(v, scale_factor=2, mode="nearest" if up_mode == 0 else "bilinear",
                                              **({} if up_mode == 0 else {"align_corners": False}))
    k = 1 if up_mode < 0 else 2
    out = torch.empty(B, C, k * H, k * W, device="cuda")
    acc = torch.full((16 * B,), 7.0, dtype=torch.float64, device="cuda")       # zeroed by the call
    stats = torch.empty(B, 2, device="cuda")
    scratch = torch.empty(2 * B * C, device="cuda")
    sums = dx.double().sum((2, 3)).contiguous() if with_sums else None
    for rep in (1, 4):
        L.check(L.lib().sln_block_tail(L.ptr(xs_small if xs_up else xs), xs_up, L.ptr(dx), B, C, H, W, L.ptr(sums) if with_sums else None,
                                       L.ptr(w0), L.ptr(w2), L.ptr(scratch), up_mode, L.ptr(out), L.ptr(acc), rep, 1e-5, L.ptr(stats),
                                       L.current_stream_ptr()), "tail")
        assert_close(out.cpu().numpy(), ref.cpu().numpy(), "tail output", rtol=1e-5, atol=1e-5)
        normed = ref if rep == 1 else F.interpolate(ref, scale_factor=2, mode="nearest")
        flat = normed.reshape(B, -1)
        assert_close(stats[:, 0].cpu().numpy(), flat.mean(1).cpu().numpy(), "mean", rtol=1e-5, atol=1e-6)
        assert_close(stats[:, 1].cpu().numpy(), (1.0 / (flat.std(1) + 1e-5)).cpu().numpy(), "inv", rtol=1e-5)
    assert L.lib().sln_block_tail(L.ptr(xs), 0, L.ptr(dx), B, C, H, 6, None, L.ptr(w0), L.ptr(w2), L.ptr(scratch), -1, L.ptr(out), None, 1, 1e-5,
                                  None, L.current_stream_ptr()) < 0                # a row must be a whole number of float4


@pytest.mark.parametrize("crop,B", [(128, 3), (256, 2)])
def test_fused_schedule_against_oracle_and_unfused(crop, B):
    """The fused schedule (conv epilogue sums, block tail, nearest upsampling read through by its consumers) against the oracle
    and against the one-launch-per-module schedule on the same weights: every block output and the image."""
    S = pkg("host.SPADE_related")
    cfg = spade_ref.SpadeConfig(ngf=8, nz=16, crop_size=crop)
    sd = spade_ref.init_state(cfg, seed=11)
    G = S.SPADEGenerator4(cfg.semantic_nc, cfg.target_nc, cfg.nz, cfg.ngf, 'spectralspadelayer3x3', cfg.crop_size, 'normal')
    G.load_state_dict(sd); G = G.cuda().eval()
    seg, z = spade_ref.synth_input(cfg, B, seed=4)
    taps_ref, taps, taps_u = {}, {}, {}
    ref = spade_ref.generator(sd, cfg, seg, z, taps_ref)
    out = G(seg.cuda(), z.cuda(), taps=taps)
    G.unfused = True
    out_u = G(seg.cuda(), z.cuda(), taps=taps_u)
    assert set(taps) == set(taps_ref) == set(taps_u)
    for n in taps_ref:
        scale = float(taps_ref[n].abs().max())
        assert_close(taps[n].cpu().numpy(), taps_ref[n].numpy(), "fused:" + n, rtol=1e-4, atol=1e-4 * scale)
        assert_close(taps[n].cpu().numpy(), taps_u[n].cpu().numpy(), "fused vs unfused:" + n, rtol=1e-5, atol=2e-5 * scale)
    assert_close(out.cpu().numpy(), ref.numpy(), "fused:image", rtol=1e-4, atol=1e-4)
    assert_close(out.cpu().numpy(), out_u.cpu().numpy(), "fused vs unfused:image", rtol=1e-5, atol=2e-4)      # see the full-size test
    # one map, many z (colorize_with_spade): gamma/beta once per map (sln_spade_apply_up) inside the fused schedule
    G.unfused = False
    shared = G(seg[:1].contiguous().cuda(), z.cuda())
    ref1 = spade_ref.generator(sd, cfg, seg[:1].contiguous(), z)
    assert_close(shared.cpu().numpy(), ref1.numpy(), "fused shared:image", rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("env", ["SLN_CONV_DMA", "SLN_CONV_STAGED"])
def test_both_conv_kernels_carry_both_epilogues(env):
    """The dispatcher sends modulation convs to the direct-to-LDS kernel and bias/activation convs to the register-staged one; each
    kernel is built with both epilogues (the environment switches are read once per process: the conv, modulation and fused-schedule
    tests run again in a child process with every conv forced through one kernel)."""
    import os, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    e = dict(os.environ); e[env] = "1"
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", os.path.join(here, "test_spade_gpu.py"), "-k",
                        "conv_against_torch_cpu or fused_spade_modulation or conv_epilogue_sums or fused_schedule"],
                       env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_batch_one_calls_on_the_same_map_reuse_its_planes_and_nothing_stale():
    """testing/test_SPADE_shade.py:77-79 calls the model once per z on the SAME tensor `total`: from the second consecutive call
    the gamma|beta planes of that map are kept (round 3).  Every call must equal the oracle's answer for its z; a modified map
    (in place: same address, new version counter), another map at a recycled address, new weights, and the switch
    `reuse_map_planes = False` must all give fresh answers."""
    S = pkg("host.SPADE_related")
    cfg = spade_ref.SpadeConfig(**CASES["spade_small"][0])
    sd = spade_ref.init_state(cfg, seed=7)
    G = S.SPADEGenerator4(cfg.semantic_nc, cfg.target_nc, cfg.nz, cfg.ngf, 'spectralspadelayer3x3', cfg.crop_size, 'normal')
    G.load_state_dict(sd); G = G.cuda().eval()
    seg, _ = spade_ref.synth_input(cfg, 1, seed=5)
    zs = torch.from_numpy(np.random.default_rng(4).standard_normal((5, cfg.nz)).astype(np.float32))
    ref = spade_ref.generator(sd, cfg, seg, zs).numpy()
    total = seg.cuda()
    outs = [G(total, zs[k:k + 1].cuda()) for k in range(5)]            # call 1: fused path; call 2: planes built; calls 3-5: reused
    assert G._map_repeat and len(G._map_memo["gb"]) > 0
    for k in range(5):
        assert_close(outs[k].cpu().numpy(), ref[k:k + 1], "call %d on the same map" % k, rtol=1e-4, atol=1e-4)
    G.reuse_map_planes = False
    plain = [G(total, zs[k:k + 1].cuda()) for k in range(5)]
    G.reuse_map_planes = True
    for k in range(5):
        assert_close(outs[k].cpu().numpy(), plain[k].cpu().numpy(), "kept planes vs fused path, call %d" % k, rtol=1e-5, atol=1e-4)
    # the map changes in place: same address, new version -> fresh planes
    G(total, zs[:1].cuda()); G(total, zs[:1].cuda())
    seg2, _ = spade_ref.synth_input(cfg, 1, seed=6)
    total.copy_(seg2.cuda())
    ref2 = spade_ref.generator(sd, cfg, seg2, zs[:1]).numpy()
    for _ in range(3):
        assert_close(G(total, zs[:1].cuda()).cpu().numpy(), ref2, "map modified in place", rtol=1e-4, atol=1e-4)
    # new weights: the kept planes are stale
    sd8 = spade_ref.init_state(cfg, seed=8)
    G.load_state_dict(sd8)
    ref3 = spade_ref.generator(sd8, cfg, seg2, zs[:1]).numpy()
    for _ in range(3):
        assert_close(G(total, zs[:1].cuda()).cpu().numpy(), ref3, "weights reloaded", rtol=1e-4, atol=1e-4)
    G.clear_map_cache()
    assert G._map_memo is None


def test_full_size_generator_at_the_bench_weights_within_1e4_of_the_fp64_oracle():
    """bench.py's `spade` leg, images 0 and 1 (torch's default initialisation under seed 0 - what the reference's constructor gives -
    with conv_img scaled so that tanh is not saturated; inputs from host/synthetic.py::spade_input): the 110 M-parameter generator
    at 256x256 (a) against the REFERENCE-generated fixture tests/golden/spade_bench.npz (the reference class itself, run on these
    weights and this input in the build container: crop, seven full rows, checksums of every block), (b) within 1e-4 of the image
    scale PLUS the CPU fp32 oracle's own distance from an fp64 evaluation, and within 2e-4 of the CPU fp32 path itself.
    What closed (b) in round 4: SEBlock2's two FCs in fp64 and blocked accumulation in the long MFMA chains (csrc/spade.hip)."""
    S = pkg("host.SPADE_related"); syn = pkg("host.synthetic")
    from oracle.gen_golden_spade import BENCH_IMG_GAIN, BENCH_SEED
    torch.manual_seed(BENCH_SEED)
    G = S.SPADEGenerator4(41, 3, 256, 64, 'spectralspadelayer3x3', 256, 'normal')
    with torch.no_grad():
        G.conv_img.weight.mul_(BENCH_IMG_GAIN); G.conv_img.bias.mul_(BENCH_IMG_GAIN)
    G = G.cuda().eval()
    seg, z = syn.spade_input(2, seed=BENCH_SEED)
    taps = {}
    with torch.no_grad():
        out = G(seg.cuda(), z.cuda()).cpu()
        out0 = G(seg[:1].cuda(), z[:1].cuda(), taps=taps).cpu()
    g = load_golden("spade_bench")
    scale = float(np.abs(g["out_rows"]).max())
    assert float(g["out_abs_mean"][0]) < 0.5
    e_crop = float(np.abs(out0[:, :, 100:132, 60:92].numpy() - g["out_crop"]).max()) / scale
    e_rows = float(np.abs(out0[:, :, ::37, :].numpy() - g["out_rows"]).max()) / scale
    assert max(e_crop, e_rows) <= 1e-4, ("HIP vs the reference's own fp32 output (north_star: 1e-4; measured 7.7e-6)", e_crop, e_rows)
    for n, t in taps.items():
        assert_close(_checks(t)[1:], g["check:" + n][1:], "spade_bench:" + n, rtol=1e-4)
    cfg = spade_ref.SpadeConfig()
    sd = {k: v.detach().cpu() for k, v in G.state_dict().items()}
    torch.set_num_threads(min(16, torch.get_num_threads()))
    with torch.no_grad():
        r32 = spade_ref.generator(sd, cfg, seg, z)
        r64 = spade_ref.generator({k: v.double() for k, v in sd.items()}, cfg, seg.double(), z.double())
    assert_close(r32[:1, :, ::37, :].numpy(), g["out_rows"], "oracle vs reference fixture at the bench weights", rtol=1e-5, atol=1e-5)
    for b in range(2):
        scale = float(r64[b].abs().max())
        e_hip = float((out[b].double() - r64[b]).abs().max()) / scale
        e_cpu = float((r32[b].double() - r64[b]).abs().max()) / scale
        e_hc = float((out[b].double() - r32[b].double()).abs().max()) / scale
        assert e_hip <= 1e-4, (b, e_hip, "north_star's 1e-4 against the fp64 oracle (measured 3.5e-6; the fp32 oracle itself: %.1e)" % e_cpu)
        assert e_hc <= 1e-4, (b, e_hc)


@pytest.mark.parametrize("Cin,Cout,H,B", [(512, 128, 16, 2), (1024, 256, 8, 1), (520, 64, 32, 2), (512, 128, 64, 16)])
def test_blocked_accumulation_of_long_chains_is_as_accurate_as_the_cpu_path(Cin, Cout, H, B):
    """K = 9 Cin >= 4 608: the accumulators restart every 16 input channels (conv_flush).  The error against an fp64 evaluation
    stays at torch's CPU level (its convolution sums in blocks too) - a serial MFMA chain is 6x further away at K = 9 216 -
    in the 8 x 16-pixel DMA variant (small launches) and the 64-row 16 x 16-pixel one (the last case: 512 workgroups)."""
    L = pkg("_lib"); S = pkg("host.SPADE_related")
    g = torch.Generator().manual_seed(Cin + Cout)
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5
    b = torch.randn(Cout, generator=g)
    ref = F.conv2d(F.pad(x.double(), (1, 1, 1, 1), mode="reflect"), w.double(), b.double())
    c32 = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w, b)
    wp, rp = S._pack(w.cuda())
    bp = torch.zeros(rp, device="cuda"); bp[:Cout] = b.cuda()
    y = torch.empty(B, Cout, H, H, device="cuda")
    L.check(L.lib().sln_spade_conv(L.ptr(x.cuda()), B, Cin, H, H, L.ptr(wp), L.ptr(bp), Cout, rp, 3, 0, 0.0, L.ptr(y), L.current_stream_ptr()),
            "conv")
    scale = float(ref.abs().max())
    rms_hip = float(((y.cpu().double() - ref) ** 2).mean().sqrt()) / scale
    rms_cpu = float(((c32.double() - ref) ** 2).mean().sqrt()) / scale
    assert rms_hip <= 1.6 * rms_cpu + 1e-9, (rms_hip, rms_cpu)
    assert_close(y.cpu().numpy(), ref.numpy(), "blocked conv", rtol=2e-6, atol=1e-7)


def test_small_convolutions_on_two_streams_do_not_share_their_split_scratch():
    """The input-channel split of small launches (batch-1 calls: 1 024 -> 256 channels at 8 x 8 is 8 workgroups unsplit) leaves
    partial sums in a scratch buffer; two streams running such convolutions at the same time each need their own (round 4 kept one
    process-wide buffer).  Interleaved launches on two streams must give, bit for bit, what each stream gives alone."""
    L = pkg("_lib"); S = pkg("host.SPADE_related")
    Cin, Cout, H = 1024, 256, 8
    g = torch.Generator().manual_seed(7)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5
    wp, rp = S._pack(w.cuda())
    bp = torch.zeros(rp, device="cuda")
    xs = [torch.randn(1, Cin, H, H, generator=g).cuda() for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def conv(x, y, st):
        L.check(L.lib().sln_spade_conv(L.ptr(x), 1, Cin, H, H, L.ptr(wp), L.ptr(bp), Cout, rp, 3, 0, 0.0, L.ptr(y), C.c_void_p(st.cuda_stream)), "conv")
    import ctypes as C
    for st in streams:
        assert L.lib().sln_spade_prepare(C.c_void_p(st.cuda_stream)) == 0
    torch.cuda.synchronize()
    alone = []
    for x, st in zip(xs, streams):
        y = torch.empty(1, Cout, H, H, device="cuda")
        conv(x, y, st); st.synchronize()
        alone.append(y.clone())
    outs = [[torch.empty(1, Cout, H, H, device="cuda") for _ in range(40)] for _ in range(2)]
    for k in range(40):                                   # no synchronisation in between: the two streams' launches overlap
        for i in range(2):
            conv(xs[i], outs[i][k], streams[i])
    torch.cuda.synchronize()
    for i in range(2):
        for k in range(40):
            assert torch.equal(outs[i][k], alone[i]), "stream %d, launch %d" % (i, k)
    ref = F.conv2d(F.pad(xs[0].cpu().double(), (1, 1, 1, 1), mode="reflect"), w.double())
    assert_close(alone[0].cpu().numpy(), ref.numpy(), "split conv", rtol=2e-6, atol=1e-7)


def test_deterministic_mode_makes_the_generator_bit_identical_run_to_run():
    """The reference's CPU path (SPADE_related.py:128-149: LayerNorm2D sums, the SE pool) gives the same bits on every run; with
    SLN_DETERMINISTIC / sln_set_deterministic the HIP generator does too - the statistics the fused schedule adds with fp64 atomics
    in arrival order come from fixed-order kernels instead - and stays within rounding of the default mode."""
    L = pkg("_lib"); S = pkg("host.SPADE_related")
    cfg = spade_ref.SpadeConfig(ngf=16, nz=16, crop_size=128)
    sd = spade_ref.init_state(cfg, seed=5)
    G = S.SPADEGenerator4(cfg.semantic_nc, cfg.target_nc, cfg.nz, cfg.ngf, 'spectralspadelayer3x3', cfg.crop_size, 'normal')
    G.load_state_dict(sd); G = G.cuda().eval()
    seg, z = spade_ref.synth_input(cfg, 5, seed=4)
    seg, z = seg.cuda(), z.cuda()
    default = G(seg, z)
    try:
        L.lib().sln_set_deterministic(1)
        runs = []
        for _ in range(4):
            taps = {}
            out = G(seg, z, taps=taps)
            runs.append((out.clone(), {k: v.clone() for k, v in taps.items()}))
        shared = [G(seg[:1].contiguous(), z).clone() for _ in range(2)]          # one map, many z
    finally:
        L.lib().sln_set_deterministic(0)
    for out, taps in runs[1:]:
        assert torch.equal(out, runs[0][0])
        for k in taps:
            assert torch.equal(taps[k], runs[0][1][k]), k
    assert torch.equal(shared[0], shared[1])
    assert_close(runs[0][0].cpu().numpy(), default.cpu().numpy(), "deterministic vs default", rtol=1e-5, atol=2e-5)
    ref = spade_ref.generator(sd, cfg, seg.cpu(), z.cpu())
    assert_close(runs[0][0].cpu().numpy(), ref.numpy(), "deterministic vs oracle", rtol=1e-4, atol=1e-4)


def test_kept_planes_contract_release_and_copy(monkeypatch):
    """The kept gamma|beta planes: a batched call releases them; a deepcopy of the module carries neither the planes
    nor the packed weights and still computes the same image; a write that does not move the version counter is NOT seen (the
    documented contract) - and is caught when SLN_SPADE_MEMO_CHECK=1 asks for the checksum."""
    import copy
    S = pkg("host.SPADE_related"); L = pkg("_lib")
    cfg = spade_ref.SpadeConfig(**CASES["spade_small"][0])
    sd = spade_ref.init_state(cfg, seed=7)
    G = S.SPADEGenerator4(cfg.semantic_nc, cfg.target_nc, cfg.nz, cfg.ngf, 'spectralspadelayer3x3', cfg.crop_size, 'normal')
    G.load_state_dict(sd); G = G.cuda().eval()
    seg, _ = spade_ref.synth_input(cfg, 2, seed=5)
    z = torch.from_numpy(np.random.default_rng(4).standard_normal((2, cfg.nz)).astype(np.float32)).cuda()
    total = seg[:1].cuda().contiguous()
    for _ in range(3):
        a = G(total, z[:1])
    assert G._map_memo is not None and len(G._map_memo["gb"]) > 0
    G(seg.cuda(), z)                                               # a batched call: the planes (and the pinned map) go
    assert G._map_memo is None
    for _ in range(3):
        G(total, z[:1])
    G2 = copy.deepcopy(G)
    assert G2._map_memo is None and G2._packed is None
    assert_close(G2(total, z[:1]).cpu().numpy(), a.cpu().numpy(), "deepcopy", rtol=1e-5, atol=2e-5)   # (its first call takes the fused path)
    # a silent write (no version bump)
    monkeypatch.setenv("SLN_SPADE_MEMO_CHECK", "1")
    G.clear_map_cache()
    for _ in range(3):
        G(total, z[:1])
    seg2, _ = spade_ref.synth_input(cfg, 1, seed=6)
    total.data.copy_(seg2.cuda())                                  # .data: the version counter of `total` does not move
    with pytest.raises(L.SlnError):
        G(total, z[:1])
    G.clear_map_cache()
    ref2 = spade_ref.generator(sd, cfg, seg2, z[:1].cpu()).numpy()
    assert_close(G(total, z[:1]).cpu().numpy(), ref2, "after clear_map_cache", rtol=1e-4, atol=1e-4)
