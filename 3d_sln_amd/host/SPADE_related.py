"""HIP-backed drop-in for the part of the reference's ``models/SPADE_related.py`` that is alive:
``SPADEGenerator4`` (:1507-1605) with ``SPADEResnetBlock4`` (:1457-1505), ``SPADE4`` (:1404-1454),
``LayerNorm2D`` (:128-149) and ``SEBlock2`` (:70-85) as instantiated by testing/test_SPADE_shade.py:9
(``SPADEGenerator4(41, 3, 256, 64, 'spectralspadelayer3x3', 256, 'normal')``, inference only, README.md:60-61).

The module tree only owns parameters with the reference's names (230 ``state_dict`` keys incl. the
spectral-norm triplets ``weight_orig / weight_u / weight_v``) so the authors' ``latest_net_G_AB.pth`` loads
unchanged; ``forward`` runs on libsln_hip.so (csrc/spade.hip).  Weights are folded (spectral sigma) and packed
into the kernels' layout once per parameter version.
"""
import ctypes as C
import os
import re
import threading

import torch
import torch.nn as nn
from torch.nn.utils import spectral_norm

from .. import _lib

_STREAM_TLS = threading.local()          # the stream of the forward() in flight on this host thread (SPADEGenerator4._st)

NHIDDEN = 128


class SEBlock2(nn.Module):
    def __init__(self, channel, reduction=4):
        super().__init__()
        self.fc = nn.Sequential(nn.Linear(channel, channel // reduction, bias=False), nn.ReLU(inplace=True),
                                nn.Linear(channel // reduction, channel, bias=False), nn.Sigmoid())


class SPADE4(nn.Module):
    def __init__(self, config_text, norm_nc, label_nc):
        super().__init__()
        parsed = re.search(r'spade(\D+)(\d)x\d', config_text)
        if parsed is None or parsed.group(1) != 'layer' or int(parsed.group(2)) != 3:
            raise NotImplementedError("only 'spadelayer3x3' (LayerNorm2D, 3x3) is on the HIP path: %r" % config_text)
        self.mlp_preshared_depth = nn.Sequential(nn.ReflectionPad2d(1), nn.Conv2d(1, NHIDDEN // 8, 3), nn.LeakyReLU(inplace=True))
        self.mlp_shared = nn.Sequential(nn.ReflectionPad2d(1), nn.Conv2d(NHIDDEN // 8 + label_nc - 1, NHIDDEN, 3), nn.ReLU(inplace=True))
        self.mlp_gamma = nn.Sequential(nn.ReflectionPad2d(1), nn.Conv2d(NHIDDEN, norm_nc, 3))
        self.mlp_beta = nn.Sequential(nn.ReflectionPad2d(1), nn.Conv2d(NHIDDEN, norm_nc, 3))
        self.norm_nc = norm_nc


class SPADEResnetBlock4(nn.Module):
    def __init__(self, fin, fout, norm, semantic_nc):
        super().__init__()
        if 'spectral' not in norm:
            raise NotImplementedError("the reference instantiates the spectral variant only")
        self.fin, self.fout, self.fmiddle = fin, fout, min(fin, fout)
        self.learned_shortcut = fin != fout
        self.conv_0 = nn.Sequential(nn.ReflectionPad2d(1), spectral_norm(nn.Conv2d(fin, self.fmiddle, 3)))
        self.conv_1 = nn.Sequential(nn.ReflectionPad2d(1), spectral_norm(nn.Conv2d(self.fmiddle, fout, 3)))
        self.se = SEBlock2(fout, reduction=8)
        if self.learned_shortcut:
            self.conv_s = spectral_norm(nn.Conv2d(fin, fout, 1, bias=False))
        cfg = norm.replace('spectral', '')
        self.norm_0 = SPADE4(cfg, fin, semantic_nc)
        self.norm_1 = SPADE4(cfg, self.fmiddle, semantic_nc)
        if self.learned_shortcut:
            self.norm_s = SPADE4(cfg, fin, semantic_nc)


def _fold_sn(sd, prefix):
    w = sd[prefix + ".weight_orig"]
    sigma = torch.dot(sd[prefix + ".weight_u"], torch.mv(w.reshape(w.shape[0], -1), sd[prefix + ".weight_v"]))
    return w / sigma


def _pack(w, rows_pad=None):
    """[Cout, Cin, k, k] -> [k*k, Cin, rows_pad] (rows contiguous, zero padded to a multiple of 64)."""
    co, ci, k, _ = w.shape
    rp = rows_pad or (co + 63) // 64 * 64
    out = torch.zeros(k * k, ci, rp, dtype=torch.float32, device=w.device)
    out[:, :, :co] = w.permute(2, 3, 1, 0).reshape(k * k, ci, co)
    return out.contiguous(), rp


def _pack_gamma_beta(wg, bg, wb, bb):
    """rows [64g, 64g+32) = gamma of channels [32g, 32g+32), rows [64g+32, 64g+64) = their beta."""
    c = wg.shape[0]
    groups = (c + 31) // 32
    rp = 64 * groups
    idx = torch.arange(c, device=wg.device)
    rg, rb = (idx // 32) * 64 + idx % 32, (idx // 32) * 64 + 32 + idx % 32
    w = torch.zeros(9, wg.shape[1], rp, dtype=torch.float32, device=wg.device)
    w[:, :, rg] = wg.permute(2, 3, 1, 0).reshape(9, wg.shape[1], c)
    w[:, :, rb] = wb.permute(2, 3, 1, 0).reshape(9, wb.shape[1], c)
    b = torch.zeros(rp, dtype=torch.float32, device=wg.device)
    b[rg], b[rb] = bg, bb
    return w.contiguous(), b.contiguous(), rp


class SPADEGenerator4(nn.Module):
    def __init__(self, semantic_nc, target_nc, nz, ngf, norm, crop_size, n_up):
        super().__init__()
        if n_up != 'normal':
            raise NotImplementedError("n_up='more'/'most' crash in the reference itself (self.up is never defined, :1587,1600)")
        if nz <= 0:
            raise NotImplementedError("the reference instantiates the z-conditioned generator (nz=256)")
        nf = ngf
        self.nf, self.n_up, self.nz, self.has_z = ngf, n_up, nz, True
        self.semantic_nc, self.target_nc, self.crop_size = semantic_nc, target_nc, crop_size
        self.sw = self.sh = crop_size // 32
        self.fc = nn.Linear(nz, 16 * nf * self.sw * self.sh)
        self.head_0 = SPADEResnetBlock4(16 * nf, 16 * nf, norm, semantic_nc)
        self.G_middle_0 = SPADEResnetBlock4(16 * nf, 16 * nf, norm, semantic_nc)
        self.G_middle_1 = SPADEResnetBlock4(16 * nf, 16 * nf, norm, semantic_nc)
        self.up_0 = SPADEResnetBlock4(16 * nf, 8 * nf, norm, semantic_nc)
        self.up_1 = SPADEResnetBlock4(8 * nf, 4 * nf, norm, semantic_nc)
        self.up_2 = SPADEResnetBlock4(4 * nf, 2 * nf, norm, semantic_nc)
        self.up_3 = SPADEResnetBlock4(2 * nf, 1 * nf, norm, semantic_nc)
        self.conv_img = nn.Conv2d(nf, target_nc, 5, padding=2)
        self._packed = None
        self._packed_key = None
        self.unfused = False          # True: one launch per module (the round-1 schedule; kept for A/B runs and odd sizes)

    # ------------------------------------------------------------------ weight packing
    reuse_map_planes = True          # see forward(): gamma|beta planes of a map kept between consecutive batch-1 calls on it
    _map_repeat = False
    _map_memo = None
    _pack_gen = 0

    def clear_map_cache(self):
        """Drop the gamma|beta planes kept for the last semantic map (about 0.2 GB at 256x256) and the captured batch-1 call."""
        self._map_memo = None
        self._b1_graph = None

    # Batch-1 calls on ONE map (testing/test_SPADE_shade.py:77-79: 50 z per room): from the third consecutive call on the same
    # tensor the launch sequence is fixed - the planes are kept, nothing depends on z but the fc input - so that call is captured
    # into a hipGraph once per map and the later ones are a copy of z, one graph launch and a copy of the image: the ~1 ms of
    # python / ctypes per call (57 launches) leaves the loop.  OFF by default: measured on the bench's 50 x 1 loop (a new map, 50 calls)
    # 103 ms per room with the capture against 91 ms eager, identical images - a batch-1 call is ~1.7 ms of GPU time (57 dependent
    # launches on one image), which the ~1 ms of host work already hides behind; the capture only adds its set-up (two eager calls
    # on a side stream, capture, instantiation) to every new map.  The one-call-many-z form (forward(map, z[50])) is the fast path
    # (1 082 images/s); this switch is kept for callers whose host is slower than the GPU.
    graph_batch1 = False

    def _graph_eligible(self, input, z, taps):
        return (self.graph_batch1 and self.reuse_map_planes and taps is None and z is not None and input.dim() == 4 and input.shape[0] == 1 and
                z.shape[0] == 1 and not self.unfused and os.environ.get("SLN_SPADE_MEMO_CHECK") != "1" and
                not torch.cuda.is_current_stream_capturing())

    def _graph_forward(self, input, z):
        # (the weights' signature as _pack_all takes it: a captured call has the packed weights' addresses baked in, and a
        #  load_state_dict / optimizer step is only noticed by the NEXT _pack_all - which a replay never reaches)
        wkey = tuple((v.data_ptr(), v._version) for v in self.parameters()) + tuple((v.data_ptr(), v._version) for v in self.buffers())
        key = (input.data_ptr(), input._version, tuple(input.shape), input.dtype, str(input.device), wkey)
        ent = getattr(self, "_b1_graph", None)
        if ent is None or ent["key"] != key:
            side = getattr(self, "_b1_stream", None)
            if side is None or side.device != input.device:
                side = self._b1_stream = torch.cuda.Stream(device=input.device)     # ONE capture stream per module: its split scratch is allocated once
            ent = self._b1_graph = dict(key=key, calls=0, graph=None, stream=side, keep=input)
            # the split scratch of the small convolutions is per stream and cannot be allocated while that stream is captured
            _lib.check(_lib.lib().sln_spade_prepare(C.c_void_p(ent["stream"].cuda_stream)), "sln_spade_prepare")
        if ent["graph"] is not None:
            ent["z"].copy_(z)
            ent["graph"].replay()                              # on the caller's current stream
            return ent["out"].clone()
        cur, s = torch.cuda.current_stream(input.device), ent["stream"]
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            if ent["calls"] < 2:                               # 1st call: the fused path; 2nd: computes and keeps the map's planes
                out = self._eager_forward(input, z, None)
                ent["calls"] += 1
            else:
                ent["z"] = z.detach().float().contiguous().clone()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=s):
                    ent["out"] = self._eager_forward(input, ent["z"], None)
                ent["graph"], ent["memo"] = g, self._map_memo   # (the captured launches read the kept planes: they live with the graph)
                g.replay()
                out = ent["out"].clone()
        cur.wait_stream(s)
        out.record_stream(cur)
        return out

    def __getstate__(self):
        """copy.deepcopy / pickling: the packed weights, the kept planes and the pinned input tensor are caches, not state."""
        st = self.__dict__.copy()
        st["_map_memo"] = st["_packed"] = st["_packed_key"] = st["_cat_cache"] = st["_b1_graph"] = st["_b1_stream"] = None
        return st

    def _pack_all(self):
        # the signature of the weights - (address, version counter) of every parameter and buffer - is taken on every call; walking
        # parameters() / buffers() costs ~0.1 ms where two state_dict() walks with 230 detach().float() copies cost ~1 ms, which at
        # batch 1 was half of the host time of a call (tools/lab/spade_b1_hostprof.py)
        key = tuple((v.data_ptr(), v._version) for v in self.parameters()) + tuple((v.data_ptr(), v._version) for v in self.buffers())
        if self._packed is not None and key == self._packed_key:
            return self._packed
        sd = {k: v.detach().float() for k, v in self.state_dict().items()}
        P = {}
        for name in ("head_0", "G_middle_0", "G_middle_1", "up_0", "up_1", "up_2", "up_3"):
            blk = getattr(self, name)
            e = {}
            for cn in ("conv_0", "conv_1"):
                w, rp = _pack(_fold_sn(sd, "%s.%s.1" % (name, cn)))
                b = torch.zeros(rp, device=w.device); b[:sd["%s.%s.1.bias" % (name, cn)].numel()] = sd["%s.%s.1.bias" % (name, cn)]
                e[cn] = (w, b, rp)
            if blk.learned_shortcut:
                w, rp = _pack(_fold_sn(sd, name + ".conv_s"))
                e["conv_s"] = (w, None, rp)
            for nn_ in ("norm_0", "norm_1", "norm_s"):
                if not hasattr(blk, nn_):
                    continue
                p = "%s.%s" % (name, nn_)
                wsh, rps = _pack(sd[p + ".mlp_shared.1.weight"])
                bsh = torch.zeros(rps, device=wsh.device); bsh[:NHIDDEN] = sd[p + ".mlp_shared.1.bias"]
                wgb, bgb, rpg = _pack_gamma_beta(sd[p + ".mlp_gamma.1.weight"], sd[p + ".mlp_gamma.1.bias"],
                                                 sd[p + ".mlp_beta.1.weight"], sd[p + ".mlp_beta.1.bias"])
                e[nn_] = dict(wpd=sd[p + ".mlp_preshared_depth.1.weight"].reshape(NHIDDEN // 8, 9).contiguous(),
                              bpd=sd[p + ".mlp_preshared_depth.1.bias"].contiguous(), wsh=wsh, bsh=bsh, rps=rps,
                              wgb=wgb, bgb=bgb, rpg=rpg)
            e["se0"], e["se2"] = sd[name + ".se.fc.0.weight"].contiguous(), sd[name + ".se.fc.2.weight"].contiguous()
            P[name] = e
        self._map_memo = None            # planes computed with the old weights
        P["fc_w"], P["fc_b"] = sd["fc.weight"].contiguous(), sd["fc.bias"].contiguous()
        P["img_w"], P["img_b"] = sd["conv_img.weight"].contiguous(), sd["conv_img.bias"].contiguous()
        self._packed, self._packed_key = P, key
        self._pack_gen += 1              # new weights: planes kept for a map are stale (forward() compares this)
        return P

    # ------------------------------------------------------------------ HIP launches

    def _st(self):
        # forward() looks the current stream up once; torch.cuda.current_stream() per launch was 59 look-ups (~0.3 ms) per call
        st = getattr(_STREAM_TLS, "st", None)
        return st if st is not None else _lib.current_stream_ptr()

    def _ln_stats(self, x):
        B = x.shape[0]
        stats = torch.empty(B, 2, device=x.device)
        scratch = torch.empty(16 * B, dtype=torch.float64, device=x.device)        # one 128-byte line per sample
        _lib.check(_lib.lib().sln_layernorm_stats(_lib.ptr(x), B, x[0].numel(), 1e-5, _lib.ptr(scratch), _lib.ptr(stats), self._st()),
                   "sln_layernorm_stats")
        return stats

    def _spade(self, e, x, stats, seg, leaky, x_up=False):
        """SPADE4.forward (:1438-1454) + the following actvn (:1503-1505) fused into the modulation conv.
        x_up: x is stored at half the resolution of seg and stands for its nearest x2 upsampling (never materialised)."""
        L = _lib.lib()
        B, C = x.shape[:2]
        H, W = seg.shape[2:]
        nd = NHIDDEN // 8
        if seg.shape[0] == 1 and (B > 1 or self._map_repeat) and (H * W) % 4 == 0:
            return self._spade_shared(e, x, stats, seg, leaky, x_up)
        if seg.shape[0] != B:
            seg = seg.expand(B, -1, -1, -1).contiguous()
        cat = self._cat_buffer(seg, nd)
        _lib.check(L.sln_spade_depth_concat(_lib.ptr(seg), B, seg.shape[1], H, W, _lib.ptr(e["wpd"]), _lib.ptr(e["bpd"]), nd,
                                            _lib.ptr(cat), 0, self._st()), "sln_spade_depth_concat")
        actv = torch.empty(B, NHIDDEN, H, W, device=x.device)
        _lib.check(L.sln_spade_conv(_lib.ptr(cat), B, cat.shape[1], H, W, _lib.ptr(e["wsh"]), _lib.ptr(e["bsh"]), NHIDDEN, e["rps"], 3,
                                    1, 0.0, _lib.ptr(actv), self._st()), "sln_spade_conv(shared)")
        out = torch.empty(B, C, H, W, device=x.device)
        _lib.check(L.sln_spade_modulate_up(_lib.ptr(actv), B, NHIDDEN, H, W, _lib.ptr(e["wgb"]), _lib.ptr(e["bgb"]), C, e["rpg"],
                                           _lib.ptr(x), 1 if x_up else 0, _lib.ptr(stats), 2 if leaky else 0, 0.2, _lib.ptr(out),
                                           self._st()), "sln_spade_modulate_up")
        return out

    def _cat_buffer(self, seg, nd):
        """[depth features (nd) | masks] input of mlp_shared for this resolution.  The mask channels are the same for every
        SPADE layer of a resolution: they are copied once per forward, later layers only overwrite the nd depth features."""
        key = (seg.data_ptr(), tuple(seg.shape))
        cache = getattr(self, "_cat_cache", None)                # a dict only while forward() runs (the pyramid is alive)
        buf = cache.get(key) if cache is not None else None
        if buf is None:
            B, Cs, H, W = seg.shape
            buf = torch.empty(B, nd + Cs - 1, H, W, device=seg.device)
            buf[:, nd:] = seg[:, 1:]
            if cache is not None:
                cache[key] = buf
        return buf

    def _spade_shared(self, e, x, stats, seg, leaky, x_up=False):
        """One semantic map for the whole batch (the reference broadcasts gamma/beta [1,C,H,W] in that case, and
        colorize_with_spade is exactly that use: 50 z per room).  gamma/beta - 72 % of the generator's MACs - are computed
        once per map instead of once per sample; the per-sample part is an HBM-bound elementwise pass."""
        L = _lib.lib()
        B, C = x.shape[:2]
        H, W = seg.shape[2:]
        nd = NHIDDEN // 8
        memo = self._map_memo["gb"] if self._map_repeat else None          # gamma|beta planes of THIS map, kept between calls
        gb = memo.get(id(e)) if memo is not None else None
        if gb is None:
            cat = self._cat_buffer(seg, nd)
            _lib.check(L.sln_spade_depth_concat(_lib.ptr(seg), 1, seg.shape[1], H, W, _lib.ptr(e["wpd"]), _lib.ptr(e["bpd"]), nd,
                                                _lib.ptr(cat), 0, self._st()), "sln_spade_depth_concat")
            actv = torch.empty(1, NHIDDEN, H, W, device=x.device)
            _lib.check(L.sln_spade_conv(_lib.ptr(cat), 1, cat.shape[1], H, W, _lib.ptr(e["wsh"]), _lib.ptr(e["bsh"]), NHIDDEN, e["rps"], 3,
                                        1, 0.0, _lib.ptr(actv), self._st()), "sln_spade_conv(shared)")
            gb = torch.empty(1, e["rpg"], H, W, device=x.device)
            _lib.check(L.sln_spade_conv(_lib.ptr(actv), 1, NHIDDEN, H, W, _lib.ptr(e["wgb"]), _lib.ptr(e["bgb"]), e["rpg"], e["rpg"], 3,
                                        0, 0.0, _lib.ptr(gb), self._st()), "sln_spade_conv(gamma|beta)")
            if memo is not None:
                memo[id(e)] = gb
        out = torch.empty(B, C, H, W, device=x.device)
        _lib.check(L.sln_spade_apply_up(_lib.ptr(x), 1 if x_up else 0, _lib.ptr(gb), B, C, H, W, e["rpg"], _lib.ptr(stats),
                                        2 if leaky else 0, 0.2, _lib.ptr(out), self._st()), "sln_spade_apply_up")
        return out

    def _conv(self, x, wbr, cout, ks, ln_acc=None, gap_acc=None):
        w, b, rp = wbr
        B, _, H, W = x.shape
        y = torch.empty(B, cout, H, W, device=x.device)
        _lib.check(_lib.lib().sln_spade_conv_sums(_lib.ptr(x), B, x.shape[1], H, W, _lib.ptr(w), _lib.ptr(b), cout, rp, ks, 0, 0.0,
                                                  _lib.ptr(y), _lib.ptr(ln_acc) if ln_acc is not None else None,
                                                  _lib.ptr(gap_acc) if gap_acc is not None else None, self._st()), "sln_spade_conv_sums")
        return y

    def _block(self, name, x, x_up, stats_x, seg, tail, want_stats=True, tap=None):
        """SPADEResnetBlock4.forward (:1487-1502) and the nn.Upsample behind it (:1585-1600), the HBM-bound passes folded into
        their neighbours: LayerNorm2D sums of conv_0's output and SEBlock2's average pool come out of the conv epilogues, the
        residual sum is written once by `sln_block_tail` together with the statistics of the NEXT block's input.
        x_up: x stands for its nearest x2 upsampling.  tail: None (no upsampling follows), 'nearest' (the result stays at this
        resolution, its consumers read it through the upsampling), 'bilinear' (written upsampled).
        Returns (out, out_up, stats_out)."""
        blk, e = getattr(self, name), self._packed[name]
        L = _lib.lib()
        B = x.shape[0]
        H, W = seg.shape[2:]
        # Deterministic mode (SLN_DETERMINISTIC / sln_set_deterministic): the sums that conv epilogues and the tail add with fp64
        # atomics in arrival order are taken by fixed-order kernels instead (sln_layernorm_stats with one slot per block, the
        # tree-reduced pool of sln_block_tail) - the reference's CPU path gives the same bits on every run, so does this one then.
        det = bool(L.sln_get_deterministic())
        # fp64 accumulators of the block: one zero-filled buffer per forward holds every block's (seven fills per call were 2 % of a
        # batch-1 call); the tail launch clears its own (ln_out)
        n_acc = 32 * B + B * blk.fout
        pool = getattr(self, "_acc_pool", None)
        if pool is not None and self._acc_off + n_acc <= pool.numel():
            accs = pool[self._acc_off:self._acc_off + n_acc]; self._acc_off += n_acc
        else:
            accs = torch.zeros(n_acc, dtype=torch.float64, device=x.device)
        ln_dx, ln_out, gap = accs[:16 * B], accs[16 * B:32 * B], accs[32 * B:]
        if blk.learned_shortcut:
            x_s, xs_up = self._conv(self._spade(e["norm_s"], x, stats_x, seg, False, x_up), e["conv_s"], blk.fout, 1), 0
        else:
            x_s, xs_up = x, 1 if x_up else 0
        dx = self._conv(self._spade(e["norm_0"], x, stats_x, seg, True, x_up), e["conv_0"], blk.fmiddle, 3, ln_acc=None if det else ln_dx)
        if det:
            stats_dx = self._ln_stats(dx)
        else:
            stats_dx = torch.empty(B, 2, device=x.device)
            _lib.check(L.sln_layernorm_finalize(_lib.ptr(ln_dx), B, blk.fmiddle * H * W, 1, 1e-5, _lib.ptr(stats_dx), self._st()),
                       "sln_layernorm_finalize")
        dx = self._conv(self._spade(e["norm_1"], dx, stats_dx, seg, True), e["conv_1"], blk.fout, 3, gap_acc=None if det else gap)
        gap_p = None if det else _lib.ptr(gap)
        up_mode = 1 if tail == 'bilinear' else -1
        k = 2 if tail == 'bilinear' else 1
        out = torch.empty(B, blk.fout, k * H, k * W, device=x.device)
        stats = torch.empty(B, 2, device=x.device) if want_stats else None
        scratch = torch.empty(2 * B * blk.fout, device=x.device)
        if tap is not None:                       # the block's own output, when the caller asked for it and `out` is upsampled
            tap[name] = torch.empty(B, blk.fout, H, W, device=x.device)
            _lib.check(L.sln_block_tail(_lib.ptr(x_s), xs_up, _lib.ptr(dx), B, blk.fout, H, W, gap_p, _lib.ptr(e["se0"]),
                                        _lib.ptr(e["se2"]), _lib.ptr(scratch), -1, _lib.ptr(tap[name]), None, 1, 1e-5, None, self._st()),
                       "sln_block_tail(tap)")
        fused_stats = want_stats and not det
        _lib.check(L.sln_block_tail(_lib.ptr(x_s), xs_up, _lib.ptr(dx), B, blk.fout, H, W, gap_p, _lib.ptr(e["se0"]),
                                    _lib.ptr(e["se2"]), _lib.ptr(scratch), up_mode, _lib.ptr(out), _lib.ptr(ln_out) if fused_stats else None,
                                    4 if tail == 'nearest' else 1, 1e-5, _lib.ptr(stats) if fused_stats else None, self._st()),
                   "sln_block_tail")
        if want_stats and det:
            scr = torch.empty(16 * B, dtype=torch.float64, device=x.device)
            _lib.check(L.sln_layernorm_stats(_lib.ptr(out), B, out[0].numel(), 1e-5, _lib.ptr(scr), _lib.ptr(stats), self._st()),
                       "sln_layernorm_stats")
            if tail == 'nearest':                   # the consumers read `out` through nearest x2: statistics of THAT tensor
                _lib.check(L.sln_layernorm_finalize(_lib.ptr(scr), B, out[0].numel(), 4, 1e-5, _lib.ptr(stats), self._st()),
                           "sln_layernorm_finalize")
        return out, tail == 'nearest', stats

    def _block_unfused(self, name, x, seg):
        """SPADEResnetBlock4.forward (:1487-1502), one launch per module (one map for many z, sizes the fused tail does not take)."""
        blk, e = getattr(self, name), self._packed[name]
        stats_x = self._ln_stats(x)                                  # norm_0 and norm_s normalise the same tensor
        if blk.learned_shortcut:
            x_s = self._conv(self._spade(e["norm_s"], x, stats_x, seg, leaky=False), e["conv_s"], blk.fout, 1)
        else:
            x_s = x
        dx = self._conv(self._spade(e["norm_0"], x, stats_x, seg, leaky=True), e["conv_0"], blk.fmiddle, 3)
        dx = self._conv(self._spade(e["norm_1"], dx, self._ln_stats(dx), seg, leaky=True), e["conv_1"], blk.fout, 3)
        B, Cc, H, W = dx.shape
        out = torch.empty_like(dx)
        scratch = torch.empty(2 * B * Cc, device=dx.device)
        _lib.check(_lib.lib().sln_se_scale_add(_lib.ptr(x_s), _lib.ptr(dx), B, Cc, H * W, _lib.ptr(e["se0"]), _lib.ptr(e["se2"]),
                                               _lib.ptr(scratch), _lib.ptr(out), self._st()), "sln_se_scale_add")
        return out

    def _resize(self, t, size, mode):
        B, Cc, H, W = t.shape
        if (H, W) == (size, size):
            return t
        out = torch.empty(B, Cc, size, size, device=t.device)
        _lib.check(_lib.lib().sln_resize(_lib.ptr(t), B * Cc, H, W, size, size, mode, _lib.ptr(out), self._st()), "sln_resize")
        return out

    def _up(self, t, mode):
        B, Cc, H, W = t.shape
        out = torch.empty(B, Cc, 2 * H, 2 * W, device=t.device)
        _lib.check(_lib.lib().sln_upsample2x(_lib.ptr(t), B * Cc, H, W, mode, _lib.ptr(out), self._st()), "sln_upsample2x")
        return out

    def forward(self, input, z=None, taps=None):
        """seg [B, semantic_nc, S, S] (channel 0 depth, 1.. masks), z [B, nz] -> image [B, target_nc, S, S] in (-1, 1)."""
        if input.device.type != 'cuda':
            raise _lib.SlnError("SPADEGenerator4 runs on the MI355X only (no CPU fallback)")
        if self._graph_eligible(input, z, taps):
            return self._graph_forward(input, z)
        return self._eager_forward(input, z, taps)

    def _eager_forward(self, input, z, taps):
        # one stream look-up per call, kept per THREAD and restored (not cleared) on the way out: a nested forward (a hook) or a
        # concurrent one on another host thread / stream no longer redirects the rest of this call's launches
        tl = _STREAM_TLS
        prev = getattr(tl, "st", None)
        tl.st = _lib.current_stream_ptr()
        try:
            return self._forward(input, z, taps)
        finally:
            tl.st = prev

    def _forward(self, input, z, taps):
        with torch.no_grad():
            seg = input.float().contiguous()
            B = seg.shape[0]
            if z is not None and seg.shape[0] == 1 and z.shape[0] > 1:
                B = z.shape[0]                                          # one map, many z: gamma/beta broadcast as in the reference
            elif z is not None and z.shape[0] != B:
                raise RuntimeError("z has %d rows for %d semantic maps" % (z.shape[0], B))
            if z is None:
                print("Missing z vector, sampling from normal")
                z = torch.randn(B, self.nz, dtype=torch.float32, device=seg.device)
            P = self._pack_all()
            L = _lib.lib()
            # colorize_with_spade calls the model once per z with the SAME semantic map (testing/test_SPADE_shade.py:77-79: 50 calls
            # at batch 1).  gamma / beta depend on the map only - 72 % of the MACs -: from the SECOND consecutive call with the very
            # same input tensor (same object, unmodified: data pointer, shape and version counter; same packed parameters; same
            # stream) the planes are computed once, kept, and the later calls run only the per-sample part.  (The first call of a
            # map takes the fused path, which never materialises them: single calls on ever-new maps pay nothing.)
            # CONTRACT of the kept planes: "the same map" means the same tensor object whose version counter did not move.  Writes
            # that do not bump the counter (a raw kernel through data_ptr(), `input.data.copy_`, a DLPack / numpy alias) are NOT
            # seen: call clear_map_cache() after such a write, or set reuse_map_planes = False.  SLN_SPADE_MEMO_CHECK=1 compares
            # a checksum of the map on every call (one reduction) and raises on a silent change.  The planes (and the input they
            # belong to) are released by the next call on another map, by any batched call, by a repack of the weights and by
            # clear_map_cache(); a copy / pickle of the module does not carry them.
            mkey = (input.data_ptr(), input._version, tuple(input.shape), input.dtype, self._pack_gen,
                    int(torch.cuda.current_stream().cuda_stream))
            if seg.shape[0] != 1:
                self._map_memo = None
            memo = getattr(self, "_map_memo", None)
            self._map_repeat = bool(self.reuse_map_planes and seg.shape[0] == 1 and memo is not None and memo["key"] == mkey)
            check = os.environ.get("SLN_SPADE_MEMO_CHECK") == "1" and seg.shape[0] == 1
            csum = float(seg.double().sum().item()) if check else None
            if self._map_repeat and check and memo.get("csum") != csum:
                raise _lib.SlnError("SPADEGenerator4: the semantic map changed without its version counter moving (kept gamma|beta "
                                    "planes would be stale): call clear_map_cache() after writing through data_ptr() / .data")
            if not self._map_repeat:
                self._map_memo = dict(key=mkey, gb={}, csum=csum,
                                      keep=input if seg.shape[0] == 1 and self.reuse_map_planes else None) if seg.shape[0] == 1 else None
            self._cat_cache = {}                                      # per-forward: keyed by the pyramid level's storage
            nfc = 16 * self.nf * self.sw * self.sh
            x = torch.empty(B, nfc, device=seg.device)
            _lib.check(L.sln_linear_forward(_lib.ptr(z.float().contiguous()), B, self.nz, _lib.ptr(P["fc_w"]), _lib.ptr(P["fc_b"]),
                                            _lib.ptr(x), nfc, None, -1, self._st()), "sln_linear_forward(fc)")
            x = x.view(B, 16 * self.nf, self.sh, self.sw)
            S = seg.shape[2]
            pyr = {S: seg}
            for r in (self.sw * 2, self.sw * 4, self.sw * 8, self.sw * 16):
                pyr[r] = self._resize(seg, r, 1)                      # SPADE4's bilinear resize of the full-size map (:1444)
            seg_1 = self._resize(seg, self.sw, 0)                     # nearest (:1579); SPADE4's bilinear to the same size is identity

            chain = (("head_0", seg_1, 'nearest'), ("G_middle_0", pyr[self.sw * 2], None), ("G_middle_1", pyr[self.sw * 2], 'nearest'),
                     ("up_0", pyr[self.sw * 4], 'nearest'), ("up_1", pyr[self.sw * 8], 'nearest'), ("up_2", pyr[self.sw * 16], 'bilinear'),
                     ("up_3", pyr[S], None))
            fused = self.sw % 4 == 0 and self.sh % 4 == 0 and not self.unfused
            if fused:
                x, x_up, stats = x.contiguous(), False, None
                self._acc_pool = torch.zeros(sum(32 * B + B * getattr(self, nm).fout for nm, _, _ in chain), dtype=torch.float64, device=seg.device)
                self._acc_off = 0
                stats = self._ln_stats(x)
                for name, s, tail in chain:
                    x, x_up, stats = self._block(name, x, x_up, stats, s, tail, want_stats=name != "up_3",
                                                 tap=taps if taps is not None and tail == 'bilinear' else None)
                    if taps is not None and tail != 'bilinear':        # the block's output as the reference module returns it
                        taps[name] = x
            else:
                x = x.contiguous()
                for name, s, tail in chain:
                    x = self._block_unfused(name, x, s)
                    if taps is not None:
                        taps[name] = x
                    if tail is not None:
                        x = self._up(x, 1 if tail == 'bilinear' else 0)
            out = torch.empty(B, self.target_nc, S, S, device=seg.device)
            _lib.check(L.sln_conv_img_tanh(_lib.ptr(x), B, self.nf, S, S, _lib.ptr(P["img_w"]), _lib.ptr(P["img_b"]), self.target_nc,
                                           _lib.ptr(out), self._st()), "sln_conv_img_tanh")
            self._cat_cache = None
            self._acc_pool = None
            return out
