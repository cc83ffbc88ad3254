"""GPU parity tests of the differentiable rasterizer (run with -m gpu).

Checker: oracle/raster_ref.{cpp,py} - the CPU restatement that defines the semantics ("parity unpinned",
the third-party neural_renderer package is absent).  Indices must be bit-exact, floats within 1e-4.
"""
import numpy as np
import pytest
import torch

from conftest import pkg
from parity import assert_close

pytestmark = pytest.mark.gpu

from oracle import raster_ref as rr       # noqa: E402


def _faces_of_room(seed, target_faces, image_size):
    V, F, ranges, box = rr.synth_room(seed, n_objects=12 if target_faces > 500 else 4, target_faces=target_faces)
    K, R, t = rr.get_cam_mat(torch.from_numpy(box))
    v = torch.from_numpy(V)[None]
    f = torch.from_numpy(F)[None]
    f2 = torch.cat((f, f[:, :, [2, 1, 0]]), 1)
    fxyz = rr.vertices_to_faces(rr.project(v, K, R, t, 512), f2)
    return fxyz.contiguous(), (V, F, ranges, box)


def _hip_forward(fxyz, image_size, near, far):
    L = pkg("_lib")
    fd = fxyz.cuda().contiguous()
    B, F = fd.shape[:2]
    fi = torch.empty(B, image_size, image_size, dtype=torch.int32, device="cuda")
    w = torch.empty(B, image_size, image_size, 3, device="cuda"); d = torch.empty(B, image_size, image_size, device="cuda")
    ws = torch.empty(int(L.lib().sln_raster_workspace_bytes(B, F)), dtype=torch.uint8, device="cuda")
    L.check(L.lib().sln_raster_forward(L.ptr(fd), B, F, image_size, near, far, L.ptr(ws), L.ptr(fi), L.ptr(w), L.ptr(d),
                                       L.current_stream_ptr()), "fwd")
    return fd, fi, w, d


@pytest.mark.parametrize("image_size,target", [(256, 2000), (64, 300), (100, 700)])
def test_forward_bit_exact(image_size, target):
    fx = torch.cat([_faces_of_room(s, target, image_size)[0] for s in (1, 2)], 0) if target != 700 else _faces_of_room(5, target, image_size)[0]
    rfi, rw, rd = rr.nmr_forward(fx.numpy(), image_size, 0.001, 100.0)
    fd, fi, w, d = _hip_forward(fx, image_size, 0.001, 100.0)
    assert (fi.cpu().numpy() == rfi).all(), "face index map differs at %d pixels" % int((fi.cpu().numpy() != rfi).sum())
    assert (rfi >= 0).mean() > 0.5
    assert (w.cpu().numpy() == rw).all() and (d.cpu().numpy() == rd).all()


def test_random_soup_bit_exact_and_ties():
    rng = np.random.default_rng(0)
    F = 1500
    xy = rng.uniform(-1.2, 1.2, size=(1, F, 1, 2)) + rng.uniform(-0.25, 0.25, size=(1, F, 3, 2))
    z = rng.choice([0.5, 1.0, 2.0, 3.0], size=(1, F, 1, 1)) * np.ones((1, F, 3, 1))      # many equal depths -> ties
    f = np.concatenate([xy, z], -1).astype(np.float32)
    f[0, :50] = f[0, 50:100]                                                             # exact duplicates
    rfi, rw, rd = rr.nmr_forward(f, 128, 0.1, 100.0)
    _, fi, w, d = _hip_forward(torch.from_numpy(f), 128, 0.1, 100.0)
    assert (fi.cpu().numpy() == rfi).all()
    assert (d.cpu().numpy() == rd).all()


def test_texture_sampling_and_backwards():
    L = pkg("_lib")
    IS = 128
    fx, _ = _faces_of_room(3, 600, IS)
    fd, fi, w, d = _hip_forward(fx, IS, 0.001, 100.0)
    B, F = fd.shape[:2]
    rng = np.random.default_rng(1)
    tex = rng.uniform(0, 1, size=(B, F, 2, 2, 2, 3)).astype(np.float32)
    rgb = torch.empty(B, IS, IS, 3, device="cuda")
    L.check(L.lib().sln_raster_texture_sample(L.ptr(fd), L.ptr(torch.from_numpy(tex).cuda()), L.ptr(fi), L.ptr(w), L.ptr(d), B, F,
                                              IS, 2, 1e-3, L.ptr(rgb), L.current_stream_ptr()), "tex")
    rfi, rw, rd = rr.nmr_forward(fx.numpy(), IS, 0.001, 100.0)
    rrgb = rr.nmr_texture_sample(fx.numpy(), tex, rfi, rw, rd)
    assert_close(rgb.cpu().numpy(), rrgb, "rgb", rtol=1e-6, atol=1e-7)
    # depth backward
    gd = rng.standard_normal((B, IS, IS)).astype(np.float32)
    g = torch.zeros(B, F, 3, 3, device="cuda")
    L.check(L.lib().sln_raster_backward_depth(L.ptr(fd), L.ptr(fi), L.ptr(w), L.ptr(d), L.ptr(torch.from_numpy(gd).cuda()), B, F, IS,
                                              L.ptr(g), L.current_stream_ptr()), "bd")
    rg = rr.nmr_backward_depth(fx.numpy(), rfi, rw, rd, gd)
    assert_close(g.cpu().numpy(), rg, "grad depth", rtol=1e-4, atol=1e-4 * np.abs(rg).max())
    # rgb pixel-map backward
    grgb = rng.standard_normal((B, IS, IS, 3)).astype(np.float32)
    g2 = torch.zeros(B, F, 3, 3, device="cuda")
    L.check(L.lib().sln_raster_backward_rgb(L.ptr(fd), L.ptr(fi), L.ptr(rgb), L.ptr(torch.from_numpy(grgb).cuda()), B, F, IS, 3, 1e-3,
                                            L.ptr(g2), L.current_stream_ptr()), "br")
    rg2 = rr.nmr_backward_pixel_map(fx.numpy(), rfi, rrgb, grgb)
    assert np.abs(rg2).max() > 0
    assert_close(g2.cpu().numpy(), rg2, "grad rgb", rtol=1e-4, atol=1e-4 * np.abs(rg2).max())


def test_renderer_class_matches_restated_renderer():
    NR = pkg("host.neural_renderer")
    V, F, ranges, box = rr.synth_room(4, n_objects=6, target_faces=400)
    K, R, t = rr.get_cam_mat(torch.from_numpy(box))
    # keep only faces entirely in front of the camera (the reference culls z < 0.06 before rendering,
    # diff_render.py:346-356; x/z of a vertex at the camera plane is numerically meaningless)
    zc = (torch.from_numpy(V) @ R[0].T + t[0])[:, 2].numpy()
    F = F[(zc[F] > 0.3).all(1)]
    ref = rr.RefRenderer(image_size=96, K=K, R=R, t=t, orig_size=512, near=0.001)
    hip = NR.Renderer(camera_mode='projection', image_size=96, K=K.cuda(), R=R.cuda(), t=t.cuda(), anti_aliasing=False,
                      orig_size=512, near=0.001, light_intensity_ambient=1.0, light_intensity_directional=0.0)
    f = torch.from_numpy(F)[None]
    tex = torch.zeros(1, F.shape[0], 2, 2, 2, 3); tex[:, ::2] = 1.0
    for mode in ("depth", "rgb"):
        v1 = torch.from_numpy(V)[None].requires_grad_(True)
        v2 = torch.from_numpy(V)[None].cuda().requires_grad_(True)
        o1 = ref(v1, f, tex, mode=mode)
        o2 = hip(v2, f.cuda(), tex.cuda(), mode=mode)
        assert_close(o2.detach().cpu().numpy(), o1.detach().numpy(), mode, rtol=1e-5, atol=1e-6)
        gen = torch.Generator().manual_seed(0)
        go = torch.randn(o1.shape, generator=gen)
        (o1 * go).sum().backward(); (o2 * go.cuda()).sum().backward()
        assert_close(v2.grad.cpu().numpy(), v1.grad.numpy(), mode + " dV", rtol=1e-4, atol=1e-4 * v1.grad.abs().max().item())


def test_renderer_reuses_the_rasterisation_of_unchanged_geometry():
    """mesh_render_func renders the same vertices 33 times with different textures (diff_render.py:366,381-398): the Renderer
    rasterises once per (near, far) while the caller passes the same unmodified tensor objects, re-rasterises after an in-place
    update or for a new tensor, and every result equals the one of a fresh Renderer."""
    NR = pkg("host.neural_renderer")
    V, F, ranges, box = rr.synth_room(4, n_objects=6, target_faces=400)
    K, R, t = [x.cuda() for x in rr.get_cam_mat(torch.from_numpy(box))]
    zc = (torch.from_numpy(V).cuda() @ R[0].T + t[0])[:, 2].cpu().numpy()
    F = F[(zc[F] > 0.3).all(1)]
    kw = dict(camera_mode='projection', image_size=96, K=K, R=R, t=t, anti_aliasing=False, orig_size=512, near=0.001,
              light_intensity_ambient=1.0, light_intensity_directional=0.0)
    f = torch.from_numpy(F)[None].cuda()
    texs = []
    for k in range(3):
        tex = torch.zeros(1, F.shape[0], 2, 2, 2, 3, device="cuda"); tex[:, k::3] = 1.0
        texs.append(tex)
    calls = []
    orig = NR._rasterize
    NR._rasterize = lambda *a: (calls.append(a[2:]), orig(*a))[1]
    try:
        v = torch.from_numpy(V)[None].cuda().requires_grad_(True)
        hip = NR.Renderer(**kw)
        outs = [hip(v, f, texs[0], mode='depth')] + [hip(v, f, tex, mode='rgb') for tex in texs]
        assert len(calls) == 2, calls                           # one depth-pass rasterisation (near 0.1), one for the class passes
        w = [torch.randn(o.shape, generator=torch.Generator().manual_seed(i)).cuda() for i, o in enumerate(outs)]
        sum((o * wi).sum() for o, wi in zip(outs, w)).backward()
        n0 = len(calls)
        v2 = torch.from_numpy(V)[None].cuda().requires_grad_(True)
        fresh = [NR.Renderer(**kw)(v2, f, texs[0], mode='depth')] + [NR.Renderer(**kw)(v2, f, tex, mode='rgb') for tex in texs]
        assert len(calls) == n0 + 4
        for a, b in zip(outs, fresh):
            assert torch.equal(a, b)
        sum((o * wi).sum() for o, wi in zip(fresh, w)).backward()
        assert_close(v.grad.cpu().numpy(), v2.grad.cpu().numpy(), "dV shared vs fresh", rtol=1e-5, atol=1e-6 * float(v2.grad.abs().max()))
        # in-place update of the vertices: the version counter changes, the maps are rebuilt
        n1 = len(calls)
        with torch.no_grad():
            v[:, :, 0] += 0.05
        moved = hip(v, f, texs[1], mode='rgb')
        assert len(calls) == n1 + 1
        v3 = v.detach().clone()
        assert torch.equal(moved, NR.Renderer(**kw)(v3, f, texs[1], mode='rgb'))
        assert not torch.equal(moved, outs[2])
        # an equal tensor that is another object: not trusted
        n2 = len(calls)
        hip(v3, f, texs[1], mode='rgb')
        assert len(calls) == n2 + 1
    finally:
        NR._rasterize = orig


def test_geometry_kept_under_no_grad_is_not_reused_when_gradients_are_wanted():
    """A pass rendered under torch.no_grad() (a target image) and then, on the SAME vertex tensor, one that is back-propagated:
    the projection kept for the first call carries no autograd graph and must not be handed to the second."""
    NR = pkg("host.neural_renderer")
    V, F, ranges, box = rr.synth_room(4, n_objects=6, target_faces=400)
    K, R, t = [x.cuda() for x in rr.get_cam_mat(torch.from_numpy(box))]
    zc = (torch.from_numpy(V).cuda() @ R[0].T + t[0])[:, 2].cpu().numpy()
    F = F[(zc[F] > 0.3).all(1)]
    hip = NR.Renderer(camera_mode='projection', image_size=64, K=K, R=R, t=t, anti_aliasing=False, orig_size=512, near=0.001,
                      light_intensity_ambient=1.0, light_intensity_directional=0.0)
    f = torch.from_numpy(F)[None].cuda()
    tex = torch.zeros(1, F.shape[0], 2, 2, 2, 3, device="cuda"); tex[:, ::2] = 1.0
    v = torch.from_numpy(V)[None].cuda().requires_grad_(True)
    with torch.no_grad():
        a = hip(v, f, tex, mode='rgb')
    b = hip(v, f, tex, mode='rgb')
    assert torch.equal(a, b) and b.requires_grad
    (b * torch.linspace(0, 1, b.numel(), device="cuda").view_as(b)).sum().backward()
    assert v.grad is not None and float(v.grad.abs().max()) > 0


def test_shared_geometry_survives_separate_backward_sweeps_and_a_render_after_backward():
    """One cached geometry (same tensor objects), passes back-propagated in SEPARATE sweeps without retain_graph
    (loss_depth.backward(); loss_rgb.backward()), then another pass on the same Renderer and tensors and a third sweep: the
    upstream package projects per call and allows all of this; the gradients must equal those of the plain
    one-rasterisation-per-call path accumulated the same way."""
    NR = pkg("host.neural_renderer"); syn = pkg("host.synthetic"); DR = pkg("host.diff_render")
    V, F, ranges, box = syn.synthetic_room(3, n_objects=6, target_faces=500)
    K, R, t = DR.get_cam_mat(torch.from_numpy(box), "cuda")
    f = torch.from_numpy(F)[None].cuda()
    tex_a = (torch.rand(1, f.shape[1], 2, 2, 2, 3, device="cuda") > 0.5).float()
    tex_b = (torch.rand(1, f.shape[1], 2, 2, 2, 3, device="cuda") > 0.5).float()
    gd, ga, gb = torch.randn(1, 64, 64, device="cuda"), torch.randn(1, 3, 64, 64, device="cuda"), torch.randn(1, 3, 64, 64, device="cuda")
    grads = {}
    keep = NR.Renderer.reuse_rasterisation
    try:
        for reuse in (True, False):
            NR.Renderer.reuse_rasterisation = reuse
            v = torch.from_numpy(V)[None].cuda().requires_grad_(True)
            r = NR.Renderer(camera_mode='projection', image_size=64, K=K, R=R, t=t, anti_aliasing=False, orig_size=DR.inter_out, near=0.001,
                            light_intensity_ambient=1.0, light_intensity_directional=0.0)
            depth = r(v, f, tex_a, mode='depth')
            rgb = r(v, f, tex_a, mode='rgb')
            (depth * gd).sum().backward()                 # first sweep: the depth pass alone
            (rgb * ga).sum().backward()                   # second sweep through the same projection / gate nodes
            rgb2 = r(v, f, tex_b, mode='rgb')             # same Renderer, same tensors, after two backward sweeps
            (rgb2 * gb).sum().backward()
            grads[reuse] = v.grad.detach().cpu().numpy().copy()
            assert np.isfinite(grads[reuse]).all() and np.abs(grads[reuse]).max() > 0
    finally:
        NR.Renderer.reuse_rasterisation = keep
    assert_close(grads[True], grads[False], "vertex gradients over three sweeps", rtol=1e-4, atol=1e-6 * np.abs(grads[False]).max())


@pytest.mark.parametrize("image_size,n_pass,dense", [(96, 5, True), (100, 3, False), (64, 70, False)])
def test_shared_geometry_passes_equal_the_plain_one_rasterisation_per_call_path(image_size, n_pass, dense):
    """The Renderer's fast path (one projection node, one launch per rgb pass, ONE deferred pixel-map backward over all passes:
    sln_raster_texture_sample_chw / sln_raster_backward_rgb_multi) against its plain path (reuse_rasterisation = False: every call
    projects, rasterises, concatenates the fill_back textures and back-propagates on its own): images bit-identical, dV to 1e-5.
    dense: random trilinear textures (every pass non-zero everywhere: all mask bits set); otherwise 0/1 class masks; 70 passes
    take two launches of the 64-pass kernel; 100 is not a power of two (row flip by division)."""
    NR = pkg("host.neural_renderer")
    V, F, ranges, box = rr.synth_room(4, n_objects=6, target_faces=400)
    K, R, t = [x.cuda() for x in rr.get_cam_mat(torch.from_numpy(box))]
    zc = (torch.from_numpy(V).cuda() @ R[0].T + t[0])[:, 2].cpu().numpy()
    F = F[(zc[F] > 0.3).all(1)]
    kw = dict(camera_mode='projection', image_size=image_size, K=K, R=R, t=t, anti_aliasing=False, orig_size=512, near=0.001,
              light_intensity_ambient=0.7 if dense else 1.0, light_intensity_directional=0.0)
    f = torch.from_numpy(F)[None].cuda()
    g = torch.Generator().manual_seed(3)
    texs = []
    for k in range(n_pass):
        if dense:
            tex = torch.rand(1, F.shape[0], 2, 2, 2, 3, generator=g).cuda()
        else:
            tex = torch.zeros(1, F.shape[0], 2, 2, 2, 3, device="cuda"); tex[:, k::max(n_pass // 2, 2)] = 1.0
        texs.append(tex)
    res = {}
    keep = NR.Renderer.reuse_rasterisation
    try:
        for reuse in (False, True):
            NR.Renderer.reuse_rasterisation = reuse
            v = torch.from_numpy(V)[None].cuda().requires_grad_(True)
            hip = NR.Renderer(**kw)
            outs = [hip(v, f, texs[0], mode='depth')] + [hip(v, f, tex, mode='rgb') for tex in texs]
            w = [torch.randn(o.shape, generator=torch.Generator().manual_seed(i)).cuda() for i, o in enumerate(outs)]
            sum((o * wi).sum() for o, wi in zip(outs, w)).backward()
            res[reuse] = ([o.detach().clone() for o in outs], v.grad.clone())
            if reuse:                                        # a second backward sweep over part of the passes (retain_graph)
                v2 = torch.from_numpy(V)[None].cuda().requires_grad_(True)
                o2 = [hip(v2, f, tex, mode='rgb') for tex in texs[:2]]
                (o2[0] * w[1]).sum().backward(retain_graph=True)
                g_first = v2.grad.clone()
                (o2[1] * w[2]).sum().backward()
                res["two_sweeps"] = (g_first, v2.grad.clone())
    finally:
        NR.Renderer.reuse_rasterisation = keep
    for a, b in zip(res[False][0], res[True][0]):
        assert a.shape == b.shape
        if dense:           # ambient light: the plain path scales the texels before it samples, the fast one the sampled pixel
            assert_close(b.cpu().numpy(), a.cpu().numpy(), "image", rtol=1e-6, atol=1e-6)
        else:
            assert torch.equal(a, b)
    gp, gs = res[False][1].cpu().numpy(), res[True][1].cpu().numpy()
    assert_close(gs, gp, "dV shared vs plain", rtol=1e-5, atol=2e-6 * float(np.abs(gp).max()))
    assert np.abs(gp).max() > 0
    # separate sweeps: each flushes what it recorded, the sum is the gradient of both passes
    g1, g12 = res["two_sweeps"]
    assert float(g1.abs().max()) > 0 and float((g12 - g1).abs().max()) > 0


@pytest.mark.parametrize("image_size,target", [(96, 500), (256, 2000), (50, 200)])
def test_fused_scene_matches_33_pass_restatement(image_size, target):
    DR = pkg("host.diff_render")
    V, F, ranges, box = rr.synth_room(7, n_objects=12 if target > 1000 else 5, target_faces=target)
    v1 = torch.from_numpy(V)[None].requires_grad_(True)
    ref = rr.scene_render(v1, torch.from_numpy(F)[None], ranges, torch.from_numpy(box), image_size=image_size)
    v2 = torch.from_numpy(V)[None].cuda().requires_grad_(True)
    out = DR.scene_render(v2, torch.from_numpy(F)[None].cuda(), ranges, torch.from_numpy(box), image_size=image_size)
    assert out.shape == (1, 70, image_size, image_size)
    o, r = out.detach().cpu().numpy(), ref.detach().numpy()
    # the camera projection runs in torch on the GPU here and on the CPU in the oracle: vertex coordinates can
    # differ in the last bit, so allow a handful of silhouette pixels to flip (identical inputs are held to
    # bit-exactness in test_forward_bit_exact / test_random_soup_bit_exact_and_ties)
    assert (np.abs(o[0, 1:41] - r[0, 1:41]) > 1e-6).sum() <= 8, "class images differ"
    worst = np.abs(o[0] - r[0]).reshape(70, -1).max(1)
    assert_close(o, r, "final (worst channels %s)" % str([(int(c), float(worst[c])) for c in np.argsort(-worst)[:4]]),
                 rtol=1e-4, atol=1e-5)
    gen = torch.Generator().manual_seed(1)
    go = torch.randn(ref.shape, generator=gen)
    (ref * go).sum().backward()
    (out * go.cuda()).sum().backward()
    assert_close(v2.grad.cpu().numpy(), v1.grad.numpy(), "dV", rtol=1e-4, atol=2e-4 * v1.grad.abs().max().item())
    if image_size <= 96:
        v3 = torch.from_numpy(V)[None].cuda().requires_grad_(True)
        out3 = DR.scene_render_passes(v3, torch.from_numpy(F)[None].cuda(), ranges, torch.from_numpy(box), image_size=image_size)
        assert_close(out3.detach().cpu().numpy(), r, "passes", rtol=1e-5, atol=1e-5)
        (out3 * go.cuda()).sum().backward()
        assert_close(v3.grad.cpu().numpy(), v1.grad.numpy(), "dV passes", rtol=1e-4, atol=2e-4 * v1.grad.abs().max().item())


@pytest.mark.parametrize("n_rooms", [3, 8, 9, 17])
def test_batched_rooms_with_padding_equal_single_room_renders(n_rooms):
    """Ragged rooms (different V / F) are batched by padding with degenerate faces of class -1: the padded batch must
    reproduce each room's own render and vertex gradients.  8 rooms and more: the image -> XCD block mappings of the tile kernel,
    the depth walk and the pixel-map backward (one-dimensional grids padded to a multiple of 8 images: exactly 8, a remainder of 1
    on one and on two rounds of images) against the single-room launches, which use the plain grids and the small-batch split."""
    DR = pkg("host.diff_render")
    spec = ((11, 4, 300), (12, 7, 600), (13, 3, 200), (14, 5, 450), (15, 6, 500), (16, 2, 150), (17, 8, 700), (18, 4, 350), (19, 5, 250))
    rooms = [rr.synth_room(spec[i % 9][0] + 100 * (i // 9), n_objects=spec[i % 9][1], target_faces=spec[i % 9][2]) for i in range(n_rooms)]
    IS = 96
    singles, grads = [], []
    go = torch.randn(n_rooms, 70, IS, IS, generator=torch.Generator().manual_seed(2)).cuda()
    for i, (V, F, ranges, box) in enumerate(rooms):
        v = torch.from_numpy(V)[None].cuda().requires_grad_(True)
        out = DR.scene_render(v, torch.from_numpy(F)[None].cuda(), ranges, torch.from_numpy(box), image_size=IS)
        (out * go[i:i + 1]).sum().backward()
        singles.append(out.detach()); grads.append(v.grad[0])
    Vmax = max(r[0].shape[0] for r in rooms)
    prepared, Fmax = [], 0
    for V, F, ranges, box in rooms:
        K, R, t = DR.get_cam_mat(torch.from_numpy(box), "cpu")
        faces, cls, classes, chan, dch = DR.cull_and_classify(torch.from_numpy(V)[None], torch.from_numpy(F)[None], ranges, R, t)
        faces = torch.cat((faces, faces[:, :, [2, 1, 0]]), 1)[0]; cls = torch.cat((cls, cls))
        vp = torch.zeros(Vmax, 3); vp[:V.shape[0]] = torch.from_numpy(V)
        prepared.append((vp, faces, cls, K[0], R[0], t[0])); Fmax = max(Fmax, faces.shape[0])
    Vb = torch.stack([p[0] for p in prepared]).cuda().requires_grad_(True)
    Fb = torch.zeros(n_rooms, Fmax, 3, dtype=torch.int32); Cb = torch.full((n_rooms, Fmax), -1, dtype=torch.int32)
    for i, p in enumerate(prepared):
        Fb[i, :p[1].shape[0]] = p[1]; Cb[i, :p[2].shape[0]] = p[2]
    out = DR.scene_render_batch(Vb, Fb.cuda(), Cb.cuda(), torch.tensor(chan, dtype=torch.int32).cuda(),
                                torch.tensor(dch, dtype=torch.int32).cuda(), torch.stack([p[3] for p in prepared]).cuda(),
                                torch.stack([p[4] for p in prepared]).cuda(), torch.stack([p[5] for p in prepared]).cuda(), IS, 0.001)
    (out * go).sum().backward()
    for i, (V, F, ranges, box) in enumerate(rooms):
        assert_close(out[i:i + 1].detach().cpu().numpy(), singles[i].cpu().numpy(), "room %d" % i, rtol=1e-6, atol=1e-6)
        assert_close(Vb.grad[i, :V.shape[0]].cpu().numpy(), grads[i].cpu().numpy(), "room %d dV" % i, rtol=1e-4,
                     atol=1e-4 * grads[i].abs().max().item())
    assert torch.isfinite(Vb.grad).all()


def test_fused_projection_and_gather_equal_the_torch_ops():
    """sln_project_faces(+_backward) against neural_renderer.projection + vertices_to_faces evaluated by torch on the same device"""
    N = pkg("host.neural_renderer"); DR = pkg("host.diff_render")
    g = torch.Generator().manual_seed(0)
    B, V, F = 3, 200, 500
    verts = (torch.rand(B, V, 3, generator=g) * torch.tensor([4.0, 2.7, 4.0])).cuda()
    faces = torch.randint(0, V, (B, F, 3), generator=g, dtype=torch.int32).cuda()
    K, R, t = DR.get_cam_mat(torch.tensor([0, 0, 0, 4.0, 2.7, 5.0]), "cuda")
    K, R, t = K.expand(B, 3, 3), R.expand(B, 3, 3), t.expand(B, 1, 3)
    v1 = verts.clone().requires_grad_(True); v2 = verts.clone().requires_grad_(True)
    ref = N.vertices_to_faces(N.projection(v1, K, R, t, None, 512), faces)
    got = N.project_faces(v2, faces, K, R, t, 512)
    assert_close(got.detach().cpu().numpy(), ref.detach().cpu().numpy(), "faces_xyz", rtol=1e-5, atol=1e-5)
    w = torch.randn(ref.shape, generator=g).cuda()
    (ref * w).sum().backward(); (got * w).sum().backward()
    assert_close(v2.grad.cpu().numpy(), v1.grad.cpu().numpy(), "d/d vertices", rtol=1e-4, atol=1e-4 * float(v1.grad.abs().max()))


def test_renderer_batch_of_nine_equals_single_image_calls():
    """nr.Renderer with a batch of 9 images (depth and rgb, forward and backward): every image must come out as in a call of
    its own - the dense pixel-map backward's image -> XCD mapping (batches of 8 and more) against its small-batch path."""
    NR = pkg("host.neural_renderer")
    V, F, ranges, box = rr.synth_room(4, n_objects=6, target_faces=400)
    K, R, t = [x.cuda() for x in rr.get_cam_mat(torch.from_numpy(box))]
    zc = (torch.from_numpy(V).cuda() @ R[0].T + t[0])[:, 2].cpu().numpy()
    F = F[(zc[F] > 0.3).all(1)]
    B = 9
    rng = np.random.default_rng(0)
    Vb = torch.from_numpy(np.stack([V + rng.normal(0, 0.03, V.shape).astype(np.float32) * (k > 0) for k in range(B)])).cuda()
    fb = torch.from_numpy(F)[None].cuda().expand(B, -1, -1).contiguous()
    tex = torch.zeros(B, F.shape[0], 2, 2, 2, 3, device="cuda")
    for k in range(B):
        tex[k, k % 3::3] = 1.0
    kw = dict(camera_mode='projection', image_size=96, anti_aliasing=False, orig_size=512, near=0.001, light_intensity_ambient=1.0,
              light_intensity_directional=0.0)
    for mode in ("depth", "rgb"):
        vb = Vb.clone().requires_grad_(True)
        out = NR.Renderer(K=K.expand(B, -1, -1).contiguous(), R=R.expand(B, -1, -1).contiguous(), t=t.expand(B, -1, -1).contiguous(), **kw)(
            vb, fb, tex, mode=mode)
        go = torch.randn(out.shape, generator=torch.Generator().manual_seed(1)).cuda()
        (out * go).sum().backward()
        for k in (0, 3, 8):
            v1 = Vb[k:k + 1].clone().requires_grad_(True)
            o1 = NR.Renderer(K=K, R=R, t=t, **kw)(v1, fb[k:k + 1], tex[k:k + 1], mode=mode)
            assert torch.equal(out[k:k + 1], o1), (mode, k)
            (o1 * go[k:k + 1]).sum().backward()
            assert_close(vb.grad[k].cpu().numpy(), v1.grad[0].cpu().numpy(), "%s image %d dV" % (mode, k), rtol=1e-4,
                         atol=1e-5 * float(v1.grad.abs().max()))
            assert float(v1.grad.abs().max()) > 0


def test_many_triangles_bit_exact():
    """20 000 triangles (40 000 after fill_back: ~160 chunks of 256 per tile) at 128 x 128: face index, weights and depth
    bit-identical to the CPU restatement."""
    fx, _ = _faces_of_room(21, 20000, 128)
    assert fx.shape[1] >= 30000
    rfi, rw, rd = rr.nmr_forward(fx.numpy(), 128, 0.001, 100.0)
    _, fi, w, d = _hip_forward(fx, 128, 0.001, 100.0)
    assert (fi.cpu().numpy() == rfi).all(), "face index map differs at %d pixels" % int((fi.cpu().numpy() != rfi).sum())
    assert (w.cpu().numpy() == rw).all() and (d.cpu().numpy() == rd).all()
    assert len(np.unique(rfi)) > 500


def test_c3_sixteen_rooms_at_size():
    """BASELINE configs[2] at its own size - 16 rooms x 2000 triangles (x2 fill_back), 256 x 256, the bench's rooms: (a) face
    index / barycentric / depth maps of ALL 16 images bit-identical to the CPU restatement on identical projected faces,
    (b) the fused 70-channel scene tensor and d/d vertices of three of the 16 rooms (first, middle, last: every XCD slot of the
    image -> XCD mapping that batches of >= 8 use) against the 33-pass restatement."""
    DR = pkg("host.diff_render"); NR = pkg("host.neural_renderer"); syn = pkg("host.synthetic")
    rooms = [syn.synthetic_room(100 + i, n_objects=12, target_faces=2000) for i in range(16)]
    b = syn.pack_rooms(rooms)
    assert b["tris"] / 16 > 1800
    with torch.no_grad():
        fxyz = NR.project_faces(b["V"], b["F"], b["K"], b["R"], b["t"], 512).contiguous()
    rfi, rw, rd = rr.nmr_forward(fxyz.cpu().numpy(), 256, 0.001, 100.0)
    _, fi, w, d = _hip_forward(fxyz, 256, 0.001, 100.0)
    assert (fi.cpu().numpy() == rfi).all(), "face index map differs at %d pixels" % int((fi.cpu().numpy() != rfi).sum())
    assert (w.cpu().numpy() == rw).all() and (d.cpu().numpy() == rd).all()
    assert min(float((rfi[k] >= 0).mean()) for k in range(16)) > 0.9
    Vb = b["V"].clone().requires_grad_(True)
    out = DR.scene_render_batch(Vb, b["F"], b["C"], b["chan"], b["dch"], b["K"], b["R"], b["t"], 256, 0.001)
    go = torch.randn(16, 70, 256, 256, generator=torch.Generator().manual_seed(3))
    (out * go.cuda()).sum().backward()
    assert out.shape == (16, 70, 256, 256) and torch.isfinite(Vb.grad).all()
    for k in (0, 7, 15):
        V, F, ranges, box = rooms[k]
        v1 = torch.from_numpy(V)[None].requires_grad_(True)
        ref = rr.scene_render(v1, torch.from_numpy(F)[None], ranges, torch.from_numpy(box), image_size=256)
        (ref * go[k:k + 1]).sum().backward()
        o, r = out[k].detach().cpu().numpy(), ref[0].detach().numpy()
        # projection on the GPU vs on the CPU: last-bit differences of a vertex may flip a handful of silhouette pixels
        assert (np.abs(o[1:41] - r[1:41]) > 1e-6).sum() <= 8, "room %d: class images differ" % k
        assert_close(o, r, "room %d final" % k, rtol=1e-4, atol=1e-5)
        assert_close(Vb.grad[k, :V.shape[0]].cpu().numpy(), v1.grad[0].numpy(), "room %d dV" % k, rtol=1e-4,
                     atol=2e-4 * v1.grad.abs().max().item())


def test_scene_with_every_face_culled_renders_the_empty_image():
    """All faces behind the near plane (diff_render.py:346-356 drops them): the reference renders an empty image - background
    depth, no class pixels - and no gradient reaches the vertices; the fused pass must not refuse the empty face list."""
    DR = pkg("host.diff_render")
    V, F, ranges, box = rr.synth_room(4, n_objects=3, target_faces=120)
    Vb = torch.from_numpy(V)[None].cuda()
    Vb = (Vb + torch.tensor([0.0, 0.0, 50.0], device="cuda")).requires_grad_(True)      # the whole room far behind the camera
    out = DR.scene_render(Vb, torch.from_numpy(F)[None].cuda(), ranges, torch.from_numpy(box), image_size=64)
    assert out.shape == (1, 70, 64, 64) and torch.isfinite(out).all()
    assert float(out[0, 1:41].abs().max()) == 0.0
    out.sum().backward()
    assert float(Vb.grad.abs().max()) == 0.0


def test_class_list_longer_than_the_depth_hot_planes_is_refused():
    DR = pkg("host.diff_render")
    names = ["wall", "floor", "ceiling"] + ["bed"] * 1
    DR.class_tables(names)
    with pytest.raises(ValueError):
        others = [c.replace(" ", "_") for c in DR.nyu_class if c not in ("wall", "floor", "ceiling")]
        DR.class_tables(["wall", "floor", "ceiling"] + others[:30])


@pytest.mark.parametrize("n_rooms,target_faces,IS", [(3, 400, 128), (9, 1200, 96)])
def test_captured_scene_pass_replays_with_new_vertices_and_gradients(n_rooms, target_faces, IS):
    """SceneRenderGraph: forward + backward of the fused pass as one hipGraph for a fixed topology (the refinement loop of
    testing/test_render_refine.py:279-359 moves vertices only): every replay must equal the eager pass on the same inputs.
    The second case is a batch of more than 16 k (image, face) pairs: its backward forks the depth chain onto the library's side
    stream INSIDE the capture (event edges as graph dependencies)."""
    DR = pkg("host.diff_render"); syn = pkg("host.synthetic")
    rooms = [syn.synthetic_room(300 + i, n_objects=6, target_faces=target_faces) for i in range(n_rooms)]
    pk = syn.pack_rooms(rooms, "cuda")
    if n_rooms > 3:
        assert n_rooms * pk["F"].shape[1] * 2 >= 16384, "batch too small to take the side stream (fill_back doubles the faces)"
    args = (pk["F"], pk["C"], pk["chan"], pk["dch"], pk["K"], pk["R"], pk["t"], IS, 0.001)
    g = DR.SceneRenderGraph(pk["V"], *args)
    gen = torch.Generator().manual_seed(4)
    for trial in range(3):
        V = (pk["V"] + 0.01 * trial * torch.randn(pk["V"].shape, generator=gen).cuda()).detach()
        go = torch.randn(n_rooms, 70, IS, IS, generator=gen).cuda()
        image, dV = g(V, go)
        Ve = V.clone().requires_grad_(True)
        ref = DR.scene_render_batch(Ve, *args)
        ref.backward(go)
        assert_close(image.detach().cpu().numpy(), ref.detach().cpu().numpy(), "captured image %d" % trial, rtol=1e-6, atol=1e-6)
        assert_close(dV.detach().cpu().numpy(), Ve.grad.cpu().numpy(), "captured dV %d" % trial, rtol=1e-4, atol=1e-4 * float(Ve.grad.abs().max()))      # float atomics of the face backward: order differs between runs
    image2, dV2 = g()                                             # nothing new: the last inputs again
    assert torch.equal(image2, image) and torch.isfinite(dV2).all()


@pytest.mark.parametrize("n_rooms", [1, 9])
def test_deterministic_mode_makes_the_scene_pass_bit_identical(n_rooms):
    """sln_set_deterministic(1): the fused scene pass forward + backward gives the same bits on every run - the scene tensor
    (class depth statistics summed in exact fixed point) and d / d vertices (one stream, no split units, the depth walk behind the
    edge scans, fixed-order gradient sums, a gather instead of the vertex scatter) - for one room (the refinement loop's shape, where
    the default mode splits the long walks over several wavefronts) and for a batch that uses the image -> XCD mapping.  The
    deterministic result agrees with the default mode's."""
    lib = pkg("_lib"); DR = pkg("host.diff_render"); syn = pkg("host.synthetic")
    rooms = [syn.synthetic_room(300 + i, n_objects=8, target_faces=700) for i in range(n_rooms)]
    b = syn.pack_rooms(rooms)
    go = torch.randn(n_rooms, 70, 128, 128, generator=torch.Generator().manual_seed(5)).cuda()

    def run():
        Vb = b["V"].clone().requires_grad_(True)
        out = DR.scene_render_batch(Vb, b["F"], b["C"], b["chan"], b["dch"], b["K"], b["R"], b["t"], 128, 0.001)
        (out * go).sum().backward()
        torch.cuda.synchronize()
        return out.detach().clone(), Vb.grad.clone()
    try:
        lib.check(lib.lib().sln_set_deterministic(1), "sln_set_deterministic")
        runs = [run() for _ in range(4)]
    finally:
        lib.lib().sln_set_deterministic(0)
    for o, g in runs[1:]:
        assert torch.equal(o, runs[0][0]), float((o - runs[0][0]).abs().max())
        assert torch.equal(g, runs[0][1]), float((g - runs[0][1]).abs().max())
    o, g = run()
    assert_close(o.cpu().numpy(), runs[0][0].cpu().numpy(), "scene tensor, default vs deterministic", rtol=1e-5, atol=1e-6)
    assert_close(g.cpu().numpy(), runs[0][1].cpu().numpy(), "dV, default vs deterministic", rtol=1e-4,
                 atol=2e-5 * float(runs[0][1].abs().max()))
