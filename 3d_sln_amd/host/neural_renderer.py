"""HIP-backed stand-in for the slice of the third-party ``neural_renderer`` package that the reference
uses (``import neural_renderer as nr``, models/misc.py:7):

    renderer = nr.Renderer(camera_mode='projection', image_size=256, K=K, R=R, t=t, anti_aliasing=False,
                           orig_size=512, near=0.001, light_intensity_ambient=1.0,
                           light_intensity_directional=0.0)              # models/diff_render.py:359-361
    depth  = renderer(vertices, faces, textures, mode='depth')           # :366   -> [B, is, is]
    images = renderer(vertices, faces, textures, mode="rgb")             # :398   -> [B, 3, is, is]

Camera projection, fill_back and the vertex->face gather are a handful of small torch ops (autograd
carries them); rasterisation, texture sampling and both backward passes are HIP kernels behind the
C ABI (include/sln_hip.h, csrc/raster.hip).  Semantics: oracle/raster_ref.cpp ("parity unpinned").
To drop it into the reference: ``sys.modules['neural_renderer'] = this module``.
"""
import ctypes as C
import weakref

import torch

from .. import _lib


def _ws(nbytes, device):
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


def projection(vertices, K, R, t, dist_coeffs=None, orig_size=512, eps=1e-9):
    """Pinhole projection to NDC with the reference README's patch applied (no distortion, README.md:12-18)."""
    v = torch.matmul(vertices, R.transpose(2, 1)) + t
    x, y, z = v[:, :, 0], v[:, :, 1], v[:, :, 2]
    h = torch.stack([x / (z + eps), y / (z + eps), torch.ones_like(z)], dim=-1)
    h = torch.matmul(h, K.transpose(1, 2))
    u, vv = h[:, :, 0], orig_size - h[:, :, 1]
    u = 2 * (u - orig_size / 2.) / orig_size
    vv = 2 * (vv - orig_size / 2.) / orig_size
    return torch.stack([u, vv, z], dim=-1)


def vertices_to_faces(vertices, faces):
    B, V = vertices.shape[:2]
    idx = faces.long() + (torch.arange(B, dtype=torch.int64, device=vertices.device) * V)[:, None, None]
    return vertices.reshape(B * V, 3)[idx]


class _ProjectFaces(torch.autograd.Function):
    """projection + vertices_to_faces in one HIP launch each way (sln_project_faces / _backward)."""

    @staticmethod
    def forward(ctx, vertices, faces, K, R, t, orig_size, eps):
        vertices = vertices.float().contiguous()
        B, V = vertices.shape[:2]
        F = faces.shape[1]
        faces = faces.to(torch.int32).contiguous()
        cam = [x.float().expand(B, *x.shape[1:]).contiguous() for x in (K, R, t)]
        out = torch.empty(B, F, 3, 3, device=vertices.device)
        _lib.check(_lib.lib().sln_project_faces(_lib.ptr(vertices), _lib.ptr(faces), _lib.ptr(cam[0]), _lib.ptr(cam[1]), _lib.ptr(cam[2]), B, V, F,
                                                float(orig_size), float(eps), _lib.ptr(out), _lib.current_stream_ptr()), "sln_project_faces")
        # Plain attributes, not save_for_backward: every pass of a cached geometry (Renderer._geometry) hangs off THIS node, and the
        # caller may back-propagate the passes in separate sweeps (loss_depth.backward(); loss_rgb.backward()) or render again
        # after a backward - autograd frees saved tensors after the first sweep ("Trying to backward through the graph a second
        # time"), it does not free attributes.  All of them are inputs or fresh detached copies: no reference cycle.
        ctx.kept = (vertices.detach(), faces, *cam)
        ctx.orig_size, ctx.eps = float(orig_size), float(eps)
        return out

    @staticmethod
    def backward(ctx, gout):
        vertices, faces, K, R, t = ctx.kept
        B, V = vertices.shape[:2]
        g = torch.empty_like(vertices)
        _lib.check(_lib.lib().sln_project_faces_backward(_lib.ptr(vertices), _lib.ptr(faces), _lib.ptr(K), _lib.ptr(R), _lib.ptr(t), B, V,
                                                         faces.shape[1], ctx.orig_size, ctx.eps, _lib.ptr(gout.contiguous()), _lib.ptr(g),
                                                         _lib.current_stream_ptr()), "sln_project_faces_backward")
        return g, None, None, None, None, None, None


def project_faces(vertices, faces, K, R, t, orig_size=512, eps=1e-9):
    """``vertices_to_faces(projection(vertices, K, R, t, None, orig_size), faces)`` -> [B,F,3,3] (x_ndc, y_ndc, z_cam)."""
    return _ProjectFaces.apply(vertices, faces, K, R, t, orig_size, eps)


def _rasterize(faces, image_size, near, far):
    """(face_index int32, weight, depth) maps of faces [B,F,3,3] (sln_raster_forward)"""
    L = _lib.lib()
    B, F = faces.shape[0], faces.shape[1]
    dev = faces.device
    fi = torch.empty(B, image_size, image_size, dtype=torch.int32, device=dev)
    w = torch.empty(B, image_size, image_size, 3, device=dev)
    d = torch.empty(B, image_size, image_size, device=dev)
    ws = _ws(L.sln_raster_workspace_bytes(B, F), dev)
    _lib.check(L.sln_raster_forward(_lib.ptr(faces), B, F, image_size, near, far, _lib.ptr(ws), _lib.ptr(fi), _lib.ptr(w),
                                    _lib.ptr(d), _lib.current_stream_ptr()), "sln_raster_forward")
    return fi, w, d


class _RasterizeDepth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, faces, image_size, near, far, maps=None):
        faces = faces.contiguous()
        fi, w, d = maps if maps is not None else _rasterize(faces, image_size, near, far)
        ctx.kept = (faces.detach(), fi, w, d)          # attributes: see _ProjectFaces (the node may be walked by several sweeps)
        ctx.image_size = image_size
        ctx.maps = (fi, w, d)
        return d.clone()

    @staticmethod
    def backward(ctx, gd):
        faces, fi, w, d = ctx.kept
        B, F = faces.shape[0], faces.shape[1]
        g = torch.zeros_like(faces)
        _lib.check(_lib.lib().sln_raster_backward_depth(_lib.ptr(faces), _lib.ptr(fi), _lib.ptr(w), _lib.ptr(d),
                                                        _lib.ptr(gd.contiguous()), B, F, ctx.image_size, _lib.ptr(g),
                                                        _lib.current_stream_ptr()), "sln_raster_backward_depth")
        return g, None, None, None, None


class _RasterizeRgb(torch.autograd.Function):
    @staticmethod
    def forward(ctx, faces, textures, image_size, near, far, eps, maps=None):
        """``maps``: (face_index, weight, depth) of an earlier rasterisation of the SAME faces / near / far (Renderer's memo)."""
        L = _lib.lib()
        faces, textures = faces.contiguous(), textures.contiguous().float()
        B, F = faces.shape[0], faces.shape[1]
        dev = faces.device
        rgb = torch.empty(B, image_size, image_size, 3, device=dev)
        st = _lib.current_stream_ptr()
        if maps is None:
            maps = _rasterize(faces, image_size, near, far)
        fi, w, d = maps
        _lib.check(L.sln_raster_texture_sample(_lib.ptr(faces), _lib.ptr(textures), _lib.ptr(fi), _lib.ptr(w), _lib.ptr(d), B, F,
                                               image_size, textures.shape[2], eps, _lib.ptr(rgb), st), "sln_raster_texture_sample")
        ctx.save_for_backward(faces, fi, rgb)
        ctx.image_size, ctx.eps = image_size, eps
        ctx.maps = (fi, w, d)
        return rgb.clone()

    @staticmethod
    def backward(ctx, grgb):
        faces, fi, rgb = ctx.saved_tensors
        B, F = faces.shape[0], faces.shape[1]
        g = torch.zeros_like(faces)
        _lib.check(_lib.lib().sln_raster_backward_rgb(_lib.ptr(faces), _lib.ptr(fi), _lib.ptr(rgb), _lib.ptr(grgb.contiguous()), B,
                                                      F, ctx.image_size, 3, ctx.eps, _lib.ptr(g), _lib.current_stream_ptr()),
                   "sln_raster_backward_rgb")
        return g, None, None, None, None, None, None     # textures do not require grad in the reference (diff_render.py:397)


class _Geometry:
    """What the Renderer keeps for the LAST geometry while the caller keeps passing the very same, unmodified tensor objects
    (mesh_render_func: 33 calls on one vertex buffer, diff_render.py:366,381-398): the projected faces (ONE autograd node for all
    passes), the rasterisations per (near, far), and the rgb passes whose pixel-map backward is still owed."""

    def __init__(self, key_objs, orig_size, fxyz):
        self.refs = [weakref.ref(o) for o in key_objs]
        self.versions = [o._version for o in key_objs]
        self.orig_size = orig_size
        self.grad_mode = bool(torch.is_grad_enabled() and fxyz.requires_grad)   # a projection recorded under no_grad carries no graph
        self.fxyz = fxyz                      # [B, F(+fill_back), 3, 3], requires grad when the vertices do
        self.gate = _Gate.apply(fxyz, self)   # what the passes consume: its backward runs once, behind all of them
        self.maps = {}
        self.pending = []                     # (rgb_chw, grad_chw) of rgb passes waiting for the shared pixel-map backward

    def matches(self, key_objs, orig_size):
        want_grad = bool(torch.is_grad_enabled() and key_objs[0].requires_grad)
        return self.orig_size == orig_size and self.grad_mode == want_grad and \
            all(r() is o and v == o._version for r, v, o in zip(self.refs, self.versions, key_objs))


class _Gate(torch.autograd.Function):
    """Identity on the projected faces.  Every pass of a geometry consumes the gate's output, so autograd runs the gate's backward
    exactly once per backward sweep, after all of them: that is where the deferred rgb passes are back-propagated together
    (sln_raster_backward_rgb_multi: one edge walk for all passes instead of one per pass)."""

    @staticmethod
    def forward(ctx, fxyz, geom):
        ctx.geom = weakref.ref(geom)
        ctx.set_materialize_grads(False)
        return fxyz.clone()

    @staticmethod
    def backward(ctx, g):
        geom = ctx.geom()
        pend = geom.pending if geom is not None else []
        if not pend:
            return g, None
        geom.pending = []
        faces = geom.fxyz.detach()
        B, F = faces.shape[0], faces.shape[1]
        out = torch.zeros_like(faces) if g is None else g.contiguous().clone()
        L = _lib.lib()
        st = _lib.current_stream_ptr()
        for (fi, image_size, eps), items in _group_pending(pend).items():
            for k in range(0, len(items), 64):
                chunk = items[k:k + 64]
                ptrs = torch.tensor([[r.data_ptr() for r, _ in chunk], [gr.data_ptr() for _, gr in chunk]], dtype=torch.int64).to(faces.device)
                mask = torch.empty(B * image_size * image_size, dtype=torch.int64, device=faces.device)
                _lib.check(L.sln_raster_backward_rgb_multi(_lib.ptr(faces), _lib.ptr(fi), C.c_void_p(ptrs[0].data_ptr()),
                                                           C.c_void_p(ptrs[1].data_ptr()), len(chunk), B, F, image_size, eps,
                                                           _lib.ptr(mask), _lib.ptr(out), st), "sln_raster_backward_rgb_multi")
        return out, None


def _group_pending(pend):
    groups = {}
    for fi, image_size, eps, rgb, grad in pend:
        groups.setdefault((fi, image_size, eps), []).append((rgb, grad))
    return groups


class _RgbPass(torch.autograd.Function):
    """mode="rgb" on cached maps: ONE launch forward (sampling + ambient factor + fill_back by index + the package's layout);
    backward only records (image, gradient) - the geometry's gate runs the pixel-map backward of all recorded passes at once."""

    @staticmethod
    def forward(ctx, gate, textures, geom, maps, image_size, eps, scale):
        faces = gate.detach()
        textures = textures.detach().contiguous().float()
        B, F = faces.shape[0], faces.shape[1]
        fi, w, d = maps
        out = torch.empty(B, 3, image_size, image_size, device=faces.device)
        _lib.check(_lib.lib().sln_raster_texture_sample_chw(_lib.ptr(faces), _lib.ptr(textures), _lib.ptr(fi), _lib.ptr(w), _lib.ptr(d), B, F,
                                                            textures.shape[1], image_size, textures.shape[2], eps, float(scale), _lib.ptr(out),
                                                            _lib.current_stream_ptr()), "sln_raster_texture_sample_chw")
        ctx.geom, ctx.fi, ctx.image_size, ctx.eps = geom, fi, image_size, eps
        ctx.image = out.detach()                       # (a detached alias: no cycle through the output's grad_fn; see _ProjectFaces)
        ctx.set_materialize_grads(False)
        return out

    @staticmethod
    def backward(ctx, gout):
        if gout is not None:
            ctx.geom.pending.append((ctx.fi, ctx.image_size, ctx.eps, ctx.image, gout.contiguous()))
        return None, None, None, None, None, None, None      # the gate adds the gradient of the projected faces


class Renderer:
    """Constructor / call signature of ``neural_renderer.Renderer`` restricted to what the reference passes."""

    reuse_rasterisation = True      # see _geometry(); False rasterises (and projects) on every call

    def __init__(self, image_size=256, anti_aliasing=True, background_color=(0, 0, 0), fill_back=True,
                 camera_mode='projection', K=None, R=None, t=None, dist_coeffs=None, orig_size=1024, perspective=True,
                 viewing_angle=30, camera_direction=(0, 0, 1), near=0.1, far=100, light_intensity_ambient=0.5,
                 light_intensity_directional=0.5, light_color_ambient=(1, 1, 1), light_color_directional=(1, 1, 1),
                 light_direction=(0, 1, 0)):
        if camera_mode != 'projection':
            raise NotImplementedError("only camera_mode='projection' is on the HIP path (the reference uses nothing else)")
        if anti_aliasing:
            raise NotImplementedError("anti_aliasing=True is not used by the reference (diff_render.py:360)")
        if tuple(background_color) != (0, 0, 0) or light_intensity_directional != 0.0:
            raise NotImplementedError("the reference renders with ambient light only on a black background")
        self.image_size, self.fill_back, self.K, self.R, self.t = image_size, fill_back, K, R, t
        self.orig_size, self.near, self.far = orig_size, near, far
        self.light_intensity_ambient, self.rasterizer_eps = light_intensity_ambient, 1e-3

    def __call__(self, vertices, faces, textures=None, mode=None, K=None, R=None, t=None, dist_coeffs=None, orig_size=None):
        return self.render(vertices, faces, textures, mode, K, R, t, orig_size)

    def render(self, vertices, faces, textures=None, mode=None, K=None, R=None, t=None, orig_size=None):
        K = self.K if K is None else K
        R = self.R if R is None else R
        t = self.t if t is None else t
        orig_size = self.orig_size if orig_size is None else orig_size
        if vertices.device.type != 'cuda':
            raise _lib.SlnError("the rasterizer runs on the MI355X only (no CPU fallback)")
        geom = self._geometry(vertices, faces, K, R, t, orig_size)
        if geom is None:
            return self._render_plain(vertices, faces, textures, mode, K, R, t, orig_size)
        if mode == 'depth':
            # the package's render_depth does not forward near/far: library defaults apply
            maps = self._maps(geom, 0.1, 100.0)
            d = _RasterizeDepth.apply(geom.gate, self.image_size, 0.1, 100.0, maps)
            return torch.flip(d, dims=[1])
        if mode in ('rgb', None):
            maps = self._maps(geom, float(self.near), float(self.far))
            return _RgbPass.apply(geom.gate, textures, geom, maps, self.image_size, self.rasterizer_eps, self.light_intensity_ambient)
        raise NotImplementedError("mode=%r (the reference uses 'depth' and 'rgb')" % (mode,))

    def _render_plain(self, vertices, faces, textures, mode, K, R, t, orig_size):
        """One rasterisation per call, every step of the package spelled out (reuse_rasterisation = False)."""
        f2 = torch.cat((faces, faces[:, :, [2, 1, 0]]), dim=1) if self.fill_back else faces
        fxyz = project_faces(vertices, f2, K, R, t, orig_size)
        if mode == 'depth':
            d = _RasterizeDepth.apply(fxyz, self.image_size, 0.1, 100.0, None)
            return torch.flip(d, dims=[1])
        if mode in ('rgb', None):
            if self.fill_back:
                textures = torch.cat((textures, textures.permute((0, 1, 4, 3, 2, 5))), dim=1)
            textures = textures * self.light_intensity_ambient          # ambient-only lighting, white light
            rgb = _RasterizeRgb.apply(fxyz, textures, self.image_size, float(self.near), float(self.far), self.rasterizer_eps, None)
            return torch.flip(rgb.permute(0, 3, 1, 2), dims=[2])
        raise NotImplementedError("mode=%r (the reference uses 'depth' and 'rgb')" % (mode,))

    # mesh_render_func renders the SAME geometry 33 times in a row (1 depth pass + 32 class masks that differ only in their
    # textures, diff_render.py:366,381-398).  While the caller keeps passing the very same tensor objects, unmodified (weak
    # references: a new tensor that merely reuses a freed address is a different object; version counters: in-place updates),
    # the Renderer keeps that geometry's projection - ONE autograd node that all passes share -, its rasterisations, and defers
    # the pixel-map backward of the rgb passes to the shared node (_Gate): a class pass is one launch forward and a list append
    # backward; all 32 are back-propagated by one edge walk.  Round 3 kept only the maps and still paid, per pass, a projection
    # each way, a copy of the texture tensor, three layout copies and a full pixel-map backward: 19.5 ms per room against 20.4
    # rasterising every time on this round's box (the driver's round-3 run read 28.0 against 20.3 - not reproduced here; either
    # way those per-pass costs are what is gone: 12.2 ms now, 9.9-12.4 of it the caller's own torch code, bench.py render_33pass).
    def _geometry(self, vertices, faces, K, R, t, orig_size):
        key_objs = (vertices, faces, K, R, t)
        if not self.reuse_rasterisation or not self.fill_back or not all(torch.is_tensor(o) for o in key_objs):
            return None                      # (fill_back=False - never used by the reference - takes the plain path)
        geom = getattr(self, "_geom", None)
        if geom is not None and geom.matches(key_objs, orig_size):
            return geom
        f2 = torch.cat((faces, faces[:, :, [2, 1, 0]]), dim=1) if self.fill_back else faces
        geom = _Geometry(key_objs, orig_size, project_faces(vertices, f2, K, R, t, orig_size))
        self._geom = geom
        return geom

    def _maps(self, geom, near, far):
        key = (near, far, self.image_size, torch.cuda.current_stream().cuda_stream)
        if key not in geom.maps:
            geom.maps[key] = _rasterize(geom.fxyz.detach().contiguous(), self.image_size, near, far)
        return geom.maps[key]
