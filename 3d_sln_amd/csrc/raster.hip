// Tiled z-buffer rasterizer + its two hand-designed backward passes for gfx950.
//
// Semantics: the `neural_renderer` package the reference calls at models/diff_render.py:359-398
// (camera projection happens on the host side; this file starts at faces[B,F,3,3] = projected x,y in
// NDC + camera z).  The arithmetic (expression order, float/double mix) is the one restated in
// oracle/raster_ref.cpp and this file is built with -ffp-contract=off so that the face-index map is
// bit-identical to that restatement.
//
// Forward: a workgroup owns a 16x16 pixel tile (one pixel per lane, 4 wavefronts).  It streams the face
// set-up records in chunks of 256 (coalesced: one record field per lane), keeps the faces whose pixel
// bounding box touches the tile by an ORDER-PRESERVING ballot/prefix compaction into LDS (ascending
// face index == the package's tie rule "lowest index wins"), stages their 18 coefficients in LDS and
// lets every lane run the edge tests against broadcast LDS reads (a pixel that already holds something nearer than the
// face's depth lower bound skips the face).  z-min / index / barycentrics live in registers.  The package's 33 passes per scene (1 depth + 32 class masks over the same geometry)
// collapse into ONE pass that tracks two z-buffers (depth pass near=0.1, class passes near=ctor value).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>
#include <mutex>

#include "../../include/sln_hip.h"
#include "sln_common.h"
#include "sln_prof.h"

namespace {



constexpr int TS = 16;          // tile side
constexpr int CHUNK = 256;      // faces tested per round

struct FaceRec {                // per-face set-up, 24 floats
  float f[9];                   // x0 y0 z0 x1 y1 z1 x2 y2 z2
  float inv[9];                 // pixel-space inverse (w_k = inv[3k] xi + inv[3k+1] yi + inv[3k+2])
  int x0, x1, y0, y1;           // conservative pixel bounding box (x1 < x0: never drawn)
  int pad_[2];
};

// the bounding boxes once more as a compact array: every tile scans ALL faces of its image, and reading 16 bytes out of each
// 96-byte FaceRec touched twelve times the cache lines an 8-byte stream needs
struct BBox8 { unsigned short x0, x1, y0, y1; };

__device__ __forceinline__ bool backfacing(const float* f) {
  return (f[7] - f[1]) * (f[3] - f[0]) < (f[4] - f[1]) * (f[6] - f[0]);
}

__device__ __forceinline__ void face_inverse(const float* f, int is, float* inv) {
  float p[3][2];
#pragma unroll
  for (int n = 0; n < 3; ++n)
#pragma unroll
    for (int d = 0; d < 2; ++d) p[n][d] = 0.5f * (f[3 * n + d] * is + is - 1);      // == (float)(0.5 * (double)(...)): x0.5 is exact
  const float m[9] = {p[1][1] - p[2][1], p[2][0] - p[1][0], p[1][0] * p[2][1] - p[2][0] * p[1][1],
                      p[2][1] - p[0][1], p[0][0] - p[2][0], p[2][0] * p[0][1] - p[0][0] * p[2][1],
                      p[0][1] - p[1][1], p[1][0] - p[0][0], p[0][0] * p[1][1] - p[1][0] * p[0][1]};
  const float den = p[2][0] * (p[0][1] - p[1][1]) + p[0][0] * (p[1][1] - p[2][1]) + p[1][0] * (p[2][1] - p[0][1]);
#pragma unroll
  for (int k = 0; k < 9; ++k) inv[k] = m[k] / den;
}

__device__ __forceinline__ void raster_prep_face(const float* __restrict__ faces, const long i, const int is, FaceRec* __restrict__ rec,
                                                 BBox8* __restrict__ bbox) {
  FaceRec r;
#pragma unroll
  for (int k = 0; k < 9; ++k) r.f[k] = faces[9 * i + k];
  r.pad_[0] = r.pad_[1] = 0;
  bool draw = !backfacing(r.f);
#pragma unroll
  for (int k = 0; k < 9; ++k) draw = draw && (r.f[k] == r.f[k]);          // NaN vertices never draw
  if (draw) {
    face_inverse(r.f, is, r.inv);
    const float s = 0.5f * is, o = 0.5f * (is - 1);
    const float xa = fminf(r.f[0], fminf(r.f[3], r.f[6])) * s + o, xb = fmaxf(r.f[0], fmaxf(r.f[3], r.f[6])) * s + o;
    const float ya = fminf(r.f[1], fminf(r.f[4], r.f[7])) * s + o, yb = fmaxf(r.f[1], fmaxf(r.f[4], r.f[7])) * s + o;
    // clamp in float first (huge coordinates), pad by one pixel: the exact edge tests decide later
    r.x0 = (int)fmaxf(floorf(xa) - 1.f, 0.f); r.x1 = (int)fminf(ceilf(xb) + 1.f, (float)(is - 1));
    r.y0 = (int)fmaxf(floorf(ya) - 1.f, 0.f); r.y1 = (int)fminf(ceilf(yb) + 1.f, (float)(is - 1));
    if (!(xb >= -2.f && xa <= is + 1.f && yb >= -2.f && ya <= is + 1.f)) { r.x0 = 1; r.x1 = 0; r.y0 = 1; r.y1 = 0; }
  } else {
#pragma unroll
    for (int k = 0; k < 9; ++k) r.inv[k] = 0.f;
    r.x0 = 1; r.x1 = 0; r.y0 = 1; r.y1 = 0;
  }
  // a lower bound of every depth the face can produce (zp is a weighted harmonic mean of the three vertex depths when they are
  // all positive; the margin covers its rounding): the tile kernel skips the barycentric / depth arithmetic of a covered pixel
  // that already holds something nearer.  -inf (never skips) for faces that reach behind the camera.
  {
    const float zmin = fminf(r.f[2], fminf(r.f[5], r.f[8]));
    r.pad_[0] = __float_as_int(draw && zmin > 0.f ? zmin * (1.0f - 1e-5f) : -__builtin_huge_valf());
  }
  rec[i] = r;
  BBox8 bb; bb.x0 = (unsigned short)r.x0; bb.x1 = (unsigned short)r.x1; bb.y0 = (unsigned short)r.y0; bb.y1 = (unsigned short)r.y1;
  bbox[i] = bb;
}
__global__ void raster_prep_kernel(const float* __restrict__ faces, long n, int is, FaceRec* __restrict__ rec, BBox8* __restrict__ bbox) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) raster_prep_face(faces, i, is, rec, bbox);
}

struct ZState { float z; int idx; float w0, w1, w2; };

// DUAL: track a second z-buffer with its own near plane (the reference's depth pass runs with the package
// default near=0.1 while its class passes use the constructor's near, SURVEY.md 2.1 "known asymmetry").
// (statistics of the fused scene pass: defined here because the tile kernel's TEX form takes them along)
constexpr int DET_BANDS = 16;        // row bands of the deterministic masked sums (scene_bwd_masked_sums_det_kernel)
// vis[c]: some pixel of the class pass belongs to a face of class c - the predicate under which scene_compose writes the class's
// semantic plane and scene_bwd_maps reads its gradient; cnt[c] counts the pixels that also pass the 0.1 mask of diff_render.py:403
// (the same set while the class textures are all ones: the sample is 1).  The live flags of the semantic planes follow vis, those of
// the depth-hot planes cnt (an empty mask makes the plane the constant mean / wall_max without a gradient, :412-421).
struct SceneStats { double sum[64]; double cnt[64]; double gsum[64]; int wall_key; int wall_any; int det_ticket; int pad_; float det_part[DET_BANDS][64]; int vis[64]; };

// order-preserving float <-> int key (atomicMax on the key == float max, negatives included)
__device__ __forceinline__ int fkey(float v) { const int b = __float_as_int(v); return b >= 0 ? b : b ^ 0x7fffffff; }
__device__ __forceinline__ float funkey(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }
__device__ __forceinline__ float wall_max_of(const SceneStats& s) { return s.wall_any ? funkey(s.wall_key) : 10.0f; }

__device__ __forceinline__ float class_image_value(float v) { float s = 0.f; s += v; s += v; s += v; return s / 3.0f; }
__device__ __forceinline__ float depth_value(float d) { return d > 15.f ? -1.f : d; }

__device__ __forceinline__ void tex_sample(const float* f, const float* tx, int ts, float eps, float w0, float w1, float w2,
                                           float depth, float* px);
// TEX (fused scene pass, round 5): the class pass's texture sample of the pixel and the per-class statistics in the same launch
// (they were two launches between the tile kernel and the compose kernel: one pixel per thread, inputs = what this kernel has
// just written)
template <bool DUAL, bool TEX = false>
__global__ __launch_bounds__(256) void raster_tile_kernel(const FaceRec* __restrict__ rec, const BBox8* __restrict__ bbox, int F, int is, float near_a,
                                                          float near_b, float far, int32_t* __restrict__ fi_a,
                                                          float* __restrict__ w_a, float* __restrict__ d_a,
                                                          int32_t* __restrict__ fi_b, float* __restrict__ w_b,
                                                          float* __restrict__ d_b, const float* __restrict__ tex_faces,
                                                          const float* __restrict__ tex, int tex_ts, float tex_eps, float* __restrict__ tex_rgb,
                                                          const int32_t* __restrict__ st_cls, int st_nc, SceneStats* __restrict__ st_out,
                                                          int xcd_images) {
  __shared__ float sf[CHUNK][19];          // 9 vertex words, 9 inverse words, the face's depth lower bound
  __shared__ int sid[CHUNK];
  __shared__ int wave_cnt[4];
  const int tiles_x = (is + TS - 1) / TS;
  // Block -> (tile, image).  Workgroups are dealt to the 8 XCDs round-robin by linear id and dispatched in block order.  With at
  // least 8 images (xcd_images = B, a one-dimensional grid of 8 * ceil(B / 8) * tiles blocks, raster_tile_grid): XCD x rasterises
  // the images x, x + 8, ... only - their face records and bounding boxes (330 KB per 4 000-face image) stay in ONE 4 MB L2 instead
  // of all images' in all eight - and tile t of its images side by side.  Round 6, 16 rooms of 4 000 faces: 98.5 -> 74.0 us against
  // the (tiles, images) grid, where every XCD walked every image's face list.  Fewer images: grid (tiles, images), xcd_images = 0.
  int tile, b;
  if (xcd_images > 0) {
    const unsigned lin = blockIdx.x, xcd = lin & 7u, j = lin >> 3, nimg = (unsigned)(xcd_images + 7) >> 3;
    tile = (int)(j / nimg);
    b = (int)(xcd + 8u * (j - (unsigned)tile * nimg));
    if (b >= xcd_images) return;
  } else { tile = blockIdx.x; b = blockIdx.y; }
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int xi = tx * TS + (tid & (TS - 1)), yi = ty * TS + (tid >> 4);
  const bool inimg = xi < is && yi < is;
  const float yp = (float)((2. * yi + 1 - is) / is), xp = (float)((2. * xi + 1 - is) / is);
  const float fxi = (float)xi, fyi = (float)yi;
  const int tx0 = tx * TS, tx1 = tx0 + TS - 1, ty0 = ty * TS, ty1 = ty0 + TS - 1;
  // NDC coordinates of the tile's corner pixel centres (same formula as xp / yp)
  const float cx_lo = (float)((2. * tx0 + 1 - is) / is), cx_hi = (float)((2. * min(tx1, is - 1) + 1 - is) / is);
  const float cy_lo = (float)((2. * ty0 + 1 - is) / is), cy_hi = (float)((2. * min(ty1, is - 1) + 1 - is) / is);
  const FaceRec* rb = rec + (size_t)b * F;
  const BBox8* bb = bbox + (size_t)b * F;

  ZState A = {far, -1, 0.f, 0.f, 0.f}, Bz = {far, -1, 0.f, 0.f, 0.f};

  for (int c0 = 0; c0 < F; c0 += CHUNK) {
    const int fn = c0 + tid;
    bool hit = false;
    if (fn < F) {
      // one 8-byte load, pinned: hipcc loaded x0 alone (global_load_ushort), waited, and fetched the other three fields behind the
      // first comparison of the short-circuit test below - two dependent round trips per chunk of 256 faces for one
      uint2 q = *reinterpret_cast<const uint2*>(bb + fn);
      asm volatile("" : "+v"(q.x), "+v"(q.y));
      const int bx0 = (int)(q.x & 0xffffu), bx1 = (int)(q.x >> 16), by0 = (int)(q.y & 0xffffu), by1 = (int)(q.y >> 16);
      hit = bx0 <= tx1 && bx1 >= tx0 && by0 <= ty1 && by1 >= ty0;
      if (hit) {
        // the bounding box of a triangle is twice its area: drop the face when the whole tile lies outside one of its
        // edges (the edge function is linear, so its maximum over the tile is at a corner).  The margin keeps the cull
        // conservative against the rounding of the per-pixel test below, which stays the only exact decision.
        const float* f = rb[fn].f;
        const float ax[3] = {f[0], f[3], f[6]}, ay[3] = {f[1], f[4], f[7]};
#pragma unroll
        for (int e3 = 0; e3 < 3; ++e3) {
          const float x0 = ax[e3], y0 = ay[e3], dx = ax[(e3 + 1) % 3] - x0, dy = ay[(e3 + 1) % 3] - y0;
          const float e00 = (cy_lo - y0) * dx - (cx_lo - x0) * dy, e01 = (cy_lo - y0) * dx - (cx_hi - x0) * dy;
          const float e10 = (cy_hi - y0) * dx - (cx_lo - x0) * dy, e11 = (cy_hi - y0) * dx - (cx_hi - x0) * dy;
          if (fmaxf(fmaxf(e00, e01), fmaxf(e10, e11)) < -1e-4f) hit = false;
        }
      }
    }
    const unsigned long long m = __ballot(hit);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int off = 0, total = 0;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) { if (wv < wave) off += wave_cnt[wv]; total += wave_cnt[wv]; }
    if (hit) sid[off + before] = fn;
    __syncthreads();
    // stage the kept faces (18 coefficients + the depth bound each), coalesced over (face, field)
    for (int e = tid; e < total * 19; e += 256) {
      const int k = e / 19, q = e % 19;
      const FaceRec& r = rb[sid[k]];
      sf[k][q] = q < 9 ? r.f[q] : (q < 18 ? r.inv[q - 9] : __int_as_float(r.pad_[0]));
    }
    __syncthreads();
    if (inimg) {
      for (int k = 0; k < total; ++k) {
        const float* f = sf[k];
        if (f[18] >= (DUAL ? fmaxf(A.z, Bz.z) : A.z)) continue;      // cannot beat what the pixel holds (strict < decides below)
        if (((yp - f[1]) * (f[3] - f[0]) < (xp - f[0]) * (f[4] - f[1])) ||
            ((yp - f[4]) * (f[6] - f[3]) < (xp - f[3]) * (f[7] - f[4])) ||
            ((yp - f[7]) * (f[0] - f[6]) < (xp - f[6]) * (f[1] - f[7]))) continue;
        float w0 = f[9] * fxi + f[10] * fyi + f[11];
        float w1 = f[12] * fxi + f[13] * fyi + f[14];
        float w2 = f[15] * fxi + f[16] * fyi + f[17];
        w0 = fminf(fmaxf(w0, 0.f), 1.f); w1 = fminf(fmaxf(w1, 0.f), 1.f); w2 = fminf(fmaxf(w2, 0.f), 1.f);
        float ws = 0.f; ws += w0; ws += w1; ws += w2;
        w0 /= ws; w1 /= ws; w2 /= ws;
        // the package computes 1. / x in double and rounds to float; with a 53-bit intermediate that double rounding is
        // innocuous for a division (53 >= 2*24+2), so the correctly rounded fp32 division gives the same bits
        const float zp = 1.0f / (w0 / f[2] + w1 / f[5] + w2 / f[8]);
        if (far <= zp) continue;
        if (!(zp <= near_a) && zp < A.z) { A.z = zp; A.idx = sid[k]; A.w0 = w0; A.w1 = w1; A.w2 = w2; }
        if (DUAL && !(zp <= near_b) && zp < Bz.z) { Bz.z = zp; Bz.idx = sid[k]; Bz.w0 = w0; Bz.w1 = w1; Bz.w2 = w2; }
      }
    }
    __syncthreads();
  }
  float tpx[3] = {0.f, 0.f, 0.f};
  if (inimg) {
    const size_t p = ((size_t)b * is + yi) * is + xi;
    fi_a[p] = A.idx; d_a[p] = A.idx >= 0 ? A.z : far;
    w_a[3 * p] = A.w0; w_a[3 * p + 1] = A.w1; w_a[3 * p + 2] = A.w2;
    if (DUAL) {
      fi_b[p] = Bz.idx; d_b[p] = Bz.idx >= 0 ? Bz.z : far;
      w_b[3 * p] = Bz.w0; w_b[3 * p + 1] = Bz.w1; w_b[3 * p + 2] = Bz.w2;
      if (TEX) {                                       // texture_sample_kernel's work for the pixel, from the registers that hold its inputs
        if (Bz.idx >= 0)
          tex_sample(tex_faces + 9 * ((size_t)b * F + Bz.idx), tex + ((size_t)b * F + Bz.idx) * tex_ts * tex_ts * tex_ts * 3, tex_ts, tex_eps,
                     Bz.w0, Bz.w1, Bz.w2, Bz.z, tpx);
        tex_rgb[3 * p] = tpx[0]; tex_rgb[3 * p + 1] = tpx[1]; tex_rgb[3 * p + 2] = tpx[2];
      }
    }
  }
  if (TEX && st_out != nullptr) {
    // ... and scene_stats_kernel's (per-class depth sums in exact fixed point, visible-face marks, the wall's maximum depth): the
    // tile's sums in LDS, one set of device atomics per tile.  Exact integer / power-of-two arithmetic: the statistics are the
    // same bits as the separate pass gave, whatever the order of the tiles.
    __shared__ unsigned long long ssum[64]; __shared__ int scnt[64]; __shared__ int svis[64];
    __shared__ int s_wkey, s_wany;
    if (tid < 64) { ssum[tid] = 0ull; scnt[tid] = 0; svis[tid] = 0; }
    if (tid == 0) { s_wkey = (int)0x80000000; s_wany = 0; }
    __syncthreads();
    if (inimg && Bz.idx >= 0) {
      const_cast<FaceRec*>(rb)[Bz.idx].pad_[1] = 1;
      const int c = st_cls[(long)b * F + Bz.idx];
      if (c >= 0 && c < st_nc) svis[c] = 1;
      if (c >= 0 && c < st_nc && class_image_value(tpx[0]) > 0.1f) {
        const float dd = depth_value(A.idx >= 0 ? A.z : far);
        atomicAdd(&ssum[c], (unsigned long long)(long long)rint((double)dd * 4294967296.0)); atomicAdd(&scnt[c], 1);
        if (c == 0) { s_wany = 1; atomicMax(&s_wkey, fkey(dd)); }
      }
    }
    __syncthreads();
    if (tid < st_nc && scnt[tid] > 0) {
      atomicAdd(&st_out[b].sum[tid], (double)(long long)ssum[tid] * (1.0 / 4294967296.0));
      atomicAdd(&st_out[b].cnt[tid], (double)scnt[tid]);
    }
    if (tid < st_nc && svis[tid]) st_out[b].vis[tid] = 1;
    if (tid == 0 && s_wany) {
      atomicMax(&st_out[b].wall_any, 1);
      atomicMax(&st_out[b].wall_key, s_wkey);
    }
  }
}

// trilinear sample of the winning face's ts^3 texture cube (package's forward_texture_sampling)
__device__ __forceinline__ void tex_sample(const float* f, const float* tx, int ts, float eps, float w0, float w1, float w2,
                                           float depth, float* px) {
  const float wk[3] = {w0, w1, w2};
  float t[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float v = wk[k] * (ts - 1) * (depth / f[3 * k + 2]);
    v = fmaxf(v, 0.f);
    v = fminf(v, (float)(ts - 1) - eps);
    t[k] = v;
  }
  px[0] = px[1] = px[2] = 0.f;
  for (int c = 0; c < 8; ++c) {
    float w = 1.f; int ti[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float fr = t[k] - (float)(int)t[k];
      if (((c >> k) & 1) == 0) { w *= 1.f - fr; ti[k] = (int)t[k]; }
      else { w *= fr; ti[k] = (int)t[k] + 1; }
    }
    const int cell = ti[0] * ts * ts + ti[1] * ts + ti[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) px[k] += w * tx[3 * cell + k];
  }
}

__global__ void texture_sample_kernel(const float* __restrict__ faces, const float* __restrict__ textures,
                                      const int32_t* __restrict__ fi, const float* __restrict__ w,
                                      const float* __restrict__ depth, int F, int is, int ts, float eps, long npix,
                                      float* __restrict__ rgb) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  const int fn = fi[i];
  float px[3] = {0.f, 0.f, 0.f};
  if (fn >= 0) {
    const long b = i / ((long)is * is);
    const float* f = faces + 9 * (b * F + fn);
    tex_sample(f, textures + (size_t)(b * F + fn) * ts * ts * ts * 3, ts, eps, w[3 * i], w[3 * i + 1], w[3 * i + 2], depth[i], px);
  }
  rgb[3 * i] = px[0]; rgb[3 * i + 1] = px[1]; rgb[3 * i + 2] = px[2];
}

// The same sampling for the Renderer's rgb mode in one launch (round 4): output [B,3,is,is] with rows flipped (what
// neural_renderer returns after permute + flip: three torch copies before), ambient light as a factor, and fill_back resolved by
// index: textures hold F_tex = F / 2 cubes, face f >= F_tex reads cube f - F_tex with its first and third texture axes swapped
// (the package concatenates textures.permute((0, 1, 4, 3, 2, 5)) - a copy of the whole texture tensor per pass).
__global__ void texture_sample_chw_kernel(const float* __restrict__ faces, const float* __restrict__ textures,
                                          const int32_t* __restrict__ fi, const float* __restrict__ w,
                                          const float* __restrict__ depth, int F, int F_tex, int is, int ts, float eps, float scale,
                                          long npix, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  const int fn = fi[i];
  const long plane = (long)is * is;
  const long b = i / plane; const int p = (int)(i - b * plane);
  float px[3] = {0.f, 0.f, 0.f};
  if (fn >= 0) {
    const float* f = faces + 9 * (b * F + fn);
    const bool back = fn >= F_tex;
    const float* tx = textures + (size_t)(b * F_tex + (back ? fn - F_tex : fn)) * ts * ts * ts * 3;
    if (!back) {
      tex_sample(f, tx, ts, eps, w[3 * i], w[3 * i + 1], w[3 * i + 2], depth[i], px);
    } else {                      // swapped axes: cell (t0, t1, t2) of the permuted cube is cell (t2, t1, t0) of the stored one
      const float wk[3] = {w[3 * i], w[3 * i + 1], w[3 * i + 2]};
      float t[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float v = wk[k] * (ts - 1) * (depth[i] / f[3 * k + 2]);
        v = fmaxf(v, 0.f); v = fminf(v, (float)(ts - 1) - eps);
        t[k] = v;
      }
      for (int c = 0; c < 8; ++c) {
        float ww = 1.f; int ti[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float fr = t[k] - (float)(int)t[k];
          if (((c >> k) & 1) == 0) { ww *= 1.f - fr; ti[k] = (int)t[k]; }
          else { ww *= fr; ti[k] = (int)t[k] + 1; }
        }
        const int cell = ti[2] * ts * ts + ti[1] * ts + ti[0];
#pragma unroll
        for (int k = 0; k < 3; ++k) px[k] += ww * tx[3 * cell + k];
      }
    }
  }
  const int y = p / is, x = p - y * is;
  float* o = out + b * 3 * plane + (long)(is - 1 - y) * is + x;
  o[0] = px[0] * scale; o[plane] = px[1] * scale; o[2 * plane] = px[2] * scale;
}

// depth backward, atomic-free: one wavefront per face gathers the pixels it won inside its bounding box
// (a per-pixel scatter serialises on the 9 atomics of large wall / floor faces: 1.45 ms per 16 rooms).
// gridDim.y > 1 (few images: the launch lasts as long as the largest face's bounding box walk): the windows of 64 pixels
// are dealt to gridDim.y wavefronts, which then add their parts atomically.
__global__ __launch_bounds__(64) void depth_backward_face_kernel(const float* __restrict__ faces, const int32_t* __restrict__ fi,
                                                                 const float* __restrict__ w, const float* __restrict__ depth,
                                                                 const float* __restrict__ gd, int F, int is,
                                                                 float* __restrict__ gfaces, int xcd_images) {
  // Block order (round 6).  xcd_images = B >= 8 (grid.x = 8 * ceil(B / 8) * F, depth_bwd_grid_x): consecutive workgroups go to
  // consecutive XCDs, XCD x walks the images x, x + 8, ... only (their maps stay in one L2) and face f of its images side by side, so
  // that every image's large faces are dispatched at the same point of the launch instead of the last image's starting when the
  // others are done: 77.2 -> 59.7 us for 16 rooms.  xcd_images = 0: image-major, grid.x = B * F.  SLN_DEPTH_BWD_ORDER=0 (lab).
  const int lane = threadIdx.x;
  int b, fn;
  if (xcd_images > 0) {
    const unsigned lin = blockIdx.x, xcd = lin & 7u, j = lin >> 3, nimg = (unsigned)(xcd_images + 7) >> 3;
    fn = (int)(j / nimg);
    b = (int)(xcd + 8u * (j - (unsigned)fn * nimg));
    if (b >= xcd_images) return;
  } else { b = (int)(blockIdx.x / (unsigned)F); fn = (int)(blockIdx.x - (unsigned)b * (unsigned)F); }                            // 32-bit divisions
  const size_t i = (size_t)b * F + fn;

  float fl[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) fl[k] = faces[9 * i + k];
  bool draw = !backfacing(fl);
#pragma unroll
  for (int k = 0; k < 9; ++k) draw = draw && (fl[k] == fl[k]);
  if (!draw) return;
  const float s = 0.5f * is, o = 0.5f * (is - 1);
  const float xa = fminf(fl[0], fminf(fl[3], fl[6])) * s + o, xb = fmaxf(fl[0], fmaxf(fl[3], fl[6])) * s + o;
  const float ya = fminf(fl[1], fminf(fl[4], fl[7])) * s + o, yb = fmaxf(fl[1], fmaxf(fl[4], fl[7])) * s + o;
  if (!(xb >= -2.f && xa <= is + 1.f && yb >= -2.f && ya <= is + 1.f)) return;
  const int x0 = (int)fmaxf(floorf(xa) - 1.f, 0.f), x1 = (int)fminf(ceilf(xb) + 1.f, (float)(is - 1));
  const int y0 = (int)fmaxf(floorf(ya) - 1.f, 0.f), y1 = (int)fminf(ceilf(yb) + 1.f, (float)(is - 1));
  float iv[9];
  face_inverse(fl, is, iv);
  float tmp[2] = {0.f, 0.f};
#pragma unroll
  for (int l = 0; l < 2; ++l)
#pragma unroll
    for (int m = 0; m < 3; ++m) tmp[l] += -iv[3 * m + l] / fl[3 * m + 2];
  float acc[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const long base = (long)b * is * is;
  // the bounding box as one flat pixel range: narrow boxes keep all 64 lanes busy (a row per iteration used a third of them)
  const int bw = x1 - x0 + 1, npx = bw * (y1 - y0 + 1);
  if (64 * (int)blockIdx.y >= npx) return;
  for (int pidx = lane + 64 * (int)blockIdx.y; pidx < npx; pidx += 64 * (int)gridDim.y) {
    {
      const int y = y0 + pidx / bw, x = x0 + pidx % bw;
      const long q = base + (long)y * is + x;
      if (fi[q] != fn) continue;
      const float g = gd[q];
      const float d2 = depth[q] * depth[q];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float wk = w[3 * q + k];
        acc[3 * k + 2] += g * wk * d2 / (fl[3 * k + 2] * fl[3 * k + 2]);
#pragma unroll
        for (int l = 0; l < 2; ++l) acc[3 * k + l] += -g * tmp[l] * wk * d2 * is / 2;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    float v = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    acc[k] = v;
  }
  if (lane == 0) {
    // always atomic: in the fused scene pass this kernel runs on a side stream NEXT TO pixel_map_backward_kernel, which adds into
    // the same nine floats of the face (a plain read-modify-write here could swallow one of its atomic adds)
#pragma unroll
    for (int k = 0; k < 9; ++k)
      if (acc[k] != 0.f) atomicAdd(gfaces + 9 * i + k, acc[k]);
  }
}

// raster_tile_kernel's grid and its xcd_images argument (see the kernel): SLN_TILE_ORDER=0 (lab) keeps the (tiles, images) grid
inline bool raster_tile_xcd(int B) {
  static const bool off = [] { const char* e = std::getenv("SLN_TILE_ORDER"); return e != nullptr && e[0] == '0'; }();
  return !off && B >= 8;
}
inline dim3 raster_tile_grid(int tiles, int B) { return raster_tile_xcd(B) ? dim3((unsigned)((B + 7) / 8 * 8 * tiles)) : dim3(tiles, B); }
inline int raster_tile_arg(int B) { return raster_tile_xcd(B) ? B : 0; }
inline int depth_bwd_xcd(int B) {
  static const bool off = [] { const char* e = std::getenv("SLN_DEPTH_BWD_ORDER"); return e != nullptr && e[0] == '0'; }();
  return !off && B >= 8 ? B : 0;
}
inline unsigned depth_bwd_grid_x(int B, int F) { return (unsigned)((long)(depth_bwd_xcd(B) ? (B + 7) / 8 * 8 : B) * F); }
// Few (image, face) pairs leave the chip idle while the largest faces are walked: split their walks (see the two kernels).
inline int small_batch_split(long units, int max_split, long budget = 32768) {
  int s = 1;
  while (s < max_split && units * s * 2 <= budget) s *= 2;        // stay below `budget` workgroups
  return s;
}
// depth backward: one wavefront per (face, slice of its bounding-box walk).  A face's walk is a serial chain of dependent loads; a
// few big faces (the wall / floor quads of a refinement room fill the view) set the launch's duration whatever the face count, and
// a slice without pixels costs a wavefront that reads nine floats and leaves: split by default while the grid stays below
// `budget` wavefronts (16 refinement rooms, 22 k faces: 247 -> us with the split of 8 the face count alone denied)
// ... but only for images of FEW faces (<= 2 048: few faces over a fixed image area are big faces): the 16 x 4 000-face batch of
// BASELINE config c3 went 0.596 -> 0.643 ms per batch with the split everywhere, 16 refinement rooms (1 400 faces each) 247 -> 130 us
inline int depth_bwd_split(long units, int F) {
  static const int few = std::getenv("SLN_DEPTH_SPLIT_FACES") ? std::atoi(std::getenv("SLN_DEPTH_SPLIT_FACES")) : 2048;
  return F <= few ? small_batch_split(units, 8, 1L << 20) : small_batch_split(units, 8);
}
inline unsigned pixel_map_grid_x(int B, int F);
inline bool pixel_map_grid_ok(int B, int F) { return (long)(B >= 8 ? (B + 7) / 8 * 8 : B) * F * 6 < (1L << 31); }   // 32-bit workgroup ids in the kernel
inline int pixel_map_scan_split(long faces_total, int B) {
  if (B >= 8) return 1;                          // the image -> XCD affinity mapping of the kernel uses a 2-D grid
  return small_batch_split(faces_total * 6, 16, 131072);
}

// ----------------------------------------------------------------------------------------------------
// pixel-map backward.  One wavefront per (face, edge, axis): see pixel_map_backward_kernel for the two-phase walk.
// PIX is a policy giving diff(q, ref) = sum_c (I_c(q) - I_c(ref)) * dI_c(q) with the package's positive-part test.
// ----------------------------------------------------------------------------------------------------
// A policy exposes, per (scan axis, image), a view with pixel offsets LOCAL to the image (32-bit: the base pointers of the image
// are wavefront-uniform, so a load is "uniform base + 32-bit lane offset" - the 64-bit per-lane address arithmetic of the first
// version was a dozen of the ~130 instructions a scan window costs, and the kernel is bound by instruction issue, see below):
// idx(d0, d1), ref_pair(p, p', ..), load(p, ref) and eval(loaded, ref) = sum_c (I_c(p) - I_c(ref)) * dI_c(p) with the package's
// positive-part test.
struct PixDenseView {
  const int32_t* fi; const float* rgb; const float* grad; int C, is, axis;      // pointers at the image's first pixel
  __device__ __forceinline__ int idx(int d0, int d1) const { return axis == 0 ? d1 * is + d0 : d0 * is + d1; }
  // Ref: what a scan needs to know about its reference pixel, fetched once per edge step (phase 1)
  struct __align__(16) Ref { int p; int fi; int pad_[2]; };
  __device__ __forceinline__ void ref_pair(int pa, int pb, Ref& ra, Ref& rb) const {
    ra.p = pa; ra.fi = fi[pa]; ra.pad_[0] = ra.pad_[1] = 0;
    rb.p = pb; rb.fi = fi[pb]; rb.pad_[0] = rb.pad_[1] = 0;
  }
  struct Loaded { int fq; float diff; };
  __device__ __forceinline__ Loaded load(int p, const Ref& ref) const {
    Loaded r; r.fq = fi[p];
    float diff = 0.f;
    for (int k = 0; k < C; ++k) diff += (rgb[(long)p * C + k] - rgb[(long)ref.p * C + k]) * grad[(long)p * C + k];
    r.diff = diff;
    return r;
  }
  __device__ __forceinline__ float eval(const Loaded& L, const Ref&, int& fq) const { fq = L.fq; return L.diff > 0.f ? L.diff : 0.f; }
  __device__ __forceinline__ Loaded issue(int p, const Ref& ref) const { return load(p, ref); }
  static __device__ __forceinline__ void pin(Loaded&) {}
};
struct PixDense {            // C-channel image: one positive-part test over the channel sum (package's rgb mode, C=3)
  const int32_t* fi; const float* rgb; const float* grad; int C, is;
  __device__ __forceinline__ PixDenseView view(int axis, int b) const {
    const long plane = (long)is * is;
    return PixDenseView{fi + b * plane, rgb + b * plane * C, grad + b * plane * C, C, is, axis};
  }
};

// P rgb passes over the SAME geometry in ONE walk (round 4).  mesh_render_func calls nr.Renderer 32 times on identical vertices
// (models/diff_render.py:381-398) and back-propagates all of them at once: the Renderer defers the pixel-map backward of its
// rgb passes until autograd reaches the shared projection node and then runs this policy - the edge walk, the prefix sums and
// the scan windows are paid once instead of P times.  Semantics = the sum over the passes of the package's per-pass gradient
// (every pass keeps its own positive-part test).  Images are [B,3,is,is] with rows flipped, exactly what the Renderer returned
// and what autograd hands back.  mask[b][pixel] has bit p set when pass p is non-zero at the pixel in any channel: a pass whose
// image is zero at both the scanned and the reference pixel contributes exactly nothing (diff = 0), so only the set bits are
// visited - one or two of 32 for class masks, all of them for dense textures (still exact).
struct PixMultiView {
  const int32_t* fi; const float* const* rgb; const float* const* grad; const unsigned long long* mask;
  unsigned img_off, plane; int is, axis, shift;          // img_off: floats of image b inside a pass tensor; shift: log2(is) or -1
  __device__ __forceinline__ int idx(int d0, int d1) const { return axis == 0 ? d1 * is + d0 : d0 * is + d1; }
  __device__ __forceinline__ int flipped(int p) const {
    const int y = shift >= 0 ? p >> shift : p / is;
    return p + (is - 1 - 2 * y) * is;
  }
  struct __align__(16) Ref { int pf; int fi; unsigned long long m; };
  __device__ __forceinline__ void ref_pair(int pa, int pb, Ref& ra, Ref& rb) const {
    ra.pf = flipped(pa); ra.fi = fi[pa]; ra.m = mask[ra.pf];
    rb.pf = flipped(pb); rb.fi = fi[pb]; rb.m = mask[rb.pf];
  }
  struct Loaded { int fq; float tot; };
  __device__ __forceinline__ Loaded load(int p, const Ref& ref) const {
    Loaded r; r.fq = fi[p];
    const int pf = flipped(p);
    unsigned long long m = mask[pf] | ref.m;
    float tot = 0.f;
    while (m) {
      const int ps = __ffsll((long long)m) - 1;
      m &= m - 1;
      const float* im = rgb[ps] + img_off; const float* g = grad[ps] + img_off;
      float diff = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) diff += (im[k * plane + pf] - im[k * plane + ref.pf]) * g[k * plane + pf];
      if (diff > 0.f) tot += diff;
    }
    r.tot = tot;
    return r;
  }
  __device__ __forceinline__ float eval(const Loaded& L, const Ref&, int& fq) const { fq = L.fq; return L.tot; }
  __device__ __forceinline__ Loaded issue(int p, const Ref& ref) const { return load(p, ref); }
  static __device__ __forceinline__ void pin(Loaded&) {}
};
struct PixMulti {
  const int32_t* fi; const float* const* rgb; const float* const* grad; const unsigned long long* mask; int is, shift;
  __device__ __forceinline__ PixMultiView view(int axis, int b) const {
    const unsigned plane = (unsigned)is * (unsigned)is;
    return PixMultiView{fi + (long)b * plane, rgb, grad, mask + (long)b * plane, (unsigned)b * 3u * plane, plane, is, axis, shift};
  }
};
// mask[b][flipped pixel] of PixMulti: one thread per pixel, P <= 64 passes
__global__ void multi_mask_kernel(const float* const* __restrict__ rgb, int P, int plane, long n, unsigned long long* __restrict__ mask) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long b = i / plane; const long q = i - b * plane;
  unsigned long long m = 0;
  for (int ps = 0; ps < P; ++ps) {
    const float* im = rgb[ps] + b * 3 * plane + q;
    if (im[0] != 0.f || im[plane] != 0.f || im[2 * (long)plane] != 0.f) m |= 1ull << ps;
  }
  mask[i] = m;
}

// The reference's 32 class passes fused (models/diff_render.py:381-398): pass c renders value(pixel) where the
// winning face belongs to class c and 0 elsewhere, into three equal rgb channels whose mean is the class
// image; every pass applies its own positive-part test (at most two classes contribute per pixel pair).
// Vertical scans (axis 0) read TRANSPOSED copies of the per-pixel maps and of the class-gradient planes so that
// the 64 lanes of a scan touch consecutive addresses (the strided version fetched 5.8 GB per 16 rooms).
// One 16-byte record per pixel = everything a scan needs about that pixel: winning face, its class, the class-image value
// and the incoming gradient of the pixel's OWN class plane.  A scan step is then one 16-byte load per lane (four separate
// 4-byte streams before); only when the reference pixel's class differs is the gradient plane of that class read as well.
struct PixRec { int fi; int cp; float v; float gown; };
struct PixClassView {
  const char* rec; const char* g; int is; unsigned plane4;        // byte pointers at the image's records / first class plane
  __device__ __forceinline__ int idx(int d0, int d1) const { return d0 * is + d1; }
  // Ref: class and value of the reference pixel of an edge step (phase 1).  Knowing the class up front makes the address of
  // the second gradient plane independent of the scanned pixel's record, so both loads of a scan pixel go out together.
  struct __align__(16) Ref { int fi, cp; float v; int pad_; };
  __device__ __forceinline__ void ref_pair(int pa, int pb, Ref& ra, Ref& rb) const {      // both loads, then both pins (see load())
    int4 ia = *reinterpret_cast<const int4*>(rec + (unsigned)pa * 16u);
    int4 ib = *reinterpret_cast<const int4*>(rec + (unsigned)pb * 16u);
    asm volatile("" : "+v"(ia.x), "+v"(ia.y), "+v"(ia.z), "+v"(ia.w), "+v"(ib.x), "+v"(ib.y), "+v"(ib.z), "+v"(ib.w));
    ra.fi = ia.x; ra.cp = ia.y; ra.v = __int_as_float(ia.z); ra.pad_ = 0;
    rb.fi = ib.x; rb.cp = ib.y; rb.v = __int_as_float(ib.z); rb.pad_ = 0;
  }
  struct Loaded { int4 iq; float g_cr; };
  __device__ __forceinline__ Loaded load(int p, const Ref& ref) const {
    Loaded r;
    r.iq = *reinterpret_cast<const int4*>(rec + (unsigned)p * 16u);               // one 16-byte load ...
    // ... and it has to stay one: value and own-plane gradient (z, w) are only used when the class (y) is >= 0, so hipcc split
    // the load into two 8-byte halves and sank the second one behind the class test - a second, dependent round trip per
    // scan window (global_load_dwordx2, s_waitcnt vmcnt(1), branch, global_load_dwordx2 offset:8, s_waitcnt vmcnt(0) in the ISA)
    // The pin below makes the four words live at that point, i.e. it is also where the wavefront waits for them: the load of
    // the reference class's gradient plane (unconditional, independent of the record) is an operand too, so that it is issued
    // BEFORE the wait and both loads share one round trip.
    r.g_cr = *reinterpret_cast<const float*>(g + ((unsigned)max(ref.cp, 0) * plane4 + (unsigned)p * 4u));
    asm volatile("" : "+v"(r.iq.x), "+v"(r.iq.y), "+v"(r.iq.z), "+v"(r.iq.w), "+v"(r.g_cr));
    return r;
  }
  // the two loads without the pin, and the pin on its own: several windows' loads go out before the first pin waits (phase 2a)
  __device__ __forceinline__ Loaded issue(int p, const Ref& ref) const {
    Loaded r;
    r.iq = *reinterpret_cast<const int4*>(rec + (unsigned)p * 16u);
    r.g_cr = *reinterpret_cast<const float*>(g + ((unsigned)max(ref.cp, 0) * plane4 + (unsigned)p * 4u));
    return r;
  }
  static __device__ __forceinline__ void pin(Loaded& r) { asm volatile("" : "+v"(r.iq.x), "+v"(r.iq.y), "+v"(r.iq.z), "+v"(r.iq.w), "+v"(r.g_cr)); }
  __device__ __forceinline__ float eval(const Loaded& L, const Ref& ref, int& fq) const {
    // the package's per-pixel term, two additions onto zero, each only when positive:
    //   own class   (cq >= 0):             3 * ((vq - [cr == cq] * vr) * g[cq][q])
    //   ref's class (cr >= 0, cr != cq):   3 * ((0 - vr) * g[cr][q])
    // (3 * x for the package's ((0 + x) + x) + x over the three identical colour channels: x + x is exact, 2x + x rounds once, to the
    // nearest float of 3x).  Round 6, fewer vector instructions for the same bits - the launch is bound by the vector instructions it
    // issues (LAB_NOTES 9a-2): a record without a class holds v = 0 and g = 0 (scene_bwd_maps_body), so its own-class term is 3 * (0 * 0)
    // and needs no class test; "add when positive" onto zero is max(x, 0) (NaN gives 0 either way), and 0 + a, a + 0 are exact.
    const int cr = ref.cp, cq = L.iq.y;
    fq = L.iq.x;
    const float vr = ref.v;
    const bool same = cr == cq;
    const float da = fmaxf(3.0f * ((__int_as_float(L.iq.z) - (same ? vr : 0.f)) * __int_as_float(L.iq.w)), 0.f);
    float db = 0.f;
    if (cr >= 0) db = same ? 0.f : fmaxf(3.0f * ((0.f - vr) * L.g_cr), 0.f);
    return da + db;
  }
};
struct PixClass {
  const PixRec *rec, *recT; const float *g, *gT; int is, NC;
  __device__ __forceinline__ PixClassView view(int axis, int b) const {
    const long plane = (long)is * is;
    return PixClassView{reinterpret_cast<const char*>((axis == 0 ? recT : rec) + b * plane),
                        reinterpret_cast<const char*>((axis == 0 ? gT : g) + (long)b * NC * plane), is, (unsigned)(plane * 4)};
  }
};

// x * 2. / is of the package (double arithmetic, rounded to float): x*2 is exact and the quotient by an integer that is
// exact in fp32 survives the double rounding (see raster_tile_kernel), so fp32 gives the same bits; for a power-of-two
// image size the division is an exact scaling.
__device__ __forceinline__ float pix_scale(float x, int is, bool pow2, float s2) { return pow2 ? x * s2 : x * 2.0f / (float)is; }

// Work unit = (face, edge x axis, chunk of PMB_DC consecutive d0 values): blockIdx = (face, 6, chunks).  One wavefront per
// face serialised up to 3*2*image_size edge steps for a wall / floor triangle that spans the image while thousands of
// small faces had finished - the kernel ran as long as its largest face.  Units outside the edge's d0 range exit at
// once; each unit adds its two partial sums to the face gradient with two atomics.
constexpr int PMB_DC = 64;
#ifndef PMB_ILP
#define PMB_ILP 2            // windows of a row in flight per pass of phase 2a (same-box A/B of 1 / 2 / 3 / 4 / 8: 0.590 / 0.549 / 0.556 / 0.571 / 0.688 ms per 16-room batch)
#endif
#ifndef PMB_LONG
#define PMB_LONG 64          // rows of at least this many pixels are scanned row by row (phase 2a); 0: every row flattened (16 / 32 / 48 / 64: 0.586 / 0.571 / 0.548 / 0.546 ms with two windows per pass)
#endif
// diff / dist of a contributing scan pixel: v_rcp_f32 (1 ulp) and a multiplication instead of the ~12-instruction IEEE division
// sequence, twice per contributing pixel in a kernel bound by instruction issue (0.774 -> 0.750 ms per 16 rooms forward + backward).
// The face gradient is a sum of thousands of such terms whose order already differs from the restatement's (lane partial sums,
// atomics): the relative change is ~1e-7, three orders below the 1e-4 tolerance of the parity tests (which pass in both builds).
// -DPMB_EXACT_DIV restores the division.
#ifdef PMB_EXACT_DIV
#define PMB_DIV(a, b) ((a) / (b))
#else
#define PMB_DIV(a, b) ((a) * __builtin_amdgcn_rcpf(b))
#endif

// Inside a unit the serial form of the edge walk (for each d0: load the face index under the edge, then scan) is a
// chain of three dependent memory round trips per edge step.  Here the walk is flattened:
//   phase 1 - lane l owns edge step d0 = c_from + l: crossing, reference pixels, the face index under the edge and
//             the two scan ranges, all 64 steps in ONE round trip; parameters + an exclusive prefix of the scan
//             lengths go to LDS;
//   phase 2 - the lanes stride over the concatenated scan pixels of all steps (a short walk along the prefix), every
//             iteration is independent of the others, so their loads overlap.
// What bounds phase 2 (round 2, measured): NOT the memory round trip of a window - two or four windows per lane with all loads
// in flight together were 1-5 % SLOWER (more registers, fewer wavefronts), and removing a second dependent round trip per
// window that hipcc had created (see PixClassView::load) gained 0.6 % - but the ~130 instructions a wavefront issues per window
// of 64 scan pixels: 1.3 M windows x 130 x 4 cycles over 1 024 SIMDs = 0.28 ms.  Hence the packed 16-byte step records in LDS
// (one ds_read_b128 instead of four ds_read_b32), the 16-byte reference records (one read, selected by address), the ratios of
// a step read only by contributing pixels, and image-local 32-bit pixel offsets.
struct __align__(16) PmbStep { int lo, ofrom, ifrom; float cross; };
// orders a wavefront's own LDS writes before its own later reads: the workgroup is ONE wavefront and LDS executes a wavefront's
// instructions in order, so this is a compiler fence, not an s_barrier
#define PMB_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
// grid of the launch: B >= 8 (image -> XCD mapping in the kernel): one dimension, 8 * ceil(B / 8) * F * 6 workgroups; fewer images:
// (B * F, 6, scan split)
inline unsigned pixel_map_grid_x(int B, int F) { return (unsigned)((long)(B >= 8 ? (B + 7) / 8 * 8 * 6 : B) * F); }
inline unsigned pixel_map_grid_y(int B) { return B >= 8 ? 1u : 6u; }
#ifdef PMB_STAMP
// lab build only: per-workgroup clock sums (prologue, phase 1, phase 2a load waits / evaluation / passes / rows, phase 2b), one slot per
// workgroup, read back by sln_lab_pmb_stamps
constexpr int PMB_STAMP_SLOTS = 1 << 19;
__device__ unsigned long long g_pmb_stamp[PMB_STAMP_SLOTS][10];      // [8], [9]: wall clock (100 MHz) at the start and the end
#define PMB_T() ((unsigned long long)clock64())
#endif
template <typename PIX, bool POW2>
__global__ __launch_bounds__(64) void pixel_map_backward_kernel(const float* __restrict__ faces, const FaceRec* __restrict__ vis,
                                                                float* __restrict__ gfaces, int B, int F, int is, float eps, PIX pix) {
  // (argument order, round 6: the pointers and scalars in front arrive in SGPRs with the wavefront - kernarg preload stops at the first
  // by-value struct; with `pix` second it took five serialised scalar loads to reach the owner test below)
  typedef decltype(pix.view(0, 0)) View;
  __shared__ int s_pre[PMB_DC + 1];          // exclusive prefix of the scan lengths
  __shared__ PmbStep s_step[PMB_DC];
  __shared__ float2 s_ratio[PMB_DC];
  __shared__ typename View::Ref s_ref[PMB_DC][2];     // [0]: reference pixel of the outward scan of a step, [1]: of the inward scan
  static_assert(sizeof(typename View::Ref) == 16, "one ds_read_b128 per reference record");
  // Image -> XCD affinity.  Workgroups are dealt to the 8 XCDs round-robin by linear id and every XCD has its own L2: with the
  // plain (face, edge) order all XCDs scan the SAME image at a time and each L2 fetches that image's maps for itself (measured:
  // 1.07 GB from memory per launch for 0.28 GB of maps).  With at least 8 images, XCD x takes the images x, x+8, ...
  // The launcher (pixel_map_grid_x) pads the grid to a multiple of 8 images for this mapping: with B = 9 a grid of 9 F x 6
  // workgroups gives XCD 0 only 6.75 F of the 12 F units of its two images (found by the 9-room parity test).
  // (32-bit arithmetic: the first version did this mapping with 64-bit integers - four software divisions, ~600 scalar
  // instructions in front of every one of the 370 k workgroups of a 16-room batch, as many as all their scan windows issue)
  const int lane = threadIdx.x;
#ifdef PMB_STAMP
  const unsigned long long T_start = PMB_T(), W_start = (unsigned long long)wall_clock64();
  unsigned long long T_p1 = 0, T_wait = 0, T_eval = 0, T_2b = 0, N_pass = 0, N_rows = 0, T_pro = 0;
#endif
  unsigned bu, fnu; int ea;
  if (B >= 8) {
    // Order inside an XCD's share (round 6): its images one after the other, FACE-MAJOR - the six (edge, axis) units of a face side by
    // side, faces in list order.  Workgroups are dispatched in block order, a unit of an owner lasts 10 us in the median and up to 90
    // (a wall edge), an ownerless one ~3 us, so the order decides how full the chip is.  With all faces per (edge, axis), the order
    // until round 5, owners and ownerless faces alternate in every sixth of an image and the long wall edges of the LAST image's fifth
    // and sixth start at 75-80 % of the launch: 15 % of it ran below a fifth of the occupancy.  Face-major keeps an image's owners
    // densely in flight and lets the ownerless run of its second half (the mirrored faces of fill_back) drain at the dispatch rate:
    // 246 -> 222 us for the 16-room batch.  Measured and NOT kept (LAB_NOTES 9a-3): images in alternating pairs with the two halves of
    // the face list interleaved (238 us: ~3 us ownerless workgroups in every slot again), 2 / 3 / 6 units per workgroup (224 / 231 /
    // 314 us: the chip retires 4-wavefront workgroups as fast as 1-wavefront ones, but a workgroup keeps its LDS until its longest
    // unit is done), 2 / 4 / 8 faces per workgroup walked one after the other (226 / 232 / 240 us).
    const unsigned lin = blockIdx.x;                                // < 6 * 8 * ceil(B / 8) * F: the launcher refuses grids beyond 2^31
    const unsigned xcd = lin & 7u, j = lin >> 3;
    const unsigned per_img = (unsigned)F * 6u;
    const unsigned img_local = j / per_img, r = j - img_local * per_img;
    fnu = r / 6u;
    bu = xcd + 8u * img_local;
    if (bu >= (unsigned)B) return;
    ea = (int)(r - fnu * 6u);
  } else {
    // fewer images than XCDs: grid (B * F, 6, scan split), all faces per (edge, axis).  Face-major was measured here too (round 6) and
    // is WORSE for a single image without owner flags (the drop-in Renderer's rgb backward: 113 -> 165 us per pass): the launch lasts
    // as long as its longest walks, and in this order they start in every sixth of the launch instead of in one burst.
    bu = blockIdx.x / (unsigned)F; fnu = blockIdx.x - bu * (unsigned)F; ea = (int)blockIdx.y;
  }
  const int b = (int)bu, fn = (int)fnu;
  const size_t i = (size_t)bu * F + fnu;
  // A face that owns no pixel of the map contributes nothing: outward scans start from a pixel of the face under the edge, inward
  // scans count the face's own pixels only.  The fused scene pass marks the owners in its forward (FaceRec::pad_[1]); their
  // complement - every back-facing face, and the front-facing ones that are hidden or fall between pixel centres - leaves here
  // instead of walking its edge and scanning its interior for nothing.
  // The owner flag, the face and the edge's vertices (rotated by e, scan axis first; read at computed offsets: picking six of the nine
  // words of a face held in scalar registers by the run-time (e, axis) was a chain of ~90 scalar selects) are ALL requested before the
  // first of them is tested: one scalar-memory round trip in front of phase 1 instead of three.
  const int e = ea >> 1, axis = ea & 1;
  int pi[3];
#pragma unroll
  for (int n = 0; n < 3; ++n) pi[n] = e + n >= 3 ? e + n - 3 : e + n;
  const float* fp = faces + 9 * i;
  int own = 1;
  if (vis != nullptr) own = vis[i].pad_[1];
  float face[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) face[k] = fp[k];
  float praw[3][2];
#pragma unroll
  for (int n = 0; n < 3; ++n)
#pragma unroll
    for (int d = 0; d < 2; ++d) praw[n][d] = fp[3 * pi[n] + ((d + axis) & 1)];
  asm volatile("" : "+s"(own), "+s"(face[0]), "+s"(face[1]), "+s"(face[3]), "+s"(face[4]), "+s"(face[6]), "+s"(face[7]),
               "+s"(praw[0][0]), "+s"(praw[0][1]), "+s"(praw[1][0]), "+s"(praw[1][1]), "+s"(praw[2][0]), "+s"(praw[2][1]));
  if (own == 0) return;
  if (backfacing(face)) return;
  // (round 6) a template parameter: as a run-time flag the power-of-two test was two scalar branches per distance term - four per
  // window - around an IEEE division that the 256 x 256 images never execute
  constexpr bool pow2 = POW2;
  const float s2 = 2.0f / (float)is;
  const View V = pix.view(axis, b);
  float p[3][2];
#pragma unroll
  for (int n = 0; n < 3; ++n)
#pragma unroll
    for (int d = 0; d < 2; ++d) p[n][d] = 0.5f * (praw[n][d] * is + is - 1);
  const int dir = (axis == 0) ? (p[0][0] < p[1][0] ? -1 : 1) : (p[0][0] < p[1][0] ? 1 : -1);
  const int d0_from = (int)fmaxf(ceilf(fminf(p[0][0], p[1][0])), 0.f);
  const int d0_to = (int)fminf(fmaxf(p[0][0], p[1][0]), (float)(is - 1));
  float acc0 = 0.f, acc1 = 0.f;                 // gradient slots pi[0] / pi[1], component (1 - axis)
  // hoisted by hand: hipcc re-read gridDim.z from the dispatch packet in every window (s_load_dword + s_waitcnt lgkmcnt(0))
  const int wfirst = 64 * (int)blockIdx.z, wstep = 64 * (int)gridDim.z;
  // an edge longer than PMB_DC steps takes several rounds (chunks) in the same wavefront: one workgroup per chunk filled the
  // grid with empty workgroups (three out of four), whose dispatch alone cost ~0.25 ms per batch of 16 rooms
#ifdef PMB_STAMP
  T_pro = PMB_T() - T_start;
#endif
  for (int c_from = d0_from; c_from <= d0_to; c_from += PMB_DC) {
  const int c_to = min(d0_to, c_from + PMB_DC - 1);
  PMB_SYNC();                                   // the LDS tables of the previous round are no longer read
#ifdef PMB_STAMP
  const unsigned long long T_c0 = PMB_T();
#endif

  // ---- phase 1: one edge step per lane ----
  const int d0 = c_from + lane;
  PmbStep sp; sp.lo = 0; sp.ofrom = 0; sp.ifrom = 0; sp.cross = 0.f;
  int li = 0;
  float r0 = 0.f, r1 = 0.f;
  typename View::Ref rin_keep, rout_keep;        // the step's two reference records (phase 2a reads them with v_readlane)
  { int4 z4 = make_int4(0, 0, 0, 0); rin_keep = *reinterpret_cast<typename View::Ref*>(&z4); rout_keep = rin_keep; }
  if (d0 <= c_to) {
    const float d1_cross = (p[1][1] - p[0][1]) / (p[1][0] - p[0][0]) * (d0 - p[0][0]) + p[0][1];
    sp.cross = d1_cross;
    const int d1_in = dir > 0 ? (int)floorf(d1_cross) : (int)ceilf(d1_cross);
    const int d1_out = d1_in + dir;
    if (!(d1_in < 0 || is <= d1_in || d1_out < 0 || is <= d1_out)) {
      const bool use0 = p[1][0] != d0, use1 = p[0][0] != d0;
      r0 = use0 ? (p[1][0] - p[0][0]) / (p[1][0] - d0) : 0.f;         // 0 marks "slot not used" (a used ratio is never 0)
      r1 = use1 ? (p[1][0] - p[0][0]) / (d0 - p[0][0]) : 0.f;
      typename View::Ref rin, rout;
      V.ref_pair(V.idx(d0, d1_in), V.idx(d0, d1_out), rin, rout);
      s_ref[lane][0] = rin; s_ref[lane][1] = rout;
      rin_keep = rin; rout_keep = rout;
      if (rin.fi == fn) {                                              // outward scan to the image border
        const int lim = dir > 0 ? is - 1 : 0;
        const int from = max(min(d1_out, lim), 0), to = min(max(d1_out, lim), is - 1);
        sp.ofrom = from; sp.lo = max(to - from + 1, 0);
      }
      float cross2;                                                    // inward scan to the opposite edge
      if ((d0 - p[0][0]) * (d0 - p[2][0]) < 0) cross2 = (p[2][1] - p[0][1]) / (p[2][0] - p[0][0]) * (d0 - p[0][0]) + p[0][1];
      else cross2 = (p[1][1] - p[2][1]) / (p[1][0] - p[2][0]) * (d0 - p[2][0]) + p[2][1];
      const int lim = dir > 0 ? (int)ceilf(cross2) : (int)floorf(cross2);
      const int from = max(min(d1_in, lim), 0), to = min(max(d1_in, lim), is - 1);
      sp.ifrom = from; li = max(to - from + 1, 0);
    }
  }
  // Rows of at least PMB_LONG pixels (94 % of the scan pixels of a furnished room: outward scans run to the image border) are
  // scanned by the whole wavefront, one row at a time (phase 2a); only the shorter rows enter the flattened space of phase 2b.
  const int lo_all = sp.lo, li_all = li;
  const bool long_o = PMB_LONG > 0 && lo_all >= PMB_LONG, long_i = PMB_LONG > 0 && li_all >= PMB_LONG;
  if (long_o) sp.lo = 0;
  if (long_i) li = 0;
  int incl = sp.lo + li;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off, 64); if (lane >= off) incl += v; }
  s_pre[lane + 1] = incl;
  if (lane == 0) s_pre[0] = 0;
  s_step[lane] = sp;
  s_ratio[lane] = make_float2(r0, r1);
  PMB_SYNC();
  const int W = s_pre[64];
#ifdef PMB_STAMP
  T_p1 += PMB_T() - T_c0;
#endif

  // ---- phase 2a: one row per pass of the wavefront (round 6) ----
  // Everything that describes the row - its step's crossing, ratios, reference record, first pixel, length - is wavefront-uniform
  // (v_readlane of the owning lane's phase-1 registers): a lane adds its offset to a scalar base, loads and evaluates; no walk along
  // the prefix, no per-lane LDS reads, PMB_ILP windows' loads in flight at once.  Measured on the 16-room batch: 293 -> 272 us
  // (four windows per pass, rows >= 32), -> ~255 us with two windows per pass and rows >= 64 (the mean row is 112 pixels: a pass of 128).
  // The launch is bound by the instructions it ISSUES, vector and scalar together (one of each per cycle and CU): halving the vector
  // instructions of a window (they became scalar ones), four windows' loads in flight instead of one, and every load redirected to one
  // cache-resident 64 KB window each moved the kernel by less than 5 % (LAB_NOTES, round 6).
  if (PMB_LONG > 0) {
    const int gz = wstep >> 6, zme = wfirst >> 6;                  // long walks of few images are split over gridDim.z workgroups
    int widx = 0;
    unsigned long long m_fast = 0ull, m_pos0 = 0ull, m_pos1 = 0ull;
    if (pow2) {
      const float a0 = fabsf(r0), a1 = fabsf(r1);
      m_fast = __ballot((r0 == 0.f || (a0 > 1e-18f && a0 < 1e18f)) && (r1 == 0.f || (a1 > 1e-18f && a1 < 1e18f)));
      m_pos0 = __ballot(r0 > 0.f); m_pos1 = __ballot(r1 > 0.f);
    }
#pragma unroll
    for (int scan = 0; scan < 2; ++scan) {
      unsigned long long m = __ballot(scan == 0 ? long_o : long_i);
      const int4 myref = scan == 0 ? *reinterpret_cast<const int4*>(&rin_keep) : *reinterpret_cast<const int4*>(&rout_keep);
      while (m != 0ull) {
        const int l = (int)__builtin_ctzll(m);
        m &= m - 1ull;
        const int len = __builtin_amdgcn_readlane(scan == 0 ? lo_all : li_all, l);
        const int from = __builtin_amdgcn_readlane(scan == 0 ? sp.ofrom : sp.ifrom, l);
        const float crs = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sp.cross), l));
        const float q0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r0), l));
        const float q1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r1), l));
        int4 ru;
        ru.x = __builtin_amdgcn_readlane(myref.x, l); ru.y = __builtin_amdgcn_readlane(myref.y, l);
        ru.z = __builtin_amdgcn_readlane(myref.z, l); ru.w = __builtin_amdgcn_readlane(myref.w, l);
        const typename View::Ref ref = *reinterpret_cast<const typename View::Ref*>(&ru);
        const int d0r = c_from + l;
        // the fast form of the distance term (below): the row's flag and the signs of its two ratios are bits of three ballots taken once
        // per round, picked and turned into +-eps by the scalar unit
        const bool fast = ((m_fast >> l) & 1ull) != 0ull;
        const float e0 = (((m_pos0 >> l) & 1ull) != 0ull) == (dir > 0) ? eps : -eps;
        const float e1 = (((m_pos1 >> l) & 1ull) != 0ull) == (dir > 0) ? eps : -eps;
        const float qs0 = q0 * s2, qs1 = q1 * s2;
#ifdef PMB_STAMP
        ++N_rows;
#endif
        for (int k = 0; k < len; k += 64 * PMB_ILP, ++widx) {
          if (gz > 1 && widx % gz != zme) continue;
#ifdef PMB_STAMP
          const unsigned long long T_a = PMB_T();
#endif
          typename View::Loaded ld[PMB_ILP];
#pragma unroll
          for (int u = 0; u < PMB_ILP; ++u) {
            const int t = min(k + 64 * u + lane, len - 1);            // (clamped: the load is unconditional, the result masked below)
            ld[u] = V.issue(V.idx(d0r, from + t), ref);
          }
#pragma unroll
          for (int u = 0; u < PMB_ILP; ++u) View::pin(ld[u]);
#ifdef PMB_STAMP
          const unsigned long long T_b = PMB_T();
          T_wait += T_b - T_a; ++N_pass;
#endif
#pragma unroll
          for (int u = 0; u < PMB_ILP; ++u) {
            const int t = k + 64 * u + lane;
            int fq;
            const float diff = V.eval(ld[u], ref, fq);
            if (t < len && (scan == 0 || fq == fn) && diff > 0.f) {       // inward: this face's pixels only
              const float dc = (from + t) - crs;
              if (pow2 && scan == 0 && fast) {
                // outward row, power-of-two image: every pixel lies beyond the crossing in direction dir (dc != 0, sign(dc) = dir), so
                // the sign of dist - hence the sign of its eps - is the row's, and (q * dc) * s2 == (q * s2) * dc bit for bit (s2 is a
                // power of two, |q| within 1e-18 .. 1e18: no under- or overflow): multiply, add, reciprocal, multiply, subtract per term
                // instead of two multiplications, compare, select, add, reciprocal, multiply, subtract
                if (q0 != 0.f) acc0 -= PMB_DIV(diff, qs0 * dc + e0);
                if (q1 != 0.f) acc1 -= PMB_DIV(diff, qs1 * dc + e1);
              } else {
              if (q0 != 0.f) {
                float dist = pix_scale(q0 * dc, is, pow2, s2);
                dist = 0 < dist ? dist + eps : dist - eps;
                acc0 -= PMB_DIV(diff, dist);
              }
              if (q1 != 0.f) {
                float dist = pix_scale(q1 * dc, is, pow2, s2);
                dist = 0 < dist ? dist + eps : dist - eps;
                acc1 -= PMB_DIV(diff, dist);
              }
              }
            }
          }
#ifdef PMB_STAMP
          asm volatile("" : "+v"(acc0), "+v"(acc1));
          T_eval += PMB_T() - T_b;
#endif
        }
      }
    }
  }
#ifdef PMB_STAMP
  const unsigned long long T_2b0 = PMB_T();
#endif

  // ---- phase 2b: the scan pixels of the remaining (short) rows of all steps, flattened ----
  // Few images (gridDim.z > 1, pixel_map_scan_split): the launch lasts as long as the longest walk - a wall edge is 64 steps
  // x up to 256 scan pixels per chunk = 256 windows of two dependent loads each, 150 us - so gridDim.z workgroups repeat
  // phase 1 (one memory round trip) and deal the windows among themselves.
  int l0 = 0;                                    // row of the previous window's last pixel: rows only move forward
  for (int w0 = wfirst; w0 < W; w0 += wstep) {
    const int w = min(w0 + lane, W - 1);
    int l = l0;                                  // largest l with s_pre[l] <= w: a short walk instead of a 6-step bisection
    int pre = s_pre[l];
    for (int nx = s_pre[l + 1]; nx <= w; nx = s_pre[l + 1]) { ++l; pre = nx; }
    l0 = __shfl(l, 63, 64);
    if (w0 + lane >= W) continue;
    const int t = w - pre;
    PmbStep st = s_step[l];
    asm volatile("" : "+v"(st.lo), "+v"(st.ofrom), "+v"(st.ifrom), "+v"(st.cross));      // one ds_read_b128 (hipcc read two words and fetched the others in two branches)
    const bool outward = t < st.lo;
    const int d1 = t + (outward ? st.ofrom : st.ifrom - st.lo);
    const typename View::Ref ref = s_ref[l][outward ? 0 : 1];
    int fq;
    const float diff = V.eval(V.load(V.idx(c_from + l, d1), ref), ref, fq);
    if (!outward && fq != fn) continue;          // inward: this face's pixels only
    if (diff > 0.f) {
      const float2 rq = s_ratio[l];
      const float dc = d1 - st.cross;
      if (rq.x != 0.f) {
        float dist = pix_scale(rq.x * dc, is, pow2, s2);
        dist = 0 < dist ? dist + eps : dist - eps;
        acc0 -= PMB_DIV(diff, dist);
      }
      if (rq.y != 0.f) {
        float dist = pix_scale(rq.y * dc, is, pow2, s2);
        dist = 0 < dist ? dist + eps : dist - eps;
        acc1 -= PMB_DIV(diff, dist);
      }
    }
  }
#ifdef PMB_STAMP
  asm volatile("" : "+v"(acc0), "+v"(acc1));
  T_2b += PMB_T() - T_2b0;
#endif
  }   // rounds
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { acc0 += __shfl_xor(acc0, off, 64); acc1 += __shfl_xor(acc1, off, 64); }
  if (lane == 0) {
    if (acc0 != 0.f) atomicAdd(gfaces + 9 * i + pi[0] * 3 + (1 - axis), acc0);
    if (acc1 != 0.f) atomicAdd(gfaces + 9 * i + pi[1] * 3 + (1 - axis), acc1);
#ifdef PMB_STAMP
    unsigned long long* o = g_pmb_stamp[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) & (PMB_STAMP_SLOTS - 1)];
    o[0] = PMB_T() - T_start; o[1] = T_pro; o[2] = T_p1; o[3] = T_wait; o[4] = T_eval; o[5] = N_pass; o[6] = T_2b; o[7] = N_rows;
    o[8] = W_start; o[9] = (unsigned long long)wall_clock64();
#endif
  }
}

#define PMB_KERNEL(PIX, is) ((((is) & ((is) - 1)) == 0) ? pixel_map_backward_kernel<PIX, true> : pixel_map_backward_kernel<PIX, false>)

// ----------------------------------------------------------------------------------------------------
// projection + vertex->face gather (neural_renderer.projection with the reference README's patch - no distortion - followed by
// vertices_to_faces), forward and backward, one thread per face corner.  In torch this is ~30 small kernels around the
// rasterizer (a quarter of a fused scene iteration); the arithmetic follows the torch expression order of
// host/neural_renderer.py::projection.
// ----------------------------------------------------------------------------------------------------
struct Cam { float K[9], R[9], t[3]; };
__device__ __forceinline__ Cam load_cam(const float* K, const float* R, const float* t, int b) {
  Cam c;
#pragma unroll
  for (int k = 0; k < 9; ++k) { c.K[k] = K[9 * b + k]; c.R[k] = R[9 * b + k]; }
#pragma unroll
  for (int k = 0; k < 3; ++k) c.t[k] = t[3 * b + k];
  return c;
}

__global__ void project_faces_kernel(const float* __restrict__ verts, const int32_t* __restrict__ faces, const float* __restrict__ K,
                                     const float* __restrict__ R, const float* __restrict__ t, int V, int F, long n, float os,
                                     float eps, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;       // (b, f, corner)
  if (i >= n) return;
  const int b = (int)(i / (3L * F));
  int vi = faces[i];
  vi = min(max(vi, 0), V - 1);
  const float* p = verts + ((long)b * V + vi) * 3;
  const Cam c = load_cam(K, R, t, b);
  const float x = p[0] * c.R[0] + p[1] * c.R[1] + p[2] * c.R[2] + c.t[0];
  const float y = p[0] * c.R[3] + p[1] * c.R[4] + p[2] * c.R[5] + c.t[1];
  const float z = p[0] * c.R[6] + p[1] * c.R[7] + p[2] * c.R[8] + c.t[2];
  const float xh = x / (z + eps), yh = y / (z + eps);
  float u = xh * c.K[0] + yh * c.K[1] + c.K[2];
  float v = os - (xh * c.K[3] + yh * c.K[4] + c.K[5]);
  u = 2.f * (u - os / 2.f) / os;
  v = 2.f * (v - os / 2.f) / os;
  out[3 * i] = u; out[3 * i + 1] = v; out[3 * i + 2] = z;
}

__global__ void project_faces_bwd_kernel(const float* __restrict__ verts, const int32_t* __restrict__ faces, const float* __restrict__ K,
                                         const float* __restrict__ R, const float* __restrict__ t, int V, int F, long n, float os,
                                         float eps, const float* __restrict__ gout, float* __restrict__ gverts) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int b = (int)(i / (3L * F));
  int vi = faces[i];
  if (vi < 0 || vi >= V) return;
  const float gu = gout[3 * i], gv = gout[3 * i + 1], gz = gout[3 * i + 2];
  if (gu == 0.f && gv == 0.f && gz == 0.f) return;                    // most faces are hidden: nothing to scatter
  const float* p = verts + ((long)b * V + vi) * 3;
  const Cam c = load_cam(K, R, t, b);
  const float x = p[0] * c.R[0] + p[1] * c.R[1] + p[2] * c.R[2] + c.t[0];
  const float y = p[0] * c.R[3] + p[1] * c.R[4] + p[2] * c.R[5] + c.t[1];
  const float z = p[0] * c.R[6] + p[1] * c.R[7] + p[2] * c.R[8] + c.t[2];
  const float iz = 1.f / (z + eps);
  // u = 2 (K0 xh + K1 yh + K2 - os/2) / os ; v = 2 (os - (K3 xh + K4 yh + K5) - os/2) / os
  const float s = 2.f / os;
  const float gxh = s * (gu * c.K[0] - gv * c.K[3]);
  const float gyh = s * (gu * c.K[1] - gv * c.K[4]);
  const float gx = gxh * iz, gy = gyh * iz;
  const float gzc = gz - (gxh * x + gyh * y) * iz * iz;
  float* g = gverts + ((long)b * V + vi) * 3;
  atomicAdd(g + 0, gx * c.R[0] + gy * c.R[3] + gzc * c.R[6]);
  atomicAdd(g + 1, gx * c.R[1] + gy * c.R[4] + gzc * c.R[7]);
  atomicAdd(g + 2, gx * c.R[2] + gy * c.R[5] + gzc * c.R[8]);
}

// Deterministic form (SLN_DETERMINISTIC): one WAVEFRONT per vertex; lane l takes the face corners l, l + 64, .. of its image in
// ascending order and adds the contributions of those that reference the vertex, then the 64 partial sums meet in a fixed
// shuffle tree - a fixed association of the same terms on every run (the scatter above adds them with float atomics in arrival
// order).  Round 3 walked all 3 F corners in ONE thread per vertex (O(V F) dependent L2 loads per thread: 1.2 ms of the 1.8 ms
// a deterministic 16-room scene pass took); the corner list of an image is ~50 KB and stays in L2.
__global__ __launch_bounds__(64) void project_faces_bwd_det_kernel(const float* __restrict__ verts, const int32_t* __restrict__ faces,
                                                                   const float* __restrict__ K, const float* __restrict__ R,
                                                                   const float* __restrict__ t, int V, int F, float os, float eps,
                                                                   const float* __restrict__ gout, float* __restrict__ gverts) {
  const int b = blockIdx.y, vi = blockIdx.x, lane = threadIdx.x;
  const float* p = verts + ((long)b * V + vi) * 3;
  const Cam c = load_cam(K, R, t, b);
  const float x = p[0] * c.R[0] + p[1] * c.R[1] + p[2] * c.R[2] + c.t[0];
  const float y = p[0] * c.R[3] + p[1] * c.R[4] + p[2] * c.R[5] + c.t[1];
  const float z = p[0] * c.R[6] + p[1] * c.R[7] + p[2] * c.R[8] + c.t[2];
  const float iz = 1.f / (z + eps), s = 2.f / os;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  const long base = (long)b * F * 3;
  auto corner = [&](long j) {
    const float gu = gout[3 * (base + j)], gv = gout[3 * (base + j) + 1], gz = gout[3 * (base + j) + 2];
    if (gu == 0.f && gv == 0.f && gz == 0.f) return;
    const float gxh = s * (gu * c.K[0] - gv * c.K[3]);
    const float gyh = s * (gu * c.K[1] - gv * c.K[4]);
    const float gx = gxh * iz, gy = gyh * iz;
    const float gzc = gz - (gxh * x + gyh * y) * iz * iz;
    a0 += gx * c.R[0] + gy * c.R[3] + gzc * c.R[6];
    a1 += gx * c.R[1] + gy * c.R[4] + gzc * c.R[7];
    a2 += gx * c.R[2] + gy * c.R[5] + gzc * c.R[8];
  };
  // four consecutive corners per lane and step (one 16-byte load when the image's corner list is 16-byte aligned: 3 F % 4 == 0),
  // two steps in flight: the walk is a chain of L2 round trips otherwise
  const long n = 3L * F;
  const bool al = ((base & 3) == 0) && ((n & 3) == 0);
  if (al) {
    const int4* f4 = reinterpret_cast<const int4*>(faces + base);
    const long n4 = n >> 2;
    for (long j = lane; j < n4; j += 128) {
      const int4 u = f4[j];
      const long j2 = j + 64;
      const int4 w = j2 < n4 ? f4[j2] : make_int4(-1, -1, -1, -1);
      if (u.x == vi) corner(4 * j); if (u.y == vi) corner(4 * j + 1); if (u.z == vi) corner(4 * j + 2); if (u.w == vi) corner(4 * j + 3);
      if (w.x == vi) corner(4 * j2); if (w.y == vi) corner(4 * j2 + 1); if (w.z == vi) corner(4 * j2 + 2); if (w.w == vi) corner(4 * j2 + 3);
    }
  } else {
    for (long j = lane; j < n; j += 64) if (faces[base + j] == vi) corner(j);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { a0 += __shfl_xor(a0, off, 64); a1 += __shfl_xor(a1, off, 64); a2 += __shfl_xor(a2, off, 64); }
  if (lane == 0) {
    float* g = gverts + ((long)b * V + vi) * 3;
    g[0] = a0; g[1] = a1; g[2] = a2;
  }
}

}  // namespace
#ifdef PMB_STAMP
extern "C" int sln_lab_pmb_stamps(unsigned long long* host, int clear) {
  if (clear) { void* p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_pmb_stamp)) != hipSuccess) return -1; return (int)hipMemset(p, 0, sizeof(g_pmb_stamp)); }
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pmb_stamp), sizeof(g_pmb_stamp));
}
#endif

// ====================================================================================================
// C ABI
// ====================================================================================================
extern "C" {

int64_t sln_raster_workspace_bytes(int B, int F) { return (int64_t)(sizeof(FaceRec) + sizeof(BBox8)) * B * F + 512; }

int sln_raster_forward(const float* faces, int B, int F, int image_size, float near, float far, void* workspace,
                       int32_t* face_index, float* weight, float* depth, void* stream) {
  if (!faces || !workspace || !face_index || !weight || !depth || B <= 0 || F < 0 || image_size <= 0) return SLN_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  FaceRec* rec = static_cast<FaceRec*>(workspace);
  const long n = (long)B * F;
  BBox8* bbox = reinterpret_cast<BBox8*>(static_cast<char*>(workspace) + ((sizeof(FaceRec) * (size_t)n + 255) & ~size_t(255)));
  SlnProfScope prof(SLN_FAM_RASTER, 36.0 * n + 20.0 * B * image_size * image_size, st);
  if (n > 0) hipLaunchKernelGGL(raster_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, faces, n, image_size, rec, bbox);
  const int tiles = sln_cdiv(image_size, TS) * sln_cdiv(image_size, TS);
  hipLaunchKernelGGL((raster_tile_kernel<false>), raster_tile_grid(tiles, B), dim3(256), 0, st, rec, bbox, F, image_size, near, near, far,
                     face_index, weight, depth, (int32_t*)nullptr, (float*)nullptr, (float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, 0.f,
                     (float*)nullptr, (const int32_t*)nullptr, 0, (SceneStats*)nullptr, raster_tile_arg(B));
  SLN_CHECK_LAUNCH();
  return 0;
}

// faces_xyz[B,F,3,3] = (x_ndc, y_ndc, z_cam) of every face corner: projection (K, R, t, orig_size; no distortion) of
// vertices[B,V,3] gathered by faces[B,F,3]
int sln_project_faces(const float* vertices, const int32_t* faces, const float* K, const float* R, const float* t, int B, int V, int F,
                      float orig_size, float eps, float* faces_xyz, void* stream) {
  if (!vertices || !faces || !K || !R || !t || !faces_xyz || B <= 0 || V <= 0 || F < 0) return SLN_E_BADARG;
  const long n = (long)B * F * 3;
  if (n == 0) return 0;
  hipLaunchKernelGGL(project_faces_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, vertices, faces, K, R, t, V, F, n,
                     orig_size, eps, faces_xyz);
  SLN_CHECK_LAUNCH();
  return 0;
}

// grad_vertices[B,V,3] = the adjoint of sln_project_faces applied to grad_faces_xyz (overwritten, fp32 atomics over shared corners)
int sln_project_faces_backward(const float* vertices, const int32_t* faces, const float* K, const float* R, const float* t, int B, int V,
                               int F, float orig_size, float eps, const float* grad_faces_xyz, float* grad_vertices, void* stream) {
  if (!vertices || !faces || !K || !R || !t || !grad_faces_xyz || !grad_vertices || B <= 0 || V <= 0 || F < 0) return SLN_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const int e = sln_zero_async(grad_vertices, sizeof(float) * 3 * (size_t)B * V, st);
  if (e != 0) return e;
  const long n = (long)B * F * 3;
  if (n == 0) return 0;
  if (g_sln_deterministic)
    hipLaunchKernelGGL(project_faces_bwd_det_kernel, dim3(V, B), dim3(64), 0, st, vertices, faces, K, R, t, V, F, orig_size, eps,
                       grad_faces_xyz, grad_vertices);
  else
  hipLaunchKernelGGL(project_faces_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, vertices, faces, K, R, t, V, F, n, orig_size,
                     eps, grad_faces_xyz, grad_vertices);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_raster_forward_dual(const float* faces, int B, int F, int image_size, float near_a, float near_b, float far,
                            void* workspace, int32_t* fi_a, float* w_a, float* d_a, int32_t* fi_b, float* w_b,
                            float* d_b, void* stream) {
  if (!faces || !workspace || !fi_a || !w_a || !d_a || !fi_b || !w_b || !d_b || B <= 0 || F < 0 || image_size <= 0)
    return SLN_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  FaceRec* rec = static_cast<FaceRec*>(workspace);
  const long n = (long)B * F;
  BBox8* bbox = reinterpret_cast<BBox8*>(static_cast<char*>(workspace) + ((sizeof(FaceRec) * (size_t)n + 255) & ~size_t(255)));
  SlnProfScope prof(SLN_FAM_RASTER, 36.0 * n + 40.0 * B * image_size * image_size, st);
  if (n > 0) hipLaunchKernelGGL(raster_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, faces, n, image_size, rec, bbox);
  const int tiles = sln_cdiv(image_size, TS) * sln_cdiv(image_size, TS);
  hipLaunchKernelGGL((raster_tile_kernel<true>), raster_tile_grid(tiles, B), dim3(256), 0, st, rec, bbox, F, image_size, near_a, near_b, far,
                     fi_a, w_a, d_a, fi_b, w_b, d_b, (const float*)nullptr, (const float*)nullptr, 0, 0.f, (float*)nullptr, (const int32_t*)nullptr, 0,
                     (SceneStats*)nullptr, raster_tile_arg(B));
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_raster_texture_sample(const float* faces, const float* textures, const int32_t* face_index, const float* weight,
                              const float* depth, int B, int F, int image_size, int texture_size, float eps, float* rgb,
                              void* stream) {
  if (!faces || !textures || !face_index || !weight || !depth || !rgb || texture_size < 2) return SLN_E_BADARG;
  const long npix = (long)B * image_size * image_size;
  hipLaunchKernelGGL(texture_sample_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, faces,
                     textures, face_index, weight, depth, F, image_size, texture_size, eps, npix, rgb);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_raster_backward_depth(const float* faces, const int32_t* face_index, const float* weight, const float* depth,
                              const float* grad_depth, int B, int F, int image_size, float* grad_faces, void* stream) {
  if (!faces || !face_index || !weight || !depth || !grad_depth || !grad_faces) return SLN_E_BADARG;
  const long npix = (long)B * image_size * image_size;
  hipStream_t st = (hipStream_t)stream;
  SlnProfScope prof(SLN_FAM_RASTER_BWD, 28.0 * npix, st);
  (void)npix;
  if ((long)B * F > 0)
    // (deterministic mode: one wavefront per face - a single add per value onto the caller's zeros)
    hipLaunchKernelGGL(depth_backward_face_kernel, dim3(depth_bwd_grid_x(B, F), g_sln_deterministic ? 1 : depth_bwd_split((long)B * F, F)), dim3(64), 0, st, faces,
                       face_index, weight, depth, grad_depth, F, image_size, grad_faces, depth_bwd_xcd(B));
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_raster_backward_rgb(const float* faces, const int32_t* face_index, const float* rgb, const float* grad_rgb, int B,
                            int F, int image_size, int channels, float eps, float* grad_faces, void* stream) {
  if (!faces || !face_index || !rgb || !grad_rgb || !grad_faces || channels <= 0) return SLN_E_BADARG;
  if ((long)image_size * image_size >= (1L << 31)) return SLN_E_UNSUPPORTED;      // 32-bit pixel offsets inside an image
  if (!pixel_map_grid_ok(B, F)) return SLN_E_UNSUPPORTED;
  const long n = (long)B * F;
  if (n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  SlnProfScope prof(SLN_FAM_RASTER_BWD, 8.0 * channels * B * image_size * image_size + 72.0 * n, st);
  PixDense pix{face_index, rgb, grad_rgb, channels, image_size};
  // (deterministic mode: no scan split - every gradient value then receives exactly two adds, one per incident edge, and a + b
  // is b + a)
  hipLaunchKernelGGL(PMB_KERNEL(PixDense, image_size), dim3(pixel_map_grid_x(B, F), pixel_map_grid_y(B), g_sln_deterministic ? 1 : pixel_map_scan_split(n, B)), dim3(64), 0, st, faces, (const FaceRec*)nullptr,
                     grad_faces, B, F, image_size, eps, pix);
  SLN_CHECK_LAUNCH();
  return 0;
}

// Renderer rgb mode in one launch: see texture_sample_chw_kernel.
int sln_raster_texture_sample_chw(const float* faces, const float* textures, const int32_t* face_index, const float* weight,
                                  const float* depth, int B, int F, int F_tex, int image_size, int texture_size, float eps, float scale,
                                  float* rgb_chw, void* stream) {
  if (!faces || !textures || !face_index || !weight || !depth || !rgb_chw || texture_size < 2) return SLN_E_BADARG;
  if (F_tex != F && 2 * F_tex != F) return SLN_E_BADARG;
  const long npix = (long)B * image_size * image_size;
  if (npix <= 0) return 0;
  hipLaunchKernelGGL(texture_sample_chw_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, (hipStream_t)stream, faces, textures,
                     face_index, weight, depth, F, F_tex, image_size, texture_size, eps, scale, npix, rgb_chw);
  SLN_CHECK_LAUNCH();
  return 0;
}

// backward_pixel_map of P rgb passes over the same face-index map in one walk: see PixMulti.  rgb_chw / grad_chw: DEVICE arrays
// of P pointers to [B,3,is,is] row-flipped images; mask_ws: B * is * is 8-byte words of scratch; P <= 64.
int sln_raster_backward_rgb_multi(const float* faces, const int32_t* face_index, const float* const* rgb_chw,
                                  const float* const* grad_chw, int P, int B, int F, int image_size, float eps, void* mask_ws,
                                  float* grad_faces, void* stream) {
  if (!faces || !face_index || !rgb_chw || !grad_chw || !mask_ws || !grad_faces || P < 1 || P > 64) return SLN_E_BADARG;
  if ((long)image_size * image_size * 3 * B >= (1L << 31)) return SLN_E_UNSUPPORTED;      // 32-bit offsets inside a pass tensor
  if (!pixel_map_grid_ok(B, F)) return SLN_E_UNSUPPORTED;
  const long n = (long)B * F;
  if (n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int plane = image_size * image_size;
  const long npix = (long)B * plane;
  SlnProfScope prof(SLN_FAM_RASTER_BWD, 24.0 * P * npix + 72.0 * n, st);
  unsigned long long* mask = static_cast<unsigned long long*>(mask_ws);
  hipLaunchKernelGGL(multi_mask_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, rgb_chw, P, plane, npix, mask);
  int shift = -1;
  if ((image_size & (image_size - 1)) == 0) { shift = 0; while ((1 << shift) < image_size) ++shift; }
  PixMulti pix{face_index, rgb_chw, grad_chw, mask, image_size, shift};
  hipLaunchKernelGGL(PMB_KERNEL(PixMulti, image_size), dim3(pixel_map_grid_x(B, F), pixel_map_grid_y(B), g_sln_deterministic ? 1 : pixel_map_scan_split(n, B)),
                     dim3(64), 0, st, faces, (const FaceRec*)nullptr, grad_faces, B, F, image_size, eps, pix);
  SLN_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"

// ====================================================================================================
// Fused scene pass: everything models/diff_render.py:359-434 does after the mesh buffers are assembled,
// in ONE rasterisation instead of 33 (1 depth + 32 class passes over identical geometry).
//   final[b, 0]           = depth (rows flipped, values > 15 -> -1)
//   final[b, 1 + chan[c]] = class image of class c (mean of three equal rgb channels of the uniform-texture pass)
//   final[b, 41 + dch[c]] = where(mask_c, depth, mean_c(depth)) / wall_max      (dch[c] >= 0 only)
// with mask_c = class image > 0.1, mean_c over mask_c (wall_max when empty), wall_max = max depth over the
// class-0 ("wall") mask (10 when empty).
// ====================================================================================================
namespace {

// Side stream of the scene backward (one per device, created on first use, kept for the life of the process).  SLN_SCENE_NO_SIDE=1
// keeps every launch on the caller's stream.
struct SceneSide { hipStream_t stream; hipEvent_t fork, mid, join; };      // stream: a pooled one that overlaps with the caller's, per call
SceneSide* scene_side() {
  static const bool off = [] { const char* v = std::getenv("SLN_SCENE_NO_SIDE"); return v && v[0] == '1'; }();
  if (off) return nullptr;
  static std::mutex mu;
  static SceneSide* per_dev[64] = {};
  static bool failed[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (per_dev[dev] == nullptr && !failed[dev]) {
    SceneSide* s = new SceneSide{nullptr, nullptr, nullptr, nullptr};
    if (hipEventCreateWithFlags(&s->fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s->mid, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&s->join, hipEventDisableTiming) != hipSuccess) { failed[dev] = true; delete s; return nullptr; }
    per_dev[dev] = s;
  }
  return per_dev[dev];
}

__global__ void scene_stats_kernel(const int32_t* __restrict__ fi_b, const float* __restrict__ val, const float* __restrict__ d_a,
                                   const int32_t* __restrict__ cls, int F, int is, int NC, SceneStats* __restrict__ st,
                                   FaceRec* __restrict__ rec) {
  const int b = blockIdx.y;
  const long plane = (long)is * is;
  // Per-class depth sums in 64-bit FIXED POINT (2^-32): depths are fp32 values below 16, i.e. multiples of 2^-30 down to 2^-7 - the
  // integer sum is exact and does not depend on the order the lanes arrive in (the LDS float atomics of rounds 1-2 did, at ulp level,
  // and carried the rounding of a 65 536-term fp32 sum); as a double it is a multiple of 2^-32 below 2^21: the cross-block fp64
  // atomics are exact too.  The statistics of the forward pass are order-independent in every mode.
  __shared__ unsigned long long ssum[64]; __shared__ int scnt[64]; __shared__ int svis[64];
  __shared__ int s_wkey, s_wany;        // the block's wall maximum: ONE pair of device atomics per block (every wall pixel - a third
                                        // of a room image - used to issue its own pair on the same two words)
  if (threadIdx.x < 64) { ssum[threadIdx.x] = 0ull; scnt[threadIdx.x] = 0; svis[threadIdx.x] = 0; }
  if (threadIdx.x == 0) { s_wkey = (int)0x80000000; s_wany = 0; }
  __syncthreads();
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < plane; p += (long)gridDim.x * blockDim.x) {
    const long q = b * plane + p;
    // the three streams of a pixel in one round trip (pinned: hipcc sinks each load behind the test that precedes its use)
    int f = fi_b[q]; float v = val[3 * q]; float d = d_a[q];
    asm volatile("" : "+v"(f), "+v"(v), "+v"(d));
    if (f < 0) continue;
    rec[(long)b * F + f].pad_[1] = 1;       // the face owns a pixel of the class pass (raster_prep_kernel cleared the flag): see pixel_map_backward_kernel
    const int c = cls[(long)b * F + f];
    if (c < 0 || c >= NC) continue;
    svis[c] = 1;
    if (!(class_image_value(v) > 0.1f)) continue;
    const float dd = depth_value(d);
    atomicAdd(&ssum[c], (unsigned long long)(long long)rint((double)dd * 4294967296.0)); atomicAdd(&scnt[c], 1);
    if (c == 0) {                       // wall_max = max depth over the wall mask (models/diff_render.py:408-411)
      s_wany = 1;
      atomicMax(&s_wkey, fkey(dd));
    }
  }
  __syncthreads();
  if (threadIdx.x < NC && scnt[threadIdx.x] > 0) {
    atomicAdd(&st[b].sum[threadIdx.x], (double)(long long)ssum[threadIdx.x] * (1.0 / 4294967296.0));
    atomicAdd(&st[b].cnt[threadIdx.x], (double)scnt[threadIdx.x]);
  }
  if (threadIdx.x < NC && svis[threadIdx.x]) st[b].vis[threadIdx.x] = 1;
  if (threadIdx.x == 0 && s_wany) {
    atomicMax(&st[b].wall_any, 1);
    atomicMax(&st[b].wall_key, s_wkey);
  }
}

struct ComposeTabs { int chan[64]; int dch[64]; int owner[32]; float fill[32]; float wall; unsigned char live[72]; };
// live flag of channel ch of an image (see sln_scene_live_channels)
__device__ __forceinline__ unsigned char scene_live_flag(const SceneStats& sb, const int* chan, const int* dch, int NC, int nch, int ch) {
  unsigned char v = (ch == 0 || ch == nch - 1) ? 3 : 0;
  if (!v) {
    if (ch < 41) {
      for (int c = 0; c < NC; ++c) if (chan[c] + 1 == ch && sb.vis[c] != 0) v = 3;
    } else {
      for (int c = 0; c < NC; ++c) if (dch[c] + 41 == ch) v |= sb.cnt[c] > 0.0 ? 3 : 1;
    }
  }
  return v;
}
// The class tables go to LDS in one round trip; the owner search then walks LDS.  (It walked dch[] in global memory with a
// break - up to 32 dependent loads in front of every workgroup's first pixel - and every pixel divided each of its 29
// depth-hot values by wall_max.)
__device__ __forceinline__ void compose_tables(ComposeTabs& t, const int32_t* __restrict__ chan, const int32_t* __restrict__ dch, int NC,
                                               int ndch, const SceneStats& sb, const bool want_live = false) {
  if (threadIdx.x < 64) {
    t.chan[threadIdx.x] = threadIdx.x < NC ? chan[threadIdx.x] : -1;      // image channel (0-based among the 40) of class c
    t.dch[threadIdx.x] = threadIdx.x < NC ? dch[threadIdx.x] : -1;        // depth channel of class c
  }
  __syncthreads();
  if (threadIdx.x < 32) {
    const int k = threadIdx.x;
    int owner = -1;                                                         // class owning depth channel k, -1 if none
    if (k < ndch)
      for (int c = 0; c < NC; ++c) if (t.dch[c] == k) { owner = c; break; }
    t.owner[k] = owner;
    const float wall_max = wall_max_of(sb);
    const int oc = max(owner, 0);
    const double cnt = sb.cnt[oc], sum = sb.sum[oc];                      // unconditional: one round trip together with the wall statistics
    float fill = 0.f;
    if (owner >= 0) fill = (cnt > 0.0 ? (float)(sum / cnt) : wall_max);
    t.fill[k] = fill / wall_max;                                            // mean_c / wall_max: the same quotient for every pixel
    if (k == 0) t.wall = wall_max;
  }
  if (want_live && threadIdx.x >= 64 && threadIdx.x < 64 + 72) {
    const int ch = threadIdx.x - 64;
    t.live[ch] = ch < 41 + ndch ? scene_live_flag(sb, t.chan, t.dch, NC, 41 + ndch, ch) : 0;
  }
  __syncthreads();
}

// One thread per pixel writes all nch channels: the per-pixel inputs are read once (a thread per (pixel, channel) re-read
// them 70 times), every store instruction of a wavefront covers 64 consecutive pixels of one channel plane.
__global__ __launch_bounds__(256) void scene_compose_kernel(const int32_t* __restrict__ fi_b, const float* __restrict__ val,
                                                            const float* __restrict__ d_a, const int32_t* __restrict__ cls,
                                                            const int32_t* __restrict__ chan, const int32_t* __restrict__ dch, int F,
                                                            int is, int NC, int nch, const SceneStats* __restrict__ st,
                                                            float* __restrict__ out, unsigned char* __restrict__ live,
                                                            unsigned char* __restrict__ null_mask) {
  __shared__ ComposeTabs t;
  const int b = blockIdx.y;
  const int ndch = nch - 41;
  compose_tables(t, chan, dch, NC, ndch, st[b], live != nullptr);
  // live != nullptr (sln_scene_forward_live): the image's flags go out, and only the planes flagged 3 are written - the others are
  // all zeros (flag 0) or the constant 1 (flag 1) and the flags say so
  const bool sparse = live != nullptr;
  if (sparse && blockIdx.x == 0 && threadIdx.x < nch) live[b * nch + threadIdx.x] = t.live[threadIdx.x];
  const long plane = (long)is * is;
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= plane) return;
  const int y = (int)((unsigned)p / (unsigned)is), x = (int)((unsigned)p - (unsigned)y * (unsigned)is);      // p < is^2 < 2^31: 32-bit division
  const long q = b * plane + p;
  const float wall_max = t.wall;
  const int f = fi_b[q];
  const int c = f >= 0 ? cls[(long)b * F + f] : -1;
  const bool cvalid = c >= 0 && c < NC;
  const float img = cvalid ? class_image_value(val[3 * q]) : 0.f;
  const float dd = depth_value(d_a[q]);
  float* o = out + ((long)b * nch * is + (is - 1 - y)) * is + x;       // channel stride = plane
  o[0] = dd;
  const int mych = cvalid ? t.chan[c] : -1;
  for (int ch = 1; ch <= 40 && ch < nch; ++ch)
    if (!sparse || t.live[ch] == 3) o[(long)ch * plane] = (mych == ch - 1) ? img : 0.f;
  const float ddq = dd / wall_max;               // one division per pixel: (own depth) / wall_max is the same in every plane it appears in
  const bool own_ok = img > 0.1f;
  float dsum = 0.f;                                // sum of the pixel's depth-hot values in channel order (refine_loss.hip null_mask_kernel)
  for (int k = 0; k < ndch; ++k) {
    const int owner = t.owner[k];
    float v = 0.f;
    if (owner >= 0) v = (c == owner && own_ok) ? ddq : t.fill[k];
    if (!sparse || t.live[41 + k] == 3) o[(long)(41 + k) * plane] = v;
    dsum += v;
  }
  if (null_mask != nullptr) null_mask[(long)b * plane + (long)(is - 1 - y) * is + x] = dsum < 0.5f ? 1 : 0;
}

// sum over the NOT-masked pixels of each depth-hot channel's incoming gradient (-> d mean_c), as (sum over ALL pixels of the
// channel) - (sum over the pixels inside the owner class's mask): the first part is a plain streaming reduction of each
// channel plane, the second reads one value per masked pixel.  (A block per (class, pixel range) re-read the per-pixel
// inputs once per class; a single pass with one accumulator per channel had 29 strided streams per thread and was no faster.)
__global__ void scene_zero_gsum_kernel(SceneStats* st, int B) {      // backward may run more than once per forward
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * 64) st[i / 64].gsum[i % 64] = 0.0;
  if (i < B) st[i].det_ticket = 0;
}

__global__ __launch_bounds__(256) void scene_bwd_plane_sums_kernel(const int32_t* __restrict__ dch, int is, int NC, int nch,
                                                                   const float* __restrict__ gout, SceneStats* __restrict__ st) {
  const int b = blockIdx.z, k = blockIdx.y;                       // depth-hot channel k
  __shared__ int s_owner;
  if (threadIdx.x == 0) s_owner = -1;
  __syncthreads();
  // owner = the lowest class whose depth channel is k: one load per lane and an LDS minimum (the scalar walk with its break was up
  // to 32 dependent loads in front of the stream)
  if (threadIdx.x < 64 && threadIdx.x < NC && dch[threadIdx.x] == k) atomicMin(reinterpret_cast<unsigned*>(&s_owner), (unsigned)threadIdx.x);
  __syncthreads();
  const int owner = s_owner;
  if (owner < 0) return;
  if (!(st[b].cnt[owner] > 0.0)) return;                          // gsum[owner] is only used at the class's own pixels (scene_bwd_depthgrad_kernel)
  const long plane = (long)is * is;
  const float4* src = reinterpret_cast<const float4*>(gout + ((long)b * nch + 41 + k) * plane);
  const long n4 = plane / 4;
  float acc = 0.f;
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x; i0 < n4; i0 += 8 * stride) {
    float4 v[8];                                                   // eight loads in flight, summed in the order of the plain loop
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[min(i0 + u * stride, n4 - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u * stride < n4) acc += (v[u].x + v[u].y) + (v[u].z + v[u].w);
  }
  if (blockIdx.x == 0)                                             // tail when the plane size is not a multiple of 4
    for (long i = n4 * 4 + threadIdx.x; i < plane; i += blockDim.x) acc += gout[((long)b * nch + 41 + k) * plane + i];
  __shared__ float red[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&st[b].gsum[owner], (double)(red[0] + red[1] + red[2] + red[3]));
}

// Deterministic form of the masked sums (SLN_DETERMINISTIC).  DET_BANDS workgroups per image, each over a band of rows: every
// thread keeps its own accumulator per class in LDS (priv[c][thread]: no atomics, a thread adds its pixels in order), thread c sums
// the 256 columns of class c in thread order -> det_part[band][c]; the workgroup that arrives LAST (a ticket: who is last does
// not matter) adds the bands in band order and adds the result to gsum[c] - the only other add into that value is the plane
// sum's, and a + b is b + a.  (Round 3: one workgroup per image walked all 65 536 pixels, 283 us per 16 rooms.)
__global__ __launch_bounds__(256) void scene_bwd_masked_sums_det_kernel(const int32_t* __restrict__ fi_b, const float* __restrict__ val,
                                                                        const int32_t* __restrict__ cls, const int32_t* __restrict__ dch,
                                                                        int F, int is, int NC, int nch, const float* __restrict__ gout,
                                                                        SceneStats* __restrict__ st) {
  extern __shared__ float priv[];            // [NC][256]
  const int b = blockIdx.y, band = blockIdx.x;
  const long plane = (long)is * is;
  const int rows = (is + DET_BANDS - 1) / DET_BANDS;
  const long p0 = (long)band * rows * is, p1 = min((long)(band + 1) * rows * is, plane);
  for (int c = 0; c < NC; ++c) priv[c * 256 + threadIdx.x] = 0.f;
  for (long p = p0 + threadIdx.x; p < p1; p += 256) {
    const long q = b * plane + p;
    const int f = fi_b[q];
    if (f < 0) continue;
    const int c = cls[(long)b * F + f];
    if (c < 0 || c >= NC || dch[c] < 0) continue;
    if (!(class_image_value(val[3 * q]) > 0.1f)) continue;
    const int y = (int)((unsigned)p / (unsigned)is), x = (int)((unsigned)p - (unsigned)y * (unsigned)is);
    priv[c * 256 + threadIdx.x] += gout[(((long)b * nch + 41 + dch[c]) * is + (is - 1 - y)) * is + x];
  }
  __syncthreads();
  if (threadIdx.x < NC) {
    float s = 0.f;
    for (int t = 0; t < 256; ++t) s += priv[threadIdx.x * 256 + t];
    st[b].det_part[band][threadIdx.x] = s;
  }
  __threadfence();
  __syncthreads();
  __shared__ int s_last;
  if (threadIdx.x == 0) s_last = atomicAdd(&st[b].det_ticket, 1) == DET_BANDS - 1;
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  if (threadIdx.x < NC) {
    float s = 0.f;
    for (int k = 0; k < DET_BANDS; ++k) s += __builtin_nontemporal_load(&st[b].det_part[k][threadIdx.x]);
    if (s != 0.f) atomicAdd(&st[b].gsum[threadIdx.x], -(double)s);
  }
}

__global__ __launch_bounds__(256) void scene_bwd_masked_sums_kernel(const int32_t* __restrict__ fi_b, const float* __restrict__ val,
                                                                    const int32_t* __restrict__ cls, const int32_t* __restrict__ dch,
                                                                    int F, int is, int NC, int nch, const float* __restrict__ gout,
                                                                    SceneStats* __restrict__ st) {
  const int b = blockIdx.y;
  const long plane = (long)is * is;
  __shared__ float ssum[64];
  if (threadIdx.x < 64) ssum[threadIdx.x] = 0.f;
  __syncthreads();
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < plane; p += (long)gridDim.x * blockDim.x) {
    const long q = b * plane + p;
    const int f = fi_b[q];
    if (f < 0) continue;
    const int c = cls[(long)b * F + f];
    if (c < 0 || c >= NC || dch[c] < 0) continue;
    if (!(class_image_value(val[3 * q]) > 0.1f)) continue;
    const int y = (int)((unsigned)p / (unsigned)is), x = (int)((unsigned)p - (unsigned)y * (unsigned)is);      // p < is^2 < 2^31: 32-bit division
    atomicAdd(&ssum[c], gout[(((long)b * nch + 41 + dch[c]) * is + (is - 1 - y)) * is + x]);
  }
  __syncthreads();
  if (threadIdx.x < NC && ssum[threadIdx.x] != 0.f) atomicAdd(&st[b].gsum[threadIdx.x], -(double)ssum[threadIdx.x]);
}

// per-pixel class / value maps of the class pass and their transposes (32x32 LDS tiles)
__device__ __forceinline__ void scene_bwd_maps_body(const int b, const int tile_x, PixRec (*tile)[33], const int32_t* __restrict__ fi_b, const float* __restrict__ val,
                                                    const int32_t* __restrict__ cls, const int32_t* __restrict__ chan,
                                                    const float* __restrict__ gout, int F, int is, int NC, int nch,
                                                    PixRec* __restrict__ rec, PixRec* __restrict__ recT) {
  const int x0 = tile_x * 32, y0 = blockIdx.y * 32;
  const long plane = (long)is * is;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int y = y0 + r, x = x0 + tx;
    PixRec pr; pr.fi = -1; pr.cp = -1; pr.v = 0.f; pr.gown = 0.f;
    if (y < is && x < is) {
      const long q = b * plane + (long)y * is + x;
      pr.fi = fi_b[q];
      if (pr.fi >= 0) { pr.cp = cls[(long)b * F + pr.fi]; if (pr.cp >= NC) pr.cp = -1; }
      if (pr.cp >= 0) {
        pr.v = val[3 * q];
        pr.gown = gout[(((long)b * nch + 1 + chan[pr.cp]) * is + (is - 1 - y)) * is + x] / 3.0f;     // == g[b, cp, y, x]
      }
      rec[q] = pr;
    }
    tile[r][tx] = pr;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int x = x0 + r, y = y0 + tx;
    if (x < is && y < is) recT[b * plane + (long)x * is + y] = tile[tx][r];
  }
}

// g[b,c,y,x] = d final[b, 1+chan[c], flip(y), x] / 3  and its transpose
__device__ __forceinline__ void scene_bwd_grad_planes_body(const int bc, const int tile_x, float (*t)[33], const float* __restrict__ gout, const int32_t* __restrict__ chan,
                                                           int is, int NC, int nch, const SceneStats* __restrict__ st,
                                                           float* __restrict__ g, float* __restrict__ gT) {
  const int b = bc / NC, c = bc % NC, x0 = tile_x * 32, y0 = blockIdx.y * 32;
  // a plane is only ever read for the class of a VISIBLE pixel (the reference pixel of a scan): classes without a single
  // visible pixel in this image (typically half of the 32) are skipped - `vis`, the predicate the scan's reference pixels and the
  // live flags of the semantic planes use (not the 0.1 mask count)
  if (st[b].vis[c] == 0) return;
  const long plane = (long)is * is;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* src = gout + ((long)b * nch + 1 + chan[c]) * plane;
  for (int r = ty; r < 32; r += 8) {
    const int y = y0 + r, x = x0 + tx;
    float vv = 0.f;
    if (y < is && x < is) { vv = src[(long)(is - 1 - y) * is + x] / 3.0f; g[(long)bc * plane + (long)y * is + x] = vv; }
    t[r][tx] = vv;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int x = x0 + r, y = y0 + tx;
    if (x < is && y < is) gT[(long)bc * plane + (long)x * is + y] = t[tx][r];
  }
}
#ifndef TABLES_TPB
#define TABLES_TPB 2         // 32 x 32 tiles per workgroup of scene_bwd_tables_kernel (same-box, 16-room batch, 1 / 2 / 4 / 8: 0.4925 / 0.4855 / 0.4859 / 0.4949 ms)
#endif
// Both tables the edge scans read - the per-pixel records and the class-gradient planes, each with its transpose - in ONE launch
// (blockIdx.z < B: records of image z; above: plane (b, c) = z - B).  Round 5: they were two launches on two streams with an event
// between them and the scan kernel; the scan kernel (the longest of the pass, on the critical path) started ~40 us after the
// records were done - the planes' launch began a fork later and the event took another ~13 us to arrive.
__global__ __launch_bounds__(256) void scene_bwd_tables_kernel(const int32_t* __restrict__ fi_b, const float* __restrict__ val,
                                                               const int32_t* __restrict__ cls, const int32_t* __restrict__ chan,
                                                               const float* __restrict__ gout, int B, int F, int is, int NC, int nch,
                                                               const SceneStats* __restrict__ st, PixRec* __restrict__ rec,
                                                               PixRec* __restrict__ recT, float* __restrict__ g, float* __restrict__ gT) {
  __shared__ PixRec tile[32][33];
  // TABLES_TPB tiles of a row of tiles per workgroup (round 6): half as many workgroups next to the depth chain's small kernels (the
  // launch alone is unchanged, 52 us; beside them 78 -> 72 us)
  const int t32 = (is + 31) / 32;
  for (int k = 0; k < TABLES_TPB; ++k) {
    const int tile_x = blockIdx.x * TABLES_TPB + k;
    if (tile_x >= t32) break;
    if (k > 0) __syncthreads();                  // the previous tile's transposed reads are done
    if ((int)blockIdx.z < B) scene_bwd_maps_body(blockIdx.z, tile_x, tile, fi_b, val, cls, chan, gout, F, is, NC, nch, rec, recT);
    else scene_bwd_grad_planes_body(blockIdx.z - B, tile_x, reinterpret_cast<float (*)[33]>(&tile[0][0]), gout, chan, is, NC, nch, st, g, gT);
  }
}

// d(loss)/d(raw depth map of the depth pass), unflipped [B,is,is]
__global__ void scene_bwd_depthgrad_kernel(const int32_t* __restrict__ fi_b, const float* __restrict__ val,
                                           const float* __restrict__ d_a, const int32_t* __restrict__ cls,
                                           const int32_t* __restrict__ dch, int F, int is, int NC, int nch,
                                           const float* __restrict__ gout, const SceneStats* __restrict__ st,
                                           float* __restrict__ gd) {
  const int b = blockIdx.y;
  const long plane = (long)is * is;
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= plane) return;
  const long q = b * plane + p;
  const int y = (int)((unsigned)p / (unsigned)is), x = (int)((unsigned)p - (unsigned)y * (unsigned)is);      // p < is^2 < 2^31: 32-bit division
  float g = 0.f;
  if (!(d_a[q] > 15.f)) {
    const float wall_max = wall_max_of(st[b]);
    const long o = ((long)b * nch * is + (is - 1 - y)) * is + x;      // channel 0
    g = gout[o];
    const int f = fi_b[q];
    const int c = f >= 0 ? cls[(long)b * F + f] : -1;
    if (c >= 0 && c < NC && dch[c] >= 0 && class_image_value(val[3 * q]) > 0.1f) {
      g += gout[o + (long)(41 + dch[c]) * plane] / wall_max;
      if (st[b].cnt[c] > 0.0) g += (float)(st[b].gsum[c] / st[b].cnt[c]) / wall_max;
    }
  }
  gd[q] = g;
}

}  // namespace

extern "C" {

int64_t sln_scene_workspace_bytes(int B, int F, int image_size) {
  const int64_t plane = (int64_t)image_size * image_size;
  // FaceRec | stats | fiA wA dA | fiB wB dB | val(3) | gd | ones texture | pixel records + transpose | g gT (64 class planes max)
  return (int64_t)(sizeof(FaceRec) + sizeof(BBox8)) * B * F + sizeof(SceneStats) * B + B * plane * (4 + 12 + 4) * 2 + B * plane * 12 + B * plane * 4 +
         (int64_t)B * F * 24 * 4 + B * plane * 16 * 2 + (int64_t)B * 64 * plane * 4 * 2 + 8192;
}

struct SceneWs { FaceRec* rec; BBox8* bbox; SceneStats* st; int32_t *fiA, *fiB; float *wA, *dA, *wB, *dB, *val, *gd, *ones;
                 PixRec *prec, *precT; float *g, *gT; };

static SceneWs carve_scene(void* ws, int B, int F, int is) {
  char* p = static_cast<char*>(ws);
  auto take = [&](size_t n) { char* r = p; p += (n + 255) & ~size_t(255); return r; };
  const size_t plane = (size_t)is * is;
  SceneWs w;
  w.rec = (FaceRec*)take(sizeof(FaceRec) * B * F); w.bbox = (BBox8*)take(sizeof(BBox8) * B * F); w.st = (SceneStats*)take(sizeof(SceneStats) * B);
  w.fiA = (int32_t*)take(4 * B * plane); w.wA = (float*)take(12 * B * plane); w.dA = (float*)take(4 * B * plane);
  w.fiB = (int32_t*)take(4 * B * plane); w.wB = (float*)take(12 * B * plane); w.dB = (float*)take(4 * B * plane);
  w.val = (float*)take(12 * B * plane); w.gd = (float*)take(4 * B * plane); w.ones = (float*)take((size_t)B * F * 24 * 4);
  w.prec = (PixRec*)take(16 * B * plane); w.precT = (PixRec*)take(16 * B * plane);
  w.g = (float*)take((size_t)4 * B * 64 * plane); w.gT = (float*)take((size_t)4 * B * 64 * plane);
  return w;
}

// The head of the fused scene pass as ONE launch (it was three, 4.6 + 5.6 + 5.4 us in a row in front of the tile kernel): the per-face
// records, the all-ones texture of the class passes (24 floats per face) and the per-image statistics' start values.
__global__ void scene_prep_kernel(const float* __restrict__ faces, long n, int is, FaceRec* __restrict__ rec, BBox8* __restrict__ bbox,
                                  float* __restrict__ ones, SceneStats* __restrict__ st, int B) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (long)B * 64) {
    const int b = (int)(i / 64), k = (int)(i % 64);
    st[b].sum[k] = 0.0; st[b].cnt[k] = 0.0; st[b].gsum[k] = 0.0; st[b].vis[k] = 0;
    if (k == 0) { st[b].wall_key = (int)0x80000000; st[b].wall_any = 0; }
  }
  if (i < n) {
    float4* o4 = reinterpret_cast<float4*>(ones + 24 * i);       // the workspace is carved in 256-byte steps: 96-byte records are 16-byte aligned
    const float4 one = make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
    for (int q = 0; q < 6; ++q) o4[q] = one;
    raster_prep_face(faces, i, is, rec, bbox);
  }
}

static int scene_forward_impl(const float* faces, const int32_t* face_class, int B, int F, int image_size, int num_classes,
                              const int32_t* class_channel, const int32_t* class_depth_channel, float near_depth, float near_rgb,
                              float far, float tex_eps, void* workspace, float* final_out, unsigned char* live, unsigned char* null_mask,
                              void* stream) {
  if (!faces || !face_class || !class_channel || !class_depth_channel || !workspace || !final_out) return SLN_E_BADARG;
  if (B <= 0 || F <= 0 || image_size <= 0 || num_classes <= 0 || num_classes > 64) return SLN_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const int is = image_size;
  const long plane = (long)is * is, npix = (long)B * plane, n = (long)B * F;
  SceneWs w = carve_scene(workspace, B, F, is);
  SlnProfScope prof(SLN_FAM_RASTER, 36.0 * n + 70.0 * 4.0 * npix, st);
  {
    const long items = n > (long)B * 64 ? n : (long)B * 64;
    hipLaunchKernelGGL(scene_prep_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, st, faces, n, is, w.rec, w.bbox, w.ones, w.st, B);
  }
  const int tiles = sln_cdiv(is, TS) * sln_cdiv(is, TS);
  static const bool tex_apart = std::getenv("SLN_SCENE_TEX_APART") != nullptr;       // lab: the texture sample as its own launch
  if (tex_apart) {
    hipLaunchKernelGGL((raster_tile_kernel<true>), raster_tile_grid(tiles, B), dim3(256), 0, st, w.rec, w.bbox, F, is, near_depth, near_rgb, far,
                       w.fiA, w.wA, w.dA, w.fiB, w.wB, w.dB, (const float*)nullptr, (const float*)nullptr, 0, 0.f, (float*)nullptr,
                       (const int32_t*)nullptr, 0, (SceneStats*)nullptr, raster_tile_arg(B));
    hipLaunchKernelGGL(texture_sample_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, faces, w.ones, w.fiB, w.wB,
                       w.dB, F, is, 2, tex_eps, npix, w.val);
    // wall_max starts at -inf surrogate
    hipLaunchKernelGGL(scene_stats_kernel, dim3(64, B), dim3(256), 0, st, w.fiB, w.val, w.dA, face_class, F, is, num_classes, w.st, w.rec);
  } else {
    hipLaunchKernelGGL((raster_tile_kernel<true, true>), raster_tile_grid(tiles, B), dim3(256), 0, st, w.rec, w.bbox, F, is, near_depth, near_rgb, far,
                       w.fiA, w.wA, w.dA, w.fiB, w.wB, w.dB, faces, (const float*)w.ones, 2, tex_eps, w.val, face_class, num_classes, w.st, raster_tile_arg(B));
  }
  hipLaunchKernelGGL(scene_compose_kernel, dim3((unsigned)((plane + 255) / 256), B), dim3(256), 0, st, w.fiB, w.val, w.dA,
                     face_class, class_channel, class_depth_channel, F, is, num_classes, 70, w.st, final_out, live, null_mask);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_scene_forward(const float* faces, const int32_t* face_class, int B, int F, int image_size, int num_classes,
                      const int32_t* class_channel, const int32_t* class_depth_channel, float near_depth, float near_rgb,
                      float far, float tex_eps, void* workspace, float* final_out, void* stream) {
  return scene_forward_impl(faces, face_class, B, F, image_size, num_classes, class_channel, class_depth_channel, near_depth, near_rgb, far,
                            tex_eps, workspace, final_out, nullptr, nullptr, stream);
}

// The same pass for a consumer that reads the image through its flags (the refinement loss, SlnRefineLoss::live_planes): `live`
// [B, 70] receives what sln_scene_live_channels would write, and only the planes flagged 3 of final_out are written - a plane
// flagged 0 WOULD hold zeros, a plane flagged 1 the constant 1; their memory is left as it was.
int sln_scene_forward_live(const float* faces, const int32_t* face_class, int B, int F, int image_size, int num_classes,
                           const int32_t* class_channel, const int32_t* class_depth_channel, float near_depth, float near_rgb,
                           float far, float tex_eps, void* workspace, float* final_out, unsigned char* live, unsigned char* null_mask,
                           void* stream) {
  if (!live) return SLN_E_BADARG;
  return scene_forward_impl(faces, face_class, B, F, image_size, num_classes, class_channel, class_depth_channel, near_depth, near_rgb, far,
                            tex_eps, workspace, final_out, live, null_mask, stream);
}

// live[b][ch] of the last sln_scene_forward, two bits.  Bit 0: the plane can hold a non-zero value (clear: it is all zeros).
// Bit 1: sln_scene_backward reads the plane's incoming gradient (clear: it never does).
//   channel 0, the last depth-hot channel (the loss fills it where no class has depth): 3
//   semantic channel 1 + k: 3 when the class mapped to NYU index k has a visible pixel in image b, else 0 - the plane is zeros
//     and scene_bwd_grad_planes_body skips it
//   depth-hot channel 41 + k: 3 when its class has a visible pixel; 1 when it has none - the plane is the constant 1
//     (scene_fill_table: mean / wall_max with the mean replaced by wall_max) and its gradient only enters gsum[owner], which
//     scene_bwd_depthgrad_kernel uses at the class's own pixels; 0 when no class owns the channel
// Lets the refinement loss skip those planes (SlnRefineLoss::live_planes).
static __global__ void scene_live_channels_kernel(const SceneStats* __restrict__ st, const int32_t* __restrict__ chan, const int32_t* __restrict__ dch,
                                           int NC, int nch, int B, unsigned char* __restrict__ live) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * nch) return;
  const int b = i / nch, ch = i % nch;
  const unsigned char v = scene_live_flag(st[b], chan, dch, NC, nch, ch);
  live[i] = v;
}

int sln_scene_live_channels(void* workspace, int B, int F, int image_size, int num_classes, const int32_t* class_channel,
                            const int32_t* class_depth_channel, unsigned char* live, void* stream) {
  if (!workspace || !class_channel || !class_depth_channel || !live || B <= 0 || F <= 0 || image_size <= 0 || num_classes <= 0 || num_classes > 64)
    return SLN_E_BADARG;
  SceneWs w = carve_scene(workspace, B, F, image_size);
  hipLaunchKernelGGL(scene_live_channels_kernel, dim3(sln_cdiv(B * 70, 256)), dim3(256), 0, (hipStream_t)stream, w.st, class_channel,
                     class_depth_channel, num_classes, 70, B, live);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_scene_backward(const float* faces, const int32_t* face_class, int B, int F, int image_size, int num_classes,
                       const int32_t* class_channel, const int32_t* class_depth_channel, float pix_eps, void* workspace,
                       const float* grad_final, float* grad_faces, void* stream) {
  if (!faces || !face_class || !class_channel || !class_depth_channel || !workspace || !grad_final || !grad_faces) return SLN_E_BADARG;
  if (B <= 0 || F <= 0 || image_size <= 0 || num_classes <= 0 || num_classes > 64) return SLN_E_BADARG;
  if ((long)num_classes * image_size * image_size >= (1L << 30)) return SLN_E_UNSUPPORTED;   // 32-bit byte offsets inside an image's class planes (pixel_map_backward_kernel)
  if (!pixel_map_grid_ok(B, F)) return SLN_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int is = image_size;
  const long plane = (long)is * is, npix = (long)B * plane, n = (long)B * F;
  SceneWs w = carve_scene(workspace, B, F, is);
  SlnProfScope prof(SLN_FAM_RASTER_BWD, 70.0 * 4.0 * npix + 36.0 * n, st);
  // Two independent chains add into grad_faces: the depth channel's (gradient sums -> depth-map gradient -> per-face walk,
  // five launches, ~0.12 ms per 16 rooms) and the class planes' (packed records, gradient planes, edge scans: ~0.25 ms, bound by
  // instruction issue and by the number of WORKING wavefronts).  The depth chain runs on a side stream next to the class
  // chain; in a stream capture the event edges become graph dependencies (fork / join inside this call).
  // Batches only: with one room (the refinement loop, a captured iteration of ~240 small launches) the three event edges cost
  // more than the overlap returns - 1.37 ms per iteration with the side stream, 1.22 ms without (same-box A/B).
  // Deterministic mode (SLN_DETERMINISTIC): ONE stream, no split units, the depth walk BEHIND the edge scans - every face-gradient
  // value then receives two adds from the scans (one per incident edge: a + b is b + a) onto the zero fill and one more from the
  // depth walk, in stream order; the gradient sums take their fixed-order forms (one workgroup per plane / per image).
  const bool det = g_sln_deterministic != 0;
  // Not inside a stream capture (round 6): a captured fork becomes a forked hipGraph, and this runtime replays forked graphs node
  // by node from the host (0.31-0.40 ms of enqueue per replayed refinement iteration against 0.09 ms for a linear graph) without
  // running the branches side by side - 16 rooms: 1.69-1.72 ms per replayed iteration with the forks, 1.66 ms without, 1.56-1.59 ms
  // eager with them (profiles/r06_refine_batch_by_rooms.txt).  SLN_CAPTURE_SIDE=1 keeps the forks in captures (lab).
  static const bool capture_side = std::getenv("SLN_CAPTURE_SIDE") != nullptr;
  SceneSide* sd = (!det && n >= 16384 && (capture_side || !sln_capturing(st))) ? scene_side() : nullptr;
  // One stream and three events per device serve every caller: two host threads (or a capturing and an eager caller) enqueuing
  // their fork / mid / join records at the same time would cross their dependencies, so the fork..join section is exclusive
  // (host-side enqueue only: microseconds).  The guard below also JOINS on every way out: an error return between fork and join
  // used to leave the side stream un-joined - work on grad_faces still in flight, and an active stream capture invalidated.
  static std::mutex side_mu;
  std::unique_lock<std::mutex> side_lock(side_mu, std::defer_lock);
  // What BOTH chains wait for runs first, on the caller's stream: the zero-fill of the face gradient.  The fork follows at once and the
  // launch that builds the scan kernel's tables (records + gradient planes) comes behind it on the caller's stream, so the depth chain's
  // four small kernels run NEXT TO the tables launch and its per-face walk next to the start of the scans.  (Round 5 forked behind the
  // tables launch.  With the scan kernel in its round-5 dispatch order that was equal - 0.540 vs 0.538 ms per 16-room batch: the scans
  // ended in a 40 us tail of low occupancy that absorbed the depth chain wherever it started.  With the face-major order the tail is
  // gone and the late fork costs: 0.549 vs 0.520 ms, same box.  SLN_SCENE_FORK_LATE=1 restores it.)
  const int t32 = sln_cdiv(is, 32);
  static const bool fork_late = std::getenv("SLN_SCENE_FORK_LATE") != nullptr;      // lab
  auto tables = [&]() {
    hipLaunchKernelGGL(scene_bwd_tables_kernel, dim3(sln_cdiv(t32, TABLES_TPB), t32, B + B * num_classes), dim3(256), 0, st, w.fiB, w.val, face_class, class_channel,
                       grad_final, B, F, is, num_classes, 70, w.st, w.prec, w.precT, w.g, w.gT);
  };
  {
    const int e = sln_zero_async(grad_faces, sizeof(float) * 9 * (size_t)n, st);
    if (e != hipSuccess) return (int)e;
    if (fork_late) tables();
  }
  hipStream_t sd_st = st;
  if (sd != nullptr) {
    side_lock.lock();
    sd->stream = sln_overlapping_stream(st);                                    // (under side_mu: one caller at a time)
    if (sd->stream != nullptr && hipEventRecord(sd->fork, st) == hipSuccess && hipStreamWaitEvent(sd->stream, sd->fork, 0) == hipSuccess) sd_st = sd->stream;
    else { sd = nullptr; side_lock.unlock(); }
  }
  if (!fork_late) tables();
  struct Join {
    SceneSide* sd; hipStream_t st; hipError_t err;
    void run() {
      if (sd == nullptr) return;
      err = hipEventRecord(sd->join, sd->stream);
      if (err == hipSuccess) err = hipStreamWaitEvent(st, sd->join, 0);
      sd = nullptr;
    }
    ~Join() { run(); }
  } join{sd, st, hipSuccess};
  hipLaunchKernelGGL(scene_zero_gsum_kernel, dim3(sln_cdiv(B * 64, 256)), dim3(256), 0, sd_st, w.st, B);
  hipLaunchKernelGGL(scene_bwd_plane_sums_kernel, dim3(det ? 1 : 8, 70 - 41, B), dim3(256), 0, sd_st, class_depth_channel, is, num_classes, 70, grad_final,
                     w.st);
  if (det)
    hipLaunchKernelGGL(scene_bwd_masked_sums_det_kernel, dim3(DET_BANDS, B), dim3(256), sizeof(float) * 256 * (size_t)num_classes, sd_st, w.fiB, w.val,
                       face_class, class_depth_channel, F, is, num_classes, 70, grad_final, w.st);
  else
  hipLaunchKernelGGL(scene_bwd_masked_sums_kernel, dim3(64, B), dim3(256), 0, sd_st, w.fiB, w.val, face_class, class_depth_channel, F, is,
                     num_classes, 70, grad_final, w.st);
  hipLaunchKernelGGL(scene_bwd_depthgrad_kernel, dim3((unsigned)((plane + 255) / 256), B), dim3(256), 0, sd_st, w.fiB, w.val, w.dA,
                     face_class, class_depth_channel, F, is, num_classes, 70, grad_final, w.st, w.gd);
  if (!det)
    hipLaunchKernelGGL(depth_backward_face_kernel, dim3(depth_bwd_grid_x(B, F), depth_bwd_split(n, F)), dim3(64), 0, sd_st, faces, w.fiA, w.wA, w.dA, w.gd, F, is,
                       grad_faces, depth_bwd_xcd(B));
  PixClass pix{w.prec, w.precT, w.g, w.gT, is, num_classes};
  hipLaunchKernelGGL(PMB_KERNEL(PixClass, is), dim3(pixel_map_grid_x(B, F), pixel_map_grid_y(B), det ? 1 : pixel_map_scan_split(n, B)), dim3(64), 0, st, faces, (const FaceRec*)w.rec,
                     grad_faces, B, F, is, pix_eps, pix);
  if (det)
    hipLaunchKernelGGL(depth_backward_face_kernel, dim3(depth_bwd_grid_x(B, F), 1), dim3(64), 0, st, faces, w.fiA, w.wA, w.dA, w.gd, F, is, grad_faces, depth_bwd_xcd(B));
  join.run();                     // whatever follows on `st` sees both chains
  if (join.err != hipSuccess) return (int)join.err;
  SLN_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
