// Refinement loss of the layout-refinement loop (testing/test_render_refine.py:192-215 PSP_pool_new, :332-356 the loss):
//
//   iter[:, -1][sum(iter[:, 41:], 1) < 0.5] = 1                                   null regions
//   depth = L1( cat_s pool_s(iter[:, 41:]), cat_s pool_s(target[:, 41:]) ) * 0.5   pool_s = bilinear(align_corners) to s x s,
//   sem   = sum_s CE( pool_s(iter[:, 1:41]), argmax labels of the target ) / 800            then bilinear to 96 x 96
//   loss  = 100 * depth + 100 * sem  (+ 2 * size_loss, added by the caller)
//
// In torch this is ~200 small launches per iteration (16 resamples forward and backward, softmax / nll per scale, fills, cats)
// and was 60 % of a refinement iteration.  Here: null mask, one resampling kernel for the 4 scales x 69 channels (both
// interpolation stages in registers, same order of operations as upsample_bilinear2d), one loss kernel that leaves
// d loss / d pooled in place of the pooled maps, and for backward ONE gather kernel through the transposed resampling
// operator (a bilinear DOWN-sample without anti-aliasing reads few input pixels, so the image gradient is sparse; every
// input pixel sums the few pooled pixels that read it - no atomics).  The target's pooled depth maps and labels are
// constants of a room; the host derives them from sln_refine_pool of the target (the same resampling kernel).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "sln_common.h"
#include "sln_hip.h"

namespace {

constexpr int MAX_SCALES = 4;

struct RefineDims {
  int B, S, P, C;                 // batch, image size, pooled size, image channels (70)
  int sem0, n_sem, dep0, n_dep;   // semantic channels [sem0, sem0 + n_sem), depth channels [dep0, dep0 + n_dep) (contiguous)
  int n_scales, pmax;             // pmax: row length of the stage-1 tables (largest intermediate size)
  int per_room, pad_;             // per_room: B independent rooms - every room's loss is normalised by its OWN element / label counts
};

// `live` (SlnRefineLoss::live_planes): a plane marked 0 is all zeros, a plane marked 1 the constant 1 - the same sum without the load
__global__ void null_mask_kernel(const float* __restrict__ img, RefineDims d, unsigned char* __restrict__ null,
                                 const unsigned char* __restrict__ live) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long plane = (long)d.S * d.S;
  if (i >= d.B * plane) return;
  const long b = i / plane, pix = i % plane;
  const float* p = img + (b * d.C + d.dep0) * plane + pix;
  float s = 0.f;
  if (live != nullptr) {
    const unsigned char* lv = live + b * d.C + d.dep0;
    for (int c = 0; c < d.n_dep; ++c) { const int k = lv[c]; s += k == 3 ? p[c * plane] : (k == 1 ? 1.f : 0.f); }
  } else
  for (int c = 0; c < d.n_dep; ++c) s += p[c * plane];
  null[i] = s < 0.5f ? 1 : 0;
}

// pooled[b][s][cc][oy][ox], cc = channel - sem0 over the n_sem + n_dep loss channels.  One thread resamples one pooled pixel
// of CG consecutive channels: the 12 table entries that locate its 16 taps are loaded once per thread, not once per channel.
constexpr int CG = 8;
__global__ __launch_bounds__(256) void pool_kernel(const float* __restrict__ img, const unsigned char* __restrict__ null, const int null_fill,
                                                   RefineDims d,
                                                   const int* __restrict__ s2_k0, const int* __restrict__ s2_k1, const float* __restrict__ s2_l1,
                                                   const int* __restrict__ s1_i0, const int* __restrict__ s1_i1, const float* __restrict__ s1_l1,
                                                   float* __restrict__ pooled) {
  const int nc = d.n_sem + d.n_dep, ng = (nc + CG - 1) / CG;
  const long pp = (long)d.P * d.P;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)d.B * d.n_scales * ng * pp) return;
  const int ox = (int)(i % d.P), oy = (int)((i / d.P) % d.P);
  const int cg = (int)((i / pp) % ng), s = (int)((i / (pp * ng)) % d.n_scales), b = (int)(i / (pp * ng * d.n_scales));
  const long plane = (long)d.S * d.S;
  const int ky[2] = {s2_k0[s * d.P + oy], s2_k1[s * d.P + oy]}, kx[2] = {s2_k0[s * d.P + ox], s2_k1[s * d.P + ox]};
  const float ly1 = s2_l1[s * d.P + oy], lx1 = s2_l1[s * d.P + ox], ly0 = 1.f - ly1, lx0 = 1.f - lx1;
  int yy[2][2], xx[2][2]; float hh[2], ww[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    yy[a][0] = s1_i0[s * d.pmax + ky[a]]; yy[a][1] = s1_i1[s * d.pmax + ky[a]]; hh[a] = s1_l1[s * d.pmax + ky[a]];
    xx[a][0] = s1_i0[s * d.pmax + kx[a]]; xx[a][1] = s1_i1[s * d.pmax + kx[a]]; ww[a] = s1_l1[s * d.pmax + kx[a]];
  }
  const unsigned char* nm = null + (long)b * plane;
  float* dst = pooled + ((long)(b * d.n_scales + s) * nc) * pp + (long)oy * d.P + ox;
  for (int cc = cg * CG; cc < min(cg * CG + CG, nc); ++cc) {
    const int c = d.sem0 + cc;
    const bool fill = null_fill && c == d.dep0 + d.n_dep - 1;       // the last depth channel is set to 1 where no class has depth
    const float* src = img + ((long)b * d.C + c) * plane;
    float inter[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int y0 = yy[a][0], y1 = yy[a][1];
      const float h1 = hh[a], h0 = 1.f - h1;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int x0 = xx[e][0], x1 = xx[e][1];
        const float w1 = ww[e], w0 = 1.f - w1;
        float v00 = src[(long)y0 * d.S + x0], v01 = src[(long)y0 * d.S + x1], v10 = src[(long)y1 * d.S + x0], v11 = src[(long)y1 * d.S + x1];
        if (fill) {
          v00 = nm[(long)y0 * d.S + x0] ? 1.f : v00; v01 = nm[(long)y0 * d.S + x1] ? 1.f : v01;
          v10 = nm[(long)y1 * d.S + x0] ? 1.f : v10; v11 = nm[(long)y1 * d.S + x1] ? 1.f : v11;
        }
        inter[a][e] = h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11);     // upsample_bilinear2d's expression
      }
    }
    dst[(long)cc * pp] = ly0 * (lx0 * inter[0][0] + lx1 * inter[0][1]) + ly1 * (lx0 * inter[1][0] + lx1 * inter[1][1]);
  }
}

// The same resampling with the INTERMEDIATE image of a (plane, scale) in LDS (round 5): one workgroup per (image b, loss channel,
// scale).  Phase A evaluates the align_corners=True stage once per intermediate pixel (4 image taps each: sz^2 evaluations instead of
// the 4 P^2 the per-pixel kernel above repeats - 16 640 against 147 456 per plane over the four scales), phase B the second stage
// from LDS with coalesced stores.  Same expressions, same order: the results are bit-identical to pool_kernel's.  With 16 rooms in
// flight pool_kernel was the largest kernel of a refinement iteration (510 us: 680 M scattered 4-byte taps).
constexpr int POOL_LDS_MAX = 96;
__global__ __launch_bounds__(256) void pool_lds_kernel(const float* __restrict__ img, const unsigned char* __restrict__ null, const int null_fill,
                                                       RefineDims d,
                                                       const int* __restrict__ s2_k0, const int* __restrict__ s2_k1, const float* __restrict__ s2_l1,
                                                       const int* __restrict__ s1_i0, const int* __restrict__ s1_i1, const float* __restrict__ s1_l1,
                                                       float* __restrict__ pooled, const unsigned char* __restrict__ live, const int skip_ones,
                                                       const int scale_rev) {
  __shared__ float inter[POOL_LDS_MAX * POOL_LDS_MAX];
  const int nc = d.n_sem + d.n_dep;
  // (planes x scales grid.  Round 5 tried the four scale-workgroups of a plane on one XCD, adjacent in launch order, so that the
  //  plane is read from HBM once: 280 us against 223 - the scales differ 9x in work and interleaving them unbalances the XCDs)
  const int cc = blockIdx.x % nc, b = blockIdx.x / nc, s = scale_rev ? d.n_scales - 1 - (int)blockIdx.y : (int)blockIdx.y;
  const int c = d.sem0 + cc;
  const int lv = live != nullptr ? live[b * d.C + c] : 3;
  if (!(lv & 1)) {                                                // an all-zero plane: its pooled plane is zero (what the taps would give)
    if (cc < d.n_sem) return;                                     // ... and loss_kernel does not read the dead semantic planes at all
    float* dz = pooled + ((long)(b * d.n_scales + s) * nc + cc) * ((long)d.P * d.P);
    for (int o = threadIdx.x; o < d.P * d.P; o += 256) dz[o] = 0.f;
    return;
  }
  const bool ones = lv == 1;                                      // the constant 1: the same expressions on 1.f instead of the four taps
  if (ones && skip_ones) return;                                  // ... or not at all: loss_kernel takes the pooled constant plane from pooled_ones
  const int sz = s2_k1[s * d.P + d.P - 1] + 1;                    // intermediate size of this scale (the last pooled index reads its last row)
  const long plane = (long)d.S * d.S;
  const bool fill = null_fill && c == d.dep0 + d.n_dep - 1;       // the last depth channel is set to 1 where no class has depth
  const float* src = img + ((long)b * d.C + c) * plane;
  const unsigned char* nm = null + (long)b * plane;
  for (int i = threadIdx.x; i < sz * sz; i += 256) {
    const int ky = i / sz, kx = i % sz;
    const int y0 = s1_i0[s * d.pmax + ky], y1 = s1_i1[s * d.pmax + ky], x0 = s1_i0[s * d.pmax + kx], x1 = s1_i1[s * d.pmax + kx];
    const float h1 = s1_l1[s * d.pmax + ky], h0 = 1.f - h1, w1 = s1_l1[s * d.pmax + kx], w0 = 1.f - w1;
    float v00 = 1.f, v01 = 1.f, v10 = 1.f, v11 = 1.f;
    if (!ones) { v00 = src[(long)y0 * d.S + x0]; v01 = src[(long)y0 * d.S + x1]; v10 = src[(long)y1 * d.S + x0]; v11 = src[(long)y1 * d.S + x1]; }
    if (fill) {
      v00 = nm[(long)y0 * d.S + x0] ? 1.f : v00; v01 = nm[(long)y0 * d.S + x1] ? 1.f : v01;
      v10 = nm[(long)y1 * d.S + x0] ? 1.f : v10; v11 = nm[(long)y1 * d.S + x1] ? 1.f : v11;
    }
    inter[i] = h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11);
  }
  __syncthreads();
  float* dst = pooled + ((long)(b * d.n_scales + s) * nc + cc) * ((long)d.P * d.P);
  for (int o = threadIdx.x; o < d.P * d.P; o += 256) {
    const int oy = o / d.P, ox = o % d.P;
    const int ky0 = s2_k0[s * d.P + oy], ky1 = s2_k1[s * d.P + oy], kx0 = s2_k0[s * d.P + ox], kx1 = s2_k1[s * d.P + ox];
    const float ly1 = s2_l1[s * d.P + oy], lx1 = s2_l1[s * d.P + ox], ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    dst[o] = ly0 * (lx0 * inter[ky0 * sz + kx0] + lx1 * inter[ky0 * sz + kx1]) + ly1 * (lx0 * inter[ky1 * sz + kx0] + lx1 * inter[ky1 * sz + kx1]);
  }
}

// one thread per pooled pixel of one scale and part (blockIdx.y: 0 = log-softmax / NLL over the semantic channels, 1.. = |.| over
// DCH depth channels); overwrites pooled with d loss / d pooled.  partial = per-block {sum |diff|, sum_s CE_s / 800 / valid count}.
constexpr int DCH = 8;               // depth channels per thread of loss_kernel's parts 1..
template <int NSEM>
__global__ __launch_bounds__(128) void loss_kernel(float* __restrict__ pooled, RefineDims d, const float* __restrict__ tgt_depth,
                                                   const int* __restrict__ labels, const float* __restrict__ inv_count,
                                                   float2* __restrict__ partial, const unsigned char* __restrict__ live,
                                                   const float* __restrict__ pooled_ones) {
  const int nc = d.n_sem + d.n_dep;
  const long pp = (long)d.P * d.P;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)d.B * d.n_scales * pp;
  float l_abs = 0.f, l_ce = 0.f;
  if (i < total) {
    const long pix = i % pp;
    const int s = (int)((i / pp) % d.n_scales), b = (int)(i / (pp * d.n_scales));
    float* p = pooled + ((long)(b * d.n_scales + s) * nc) * pp + pix;
    if (blockIdx.y == 0) {
      const int t = labels[i];
      // semantic planes marked all-zero (live_planes) enter as the zeros they are, unread (pool_lds_kernel did not write them), and
      // their gradient, which nobody reads, is not stored
      unsigned long long lm = ~0ull;
      if (live != nullptr) {
        lm = 0;
        const unsigned char* lv = live + b * d.C + d.sem0;
        for (int c = 0; c < NSEM; ++c) lm |= (unsigned long long)(lv[c] & 1) << c;
      }
      float v[NSEM];
#pragma unroll
      for (int c = 0; c < NSEM; ++c) v[c] = (lm >> c & 1) ? p[c * pp] : 0.f;
      if (t >= 0) {
        float m = v[0];
#pragma unroll
        for (int c = 1; c < NSEM; ++c) m = fmaxf(m, v[c]);
        float z = 0.f;
#pragma unroll
        for (int c = 0; c < NSEM; ++c) z += expf(v[c] - m);
        const float lse = m + logf(z);
        const float k = inv_count[d.per_room ? b * d.n_scales + s : s] * (1.f / 800.f);
        float vt = 0.f;
#pragma unroll
        for (int c = 0; c < NSEM; ++c) {
          vt = c == t ? v[c] : vt;
          if (lm >> c & 1) p[c * pp] = (expf(v[c] - lse) - (c == t ? 1.f : 0.f)) * (k * 100.f);
        }
        l_ce = (lse - vt) * k;
      } else {
#pragma unroll
        for (int c = 0; c < NSEM; ++c) if (lm >> c & 1) p[c * pp] = 0.f;
      }
    } else {                                                                       // depth channels, DCH per thread
      const float gd = 50.f / (float)((double)(d.per_room ? 1 : d.B) * d.n_scales * d.n_dep * pp);      // 100 * 0.5 / numel
      const float* tg = tgt_depth + ((long)(b * d.n_scales + s) * d.n_dep) * pp + pix;
      float* q = p + (long)d.n_sem * pp;
      const int c0 = ((int)blockIdx.y - 1) * DCH;
      // a depth-hot plane flagged "the constant 1" (live_planes == 1) was not pooled per room: its pooled plane is the same for every
      // room and lives in pooled_ones [n_scales][P][P] (built with the pooling kernel itself); its gradient, which nobody reads, is
      // not stored
      const float one_p = (live != nullptr && pooled_ones != nullptr) ? pooled_ones[(long)s * pp + pix] : 0.f;
      float a[DCH], t[DCH]; bool cst[DCH];
#pragma unroll
      for (int u = 0; u < DCH; ++u) {
        const int c = min(c0 + u, d.n_dep - 1);
        cst[u] = live != nullptr && pooled_ones != nullptr && live[b * d.C + d.dep0 + c] == 1;
        a[u] = cst[u] ? one_p : q[c * pp]; t[u] = tg[c * pp];
      }
#pragma unroll
      for (int u = 0; u < DCH; ++u) {
        if (c0 + u < d.n_dep) {
          const float diff = a[u] - t[u];
          l_abs += fabsf(diff);
          if (!cst[u]) q[(c0 + u) * pp] = diff > 0.f ? gd : (diff < 0.f ? -gd : 0.f);
        }
      }
    }
  }
  // block reduction; one partial per block (a same-address atomic per block is serialised on the memory side: with 1440
  // blocks the three atomics of the first version cost 60 us), summed in a fixed order by loss_finalize_kernel
  __shared__ float red[2][2];
  for (int o = 32; o > 0; o >>= 1) { l_abs += __shfl_down(l_abs, o, 64); l_ce += __shfl_down(l_ce, o, 64); }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[wave][0] = l_abs; red[wave][1] = l_ce; }
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = make_float2(red[0][0] + red[1][0], red[0][1] + red[1][1]);
}

// per_room: workgroup b sums room b's partials - gx blocks per part row of which the room owns bpr consecutive ones (a block of
// loss_kernel never straddles two rooms, checked by the launcher) - in the order the B = 1 launch sums its own
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float2* __restrict__ partial, int n, RefineDims d, float* __restrict__ loss_out,
                                                            const int gx, const int bpr) {
  __shared__ double red[4][2];
  double a = 0.0, c = 0.0;
  if (d.per_room) {
    const int b = blockIdx.x, ny = n / gx;
    loss_out += 3 * b;
    for (int i = threadIdx.x; i < ny * bpr; i += 256) { const float2 v = partial[(i / bpr) * gx + b * bpr + (i % bpr)]; a += (double)v.x; c += (double)v.y; }
  } else
  for (int i = threadIdx.x; i < n; i += 256) { const float2 v = partial[i]; a += (double)v.x; c += (double)v.y; }
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); c += __shfl_down(c, o, 64); }
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = a; red[threadIdx.x >> 6][1] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a = red[0][0] + red[1][0] + red[2][0] + red[3][0]; c = red[0][1] + red[1][1] + red[2][1] + red[3][1];
    const double depth = 0.5 * a / ((double)(d.per_room ? 1 : d.B) * d.n_scales * d.n_dep * d.P * d.P);
    loss_out[0] = (float)(100.0 * depth + 100.0 * c); loss_out[1] = (float)depth; loss_out[2] = (float)c;
  }
}

// d loss / d image[b][c][y][x] = gscale * sum_s sum_{(oy, wy) in col_s(y)} sum_{(ox, wx) in col_s(x)} wy wx dpooled[b][s][c][oy][ox]
__global__ __launch_bounds__(256) void refine_bwd_kernel(const float* __restrict__ dpooled, const unsigned char* __restrict__ null, RefineDims d,
                                                         const int* __restrict__ col_ptr, const int* __restrict__ col_out,
                                                         const float* __restrict__ col_w, const float* __restrict__ gscale,
                                                         float* __restrict__ dimg) {
  const long plane = (long)d.S * d.S;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)d.B * d.C * plane) return;
  const int x = (int)(i % d.S), y = (int)((i / d.S) % d.S);
  const int c = (int)((i / plane) % d.C), b = (int)(i / (plane * d.C));
  const int nc = d.n_sem + d.n_dep, cc = c - d.sem0;
  float g = 0.f;
  if (cc >= 0 && cc < nc && !(c == d.dep0 + d.n_dep - 1 && null[(long)b * plane + (long)y * d.S + x])) {
    const long pp = (long)d.P * d.P;
    for (int s = 0; s < d.n_scales; ++s) {
      const int* cp = col_ptr + s * (d.S + 1);
      const int yb = cp[y], ye = cp[y + 1], xb = cp[x], xe = cp[x + 1];
      if (yb == ye || xb == xe) continue;
      const float* dp = dpooled + ((long)(b * d.n_scales + s) * nc + cc) * pp;
      for (int e = yb; e < ye; ++e) {
        const float wy = col_w[e];
        const float* row = dp + (long)col_out[e] * d.P;
        float r = 0.f;
        for (int f = xb; f < xe; ++f) r = fmaf(col_w[f], row[col_out[f]], r);
        g = fmaf(wy, r, g);
      }
    }
    g *= gscale[0];
  }
  dimg[i] = g;
}

// The same sum, separable, with one workgroup per (strip of ROWS image rows, channel):
//   stage 1  T[s][r][ox] = sum_{(oy, wy) in col_s(y0 + r)} wy * dpooled[s][c][oy][ox]      rows of dpooled, coalesced, into LDS
//   stage 2  g[r][x]     = sum_s sum_{(ox, wx) in col_s(x)} wx * T[s][r][ox]                 LDS reads only
// A lane owns one image column and keeps that column's tap lists (padded to MAXX entries per scale, weight 0) in registers
// for all rows of the strip.  The generic kernel above walks four CSR ranges per (pixel, channel) through dependent global
// loads (136 us per 256 x 256 x 70 gradient; 61 us with its inner loop padded, L1-issue bound); this one takes 41 us.
template <int MAXX, int ROWS, int PMAX>
__global__ __launch_bounds__(256) void refine_bwd_sep_kernel(const float* __restrict__ dpooled, const unsigned char* __restrict__ null,
                                                             RefineDims d, const int* __restrict__ col_ptr, const int* __restrict__ col_out,
                                                             const float* __restrict__ col_w, const float* __restrict__ gscale,
                                                             float* __restrict__ dimg, const int abl, const unsigned char* __restrict__ live) {
  __shared__ float T[MAX_SCALES][ROWS][PMAX];
  const int x = blockIdx.z * 256 + threadIdx.x;
  const int c = blockIdx.y % d.C, b = blockIdx.y / d.C;
  const int y0 = blockIdx.x * ROWS;
  const bool active = x < d.S;
  const int xs = active ? x : d.S - 1;
  const int nc = d.n_sem + d.n_dep, cc = c - d.sem0;
  float* out = dimg + ((long)(b * d.C + c) * d.S) * d.S;
  if (live != nullptr && cc >= 0 && cc < nc && !(live[b * d.C + c] & 2)) return;      // a plane nobody reads the gradient of (live_planes): left as it is
  if (cc < 0 || cc >= nc) {                                                            // channel 0
    if (active) for (int r = 0; r < ROWS && y0 + r < d.S; ++r) out[(long)(y0 + r) * d.S + x] = 0.f;
    return;
  }
  int ox[MAX_SCALES][MAXX]; float wx[MAX_SCALES][MAXX];
#pragma unroll
  for (int s = 0; s < MAX_SCALES; ++s) {
    const int sv = s < d.n_scales ? s : 0;
    const int xb = col_ptr[sv * (d.S + 1) + xs], xe = s < d.n_scales ? col_ptr[sv * (d.S + 1) + xs + 1] : xb;
#pragma unroll
    for (int f = 0; f < MAXX; ++f) {
      const bool v = xb + f < xe;
      ox[s][f] = v ? col_out[xb + f] : 0;
      wx[s][f] = v ? col_w[xb + f] : 0.f;
    }
  }
  // the strip's row lists, padded like the column lists, so that stage 1 issues MAXX independent loads per element
  __shared__ int yo[MAX_SCALES * ROWS][MAXX];
  __shared__ float yw[MAX_SCALES * ROWS][MAXX];
  for (int i = threadIdx.x; i < d.n_scales * ROWS * MAXX; i += 256) {
    const int e = i % MAXX, r = (i / MAXX) % ROWS, s = i / (MAXX * ROWS);
    const int y = min(y0 + r, d.S - 1);
    const int yb = col_ptr[s * (d.S + 1) + y], ye = col_ptr[s * (d.S + 1) + y + 1];
    const bool v = yb + e < ye && y0 + r < d.S;
    yo[s * ROWS + r][e] = v ? col_out[yb + e] : 0;
    yw[s * ROWS + r][e] = v ? col_w[yb + e] : 0.f;
  }
  __syncthreads();
  const long pp = (long)d.P * d.P;
  // four elements per pass, their 4 x MAXX loads issued together: one element at a time (MAXX loads in flight per thread, 24 dependent
  // round trips per workgroup) left the kernel latency-bound - 405 us for 16 rooms against ~130 us of LDS + L1 issue time
  {
    const int n1 = (abl & 1) ? 0 : d.n_scales * ROWS * d.P;
    constexpr int U = 4;
    for (int i0 = threadIdx.x; i0 < n1; i0 += 256 * U) {
      float v[U][MAXX]; float wgt[U][MAXX]; int si[U], ri[U], oi[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = min(i0 + 256 * u, n1 - 1);
        oi[u] = i % d.P; ri[u] = (i / d.P) % ROWS; si[u] = i / (d.P * ROWS);
        const float* dp = dpooled + ((long)(b * d.n_scales + si[u]) * nc + cc) * pp + oi[u];
#pragma unroll
        for (int e = 0; e < MAXX; ++e) { wgt[u][e] = yw[si[u] * ROWS + ri[u]][e]; v[u][e] = dp[(long)yo[si[u] * ROWS + ri[u]][e] * d.P]; }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (i0 + 256 * u < n1) {
          float t = 0.f;
#pragma unroll
          for (int e = 0; e < MAXX; ++e) t = fmaf(wgt[u][e], v[u][e], t);
          T[si[u]][ri[u]][oi[u]] = t;
        }
      }
    }
  }
  __syncthreads();
  if (!active) return;
  const float gs = gscale[0];
  const bool last = c == d.dep0 + d.n_dep - 1;
  for (int r = 0; r < ROWS && y0 + r < d.S; ++r) {
    float g = 0.f;
    if (!(abl & 2))
#pragma unroll
    for (int s = 0; s < MAX_SCALES; ++s)
#pragma unroll
      for (int f = 0; f < MAXX; ++f) g = fmaf(wx[s][f], T[s][r][ox[s][f]], g);
    if (last && null[(long)b * d.S * d.S + (long)(y0 + r) * d.S + x]) g = 0.f;
    out[(long)(y0 + r) * d.S + x] = g * gs;
  }
}

// Round 5: the same separable sum for NSTRIP strips of ROWS image rows per workgroup, rebuilt around what the kernel above is
// actually bound by - LDS instruction issue (SLN_RBWD_ABL: stage 1 = 235 of its 390 us for 16 rooms; per T element it issues 10
// LDS reads for the row's tap list next to 5 operand loads, per output pixel 20 single-word reads of T):
//   * the pooled-gradient rows the workgroup's image rows touch are staged in LDS once (coalesced; each pooled row is read ~1.3
//     times overall instead of by 13 strips: 2.2 GB through L2 before);
//   * stage 1 runs one (scale, image row) pair per WAVEFRONT pass: the pair's tap list is wave-uniform and comes through SCALAR
//     loads (no LDS traffic), the lanes walk the pooled columns: 5 LDS reads per T element instead of 15;
//   * stage 2: a column's taps are consecutive pooled columns (checked per lane when the list is loaded), so the five T values
//     of a scale are adjacent words: ds_read2 pairs, 12 LDS instructions per pixel instead of 20.
// Same products, same order of additions (absent taps are skipped instead of added with weight 0): bit-identical output
// (tools/lab/rbwd_compare.py).
template <int MAXX, int ROWS, int NSTRIP, int PMAX, int RMAX>
__global__ __launch_bounds__(256) void refine_bwd_sep2_kernel(const float* __restrict__ dpooled, const unsigned char* __restrict__ null,
                                                              RefineDims d, const int* __restrict__ col_ptr, const int* __restrict__ col_out,
                                                              const float* __restrict__ col_w, const float* __restrict__ gscale,
                                                              float* __restrict__ dimg) {
  static_assert(MAXX == 5, "the stage-2 read pattern is written for five taps");
  constexpr int WR = ROWS * NSTRIP;                              // image rows per workgroup
  constexpr int TP = PMAX + 4;                                   // T row: four words of slack behind the last pooled column
  extern __shared__ __attribute__((aligned(16))) float sm2[];
  float (*dP)[RMAX][PMAX] = reinterpret_cast<float (*)[RMAX][PMAX]>(sm2);                              // [MAX_SCALES]
  float (*T)[ROWS][TP] = reinterpret_cast<float (*)[ROWS][TP]>(sm2 + MAX_SCALES * RMAX * PMAX);         // [MAX_SCALES]
  __shared__ int lo_s[MAX_SCALES], n_s[MAX_SCALES];
  const int x = blockIdx.z * 256 + threadIdx.x;
  const int c = blockIdx.y % d.C, b = blockIdx.y / d.C;
  const int yw0 = blockIdx.x * WR;                               // first image row of the workgroup
  const bool active = x < d.S;
  const int xs = active ? x : d.S - 1;
  const int nc = d.n_sem + d.n_dep, cc = c - d.sem0;
  float* out = dimg + ((long)(b * d.C + c) * d.S) * d.S;
  if (cc < 0 || cc >= nc) {
    if (active) for (int r = 0; r < WR && yw0 + r < d.S; ++r) out[(long)(yw0 + r) * d.S + x] = 0.f;
    return;
  }
  int ox[MAX_SCALES][MAXX]; float wx[MAX_SCALES][MAXX];
  bool contig = true;
#pragma unroll
  for (int s = 0; s < MAX_SCALES; ++s) {
    const int sv = s < d.n_scales ? s : 0;
    const int xb = col_ptr[sv * (d.S + 1) + xs], xe = s < d.n_scales ? col_ptr[sv * (d.S + 1) + xs + 1] : xb;
#pragma unroll
    for (int f = 0; f < MAXX; ++f) {
      const bool v = xb + f < xe;
      ox[s][f] = v ? col_out[xb + f] : 0;
      wx[s][f] = v ? col_w[xb + f] : 0.f;
    }
#pragma unroll
    for (int f = 1; f < MAXX; ++f) contig = contig && (wx[s][f] == 0.f || ox[s][f] == ox[s][0] + f);
  }
  const int ylast = min(yw0 + WR, d.S) - 1;
  if (threadIdx.x < d.n_scales) {
    // pooled rows the workgroup's image rows touch (the lists are ascending): first entry of the first row, last entry of the last
    const int s = threadIdx.x;
    const int* cp = col_ptr + s * (d.S + 1);
    int lo = 1 << 30, hi = -1;
    const int e0 = cp[yw0], e1 = cp[ylast + 1];
    if (e1 > e0) { lo = col_out[e0]; hi = col_out[e1 - 1]; }
    lo_s[s] = hi >= lo ? lo : 0; n_s[s] = hi >= lo ? hi - lo + 1 : 0;
  }
  for (int i = threadIdx.x; i < MAX_SCALES * ROWS * 4; i += 256) T[i / (ROWS * 4)][(i / 4) % ROWS][PMAX + i % 4] = 0.f;     // the slack words: read with weight 0
  __syncthreads();
  const long pp = (long)d.P * d.P;
  bool fits = true;
  for (int s = 0; s < d.n_scales; ++s) fits = fits && n_s[s] <= RMAX;
  if (fits)
    for (int s = 0; s < d.n_scales; ++s) {
      const float* dp = dpooled + ((long)(b * d.n_scales + s) * nc + cc) * pp + (long)lo_s[s] * d.P;
      for (int i = threadIdx.x; i < n_s[s] * d.P; i += 256) dP[s][i / d.P][i % d.P] = dp[i];
    }
  const float gs = gscale[0];
  const bool last = c == d.dep0 + d.n_dep - 1;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int st = 0; st < NSTRIP; ++st) {
    const int y0 = yw0 + st * ROWS;
    if (y0 >= d.S) break;
    __syncthreads();                                            // dP staged (first strip) / T of the previous strip consumed
    // stage 1: wavefront w takes the pairs (scale, row) = w, w + 4, ...; the pair's taps are scalars
    for (int p = wave; p < d.n_scales * ROWS; p += 4) {
      const int s = p / ROWS, r = p % ROWS;
      const int y = min(y0 + r, d.S - 1);
      const int yb = col_ptr[s * (d.S + 1) + y];
      const int ne = (y0 + r < d.S) ? min(col_ptr[s * (d.S + 1) + y + 1] - yb, MAXX) : 0;
      int ro[MAXX]; float we[MAXX];
#pragma unroll
      for (int e = 0; e < MAXX; ++e) { const int q = yb + min(e, max(ne - 1, 0)); ro[e] = col_out[q]; we[e] = col_w[q]; }
      const float* dg = dpooled + ((long)(b * d.n_scales + s) * nc + cc) * pp;
      for (int o = lane; o < d.P; o += 64) {
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < MAXX; ++e)
          if (e < ne) t = fmaf(we[e], fits ? dP[s][ro[e] - lo_s[s]][o] : dg[(long)ro[e] * d.P + o], t);
        T[s][r][o] = t;
      }
    }
    __syncthreads();
    if (active)
      for (int r = 0; r < ROWS && y0 + r < d.S; ++r) {
        float g = 0.f;
        if (contig) {
#pragma unroll
          for (int s = 0; s < MAX_SCALES; ++s) {
            const float* tp = &T[s][r][ox[s][0]];               // five adjacent words (the slack covers ox + 4 > P - 1, weight 0 there)
            const float t0 = tp[0], t1 = tp[1], t2 = tp[2], t3 = tp[3], t4 = tp[4];
            g = fmaf(wx[s][0], t0, g); g = fmaf(wx[s][1], t1, g); g = fmaf(wx[s][2], t2, g); g = fmaf(wx[s][3], t3, g); g = fmaf(wx[s][4], t4, g);
          }
        } else {
#pragma unroll
          for (int s = 0; s < MAX_SCALES; ++s)
#pragma unroll
            for (int f = 0; f < MAXX; ++f) g = fmaf(wx[s][f], T[s][r][ox[s][f]], g);
        }
        if (last && null[(long)b * d.S * d.S + (long)(y0 + r) * d.S + x]) g = 0.f;
        out[(long)(y0 + r) * d.S + x] = g * gs;
      }
  }
}

}  // namespace

extern "C" {

int64_t sln_refine_loss_workspace_bytes(int B, int image_size, int pooled_size, int n_scales, int n_sem, int n_dep) {
  if (B <= 0 || image_size <= 0 || pooled_size <= 0 || n_scales <= 0 || n_scales > MAX_SCALES || n_sem <= 0 || n_dep <= 0) return SLN_E_BADARG;
  const int64_t pooled = (int64_t)B * n_scales * (n_sem + n_dep) * pooled_size * pooled_size * 4;
  const int64_t mask = ((int64_t)B * image_size * image_size + 255) / 256 * 256;
  const int64_t nblk = (((int64_t)B * n_scales * pooled_size * pooled_size + 127) / 128) * (1 + (n_dep + DCH - 1) / DCH);
  return pooled + mask + nblk * 8;                    // + one float2 partial per block of loss_kernel
}

static int carve(const SlnRefineLoss* L, void* workspace, float** pooled, unsigned char** mask, float2** partial) {
  char* p = (char*)workspace;
  const int64_t np = (int64_t)L->B * L->n_scales * (L->n_sem + L->n_dep) * L->pooled_size * L->pooled_size * 4;
  const int64_t nm = ((int64_t)L->B * L->image_size * L->image_size + 255) / 256 * 256;
  *pooled = (float*)p; *mask = (unsigned char*)(p + np); *partial = (float2*)(p + np + nm);
  return 0;
}

static int check(const SlnRefineLoss* L) {
  if (!L || L->B <= 0 || L->image_size <= 0 || L->pooled_size <= 0 || L->n_scales <= 0 || L->n_scales > MAX_SCALES) return SLN_E_BADARG;
  if (L->n_sem != 40) return SLN_E_UNSUPPORTED;      // NYU-40 one-hot block of the scene tensor (models/diff_render.py:3)
  if (L->sem0 < 0 || L->dep0 != L->sem0 + L->n_sem || L->dep0 + L->n_dep > L->channels || L->n_dep <= 0) return SLN_E_BADARG;
  if (!L->s2_k0 || !L->s2_k1 || !L->s2_l1 || !L->s1_i0 || !L->s1_i1 || !L->s1_l1 || !L->col_ptr || !L->col_out || !L->col_w) return SLN_E_BADARG;
  // per_room: a 128-row block of loss_kernel must not cover two rooms (checked here - init and every entry point - so that an
  // unsupported geometry fails before anything is launched or overwritten)
  if (L->per_room && ((long)L->n_scales * L->pooled_size * L->pooled_size) % 128 != 0) return SLN_E_UNSUPPORTED;
  return 0;
}

static RefineDims dims_of(const SlnRefineLoss* L) {
  RefineDims d;
  d.B = L->B; d.S = L->image_size; d.P = L->pooled_size; d.C = L->channels; d.sem0 = L->sem0; d.n_sem = L->n_sem; d.dep0 = L->dep0;
  d.n_dep = L->n_dep; d.n_scales = L->n_scales; d.pmax = L->stage1_stride;
  d.per_room = L->per_room != 0; d.pad_ = 0;
  return d;
}

int sln_refine_loss_init(const SlnRefineLoss* L, void* workspace, void* stream) {
  int r = check(L);
  if (r) return r;
  if (!workspace) return SLN_E_BADARG;
  float* pooled; unsigned char* mask; float2* partial;
  carve(L, workspace, &pooled, &mask, &partial);
  (void)pooled; (void)mask; (void)partial; (void)stream;       // nothing to arm: every call rewrites what it reads
  return 0;
}

// live_planes is honoured only where every kernel of the forward / backward pair knows about it (the LDS pooling kernel and the
// separable backward kernel: the shapes of the refinement loop); elsewhere every plane is processed
static const unsigned char* live_of(const SlnRefineLoss* L, const RefineDims& d) {
  static const bool lab = std::getenv("SLN_POOL_NO_LDS") != nullptr || std::getenv("SLN_RBWD_NEW") != nullptr;
  const bool ok = !lab && d.pmax <= POOL_LDS_MAX && d.P <= POOL_LDS_MAX && L->max_col_entries > 0 && L->max_col_entries <= 5 && d.n_sem == 40;
  return ok ? L->live_planes : nullptr;
}

static void launch_pool(const SlnRefineLoss* L, const RefineDims& d, const float* image, int null_fill, unsigned char* mask, float* pooled,
                        const unsigned char* live, hipStream_t st) {
  const long npix = (long)d.B * d.S * d.S;
  if (null_fill == 1) hipLaunchKernelGGL(null_mask_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, image, d, mask, live);      // 2: `mask` is given
  static const bool no_lds = std::getenv("SLN_POOL_NO_LDS") != nullptr;      // lab: the per-pixel kernel
  // the scales' workgroups differ 9x in work (intermediate images of 32^2 .. 96^2 pixels) and workgroups are dispatched in block order:
  // the largest scale first (round 6: 76 -> 62 us with 16 rooms in flight; SLN_POOL_SCALE_ORDER=0 restores the ascending order)
  static const int scale_rev = [] { const char* e = std::getenv("SLN_POOL_SCALE_ORDER"); return e != nullptr && e[0] == '0' ? 0 : 1; }();
  if (!no_lds && d.pmax <= POOL_LDS_MAX && d.P <= POOL_LDS_MAX) {
    hipLaunchKernelGGL(pool_lds_kernel, dim3(d.B * (d.n_sem + d.n_dep), d.n_scales), dim3(256), 0, st, image, mask, null_fill, d, L->s2_k0, L->s2_k1,
                       L->s2_l1, L->s1_i0, L->s1_i1, L->s1_l1, pooled, live, (live != nullptr && L->pooled_ones != nullptr) ? 1 : 0, scale_rev);
    return;
  }
  const long np = (long)d.B * d.n_scales * sln_cdiv(d.n_sem + d.n_dep, CG) * d.P * d.P;
  hipLaunchKernelGGL(pool_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, st, image, mask, null_fill, d, L->s2_k0, L->s2_k1, L->s2_l1,
                     L->s1_i0, L->s1_i1, L->s1_l1, pooled);
}

int sln_refine_pool(const SlnRefineLoss* L, const float* image, int null_fill, void* workspace, float* pooled_out, void* stream) {
  int r = check(L);
  if (r) return r;
  if (!image || !workspace || !pooled_out) return SLN_E_BADARG;
  const RefineDims d = dims_of(L);
  float* pooled; unsigned char* mask; float2* partial;
  carve(L, workspace, &pooled, &mask, &partial);
  launch_pool(L, d, image, null_fill, mask, pooled_out, nullptr, (hipStream_t)stream);      // any image: no plane is known dead
  SLN_CHECK_LAUNCH();
  return 0;
}

// 1 when live_planes (and with it null_mask / pooled_ones) of this descriptor would be honoured - the shapes the LDS pooling kernel
// and the separable backward kernel take; 0 when every plane is processed regardless.  A producer that leaves dead planes of the
// image unwritten (sln_scene_forward_live) must not be paired with a loss that reads them: ask first.
int sln_refine_loss_live_ok(const SlnRefineLoss* L) {
  if (check(L)) return 0;
  SlnRefineLoss probe = *L;
  static const unsigned char one = 1;
  probe.live_planes = &one;                       // (only compared against NULL)
  return live_of(&probe, dims_of(&probe)) != nullptr ? 1 : 0;
}

int sln_refine_loss_forward(const SlnRefineLoss* L, const float* image, const float* target_depth_pooled, const int32_t* labels,
                            const float* inv_count, void* workspace, float* loss_out, void* stream) {
  int r = check(L);
  if (r) return r;
  if (!image || !target_depth_pooled || !labels || !inv_count || !workspace || !loss_out) return SLN_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const RefineDims d = dims_of(L);
  float* pooled; unsigned char* mask; float2* partial;
  carve(L, workspace, &pooled, &mask, &partial);
  const unsigned char* live = live_of(L, d);
  // the null mask as the producer of `image` computed it (SlnRefineLoss::null_mask: the scene pass's compose kernel sums the 29
  // depth-hot values of a pixel in this kernel's order while it has them in registers): null_mask_kernel's launch is skipped
  const bool ext_mask = live != nullptr && L->null_mask != nullptr;
  if (ext_mask) mask = const_cast<unsigned char*>(L->null_mask);
  launch_pool(L, d, image, ext_mask ? 2 : 1, mask, pooled, live, st);
  const long nl = (long)d.B * d.n_scales * d.P * d.P;
  const dim3 lg((unsigned)((nl + 127) / 128), 1 + sln_cdiv(d.n_dep, DCH));
  hipLaunchKernelGGL((loss_kernel<40>), lg, dim3(128), 0, st, pooled, d, target_depth_pooled, labels, inv_count, partial, live, L->pooled_ones);
  const long per_room_rows = (long)d.n_scales * d.P * d.P;
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(d.per_room ? d.B : 1), dim3(256), 0, st, partial, (int)(lg.x * lg.y), d, loss_out, (int)lg.x,
                     (int)(per_room_rows / 128));
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_refine_loss_backward(const SlnRefineLoss* L, const void* workspace, const float* grad_scale, float* grad_image, void* stream) {
  int r = check(L);
  if (r) return r;
  if (!workspace || !grad_scale || !grad_image) return SLN_E_BADARG;
  const RefineDims d = dims_of(L);
  float* pooled; unsigned char* mask; float2* partial;
  carve(L, const_cast<void*>(workspace), &pooled, &mask, &partial);
  const unsigned char* live = live_of(L, d);
  if (live != nullptr && L->null_mask != nullptr) mask = const_cast<unsigned char*>(L->null_mask);
  static const int abl = std::getenv("SLN_RBWD_ABL") ? std::atoi(std::getenv("SLN_RBWD_ABL")) : 0;      // lab: 1 no stage-1 loads, 2 no stage-2 sums
  static const bool new_sep = std::getenv("SLN_RBWD_NEW") != nullptr;          // lab: the LDS-staged multi-strip kernel (slower so far, see LAB_NOTES)
  static const int rows8 = std::getenv("SLN_RBWD_ROWS") ? std::atoi(std::getenv("SLN_RBWD_ROWS")) : 16;
  if (!new_sep && rows8 == 8 && L->max_col_entries > 0 && L->max_col_entries <= 5 && d.P <= 96) {
    hipLaunchKernelGGL((refine_bwd_sep_kernel<5, 8, 96>), dim3(sln_cdiv(d.S, 8), d.B * d.C, sln_cdiv(d.S, 256)), dim3(256), 0,
                       (hipStream_t)stream, pooled, mask, d, L->col_ptr, L->col_out, L->col_w, grad_scale, grad_image, abl, live);
    SLN_CHECK_LAUNCH();
    return 0;
  }
  if (!new_sep && rows8 == 32 && L->max_col_entries > 0 && L->max_col_entries <= 5 && d.P <= 96) {
    hipLaunchKernelGGL((refine_bwd_sep_kernel<5, 32, 96>), dim3(sln_cdiv(d.S, 32), d.B * d.C, sln_cdiv(d.S, 256)), dim3(256), 0,
                       (hipStream_t)stream, pooled, mask, d, L->col_ptr, L->col_out, L->col_w, grad_scale, grad_image, abl, live);
    SLN_CHECK_LAUNCH();
    return 0;
  }
  if (new_sep && abl == 0 && L->max_col_entries > 0 && L->max_col_entries <= 5 && d.P <= 96 && d.n_scales <= MAX_SCALES) {
    constexpr int ROWS = 8, NSTRIP = 4, RMAX = 20;
    constexpr size_t smem = sizeof(float) * (MAX_SCALES * RMAX * 96 + MAX_SCALES * ROWS * (96 + 4));
    hipLaunchKernelGGL((refine_bwd_sep2_kernel<5, ROWS, NSTRIP, 96, RMAX>), dim3(sln_cdiv(d.S, ROWS * NSTRIP), d.B * d.C, sln_cdiv(d.S, 256)), dim3(256), smem,
                       (hipStream_t)stream, pooled, mask, d, L->col_ptr, L->col_out, L->col_w, grad_scale, grad_image);
    SLN_CHECK_LAUNCH();
    return 0;
  }
  if (L->max_col_entries > 0 && L->max_col_entries <= 5 && d.P <= 96) {
    constexpr int ROWS = 16;
    hipLaunchKernelGGL((refine_bwd_sep_kernel<5, ROWS, 96>), dim3(sln_cdiv(d.S, ROWS), d.B * d.C, sln_cdiv(d.S, 256)), dim3(256), 0,
                       (hipStream_t)stream, pooled, mask, d, L->col_ptr, L->col_out, L->col_w, grad_scale, grad_image, abl, live);
  } else {
    const long n = (long)d.B * d.C * d.S * d.S;
    hipLaunchKernelGGL(refine_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pooled, mask, d, L->col_ptr,
                       L->col_out, L->col_w, grad_scale, grad_image);
  }
  SLN_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
