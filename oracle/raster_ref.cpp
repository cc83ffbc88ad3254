// CPU restatement of the rasterizer the reference borrows from the third-party `neural_renderer`
// package (github daniilidis-group/neural_renderer, PyTorch port of Kato et al.'s Neural 3D Mesh
// Renderer).  TEST INFRASTRUCTURE ONLY: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
// may load it; nothing under 3d_sln_amd/ does.
//
// PARITY UNPINNED.  The package is neither vendored in /root/reference nor pinned to a version
// (README.md:12-18 links the repository without tag/commit and asks for a manual edit of its
// projection.py); it is not installed in the build image and has no source on disk.  The reference's own
// call sites are models/misc.py:7 (import), models/diff_render.py:359-361 (Renderer construction),
// :366 (mode='depth') and :398 (mode="rgb"); the reference holds no test or golden vector at that boundary.
// This file restates the package's published algorithm and thereby DEFINES the
// semantics the HIP kernels are held to (bit-exact face indices; everything else within fp32 rounding):
//   forward  : per-face back-face test + inverse of the pixel-space [x y 1] matrix, then per pixel a scan
//              over ALL faces in ascending index with three edge tests, clamped/renormalised barycentrics,
//              perspective depth 1/sum(w/z), near/far rejection and a strict '<' z-test (lowest index wins ties);
//   sampling : trilinear fetch in the per-face ts^3 texture cube at w_k*(ts-1)*depth/z_k;
//   backward : depth  - d zp/d z_k = w_k zp^2/z_k^2 and d zp/d(x,y)_k through the stored inverse matrix;
//              rgb    - the hand-designed "pixel map" gradient: per front-facing face, per edge, per axis,
//                       walk the pixels the edge crosses and scan outward / inward accumulating
//                       -(I_pixel - I_inside|outside).grad / signed_distance where that product is positive.
// Expressions keep the published code's mixed float/double arithmetic (double literals promote), because it
// decides on which side of an edge a pixel centre falls.  Build with -ffp-contract=off.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

inline bool backfacing(const float* f) {
  return (f[7] - f[1]) * (f[3] - f[0]) < (f[4] - f[1]) * (f[6] - f[0]);
}

// pixel-space inverse of [[x0 x1 x2],[y0 y1 y2],[1 1 1]] arranged so that w_k = inv[3k]*xi + inv[3k+1]*yi + inv[3k+2]
inline void face_inverse(const float* f, int is, float* inv) {
  float p[3][2];
  for (int n = 0; n < 3; ++n)
    for (int d = 0; d < 2; ++d) p[n][d] = (float)(0.5 * (double)(f[3 * n + d] * is + is - 1));
  float m[9] = {p[1][1] - p[2][1], p[2][0] - p[1][0], p[1][0] * p[2][1] - p[2][0] * p[1][1],
                p[2][1] - p[0][1], p[0][0] - p[2][0], p[2][0] * p[0][1] - p[0][0] * p[2][1],
                p[0][1] - p[1][1], p[1][0] - p[0][0], p[0][0] * p[1][1] - p[1][0] * p[0][1]};
  const float den = p[2][0] * (p[0][1] - p[1][1]) + p[0][0] * (p[1][1] - p[2][1]) + p[1][0] * (p[2][1] - p[0][1]);
  for (int k = 0; k < 9; ++k) inv[k] = m[k] / den;
}

}  // namespace

extern "C" {

// faces [B,F,9] = (x,y,z) of the three projected vertices (x,y in NDC, z camera depth).
// Outputs (row 0 = bottom row; the package flips rows afterwards): face_index [B,is,is] (-1 = none),
// weight [B,is,is,3], depth [B,is,is] (far where empty).
void nmr_forward(const float* faces, int B, int F, int is, float near, float far, int32_t* face_index, float* weight,
                 float* depth) {
  std::vector<float> inv((size_t)B * F * 9, 0.f);
  std::vector<uint8_t> back((size_t)B * F);
  for (long i = 0; i < (long)B * F; ++i) {
    back[i] = backfacing(faces + 9 * i);
    if (!back[i]) face_inverse(faces + 9 * i, is, inv.data() + 9 * i);
  }
#pragma omp parallel for schedule(dynamic, 256)
  for (long i = 0; i < (long)B * is * is; ++i) {
    const int b = (int)(i / ((long)is * is)), pn = (int)(i % ((long)is * is)), yi = pn / is, xi = pn % is;
    const float yp = (float)((2. * yi + 1 - is) / is), xp = (float)((2. * xi + 1 - is) / is);
    float zmin = far; int best = -1; float wb[3] = {0.f, 0.f, 0.f};
    for (int fn = 0; fn < F; ++fn) {
      const long fi = (long)b * F + fn;
      if (back[fi]) continue;
      const float* f = faces + 9 * fi;
      if (((yp - f[1]) * (f[3] - f[0]) < (xp - f[0]) * (f[4] - f[1])) ||
          ((yp - f[4]) * (f[6] - f[3]) < (xp - f[3]) * (f[7] - f[4])) ||
          ((yp - f[7]) * (f[0] - f[6]) < (xp - f[6]) * (f[1] - f[7]))) continue;
      const float* iv = inv.data() + 9 * fi;
      float w[3], ws = 0.f;
      for (int k = 0; k < 3; ++k) {
        w[k] = iv[3 * k] * xi + iv[3 * k + 1] * yi + iv[3 * k + 2];
        w[k] = std::min(std::max(w[k], 0.f), 1.f);
        ws += w[k];
      }
      for (int k = 0; k < 3; ++k) w[k] /= ws;
      const float zp = (float)(1. / (double)(w[0] / f[2] + w[1] / f[5] + w[2] / f[8]));
      if (zp <= near || far <= zp) continue;
      if (zp < zmin) { zmin = zp; best = fn; wb[0] = w[0]; wb[1] = w[1]; wb[2] = w[2]; }
    }
    face_index[i] = best;
    depth[i] = best >= 0 ? zmin : far;
    for (int k = 0; k < 3; ++k) weight[3 * i + k] = wb[k];
  }
}

// Trilinear texture fetch: textures [B,F,ts,ts,ts,3]; rgb [B,is,is,3] (background 0).
void nmr_texture_sample(const float* faces, const float* textures, const int32_t* face_index, const float* weight,
                        const float* depth, int B, int F, int is, int ts, float eps, float* rgb) {
#pragma omp parallel for
  for (long i = 0; i < (long)B * is * is; ++i) {
    const int fn = face_index[i];
    float px[3] = {0.f, 0.f, 0.f};
    if (fn >= 0) {
      const int b = (int)(i / ((long)is * is));
      const float* f = faces + 9 * ((long)b * F + fn);
      const float* tx = textures + (size_t)((long)b * F + fn) * ts * ts * ts * 3;
      float t[3];
      for (int k = 0; k < 3; ++k) {
        float v = weight[3 * i + k] * (ts - 1) * (depth[i] / f[3 * k + 2]);
        v = std::max(v, 0.f);
        v = std::min(v, (float)(ts - 1) - eps);
        t[k] = v;
      }
      for (int c = 0; c < 8; ++c) {
        float w = 1.f; int ti[3];
        for (int k = 0; k < 3; ++k) {
          const float fr = t[k] - (float)(int)t[k];
          if (((c >> k) & 1) == 0) { w *= 1.f - fr; ti[k] = (int)t[k]; }
          else { w *= fr; ti[k] = (int)t[k] + 1; }
        }
        const int cell = ti[0] * ts * ts + ti[1] * ts + ti[2];
        for (int k = 0; k < 3; ++k) px[k] += w * tx[3 * cell + k];
      }
    }
    for (int k = 0; k < 3; ++k) rgb[3 * i + k] = px[k];
  }
}

// grad_faces [B,F,9] += d(loss)/d(faces) through the depth map.
void nmr_backward_depth(const float* faces, const int32_t* face_index, const float* weight, const float* depth,
                        const float* grad_depth, int B, int F, int is, float* grad_faces) {
  for (long i = 0; i < (long)B * is * is; ++i) {        // serial: deterministic accumulation order
    const int fn = face_index[i];
    if (fn < 0) continue;
    const int b = (int)(i / ((long)is * is));
    const float* f = faces + 9 * ((long)b * F + fn);
    float* gf = grad_faces + 9 * ((long)b * F + fn);
    float iv[9];
    face_inverse(f, is, iv);
    const float d2 = depth[i] * depth[i], g = grad_depth[i];
    for (int k = 0; k < 3; ++k) gf[3 * k + 2] += g * weight[3 * i + k] * d2 / (f[3 * k + 2] * f[3 * k + 2]);
    float tmp[2] = {0.f, 0.f};
    for (int l = 0; l < 2; ++l)
      for (int m = 0; m < 3; ++m) tmp[l] += -iv[3 * m + l] / f[3 * m + 2];
    for (int k = 0; k < 3; ++k)
      for (int l = 0; l < 2; ++l) gf[3 * k + l] += -g * tmp[l] * weight[3 * i + k] * d2 * is / 2;
  }
}

// Pixel-map gradient of a C-channel image (rgb [B,is,is,C], grad_rgb same shape) w.r.t. the x,y of the faces.
// One positive-part test per pixel pair over the SUM of the C channels (C=3 reproduces the package's rgb mode).
void nmr_backward_pixel_map(const float* faces, const int32_t* face_index, const float* rgb, const float* grad_rgb,
                            int B, int F, int is, int C, float eps, float* grad_faces) {
#pragma omp parallel for schedule(dynamic, 16)
  for (long i = 0; i < (long)B * F; ++i) {
    const int b = (int)(i / F), fn = (int)(i % F);
    const float* face = faces + 9 * i;
    float gface[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (backfacing(face)) continue;
    const long base = (long)b * is * is;
    for (int e = 0; e < 3; ++e) {
      int pi[3]; float pp[3][2];
      for (int n = 0; n < 3; ++n) pi[n] = (e + n) % 3;
      for (int n = 0; n < 3; ++n)
        for (int d = 0; d < 2; ++d) pp[n][d] = (float)(0.5 * (double)(face[3 * pi[n] + d] * is + is - 1));
      for (int axis = 0; axis < 2; ++axis) {
        float p[3][2];
        for (int n = 0; n < 3; ++n)
          for (int d = 0; d < 2; ++d) p[n][d] = pp[n][(d + axis) % 2];
        const int dir = (axis == 0) ? (p[0][0] < p[1][0] ? -1 : 1) : (p[0][0] < p[1][0] ? 1 : -1);
        const int d0_from = (int)std::max(std::ceil(std::min(p[0][0], p[1][0])), 0.f);
        const int d0_to = (int)std::min(std::max(p[0][0], p[1][0]), (float)(is - 1));
        for (int d0 = d0_from; d0 <= d0_to; ++d0) {
          const float d1_cross = (p[1][1] - p[0][1]) / (p[1][0] - p[0][0]) * (d0 - p[0][0]) + p[0][1];
          const int d1_in = dir > 0 ? (int)std::floor(d1_cross) : (int)std::ceil(d1_cross);
          const int d1_out = d1_in + dir;
          if (d1_in < 0 || is <= d1_in || d1_out < 0 || is <= d1_out) continue;
          const long idx_in = axis == 0 ? base + (long)d1_in * is + d0 : base + (long)d0 * is + d1_in;
          const long idx_out = axis == 0 ? base + (long)d1_out * is + d0 : base + (long)d0 * is + d1_out;
          const long step = axis == 0 ? is : 1;
          auto accumulate = [&](int d1, float diff) {
            if (p[1][0] != d0) {
              float dist = (float)((double)((p[1][0] - p[0][0]) / (p[1][0] - d0) * (d1 - d1_cross)) * 2. / is);
              dist = 0 < dist ? dist + eps : dist - eps;
              gface[pi[0] * 3 + (1 - axis)] -= diff / dist;
            }
            if (p[0][0] != d0) {
              float dist = (float)((double)((p[1][0] - p[0][0]) / (d0 - p[0][0]) * (d1 - d1_cross)) * 2. / is);
              dist = 0 < dist ? dist + eps : dist - eps;
              gface[pi[1] * 3 + (1 - axis)] -= diff / dist;
            }
          };
          // outward scan: from the first outside pixel to the image border
          if (face_index[idx_in] == fn) {
            const int lim = dir > 0 ? is - 1 : 0;
            const int from = std::max(std::min(d1_out, lim), 0), to = std::min(std::max(d1_out, lim), is - 1);
            long q = axis == 0 ? base + (long)from * is + d0 : base + (long)d0 * is + from;
            for (int d1 = from; d1 <= to; ++d1, q += step) {
              float diff = 0.f;
              for (int k = 0; k < C; ++k) diff += (rgb[q * C + k] - rgb[idx_in * C + k]) * grad_rgb[q * C + k];
              if (diff <= 0) continue;
              accumulate(d1, diff);
            }
          }
          // inward scan: from the first inside pixel to the opposite edge, only over this face's pixels
          {
            float cross2;
            if ((d0 - p[0][0]) * (d0 - p[2][0]) < 0) cross2 = (p[2][1] - p[0][1]) / (p[2][0] - p[0][0]) * (d0 - p[0][0]) + p[0][1];
            else cross2 = (p[1][1] - p[2][1]) / (p[1][0] - p[2][0]) * (d0 - p[2][0]) + p[2][1];
            const int lim = dir > 0 ? (int)std::ceil(cross2) : (int)std::floor(cross2);
            const int from = std::max(std::min(d1_in, lim), 0), to = std::min(std::max(d1_in, lim), is - 1);
            long q = axis == 0 ? base + (long)from * is + d0 : base + (long)d0 * is + from;
            for (int d1 = from; d1 <= to; ++d1, q += step) {
              if (face_index[q] != fn) continue;
              float diff = 0.f;
              for (int k = 0; k < C; ++k) diff += (rgb[q * C + k] - rgb[idx_out * C + k]) * grad_rgb[q * C + k];
              if (diff <= 0) continue;
              accumulate(d1, diff);
            }
          }
        }
      }
    }
    for (int k = 0; k < 9; ++k) grad_faces[9 * i + k] += gface[k];
  }
}

}  // extern "C"
