// Object placement of mesh_render_func (models/diff_render.py:76-159) fused with the camera projection, the near-plane cull
// (:346-356) and fill_back, forward and backward, for ONE room whose meshes are resident on the device:
//
//   bmin, bmax = box[:3] * room, box[3:] * room;  centre = (bmax + bmin) / 2;  size = bmax - bmin
//   theta = -angle * 2 pi / 24;  scale = min_j(size_j / model_size_j);  A = scale * R_y(theta)
//   v' = A (v - model_centre) + centre                                      [evaluated as A v + (centre - A model_centre)]
//   face corner -> camera (R v' + t) -> (x_ndc, y_ndc, z_cam) as sln_project_faces; a face with a corner closer than
//   cull_eps is degenerated to a point (it never covers a pixel; shapes stay fixed); faces F..2F-1 are the same faces with
//   corners (2, 1, 0);  size_loss = sum_k mean_j (size_kj - target_kj)^2 (:98-100,160-165).
//
// In torch this is ~50 small launches forward and ~55 backward per refinement iteration (stack / cat / matmul / index /
// where / flip and their autograd nodes), a third of the iteration.  Here: one kernel each way.  Backward runs one workgroup
// per object over that object's faces (the face list is grouped by object), reduces d centre (3) and d A (3x3) in
// registers / LDS and applies the chain rule to the box row and the angle - no atomics, deterministic.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <hip/hip_runtime.h>

#include <cstdint>

#include "sln_common.h"
#include "sln_hip.h"

namespace {

struct ObjXform { float A[9]; float tr[3]; float scale, c, s; int jmin; float size[3]; };

__device__ __forceinline__ ObjXform object_xform(const SlnPlacement& P, const float* __restrict__ boxes, const float* __restrict__ angles, int k) {
  ObjXform x;
  const int row = P.vis[k];
  const float* b = boxes + 6 * row;
  float centre[3];
  x.scale = 0.f; x.jmin = 0;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float lo = b[j] * P.ext[j], hi = b[3 + j] * P.ext[j];
    centre[j] = (hi + lo) / 2.f;
    x.size[j] = hi - lo;
    const float r = x.size[j] / P.msize[3 * k + j];
    if (j == 0 || r < x.scale) { x.scale = r; x.jmin = j; }          // torch.min: first minimum wins
  }
  const float theta = -angles[row] * (float)(2.0 * 3.14159265358979323846 / 24.0);
  x.c = cosf(theta); x.s = sinf(theta);
  // R_y = [[c, 0, s], [0, 1, 0], [-s, 0, c]]
  x.A[0] = x.scale * x.c; x.A[1] = 0.f;     x.A[2] = x.scale * x.s;
  x.A[3] = 0.f;           x.A[4] = x.scale; x.A[5] = 0.f;
  x.A[6] = -x.scale * x.s; x.A[7] = 0.f;    x.A[8] = x.scale * x.c;
  const float* mc = P.mcenter + 3 * k;
#pragma unroll
  for (int j = 0; j < 3; ++j) x.tr[j] = centre[j] - (x.A[3 * j] * mc[0] + x.A[3 * j + 1] * mc[1] + x.A[3 * j + 2] * mc[2]);
  return x;
}

struct Corner { float u, v, z; float x, y; };      // ndc + camera-space coordinates

__device__ __forceinline__ Corner project(const SlnPlacement& P, const float p[3]) {
  Corner c;
  c.x = p[0] * P.R[0] + p[1] * P.R[1] + p[2] * P.R[2] + P.t[0];
  c.y = p[0] * P.R[3] + p[1] * P.R[4] + p[2] * P.R[5] + P.t[1];
  c.z = p[0] * P.R[6] + p[1] * P.R[7] + p[2] * P.R[8] + P.t[2];
  const float os = P.orig_size;
  const float xh = c.x / (c.z + P.proj_eps), yh = c.y / (c.z + P.proj_eps);
  float u = xh * P.K[0] + yh * P.K[1] + P.K[2];
  float v = os - (xh * P.K[3] + yh * P.K[4] + P.K[5]);
  c.u = 2.f * (u - os / 2.f) / os;
  c.v = 2.f * (v - os / 2.f) / os;
  return c;
}

__device__ __forceinline__ int object_of_face(const SlnPlacement& P, int f) {
  int k = 0;                                        // n_vis is a dozen: linear search over the range table
  while (k < P.n_vis && f >= P.obj_face_ptr[k + 1]) ++k;
  return k;                                         // == n_vis: room shell
}

__device__ __forceinline__ void place_forward_body(const SlnPlacement& P, const float* __restrict__ boxes, const float* __restrict__ angles,
                                                   const float* __restrict__ size_target, float* __restrict__ fxyz,
                                                   float* __restrict__ sizes, float* __restrict__ size_loss) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.x == 0 && threadIdx.x < 64) {        // sizes and the size loss: one wavefront
    float l = 0.f;
    for (int k = threadIdx.x; k < P.n_vis; k += 64) {
      const ObjXform x = object_xform(P, boxes, angles, k);
      float m = 0.f;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        sizes[3 * k + j] = x.size[j];
        if (size_target) { const float d = x.size[j] - size_target[3 * k + j]; m += d * d; }
      }
      l += m / 3.f;
    }
    for (int o = 32; o > 0; o >>= 1) l += __shfl_down(l, o, 64);
    if (threadIdx.x == 0) size_loss[0] = l;
  }
  if (f >= P.F) return;
  const int k = object_of_face(P, f);
  ObjXform x;
  if (k < P.n_vis) x = object_xform(P, boxes, angles, k);
  Corner c[3];
  bool cull = false;
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const int vi = P.faces[3 * f + q];
    float p[3];
    if (k < P.n_vis) {
      const float* m = P.model_v + 3L * vi;           // vi indexes the flattened [n_vis * Vm] model table
#pragma unroll
      for (int j = 0; j < 3; ++j) p[j] = x.A[3 * j] * m[0] + x.A[3 * j + 1] * m[1] + x.A[3 * j + 2] * m[2] + x.tr[j];
    } else {
      const float* m = P.shell_v + 3L * (vi - P.n_vis * P.Vm);
      p[0] = m[0]; p[1] = m[1]; p[2] = m[2];
    }
    c[q] = project(P, p);
    cull = cull || c[q].z < P.cull_eps;
  }
  float* o0 = fxyz + 9L * f;
  float* o1 = fxyz + 9L * (P.F + f);
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const float u = cull ? 0.f : c[q].u, v = cull ? 0.f : c[q].v, z = cull ? 0.f : c[q].z;
    o0[3 * q] = u; o0[3 * q + 1] = v; o0[3 * q + 2] = z;
    o1[3 * (2 - q)] = u; o1[3 * (2 - q) + 1] = v; o1[3 * (2 - q) + 2] = z;
  }
}

__global__ __launch_bounds__(256) void place_forward_kernel(SlnPlacement P, const float* __restrict__ boxes, const float* __restrict__ angles,
                                                            const float* __restrict__ size_target, float* __restrict__ fxyz,
                                                            float* __restrict__ sizes, float* __restrict__ size_loss) {
  place_forward_body(P, boxes, angles, size_target, fxyz, sizes, size_loss);
}
// (defined with the stand-alone head kernels below)
__device__ __forceinline__ void head_forward_row(const int i, const int room, const bool lastrow, const int na, const float* __restrict__ boxes_pred,
                                                 const float* __restrict__ angles_pred, const float* __restrict__ noise,
                                                 const float* __restrict__ box_last, const float* __restrict__ angle_last, const float beta,
                                                 float* __restrict__ box_out, float* __restrict__ idx_out);
__device__ __forceinline__ void head_backward_row(const int i, const bool lastrow, const int na, const float* __restrict__ angles_pred,
                                                  const float* __restrict__ g_box_row, const float g_idx_row, const float beta,
                                                  float* __restrict__ g_boxes_pred, float* __restrict__ g_angles_pred, const int ld_gb);
// With the head fields of a room set (SlnPlacementRoom::boxes_pred), the glue between the decoder and the placement runs inside the
// two placement launches (round 5: two launches less per iteration): every workgroup derives the room's [n, 6] boxes and [n] angle
// indices from the decoder's outputs into LDS - a dozen rows, the stand-alone head kernel's own arithmetic - and places its faces
// from there; workgroup 0 also stores them (backward and the caller read them).
constexpr int HEAD_ROWS = 128;
__global__ __launch_bounds__(256) void place_forward_rooms_kernel(const SlnPlacementRoom* __restrict__ rooms) {
  const SlnPlacementRoom& r = rooms[blockIdx.y];
  if ((int)blockIdx.x * 256 >= r.P.F && blockIdx.x != 0) return;
  __shared__ float s_box[HEAD_ROWS * 6];
  __shared__ float s_idx[HEAD_ROWS];
  const float* boxes = r.boxes; const float* angles = r.angles;
  if (r.boxes_pred != nullptr) {
    const int n = r.P.n;
    if (n > HEAD_ROWS) __builtin_trap();                       // the host checks this before it sets the head fields
    // the noise of iteration k = noise_step[0] (a device counter, advanced by the backward launch): row k of a [iterations, stride] table
    const float* nz = r.noise != nullptr && r.noise_step != nullptr ? r.noise + (long)r.noise_step[0] * r.noise_stride : r.noise;
    for (int i = threadIdx.x; i < n; i += 256) {
      head_forward_row(i, 0, i == n - 1, r.n_angle, r.boxes_pred, r.angles_pred, nz, r.box_last, r.angle_last, r.beta, s_box + 6 * i, s_idx + i);
      if (blockIdx.x == 0) {
        for (int j = 0; j < 6; ++j) const_cast<float*>(r.boxes)[6 * i + j] = s_box[6 * i + j];
        const_cast<float*>(r.angles)[i] = s_idx[i];
      }
    }
    __syncthreads();
    boxes = s_box; angles = s_idx;
  }
  place_forward_body(r.P, boxes, angles, r.size_target, r.faces_out, r.sizes, r.size_loss);
}

// one workgroup per row of `boxes`; rows without a visible object get zero gradients
__device__ __forceinline__ void place_backward_body(const SlnPlacement& P, const float* __restrict__ boxes, const float* __restrict__ angles,
                                                    const float* __restrict__ size_target, const float* __restrict__ gf,
                                                    const float* __restrict__ g_size_loss, float* __restrict__ g_boxes,
                                                    float* __restrict__ g_angles) {
  const int row = blockIdx.x;
  int k = -1;
  for (int q = 0; q < P.n_vis; ++q) k = P.vis[q] == row ? q : k;
  if (k < 0) {
    if (threadIdx.x < 6) g_boxes[6 * row + threadIdx.x] = 0.f;
    if (threadIdx.x == 6) g_angles[row] = 0.f;
    return;
  }
  const ObjXform x = object_xform(P, boxes, angles, k);
  float acc[12];                                     // d centre (3), d A (3x3, row-major)
#pragma unroll
  for (int j = 0; j < 12; ++j) acc[j] = 0.f;
  const float* mc = P.mcenter + 3 * k;
  const float s2 = 2.f / P.orig_size;
  for (int f = P.obj_face_ptr[k] + threadIdx.x; f < P.obj_face_ptr[k + 1]; f += 256) {
    float p[3][3]; Corner c[3];
    bool cull = false;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const float* m = P.model_v + 3L * P.faces[3 * f + q];
#pragma unroll
      for (int j = 0; j < 3; ++j) p[q][j] = x.A[3 * j] * m[0] + x.A[3 * j + 1] * m[1] + x.A[3 * j + 2] * m[2] + x.tr[j];
      c[q] = project(P, p[q]);
      cull = cull || c[q].z < P.cull_eps;
    }
    if (cull) continue;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const float* a = gf + 9L * f + 3 * q;
      const float* b = gf + 9L * (P.F + f) + 3 * (2 - q);
      const float gu = a[0] + b[0], gv = a[1] + b[1], gz = a[2] + b[2];
      const float iz = 1.f / (c[q].z + P.proj_eps);
      const float gxh = s2 * (gu * P.K[0] - gv * P.K[3]);
      const float gyh = s2 * (gu * P.K[1] - gv * P.K[4]);
      const float gx = gxh * iz, gy = gyh * iz;
      const float gzc = gz - (gxh * c[q].x + gyh * c[q].y) * iz * iz;
      float gp[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) gp[j] = gx * P.R[j] + gy * P.R[3 + j] + gzc * P.R[6 + j];
      const float* m = P.model_v + 3L * P.faces[3 * f + q];
      const float d[3] = {m[0] - mc[0], m[1] - mc[1], m[2] - mc[2]};
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        acc[j] += gp[j];
#pragma unroll
        for (int e = 0; e < 3; ++e) acc[3 + 3 * j + e] += gp[j] * d[e];
      }
    }
  }
  __shared__ float red[4][12];
#pragma unroll
  for (int j = 0; j < 12; ++j) {
    float v = acc[j];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][j] = v;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  float g[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) g[j] = red[0][j] + red[1][j] + red[2][j] + red[3][j];
  const float* dA = g + 3;
  // A = scale * R_y:  d scale = <dA, R_y>,  d theta = scale * <dA, dR_y/dtheta>
  const float dscale = dA[0] * x.c + dA[2] * x.s + dA[4] - dA[6] * x.s + dA[8] * x.c;
  const float dtheta = x.scale * (-dA[0] * x.s + dA[2] * x.c - dA[6] * x.c - dA[8] * x.s);
  g_angles[row] = -dtheta * (float)(2.0 * 3.14159265358979323846 / 24.0);
  const float gsl = g_size_loss ? g_size_loss[0] : 0.f;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    float dsize = j == x.jmin ? dscale / P.msize[3 * k + j] : 0.f;
    if (size_target) dsize += gsl * 2.f * (x.size[j] - size_target[3 * k + j]) / 3.f;
    g_boxes[6 * row + j] = (g[j] / 2.f - dsize) * P.ext[j];
    g_boxes[6 * row + 3 + j] = (g[j] / 2.f + dsize) * P.ext[j];
  }
}

__global__ __launch_bounds__(256) void place_backward_kernel(SlnPlacement P, const float* __restrict__ boxes, const float* __restrict__ angles,
                                                             const float* __restrict__ size_target, const float* __restrict__ gf,
                                                             const float* __restrict__ g_size_loss, float* __restrict__ g_boxes,
                                                             float* __restrict__ g_angles) {
  place_backward_body(P, boxes, angles, size_target, gf, g_size_loss, g_boxes, g_angles);
}
__global__ __launch_bounds__(256) void place_backward_rooms_kernel(const SlnPlacementRoom* __restrict__ rooms) {
  const SlnPlacementRoom& r = rooms[blockIdx.y];
  if ((int)blockIdx.x >= r.P.n) return;
  place_backward_body(r.P, r.boxes, r.angles, r.size_target, r.grad_faces, r.grad_size_loss, r.grad_boxes, r.grad_angles);
  if (r.boxes_pred == nullptr) return;
  __syncthreads();                                             // the row's gradients are in memory (written by this workgroup)
  if (threadIdx.x == 0) {
    const int row = blockIdx.x;
    head_backward_row(row, row == r.P.n - 1, r.n_angle, r.angles_pred, r.grad_boxes + 6 * row, r.grad_angles[row], r.beta, r.grad_boxes_pred,
                      r.grad_angles_pred, r.ld_gb);
    if (row == 0 && blockIdx.y == 0 && r.noise_step != nullptr) r.noise_step[0] += 1;      // the next forward reads the next row of the noise table
  }
}

// ---- the torch glue between the decoder and the placement as one kernel each way (testing/test_render_refine.py:296-306) ----
// forward: boxes_full = [boxes_pred[:-1] ; box_last], idx = [softargmax(angles_pred, beta)[:-1] + noise[:-1] / 10 ; angle_last]
// with softargmax(x) = sum_j softmax(beta x)_j (j + 1) - 1 (:20-25).  One thread per row.
// (room_of_row / last_row: rows of several rooms concatenated - box_last [R,6], angle_last [R]; nullptr: one room of n rows)
__device__ __forceinline__ void head_forward_row(const int i, const int room, const bool lastrow, const int na, const float* __restrict__ boxes_pred,
                                                 const float* __restrict__ angles_pred, const float* __restrict__ noise,
                                                 const float* __restrict__ box_last, const float* __restrict__ angle_last, const float beta,
                                                 float* __restrict__ box_out /* [6] */, float* __restrict__ idx_out /* [1] */) {
  for (int j = 0; j < 6; ++j) box_out[j] = lastrow ? box_last[6 * room + j] : boxes_pred[6 * i + j];
  if (lastrow) { idx_out[0] = angle_last[room]; return; }
  const float* a = angles_pred + (size_t)i * na;
  float m = -INFINITY;
  for (int j = 0; j < na; ++j) m = fmaxf(m, a[j] * beta);
  float den = 0.f, num = 0.f;
  for (int j = 0; j < na; ++j) { const float e = expf(a[j] * beta - m); den += e; num += e * (float)(j + 1); }
  idx_out[0] = num / den - 1.0f + (noise ? noise[i] / 10.0f : 0.f);
}
__global__ void refine_head_forward_kernel(int n, int na, const float* __restrict__ boxes_pred, const float* __restrict__ angles_pred,
                                           const float* __restrict__ noise, const float* __restrict__ box_last,
                                           const float* __restrict__ angle_last, float beta, float* __restrict__ boxes_full,
                                           float* __restrict__ idx, const int* __restrict__ room_of_row, const int* __restrict__ last_row) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int room = room_of_row ? room_of_row[i] : 0;
  const bool lastrow = i == (room_of_row ? last_row[room] : n - 1);
  head_forward_row(i, room, lastrow, na, boxes_pred, angles_pred, noise, box_last, angle_last, beta, boxes_full + 6 * i, idx + i);
}
// backward, with the two gradient hooks of the reference folded in: quad_grad (x4 on d idx, :226-228) and fix_grad (both halves of a
// box row receive the mean of the two halves' gradients, :217-224); the frozen last row receives zeros.
__device__ __forceinline__ void head_backward_row(const int i, const bool lastrow, const int na, const float* __restrict__ angles_pred,
                                                  const float* __restrict__ g_box_row /* [6] */, const float g_idx_row, const float beta,
                                                  float* __restrict__ g_boxes_pred, float* __restrict__ g_angles_pred, const int ld_gb) {
  for (int j = 0; j < 3; ++j) {
    const float avg = lastrow ? 0.f : g_box_row[3 + j] / 2.0f + g_box_row[j] / 2.0f;
    g_boxes_pred[(size_t)ld_gb * i + j] = avg; g_boxes_pred[(size_t)ld_gb * i + 3 + j] = avg;
  }
  for (int j = 6; j < ld_gb; ++j) g_boxes_pred[(size_t)ld_gb * i + j] = 0.f;
  float* ga = g_angles_pred + (size_t)i * na;
  if (lastrow) { for (int j = 0; j < na; ++j) ga[j] = 0.f; return; }
  const float* a = angles_pred + (size_t)i * na;
  float m = -INFINITY;
  for (int j = 0; j < na; ++j) m = fmaxf(m, a[j] * beta);
  float den = 0.f, num = 0.f;
  for (int j = 0; j < na; ++j) { const float e = expf(a[j] * beta - m); den += e; num += e * (float)(j + 1); }
  const float ev = num / den, g = g_idx_row * 4.0f;
  // d idx / d a_j = beta p_j ((j + 1) - E[j + 1])
  for (int j = 0; j < na; ++j) ga[j] = g * beta * (expf(a[j] * beta - m) / den) * ((float)(j + 1) - ev);
}
__global__ void refine_head_backward_kernel(int n, int na, const float* __restrict__ angles_pred, const float* __restrict__ g_boxes_full,
                                            const float* __restrict__ g_idx, float beta, float* __restrict__ g_boxes_pred,
                                            float* __restrict__ g_angles_pred, const int* __restrict__ room_of_row,
                                            const int* __restrict__ last_row, const int ld_gb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool lastrow = i == (room_of_row ? last_row[room_of_row[i]] : n - 1);
  head_backward_row(i, lastrow, na, angles_pred, g_boxes_full + 6 * i, g_idx[i], beta, g_boxes_pred, g_angles_pred, ld_gb);
}
// p -= step * g, g = 0 (the next backward accumulates into it) and z -= step_z * gz in one launch: the closed form of the
// reference's per-iteration SGD(momentum = 0.1, nesterov) on a fresh optimizer (:286-292: step = lr * 1.1)
__global__ void refine_sgd_kernel(float* __restrict__ p, float* __restrict__ g, long n, float step, float* __restrict__ z,
                                  const float* __restrict__ gz, long nz, float step_z) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n4 = n >> 2;
  if (i < n4) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    pv.x -= step * gv.x; pv.y -= step * gv.y; pv.z -= step * gv.z; pv.w -= step * gv.w;
    reinterpret_cast<float4*>(p)[i] = pv;
    reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  } else if (i < n4 + (n & 3)) {
    const long k = 4 * n4 + (i - n4);
    p[k] -= step * g[k]; g[k] = 0.f;
  }
  if (i < nz) z[i] -= step_z * gz[i];
}

// R parameter copies, up to SGD_MAX_RANGES ranges each (see sln_refine_sgd_rooms); blockIdx.y = room * n_ranges + range
constexpr int SGD_MAX_RANGES = 96;
struct SgdRanges { long off[SGD_MAX_RANGES], len[SGD_MAX_RANGES]; int n; };
__global__ void refine_sgd_rooms_kernel(float* __restrict__ p, float* __restrict__ g, long stride, SgdRanges rg, float step, float* __restrict__ z,
                                        const float* __restrict__ gz, long nz, float step_z) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int room = blockIdx.y / rg.n, k = blockIdx.y % rg.n;
  if (i < (rg.len[k] >> 2)) {
    const long e = (long)room * stride + rg.off[k] + 4 * i;
    float4 pv = *reinterpret_cast<float4*>(p + e);
    const float4 gv = *reinterpret_cast<const float4*>(g + e);
    pv.x -= step * gv.x; pv.y -= step * gv.y; pv.z -= step * gv.z; pv.w -= step * gv.w;
    *reinterpret_cast<float4*>(p + e) = pv;
    *reinterpret_cast<float4*>(g + e) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (blockIdx.y == 0 && i < nz) z[i] -= step_z * gz[i];
}

int check(const SlnPlacement* P) {
  if (!P || P->n <= 0 || P->n_vis < 0 || P->n_vis > P->n || P->F <= 0 || P->Vm <= 0 || P->Vs < 0) return SLN_E_BADARG;
  if (!P->faces || !P->obj_face_ptr || (P->n_vis > 0 && (!P->vis || !P->model_v || !P->msize || !P->mcenter)) || (P->Vs > 0 && !P->shell_v))
    return SLN_E_BADARG;
  return 0;
}

}  // namespace

extern "C" {

int sln_place_forward(const SlnPlacement* P, const float* boxes, const float* angles, const float* size_target, float* faces_out,
                      float* sizes, float* size_loss, void* stream) {
  int r = check(P);
  if (r) return r;
  if (!boxes || !angles || !faces_out || !sizes || !size_loss) return SLN_E_BADARG;
  hipLaunchKernelGGL(place_forward_kernel, dim3(sln_cdiv(P->F, 256)), dim3(256), 0, (hipStream_t)stream, *P, boxes, angles, size_target,
                     faces_out, sizes, size_loss);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_place_backward(const SlnPlacement* P, const float* boxes, const float* angles, const float* size_target, const float* grad_faces,
                       const float* grad_size_loss, float* grad_boxes, float* grad_angles, void* stream) {
  int r = check(P);
  if (r) return r;
  if (!boxes || !angles || !grad_faces || !grad_boxes || !grad_angles) return SLN_E_BADARG;
  hipLaunchKernelGGL(place_backward_kernel, dim3(P->n), dim3(256), 0, (hipStream_t)stream, *P, boxes, angles, size_target, grad_faces,
                     grad_size_loss, grad_boxes, grad_angles);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_refine_head_forward(int n, int n_angle, const float* boxes_pred, const float* angles_pred, const float* noise,
                            const float* box_last, const float* angle_last, float beta, float* boxes_full, float* idx, void* stream) {
  if (n <= 0 || n_angle <= 0 || !boxes_pred || !angles_pred || !box_last || !angle_last || !boxes_full || !idx) return SLN_E_BADARG;
  hipLaunchKernelGGL(refine_head_forward_kernel, dim3(sln_cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, n, n_angle, boxes_pred,
                     angles_pred, noise, box_last, angle_last, beta, boxes_full, idx, (const int*)nullptr, (const int*)nullptr);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_refine_head_backward(int n, int n_angle, const float* angles_pred, const float* grad_boxes_full, const float* grad_idx,
                             float beta, float* grad_boxes_pred, float* grad_angles_pred, void* stream) {
  if (n <= 0 || n_angle <= 0 || !angles_pred || !grad_boxes_full || !grad_idx || !grad_boxes_pred || !grad_angles_pred) return SLN_E_BADARG;
  hipLaunchKernelGGL(refine_head_backward_kernel, dim3(sln_cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, n, n_angle, angles_pred,
                     grad_boxes_full, grad_idx, beta, grad_boxes_pred, grad_angles_pred, (const int*)nullptr, (const int*)nullptr, 6);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_refine_head_forward_rooms(int rows_total, int n_angle, const int32_t* room_of_row, const int32_t* last_row, const float* boxes_pred,
                                  const float* angles_pred, const float* noise, const float* box_last, const float* angle_last, float beta,
                                  float* boxes_full, float* idx, void* stream) {
  if (rows_total <= 0 || n_angle <= 0 || !room_of_row || !last_row || !boxes_pred || !angles_pred || !box_last || !angle_last || !boxes_full || !idx)
    return SLN_E_BADARG;
  hipLaunchKernelGGL(refine_head_forward_kernel, dim3(sln_cdiv(rows_total, 64)), dim3(64), 0, (hipStream_t)stream, rows_total, n_angle, boxes_pred,
                     angles_pred, noise, box_last, angle_last, beta, boxes_full, idx, room_of_row, last_row);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_refine_head_backward_rooms(int rows_total, int n_angle, const int32_t* room_of_row, const int32_t* last_row, const float* angles_pred,
                                   const float* grad_boxes_full, const float* grad_idx, float beta, float* grad_boxes_pred, int ld_gb,
                                   float* grad_angles_pred, void* stream) {
  if (rows_total <= 0 || n_angle <= 0 || !room_of_row || !last_row || !angles_pred || !grad_boxes_full || !grad_idx || !grad_boxes_pred ||
      !grad_angles_pred || ld_gb < 6) return SLN_E_BADARG;
  hipLaunchKernelGGL(refine_head_backward_kernel, dim3(sln_cdiv(rows_total, 64)), dim3(64), 0, (hipStream_t)stream, rows_total, n_angle, angles_pred,
                     grad_boxes_full, grad_idx, beta, grad_boxes_pred, grad_angles_pred, room_of_row, last_row, ld_gb);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_place_forward_rooms(const SlnPlacementRoom* rooms, int R, int F_max, void* stream) {
  if (!rooms || R <= 0 || F_max <= 0) return SLN_E_BADARG;
  hipLaunchKernelGGL(place_forward_rooms_kernel, dim3(sln_cdiv(F_max, 256), R), dim3(256), 0, (hipStream_t)stream, rooms);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_place_backward_rooms(const SlnPlacementRoom* rooms, int R, int n_max, void* stream) {
  if (!rooms || R <= 0 || n_max <= 0) return SLN_E_BADARG;
  hipLaunchKernelGGL(place_backward_rooms_kernel, dim3(n_max, R), dim3(256), 0, (hipStream_t)stream, rooms);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_refine_sgd_rooms(float* params, float* grads, int R, int64_t stride, const int64_t* off_host, const int64_t* len_host, int n_ranges,
                         float step, float* z, const float* grad_z, int64_t nz, float step_z, void* stream) {
  if (!params || !grads || R <= 0 || n_ranges < 1 || n_ranges > SGD_MAX_RANGES || !off_host || !len_host || nz < 0 || (nz > 0 && (!z || !grad_z))) return SLN_E_BADARG;
  if (((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads)) & 15) || (stride & 3)) return SLN_E_BADARG;
  SgdRanges rg; std::memset(&rg, 0, sizeof(rg));
  rg.n = n_ranges;
  long work = nz;
  for (int k = 0; k < n_ranges; ++k) {
    if ((off_host[k] & 3) || (len_host[k] & 3) || off_host[k] < 0 || len_host[k] < 0 || off_host[k] + len_host[k] > stride) return SLN_E_BADARG;
    rg.off[k] = (long)off_host[k]; rg.len[k] = (long)len_host[k];
    work = std::max<long>(work, len_host[k] >> 2);
  }
  if (work <= 0) return 0;
  hipLaunchKernelGGL(refine_sgd_rooms_kernel, dim3((unsigned)((work + 255) / 256), R * n_ranges), dim3(256), 0, (hipStream_t)stream, params, grads,
                     (long)stride, rg, step, z, grad_z, (long)nz, step_z);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_refine_sgd(float* params, float* grads, int64_t n, float step, float* z, const float* grad_z, int64_t nz, float step_z,
                   void* stream) {
  if (n < 0 || nz < 0 || (n > 0 && (!params || !grads)) || (nz > 0 && (!z || !grad_z))) return SLN_E_BADARG;
  if ((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads)) & 15) return SLN_E_BADARG;
  const long work = std::max<long>((n >> 2) + (n & 3), nz);
  if (work <= 0) return 0;
  hipLaunchKernelGGL(refine_sgd_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream, params, grads, (long)n, step,
                     z, grad_z, (long)nz, step_z);
  SLN_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
