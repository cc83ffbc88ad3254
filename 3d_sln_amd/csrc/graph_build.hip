// Scene-graph builder on the device: the work of SuncgDataset.__getitem__ + suncg_collate_fn
// (reference data/suncg_dataset.py:110-353, utils.py:36-80 compute_rel) for a whole batch of rooms.
//
// The room table (classes, raw boxes, rotations, room boxes, size thresholds) stays resident in HBM; a batch is a
// list of room indices plus the random decisions the reference draws from python's `random` (which partner, subject /
// object order, which size attribute) - injected, so that the result is bit-identical to the reference given the same
// draws.  One wavefront per room:
//   plan : number of rows / triples of every room ('on' pairs depend on the geometry) + exclusive scan
//   emit : objs, boxes (room-normalised), angles, attributes, triples (reference order: 'on' pairs by (cur, other),
//          one drawn pair per object, then the __in_room__ rows), obj_to_img, triple_to_img.
// compute_rel is evaluated in float32 exactly as the reference does (left-to-right, no contraction: this file is
// built with -ffp-contract=off); its atan2 sector test is replaced by the equivalent comparisons
// (oracle/graph_build_ref.py::sector_by_compare, tests/test_oracle_graph_build.py).
#include "sln_common.h"
#include "../../include/sln_hip.h"

namespace {

enum { P_IN_ROOM = 0, P_LEFT = 1, P_RIGHT = 2, P_BEHIND = 3, P_FRONT = 4, P_INSIDE = 5, P_SURROUND = 6, P_LEFT_T = 7,
       P_RIGHT_T = 8, P_FRONT_T = 9, P_BEHIND_T = 10, P_ON = 15 };

struct Box { float x0, y0, z0, x1, y1, z1; };

__device__ __forceinline__ Box load_box(const float* p) { Box b; b.x0 = p[0]; b.y0 = p[1]; b.z0 = p[2]; b.x1 = p[3]; b.y1 = p[4]; b.z1 = p[5]; return b; }

__device__ __forceinline__ bool rel_is_on(const Box& s, const Box& o) {   // utils.py:45-52
  const float c1x = (s.x0 + s.x1) / 2.f, c1y = (s.y0 + s.y1) / 2.f, c1z = (s.z0 + s.z1) / 2.f;
  const float c2y = (o.y0 + o.y1) / 2.f;
  if (c1x >= o.x0 && c1x <= o.x1 && c1z >= o.z0 && c1z <= o.z1) {
    const float delta1 = c1y - c2y;
    const float delta2 = (s.y1 - s.y0 + o.y1 - o.y0) / 2.f;
    return fabsf(delta1 - delta2) < 0.05f;
  }
  return false;
}

__device__ __forceinline__ int compute_rel(const Box& s, const Box& o) {  // utils.py:36-80
  if (rel_is_on(s, o)) return P_ON;
  const float dx = (s.x0 + s.x1) / 2.f - (o.x0 + o.x1) / 2.f;
  const float dz = (s.z0 + s.z1) / 2.f - (o.z0 + o.z1) / 2.f;
  const float area_s = (s.x1 - s.x0) * (s.z1 - s.z0);
  const float area_o = (o.x1 - o.x0) * (o.z1 - o.z0);
  const float ix0 = fmaxf(s.x0, o.x0), ix1 = fminf(s.x1, o.x1);
  const float iz0 = fmaxf(s.z0, o.z0), iz1 = fminf(s.z1, o.z1);
  const float area_i = fmaxf(0.f, ix1 - ix0) * fmaxf(0.f, iz1 - iz0);
  const float iou = area_i / (area_s + area_o - area_i);
  const bool touching = 0.0001f < iou && iou < 0.5f;
  if (s.x0 < o.x0 && s.x1 > o.x1 && s.z0 < o.z0 && s.z1 > o.z1) return P_SURROUND;
  if (s.x0 > o.x0 && s.x1 < o.x1 && s.z0 > o.z0 && s.z1 < o.z1) return P_INSIDE;
  if (dx < 0.f && fabsf(dz) <= -dx) return touching ? P_RIGHT_T : P_LEFT;        // theta >= 3pi/4 or <= -3pi/4
  if (dz < 0.f && fabsf(dx) < -dz) return touching ? P_BEHIND_T : P_BEHIND;      // -3pi/4 <= theta < -pi/4
  if ((dx > 0.f && -dx <= dz && dz < dx) || (dx == 0.f && dz == 0.f)) return touching ? P_LEFT_T : P_RIGHT;
  return touching ? P_FRONT_T : P_FRONT;
}

__device__ __forceinline__ int lane_prefix(unsigned long long m, int lane) { return __popcll(m & ((1ull << lane) - 1ull)); }

// number of 'on' pairs of one room, counted by the whole wavefront (uniform result)
__device__ int count_on_pairs(const float* bbox, int n, int lane) {
  int total = 0;
  for (int cur = 0; cur < n; ++cur) {
    const Box s = load_box(bbox + (size_t)cur * 6);
    for (int o0 = 0; o0 < n; o0 += 64) {
      const int oth = o0 + lane;
      bool hit = false;
      if (oth < n && oth != cur) hit = rel_is_on(s, load_box(bbox + (size_t)oth * 6));
      total += __popcll(__ballot(hit));
    }
  }
  return total;
}

__global__ __launch_bounds__(64) void graph_count_kernel(SlnRoomTable tab, const int* __restrict__ room_idx, int B, int* __restrict__ counts) {
  const int b = blockIdx.x, lane = threadIdx.x;
  int r = room_idx[b];
  const bool valid = r >= 0 && r < tab.n_rooms;
  r = valid ? r : 0;
  const int first = tab.room_off[r], n = tab.room_off[r + 1] - first;
  const int n_on = count_on_pairs(tab.bbox + (size_t)first * 6, n, lane);
  if (lane == 0) {
    counts[2 * b + 0] = valid ? n + 1 : -1;                 // rows: objects + the room (negative: bad room index)
    counts[2 * b + 1] = n_on + (n >= 2 ? n : 0) + n;        // 'on' pairs + one drawn pair per object + __in_room__
  }
}

// exclusive scan of the per-room counts: off[0..B] rows, off[B+1 .. 2B+1] triples, off[2B+2] = number of bad room indices
__global__ __launch_bounds__(256) void graph_scan_kernel(const int* __restrict__ counts, int B, int* __restrict__ off) {
  __shared__ int part[2][256];
  __shared__ int bad_s;
  const int tid = threadIdx.x;
  const int per = (B + 255) / 256, lo = min(B, tid * per), hi = min(B, lo + per);
  if (tid == 0) bad_s = 0;
  __syncthreads();
  int s0 = 0, s1 = 0, bad = 0;
  for (int i = lo; i < hi; ++i) { const int c = counts[2 * i]; bad += c < 0; s0 += max(c, 0); s1 += counts[2 * i + 1]; }
  part[0][tid] = s0; part[1][tid] = s1;
  if (bad) atomicAdd(&bad_s, bad);
  __syncthreads();
  if (tid == 0) {
    int a = 0, c = 0;
    for (int i = 0; i < 256; ++i) { const int t0 = part[0][i], t1 = part[1][i]; part[0][i] = a; part[1][i] = c; a += t0; c += t1; }
    off[B] = a; off[2 * B + 1] = c; off[2 * B + 2] = bad_s;
  }
  __syncthreads();
  int a = part[0][tid], c = part[1][tid];
  for (int i = lo; i < hi; ++i) { off[i] = a; off[B + 1 + i] = c; a += max(counts[2 * i], 0); c += counts[2 * i + 1]; }
}

__global__ __launch_bounds__(64) void graph_emit_kernel(SlnRoomTable tab, const int* __restrict__ room_idx, int B,
                                                        const int* __restrict__ off, SlnGraphDraws draws, SlnGraphBatch out) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int r = room_idx[b];
  if (r < 0 || r >= tab.n_rooms) return;
  const int first = tab.room_off[r], n = tab.room_off[r + 1] - first;
  const int row0 = off[b], trip0 = off[B + 1 + b];
  const int draw0 = row0 - b;                                  // draws hold one entry per NON-room row, in batch order
  const float* bbox = tab.bbox + (size_t)first * 6;
  const float rx = tab.room_bbox[3 * r + 0], ry = tab.room_bbox[3 * r + 1], rz = tab.room_bbox[3 * r + 2];
  if (lane == 0 && out.ids) out.ids[b] = tab.room_id ? tab.room_id[r] : r;

  // ---- rows: objs, boxes (normalised after the triples in the reference, independent here), angles, attributes ----
  for (int i = lane; i <= n; i += 64) {
    const size_t row = (size_t)row0 + i;
    float* bo = out.boxes + row * 6;
    int attr = 0;
    if (i < n) {
      const Box s = load_box(bbox + (size_t)i * 6);
      const float x0 = s.x0 / rx, y0 = s.y0 / ry, z0 = s.z0 / rz, x1 = s.x1 / rx, y1 = s.y1 / ry, z1 = s.z1 / rz;
      bo[0] = x0; bo[1] = y0; bo[2] = z0; bo[3] = x1; bo[4] = y1; bo[5] = z1;
      const int cls = tab.cls[first + i];
      const int mode = draws.attr_mode[draw0 + i];               // 0 none, 1 height test, 2 volume test
      const bool known = cls >= 0 && cls < tab.n_classes && tab.has_size[cls];
      if (mode != 0 && known) {
        const float* th = tab.size_thr + 4 * cls;
        const float height = y1 - y0;
        const float volume = (x1 - x0) * (y1 - y0) * (z1 - z0);
        if (!tab.use_attr_30) {
          if (mode == 1) attr = height > th[0] ? 1 : 2;          // tall / short
          else attr = volume > th[1] ? 3 : 4;                    // large / small
        } else {
          if (mode == 1) attr = height > th[0] ? 1 : (height < th[1] ? 2 : 0);
          else attr = volume > th[2] ? 3 : (volume < th[3] ? 4 : 0);
        }
      }
      out.objs[row] = cls; out.angles[row] = tab.rot[first + i];
    } else {
      bo[0] = 0.f; bo[1] = 0.f; bo[2] = 0.f; bo[3] = rx; bo[4] = ry; bo[5] = rz;
      out.objs[row] = 0; out.angles[row] = 0;                    // '__room__' (vocab index 0), angle 0
    }
    out.attributes[row] = attr;
    out.obj_to_img[row] = b;
  }

  // ---- triples ----
  int t = trip0;
  for (int cur = 0; cur < n; ++cur) {                            // 'on' pairs, (cur, other) order kept by the compaction
    const Box s = load_box(bbox + (size_t)cur * 6);
    for (int o0 = 0; o0 < n; o0 += 64) {
      const int oth = o0 + lane;
      bool hit = false;
      if (oth < n && oth != cur) hit = rel_is_on(s, load_box(bbox + (size_t)oth * 6));
      const unsigned long long m = __ballot(hit);
      if (hit) {
        int64_t* tr = out.triples + (size_t)(t + lane_prefix(m, lane)) * 3;
        tr[0] = row0 + cur; tr[1] = P_ON; tr[2] = row0 + oth;
      }
      t += __popcll(m);
    }
  }
  if (n >= 2) {
    for (int cur = lane; cur < n; cur += 64) {                   // one drawn pair per object (suncg_dataset.py:189-205)
      int oth = draws.other[draw0 + cur];
      oth = min(max(oth, 0), n - 1);
      const bool keep = draws.swap[draw0 + cur] != 0;
      const int s = keep ? cur : oth, o = keep ? oth : cur;
      const int p = compute_rel(load_box(bbox + (size_t)s * 6), load_box(bbox + (size_t)o * 6));
      int64_t* tr = out.triples + (size_t)(t + cur) * 3;
      tr[0] = row0 + s; tr[1] = p; tr[2] = row0 + o;
    }
    t += n;
  }
  for (int i = lane; i < n; i += 64) {
    int64_t* tr = out.triples + (size_t)(t + i) * 3;
    tr[0] = row0 + i; tr[1] = P_IN_ROOM; tr[2] = row0 + n;
  }
  t += n;
  for (int i = trip0 + lane; i < t; i += 64) out.triple_to_img[i] = b;
}

// The random decisions of a batch drawn on the device in ONE launch (round 3; host/suncg_dataset.py::device_draws used to spell
// them as ~15 ATen launches): Philox-4x32-10 keyed by two 64-bit words in device memory (the caller takes them from its torch
// generator, so the generator's stream advances and nothing is read back), counter = the object's row among the non-room rows
// of the batch; the four words of a counter are the four uniforms of one object (suncg_dataset.py:189-196, 236-282):
//   other = uniform choice among the n - 1 other objects, swap = u > 0.5, attribute mode = none / height / volume.
__device__ __forceinline__ void gb_philox(unsigned int (&c)[4], unsigned int k0, unsigned int k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
    const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned int)p1;
    const unsigned int n2 = (unsigned int)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned int)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

__global__ __launch_bounds__(64) void graph_draw_kernel(SlnRoomTable tab, const int* __restrict__ room_idx, int B,
                                                        const int* __restrict__ off, const long long* __restrict__ key,
                                                        int* __restrict__ other, unsigned char* __restrict__ swap,
                                                        unsigned char* __restrict__ mode) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int r = room_idx[b];
  if (r < 0 || r >= tab.n_rooms) return;
  const int first = tab.room_off[r], n = tab.room_off[r + 1] - first;
  const int base = off[b] - b;                       // rows of earlier rooms minus their room rows = first draw slot of this room
  const unsigned long long k0 = (unsigned long long)key[0], k1 = (unsigned long long)key[1];
  for (int cur = lane; cur < n; cur += 64) {
    const unsigned int j = (unsigned int)(base + cur);
    unsigned int c[4] = {j, (unsigned int)(k1 >> 32), (unsigned int)k1, 0x5ce9e6a1u};
    gb_philox(c, (unsigned int)k0, (unsigned int)(k0 >> 32));
    const float u0 = (float)(c[0] >> 8) * 5.9604644775390625e-08f, u1 = (float)(c[1] >> 8) * 5.9604644775390625e-08f;   // [0, 1)
    const float u2 = (float)(c[2] >> 8) * 5.9604644775390625e-08f, u3 = (float)(c[3] >> 8) * 5.9604644775390625e-08f;
    int k = min((int)(u0 * (float)(n - 1)), n - 2);
    k = max(k, 0);
    other[j] = k + (k >= cur ? 1 : 0);
    swap[j] = u1 > 0.5f ? 1 : 0;
    const bool known = tab.has_size[tab.cls[first + cur]] != 0;
    mode[j] = (u2 > 0.5f || !known) ? 0 : (u3 > 0.5f ? 1 : 2);
  }
}

}  // namespace

extern "C" int sln_graph_draw(const SlnRoomTable* tab, const int* room_idx, int B, const int* offsets, const int64_t* key,
                              int* other, unsigned char* swap, unsigned char* attr_mode, void* stream) {
  if (!tab || !room_idx || !offsets || !key || !other || !swap || !attr_mode || B < 0) return -1;
  if (B == 0) return 0;
  hipLaunchKernelGGL(graph_draw_kernel, dim3(B), dim3(64), 0, static_cast<hipStream_t>(stream), *tab, room_idx, B, offsets,
                     reinterpret_cast<const long long*>(key), other, swap, attr_mode);
  SLN_CHECK_LAUNCH();
  return 0;
}

extern "C" int sln_graph_plan(const SlnRoomTable* tab, const int* room_idx, int B, int* counts, int* offsets, void* stream) {
  if (!tab || !room_idx || !counts || !offsets || B < 0) return -1;
  if (B == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(graph_count_kernel, dim3(B), dim3(64), 0, st, *tab, room_idx, B, counts);
  SLN_CHECK_LAUNCH();
  hipLaunchKernelGGL(graph_scan_kernel, dim3(1), dim3(256), 0, st, counts, B, offsets);
  SLN_CHECK_LAUNCH();
  return 0;
}

extern "C" int sln_graph_emit(const SlnRoomTable* tab, const int* room_idx, int B, const int* offsets, const SlnGraphDraws* draws,
                              const SlnGraphBatch* out, void* stream) {
  if (!tab || !room_idx || !offsets || !draws || !out || B < 0) return -1;
  if (!out->objs || !out->boxes || !out->triples || !out->angles || !out->attributes || !out->obj_to_img || !out->triple_to_img) return -1;
  if (!draws->other || !draws->swap || !draws->attr_mode) return -1;
  if (B == 0) return 0;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(graph_emit_kernel, dim3(B), dim3(64), 0, st, *tab, room_idx, B, offsets, *draws, *out);
  SLN_CHECK_LAUNCH();
  return 0;
}
