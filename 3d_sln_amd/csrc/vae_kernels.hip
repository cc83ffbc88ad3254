// Non-GEMM kernels of the scene-graph VAE path (gfx950): graph CSR, edge aggregation
// (scatter/gather), embedding assembly, loss, BatchNorm bookkeeping, transposes, Adam.
//
// These are HBM/L2-bound streaming kernels: coalesced 64-column row segments per wave, per-column
// statistics reduced in registers -> LDS -> one fp64 atomic per column per block.
#include <cstring>
#include "vae_kernels.h"
#include "sln_prof.h"

namespace {

constexpr int CB = 64;   // columns per block
constexpr int RL = 4;    // row lanes per block
constexpr int BOX_ROWS = 64;   // rows per block of box_embed_bwd_kernel
constexpr int RB = 32;   // rows per block (dense row kernels)

__device__ __forceinline__ void commit_col_stats(float s1, float s2, bool valid, double* out, int cstride, int col) {
  __shared__ float red[2][RL][CB];
  red[0][threadIdx.y][threadIdx.x] = s1;
  red[1][threadIdx.y][threadIdx.x] = s2;
  __syncthreads();
  if (threadIdx.y == 0 && valid && out != nullptr) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int i = 0; i < RL; ++i) { a += red[0][i][threadIdx.x]; b += red[1][i][threadIdx.x]; }
    atomicAdd(out + col, sln_qd((double)a, SLN_Q_BWD));
    atomicAdd(out + cstride + col, sln_qd((double)b, SLN_Q_BWD));
  }
}

// ----------------------------------------------------------------------------------------------
// graph CSR
// ----------------------------------------------------------------------------------------------
__global__ void prep_split_kernel(const int64_t* __restrict__ tri, int T, int O, int num_preds, GraphCsr g, int* err, int stride,
                                  int poff, int ooff) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  // triples [T,3] = (s, p, o) (stride 3) or edges [T,2] = (s, o) (stride 2, no predicate)
  const int64_t s = tri[stride * t], p = poff >= 0 ? tri[stride * t + poff] : 0, o = tri[stride * t + ooff];
  // out-of-range ids (the reference's index ops would raise): flag them and neutralise the triple
  if (s < 0 || s >= O || o < 0 || o >= O || p < 0 || p >= num_preds) {
    if (err) atomicOr(err, 1);
    g.s[t] = 0; g.p[t] = 0; g.o[t] = 0;
    atomicAdd(g.deg, 2);
    return;
  }
  g.s[t] = (int)s; g.p[t] = (int)p; g.o[t] = (int)o;
  atomicAdd(g.deg + (int)s, 1);
  atomicAdd(g.deg + (int)o, 1);
}

__global__ __launch_bounds__(1024) void csr_scan_kernel(GraphCsr g, int O) {
  __shared__ int part[1024];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < O; base += 1024) {
    const int i = base + threadIdx.x;
    const int d = i < O ? g.deg[i] : 0;
    part[threadIdx.x] = d;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {          // Hillis-Steele inclusive scan
      int v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
      __syncthreads();
      part[threadIdx.x] += v;
      __syncthreads();
    }
    const int excl = carry + part[threadIdx.x] - d;
    if (i < O) {
      g.rowptr[i] = excl; g.cursor[i] = excl;
      g.invdeg[i] = 1.0f / (float)(d > 1 ? d : 1);
    }
    __syncthreads();
    if (threadIdx.x == 1023) carry += part[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) g.rowptr[O] = carry;
}

__global__ void csr_fill_kernel(GraphCsr g, int T) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 2 * T) return;
  const int node = e < T ? g.s[e] : g.o[e - T];
  const int pos = atomicAdd(g.cursor + node, 1);
  g.ent[pos] = e;
}

// ascending entry id inside each row == reference scatter_add order (all subject rows in triple order, then all object rows,
// models/graph.py:97-98).  One wavefront per row, rank by counting: a lane's entry goes to position #(smaller entries).  (One
// THREAD per row with an insertion sort took 23 us per batch - the room node of every graph has 31+ entries - on the critical
// path of every training step, outside the captured iteration.)
constexpr int SORT_MAXDEG = 1024;      // entries of a row held in LDS; longer rows fall back to the serial sort of one lane
__global__ __launch_bounds__(64) void csr_sort_kernel(GraphCsr g, int O) {
  __shared__ int keys[SORT_MAXDEG];
  const int i = blockIdx.x;
  if (i >= O) return;
  const int b = g.rowptr[i], e = g.rowptr[i + 1], deg = e - b;
  if (deg <= 1) return;
  if (deg > SORT_MAXDEG) {
    if (threadIdx.x == 0)
      for (int a = b + 1; a < e; ++a) {
        const int key = g.ent[a];
        int j = a - 1;
        while (j >= b && g.ent[j] > key) { g.ent[j + 1] = g.ent[j]; --j; }
        g.ent[j + 1] = key;
      }
    return;
  }
  for (int k = threadIdx.x; k < deg; k += 64) keys[k] = g.ent[b + k];
  __syncthreads();
  for (int k = threadIdx.x; k < deg; k += 64) {
    const int key = keys[k];
    int rank = 0;
    for (int j = 0; j < deg; ++j) rank += keys[j] < key ? 1 : 0;       // entry ids of a row are distinct
    g.ent[b + rank] = key;
  }
}

// ----------------------------------------------------------------------------------------------
// edge aggregation
// ----------------------------------------------------------------------------------------------
// ----------------------------------------------------------------------------------------------
// Edge kernels (every row stride / column offset is a multiple of 4 floats - the engine's configurations guarantee it -
// and the base pointers are 16-byte aligned).  These kernels are chains of dependent loads (rowptr -> entry -> row
// data), not bandwidth: a scalar one-column-per-thread version spent ~20 us on ~20 MB.  Here a thread owns 4 consecutive columns (float4
// loads, a block of XT x YT threads covers 4*XT columns x YT rows), every row lane walks ONE row (CSR kernels) or a
// short unrolled run of rows (dense kernel), entry indices and row data of 4 entries are in flight together, and the
// BatchNorm coefficients are computed once per block (one column per thread) and shared through LDS.
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 ld4g(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4g(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// coefficient table of the block's columns [c0, c0 + 4*XT): (scale, shift, mean, istd); one column per thread.
// PLANAR (round 4): column j = 4 x + q of the block sits at tab[q * XT + x], so the read of thread x for its q-th column
// (tab[q * XT + x], one ds_read_b128) is 16 bytes from its neighbour's - conflict-free.  In column order (tab[4 x + q]) the lanes
// of a read were 64 bytes apart and touched 16 of the 64 banks: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.80 in the forward
// scatter, which re-reads its two tables for every entry of a row (profiles/r03_vae_sq_stalls.csv).
// rows of XT + 4 entries: the FILL (lane j -> row j & 3, entry j >> 2) then spreads every 16 consecutive lanes over the 16
// 16-byte bank groups as well (rows of XT entries put its four rows on the same banks: a 4-way conflict on each table write)
#define SLN_CTAB_ROW (XT + 4)
#define SLN_CTAB(j) ((((j) & 3) * SLN_CTAB_ROW) + ((j) >> 2))
template <int XT, int YT>
__device__ __forceinline__ void fill_coef_table(float4* tab, const BnView& bn, int c0, int ncols, int coloff) {
  const int tid = threadIdx.y * XT + threadIdx.x;
  for (int j = tid; j < 4 * XT; j += XT * YT) {
    float4 v = make_float4(1.f, 0.f, 0.f, 1.f);
    if (c0 + j < ncols) v = bn_fwd_coef4(bn, coloff + c0 + j);
    tab[SLN_CTAB(j)] = v;
  }
}

// Column statistics of a block: reduce over the row lanes in LDS, then ONE column per thread so that a wavefront's
// atomics hit consecutive doubles (the memory side serialises same-line atomics: lanes owning 4 columns each would
// touch every 128-byte line from four different instructions).
template <int XT, int YT>
__device__ __forceinline__ void commit_col_stats4(const float4& s1, const float4& s2, int ncols_valid, double* out, int cstride, int col0) {
  __shared__ float red[2][YT][4 * XT];
  *reinterpret_cast<float4*>(&red[0][threadIdx.y][4 * threadIdx.x]) = s1;
  *reinterpret_cast<float4*>(&red[1][threadIdx.y][4 * threadIdx.x]) = s2;
  __syncthreads();
  if (out == nullptr) return;
  const int tid = threadIdx.y * XT + threadIdx.x;
  for (int j = tid; j < 8 * XT; j += XT * YT) {
    const int which = j / (4 * XT), cj = j % (4 * XT);
    if (cj < ncols_valid) {
      float a = 0.f;
#pragma unroll
      for (int i = 0; i < YT; ++i) a += red[which][i][cj];
      atomicAdd(out + which * cstride + col0 + cj, sln_qd((double)a, SLN_Q_BWD));
    }
  }
}

// entry list of the block's YT rows, cached in LDS (coalesced) so that the per-entry chain is one level of row loads;
// rows longer than ECACHE (only the room node of a big graph) read the tail from global memory
constexpr int ECACHE = 64;
constexpr int ESTRIDE = ECACHE + 1;         // row stride of the LDS entry cache: the rows a wavefront covers (XT < 64) sit on different banks
constexpr int EB = 16;                     // row loads in flight per thread

// The prologue of an edge kernel is a chain of memory round trips in front of the first row: the CSR bounds of the row are loaded
// FIRST (row_bounds), the coefficient tables' loads go out next to them, the entry list follows (cache_entries): two round trips
// (the first version took eight: gamma / beta, sums, sums again per table, two tables, row pointers, entries).
__device__ __forceinline__ void row_bounds(const GraphCsr& g, int row, int nrows, int& b, int& e) {
  const int r = min(row, nrows - 1);               // unconditional, clamped
  b = g.rowptr[r]; e = g.rowptr[r + 1];
  if (row >= nrows) { b = 0; e = 0; }
}
template <int XT, int YT>
__device__ __forceinline__ void cache_entries(int (*ents)[ESTRIDE], const GraphCsr& g, int b, int e) {
  for (int k = threadIdx.x; k < min(e - b, ECACHE); k += XT) ents[threadIdx.y][k] = g.ent[b + k];
  __syncthreads();
}
// the two coefficient tables of the forward scatter (columns c0.. of the subject half and of the object half)
template <int XT, int YT>
// (scale, shift) only, 8 bytes per column: the forward scatter uses nothing else, and reading half of a 16-byte entry leaves
// the lanes of its ds_read_b64 16 bytes apart - every second bank pair idle, SQ_LDS_BANK_CONFLICT share 0.41)
__device__ __forceinline__ void fill_coef_table2(float2* ta, float2* tb, const BnView& bn, int c0, int ncols, int coloff_b) {
  const int tid = threadIdx.y * XT + threadIdx.x;
  for (int j = tid; j < 4 * XT; j += XT * YT) {
    float4 va = make_float4(1.f, 0.f, 0.f, 1.f), vb = va;
    if (c0 + j < ncols) bn_fwd_coef4x2(bn, c0 + j, coloff_b + c0 + j, va, vb);
    ta[SLN_CTAB(j)] = make_float2(va.x, va.y); tb[SLN_CTAB(j)] = make_float2(vb.x, vb.y);
  }
}

template <int XT, int YT>
__device__ __forceinline__ void scatter_avg_fwd_v4_body(const float* __restrict__ A2, int ld, int H, int D, BnView bn,
                                                                    GraphCsr g, int O, int T, float* __restrict__ pooled) {
  __shared__ float2 cs[4 * SLN_CTAB_ROW], co[4 * SLN_CTAB_ROW];
  __shared__ int ents[YT][ESTRIDE];
  const int c0 = blockIdx.x * 4 * XT;
  const int i = blockIdx.y * YT + threadIdx.y;
  int b, e;
  row_bounds(g, i, O, b, e);
  fill_coef_table2<XT, YT>(cs, co, bn, c0, H, H + D);
  cache_entries<XT, YT>(ents, g, b, e);
  const int c = c0 + 4 * threadIdx.x;
  if (c >= H || i >= O) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const float2* ks = cs + threadIdx.x; const float2* ko = co + threadIdx.x;      // planar: column q at [q * SLN_CTAB_ROW]
  const int deg = e - b;
  for (int k = 0; k < deg; k += EB) {
    int en[EB]; float4 x[EB];
    // (a select between the LDS copy and the global list makes hipcc emit flat loads, which count on both wait counters: the
    //  common case - the whole list is cached - reads LDS only; the branch is uniform per row)
    if (deg <= ECACHE) {
#pragma unroll
      for (int u = 0; u < EB; ++u) en[u] = ents[threadIdx.y][min(k + u, deg - 1)];
    } else {
#pragma unroll
      for (int u = 0; u < EB; ++u) { const int kk = min(k + u, deg - 1); en[u] = kk < ECACHE ? ents[threadIdx.y][kk] : g.ent[b + kk]; }
    }
#pragma unroll
    for (int u = 0; u < EB; ++u) {
      const bool isobj = en[u] >= T;
      x[u] = ld4g(A2 + (size_t)(isobj ? en[u] - T : en[u]) * ld + (isobj ? H + D : 0) + c);
    }
#pragma unroll
    for (int u = 0; u < EB; ++u) {
      if (k + u < deg) {                     // same accumulation order as the scalar kernel / the reference scatter_add
        const float2* kk = en[u] >= T ? ko : ks;
        acc.x += fmaxf(fmaf(kk[0].x, x[u].x, kk[0].y), 0.f); acc.y += fmaxf(fmaf(kk[1 * SLN_CTAB_ROW].x, x[u].y, kk[1 * SLN_CTAB_ROW].y), 0.f);
        acc.z += fmaxf(fmaf(kk[2 * SLN_CTAB_ROW].x, x[u].z, kk[2 * SLN_CTAB_ROW].y), 0.f); acc.w += fmaxf(fmaf(kk[3 * SLN_CTAB_ROW].x, x[u].w, kk[3 * SLN_CTAB_ROW].y), 0.f);
      }
    }
  }
  const float w = g.invdeg[i];
  st4g(pooled + (size_t)i * H + c, make_float4(acc.x * w, acc.y * w, acc.z * w, acc.w * w));
}
template <int XT, int YT>
__global__ __launch_bounds__(XT* YT) void scatter_avg_fwd_v4_kernel(const float* __restrict__ A2, int ld, int H, int D, BnView bn, GraphCsr g, int O, int T, float* __restrict__ pooled) {
  scatter_avg_fwd_v4_body<XT, YT>(A2, ld, H, D, bn, g, O, T, pooled);
}

template <int XT, int YT, int RPT, int NIT>
__device__ __forceinline__ void scatter_avg_bwd_v4_body(const float* __restrict__ dM, const float* __restrict__ dP, int lddp,
                                                                    int dpcol0, const float* __restrict__ A2, int ld, int H, int D,
                                                                    BnView bn, GraphCsr g, int T, float* __restrict__ g2,
                                                                    double* gsums, int cstride) {
  __shared__ float4 cf[4 * SLN_CTAB_ROW];
  const int C = 2 * H + D;
  const int c0 = blockIdx.x * 4 * XT;
  fill_coef_table<XT, YT>(cf, bn, c0, C, 0);
  __syncthreads();
  const int c = c0 + 4 * threadIdx.x;
  const bool cv = c < C;
  const int part = c < H ? 0 : (c < H + D ? 1 : 2);
  const float4* kk = cf + threadIdx.x;                 // planar: column q at [q * SLN_CTAB_ROW]
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  if (cv) {
   for (int it = 0; it < NIT; ++it) {
    const int t0 = ((blockIdx.y * NIT + it) * YT + threadIdx.y) * RPT;
    if (t0 >= T) break;
    // Every load of the pass is unconditional and its address a select (profiles / ISA of round 3: with `part == 1 ? .. : ..`
    // BRANCHES around the node, gradient and weight loads hipcc waited for each row's loads before it issued the next row's -
    // eight dependent memory round trips per pass of four rows, most of the kernel's 12.6 us).  A lane of the predicate part reads
    // the object index and a pooled row it does not use; its values are dropped by the selects below.
    int node[RPT]; float4 d[RPT], x[RPT]; float w[RPT];
    const int* np = part == 0 ? g.s : g.o;
#pragma unroll
    for (int r = 0; r < RPT; ++r) { const int t = min(t0 + r, T - 1); node[r] = np[t]; x[r] = ld4g(A2 + (size_t)t * ld + c); }
    const bool pred = part == 1;
    const float* dpb = dP ? dP + dpcol0 + (c - H) : A2 + c;          // dP == nullptr: any valid address, the value is replaced by zero
    const int dpld = dP ? lddp : ld;
    const float* dmb = dM + (part == 0 ? c : c - H - D);
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const int t = min(t0 + r, T - 1);
      const float* src = pred ? dpb + (size_t)t * dpld : dmb + (size_t)node[r] * H;
      d[r] = ld4g(src);
      w[r] = g.invdeg[node[r]];
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      if (pred) { w[r] = 1.f; if (!dP) d[r] = make_float4(0.f, 0.f, 0.f, 0.f); }
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      if (t0 + r < T) {
        float4 gv;
        gv.x = fmaf(kk[0].x, x[r].x, kk[0].y) > 0.f ? d[r].x * w[r] : 0.f; gv.y = fmaf(kk[1 * SLN_CTAB_ROW].x, x[r].y, kk[1 * SLN_CTAB_ROW].y) > 0.f ? d[r].y * w[r] : 0.f;
        gv.z = fmaf(kk[2 * SLN_CTAB_ROW].x, x[r].z, kk[2 * SLN_CTAB_ROW].y) > 0.f ? d[r].z * w[r] : 0.f; gv.w = fmaf(kk[3 * SLN_CTAB_ROW].x, x[r].w, kk[3 * SLN_CTAB_ROW].y) > 0.f ? d[r].w * w[r] : 0.f;
        st4g(g2 + (size_t)(t0 + r) * ld + c, gv);
        s1.x += gv.x; s1.y += gv.y; s1.z += gv.z; s1.w += gv.w;
        s2.x = fmaf(gv.x, (x[r].x - kk[0].z) * kk[0].w, s2.x); s2.y = fmaf(gv.y, (x[r].y - kk[1 * SLN_CTAB_ROW].z) * kk[1 * SLN_CTAB_ROW].w, s2.y);
        s2.z = fmaf(gv.z, (x[r].z - kk[2 * SLN_CTAB_ROW].z) * kk[2 * SLN_CTAB_ROW].w, s2.z); s2.w = fmaf(gv.w, (x[r].w - kk[3 * SLN_CTAB_ROW].z) * kk[3 * SLN_CTAB_ROW].w, s2.w);
      }
    }
   }
  }
  commit_col_stats4<XT, YT>(s1, s2, min(4 * XT, C - c0), gsums, cstride, c0);
}
template <int XT, int YT, int RPT, int NIT>
__global__ __launch_bounds__(XT* YT) void scatter_avg_bwd_v4_kernel(const float* __restrict__ dM, const float* __restrict__ dP, int lddp, int dpcol0, const float* __restrict__ A2, int ld, int H, int D,
                                                                    BnView bn, GraphCsr g, int T, float* __restrict__ g2, double* gsums, int cstride) {
  scatter_avg_bwd_v4_body<XT, YT, RPT, NIT>(dM, dP, lddp, dpcol0, A2, ld, H, D, bn, g, T, g2, gsums, cstride);
}

template <int XT, int YT>
__device__ __forceinline__ void gather_bwd_v4_body(const float* __restrict__ dG, int ldg, int D, GraphCsr g, int O, int T,
                                                               const float* __restrict__ add1, int ldadd1,
                                                               const float* __restrict__ xprev, int ldx, BnView bn, int masked,
                                                               float* __restrict__ out, int ldo, double* gsums, int cstride) {
  __shared__ float4 cf[4 * SLN_CTAB_ROW];
  __shared__ int ents[YT][ESTRIDE];
  const int c0 = blockIdx.x * 4 * XT;
  const int i = blockIdx.y * YT + threadIdx.y;
  int b, e;
  row_bounds(g, i, O, b, e);
  if (masked) fill_coef_table<XT, YT>(cf, bn, c0, D, 0);
  cache_entries<XT, YT>(ents, g, b, e);
  const int c = c0 + 4 * threadIdx.x;
  const bool cv = c < D && i < O;
  const float4* kk = cf + threadIdx.x;                 // planar: column q at [q * SLN_CTAB_ROW]
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  if (cv) {
    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 xp = d, ad = d;
    if (masked) xp = ld4g(xprev + (size_t)i * ldx + c);
    if (add1) ad = ld4g(add1 + (size_t)i * ldadd1 + c);
    const int deg = e - b;
    for (int k = 0; k < deg; k += EB) {
      int en[EB]; float4 x[EB];
      if (deg <= ECACHE) {           // see scatter_avg_fwd_v4_kernel
#pragma unroll
        for (int u = 0; u < EB; ++u) en[u] = ents[threadIdx.y][min(k + u, deg - 1)];
      } else {
#pragma unroll
        for (int u = 0; u < EB; ++u) { const int q = min(k + u, deg - 1); en[u] = q < ECACHE ? ents[threadIdx.y][q] : g.ent[b + q]; }
      }
#pragma unroll
      for (int u = 0; u < EB; ++u) {
        const bool isobj = en[u] >= T;
        x[u] = ld4g(dG + (size_t)(isobj ? en[u] - T : en[u]) * ldg + (isobj ? 2 * D : 0) + c);
      }
#pragma unroll
      for (int u = 0; u < EB; ++u)
        if (k + u < deg) { d.x += x[u].x; d.y += x[u].y; d.z += x[u].z; d.w += x[u].w; }
    }
    if (add1) { d.x += ad.x; d.y += ad.y; d.z += ad.z; d.w += ad.w; }
    if (masked) {
      d.x = fmaf(kk[0].x, xp.x, kk[0].y) > 0.f ? d.x : 0.f; d.y = fmaf(kk[1 * SLN_CTAB_ROW].x, xp.y, kk[1 * SLN_CTAB_ROW].y) > 0.f ? d.y : 0.f;
      d.z = fmaf(kk[2 * SLN_CTAB_ROW].x, xp.z, kk[2 * SLN_CTAB_ROW].y) > 0.f ? d.z : 0.f; d.w = fmaf(kk[3 * SLN_CTAB_ROW].x, xp.w, kk[3 * SLN_CTAB_ROW].y) > 0.f ? d.w : 0.f;
      s1 = d;
      s2.x = d.x * ((xp.x - kk[0].z) * kk[0].w); s2.y = d.y * ((xp.y - kk[1 * SLN_CTAB_ROW].z) * kk[1 * SLN_CTAB_ROW].w);
      s2.z = d.z * ((xp.z - kk[2 * SLN_CTAB_ROW].z) * kk[2 * SLN_CTAB_ROW].w); s2.w = d.w * ((xp.w - kk[3 * SLN_CTAB_ROW].z) * kk[3 * SLN_CTAB_ROW].w);
    }
    st4g(out + (size_t)i * ldo + c, d);
  }
  if (masked) commit_col_stats4<XT, YT>(s1, s2, min(4 * XT, D - c0), gsums, cstride, c0);
}
template <int XT, int YT>
__global__ __launch_bounds__(XT* YT) void gather_bwd_v4_kernel(const float* __restrict__ dG, int ldg, int D, GraphCsr g, int O, int T, const float* __restrict__ add1, int ldadd1,
                                                               const float* __restrict__ xprev, int ldx, BnView bn, int masked, float* __restrict__ out, int ldo,
                                                               double* gsums, int cstride) {
  gather_bwd_v4_body<XT, YT>(dG, ldg, D, g, O, T, add1, ldadd1, xprev, ldx, bn, masked, out, ldo, gsums, cstride);
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

__device__ __forceinline__ void mask_gstats_body(const float* __restrict__ d1, int ld1,
                                                             const float* __restrict__ d2, int ld2,
                                                             const float* __restrict__ xprev, int ldx, BnView bn,
                                                             int rows, int cols, float* __restrict__ out, int ldo,
                                                             double* gsums, int cstride) {
  const int c = blockIdx.x * CB + threadIdx.x;
  const bool cv = c < cols;
  float sc = 1.f, sh = 0.f, mean = 0.f, istd = 1.f;
  if (cv) { const float4 k4 = bn_fwd_coef4(bn, c); sc = k4.x; sh = k4.y; mean = k4.z; istd = k4.w; }
  float s1 = 0.f, s2 = 0.f;
  const int r1 = min(rows, (int)(blockIdx.y + 1) * RB);
  if (cv) {
    // the eight rows of a thread: all their loads first (clamped rows, a uniform select instead of the `if (d2)` branch around the
    // second load) - row by row this loop was eight dependent memory round trips, half of the kernel's 9 us.  Same order of sums.
    constexpr int NI = RB / RL;
    const float* d2p = d2 ? d2 : d1;
    const int ld2p = d2 ? ld2 : ld1;
    float dv[NI], ev[NI], xv[NI];
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      const int r = min((int)blockIdx.y * RB + (int)threadIdx.y + u * RL, rows - 1);
      dv[u] = d1[(size_t)r * ld1 + c]; ev[u] = d2p[(size_t)r * ld2p + c]; xv[u] = xprev[(size_t)r * ldx + c];
    }
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      const int r = blockIdx.y * RB + threadIdx.y + u * RL;
      if (r < r1) {
        float d = d2 ? dv[u] + ev[u] : dv[u];
        const float x = xv[u];
        d = fmaf(sc, x, sh) > 0.f ? d : 0.f;
        out[(size_t)r * ldo + c] = d;
        s1 += d; s2 = fmaf(d, (x - mean) * istd, s2);
      }
    }
  }
  commit_col_stats(s1, s2, cv, gsums, cstride, c);
}
__global__ __launch_bounds__(CB* RL) void mask_gstats_kernel(const float* __restrict__ d1, int ld1, const float* __restrict__ d2, int ld2, const float* __restrict__ xprev, int ldx, BnView bn,
                                                             int rows, int cols, float* __restrict__ out, int ldo, double* gsums, int cstride) {
  mask_gstats_body(d1, ld1, d2, ld2, xprev, ldx, bn, rows, cols, out, ldo, gsums, cstride);
}

// ----------------------------------------------------------------------------------------------
// embeddings
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ void enc_assemble_body(const EncAssemble& a, long bid) {
  const int W = a.n_obj + a.n_attr + a.n_box + a.n_angle;
  const long idx = bid * blockDim.x + threadIdx.x;
  if (idx >= (long)a.O * W) return;
  const int r = (int)(idx / W);
  int c = (int)(idx % W);
  float v;
  if (c < a.n_obj) v = a.obj_emb[(size_t)a.objs[r] * a.n_obj + c];
  else if ((c -= a.n_obj) < a.n_attr) v = a.attr_emb[(size_t)a.attrs[r] * a.n_attr + c];
  else if ((c -= a.n_attr) < a.n_box) {
    v = a.bb[c];
    for (int k = 0; k < a.box_dim; ++k) v = fmaf(a.boxes[(size_t)r * a.box_dim + k], a.wb[c * a.box_dim + k], v);
  } else { c -= a.n_box; v = a.angle_emb[(size_t)a.angles[r] * a.n_angle + c]; }
  a.x0[idx] = v;
}
__global__ void enc_assemble_kernel(EncAssemble a) { enc_assemble_body(a, blockIdx.x); }

__global__ void enc_assemble_bwd_kernel(EncAssembleBwd a) {
  // embedding parts: atomics straight into the (small) tables
  const int W = a.n_obj + a.n_attr + a.n_box + a.n_angle;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)a.O * W) return;
  const int r = (int)(idx / W);
  int c = (int)(idx % W);
  const float d = a.dx0[idx];
  if (c < a.n_obj) atomicAdd(a.d_obj_emb + (size_t)a.objs[r] * a.n_obj + c, d);
  else if ((c -= a.n_obj) < a.n_attr) atomicAdd(a.d_attr_emb + (size_t)a.attrs[r] * a.n_attr + c, d);
  else if ((c -= a.n_attr) < a.n_box) { /* Linear(box_dim -> n_box): box_embed_bwd_kernel */ }
  else { c -= a.n_box; atomicAdd(a.d_angle_emb + (size_t)a.angles[r] * a.n_angle + c, d); }
}

__global__ __launch_bounds__(CB* RL) void box_embed_bwd_kernel(EncAssembleBwd a, const int box_rows) {
  // d_bb[j] += sum_r d[r,j];  d_wb[j,k] += sum_r d[r,j]*boxes[r,k]   (box_dim <= 6)
  const int W = a.n_obj + a.n_attr + a.n_box + a.n_angle, off = a.n_obj + a.n_attr;
  const int j = blockIdx.x * CB + threadIdx.x;
  const bool jv = j < a.n_box;
  float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // BOX_ROWS rows per block in batches of 16 (4 per thread, their loads issued together: one memory round trip per batch).  Every
  // block ends in 7 atomics per column on the same 7 x n_box addresses, which the L2 serialises - measured at 2 048 rows: 64 rows
  // walked one at a time 20.6 us; batches, 16 rows x 128 blocks 21.6 us, 256 x 8 15.9 us, 64 x 32 9.8 us
  // (deterministic mode: box_rows = O, one workgroup per column group walks every row - one add per element)
  const int rb0 = (int)blockIdx.y * box_rows, r1 = min(a.O, rb0 + box_rows);
  if (jv) {
    for (int rb = rb0; rb < r1; rb += 4 * RL) {
      float d[4], bx[4][6];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int r = min(rb + (int)threadIdx.y + RL * u, a.O - 1);
        d[u] = a.dx0[(size_t)r * W + off + j];
#pragma unroll
        for (int k = 0; k < 6; ++k) bx[u][k] = a.boxes[(size_t)r * a.box_dim + min(k, a.box_dim - 1)];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (rb + (int)threadIdx.y + RL * u >= r1) continue;
        acc[6] += d[u];
#pragma unroll
        for (int k = 0; k < 6; ++k) if (k < a.box_dim) acc[k] = fmaf(d[u], bx[u][k], acc[k]);
      }
    }
  }
  __shared__ float red[7][RL][CB];
  for (int k = 0; k < 7; ++k) red[k][threadIdx.y][threadIdx.x] = acc[k];
  __syncthreads();
  if (threadIdx.y == 0 && jv) {
    for (int k = 0; k < 7; ++k) {
      float s = 0.f;
      for (int i = 0; i < RL; ++i) s += red[k][i][threadIdx.x];
      if (k == 6) atomicAdd(a.d_bb + j, s);
      else if (k < a.box_dim) atomicAdd(a.d_wb + j * a.box_dim + k, s);
    }
  }
}

__device__ __forceinline__ void dec_assemble_body(DecAssemble a) {
  const int W = a.n_obj + a.n_attr + a.n_z;
  const int Wx = a.z_in_x0 ? W : a.n_obj + a.n_attr;      // row stride of x0
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)a.O * W) return;
  const int r = (int)(idx / W);
  int c = (int)(idx % W);
  const int cx = c;
  float v;
  if (c < a.n_obj) v = a.obj_emb[(size_t)a.objs[r] * a.n_obj + c];
  else if ((c -= a.n_obj) < a.n_attr) v = a.attr_emb[(size_t)a.attrs[r] * a.n_attr + c];
  else {
    c -= a.n_attr;
    const size_t zi = (size_t)r * a.n_z + c;
    if (a.z_in) v = a.z_in[zi];
    else if (a.use_ae) v = a.mu[zi];
    else v = a.eps[zi] * expf(0.5f * a.logvar[zi]) + a.mu[zi];     // Sg2ScVAE_model.py:180-183
    if (a.z) a.z[zi] = v;
    if (!a.z_in_x0) return;
  }
  a.x0[(size_t)r * Wx + cx] = v;
}
__global__ void dec_assemble_kernel(DecAssemble a) {
  dec_assemble_body(a);
}

__device__ __forceinline__ void dec_assemble_bwd_body(DecAssembleBwd a) {
  const int W = a.n_obj + a.n_attr + (a.z_in_x0 ? a.n_z : 0);
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)a.O * W) return;
  const int r = (int)(idx / W);
  int c = (int)(idx % W);
  const float d = a.dx0[idx];
  if (c < a.n_obj) atomicAdd(a.d_obj_emb + (size_t)a.objs[r] * a.n_obj + c, d);
  else if ((c -= a.n_obj) < a.n_attr) atomicAdd(a.d_attr_emb + (size_t)a.attrs[r] * a.n_attr + c, d);
  else { c -= a.n_attr; if (a.dz) a.dz[(size_t)r * a.n_z + c] = d; }
}
__global__ void dec_assemble_bwd_kernel(DecAssembleBwd a) {
  dec_assemble_bwd_body(a);
}

// Small tables (<= 8192 floats): accumulate the block's rows in an LDS copy of the table, then flush the
// touched entries with one global atomic each (pred table: 4096x128 adds onto 16x128 entries).
template <typename IdxT>
__device__ __forceinline__ void embed_bwd_lds_body(const IdxT* __restrict__ idx, const float* __restrict__ d, int ld,
                                                            int col0, int rows, int n, int table_rows, int rows_per_block,
                                                            float* __restrict__ d_emb) {
  extern __shared__ float tab[];
  const int tsz = table_rows * n;
  for (int i = threadIdx.x; i < tsz; i += 256) tab[i] = 0.f;
  __syncthreads();
  const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  const long iend = (long)r1 * n;
  for (long i = (long)r0 * n + threadIdx.x; i < iend; i += 256 * 8) {       // 8 independent loads per thread before the LDS atomics
    float v[8]; int slot[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long q = min(i + 256L * u, iend - 1);
      const int r = (int)(q / n), c = (int)(q % n);
      slot[u] = (int)idx[r] * n + c;
      v[u] = d[(size_t)r * ld + col0 + c];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i + 256L * u < iend) atomicAdd(&tab[slot[u]], v[u]);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < tsz; i += 256) {
    const float v = tab[i];
    if (v != 0.f) atomicAdd(d_emb + i, v);
  }
}
template <typename IdxT>
__global__ __launch_bounds__(256) void embed_bwd_lds_kernel(const IdxT* __restrict__ idx, const float* __restrict__ d, int ld, int col0, int rows, int n, int table_rows, int rows_per_block,
                                                            float* __restrict__ d_emb) {
  embed_bwd_lds_body<IdxT>(idx, d, ld, col0, rows, n, table_rows, rows_per_block, d_emb);
}

// Deterministic form (SLN_DETERMINISTIC): one workgroup per (table row e, 64 columns).  Its four row lanes walk the source rows
// in order and keep those with idx[r] == e; the four partial sums meet in a fixed order; the result is added to the table with a
// plain read-modify-write (one writer per element and launch).  No atomics, no arrival order.
template <typename IdxT>
__device__ __forceinline__ void embed_bwd_det_body(const IdxT* __restrict__ idx, const float* __restrict__ d, int ld, int col0,
                                                             int rows, int n, float* __restrict__ d_emb) {
  // 16 row lanes x 64 columns; a row lane takes rows lane_r, lane_r + 16, ... eight at a time (their indices in one round trip,
  // then the matching rows' values: the walk is latency-bound), always in ascending order
  const int e = blockIdx.x, x = threadIdx.x & 63, c = blockIdx.y * 64 + x, lane_r = threadIdx.x >> 6;
  float acc = 0.f;
  if (c < n)
    for (int r0 = lane_r; r0 < rows; r0 += 16 * 8) {
      int id[8]; float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) id[u] = (int)idx[min(r0 + 16 * u, rows - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = r0 + 16 * u;
        v[u] = (r < rows && id[u] == e) ? d[(size_t)r * ld + col0 + c] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];          // adding 0.f for the other rows changes nothing
    }
  __shared__ float red[16][64];
  red[lane_r][x] = acc;
  __syncthreads();
  if (lane_r == 0 && c < n) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += red[i][x];
    d_emb[(size_t)e * n + c] += s;
  }
}
template <typename IdxT>
__global__ __launch_bounds__(1024) void embed_bwd_det_kernel(const IdxT* __restrict__ idx, const float* __restrict__ d, int ld, int col0, int rows, int n, float* __restrict__ d_emb) {
  embed_bwd_det_body<IdxT>(idx, d, ld, col0, rows, n, d_emb);
}

template <typename IdxT>
__device__ __forceinline__ void embed_bwd_body(const IdxT* __restrict__ idx, const float* __restrict__ d, int ld, int col0, int rows,
                                 int n, float* __restrict__ d_emb) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * n) return;
  const int r = (int)(i / n), c = (int)(i % n);
  atomicAdd(d_emb + (size_t)idx[r] * n + c, d[(size_t)r * ld + col0 + c]);
}
template <typename IdxT>
__global__ void embed_bwd_kernel(const IdxT* __restrict__ idx, const float* __restrict__ d, int ld, int col0, int rows, int n, float* __restrict__ d_emb) {
  embed_bwd_body<IdxT>(idx, d, ld, col0, rows, n, d_emb);
}

__device__ __forceinline__ void embed_gather_body(const int* __restrict__ idx, const float* __restrict__ emb, int rows, int n,
                                                  float* __restrict__ out, long bid) {
  const long i = bid * blockDim.x + threadIdx.x;
  if (i >= (long)rows * n) return;
  const int r = (int)(i / n), c = (int)(i % n);
  out[i] = emb[(size_t)idx[r] * n + c];
}
__global__ void embed_gather_kernel(const int* __restrict__ idx, const float* __restrict__ emb, int rows, int n,
                                    float* __restrict__ out) {
  embed_gather_body(idx, emb, rows, n, out, blockIdx.x);
}

__global__ void i64_to_i32_kernel(const int64_t* __restrict__ src, int* __restrict__ dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (int)src[i];
}

// ----------------------------------------------------------------------------------------------
// loss
// ----------------------------------------------------------------------------------------------
__global__ void log_softmax_bwd_kernel(const float* __restrict__ lp, const float* __restrict__ dlp,
                                       float* __restrict__ dx, int O, int n) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= O) return;
  float s = 0.f;
  for (int k = 0; k < n; ++k) s += dlp[(size_t)r * n + k];
  for (int k = 0; k < n; ++k) dx[(size_t)r * n + k] = dlp[(size_t)r * n + k] - expf(lp[(size_t)r * n + k]) * s;
}

__global__ void log_softmax_kernel(const float* __restrict__ x, float* __restrict__ y, int O, int n) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= O) return;
  const float* xr = x + (size_t)r * n;
  float m = xr[0];
  for (int k = 1; k < n; ++k) m = fmaxf(m, xr[k]);
  float s = 0.f;
  for (int k = 0; k < n; ++k) s += expf(xr[k] - m);
  const float ls = logf(s) + m;
  for (int k = 0; k < n; ++k) y[(size_t)r * n + k] = xr[k] - ls;
}

// element-parallel (coalesced) over the three parts of the loss; LOSS_BLOCKS blocks stride over the elements so that the
// three fp64 accumulators see few same-address atomics
constexpr int LOSS_BLOCKS = 64;
// Round 3: ONE launch.  (i) With `from_logits` the row-wise log_softmax (Sg2ScVAE_model.py:171) is taken here, one thread per
// object row, instead of by a launch of its own in front; (ii) the last block to arrive (ticket in the unused fourth accumulator,
// cleared with the others) turns the three sums into the four loss values - the single-thread finalize launch is gone.
__global__ __launch_bounds__(256) void loss_kernel(LossArgs a) {
  const long stride = (long)gridDim.x * 256;
  const long i0 = (long)blockIdx.x * 256 + threadIdx.x;
  double l1 = 0.0, nll = 0.0, kl = 0.0;
  const float gb = 1.0f / ((float)a.O * (float)a.box_dim), go = 1.0f / (float)a.O;
  for (long i = i0; i < (long)a.O * a.box_dim; i += stride) {
    const int r = (int)(i / a.box_dim), k = (int)(i % a.box_dim);
    const float diff = a.boxes_pred[i] - a.boxes[i];
    l1 += fabsf(diff);
    if (a.d_boxes_pred) a.d_boxes_pred[(size_t)r * a.ld_dbp + k] = diff > 0.f ? gb : (diff < 0.f ? -gb : 0.f);
  }
  if (a.from_logits) {
    // eight lanes per object row (64 blocks x 256 threads = 8 x 2048 rows at 64 graphs): bins sub, sub + 8, .. per lane, the row
    // maximum and the sum of exponentials meet through three xor-shuffles (every lane of a group runs the same trip count)
    const int sub = threadIdx.x & 7;
    for (long r = i0 >> 3; r < (((long)a.O + 31) & ~31L); r += stride >> 3) {
      const bool rv = r < a.O;
      const float* xr = a.logits + (size_t)(rv ? r : 0) * a.n_angle;
      const int tgt = (int)a.angles[rv ? r : 0];              // requested with the row, not behind the two reductions
      float m = -INFINITY;
      for (int k = sub; k < a.n_angle; k += 8) m = fmaxf(m, xr[k]);
      m = fmaxf(m, __shfl_xor(m, 1, 64)); m = fmaxf(m, __shfl_xor(m, 2, 64)); m = fmaxf(m, __shfl_xor(m, 4, 64));
      float s = 0.f;
      for (int k = sub; k < a.n_angle; k += 8) s += expf(xr[k] - m);
      s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
      const float ls = logf(s) + m;
      if (rv) {
        for (int k = sub; k < a.n_angle; k += 8) {
          const float lp = xr[k] - ls;
          a.angles_pred[(size_t)r * a.n_angle + k] = lp;
          if (k == tgt) nll -= (double)lp;
          if (a.d_logits) a.d_logits[(size_t)r * a.n_angle + k] = (expf(lp) - (k == tgt ? 1.f : 0.f)) * go;
        }
      }
    }
  } else {
    for (long i = i0; i < (long)a.O * a.n_angle; i += stride) {
      const int r = (int)(i / a.n_angle), k = (int)(i % a.n_angle);
      const int tgt = (int)a.angles[r];
      const float lp = a.angles_pred[i];
      if (k == tgt) nll -= (double)lp;
      if (a.d_logits) a.d_logits[i] = (expf(lp) - (k == tgt ? 1.f : 0.f)) * go;
    }
  }
  if (!a.use_ae) {
    // four elements of a thread per trip, their loads first (clamped, the surplus ones add zero): element by element every trip of
    // this loop was a memory round trip of its own - eight in a row at 64 graphs, the longest phase of the kernel.  Same order of
    // additions as before.
    float s = 0.f;
    const long nkl = (long)a.O * a.n_z;
    for (long i = i0; i < nkl; i += 4 * stride) {
      float m[4], lv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const long j = min(i + u * stride, nkl - 1); m[u] = a.mu[j]; lv[u] = a.logvar[j]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float term = 1.f + lv[u] - m[u] * m[u] - expf(lv[u]);
        s += (i + u * stride < nkl) ? term : 0.f;
      }
    }
    kl = s;
  }
  __shared__ double red[3][256];
  red[0][threadIdx.x] = l1; red[1][threadIdx.x] = nll; red[2][threadIdx.x] = kl;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off)
      for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x < 3) atomicAdd(a.acc + threadIdx.x, sln_qd(red[threadIdx.x][0], SLN_Q_LOSS));
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int t = atomicAdd(reinterpret_cast<unsigned int*>(a.acc + 3), 1u);
    if (t == gridDim.x - 1) {                            // every other block's sums are in (their fence precedes their ticket)
      __threadfence();
      const double s0 = atomicAdd(a.acc + 0, 0.0), s1 = atomicAdd(a.acc + 1, 0.0), s2 = atomicAdd(a.acc + 2, 0.0);
      const double O = (double)a.O;
      const float lb = (float)(s0 / (O * a.box_dim));
      const float la = (float)(s1 / O);
      float lk = 0.f;
      if (!a.use_ae) lk = (float)(-0.5 * s2 / O) * a.kl_weight[0];
      a.losses[0] = lb; a.losses[1] = la; a.losses[2] = lk; a.losses[3] = lb + la + lk;
    }
  }
}

__global__ void latent_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv,
                                  const float* __restrict__ eps, const float* __restrict__ dz,
                                  const float* __restrict__ klw, int O, int nz, int use_ae, float* __restrict__ dmu,
                                  float* __restrict__ dlv) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)O * nz) return;
  const float g = dz[i];
  if (use_ae) { dmu[i] = g; dlv[i] = 0.f; return; }
  const float w = klw[0] / (float)O;
  const float l = lv[i];
  dmu[i] = fmaf(w, mu[i], g);
  dlv[i] = w * 0.5f * (expf(l) - 1.f) + g * eps[i] * 0.5f * expf(0.5f * l);
}

// ----------------------------------------------------------------------------------------------
// BatchNorm bookkeeping, transposes, Adam
// ----------------------------------------------------------------------------------------------
// rows_t / rows_o: the batch's triple / object counts; a table entry with rows == -1 / -2 means "the triples" / "the objects" (the
// engine's table then does not change with the batch's shape: no upload per step on real rooms), rows > 0 is taken as it is
__global__ void bn_running_update_kernel(const BnTableEntry* __restrict__ tab, int n, float mom, int per_entry, int rows_t, int rows_o) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int e0 = per_entry ? blockIdx.y : 0, e1 = per_entry ? blockIdx.y + 1 : n;
  for (int e = e0; e < e1; ++e) {             // sequential form: application order, shared modules see ordered updates
    const BnTableEntry t = tab[e];
    if (c == 0 && t.nbt) t.nbt[0] += 1;
    if (c >= t.C || t.rmean == nullptr) continue;
    const int rows = t.rows == -1 ? rows_t : (t.rows == -2 ? rows_o : t.rows);
    const double N = (double)rows;
    const double m = t.sums[c] / N;
    double v = t.sums[t.cstride + c] / N - m * m;
    v = v < 0.0 ? 0.0 : v;
    const double vu = rows > 1 ? v * N / (N - 1.0) : v;
    t.rmean[c] = (1.f - mom) * t.rmean[c] + mom * (float)m;
    t.rvar[c] = (1.f - mom) * t.rvar[c] + mom * (float)vu;
  }
}

__global__ void bn_param_grads_kernel(const BnTableEntry* __restrict__ tab, int n, int per_entry) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int e0 = per_entry ? blockIdx.y : 0, e1 = per_entry ? blockIdx.y + 1 : n;
  for (int e = e0; e < e1; ++e) {
    const BnTableEntry t = tab[e];
    if (c >= t.C || t.dgamma == nullptr) continue;
    t.dbeta[c] += (float)t.gsums[c];
    t.dgamma[c] += (float)t.gsums[t.cstride + c];
  }
}

__global__ __launch_bounds__(256) void transpose_table_kernel(const TransposeEntry* __restrict__ tab) {
  __shared__ float tile[32][33];
  const TransposeEntry t = tab[blockIdx.y];
  const int tr = (t.rows + 31) / 32, tc = (t.cols + 31) / 32;
  if ((int)blockIdx.x >= tr * tc) return;
  const int r0 = (blockIdx.x / tc) * 32, c0 = (blockIdx.x % tc) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < t.rows && c < t.cols) ? t.src[(size_t)r * t.cols + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (r < t.rows && c < t.cols) t.dst[(size_t)c * t.dst_ld + r] = tile[tx][i];
  }
}

// torch.optim.Adam defaults (train.py:15): no weight decay, no amsgrad.  train.py:79-81: a non-finite total loss is reported and
// the iteration is skipped (no optimizer step, the step count stays).  Round 3: the scalar bookkeeping (step + 1, the two bias
// corrections, the skip decision) is taken by every block from the OLD scalars instead of by a one-thread launch in front; the last
// block to finish commits the new step (every other block has read the old one by then: it arrived before).
// Round 3, last session (tools/lab/adam_lab.hip, 3.7 M parameters, operands cold as in the step): the update itself streams at
// 4.7 TB/s (22 us) - the kernel took 39 us because its 2 048 workgroups each added 1 to the SAME arrival counter, and same-address
// atomics are served one at a time (~6 ns each): 27 us without the ticket.  Now 256 workgroups of 1 024 threads (256 arrivals),
// 16-byte accesses, nontemporal for the gradient and the two moments (touched once per step; 18 us in the lab with everything
// nontemporal - but the parameters are the next step's weights and stay cached); the per-thread pow() calls, suspected first, cost
// nothing measurable.
typedef float adam_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void adam_update(float& p, const float g, float& m, float& v, const float b1, const float b2, const float step_size, const float rs2,
                                            const float eps) {
  const float mi = b1 * m + (1.f - b1) * g;
  const float vi = b2 * v + (1.f - b2) * g * g;
  m = mi; v = vi;
  p -= step_size * mi / (sqrtf(vi) * rs2 + eps);
}
__global__ __launch_bounds__(1024) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, int vec, AdamScalars* __restrict__ sc,
                                                    const float* __restrict__ total_loss) {
  const bool skip = total_loss != nullptr && !isfinite(total_loss[0]);
  const int64_t step = sc->step + 1;
  const float b1 = sc->beta1, b2 = sc->beta2, eps = sc->eps;
  const float bc1 = (float)(1.0 - pow((double)b1, (double)step)), bc2 = (float)(1.0 - pow((double)b2, (double)step));
  const float step_size = sc->lr / bc1, rs2 = 1.0f / sqrtf(bc2);
  if (!skip) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
    const long n4 = vec ? n / 4 : 0;                      // vec: all four arrays 16-byte aligned (checked by the launcher)
    adam_v4f* p4 = reinterpret_cast<adam_v4f*>(p); const adam_v4f* g4 = reinterpret_cast<const adam_v4f*>(g);
    adam_v4f* m4 = reinterpret_cast<adam_v4f*>(m); adam_v4f* v4 = reinterpret_cast<adam_v4f*>(v);
    for (long i = tid; i < n4; i += stride) {
      adam_v4f pi = p4[i], mi = __builtin_nontemporal_load(m4 + i), vi = __builtin_nontemporal_load(v4 + i);
      const adam_v4f gi = __builtin_nontemporal_load(g4 + i);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float pk = pi[k], mk = mi[k], vk = vi[k];
        adam_update(pk, gi[k], mk, vk, b1, b2, step_size, rs2, eps);
        pi[k] = pk; mi[k] = mk; vi[k] = vk;
      }
      __builtin_nontemporal_store(mi, m4 + i); __builtin_nontemporal_store(vi, v4 + i);
      p4[i] = pi;                                         // cached: the next step's first GEMMs read the weights (nontemporal: +15 us there)
    }
    for (long i = 4 * n4 + tid; i < n; i += stride) {      // the tail (or everything, for unaligned arrays)
      float pk = p[i], mk = m[i], vk = v[i];
      adam_update(pk, g[i], mk, vk, b1, b2, step_size, rs2, eps);
      m[i] = mk; v[i] = vk; p[i] = pk;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int done = atomicAdd(&sc->adam_done, 1u);
    if (done == gridDim.x - 1) {
      sc->adam_done = 0; sc->skip = skip ? 1 : 0;
      if (!skip) { sc->step = step; sc->bc1 = bc1; sc->bc2 = bc2; }
    }
  }
}

// Philox-4x32-10 (Salmon et al., SC'11), the generator torch.randn uses on the device; Box-Muller on the four words.
__device__ __forceinline__ void philox_round(unsigned int (&c)[4], unsigned int k0, unsigned int k1) {
  const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
  const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned int)p1;
  const unsigned int n2 = (unsigned int)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned int)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__device__ __forceinline__ void randn_body(float* __restrict__ eps, long n, AdamScalars* sc, long bid, unsigned int nblocks) {
  const unsigned long long seed = sc->rng_seed, off = sc->rng_offset;
  const long q = bid * 256 + threadIdx.x;
  if (4 * q < n) {
    unsigned int c[4] = {(unsigned int)q, (unsigned int)((unsigned long long)q >> 32), (unsigned int)off, (unsigned int)(off >> 32)};
    unsigned int k0 = (unsigned int)seed, k1 = (unsigned int)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    float v[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float u1 = ((float)c[2 * h] + 1.0f) * 2.3283064365386963e-10f;          // (0, 1]
      const float u2 = (float)c[2 * h + 1] * 2.3283064365386963e-10f;               // [0, 1]
      const float rad = sqrtf(-2.0f * logf(u1));
      float sn, cs;
      sincosf(6.283185307179586f * u2, &sn, &cs);
      v[2 * h] = rad * cs; v[2 * h + 1] = rad * sn;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (4 * q + k < n) eps[4 * q + k] = v[k];
  }
  // every block has read the offset before it gets here; the last one to arrive advances it
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int done = atomicAdd(&sc->rng_done, 1u);
    if (done == nblocks - 1) { sc->rng_done = 0; sc->rng_offset = off + 1; }
  }
}
__global__ __launch_bounds__(256) void randn_kernel(float* __restrict__ eps, long n, AdamScalars* sc) {
  randn_body(eps, n, sc, blockIdx.x, gridDim.x);
}

// blocks [0, b0) draw eps, [b0, b1) assemble the encoder input, [b1, b2) / [b2, ..) gather the two predicate embeddings
__global__ __launch_bounds__(256) void step_prologue_kernel(StepPrologue a, unsigned int b0, unsigned int b1, unsigned int b2, unsigned int b3) {
  const unsigned int b = blockIdx.x;
  if (b < b0) randn_body(a.eps, a.n_eps, a.scalars, b, b0);
  else if (b < b1) enc_assemble_body(a.enc, b - b0);
  else if (b < b2) embed_gather_body(a.pidx, a.pemb_ec, a.T, a.n_ec, a.p0e, b - b1);
  else if (b < b3) embed_gather_body(a.pidx, a.pemb_dc, a.T, a.n_dc, a.p0d, b - b2);
  else {                                  // the iteration's accumulators (one launch less than a memset node of their own)
    const long i = (long)(b - b3) * 256 + threadIdx.x;
    if (i * 16 < a.zero_bytes) reinterpret_cast<uint4*>(a.zero_ptr)[i] = make_uint4(0u, 0u, 0u, 0u);
  }
}

// out[r, c] = relu(bn(x[r, col0 + c]))  (materialise a post-activation, standalone GraphTripleConv API only)
__global__ void bn_relu_apply_kernel(const float* __restrict__ x, int ld, int col0, int cols, long n, BnView bn,
                                     float* __restrict__ out, int ldo) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long r = i / cols; const int c = (int)(i % cols);
  float sc, sh;
  bn_fwd_coef(bn, c, sc, sh);
  out[r * ldo + c] = fmaxf(fmaf(sc, x[r * ld + col0 + c], sh), 0.f);
}

__device__ __forceinline__ void add2_body(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb, int rows, int cols,
                            float* __restrict__ out, int ldo) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i % cols);
  out[(size_t)r * ldo + c] = a[(size_t)r * lda + c] + b[(size_t)r * ldb + c];
}
__global__ void add2_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb, int rows, int cols, float* __restrict__ out, int ldo) {
  add2_body(a, lda, b, ldb, rows, cols, out, ldo);
}

inline dim3 colgrid(int cols, int rows) { return dim3(sln_cdiv(cols, CB), sln_cdiv(rows, RB)); }

}  // namespace

__global__ void validate_ids_kernel(const int64_t* __restrict__ objs, const int64_t* __restrict__ attrs,
                                    const int64_t* __restrict__ angles, int O, int n_objs, int n_attrs, int n_angle, int* err) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= O) return;
  if (objs[i] < 0 || objs[i] >= n_objs) atomicOr(err, 2);
  if (attrs[i] < 0 || attrs[i] >= n_attrs) atomicOr(err, 4);
  if (angles && (angles[i] < 0 || angles[i] >= n_angle)) atomicOr(err, 8);
}

// One launch for everything sln_vae_set_batch does per object row: the engine-owned copies of the inputs, the id checks, the
// int32 attribute ids and the cleared degree counters (six stream operations before).
__global__ void stage_batch_kernel(StageBatch a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.O) return;
  const int64_t ob = a.objs[i], at = a.attrs[i], an = a.angles[i];
  a.st_objs[i] = ob; a.st_attrs[i] = at; a.st_angles[i] = an;
  a.attrs32[i] = (int)at;
  a.deg[i] = 0;
  for (int k = 0; k < a.box_dim; ++k) a.st_boxes[(size_t)i * a.box_dim + k] = a.boxes[(size_t)i * a.box_dim + k];
  if (ob < 0 || ob >= a.n_objs) atomicOr(a.err, 2);
  if (a.n_attrs > 0 && (at < 0 || at >= a.n_attrs)) atomicOr(a.err, 4);       // use_attr off: attributes are not looked at
  if (an < 0 || an >= a.n_angle) atomicOr(a.err, 8);
}

int sln_launch_stage_batch(const StageBatch& a, hipStream_t st) {
  if (a.O <= 0) return 0;
  hipLaunchKernelGGL(stage_batch_kernel, dim3(sln_cdiv(a.O, 256)), dim3(256), 0, st, a);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_launch_validate_ids(const int64_t* objs, const int64_t* attrs, const int64_t* angles, int O, int n_objs, int n_attrs,
                            int n_angle, int* err_flag, hipStream_t st) {
  if (O <= 0) return 0;
  hipLaunchKernelGGL(validate_ids_kernel, dim3(sln_cdiv(O, 256)), dim3(256), 0, st, objs, attrs, angles, O, n_objs, n_attrs, n_angle,
                     err_flag);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_launch_graph_prep(const int64_t* triples, int T, int O, int num_preds, GraphCsr g, int* err_flag, hipStream_t st,
                          int edges_only, int deg_is_zero) {
  if (!deg_is_zero) {
    const int e = sln_zero_async(g.deg, sizeof(int) * (size_t)O, st);
    if (e != 0) return e;
  }
  if (T > 0) hipLaunchKernelGGL(prep_split_kernel, dim3(sln_cdiv(T, 256)), dim3(256), 0, st, triples, T, O, num_preds, g, err_flag,
                              edges_only ? 2 : 3, edges_only ? -1 : 1, edges_only ? 1 : 2);
  hipLaunchKernelGGL(csr_scan_kernel, dim3(1), dim3(1024), 0, st, g, O);
  if (T > 0) hipLaunchKernelGGL(csr_fill_kernel, dim3(sln_cdiv(2 * T, 256)), dim3(256), 0, st, g, T);
  hipLaunchKernelGGL(csr_sort_kernel, dim3(O), dim3(64), 0, st, g, O);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_launch_bn_relu_apply(const float* x, int ld, int col0, int cols, int rows, BnView bn, float* out, int ldo, hipStream_t st) {
  const long n = (long)rows * cols;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(bn_relu_apply_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, ld, col0, cols, n, bn, out, ldo);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_launch_scatter_avg_fwd(const float* A2, int ld, int H, int D, BnView bn2, GraphCsr g, int O, float* pooled,
                               hipStream_t st) {
  if (O <= 0) return 0;
  // lab (SLN_EDGE_ABL: bit 0 this launcher skipped - outputs are then garbage): bounds what folding the edge launches into their
  // neighbours could return at most (round 6, LAB_NOTES)
  { static const int a = std::getenv("SLN_EDGE_ABL") ? std::atoi(std::getenv("SLN_EDGE_ABL")) : 0; if (a & 1) return 0; }
  // algorithmic bytes: both halves of A2 once, pooled once, the entry list
  SlnProfScope prof(SLN_FAM_EDGE, 4.0 * g.T * 2 * H + 4.0 * O * H + 16.0 * g.T, st);
  if (H % 4 == 0 && D % 4 == 0 && ld % 4 == 0 && al16(A2) && al16(pooled)) {
    if (H > 128) hipLaunchKernelGGL((scatter_avg_fwd_v4_kernel<64, 4>), dim3(sln_cdiv(H, 256), sln_cdiv(O, 4)), dim3(64, 4), 0, st, A2, ld, H, D, bn2, g, O, g.T, pooled);
    else hipLaunchKernelGGL((scatter_avg_fwd_v4_kernel<32, 8>), dim3(sln_cdiv(H, 128), sln_cdiv(O, 8)), dim3(32, 8), 0, st, A2, ld, H, D, bn2, g, O, g.T, pooled);
    SLN_CHECK_LAUNCH();
    return 0;
  }
  return -2;   // SLN_E_UNSUPPORTED: misaligned rows (the engine never produces them)
}

int sln_launch_scatter_avg_bwd(const float* dM, const float* dP, int lddp, int dpcol0, const float* A2, int ld, int H,
                               int D, BnView bn2, GraphCsr g, int T, float* g2, double* gsums, int cstride,
                               hipStream_t st) {
  if (T <= 0) return 0;
  { static const int a = std::getenv("SLN_EDGE_ABL") ? std::atoi(std::getenv("SLN_EDGE_ABL")) : 0; if (a & 2) return 0; }      // lab, see sln_launch_scatter_avg_fwd
  SlnProfScope prof(SLN_FAM_EDGE, 4.0 * T * (2 * H) + (dP ? 4.0 * T * D : 0.0) + 2.0 * 4.0 * T * (2 * H + D) + 8.0 * T, st);
  if (H % 4 == 0 && D % 4 == 0 && ld % 4 == 0 && (!dP || (lddp % 4 == 0 && dpcol0 % 4 == 0 && al16(dP))) && al16(dM) && al16(A2) && al16(g2)) {
    constexpr int RPT = 4;
    static const int nit = std::getenv("SLN_SCATTER_NIT") ? std::atoi(std::getenv("SLN_SCATTER_NIT")) : 2;   // lab: passes of 16 rows per workgroup
    if (nit == 4) {
      hipLaunchKernelGGL((scatter_avg_bwd_v4_kernel<64, 4, RPT, 4>), dim3(sln_cdiv(2 * H + D, 256), sln_cdiv(T, 4 * RPT * 4)), dim3(64, 4), 0, st, dM, dP,
                         lddp, dpcol0, A2, ld, H, D, bn2, g, T, g2, gsums, cstride);
      SLN_CHECK_LAUNCH();
      return 0;
    }
    constexpr int NIT = 2;         // 32 rows per block: as many column-statistics atomics as the scalar kernel
    hipLaunchKernelGGL((scatter_avg_bwd_v4_kernel<64, 4, RPT, NIT>), dim3(sln_cdiv(2 * H + D, 256), sln_cdiv(T, 4 * RPT * NIT)), dim3(64, 4), 0, st, dM, dP,
                       lddp, dpcol0, A2, ld, H, D, bn2, g, T, g2, gsums, cstride);
    SLN_CHECK_LAUNCH();
    return 0;
  }
  return -2;   // SLN_E_UNSUPPORTED: misaligned rows (the engine never produces them)
}

int sln_launch_gather_bwd(const float* dG, int ldg, int D, GraphCsr g, int O, const float* add1, int ldadd1,
                          const float* xprev, int ldx, BnView bn, int masked, float* out, int ldo, double* gsums,
                          int cstride, hipStream_t st) {
  if (O <= 0) return 0;
  { static const int a = std::getenv("SLN_EDGE_ABL") ? std::atoi(std::getenv("SLN_EDGE_ABL")) : 0; if (a & 4) return 0; }      // lab, see sln_launch_scatter_avg_fwd
  SlnProfScope prof(SLN_FAM_EDGE, 4.0 * g.T * 2 * D + (masked ? 2.0 : 1.0) * 4.0 * O * D + 16.0 * g.T, st);
  if (D % 4 == 0 && ldg % 4 == 0 && ldo % 4 == 0 && (!add1 || (ldadd1 % 4 == 0 && al16(add1))) && (!masked || (ldx % 4 == 0 && al16(xprev))) &&
      al16(dG) && al16(out)) {
    // 16 rows per workgroup: every workgroup ends in one fp64 atomic per statistics address, same-address atomics are served one
    // at a time, and all workgroups finish together - 256 arrivals (8 rows) cost the launch 0.7 us more than 128; 32 rows are
    // slower again (64 workgroups of 1 024 threads).  SLN_GATHER_YT = 8 / 16 / 32 for the lab.
    static const int yt = std::getenv("SLN_GATHER_YT") ? std::atoi(std::getenv("SLN_GATHER_YT")) : 16;
    if (D > 64 && yt == 16) hipLaunchKernelGGL((gather_bwd_v4_kernel<32, 16>), dim3(sln_cdiv(D, 128), sln_cdiv(O, 16)), dim3(32, 16), 0, st, dG, ldg, D, g, O, g.T, add1,
                                               ldadd1, xprev, ldx, bn, masked, out, ldo, gsums, cstride);
    else if (D > 64 && yt == 32) hipLaunchKernelGGL((gather_bwd_v4_kernel<32, 32>), dim3(sln_cdiv(D, 128), sln_cdiv(O, 32)), dim3(32, 32), 0, st, dG, ldg, D, g, O, g.T, add1,
                                               ldadd1, xprev, ldx, bn, masked, out, ldo, gsums, cstride);
    else if (D > 64) hipLaunchKernelGGL((gather_bwd_v4_kernel<32, 8>), dim3(sln_cdiv(D, 128), sln_cdiv(O, 8)), dim3(32, 8), 0, st, dG, ldg, D, g, O, g.T, add1,
                                   ldadd1, xprev, ldx, bn, masked, out, ldo, gsums, cstride);
    else hipLaunchKernelGGL((gather_bwd_v4_kernel<16, 16>), dim3(sln_cdiv(D, 64), sln_cdiv(O, 16)), dim3(16, 16), 0, st, dG, ldg, D, g, O, g.T, add1,
                            ldadd1, xprev, ldx, bn, masked, out, ldo, gsums, cstride);
    SLN_CHECK_LAUNCH();
    return 0;
  }
  return -2;   // SLN_E_UNSUPPORTED: misaligned rows (the engine never produces them)
}

int sln_launch_mask_gstats(const float* d1, int ld1, const float* d2, int ld2, const float* xprev, int ldx, BnView bn,
                           int rows, int cols, float* out, int ldo, double* gsums, int cstride, hipStream_t st) {
  if (rows <= 0) return 0;
  SlnProfScope prof(SLN_FAM_OTHER, 4.0 * rows * cols * (d2 ? 4.0 : 3.0), st);
  hipLaunchKernelGGL(mask_gstats_kernel, colgrid(cols, rows), dim3(CB, RL), 0, st, d1, ld1, d2, ld2, xprev, ldx, bn, rows,
                     cols, out, ldo, gsums, cstride);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_launch_enc_assemble(EncAssemble a, hipStream_t st) {
  const long n = (long)a.O * (a.n_obj + a.n_attr + a.n_box + a.n_angle);
  if (n <= 0) return 0;
  hipLaunchKernelGGL(enc_assemble_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
  SLN_CHECK_LAUNCH();
  return 0;
}

// Embedding gradients of the assembled encoder / decoder input: d_emb[idx[r], :] += dx0[r, col0 : col0 + n] for up to three
// tables.  2 048 rows draw from 5-35 table rows, so element-wise global atomics pile 60-400 adds on every table element (51 and
// 42 us per step for 1 MB of gradient); a workgroup accumulates ITS rows into LDS copies of the tables first and adds the
// touched elements once.
struct EmbSeg { const int64_t* idx; float* d_emb; int n, rows, col0, lds0; };
struct AssembleBwdLds {
  EmbSeg seg[3]; int nseg;
  const float* dx0; int ld, O, rows_per_block, lds_floats;
  float* dz; int z_col0, n_z;               // decoder: dz[r, :] = dx0[r, z_col0 : z_col0 + n_z]  (dz may be null)
};
__device__ __forceinline__ void assemble_bwd_lds_body(AssembleBwdLds a) {
  extern __shared__ float tab[];
  for (int i = threadIdx.x; i < a.lds_floats; i += 256) tab[i] = 0.f;
  __syncthreads();
  const int r0 = blockIdx.x * a.rows_per_block, r1 = min(a.O, r0 + a.rows_per_block);
  for (int s = 0; s < a.nseg; ++s) {
    const EmbSeg sg = a.seg[s];
    const long iend = (long)r1 * sg.n;
    for (long i = (long)r0 * sg.n + threadIdx.x; i < iend; i += 256 * 4) {     // independent loads first, then the LDS atomics
      float v[4]; int slot[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long q = min(i + 256L * u, iend - 1);
        const int r = (int)(q / sg.n), c = (int)(q % sg.n);
        slot[u] = sg.lds0 + (int)sg.idx[r] * sg.n + c;
        v[u] = a.dx0[(size_t)r * a.ld + sg.col0 + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (i + 256L * u < iend) atomicAdd(&tab[slot[u]], v[u]);
    }
  }
  if (a.dz) {
    const long iend = (long)r1 * a.n_z;
    for (long i = (long)r0 * a.n_z + threadIdx.x; i < iend; i += 256) {
      const int r = (int)(i / a.n_z), c = (int)(i % a.n_z);
      a.dz[i] = a.dx0[(size_t)r * a.ld + a.z_col0 + c];
    }
  }
  __syncthreads();
  for (int s = 0; s < a.nseg; ++s) {
    const EmbSeg sg = a.seg[s];
    for (int i = threadIdx.x; i < sg.rows * sg.n; i += 256) {
      const float v = tab[sg.lds0 + i];
      if (v != 0.f) atomicAdd(sg.d_emb + i, v);
    }
  }
}
__global__ __launch_bounds__(256) void assemble_bwd_lds_kernel(AssembleBwdLds a) {
  assemble_bwd_lds_body(a);
}
static int launch_assemble_bwd_lds(AssembleBwdLds a, hipStream_t st) {
  int off = 0;
  for (int s = 0; s < a.nseg; ++s) { a.seg[s].lds0 = off; off += a.seg[s].rows * a.seg[s].n; }
  a.lds_floats = off;
  a.rows_per_block = a.O <= 8192 ? 16 : 64;
  hipLaunchKernelGGL(assemble_bwd_lds_kernel, dim3(sln_cdiv(a.O, a.rows_per_block)), dim3(256), sizeof(float) * off, st, a);
  return (int)hipGetLastError();
}
constexpr int ASSEMBLE_LDS_MAX_FLOATS = 10240;      // 40 KB of tables per workgroup; larger vocabularies keep the global atomics

int sln_launch_enc_assemble_bwd(EncAssembleBwd a, hipStream_t st) {
  const long n = (long)a.O * (a.n_obj + a.n_attr + a.n_box + a.n_angle);
  if (n <= 0) return 0;
  const long tabs = (long)a.rows_obj * a.n_obj + (long)a.rows_attr * a.n_attr + (long)a.rows_angle * a.n_angle;
  if (g_sln_deterministic && a.rows_obj > 0 && a.rows_angle > 0 && (a.n_attr == 0 || a.rows_attr > 0)) {
    const int ld = a.n_obj + a.n_attr + a.n_box + a.n_angle;
    int r = sln_launch_embed_bwd_i64(a.objs, a.dx0, ld, 0, a.O, a.n_obj, a.rows_obj, a.d_obj_emb, st);
    if (!r && a.n_attr > 0) r = sln_launch_embed_bwd_i64(a.attrs, a.dx0, ld, a.n_obj, a.O, a.n_attr, a.rows_attr, a.d_attr_emb, st);
    if (!r) r = sln_launch_embed_bwd_i64(a.angles, a.dx0, ld, a.n_obj + a.n_attr + a.n_box, a.O, a.n_angle, a.rows_angle, a.d_angle_emb, st);
    if (r) return r;
    hipLaunchKernelGGL(box_embed_bwd_kernel, dim3(sln_cdiv(a.n_box, CB), 1), dim3(CB, RL), 0, st, a, a.O);
    SLN_CHECK_LAUNCH();
    return 0;
  }
  if (a.rows_obj > 0 && a.rows_angle > 0 && (a.n_attr == 0 || a.rows_attr > 0) && tabs <= ASSEMBLE_LDS_MAX_FLOATS) {
    AssembleBwdLds l; std::memset(&l, 0, sizeof(l));
    l.dx0 = a.dx0; l.ld = a.n_obj + a.n_attr + a.n_box + a.n_angle; l.O = a.O;
    int k = 0;
    l.seg[k++] = EmbSeg{a.objs, a.d_obj_emb, a.n_obj, a.rows_obj, 0, 0};
    if (a.n_attr > 0) l.seg[k++] = EmbSeg{a.attrs, a.d_attr_emb, a.n_attr, a.rows_attr, a.n_obj, 0};
    l.seg[k++] = EmbSeg{a.angles, a.d_angle_emb, a.n_angle, a.rows_angle, a.n_obj + a.n_attr + a.n_box, 0};
    l.nseg = k;
    const int e = launch_assemble_bwd_lds(l, st);
    if (e) return e;
  } else
  hipLaunchKernelGGL(enc_assemble_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
  hipLaunchKernelGGL(box_embed_bwd_kernel, dim3(sln_cdiv(a.n_box, CB), sln_cdiv(a.O, BOX_ROWS)), dim3(CB, RL), 0, st, a, (int)BOX_ROWS);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_launch_dec_assemble(DecAssemble a, hipStream_t st) {
  const long n = (long)a.O * (a.n_obj + a.n_attr + a.n_z);
  if (n <= 0) return 0;
  hipLaunchKernelGGL(dec_assemble_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_launch_dec_assemble_bwd(DecAssembleBwd a, hipStream_t st) {
  const long n = (long)a.O * (a.n_obj + a.n_attr + (a.z_in_x0 ? a.n_z : 0));
  if (n <= 0) return 0;
  const long tabs = (long)a.rows_obj * a.n_obj + (long)a.rows_attr * a.n_attr;
  if (g_sln_deterministic && a.rows_obj > 0 && (a.n_attr == 0 || a.rows_attr > 0)) {
    const int ld = a.n_obj + a.n_attr + (a.z_in_x0 ? a.n_z : 0);
    int r = sln_launch_embed_bwd_i64(a.objs, a.dx0, ld, 0, a.O, a.n_obj, a.rows_obj, a.d_obj_emb, st);
    if (!r && a.n_attr > 0) r = sln_launch_embed_bwd_i64(a.attrs, a.dx0, ld, a.n_obj, a.O, a.n_attr, a.rows_attr, a.d_attr_emb, st);
    if (r) return r;
    if (a.z_in_x0 && a.dz)
      return (int)hipMemcpy2DAsync(a.dz, sizeof(float) * a.n_z, a.dx0 + a.n_obj + a.n_attr, sizeof(float) * ld, sizeof(float) * a.n_z,
                                   (size_t)a.O, hipMemcpyDeviceToDevice, st);
    return 0;
  }
  if (a.rows_obj > 0 && (a.n_attr == 0 || a.rows_attr > 0) && tabs <= ASSEMBLE_LDS_MAX_FLOATS) {
    AssembleBwdLds l; std::memset(&l, 0, sizeof(l));
    l.dx0 = a.dx0; l.ld = a.n_obj + a.n_attr + (a.z_in_x0 ? a.n_z : 0); l.O = a.O;
    int k = 0;
    l.seg[k++] = EmbSeg{a.objs, a.d_obj_emb, a.n_obj, a.rows_obj, 0, 0};
    if (a.n_attr > 0) l.seg[k++] = EmbSeg{a.attrs, a.d_attr_emb, a.n_attr, a.rows_attr, a.n_obj, 0};
    l.nseg = k;
    if (a.z_in_x0 && a.dz) { l.dz = a.dz; l.z_col0 = a.n_obj + a.n_attr; l.n_z = a.n_z; }
    return launch_assemble_bwd_lds(l, st);
  }
  hipLaunchKernelGGL(dec_assemble_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_launch_embed_bwd_i32(const int* idx, const float* d, int ld, int col0, int rows, int n, int table_rows,
                             float* d_emb, hipStream_t st) {
  const long tot = (long)rows * n;
  if (tot <= 0) return 0;
  if (g_sln_deterministic && table_rows > 0) {
    hipLaunchKernelGGL(embed_bwd_det_kernel<int>, dim3(table_rows, sln_cdiv(n, 64)), dim3(1024), 0, st, idx, d, ld, col0, rows, n, d_emb);
    SLN_CHECK_LAUNCH();
    return 0;
  }
  if (table_rows > 0 && (long)table_rows * n <= 8192) {
    // 16 rows per workgroup: 256 workgroups at 64 graphs (64 rows left three quarters of the CUs idle: 16 -> 6 us per launch;
    // 8 / 4 rows pay more flush atomics than they gain: SLN_EMB_RPB)
    static const int rpb = std::getenv("SLN_EMB_RPB") ? std::atoi(std::getenv("SLN_EMB_RPB")) : 16;
    hipLaunchKernelGGL(embed_bwd_lds_kernel<int>, dim3(sln_cdiv(rows, rpb)), dim3(256), sizeof(float) * table_rows * n, st, idx,
                       d, ld, col0, rows, n, table_rows, rpb, d_emb);
    SLN_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(embed_bwd_kernel<int>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, idx, d, ld, col0, rows,
                     n, d_emb);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_launch_embed_bwd_i64(const int64_t* idx, const float* d, int ld, int col0, int rows, int n, int table_rows,
                             float* d_emb, hipStream_t st) {
  const long tot = (long)rows * n;
  if (tot <= 0) return 0;
  if (g_sln_deterministic && table_rows > 0) {
    hipLaunchKernelGGL(embed_bwd_det_kernel<int64_t>, dim3(table_rows, sln_cdiv(n, 64)), dim3(1024), 0, st, idx, d, ld, col0, rows, n, d_emb);
    SLN_CHECK_LAUNCH();
    return 0;
  }
  if (table_rows > 0 && (long)table_rows * n <= 8192) {
    // 16 rows per workgroup: 256 workgroups at 64 graphs (64 rows left three quarters of the CUs idle: 16 -> 6 us per launch;
    // 8 / 4 rows pay more flush atomics than they gain: SLN_EMB_RPB)
    static const int rpb = std::getenv("SLN_EMB_RPB") ? std::atoi(std::getenv("SLN_EMB_RPB")) : 16;
    hipLaunchKernelGGL(embed_bwd_lds_kernel<int64_t>, dim3(sln_cdiv(rows, rpb)), dim3(256), sizeof(float) * table_rows * n, st,
                       idx, d, ld, col0, rows, n, table_rows, rpb, d_emb);
    SLN_CHECK_LAUNCH();
    return 0;
  }
  hipLaunchKernelGGL(embed_bwd_kernel<int64_t>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, idx, d, ld, col0,
                     rows, n, d_emb);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_launch_embed_gather_i32(const int* idx, const float* emb, int rows, int n, float* out, hipStream_t st) {
  const long tot = (long)rows * n;
  if (tot <= 0) return 0;
  hipLaunchKernelGGL(embed_gather_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, idx, emb, rows, n, out);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_launch_i64_to_i32(const int64_t* src, int* dst, int n, hipStream_t st) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(i64_to_i32_kernel, dim3(sln_cdiv(n, 256)), dim3(256), 0, st, src, dst, n);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_launch_log_softmax_bwd(const float* logprob, const float* d_logprob, float* d_logits, int O, int n,
                               hipStream_t st) {
  if (O <= 0) return 0;
  hipLaunchKernelGGL(log_softmax_bwd_kernel, dim3(sln_cdiv(O, 128)), dim3(128), 0, st, logprob, d_logprob, d_logits, O, n);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_launch_log_softmax(const float* logits, float* out, int O, int n, hipStream_t st) {
  if (O <= 0) return 0;
  hipLaunchKernelGGL(log_softmax_kernel, dim3(sln_cdiv(O, 128)), dim3(128), 0, st, logits, out, O, n);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_launch_loss(LossArgs a, hipStream_t st) {
  if (!a.acc_prezeroed) {
    const int e = sln_zero_async(a.acc, sizeof(double) * 4, st);
    if (e != 0) return e;
  }
  hipLaunchKernelGGL(loss_kernel, dim3(LOSS_BLOCKS), dim3(256), 0, st, a);      // O == 0: every loop is empty, the sums are 0
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_launch_latent_bwd(const float* mu, const float* logvar, const float* eps, const float* dz, const float* kl_weight,
                          int O, int n_z, int use_ae, float* dmu, float* dlogvar, hipStream_t st) {
  const long n = (long)O * n_z;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(latent_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, mu, logvar, eps, dz,
                     kl_weight, O, n_z, use_ae, dmu, dlogvar);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_launch_bn_running_update(const BnTableEntry* table, int n, int max_c, float momentum, int independent,
                                 hipStream_t st, int rows_t, int rows_o) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(bn_running_update_kernel, dim3(sln_cdiv(max_c, 256), independent ? n : 1), dim3(256), 0, st, table, n,
                     momentum, independent, rows_t, rows_o);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_launch_bn_param_grads(const BnTableEntry* table, int n, int max_c, int independent, hipStream_t st) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(bn_param_grads_kernel, dim3(sln_cdiv(max_c, 256), independent ? n : 1), dim3(256), 0, st, table, n,
                     independent);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_launch_transpose_table(const TransposeEntry* table, int n, int max_tiles, hipStream_t st) {
  if (n <= 0 || max_tiles <= 0) return 0;
  hipLaunchKernelGGL(transpose_table_kernel, dim3(max_tiles, n), dim3(256), 0, st, table);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_launch_adam(float* params, const float* grads, float* m, float* v, long n, AdamScalars* scalars, const float* total_loss,
                    hipStream_t st) {
  SlnProfScope prof(SLN_FAM_OTHER, 28.0 * n, st);
  long blocks = (n + 4095) / 4096;    // few workgroups: every one of them is an arrival at ONE counter (see adam_kernel)
  if (blocks > 256) blocks = 256;
  if (blocks < 1) blocks = 1;
  const int vec = ((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(1024), 0, st, params, grads, m, v, n, vec, scalars, total_loss);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_launch_randn(float* eps, long n, AdamScalars* scalars, hipStream_t st) {
  if (n <= 0) return 0;
  const long q = (n + 3) / 4;
  hipLaunchKernelGGL(randn_kernel, dim3((unsigned)((q + 255) / 256)), dim3(256), 0, st, eps, n, scalars);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_launch_step_prologue(const StepPrologue& a, hipStream_t st) {
  auto blocks = [](long n) { return (unsigned int)((n + 255) / 256); };
  const unsigned int nr = a.eps ? blocks((a.n_eps + 3) / 4) : 0;
  const unsigned int ne = blocks((long)a.enc.O * (a.enc.n_obj + a.enc.n_attr + a.enc.n_box + a.enc.n_angle));
  const unsigned int n1 = blocks((long)a.T * a.n_ec), n2 = blocks((long)a.T * a.n_dc);
  const unsigned int nz = a.zero_ptr ? blocks(a.zero_bytes / 16) : 0;
  const unsigned int tot = nr + ne + n1 + n2 + nz;
  if (tot == 0) return 0;
  hipLaunchKernelGGL(step_prologue_kernel, dim3(tot), dim3(256), 0, st, a, nr, nr + ne, nr + ne + n1, nr + ne + n1 + n2);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_launch_add2(const float* a, int lda, const float* b, int ldb, int rows, int cols, float* out, int ldo, hipStream_t st) {
  const long n = (long)rows * cols;
  if (n <= 0) return 0;
  hipLaunchKernelGGL(add2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, lda, b, ldb, rows, cols, out, ldo);
  SLN_CHECK_LAUNCH();
  return 0;
}

// =================================================================================================
// Multi-room launches (vae_multi.h): blockIdx.z = room, the room's argument block comes out of a device table (uniform address:
// scalar loads), workgroups beyond the room's own grid leave.  Bodies = the single-room kernels above.
// =================================================================================================
#include "vae_multi.h"
namespace {

template <int XT, int YT>
__global__ __launch_bounds__(XT* YT) void scatter_avg_fwd_multi_kernel(const MScatterFwd* __restrict__ tab) {
  const MScatterFwd& a = tab[blockIdx.z];
  if ((int)blockIdx.x >= a.gx || (int)blockIdx.y >= a.gy) return;
  scatter_avg_fwd_v4_body<XT, YT>(a.A2, a.ld, a.H, a.D, a.bn, a.g, a.O, a.g.T, a.pooled);
}
template <int XT, int YT, int RPT, int NIT>
__global__ __launch_bounds__(XT* YT) void scatter_avg_bwd_multi_kernel(const MScatterBwd* __restrict__ tab) {
  const MScatterBwd& a = tab[blockIdx.z];
  if ((int)blockIdx.x >= a.gx || (int)blockIdx.y >= a.gy) return;
  scatter_avg_bwd_v4_body<XT, YT, RPT, NIT>(a.dM, a.dP, a.lddp, a.dpcol0, a.A2, a.ld, a.H, a.D, a.bn, a.g, a.T, a.g2, a.gsums, a.cstride);
}
template <int XT, int YT>
__global__ __launch_bounds__(XT* YT) void gather_bwd_multi_kernel(const MGatherBwd* __restrict__ tab) {
  const MGatherBwd& a = tab[blockIdx.z];
  if ((int)blockIdx.x >= a.gx || (int)blockIdx.y >= a.gy) return;
  gather_bwd_v4_body<XT, YT>(a.dG, a.ldg, a.D, a.g, a.O, a.g.T, a.add1, a.ldadd1, a.xprev, a.ldx, a.bn, a.masked, a.out, a.ldo, a.gsums, a.cstride);
}
__global__ __launch_bounds__(CB* RL) void mask_gstats_multi_kernel(const MMaskGstats* __restrict__ tab) {
  const MMaskGstats& a = tab[blockIdx.z];
  if ((int)blockIdx.x >= a.gx || (int)blockIdx.y >= a.gy) return;
  mask_gstats_body(a.d1, a.ld1, a.d2, a.ld2, a.xprev, a.ldx, a.bn, a.rows, a.cols, a.out, a.ldo, a.gsums, a.cstride);
}
__global__ void dec_assemble_multi_kernel(const MDecAssemble* __restrict__ tab) {
  const MDecAssemble& a = tab[blockIdx.z];
  if ((int)blockIdx.x >= a.gx) return;
  dec_assemble_body(a.a);
}
__global__ void dec_assemble_bwd_multi_kernel(const MDecAssembleBwd* __restrict__ tab) {
  const MDecAssembleBwd& a = tab[blockIdx.z];
  if ((int)blockIdx.x >= a.gx) return;
  dec_assemble_bwd_body(a.a);
}
struct MAsmLds { AssembleBwdLds a; int gx, pad_; };
static_assert(sizeof(MAsmLds) <= SLN_ASM_BLOB && sizeof(MDecAssembleBwd) <= SLN_ASM_BLOB, "SLN_ASM_BLOB too small");
__global__ __launch_bounds__(256) void assemble_bwd_lds_multi_kernel(const char* __restrict__ tab) {
  const MAsmLds& a = *reinterpret_cast<const MAsmLds*>(tab + (size_t)blockIdx.z * SLN_ASM_BLOB);
  if ((int)blockIdx.x >= a.gx) return;
  assemble_bwd_lds_body(a.a);
}
__global__ void dec_assemble_bwd_plain_multi_kernel(const char* __restrict__ tab) {
  const MDecAssembleBwd& a = *reinterpret_cast<const MDecAssembleBwd*>(tab + (size_t)blockIdx.z * SLN_ASM_BLOB);
  if ((int)blockIdx.x >= a.gx) return;
  dec_assemble_bwd_body(a.a);
}
__global__ void embed_gather_multi_kernel(const MEmbedGather* __restrict__ tab) {
  const MEmbedGather& a = tab[blockIdx.z];
  if ((int)blockIdx.x >= a.gx) return;
  embed_gather_body(a.idx, a.emb, a.rows, a.n, a.out, blockIdx.x);
}
template <typename IdxT, int V>
__global__ __launch_bounds__(V == MV_EMBED_DET ? 1024 : 256) void embed_bwd_multi_kernel(const MEmbedBwd* __restrict__ tab) {
  const MEmbedBwd& a = tab[blockIdx.z];
  if ((int)blockIdx.x >= a.gx || (int)blockIdx.y >= a.gy) return;
  const IdxT* idx = static_cast<const IdxT*>(a.idx);
  if (V == MV_EMBED_DET) embed_bwd_det_body<IdxT>(idx, a.d, a.ld, a.col0, a.rows, a.n, a.d_emb);
  else if (V == MV_EMBED_LDS) embed_bwd_lds_body<IdxT>(idx, a.d, a.ld, a.col0, a.rows, a.n, a.table_rows, a.rows_per_block, a.d_emb);
  else embed_bwd_body<IdxT>(idx, a.d, a.ld, a.col0, a.rows, a.n, a.d_emb);
}
__global__ void add2_multi_kernel(const MAdd2* __restrict__ tab) {
  const MAdd2& a = tab[blockIdx.z];
  if ((int)blockIdx.x >= a.gx) return;
  add2_body(a.a, a.lda, a.b, a.ldb, a.rows, a.cols, a.out, a.ldo);
}
__global__ void copy2d_multi_kernel(const MAdd2* __restrict__ tab) {
  const MAdd2& a = tab[blockIdx.z];
  if ((int)blockIdx.x >= a.gx) return;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)a.rows * a.cols) return;
  const int r = (int)(i / a.cols), c = (int)(i % a.cols);
  a.out[(size_t)r * a.ldo + c] = a.a[(size_t)r * a.lda + c];
}
__global__ void zero_multi_kernel(const MZero* __restrict__ tab) {
  const MZero a = tab[blockIdx.y];
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < a.n16) static_cast<uint4*>(a.p)[i] = make_uint4(0u, 0u, 0u, 0u);
}

}  // namespace

int sln_plan_scatter_avg_fwd(MScatterFwd& a) {
  if (a.O <= 0) { a.gx = a.gy = 0; return a.H > 128 ? MV_SCATTER_FWD_64x4 : MV_SCATTER_FWD_32x8; }
  if (!(a.H % 4 == 0 && a.D % 4 == 0 && a.ld % 4 == 0 && al16(a.A2) && al16(a.pooled))) return -1;
  if (a.H > 128) { a.gx = sln_cdiv(a.H, 256); a.gy = sln_cdiv(a.O, 4); return MV_SCATTER_FWD_64x4; }
  a.gx = sln_cdiv(a.H, 128); a.gy = sln_cdiv(a.O, 8);
  return MV_SCATTER_FWD_32x8;
}
int sln_launch_scatter_avg_fwd_multi(const MScatterFwd* tab, int R, int variant, int gx, int gy, hipStream_t st) {
  if (R <= 0 || gx <= 0 || gy <= 0) return 0;
  if (variant == MV_SCATTER_FWD_64x4) hipLaunchKernelGGL((scatter_avg_fwd_multi_kernel<64, 4>), dim3(gx, gy, R), dim3(64, 4), 0, st, tab);
  else hipLaunchKernelGGL((scatter_avg_fwd_multi_kernel<32, 8>), dim3(gx, gy, R), dim3(32, 8), 0, st, tab);
  SLN_CHECK_LAUNCH();
  return 0;
}
int sln_plan_scatter_avg_bwd(MScatterBwd& a) {
  if (a.T <= 0) { a.gx = a.gy = 0; return 0; }
  if (!(a.H % 4 == 0 && a.D % 4 == 0 && a.ld % 4 == 0 && (!a.dP || (a.lddp % 4 == 0 && a.dpcol0 % 4 == 0 && al16(a.dP))) && al16(a.dM) && al16(a.A2) &&
        al16(a.g2))) return -1;
  a.gx = sln_cdiv(2 * a.H + a.D, 256); a.gy = sln_cdiv(a.T, 4 * 4 * 2);       // <64, 4, RPT = 4, NIT = 2>: the single-room default
  return 0;
}
int sln_launch_scatter_avg_bwd_multi(const MScatterBwd* tab, int R, int variant, int gx, int gy, hipStream_t st) {
  (void)variant;
  if (R <= 0 || gx <= 0 || gy <= 0) return 0;
  hipLaunchKernelGGL((scatter_avg_bwd_multi_kernel<64, 4, 4, 2>), dim3(gx, gy, R), dim3(64, 4), 0, st, tab);
  SLN_CHECK_LAUNCH();
  return 0;
}
int sln_plan_gather_bwd(MGatherBwd& a) {
  if (a.O <= 0) { a.gx = a.gy = 0; return a.D > 64 ? MV_GATHER_32x16 : MV_GATHER_16x16; }
  if (!(a.D % 4 == 0 && a.ldg % 4 == 0 && a.ldo % 4 == 0 && (!a.add1 || (a.ldadd1 % 4 == 0 && al16(a.add1))) &&
        (!a.masked || (a.ldx % 4 == 0 && al16(a.xprev))) && al16(a.dG) && al16(a.out))) return -1;
  if (a.D > 64) { a.gx = sln_cdiv(a.D, 128); a.gy = sln_cdiv(a.O, 16); return MV_GATHER_32x16; }
  a.gx = sln_cdiv(a.D, 64); a.gy = sln_cdiv(a.O, 16);
  return MV_GATHER_16x16;
}
int sln_launch_gather_bwd_multi(const MGatherBwd* tab, int R, int variant, int gx, int gy, hipStream_t st) {
  if (R <= 0 || gx <= 0 || gy <= 0) return 0;
  if (variant == MV_GATHER_32x16) hipLaunchKernelGGL((gather_bwd_multi_kernel<32, 16>), dim3(gx, gy, R), dim3(32, 16), 0, st, tab);
  else hipLaunchKernelGGL((gather_bwd_multi_kernel<16, 16>), dim3(gx, gy, R), dim3(16, 16), 0, st, tab);
  SLN_CHECK_LAUNCH();
  return 0;
}
int sln_plan_mask_gstats(MMaskGstats& a) {
  if (a.rows <= 0) { a.gx = a.gy = 0; return 0; }
  const dim3 g = colgrid(a.cols, a.rows);
  a.gx = (int)g.x; a.gy = (int)g.y;
  return 0;
}
int sln_launch_mask_gstats_multi(const MMaskGstats* tab, int R, int gx, int gy, hipStream_t st) {
  if (R <= 0 || gx <= 0 || gy <= 0) return 0;
  hipLaunchKernelGGL(mask_gstats_multi_kernel, dim3(gx, gy, R), dim3(CB, RL), 0, st, tab);
  SLN_CHECK_LAUNCH();
  return 0;
}
int sln_plan_dec_assemble(MDecAssemble& a) {
  const long n = (long)a.a.O * (a.a.n_obj + a.a.n_attr + a.a.n_z);
  a.gx = (int)((n + 255) / 256);
  return 0;
}
int sln_launch_dec_assemble_multi(const MDecAssemble* tab, int R, int gx, hipStream_t st) {
  if (R <= 0 || gx <= 0) return 0;
  hipLaunchKernelGGL(dec_assemble_multi_kernel, dim3(gx, 1, R), dim3(256), 0, st, tab);
  SLN_CHECK_LAUNCH();
  return 0;
}
int sln_plan_dec_assemble_bwd(const DecAssembleBwd& a, void* blob, int* gx, int* smem_floats) {
  std::memset(blob, 0, SLN_ASM_BLOB);
  const long n = (long)a.O * (a.n_obj + a.n_attr + (a.z_in_x0 ? a.n_z : 0));
  const long tabs = (long)a.rows_obj * a.n_obj + (long)a.rows_attr * a.n_attr;
  if (g_sln_deterministic) return -1;
  if (a.rows_obj > 0 && (a.n_attr == 0 || a.rows_attr > 0) && tabs <= ASSEMBLE_LDS_MAX_FLOATS) {
    MAsmLds m; std::memset(&m, 0, sizeof(m));
    AssembleBwdLds& l = m.a;
    l.dx0 = a.dx0; l.ld = a.n_obj + a.n_attr + (a.z_in_x0 ? a.n_z : 0); l.O = a.O;
    int k = 0;
    l.seg[k++] = EmbSeg{a.objs, a.d_obj_emb, a.n_obj, a.rows_obj, 0, 0};
    if (a.n_attr > 0) l.seg[k++] = EmbSeg{a.attrs, a.d_attr_emb, a.n_attr, a.rows_attr, a.n_obj, 0};
    l.nseg = k;
    if (a.z_in_x0 && a.dz) { l.dz = a.dz; l.z_col0 = a.n_obj + a.n_attr; l.n_z = a.n_z; }
    int off = 0;
    for (int s = 0; s < l.nseg; ++s) { l.seg[s].lds0 = off; off += l.seg[s].rows * l.seg[s].n; }
    l.lds_floats = off;
    l.rows_per_block = 16;                        // (launch_assemble_bwd_lds: 16 up to 8 192 rows; rooms have a few dozen)
    if (a.O > 8192) return -1;
    m.gx = n > 0 ? sln_cdiv(a.O, l.rows_per_block) : 0;
    std::memcpy(blob, &m, sizeof(m));
    *gx = m.gx; *smem_floats = off;
    return MV_ASM_BWD_LDS;
  }
  MDecAssembleBwd m; std::memset(&m, 0, sizeof(m));
  m.a = a; m.gx = (int)((n + 255) / 256);
  std::memcpy(blob, &m, sizeof(m));
  *gx = m.gx; *smem_floats = 0;
  return MV_ASM_BWD_PLAIN;
}
int sln_launch_dec_assemble_bwd_multi(const void* tab, int R, int variant, int gx, int smem_floats, hipStream_t st) {
  if (R <= 0 || gx <= 0) return 0;
  if (variant == MV_ASM_BWD_LDS)
    hipLaunchKernelGGL(assemble_bwd_lds_multi_kernel, dim3(gx, 1, R), dim3(256), sizeof(float) * (size_t)smem_floats, st, static_cast<const char*>(tab));
  else hipLaunchKernelGGL(dec_assemble_bwd_plain_multi_kernel, dim3(gx, 1, R), dim3(256), 0, st, static_cast<const char*>(tab));
  SLN_CHECK_LAUNCH();
  return 0;
}
int sln_plan_embed_gather(MEmbedGather& a) {
  a.gx = (int)(((long)a.rows * a.n + 255) / 256);
  return 0;
}
int sln_launch_embed_gather_multi(const MEmbedGather* tab, int R, int gx, hipStream_t st) {
  if (R <= 0 || gx <= 0) return 0;
  hipLaunchKernelGGL(embed_gather_multi_kernel, dim3(gx, 1, R), dim3(256), 0, st, tab);
  SLN_CHECK_LAUNCH();
  return 0;
}
int sln_plan_embed_bwd(MEmbedBwd& a, int idx64) {
  const long tot = (long)a.rows * a.n;
  const int w = idx64 ? 4 : 0;
  a.rows_per_block = 16;
  if (g_sln_deterministic && a.table_rows > 0) { a.gx = tot > 0 ? a.table_rows : 0; a.gy = sln_cdiv(a.n, 64); return w + MV_EMBED_DET; }
  if (a.table_rows > 0 && (long)a.table_rows * a.n <= 8192) { a.gx = tot > 0 ? sln_cdiv(a.rows, a.rows_per_block) : 0; a.gy = 1; return w + MV_EMBED_LDS; }
  a.gx = (int)((tot + 255) / 256); a.gy = 1;
  return w + MV_EMBED_PLAIN;
}
int sln_launch_embed_bwd_multi(const MEmbedBwd* tab, int R, int variant, int gx, int gy, int smem_floats, hipStream_t st) {
  if (R <= 0 || gx <= 0 || gy <= 0) return 0;
  const dim3 grid(gx, gy, R);
  switch (variant) {
    case MV_EMBED_DET: hipLaunchKernelGGL((embed_bwd_multi_kernel<int, MV_EMBED_DET>), grid, dim3(1024), 0, st, tab); break;
    case MV_EMBED_LDS: hipLaunchKernelGGL((embed_bwd_multi_kernel<int, MV_EMBED_LDS>), grid, dim3(256), sizeof(float) * (size_t)smem_floats, st, tab); break;
    case MV_EMBED_PLAIN: hipLaunchKernelGGL((embed_bwd_multi_kernel<int, MV_EMBED_PLAIN>), grid, dim3(256), 0, st, tab); break;
    case 4 + MV_EMBED_DET: hipLaunchKernelGGL((embed_bwd_multi_kernel<int64_t, MV_EMBED_DET>), grid, dim3(1024), 0, st, tab); break;
    case 4 + MV_EMBED_LDS: hipLaunchKernelGGL((embed_bwd_multi_kernel<int64_t, MV_EMBED_LDS>), grid, dim3(256), sizeof(float) * (size_t)smem_floats, st, tab); break;
    case 4 + MV_EMBED_PLAIN: hipLaunchKernelGGL((embed_bwd_multi_kernel<int64_t, MV_EMBED_PLAIN>), grid, dim3(256), 0, st, tab); break;
    default: return -1;
  }
  SLN_CHECK_LAUNCH();
  return 0;
}
int sln_plan_add2(MAdd2& a) {
  a.gx = (int)(((long)a.rows * a.cols + 255) / 256);
  return 0;
}
int sln_launch_add2_multi(const MAdd2* tab, int R, int gx, hipStream_t st) {
  if (R <= 0 || gx <= 0) return 0;
  hipLaunchKernelGGL(add2_multi_kernel, dim3(gx, 1, R), dim3(256), 0, st, tab);
  SLN_CHECK_LAUNCH();
  return 0;
}
int sln_launch_copy2d_multi(const MAdd2* tab, int R, int gx, hipStream_t st) {
  if (R <= 0 || gx <= 0) return 0;
  hipLaunchKernelGGL(copy2d_multi_kernel, dim3(gx, 1, R), dim3(256), 0, st, tab);
  SLN_CHECK_LAUNCH();
  return 0;
}
int sln_launch_zero_multi(const MZero* tab, int R, long max_n16, hipStream_t st) {
  if (R <= 0 || max_n16 <= 0) return 0;
  hipLaunchKernelGGL(zero_multi_kernel, dim3((unsigned)((max_n16 + 255) / 256), R), dim3(256), 0, st, tab);
  SLN_CHECK_LAUNCH();
  return 0;
}

// ---- posterior heat map (testing/test_heatmap.py:80-99): counts[obj][rd_z][rd_x] += 1 over the trials, one thread per (trial, object) ----
namespace {
__global__ void layout_heatmap_kernel(const float* __restrict__ boxes, long n_trials, int O, int cs, int clip, float* __restrict__ counts) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_trials * (O - 1)) return;
  const long trial = i / (O - 1); const int obj = (int)(i % (O - 1));
  const float* room = boxes + (trial * O + (O - 1)) * 6;
  const float* b = boxes + (trial * O + obj) * 6;
  float ct[3]; bool keep = true;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float ext = room[3 + j] - room[j];
    ct[j] = (b[j] * ext + b[3 + j] * ext) * 0.5f;
    if (!clip) keep = keep && ct[j] > 0.f && ct[j] < 1.f;
    ct[j] = fminf(fmaxf(ct[j], 0.f), 1.f);
  }
  if (!keep) return;
  const int rz = (int)floorf(ct[2] * (float)(cs - 1)), rx = (int)floorf(ct[0] * (float)(cs - 1));
  atomicAdd(counts + ((long)obj * cs + rz) * cs + rx, 1.0f);
}
}  // namespace
int sln_launch_layout_heatmap(const float* boxes, long n_trials, int O, int cs, int clip, float* counts, hipStream_t st) {
  const long n = n_trials * (O - 1);
  if (n <= 0) return 0;
  hipLaunchKernelGGL(layout_heatmap_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, boxes, n_trials, O, cs, clip, counts);
  SLN_CHECK_LAUNCH();
  return 0;
}
