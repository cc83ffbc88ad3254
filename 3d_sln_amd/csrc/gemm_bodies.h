// Templates shared by the GEMM translation units (gemm_f32.hip: single / dual launches, gemm_group.hip: grouped launches).
// Everything lives in an anonymous namespace: each translation unit gets its own instantiations.
#pragma once
#include "sln_common.h"
#include "sln_gemm.h"
#include "sln_prof.h"

#include <cstdlib>
#include <cstring>
#include <type_traits>
namespace {


// phase stamps for tools/gemm_lab.hip (a no-op in the product build)
#ifndef SLN_TRACE
#define SLN_TRACE(i)
#endif
#ifndef SLN_TRACEH          // the same for the first helper wavefront (tools/lab/gemm_lab.hip)
#define SLN_TRACEH(i)
#endif

// 1: the 64 x 64 NT tile runs its K loop in a hand-ordered schedule (see gemm_nt_body); 0: hipcc's order (lab A/B)
#ifndef SLN_NT_SCHED
#define SLN_NT_SCHED 1
#endif
#ifndef SLN_TN_SCHED      // 1: the wgrad K loop in a hand-ordered schedule (gemm_tn_body), 0: hipcc's order (lab A/B)
#define SLN_TN_SCHED 1
#endif
#ifndef SLN_TN_ABL        // lab builds only (SLN_HIPCC_EXTRA=-DSLN_TN_ABL=n): leave parts of the wgrad loop out - 1 loads, 2 staging + LDS writes, 4 barrier, 8 fragment reads
#define SLN_TN_ABL 0
#endif
#ifndef SLN_ABL           // tools/lab/gemm_lab.hip only: leave parts of the scheduled loop out (1 loads, 2 LDS writes, 4 barrier, 8 fragment reads)
#define SLN_ABL 0
#endif

constexpr int BK = 32;
constexpr int TN_PAD = 32;     // LDS row padding of the TN (wgrad) tiles, see gemm_tn_body

// Operand loads name the GLOBAL address space.  A pointer that a kernel reads out of device memory (the problem table of the
// per-pass wgrad launch) has no known address space, hipcc then emits flat_load - which counts on BOTH vmcnt and lgkmcnt: every wait
// for an LDS read also waits for all operand loads in flight, and the prefetch distance of the K loop is gone.  (Pointers that
// arrive in the kernel arguments are known to be global; the cast costs nothing there.)
// (a native vector type: dereferencing a HIP float4 goes through its copy constructor, which takes a generic reference)
typedef float sln_v4f __attribute__((ext_vector_type(4)));
typedef const sln_v4f __attribute__((address_space(1)))* sln_gf4p;
typedef const int __attribute__((address_space(1)))* sln_gip;
__device__ __forceinline__ float4 ld4(const float* p) {
  const sln_v4f v = *(sln_gf4p)(p);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ int ldi(const int* p) { return *(sln_gip)(p); }
// dW / db += v as global_atomic_add_f32 (atomicAdd on a pointer of unknown address space is a flat atomic)
__device__ __forceinline__ void sln_gatomic_add(float* p, float v) {
  __hip_atomic_fetch_add((float __attribute__((address_space(1)))*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// v_max_f32 as the instruction: fmaxf() makes hipcc canonicalise both inputs first (one more v_max_f32 per coefficient that comes out
// of LDS), and every VALU instruction of a GEMM wave is paid in matrix-pipe time (tools/lab/overlap.hip).  Same result: the
// hardware instruction quiets NaNs itself.
__device__ __forceinline__ float sln_vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

__device__ __forceinline__ float4 xform(float4 x1, float4 x2, const float4* cf) {
  float4 r;
  float4 c;
  c = cf[0]; r.x = sln_vmax(fmaf(c.x, x1.x, fmaf(c.y, x2.x, c.z)), c.w);
  c = cf[1]; r.y = sln_vmax(fmaf(c.x, x1.y, fmaf(c.y, x2.y, c.z)), c.w);
  c = cf[2]; r.z = sln_vmax(fmaf(c.x, x1.z, fmaf(c.y, x2.z, c.z)), c.w);
  c = cf[3]; r.w = sln_vmax(fmaf(c.x, x1.w, fmaf(c.y, x2.w, c.z)), c.w);
  return r;
}

// the same with the four coefficient kinds of the thread's four columns as one float4 each (planar table: conflict-free LDS reads)
__device__ __forceinline__ float4 xform_planar(float4 x1, float4 x2, float4 c0, float4 c1, float4 c2, float4 cw) {
  float4 r;
  r.x = sln_vmax(fmaf(c0.x, x1.x, fmaf(c1.x, x2.x, c2.x)), cw.x);
  r.y = sln_vmax(fmaf(c0.y, x1.y, fmaf(c1.y, x2.y, c2.y)), cw.y);
  r.z = sln_vmax(fmaf(c0.z, x1.z, fmaf(c1.z, x2.z, c2.z)), cw.z);
  r.w = sln_vmax(fmaf(c0.w, x1.w, fmaf(c1.w, x2.w, c2.w)), cw.w);
  return r;
}

// single-source operands: fmaf(c.y, 0, c.z) == c.z, so this is xform(x1, 0, cf) bit for bit without keeping c.y alive
__device__ __forceinline__ float4 xform1(float4 x1, const float4* cf) {
  float4 r;
  r.x = sln_vmax(fmaf(cf[0].x, x1.x, cf[0].z), cf[0].w);
  r.y = sln_vmax(fmaf(cf[1].x, x1.y, cf[1].z), cf[1].w);
  r.z = sln_vmax(fmaf(cf[2].x, x1.z, cf[2].z), cf[2].w);
  r.w = sln_vmax(fmaf(cf[3].x, x1.w, cf[3].z), cf[3].w);
  return r;
}

// XCD-aware bijective remap of a linear block id: consecutive logical ids share an XCD (and its L2).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

struct SegSel {   // block-uniform view of the segment that holds logical column k0
  const float* x1; const float* x2; int ld1, ld2, c1, c2, which, base, end;
};

// MULTI = false: the operand has a single segment, so its fields are loop invariants that hipcc keeps in SGPRs.  With
// the select chain below it re-reads the chosen segment's fields from the kernarg segment (s_load_dword) in every K
// tile, and each such read is followed by s_waitcnt lgkmcnt(0), which also drains the LDS queue.
// MULTI = 2: segment boundaries that are not multiples of BK (decoder_cat off at embedding_dim 16: [16 | 16 | 16]); the
// segment is then chosen per THREAD from the column of its float4 (`kc`; segment lengths are multiples of 4, a float4 never
// straddles two segments) instead of per K tile.  Only built for the 64x64 tile.
template <int MULTI>
__device__ __forceinline__ SegSel pick_seg(const Operand& op, int k0, int kc) {
  SegSel r;
  if (MULTI == 2) k0 = kc;
  if (!MULTI) {
    const Seg& g = op.seg[0];
    r.x1 = g.x1; r.x2 = g.x2; r.ld1 = g.ld1; r.ld2 = g.ld2; r.c1 = g.c1; r.c2 = g.c2; r.which = g.which; r.base = 0; r.end = g.len;
    return r;
  }
  const int e0 = op.seg[0].len, e1 = e0 + op.seg[1].len;
  const int s = (op.nseg > 1 && k0 >= e0) ? ((op.nseg > 2 && k0 >= e1) ? 2 : 1) : 0;
  r.x1 = s == 0 ? op.seg[0].x1 : (s == 1 ? op.seg[1].x1 : op.seg[2].x1);
  r.x2 = s == 0 ? op.seg[0].x2 : (s == 1 ? op.seg[1].x2 : op.seg[2].x2);
  r.ld1 = s == 0 ? op.seg[0].ld1 : (s == 1 ? op.seg[1].ld1 : op.seg[2].ld1);
  r.ld2 = s == 0 ? op.seg[0].ld2 : (s == 1 ? op.seg[1].ld2 : op.seg[2].ld2);
  r.c1 = s == 0 ? op.seg[0].c1 : (s == 1 ? op.seg[1].c1 : op.seg[2].c1);
  r.c2 = s == 0 ? op.seg[0].c2 : (s == 1 ? op.seg[1].c2 : op.seg[2].c2);
  r.which = s == 0 ? op.seg[0].which : (s == 1 ? op.seg[1].which : op.seg[2].which);
  r.base = s == 0 ? 0 : (s == 1 ? e0 : e1);
  r.end = r.base + (s == 0 ? op.seg[0].len : (s == 1 ? op.seg[1].len : op.seg[2].len));
  return r;
}

// MULTI = 1, round 4: the three segments' fields loaded ONCE into scalar registers in front of the K loop (sln_seg_preload) and
// chosen by value per K tile (pick_pre: s_cselect on registers).  With pick_seg<1> inside the loop hipcc selected the ADDRESS of
// the chosen segment's fields and re-read them from the kernarg segment: ten s_load_dword and four to five s_waitcnt lgkmcnt(0) -
// each a scalar-cache round trip that also drains the LDS queue - per pair of K tiles in the gathered first Linear of every
// GraphTripleConv (ISA of gemm_nt_kernel<64, 64, 2, 2, 0, 1, 1>).
__device__ __forceinline__ SegSel sln_seg_preload1(const Seg& g, int base) {
  SegSel r;
  r.x1 = g.x1; r.x2 = g.x2; r.ld1 = g.ld1; r.ld2 = g.ld2; r.c1 = g.c1; r.c2 = g.c2; r.which = g.which;
  r.base = base; r.end = base + g.len;
  asm volatile("" : "+s"(r.x1), "+s"(r.x2), "+s"(r.ld1), "+s"(r.ld2), "+s"(r.c1), "+s"(r.c2), "+s"(r.which), "+s"(r.base), "+s"(r.end));
  return r;
}
// (bit masks, not ?: between the three records: hipcc turns a select between aggregate elements into an indexed load from a
// scratch copy - the first version of this function did exactly that, 160 bytes of scratch and seven scratch loads per tile)
__device__ __forceinline__ SegSel pick_pre(const SegSel& p0, const SegSel& p1, const SegSel& p2, int nseg, int e0, int e1, int k0) {
  const int s = (nseg > 1 && k0 >= e0) ? ((nseg > 2 && k0 >= e1) ? 2 : 1) : 0;
  const int m1 = -(int)(s == 1), m2 = -(int)(s == 2), m0 = ~(m1 | m2);
  const unsigned long long M0 = (unsigned long long)(long long)m0, M1 = (unsigned long long)(long long)m1, M2 = (unsigned long long)(long long)m2;
  auto selp = [&](const float* a, const float* b, const float* c) {
    return reinterpret_cast<const float*>((reinterpret_cast<unsigned long long>(a) & M0) | (reinterpret_cast<unsigned long long>(b) & M1) |
                                          (reinterpret_cast<unsigned long long>(c) & M2));
  };
  auto seli = [&](int a, int b, int c) { return (a & m0) | (b & m1) | (c & m2); };
  SegSel r;
  r.x1 = selp(p0.x1, p1.x1, p2.x1); r.x2 = selp(p0.x2, p1.x2, p2.x2);
  r.ld1 = seli(p0.ld1, p1.ld1, p2.ld1); r.ld2 = seli(p0.ld2, p1.ld2, p2.ld2);
  r.c1 = seli(p0.c1, p1.c1, p2.c1); r.c2 = seli(p0.c2, p1.c2, p2.c2);
  r.which = seli(p0.which, p1.which, p2.which);
  r.base = seli(p0.base, p1.base, p2.base); r.end = seli(p0.end, p1.end, p2.end);
  return r;
}

// What the helper wavefronts of an NT kernel do (threads ht = 0 .. nthreads - 1 behind the staging wavefronts): the operand's
// coefficient table (its zero rows up to kend included) and, for the masked epilogue, the forward coefficients of the TW output
// columns from n0.  The epilogue table's loads go out first and its arithmetic comes last: its round trip runs under the operand
// table's instead of behind it.
template <int NSEG_MAX, bool IDENT, int EPI, int TW>
__device__ __forceinline__ void nt_helper_tables(const GemmNTArgs& a, float4* coef, float4* ecoef, const int ht, const int nthreads, const int kend, const int n0) {
  static_assert(TW <= 256, "one epilogue column per helper thread");
  const bool etrain = EPI == EPI_MASK && a.obn.mode == SLN_BN_TRAIN;
  BnFwdRaw er;
  if (etrain) er = bn_fwd_train_load(a.obn, min(n0 + (ht < TW ? ht : 0), a.N - 1));
  if (!IDENT) {
    sln_fill_coefs<NSEG_MAX>(a.A, coef, ht, nthreads);
    for (int c = a.K + ht; c < kend; c += nthreads) coef[sln_cidx(c)] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (EPI == EPI_MASK) {
    if (etrain) {
      const float4 e = bn_fwd_train_finish(a.obn, er);
      if (ht < TW) ecoef[ht] = n0 + ht < a.N ? e : make_float4(1.f, 0.f, 0.f, 1.f);
    } else {
      for (int c = ht; c < TW; c += nthreads) {
        float4 e = make_float4(1.f, 0.f, 0.f, 1.f);
        if (n0 + c < a.N) e = bn_fwd_coef4(a.obn, n0 + c);
        ecoef[c] = e;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// NT kernel
// ---------------------------------------------------------------------------------------------
// // (An intra-block split-K variant, 8 waves per 64x64 tile, was measured and dropped: same time, see DESIGN.md.)
// HELP: the workgroup was launched with 512 threads; wavefronts 4 .. 7 only build the coefficient tables and leave behind the first
// barrier.  The loads of that set-up (gamma / beta and the fp64 column sums the previous kernel left behind with device-scope
// atomics) are the slowest round trip in front of a block's first MFMA, and in the staging waves they queue behind the operand
// loads and hold back - vmcnt retires in order - every later wait.  In wavefronts of their own they go out at once, next to the
// operand loads of the others.
template <int BM, int BN, int WM, int WN, int AMODE, int EPI, int MULTI, bool HELP = false>
__device__ __forceinline__ void gemm_nt_body(const GemmNTArgs& a, const int bid, const int nwg, char* smem) {
  constexpr int NT = 256;
  constexpr bool HAS_X2 = AMODE == 1;        // AMODE: 0 = per-column affine (+relu), 1 = two sources (BatchNorm backward), 2 = identity
  constexpr bool IDENT = AMODE == 2;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int LDT = BK + 4;                 // k-contiguous LDS rows, +4 floats: ds_read_b128 conflict-free
  constexpr int RP = 32;                       // rows staged per pass
  constexpr int PA = BM / RP, PB = BN / RP;
  constexpr int NST = 2;                      // register stages: tiles kt+1, kt+2 (and kt+3 once kt+1 is in LDS) in flight while kt computes
  static_assert(WM * WN == 4, "4 waves per block");
  const int kpad = (a.K + 31) & ~31;
  float4* coef = reinterpret_cast<float4*>(smem);
  float* As = reinterpret_cast<float*>(coef + sln_crows(kpad + 4));  // [2][BM][LDT]; columns kpad .. kpad + 3 of the table = 0 (surplus tiles of the scheduled loop)
  float* Bs = As + 2 * BM * LDT;                          // [2][BN][LDT]
  float4* ecoef = reinterpret_cast<float4*>(Bs + 2 * BN * LDT);
  float* red = reinterpret_cast<float*>(ecoef + BN);      // [WM][BN][2] floats (EPI_MASK) or doubles (EPI_STATS)
  double* redd = reinterpret_cast<double*>(ecoef + BN);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  SLN_TRACE(0);
  // (the helpers' branch comes first and computes the tile origin only for the masked epilogue's table: behind the common tile
  //  arithmetic their kernel-argument loads were a second scalar-cache round trip, ~400 clocks on the block's critical path)
  if (HELP && tid >= NT) {                     // the helper wavefronts: coefficient tables, first barrier, done
    const int ht = tid - NT;
    int n0 = 0;
    if (EPI == EPI_MASK) { const int tiles_n = (a.N + BN - 1) / BN; n0 = (xcd_remap(bid, nwg) % tiles_n) * BN; }
    SLN_TRACEH(8);
    nt_helper_tables<(MULTI ? 3 : 1), AMODE == 2, EPI, BN>(a, coef, ecoef, ht, NT, kpad + 4, n0);
    SLN_TRACEH(9);
    __syncthreads();
    SLN_TRACEH(10);
    return;
  }
  const int tiles_n = (a.N + BN - 1) / BN;
  const int lb = xcd_remap(bid, nwg);
  const int m0 = (lb / tiles_n) * BM, n0 = (lb % tiles_n) * BN;
  const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);

  const int kq = tid & 7, r0 = tid >> 3;
  int ra_idx[PA], rb_idx[PA];
#pragma unroll
  for (int p = 0; p < PA; ++p) {
    const int row = min(m0 + r0 + RP * p, a.M - 1);      // clamped: loads are unconditional, masking happens at the LDS store
    ra_idx[p] = a.A.idx_a ? a.A.idx_a[row] : row;
    rb_idx[p] = a.A.idx_b ? a.A.idx_b[row] : row;
  }

  int rid[PA];                                // source row of each staged row (single segment: gather resolved once)
#pragma unroll
  for (int p = 0; p < PA; ++p) {
    const int row = min(m0 + r0 + RP * p, a.M - 1);
    const int w0 = a.A.seg[0].which;
    rid[p] = MULTI ? row : (w0 == 0 ? row : (w0 == 1 ? ra_idx[p] : rb_idx[p]));
  }
  SLN_TRACE(5);
  constexpr bool kPreload = MULTI == 1;
  const int pe0 = a.A.seg[0].len, pe1 = pe0 + a.A.seg[1].len, pnseg = a.A.nseg;
  SegSel pre0 = {}, pre1 = {}, pre2 = {};
  if (kPreload) { pre0 = sln_seg_preload1(a.A.seg[0], 0); pre1 = sln_seg_preload1(a.A.seg[1], pe0); pre2 = sln_seg_preload1(a.A.seg[2], pe1); }
#define SLN_PICK(k0_, kc_) (kPreload ? pick_pre(pre0, pre1, pre2, pnseg, pe0, pe1, (k0_)) : pick_seg<MULTI>(a.A, (k0_), (kc_)))
  float4 ga1[NST][PA], ga2[NST][PA], gb[NST][PB];
  const int ntiles = kpad / BK;
  const int last = ntiles - 1;
  const float* Wp = a.W; const int ldw = a.ldw, Mr = a.M, Nr = a.N, Kr = a.K;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);

  // NOTE: every global load below is unconditional (addresses clamped into the operand); a
  // `valid ? load : 0` select makes hipcc branch around each load and drain vmcnt(0) per element,
  // which serialises the whole register pipeline.  Out-of-range lanes are zeroed in lstore().
  auto gload = [&](int kt, auto stage) {
    constexpr int S = decltype(stage)::value;
    const int k0 = kt * BK;
    const SegSel sg = SLN_PICK(k0, k0 + 4 * kq);
    const int cs = min(k0 + 4 * kq, sg.end - 4) - sg.base;      // column inside the segment, clamped
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const int r = MULTI ? (sg.which == 0 ? rid[p] : (sg.which == 1 ? ra_idx[p] : rb_idx[p])) : rid[p];
      ga1[S][p] = ld4(sg.x1 + (size_t)r * sg.ld1 + sg.c1 + cs);
      if (HAS_X2) {
        const float* x2 = sg.x2 ? sg.x2 : sg.x1;                 // block-uniform select
        const int ld2 = sg.x2 ? sg.ld2 : sg.ld1, c2 = sg.x2 ? sg.c2 : sg.c1;
        ga2[S][p] = ld4(x2 + (size_t)r * ld2 + c2 + cs);
      }
    }
    const int cw = min(k0 + 4 * kq, Kr - 4);
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      const int n = min(n0 + r0 + RP * p, Nr - 1);
      gb[S][p] = ld4(Wp + (size_t)n * ldw + cw);
    }
  };
  // parts: 1 = the operand rows (A: needs the coefficient table), 2 = the weight rows (B), 3 = both
  auto lstore = [&](int kt_raw, int buf, auto stage, const int parts = 3) __attribute__((always_inline)) {
    constexpr int S = decltype(stage)::value;
    const int kt = min(kt_raw, last);
    const int k0 = kt * BK, col = k0 + 4 * kq;
    if (parts & 1) {
      const SegSel sg = SLN_PICK(k0, col);
      const bool cv = col < sg.end && kt_raw <= last;          // surplus tiles (loop padded to a multiple of 3) are stored as zeros
      const bool x2v = HAS_X2 && sg.x2 != nullptr;
      float* as = As + buf * BM * LDT + 4 * kq;
      const float4* cf = coef + sln_cidx(min(col, kpad - 4));
#pragma unroll
      for (int p = 0; p < PA; ++p) {
        const int rl = r0 + RP * p;
        const bool v = cv && (m0 + rl) < Mr;
        float4 t = IDENT ? ga1[S][p] : xform(ga1[S][p], x2v ? ga2[S][p] : z4, cf);
        t.x = v ? t.x : 0.f; t.y = v ? t.y : 0.f; t.z = v ? t.z : 0.f; t.w = v ? t.w : 0.f;
        *reinterpret_cast<float4*>(as + rl * LDT) = t;
      }
    }
    if (parts & 2) {
      float* bs = Bs + buf * BN * LDT + 4 * kq;
      const bool kv = col < Kr && kt_raw <= last;
#pragma unroll
      for (int p = 0; p < PB; ++p) {
        const bool v = kv && (n0 + r0 + RP * p) < Nr;
        float4 t = gb[S][p];
        t.x = v ? t.x : 0.f; t.y = v ? t.y : 0.f; t.z = v ? t.z : 0.f; t.w = v ? t.w : 0.f;
        *reinterpret_cast<float4*>(bs + (r0 + RP * p) * LDT) = t;
      }
    }
  };

  // issue the first tiles' loads before the (dependent, sqrt-heavy) coefficient set-up
  using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
  gload(0, S0{});
  gload(min(1, last), S1{});
  SLN_TRACE(6);

  if (!HELP) {
    if (!IDENT) {
      sln_fill_coefs<(MULTI ? 3 : 1)>(a.A, coef, tid, NT);
      for (int c = a.K + tid; c < kpad + 4; c += NT) coef[sln_cidx(c)] = z4;
    }
    if (EPI == EPI_MASK) {
      for (int c = tid; c < BN; c += NT) {
        float4 e = make_float4(1.f, 0.f, 0.f, 1.f);
        if (n0 + c < a.N) {
          e = bn_fwd_coef4(a.obn, n0 + c);
        }
        ecoef[c] = e;
      }
    }
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // what the epilogue reads from memory is requested HERE, in front of the K loop: behind it the bias (and, for the masked dgrad,
  // the 16 pre-activations of a 64 x 64 tile's lane) were one more memory round trip between the last MFMA and the first store
  float bias_pre[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = n0 + wn0 + 32 * j + (lane & 31);
    bias_pre[j] = a.bias ? a.bias[min(col, a.N - 1)] : 0.f;
  }
  constexpr bool XP_PRE = EPI == EPI_MASK && TM == 1 && TN == 1;
  float xp_pre[XP_PRE ? 16 : 1];
  if (XP_PRE) {
    const int ccl = min(n0 + wn0 + (lane & 31), a.N - 1);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = min(m0 + wm0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), a.M - 1);
      xp_pre[r] = a.xprev[(size_t)row * a.ldx + a.xcol0 + ccl];
    }
  }

  // The staging wavefronts reach this barrier before the coefficient tables are written (tools/lab/gemm_lab.hip: 2 300-2 800 clocks
  // against 3 200-3 900 for the helpers): the weight rows of the first tile - they need no table - go to LDS while they would wait.
  lstore(0, 0, S0{}, 2);
  SLN_TRACE(7);
  __syncthreads();            // coef tables visible
  SLN_TRACE(1);
  lstore(0, 0, S0{}, 1);
  gload(min(2, last), S0{});
  __syncthreads();
  SLN_TRACE(2);
  const int lrow = lane & 31, lk = lane >> 5;

  // Fragments ping-pong between two register sets; on entry to body(kt) set 0 already holds the first 8-wide k chunk of
  // tile kt.  It was read right after the barrier that ended body(kt-1), under the MFMAs of that tile's last chunk, so
  // neither the barrier nor the LDS read latency leaves the matrix pipe idle (each wave has a single dependent MFMA
  // chain when the wave tile is 32x32, nothing else could cover them).
  float4 fa[2][TM], fb[2][TN];
  auto rd = [&](int buf, int kb, auto set) __attribute__((always_inline)) {
    constexpr int F = decltype(set)::value;
    const float* as = As + buf * BM * LDT + (wm0 + lrow) * LDT + 4 * lk + kb;
    const float* bs = Bs + buf * BN * LDT + (wn0 + lrow) * LDT + 4 * lk + kb;
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[F][i] = *reinterpret_cast<const float4*>(as + 32 * i * LDT);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[F][j] = *reinterpret_cast<const float4*>(bs + 32 * j * LDT);
  };
  auto mma = [&](auto set) {
    constexpr int F = decltype(set)::value;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[F][i].x, fb[F][j].x, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[F][i].y, fb[F][j].y, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[F][i].z, fb[F][j].z, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[F][i].w, fb[F][j].w, acc[i][j], 0, 0, 0);
      }
  };
  rd(0, 0, S0{});

  // Tile j lives in register stage j % 2.  body(kt) stages tile kt+1 from its register stage into LDS, refills that stage
  // with tile kt+3, and computes tile kt from LDS.  Every body issues the SAME loads unconditionally (tile indices clamped;
  // surplus tiles are stored as zeros) so that the loop is one straight block.
  // The next tile is staged (and its register stage refilled) FIRST, under the MFMAs the previous body issued last: hipcc
  // waits for ALL loads in flight at an lstore (vmcnt(0), never a partial count - seen in the ISA of every formulation tried),
  // so the youngest load must be a whole tile old by then: with the refill issued right behind the lstore it is.
  auto body = [&](int kt, auto stage_next) {
    const int buf = kt & 1;
    lstore(kt + 1, buf ^ 1, stage_next);
    gload(min(kt + 3, last), stage_next);
    __builtin_amdgcn_sched_barrier(0);      // hipcc otherwise sinks these loads to the end of the body, half a tile before their wait
    rd(buf, 8, S1{});
    mma(S0{});
    rd(buf, 16, S0{});
    mma(S1{});
    rd(buf, 24, S1{});
    mma(S0{});
    __syncthreads();
    rd(buf ^ 1, 0, S0{});
    mma(S1{});
  };
  // ONE straight loop over pairs of tiles; an odd tile count (K % 64 == 32: only box_net's 2E + E/4 columns) runs one
  // surplus tile of zeros.  With three stages and separate remainder bodies behind the loop (round 1) hipcc could not keep the
  // register stages in place: it copied all of them (and the accumulators) at the top of every iteration, and a copy of a
  // register with a load in flight is an s_waitcnt vmcnt(0) - 16 v_mov_b64 + 16 v_accvgpr_mov behind a vmcnt(0) per 3 tiles.
#if SLN_NT_SCHED
  // ---- 64 x 64 tile: the same loop with its instruction ORDER written out ---------------------------------------------------
  // A wavefront owns one 32 x 32 accumulator: its 16 MFMAs per K tile form ONE dependent chain, issue is in order, so MFMA n+1
  // waits at the issue stage until MFMA n leaves the pipe (64 cycles) - and everything hipcc schedules BEHIND a group of four
  // MFMAs (the staging arithmetic of the next tile, its LDS writes, the fragment reads, the global loads) only starts when the
  // last of the four has issued: it overlaps with 64 of the group's 256 cycles and the rest runs with the matrix pipe idle
  // (tools/lab/gemm_lab.hip: ~1 760 clocks per K tile for ~1 000 of MFMA, one workgroup per CU).  Here every MFMA is followed by
  // the piece of other work that fits under it, and sched_barrier(0) keeps hipcc from regrouping:
  //   c0.x  B rows 0-31: mask, ds_write, reload        c1.x  A rows 32-63: transform        c2.*  (nothing: LDS writes land,
  //   c0.y  B rows 32-63: the same                     c1.y  A rows 32-63: mask, write,            the loads fly)
  //   c0.z  A rows 0-31: BatchNorm / ReLU transform          reload                          -- barrier --
  //   c0.w  A rows 0-31: mask, write, reload;          c1.w  fragments of chunk 3            c3.x  next tile's segment / masks
  //         fragments of chunk 2                                                             c3.y  next tile's coefficients
  // A register stage is reloaded in the slot that stored it (two K tiles of flight time for every load); the per-tile scalars
  // and the four coefficient rows of the NEXT body are fetched under the four MFMAs behind the barrier (loop-carried registers).
  // Arithmetic and summation order are those of the loop above: results are bit-identical.
  if constexpr (TM == 1 && TN == 1) {
    // No masks in this loop.  Rows behind M and weight rows behind N are loaded from clamped (valid) addresses and only reach output
    // rows / columns that the epilogue drops.  Columns behind K: the staged operand is EXACTLY zero there because its coefficient
    // rows are (coef[K .. kpad) and the four rows at coef[kpad], which a surplus tile of an odd tile count is pointed at), so the
    // clamped - finite - weight columns meet zeros.  Identity operands (no coefficients) keep a column mask.  A v_cndmask costs 8
    // clocks of the SIMD and VALU work does NOT run in the shadow of the wave's own MFMAs (tools/lab/overlap.hip: 64 clocks per
    // MFMA alone, + 4.7 per v_fma, + 8 per v_cndmask behind it; half of that with a second wave on the SIMD).
    struct Plan { bool cv, x2v; int coff; unsigned cw, cs; const float* pa1[PA]; const float* pa2[PA]; };
    const float* rowB[PB];
#pragma unroll
    for (int q = 0; q < PB; ++q) rowB[q] = Wp + (size_t)min(n0 + r0 + RP * q, Nr - 1) * ldw;
    const float* rowA1[PA]; const float* rowA2[PA];      // single-segment operands: the row pointers are loop invariants
#pragma unroll
    for (int q = 0; q < PA; ++q) {
      const Seg& g = a.A.seg[0];
      rowA1[q] = g.x1 + (size_t)rid[q] * g.ld1 + g.c1;
      rowA2[q] = g.x2 ? g.x2 + (size_t)rid[q] * g.ld2 + g.c2 : rowA1[q];
    }
    auto plan = [&](int kt) __attribute__((always_inline)) {               // body(kt) stores tile kt + 1 and loads tile kt + 3
      Plan p;
      const int ks_raw = kt + 1, ks = min(ks_raw, last), k0 = ks * BK, col = k0 + 4 * kq;
      const SegSel ss = SLN_PICK(k0, col);
      p.cv = col < ss.end && ks_raw <= last;
      p.x2v = HAS_X2 && ss.x2 != nullptr;
      p.coff = ks_raw <= last ? col : kpad;
      const int l0 = min(kt + 3, last) * BK;
      const SegSel sl = SLN_PICK(l0, l0 + 4 * kq);
      p.cs = (unsigned)(min(l0 + 4 * kq, sl.end - 4) - sl.base);
      p.cw = (unsigned)min(l0 + 4 * kq, Kr - 4);
      if (MULTI) {
        const float* x2 = sl.x2 ? sl.x2 : sl.x1;                   // block-uniform select
        const int ld2 = sl.x2 ? sl.ld2 : sl.ld1, c2 = sl.x2 ? sl.c2 : sl.c1;
        const int mw1 = -(int)(sl.which == 1), mw2 = -(int)(sl.which == 2);
#pragma unroll
        for (int q = 0; q < PA; ++q) {
          // (bit masks instead of ?: - hipcc turns a select between array elements into an indexed load from a scratch copy)
          const int r = (rid[q] & ~(mw1 | mw2)) | (ra_idx[q] & mw1) | (rb_idx[q] & mw2);
          p.pa1[q] = sl.x1 + (size_t)r * sl.ld1 + sl.c1;
          p.pa2[q] = HAS_X2 ? x2 + (size_t)r * ld2 + c2 : nullptr;
        }
      } else {
#pragma unroll
        for (int q = 0; q < PA; ++q) { p.pa1[q] = nullptr; p.pa2[q] = nullptr; }      // unused: rowA1 / rowA2
      }
      return p;
    };
    Plan pl = plan(0);
    float4 cfr[4] = {z4, z4, z4, z4};
    if (!IDENT) {
#pragma unroll
      for (int j = 0; j < 4; ++j) cfr[j] = coef[sln_cidx(pl.coff) + j];
    }
#define SLN_SB __builtin_amdgcn_sched_barrier(0)
    auto stB = [&](int buf, int p, auto stage) __attribute__((always_inline)) {
      constexpr int S = decltype(stage)::value;
      if (!(SLN_ABL & 2)) *reinterpret_cast<float4*>(Bs + buf * BN * LDT + (r0 + RP * p) * LDT + 4 * kq) = gb[S][p];
      SLN_SB;      // the reload stays behind the write (hipcc otherwise loads into fresh registers and copies them later)
      if (!(SLN_ABL & 1)) gb[S][p] = ld4(rowB[p] + pl.cw);
    };
    auto xfA = [&](int p, auto stage) __attribute__((always_inline)) -> float4 {
      constexpr int S = decltype(stage)::value;
      // a segment without a second source has c.y == 0 (sln_coef_for) and its ga2 holds x1 again: fma(0, x1, c.z) = c.z, no select
      return IDENT ? ga1[S][p] : (HAS_X2 ? xform(ga1[S][p], ga2[S][p], cfr) : xform1(ga1[S][p], cfr));
    };
    auto wrA = [&](int buf, int p, float4 t, auto stage) __attribute__((always_inline)) {
      constexpr int S = decltype(stage)::value;
      if (IDENT) { const bool v = pl.cv; t.x = v ? t.x : 0.f; t.y = v ? t.y : 0.f; t.z = v ? t.z : 0.f; t.w = v ? t.w : 0.f; }
      if (!(SLN_ABL & 2)) *reinterpret_cast<float4*>(As + buf * BM * LDT + (r0 + RP * p) * LDT + 4 * kq) = t;
      SLN_SB;
      if (!(SLN_ABL & 1)) {
        ga1[S][p] = ld4((MULTI ? pl.pa1[p] : rowA1[p]) + pl.cs);
        if (HAS_X2) ga2[S][p] = ld4((MULTI ? pl.pa2[p] : rowA2[p]) + pl.cs);
      }
    };
    auto mf = [&](auto set, auto comp) __attribute__((always_inline)) {
      constexpr int F = decltype(set)::value;
      constexpr int C = decltype(comp)::value;
      const float av = C == 0 ? fa[F][0].x : (C == 1 ? fa[F][0].y : (C == 2 ? fa[F][0].z : fa[F][0].w));
      const float bv = C == 0 ? fb[F][0].x : (C == 1 ? fb[F][0].y : (C == 2 ? fb[F][0].z : fb[F][0].w));
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[0][0], 0, 0, 0);
    };
    using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>; using C3 = std::integral_constant<int, 3>;
    auto sbody = [&](int kt, auto stage_next) __attribute__((always_inline)) {
      const int buf = kt & 1;
      if (!(SLN_ABL & 8)) rd(buf, 8, S1{});
      SLN_SB;
      mf(S0{}, C0{}); SLN_SB; stB(buf ^ 1, 0, stage_next); SLN_SB;
      mf(S0{}, C1{}); SLN_SB; stB(buf ^ 1, 1, stage_next); SLN_SB;
      mf(S0{}, C2{}); SLN_SB; float4 t0 = xfA(0, stage_next); SLN_SB;
      mf(S0{}, C3{}); SLN_SB; wrA(buf ^ 1, 0, t0, stage_next); if (!(SLN_ABL & 8)) rd(buf, 16, S0{}); SLN_SB;
      mf(S1{}, C0{}); SLN_SB; float4 t1 = xfA(1, stage_next); SLN_SB;
      mf(S1{}, C1{}); SLN_SB; wrA(buf ^ 1, 1, t1, stage_next); SLN_SB;
      mf(S1{}, C2{}); SLN_SB;
      mf(S1{}, C3{}); SLN_SB; if (!(SLN_ABL & 8)) rd(buf, 24, S1{}); SLN_SB;
      mf(S0{}, C0{}); SLN_SB;
      mf(S0{}, C1{}); SLN_SB;
      mf(S0{}, C2{}); SLN_SB;
      mf(S0{}, C3{}); SLN_SB;
      if (!(SLN_ABL & 4)) __syncthreads();
      if (!(SLN_ABL & 8)) rd(buf ^ 1, 0, S0{});
      SLN_SB;
      mf(S1{}, C0{}); SLN_SB; pl = plan(kt + 1); SLN_SB;
      mf(S1{}, C1{}); SLN_SB;
      if (!IDENT) {
#pragma unroll
        for (int j = 0; j < 4; ++j) cfr[j] = coef[sln_cidx(pl.coff) + j];
      }
      SLN_SB;
      mf(S1{}, C2{}); SLN_SB;
      mf(S1{}, C3{}); SLN_SB;
    };
#undef SLN_SB
    for (int kt = 0; kt < ntiles; kt += 2) { sbody(kt, S1{}); sbody(kt + 1, S0{}); }
  } else
#endif
  for (int kt = 0; kt < ntiles; kt += 2) { body(kt, S1{}); body(kt + 1, S0{}); }

  // ------------------------------- epilogue -------------------------------
  SLN_TRACE(3);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int cl = wn0 + 32 * j + lrow;          // column inside the block tile
    const int col = n0 + cl;
    const bool cvalid = col < a.N;
    const float bias = cvalid ? bias_pre[j] : 0.f;
    float4 ec = make_float4(1.f, 0.f, 0.f, 1.f);
    if (EPI == EPI_MASK) ec = ecoef[cl];
    float s1 = 0.f, s2 = 0.f;
    // forward BatchNorm statistics in fp64 (y * y is exact there): var = E[y^2] - mean^2 cancels, and with fp32 partial sums a
    // column whose rows nearly agree (mean >> sigma: 8-row batches, BASELINE config c1) lost 1e-7 (mean / sigma)^2 of its
    // variance - torch's two-pass batch_norm does not.  Two fp64 ops per output element next to 2K MFMA flops.
    double d1 = 0.0, d2 = 0.0;
    // Loads first (addend, xprev: unconditional, clamped rows), then arithmetic, then the stores: written element by element
    // (load, wait, store, wait for the store before the next conditional load) the 16 values of a lane cost 5 000 cycles.
    const int ccl = cvalid ? col : 0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float yv[16], xpv[16];
      const bool has_add = a.addend != nullptr;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = min(m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lk, a.M - 1);
        yv[r] = acc[i][j][r] + bias;
        xpv[r] = 0.f;
        if (EPI == EPI_MASK) xpv[r] = XP_PRE ? xp_pre[r] : a.xprev[(size_t)row * a.ldx + a.xcol0 + ccl];
      }
      if (has_add) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = min(m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lk, a.M - 1);
          yv[r] += a.addend[(size_t)row * a.ldadd + a.addcol0 + ccl];
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lk;
        const bool ok = cvalid && row < a.M;
        float y = yv[r];
        if (EPI == EPI_MASK) y = fmaf(ec.x, xpv[r], ec.y) > 0.f ? y : 0.f;
        y = ok ? y : 0.f;
        if (EPI == EPI_STATS) { d1 += (double)y; d2 = fma((double)y, (double)y, d2); }
        if (EPI == EPI_MASK) { s1 += y; s2 = fmaf(y, (xpv[r] - ec.z) * ec.w, s2); }
        if (ok) a.Y[(size_t)row * a.ldy + a.ycol0 + col] = y;
      }
    }
    if (EPI == EPI_STATS) {
      d1 += __shfl_xor(d1, 32, 64); d2 += __shfl_xor(d2, 32, 64);
      if (lk == 0) { redd[((wave / WN) * BN + cl) * 2 + 0] = d1; redd[((wave / WN) * BN + cl) * 2 + 1] = d2; }
    } else if (EPI != EPI_PLAIN) {
      s1 = wave_sum_halves(s1); s2 = wave_sum_halves(s2);
      if (lk == 0) { red[((wave / WN) * BN + cl) * 2 + 0] = s1; red[((wave / WN) * BN + cl) * 2 + 1] = s2; }
    }
  }
  if (EPI != EPI_PLAIN) {
    __syncthreads();
    double* out = (EPI == EPI_STATS) ? a.osums : a.ogsums;
    if (out != nullptr) {
      for (int c = tid; c < BN; c += NT) {
        if (n0 + c < a.N) {
          double s1 = 0.0, s2 = 0.0;
#pragma unroll
          for (int w = 0; w < WM; ++w) {
            if (EPI == EPI_STATS) { s1 += redd[(w * BN + c) * 2]; s2 += redd[(w * BN + c) * 2 + 1]; }
            else { s1 += (double)red[(w * BN + c) * 2]; s2 += (double)red[(w * BN + c) * 2 + 1]; }
          }
          const double qs = EPI == EPI_STATS ? sln_q_fwd(a.M) : SLN_Q_BWD;      // order-independent sums (sln_common.h)
          atomicAdd(out + n0 + c, sln_qd(s1, qs));
          atomicAdd(out + a.ocstride + n0 + c, sln_qd(s2, qs));
        }
      }
    }
  }
  SLN_TRACE(4);
#undef SLN_PICK
}


// ---------------------------------------------------------------------------------------------
// NT kernel, 64 x 32J tile on v_mfma_f32_16x16x4_f32: ONE workgroup per CU for the triple-side widths
// ---------------------------------------------------------------------------------------------
// At 64 graphs the triple-side Linears have M = 4096 rows: 64 row blocks.  With 64 x 64 tiles N = 640 (net1's second Linear) is
// 640 tiles = 2.5 per CU and N = 384 (the dgrad of net1's first Linear) 1.5 per CU: the launch lasts as long as the CUs that got
// the extra tile.  Here the tile is 64 x 160 (J = 5) or 64 x 96 (J = 3): 256 tiles, every CU gets the same work.  A wavefront owns
// 32 rows x 16J columns as 2 x J accumulators of 16 x 16 (4 registers each), so its MFMAs of a K chunk are independent of one
// another, and a K tile is two chunks of 16 (lane l reads A[row l % 16][4 (l / 16) .. + 3] with one ds_read_b128: the four MFMAs
// that consume .x .. .w each see k = c, c + 4, c + 8, c + 12 - any fixed permutation of k is as good as another, A and B use the
// same one).  Single-segment operands only.  Staging, masks (none: zero coefficient rows) and the statistics are those of
// gemm_nt_body; sums are taken in a different order (per 16 x 16 block), results agree to fp32 rounding.
template <int J, int AMODE, int EPI, bool HELP = false>      // HELP: 512 threads, wavefronts 4 .. 7 build the coefficient tables (see gemm_nt_body)
__device__ __forceinline__ void gemm_nt_body16(const GemmNTArgs& a, const int bid, const int nwg, char* smem) {
  constexpr int NT = 256, BM = 64, BN = 32 * J, WN16 = 16 * J;
  constexpr bool HAS_X2 = AMODE == 1;
  constexpr bool IDENT = AMODE == 2;
  // LDS rows of BK + 8 floats: the 16 x 16 fragment pattern (lane l: row l % 16, floats 4 (l / 16) ..) is conflict-free at a row
  // stride of 40 floats and 2-way conflicted at the 36 of the 32 x 32 body (ds_read_b128 is served in four groups of 16 lanes over
  // 64 banks, MI355X_MICROARCH.md)
  constexpr int LDT = BK + 8, RP = 32, PA = BM / RP, PB = BN / RP;
  const int kpad = (a.K + 31) & ~31;
  float4* coef = reinterpret_cast<float4*>(smem);               // [kpad + 4], the last four rows zero
  float* As = reinterpret_cast<float*>(coef + sln_crows(kpad + 4));        // [2][BM][LDT]
  float* Bs = As + 2 * BM * LDT;                                // [2][BN][LDT]
  float4* ecoef = reinterpret_cast<float4*>(Bs + 2 * BN * LDT); // [BN]
  double* redd = reinterpret_cast<double*>(ecoef + BN);         // [2 row halves][BN][2]
  float* red = reinterpret_cast<float*>(redd);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  SLN_TRACE(0);
  const int wr = wave >> 1, wc = wave & 1;
  if (HELP && tid >= NT) {                     // first, see gemm_nt_body
    int n0 = 0;
    if (EPI == EPI_MASK) { const int tiles_n = (a.N + BN - 1) / BN; n0 = (xcd_remap(bid, nwg) % tiles_n) * BN; }
    nt_helper_tables<1, IDENT, EPI, BN>(a, coef, ecoef, tid - NT, NT, kpad + 4, n0);
    __syncthreads();
    return;
  }
  const int tiles_n = (a.N + BN - 1) / BN;
  const int lb = xcd_remap(bid, nwg);
  const int m0 = (lb / tiles_n) * BM, n0 = (lb % tiles_n) * BN;
  const int kq = tid & 7, r0 = tid >> 3;
  const Seg& g = a.A.seg[0];
  const float* rowA1[PA]; const float* rowA2[PA]; const float* rowB[PB];
#pragma unroll
  for (int p = 0; p < PA; ++p) {
    const int row = min(m0 + r0 + RP * p, a.M - 1);
    const int r = g.which == 0 ? row : (g.which == 1 ? a.A.idx_a[row] : a.A.idx_b[row]);
    rowA1[p] = g.x1 + (size_t)r * g.ld1 + g.c1;
    rowA2[p] = g.x2 ? g.x2 + (size_t)r * g.ld2 + g.c2 : rowA1[p];
  }
#pragma unroll
  for (int p = 0; p < PB; ++p) rowB[p] = a.W + (size_t)min(n0 + r0 + RP * p, a.N - 1) * a.ldw;
  const int ntiles = kpad / BK, last = ntiles - 1;
  const int Kr = a.K, glen = g.len;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 ga1[2][PA], ga2[2][PA], gb[2][PB];
  unsigned cs_n = 0, cw_n = 0;              // column offsets of the loads the next body issues (set under the MFMAs before it)
  int coff_n = 0; bool cv_n = true;         // coefficient row / identity-operand column mask of the tile the next body stages
  auto plan = [&](int kt) __attribute__((always_inline)) {        // for body(kt): stores tile kt + 1, loads tile kt + 3
    const int ks_raw = kt + 1, col = min(ks_raw, last) * BK + 4 * kq;
    cv_n = col < Kr && ks_raw <= last;
    coff_n = ks_raw <= last ? col : kpad;                          // surplus tile (odd tile count): the zero rows
    const int l0 = min(kt + 3, last) * BK + 4 * kq;
    cs_n = (unsigned)min(l0, glen - 4); cw_n = (unsigned)min(l0, Kr - 4);
  };
  auto ldB = [&](int p, auto stage, unsigned cw) __attribute__((always_inline)) {
    constexpr int S = decltype(stage)::value;
    gb[S][p] = ld4(rowB[p] + cw);
  };
  auto ldA = [&](int p, auto stage, unsigned cs) __attribute__((always_inline)) {
    constexpr int S = decltype(stage)::value;
    ga1[S][p] = ld4(rowA1[p] + cs);
    if (HAS_X2) ga2[S][p] = ld4(rowA2[p] + cs);
  };
  auto stB = [&](int buf, int p, auto stage) __attribute__((always_inline)) {
    constexpr int S = decltype(stage)::value;
    *reinterpret_cast<float4*>(Bs + buf * BN * LDT + (r0 + RP * p) * LDT + 4 * kq) = gb[S][p];
  };
  auto xfA = [&](int p, auto stage, const float4* cf, bool cv) __attribute__((always_inline)) -> float4 {
    constexpr int S = decltype(stage)::value;
    float4 t = IDENT ? ga1[S][p] : (HAS_X2 ? xform(ga1[S][p], ga2[S][p], cf) : xform1(ga1[S][p], cf));
    if (IDENT) { t.x = cv ? t.x : 0.f; t.y = cv ? t.y : 0.f; t.z = cv ? t.z : 0.f; t.w = cv ? t.w : 0.f; }
    return t;
  };
  auto wrA = [&](int buf, int p, float4 t) __attribute__((always_inline)) {
    *reinterpret_cast<float4*>(As + buf * BM * LDT + (r0 + RP * p) * LDT + 4 * kq) = t;
  };
  using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
  {   // tiles 0 and 1 into the two register stages
    const unsigned c0s = (unsigned)min(4 * kq, glen - 4), c0w = (unsigned)min(4 * kq, Kr - 4);
    const int l1 = min(1, last) * BK + 4 * kq;
    const unsigned c1s = (unsigned)min(l1, glen - 4), c1w = (unsigned)min(l1, Kr - 4);
#pragma unroll
    for (int p = 0; p < PB; ++p) ldB(p, S0{}, c0w);
#pragma unroll
    for (int p = 0; p < PA; ++p) ldA(p, S0{}, c0s);
#pragma unroll
    for (int p = 0; p < PB; ++p) ldB(p, S1{}, c1w);
#pragma unroll
    for (int p = 0; p < PA; ++p) ldA(p, S1{}, c1s);
  }
  if (!HELP) {
    if (!IDENT) {
      sln_fill_coefs<1>(a.A, coef, tid, NT);
      for (int c = a.K + tid; c < kpad + 4; c += NT) coef[sln_cidx(c)] = z4;
    }
    if (EPI == EPI_MASK) {
      for (int c = tid; c < BN; c += NT) {
        float4 e = make_float4(1.f, 0.f, 0.f, 1.f);
        if (n0 + c < a.N) e = bn_fwd_coef4(a.obn, n0 + c);
        ecoef[c] = e;
      }
    }
  }
  f32x4 acc[2][J];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < J; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bias_pre[J];                    // requested in front of the K loop (see gemm_nt_body)
#pragma unroll
  for (int j = 0; j < J; ++j) bias_pre[j] = a.bias ? a.bias[min(n0 + WN16 * wc + 16 * j + (lane & 15), a.N - 1)] : 0.f;
#pragma unroll
  for (int p = 0; p < PB; ++p) stB(0, p, S0{});       // tile 0's weight rows: in front of the barrier (see gemm_nt_body)
  __syncthreads();
  SLN_TRACE(1);
  {   // tile 0 into LDS buffer 0, tile 2 into stage 0
    const bool cv0 = 4 * kq < Kr;
    const float4* cf0 = coef + sln_cidx(4 * kq);
#pragma unroll
    for (int p = 0; p < PA; ++p) wrA(0, p, xfA(p, S0{}, cf0, cv0));
    const int l2 = min(2, last) * BK + 4 * kq;
    const unsigned c2s = (unsigned)min(l2, glen - 4), c2w = (unsigned)min(l2, Kr - 4);
#pragma unroll
    for (int p = 0; p < PB; ++p) ldB(p, S0{}, c2w);
#pragma unroll
    for (int p = 0; p < PA; ++p) ldA(p, S0{}, c2s);
  }
  __syncthreads();
  SLN_TRACE(2);

  const int l16 = lane & 15, lg = lane >> 4;
  float4 fa[2][2], fb[2][J];
  // fragment q of a chunk: q < 2 the A row blocks, else the B column blocks
  auto rd1 = [&](int buf, int kb, auto set, int q) __attribute__((always_inline)) {
    constexpr int F = decltype(set)::value;
    if (q < 2) fa[F][q] = *reinterpret_cast<const float4*>(As + buf * BM * LDT + (32 * wr + 16 * q + l16) * LDT + 4 * lg + kb);
    else fb[F][q - 2] = *reinterpret_cast<const float4*>(Bs + buf * BN * LDT + (WN16 * wc + 16 * (q - 2) + l16) * LDT + 4 * lg + kb);
  };
#define SLN_SB __builtin_amdgcn_sched_barrier(0)
  // the 8J MFMAs of a chunk, component-major (consecutive MFMAs go to different accumulators), hook(n) behind the n-th
  auto mma = [&](auto set, auto&& hook) __attribute__((always_inline)) {
    constexpr int F = decltype(set)::value;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < J; ++j) {
          const float av = c == 0 ? fa[F][i].x : (c == 1 ? fa[F][i].y : (c == 2 ? fa[F][i].z : fa[F][i].w));
          const float bv = c == 0 ? fb[F][j].x : (c == 1 ? fb[F][j].y : (c == 2 ? fb[F][j].z : fb[F][j].w));
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i][j], 0, 0, 0);
          SLN_SB;
          hook((c * 2 + i) * J + j);
          SLN_SB;
        }
  };
#pragma unroll
  for (int q = 0; q < 2 + J; ++q) rd1(0, 0, S0{}, q);
  plan(0);
  // body(kt): chunk 0 of tile kt sits in fragment set 0.  Memory instructions issued back to back stall the wave at the issue stage
  // (tools/lab/overlap.hip: a second ds_write_b128 behind an MFMA costs 52 clocks, a second global_load_dwordx4 64) and the MFMAs
  // queue up behind them, so every one of them gets its own MFMA to hide under:
  //   chunk 0   odd slots 1, 3, ..: the 2 + J fragment reads of chunk 1
  //             slots 4p (p < J): weight rows p: ds_write + the reload of that register stage
  //             slots 4J, 4J + 2 / 4J + 4, 4J + 6: operand rows: transform, then ds_write + reload
  //   -- barrier --
  //   chunk 1   odd slots: the fragment reads of the next tile's chunk 0; slot 0: the next body's column offsets / coefficient row
  float4 cfr[4] = {z4, z4, z4, z4};          // coefficient rows of the tile the next body stages (fetched one per MFMA in chunk 1)
  if (!IDENT) {
#pragma unroll
    for (int q = 0; q < 4; ++q) cfr[q] = coef[sln_cidx(coff_n) + q];
  }
  auto body = [&](int kt, auto stage_next) __attribute__((always_inline)) {
    const int buf = kt & 1;
    const float4* cf = cfr;
    const bool cv = cv_n;
    const unsigned cs = cs_n, cw = cw_n;
    float4 t0 = z4, t1 = z4;
    mma(S0{}, [&](int n) __attribute__((always_inline)) {
      if ((n & 1) && (n >> 1) < 2 + J) rd1(buf, 16, S1{}, n >> 1);
      // (the fence keeps the reload behind the write: hipcc otherwise loads into fresh registers first and copies them later)
      if (!(n & 3) && (n >> 2) < J) { stB(buf ^ 1, n >> 2, stage_next); SLN_SB; ldB(n >> 2, stage_next, cw); }
      if (n == 4 * J) t0 = xfA(0, stage_next, cf, cv);
      if (n == 4 * J + 2) { wrA(buf ^ 1, 0, t0); SLN_SB; ldA(0, stage_next, cs); }
      if (n == 4 * J + 4) t1 = xfA(1, stage_next, cf, cv);
      if (n == 4 * J + 6) { wrA(buf ^ 1, 1, t1); SLN_SB; ldA(1, stage_next, cs); }
    });
    __syncthreads();
    mma(S1{}, [&](int n) __attribute__((always_inline)) {
      if ((n & 1) && (n >> 1) < 2 + J) rd1(buf ^ 1, 0, S0{}, n >> 1);
      if (n == 0) plan(kt + 1);
      if (!IDENT && n >= 2 && n <= 8 && !(n & 1)) cfr[(n >> 1) - 1] = coef[sln_cidx(coff_n) + (n >> 1) - 1];
    });
  };
#undef SLN_SB
  for (int kt = 0; kt < ntiles; kt += 2) { body(kt, S1{}); body(kt + 1, S0{}); }

  // ------------------------------- epilogue -------------------------------
  SLN_TRACE(3);
  // accumulator layout of the 16 x 16 x 4 shape: register r of lane l is row 4 (l / 16) + r, column l % 16
  const bool has_add = a.addend != nullptr;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int cl = WN16 * wc + 16 * j + l16;
    const int col = n0 + cl;
    const bool cvalid = col < a.N;
    const int ccl = cvalid ? col : 0;
    const float bias = cvalid ? bias_pre[j] : 0.f;
    float4 ec = make_float4(1.f, 0.f, 0.f, 1.f);
    if (EPI == EPI_MASK) ec = ecoef[cl];
    float yv[8], xpv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int row = min(m0 + 32 * wr + 16 * (q >> 2) + 4 * lg + (q & 3), a.M - 1);
      yv[q] = acc[q >> 2][j][q & 3] + bias;
      xpv[q] = 0.f;
      if (EPI == EPI_MASK) xpv[q] = a.xprev[(size_t)row * a.ldx + a.xcol0 + ccl];
    }
    if (has_add) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int row = min(m0 + 32 * wr + 16 * (q >> 2) + 4 * lg + (q & 3), a.M - 1);
        yv[q] += a.addend[(size_t)row * a.ldadd + a.addcol0 + ccl];
      }
    }
    float s1 = 0.f, s2 = 0.f;
    double d1 = 0.0, d2 = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int row = m0 + 32 * wr + 16 * (q >> 2) + 4 * lg + (q & 3);
      const bool ok = cvalid && row < a.M;
      float y = yv[q];
      if (EPI == EPI_MASK) y = fmaf(ec.x, xpv[q], ec.y) > 0.f ? y : 0.f;
      y = ok ? y : 0.f;
      if (EPI == EPI_STATS) { d1 += (double)y; d2 = fma((double)y, (double)y, d2); }
      if (EPI == EPI_MASK) { s1 += y; s2 = fmaf(y, (xpv[q] - ec.z) * ec.w, s2); }
      if (ok) a.Y[(size_t)row * a.ldy + a.ycol0 + col] = y;
    }
    if (EPI == EPI_STATS) {
      d1 += __shfl_xor(d1, 16, 64); d2 += __shfl_xor(d2, 16, 64);
      d1 += __shfl_xor(d1, 32, 64); d2 += __shfl_xor(d2, 32, 64);
      if (lg == 0) { redd[(wr * BN + cl) * 2 + 0] = d1; redd[(wr * BN + cl) * 2 + 1] = d2; }
    } else if (EPI == EPI_MASK) {
      s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
      s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
      if (lg == 0) { red[(wr * BN + cl) * 2 + 0] = s1; red[(wr * BN + cl) * 2 + 1] = s2; }
    }
  }
  if (EPI != EPI_PLAIN) {
    __syncthreads();
    double* out = (EPI == EPI_STATS) ? a.osums : a.ogsums;
    if (out != nullptr) {
      for (int c = tid; c < BN; c += NT) {
        if (n0 + c < a.N) {
          double s1, s2;
          if (EPI == EPI_STATS) { s1 = redd[c * 2] + redd[(BN + c) * 2]; s2 = redd[c * 2 + 1] + redd[(BN + c) * 2 + 1]; }
          else { s1 = (double)red[c * 2] + (double)red[(BN + c) * 2]; s2 = (double)red[c * 2 + 1] + (double)red[(BN + c) * 2 + 1]; }
          const double qs = EPI == EPI_STATS ? sln_q_fwd(a.M) : SLN_Q_BWD;
          atomicAdd(out + n0 + c, sln_qd(s1, qs));
          atomicAdd(out + a.ocstride + n0 + c, sln_qd(s2, qs));
        }
      }
    }
  }
  SLN_TRACE(4);
}

inline size_t nt16_smem_bytes(int K, int J) {
  const int kpad = (K + 31) & ~31;
  return (size_t)sln_crows(kpad + 4) * 16 + (size_t)2 * (64 + 32 * J) * (BK + 8) * 4 + (size_t)32 * J * 16 + (size_t)2 * 32 * J * 16;
}

// Which tile for a single-segment problem?  Cost model: rounds of 256 workgroups x tile width.  Returns J (3 or 5) when the
// 16 x 16 body with 64 x 32J tiles needs fewer width-rounds than 64 x 64 tiles (J = 2), else 0.
inline int nt16_pick(const GemmNTArgs& a) {
  static const int enabled = std::getenv("SLN_NT16") ? std::atoi(std::getenv("SLN_NT16")) : 1;
  if (!enabled || a.A.nseg != 1 || a.K > 2048) return 0;
  // (round 4, measured and dropped: J = 1 / J = 2 - 64 x 32 / 64 x 64 tiles of 16 x 16 MFMAs, two co-resident workgroups per CU - for
  // the N = 256 dgrad of net1's second Linear: 1.8585 / 1.8688 ms per step against 1.8619-1.8637, same box)
  const long mb = sln_cdiv(a.M, 64);
  const long base = ((mb * sln_cdiv(a.N, 64) + 255) / 256) * 2;
  int best = 0; long bc = base;
  const int cand[2] = {3, 5};
  for (int q = 0; q < 2; ++q) {
    const int Jc = cand[q];
    const long c = ((mb * sln_cdiv(a.N, 32 * Jc) + 255) / 256) * Jc;
    if (c < bc) { bc = c; best = Jc; }
  }
  return best;
}

// ---------------------------------------------------------------------------------------------
// NT kernel, small tile: 32 x 32 outputs per workgroup, the K loop SPLIT over its four wavefronts
// ---------------------------------------------------------------------------------------------
// The object-side GEMMs of a batch of 64 graphs (M = 2048 rows, N = 128 / 256) have 64-128 tiles of 64 x 64 for 256 CUs, and on
// its CU a block walks K tile by tile with one wave per SIMD (~1 650 cycles per tile for 1 024 cycles of MFMA): 8 tiles at
// K = 256 whatever the tile shape.  Here a workgroup owns a 32 x 32 tile (4x the workgroups: every CU gets one), wave w takes the
// k-tiles w, w + 4, ... through its OWN LDS tiles (no workgroup barrier in the loop; the LDS queue of a wave is in order), the
// four partial accumulators meet in LDS once, and every wave finishes 8 of the 32 rows (bias / mask / statistics / store).
// NSEG = 1: single-segment operands.  NSEG = 3 (round 4): the gathered concat [obj[s] | pred | obj[o]] of a GraphTripleConv's first
// Linear with tile-aligned segments - for graphs of a few rows only (the refinement loop's 13-object room, BASELINE config c1):
// there the 64 x 64 body is 4 workgroups walking 12 k-tiles each (14.8 us per launch, the slowest kernel of a refinement
// iteration's decoder); at 64 graphs the 64 x 64 body stays (measured in round 2: 2.54 -> 2.57 ms per step with this one).
template <int AMODE, int EPI, bool HELP = false, int NSEG = 1>   // HELP: 512 threads, wavefronts 4 .. 7 build the coefficient tables (see gemm_nt_body)
__device__ __forceinline__ void gemm_nt_small_body(const GemmNTArgs& a, const int bid, char* smem, const bool small_xcd = true) {
  constexpr bool HAS_X2 = AMODE == 1;
  constexpr bool IDENT = AMODE == 2;
  constexpr int LDT = BK + 4;
  constexpr int TS = 32;                                   // tile edge
  const int kpad = (a.K + 31) & ~31;
  float4* coef = reinterpret_cast<float4*>(smem);           // [kpad]
  float4* ecoef = coef + sln_crows(kpad);                   // [TS]
  double* sred = reinterpret_cast<double*>(ecoef + TS);     // [4][TS][2] column statistics of the four waves
  float* wl = reinterpret_cast<float*>(sred + 4 * TS * 2);  // per wave: A [TS][LDT] | B [TS][LDT]; later its 16 x 64 partial accumulator
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (HELP && tid >= 256) {                    // first, see gemm_nt_body
    int n0 = 0;
    if (EPI == EPI_MASK) {
      const int tiles_n = (a.N + TS - 1) / TS;
      n0 = ((small_xcd ? xcd_remap(bid, ((a.M + TS - 1) / TS) * tiles_n) : bid) % tiles_n) * TS;
    }
    nt_helper_tables<NSEG, IDENT, EPI, TS>(a, coef, ecoef, tid - 256, 256, kpad, n0);
    __syncthreads();
    return;
  }
  const int tiles_n = (a.N + TS - 1) / TS;
  // workgroups that share A rows (same m tile, all n tiles) share an XCD and its L2: in plain order the eight XCDs each fetched
  // every A row (profiles/r03: 34 MB per launch for 7 MB of operands)
  const int lb = small_xcd ? xcd_remap(bid, ((a.M + TS - 1) / TS) * tiles_n) : bid;      // = gridDim.x (launch_nt_small), without the dispatch packet read
  const int m0 = (lb / tiles_n) * TS, n0 = (lb % tiles_n) * TS;
  float* As = wl + wave * 2 * TS * LDT;
  float* Bs = As + TS * LDT;

  const int kq = lane & 7, r0 = lane >> 3;                  // float4 column, first of this lane's 4 rows (stride 8)
  // segment of a k-tile (NSEG = 3: every segment is a whole number of tiles, checked by nt_wants_small): first column of the tile
  // inside its segment, and the segment itself - wave-uniform
  const int e0 = NSEG > 1 ? a.A.seg[0].len : 0, e1 = NSEG > 1 ? e0 + a.A.seg[1].len : 0;
  // NSEG = 3: the segments' fields preloaded into scalar registers and chosen by masks (pick_pre) - `a.A.seg[si]` with a run-time
  // si made hipcc keep a 64-byte scratch copy of the records and read the fields back from it per tile
  SegSel pre0 = sln_seg_preload1(a.A.seg[0], 0), pre1 = pre0, pre2 = pre0;
  if (NSEG > 1) { pre1 = sln_seg_preload1(a.A.seg[1], e0); pre2 = sln_seg_preload1(a.A.seg[2], e1); }
  int rid[NSEG][4];
#pragma unroll
  for (int sgi = 0; sgi < NSEG; ++sgi) {
    const int which = a.A.seg[sgi].which;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int row = min(m0 + r0 + 8 * p, a.M - 1);
      rid[sgi][p] = which == 0 ? row : (which == 1 ? a.A.idx_a[row] : a.A.idx_b[row]);
    }
  }
  const int ntiles = kpad / BK;
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 ga1[4], ga2[4], gb[4];
  auto gload = [&](int kt) {                               // kt clamped by the caller; everything unconditional
    const int k0 = kt * BK;
    const SegSel sg = NSEG > 1 ? pick_pre(pre0, pre1, pre2, NSEG, e0, e1, k0) : pre0;
    const int cs = min(k0 + 4 * kq, sg.end - 4) - sg.base;
    const float* x2p = sg.x2 ? sg.x2 : sg.x1;
    const int ld2 = sg.x2 ? sg.ld2 : sg.ld1, c2 = sg.x2 ? sg.c2 : sg.c1;
    const int ms1 = -(int)(NSEG > 1 && k0 >= e0 && k0 < e1), ms2 = -(int)(NSEG > 1 && k0 >= e1);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int rr = NSEG > 1 ? ((rid[0][p] & ~(ms1 | ms2)) | (rid[NSEG > 1 ? 1 : 0][p] & ms1) | (rid[NSEG - 1][p] & ms2)) : rid[0][p];
      ga1[p] = ld4(sg.x1 + (size_t)rr * sg.ld1 + sg.c1 + cs);
      if (HAS_X2) ga2[p] = ld4(x2p + (size_t)rr * ld2 + c2 + cs);
    }
    const int cw = min(k0 + 4 * kq, a.K - 4);
#pragma unroll
    for (int p = 0; p < 4; ++p) gb[p] = ld4(a.W + (size_t)min(n0 + r0 + 8 * p, a.N - 1) * a.ldw + cw);
  };
  auto lstore = [&](int kt) {
    const int col = kt * BK + 4 * kq;
    const int k0 = kt * BK;
    const SegSel sg = NSEG > 1 ? pick_pre(pre0, pre1, pre2, NSEG, e0, e1, k0) : pre0;
    const bool cv = NSEG > 1 ? col < a.K : col < sg.end, x2v = HAS_X2 && sg.x2 != nullptr, kv = col < a.K;
    const float4* cf = coef + sln_cidx(min(col, kpad - 4));
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int rl = r0 + 8 * p;
      const bool v = cv && (m0 + rl) < a.M;
      float4 t = IDENT ? ga1[p] : xform(ga1[p], x2v ? ga2[p] : z4, cf);
      t.x = v ? t.x : 0.f; t.y = v ? t.y : 0.f; t.z = v ? t.z : 0.f; t.w = v ? t.w : 0.f;
      *reinterpret_cast<float4*>(As + rl * LDT + 4 * kq) = t;
      const bool vb = kv && (n0 + rl) < a.N;
      float4 u = gb[p];
      u.x = vb ? u.x : 0.f; u.y = vb ? u.y : 0.f; u.z = vb ? u.z : 0.f; u.w = vb ? u.w : 0.f;
      *reinterpret_cast<float4*>(Bs + rl * LDT + 4 * kq) = u;
    }
  };

  if (wave < ntiles) gload(wave);                           // first tile of this wave, issued before the coefficient set-up
  if (!HELP) {
    if (!IDENT) {
      sln_fill_coefs<NSEG>(a.A, coef, tid, 256);
      for (int c = a.K + tid; c < kpad; c += 256) coef[sln_cidx(c)] = z4;
    }
    if (EPI == EPI_MASK) {
      for (int c = tid; c < TS; c += 256) {
        float4 e = make_float4(1.f, 0.f, 0.f, 1.f);
        if (n0 + c < a.N) e = bn_fwd_coef4(a.obn, n0 + c);
        ecoef[c] = e;
      }
    }
  }
  __syncthreads();                                          // coefficient tables visible

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int lrow = lane & 31, lk = lane >> 5;
  // the epilogue's memory reads (bias, the mask's pre-activations, the addend: 4 rows per lane) are requested in front of the loop
  const int pcol = min(n0 + lrow, a.N - 1);
  const float bias_pre = a.bias ? a.bias[pcol] : 0.f;
  float xp_pre[4], add_pre[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = min(m0 + 8 * wave + q + 4 * lk, a.M - 1);
    xp_pre[q] = 0.f; add_pre[q] = 0.f;
    if (EPI == EPI_MASK) xp_pre[q] = a.xprev[(size_t)row * a.ldx + a.xcol0 + pcol];
    if (a.addend) add_pre[q] = a.addend[(size_t)row * a.ldadd + a.addcol0 + pcol];
  }
  for (int kt = wave; kt < ntiles; kt += 4) {
    lstore(kt);
    if (kt + 4 < ntiles) gload(kt + 4);                     // wave-uniform branch; the refill flies under this tile's MFMAs
    __builtin_amdgcn_wave_barrier();
    // all eight fragment reads of the tile first, then its sixteen MFMAs: read -> wait -> four MFMAs per 8-wide chunk (what hipcc
    // made of the plain loop: six s_waitcnt lgkmcnt(0) per tile) exposed the LDS latency four times per tile on the one chain a
    // wave has
    float4 fa[BK / 8], fb[BK / 8];
#pragma unroll
    for (int q = 0; q < BK / 8; ++q) {
      fa[q] = *reinterpret_cast<const float4*>(As + lrow * LDT + 4 * lk + 8 * q);
      fb[q] = *reinterpret_cast<const float4*>(Bs + lrow * LDT + 4 * lk + 8 * q);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < BK / 8; ++q) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q].x, fb[q].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q].y, fb[q].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q].z, fb[q].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q].w, fb[q].w, acc, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  }

  // ---- the four partial tiles meet: wave w keeps registers 4w .. 4w+3 (rows 8w + (r & 3) + 4 lk) ----
  float* part = wl + wave * 2 * TS * LDT;                   // this wave's own LDS region: 16 x 64 floats fit into its two tiles
#pragma unroll
  for (int r = 0; r < 16; ++r) part[r * 64 + lane] = acc[r];
  __syncthreads();
  float yv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) v += wl[w * 2 * TS * LDT + (4 * wave + q) * 64 + lane];
    yv[q] = v;
  }
  const int col = n0 + lrow;
  const bool cvalid = col < a.N;
  const int ccl = cvalid ? col : 0;
  const float bias = cvalid ? bias_pre : 0.f;
  float4 ec = make_float4(1.f, 0.f, 0.f, 1.f);
  if (EPI == EPI_MASK) ec = ecoef[lrow];
  float xpv[4], addv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { xpv[q] = xp_pre[q]; addv[q] = add_pre[q]; }
  (void)ccl;
  float s1 = 0.f, s2 = 0.f;
  double d1 = 0.0, d2 = 0.0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = m0 + 8 * wave + q + 4 * lk;
    const bool ok = cvalid && row < a.M;
    float y = yv[q] + bias + addv[q];
    if (EPI == EPI_MASK) y = fmaf(ec.x, xpv[q], ec.y) > 0.f ? y : 0.f;
    y = ok ? y : 0.f;
    if (EPI == EPI_STATS) { d1 += (double)y; d2 = fma((double)y, (double)y, d2); }
    if (EPI == EPI_MASK) { s1 += y; s2 = fmaf(y, (xpv[q] - ec.z) * ec.w, s2); }
    if (ok) a.Y[(size_t)row * a.ldy + a.ycol0 + col] = y;
  }
  if (EPI != EPI_PLAIN) {
    if (EPI == EPI_MASK) { d1 = (double)wave_sum_halves(s1); d2 = (double)wave_sum_halves(s2); }
    else { d1 += __shfl_xor(d1, 32, 64); d2 += __shfl_xor(d2, 32, 64); }
    if (lk == 0) { sred[(wave * TS + lrow) * 2] = d1; sred[(wave * TS + lrow) * 2 + 1] = d2; }
    __syncthreads();
    double* out = (EPI == EPI_STATS) ? a.osums : a.ogsums;
    if (out != nullptr && tid < TS && n0 + tid < a.N) {
      double t1 = 0.0, t2 = 0.0;
#pragma unroll
      for (int w = 0; w < 4; ++w) { t1 += sred[(w * TS + tid) * 2]; t2 += sred[(w * TS + tid) * 2 + 1]; }
      const double qs = EPI == EPI_STATS ? sln_q_fwd(a.M) : SLN_Q_BWD;
      atomicAdd(out + n0 + tid, sln_qd(t1, qs));
      atomicAdd(out + a.ocstride + n0 + tid, sln_qd(t2, qs));
    }
  }
}

inline size_t nt_small_smem_bytes(int K) {
  const int kpad = (K + 31) & ~31;
  return (size_t)sln_crows(kpad) * 16 + 32 * 16 + 4 * 32 * 2 * 8 + (size_t)4 * 2 * 32 * (BK + 4) * 4;
}

// under-filled single-segment problems: fewer than this many 64 x 64 tiles (the object-side GEMMs of a 64-graph batch, the heads)
inline bool nt_wants_small(const GemmNTArgs& a) {
  static const int max_tiles = std::getenv("SLN_NT_SMALL_TILES") ? std::atoi(std::getenv("SLN_NT_SMALL_TILES")) : 160;
  return a.A.nseg == 1 && (long)sln_cdiv(a.M, 64) * sln_cdiv(a.N, 64) <= max_tiles && a.K <= 2048;
}
// ... and the three-segment gathered operand of graphs of a few rows (see gemm_nt_small_body, NSEG = 3): at most 64 rows, every
// segment a whole number of k-tiles, the segments filling K exactly
inline bool nt_wants_small3(const GemmNTArgs& a) {
  static const int max_rows = std::getenv("SLN_NT_SMALL3_ROWS") ? std::atoi(std::getenv("SLN_NT_SMALL3_ROWS")) : 64;
  if (a.A.nseg != 3 || a.M > max_rows || a.K > 2048) return false;
  int tot = 0;
  for (int s = 0; s < 3; ++s) { if (a.A.seg[s].len % BK != 0 || a.A.seg[s].len < BK) return false; tot += a.A.seg[s].len; }
  return tot == a.K;
}

// ---------------------------------------------------------------------------------------------
// TN kernel (wgrad)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float tn_acc_read(const float& v) {   // see the epilogue of gemm_tn_body
  float r;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r) : "a"(v));
  return r;
}

struct ColSel { const float* x1; const float* x2; int ld1, ld2, which; bool valid; };

__device__ __forceinline__ ColSel pick_col(const Operand& op, int col) {
  // per-thread (non-uniform) choice of the segment holding logical column `col`
  ColSel r;
  r.valid = col < op.cols;
  col = r.valid ? col : 0;                  // keep the address inside the operand; invalid lanes are masked later
  const int e0 = op.seg[0].len, e1 = e0 + op.seg[1].len;
  const int s = (op.nseg > 1 && col >= e0) ? ((op.nseg > 2 && col >= e1) ? 2 : 1) : 0;
  const int base = s == 0 ? 0 : (s == 1 ? e0 : e1);
  const int c = col - base;
  const Seg& g0 = op.seg[0]; const Seg& g1 = op.seg[1]; const Seg& g2 = op.seg[2];
  r.x1 = (s == 0 ? g0.x1 + g0.c1 : (s == 1 ? g1.x1 + g1.c1 : g2.x1 + g2.c1)) + c;
  const float* b2 = s == 0 ? g0.x2 : (s == 1 ? g1.x2 : g2.x2);
  r.x2 = b2 ? b2 + (s == 0 ? g0.c2 : (s == 1 ? g1.c2 : g2.c2)) + c : nullptr;
  r.ld1 = s == 0 ? g0.ld1 : (s == 1 ? g1.ld1 : g2.ld1);
  r.ld2 = s == 0 ? g0.ld2 : (s == 1 ? g1.ld2 : g2.ld2);
  r.which = s == 0 ? g0.which : (s == 1 ? g1.which : g2.which);
  return r;
}

__device__ __forceinline__ float4 coef_for_col(const Operand& op, int col) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < op.cols) {
    const int e0 = op.seg[0].len, e1 = e0 + op.seg[1].len;
    if (op.nseg > 2 && col >= e1) v = sln_coef_for(op.seg[2], col - e1);
    else if (op.nseg > 1 && col >= e0) v = sln_coef_for(op.seg[1], col - e0);
    else v = sln_coef_for(op.seg[0], col);
  }
  return v;
}

// XG: the X operand has row-gathered segments (GraphTripleConv input).  Its row indices then run through their own
// three-stage register pipeline, one tile ahead of the data loads that use them, so that no load waits on another.
// G (a gradient) is never gathered.
//
// Round 3 layout of the compute half.  The tiles stay ROW-major in LDS, as the rows arrive ([BK rows][64 columns + pad]: one
// ds_write_b128 per staged float4, no transposing store).  The four wavefronts of a block no longer own a 32 x 32 quarter of the
// 64 x 64 output tile each (every operand value fetched with its own ds_read_b32: 32 per wave and tile).  Wave (wr, wc) takes
// the row half wr of a tile (16 rows = 8 MFMA k-steps) and the column half wc of X, and accumulates a 64 x 32 piece of dW in two
// 32 x 32 accumulators: a lane reads two ADJACENT G columns with one ds_read_b64 - they become output rows 2i and 2i+1, any fixed
// permutation of dW's rows is as good as another - and one X value (columns stay lane-contiguous for the atomics); hipcc pairs
// the reads of two k-steps (ds_read2_b64 / ds_read2_b32): 8 LDS instructions per wave and tile instead of 32.  The two row
// halves meet in LDS once, behind the loop (each wave hands over one accumulator and finishes the other: 16 KB through the tile
// buffers), so a block issues the same 64 x 64 atomics as before.  (Four accumulators per wave - the whole 64 x 64 tile, rows
// split four ways - need 6 LDS instructions per tile but 64 AGPRs: 181-196 registers, two workgroups per CU instead of three.)
// The per-column coefficients of the two operands are loop invariants of a thread (its four columns never change): they sit in
// registers instead of being read from LDS for every staged float4 (16 ds_read_b128 per tile and thread).
template <int BM, int BN, int WM, int WN, bool G_X2, bool XG>
__device__ __forceinline__ void gemm_tn_body(const GemmTNArgs& a, const int bx, const int by, char* smem) {
  static_assert(BM == 64 && BN == 64 && WM * WN == 4, "64 x 64 tile, 4 waves");
  // row stride = 32 (mod 64) floats: the two rows a wavefront reads per fragment (lanes 0-31 / 32-63) fall into disjoint bank halves
  constexpr int SA = BM + TN_PAD, SB = BN + TN_PAD;
  constexpr int TPRA = BM / 4, TPRB = BN / 4;         // threads per row
  constexpr int RPA = 256 / TPRA, RPB = 256 / TPRB;   // rows per pass
  constexpr int PA = BK / RPA, PB = BK / RPB;
  float4* coefG = reinterpret_cast<float4*>(smem);          // [BM]
  float4* coefX = coefG + BM;                               // [BN]
  float* As = reinterpret_cast<float*>(coefX + BN);         // [2][BK][SA]
  float* Bs = As + 2 * BK * SA;                             // [2][BK][SB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_k = (a.Kin + BN - 1) / BN;
  const int n0 = (bx / tiles_k) * BM, k0 = (bx % tiles_k) * BN;
  const int rbeg = by * a.rows_per_block;
  const int rend = min(a.R, rbeg + a.rows_per_block);

  // per-column coefficient tables for this block's column ranges.  The two-source path re-reads G's coefficients from LDS for every
  // staged tile; as an array of float4 per column a lane's four ds_read_b128 (columns 4 i .. 4 i + 3 of lane i % 16: a 64-byte
  // stride) ran 4-way bank-conflicted - 64 LDS cycles per wave and tile, half of the kernel's LDS time with three workgroups per
  // CU.  That table is PLANAR here ([kind][column]): the four coefficients of a kind for a lane's four columns are one aligned
  // float4, 16 lanes read 256 contiguous bytes.
  float* coefGp = reinterpret_cast<float*>(coefG);            // G_X2: [4 kinds][BM]
  // wavefront 0 builds G's table, wavefront 1 X's: one memory round trip instead of two in a row (64 + 64 columns)
  static_assert(BM == 64 && BN == 64, "one wavefront per table");
  if (tid < BM) {
    const int c = tid;
    const float4 v = coef_for_col(a.G, n0 + c);
    if (G_X2) { coefGp[c] = v.x; coefGp[BM + c] = v.y; coefGp[2 * BM + c] = v.z; coefGp[3 * BM + c] = v.w; }
    else coefG[c] = v;
  } else if (tid < BM + BN) {
    const int c = tid - BM;
    coefX[c] = coef_for_col(a.X, k0 + c);
  }

  const int ca = 4 * (tid % TPRA), ra0 = tid / TPRA;
  const int cb = 4 * (tid % TPRB), rb0 = tid / TPRB;
  const ColSel gs = pick_col(a.G, n0 + ca);
  const ColSel xs = pick_col(a.X, k0 + cb);

  constexpr int NST = 2;      // register stages, see gemm_nt_body
  float4 g1[NST][PA], g2[NST][PA], x1[NST][PB];
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // unconditional, clamped loads (see the note in gemm_nt_body); masking happens in lstore()
  const int* x_ip = xs.which == 2 ? a.X.idx_b : a.X.idx_a;
  if (x_ip == nullptr) x_ip = a.X.idx_a ? a.X.idx_a : a.X.idx_b;   // plain-row columns still issue the (unused) index loads
  const float* g_x2 = gs.x2 ? gs.x2 : gs.x1;
  const int g_ld2 = gs.x2 ? gs.ld2 : gs.ld1;
  int xi[NST][PB];
  auto iload = [&](int rt, auto stage) {
    constexpr int S = decltype(stage)::value;
#pragma unroll
    for (int p = 0; p < PB; ++p) xi[S][p] = ldi(x_ip + min(rbeg + rt * BK + rb0 + RPB * p, rend - 1));
  };
  auto gload = [&](int rt, int rt_idx, auto stage) {
    constexpr int S = decltype(stage)::value;
    if (SLN_TN_ABL & 1) return;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const int row = min(rbeg + rt * BK + ra0 + RPA * p, rend - 1);
      g1[S][p] = ld4(gs.x1 + (size_t)row * gs.ld1);
      if (G_X2) g2[S][p] = ld4(g_x2 + (size_t)row * g_ld2);
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      const int row = min(rbeg + rt * BK + rb0 + RPB * p, rend - 1);
      const int r = XG ? (xs.which ? xi[S][p] : row) : row;
      x1[S][p] = ld4(xs.x1 + (size_t)r * xs.ld1);
    }
    if (XG) iload(rt_idx, stage);          // indices of the tile this stage will load next
  };
  float4 dbacc = z4;
  float4 cG[4], cX[4];                     // this thread's coefficients (filled behind the table barrier)
  // rt is NOT clamped here: the rows of a surplus tile (rt > last, the loop runs whole pairs of tiles) lie behind rend and are
  // stored as zeros
  auto lstore = [&](int rt, int buf, auto stage) {
    constexpr int S = decltype(stage)::value;
    if (SLN_TN_ABL & 2) return;
#pragma unroll
    for (int p = 0; p < PA; ++p) {
      const int rl = ra0 + RPA * p;
      // Masks cost matrix-pipe time (8 clocks per v_cndmask, tools/lab/overlap.hip), so only the one that is needed stays: G rows
      // behind the chunk's end are zeroed - a zero row of G adds nothing to dW or db whatever the (clamped, finite) row of X holds.
      // Columns behind the operand's width come out as exact zeros by themselves (their coefficient rows are zero), and a segment
      // without a second source has c.y == 0 and reads x1 twice.
      const bool v = (rbeg + rt * BK + rl) < rend;
      // two-source gradients (BatchNorm backward) keep their coefficients in LDS: with them in registers the dgrad + wgrad kernels
      // need 172-180 registers and lose the third workgroup per CU
      float4 t = G_X2 ? xform_planar(g1[S][p], g2[S][p], *reinterpret_cast<const float4*>(coefGp + ca),
                                     *reinterpret_cast<const float4*>(coefGp + BM + ca), *reinterpret_cast<const float4*>(coefGp + 2 * BM + ca),
                                     *reinterpret_cast<const float4*>(coefGp + 3 * BM + ca))
                      : xform1(g1[S][p], cG);
      t.x = v ? t.x : 0.f; t.y = v ? t.y : 0.f; t.z = v ? t.z : 0.f; t.w = v ? t.w : 0.f;
      dbacc.x += t.x; dbacc.y += t.y; dbacc.z += t.z; dbacc.w += t.w;
      *reinterpret_cast<float4*>(As + buf * BK * SA + rl * SA + ca) = t;
    }
#pragma unroll
    for (int p = 0; p < PB; ++p) {
      const int rl = rb0 + RPB * p;
      *reinterpret_cast<float4*>(Bs + buf * BK * SB + rl * SB + cb) = xform1(x1[S][p], cX);
    }
  };

  f32x16 acc[2];                           // [ja]: output rows n0 + 2 i + ja, columns k0 + 32 wc + lane % 32
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  const int ntiles = (rend - rbeg + BK - 1) / BK;
  using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
  const int last = ntiles - 1;          // ntiles >= 1: every launched block owns at least one row
  if (XG) { iload(0, S0{}); iload(min(1, last), S1{}); }
  gload(0, min(2, last), S0{});
  gload(min(1, last), min(3, last), S1{});
  __syncthreads();            // coefficient tables visible
#pragma unroll
  for (int j = 0; j < 4; ++j) { if (!G_X2) cG[j] = coefG[ca + j]; cX[j] = coefX[cb + j]; }
  lstore(0, 0, S0{});
  gload(min(2, last), min(4, last), S0{});
  __syncthreads();
  const int lrow = lane & 31, lk = lane >> 5;
  // fragments of two k-steps (4 rows of the tile) ping-pong between two register sets; set 0 is refilled with the next tile's
  // first quarter right after the barrier, under the MFMAs of the current tile's last quarter (as in gemm_nt_body)
  const int wr = wave >> 1, wc = wave & 1;
  float2 fa[2][2]; float fb[2][2];
  auto rd = [&](int buf, int quarter, auto set) __attribute__((always_inline)) {
    constexpr int F = decltype(set)::value;
    const float* as = As + buf * BK * SA + (16 * wr + 4 * quarter + lk) * SA + 2 * lrow;
    const float* bs = Bs + buf * BK * SB + (16 * wr + 4 * quarter + lk) * SB + 32 * wc + lrow;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      fa[F][q] = *reinterpret_cast<const float2*>(as + 2 * q * SA);
      fb[F][q] = bs[2 * q * SB];
    }
  };
  auto mma = [&](auto set) {
    constexpr int F = decltype(set)::value;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[F][q].x, fb[F][q], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[F][q].y, fb[F][q], acc[1], 0, 0, 0);
    }
  };
  rd(0, 0, S0{});
  // same schedule as gemm_nt_body: stage the next tile and refill its register stage first, one straight loop over pairs
  auto body = [&](int rt, auto stage_next) {
    const int buf = rt & 1;
    lstore(rt + 1, buf ^ 1, stage_next);
    gload(min(rt + 3, last), min(rt + 5, last), stage_next);
    __builtin_amdgcn_sched_barrier(0);
    if (!(SLN_TN_ABL & 8)) rd(buf, 1, S1{});
    mma(S0{});
    if (!(SLN_TN_ABL & 8)) rd(buf, 2, S0{});
    mma(S1{});
    if (!(SLN_TN_ABL & 8)) rd(buf, 3, S1{});
    mma(S0{});
    if (!(SLN_TN_ABL & 4)) __syncthreads();
    if (!(SLN_TN_ABL & 8)) rd(buf ^ 1, 0, S0{});
    mma(S1{});
  };
#if SLN_TN_SCHED
  // The same loop with its instruction order written out (as gemm_nt_body's): hipcc issues the tile's four to six global loads and
  // four LDS writes back to back at the top of the body, and a second memory instruction behind the first stalls the wave at the
  // issue stage (64 / 52 clocks, tools/lab/overlap.hip) with all its MFMAs queued behind - without the loads the stand-alone
  // 32 k-row wgrad runs at 100 TF instead of 85 (SLN_TN_ABL).  Here every store / reload pair of a staging pass sits behind its own
  // MFMA: G passes behind MFMAs 0 .. 3 (the second source one slot later), X passes behind 4 and 5, the row indices of a gathered
  // X behind 6 and 7; fragment reads a quarter ahead as before.  Same arithmetic, same sums.
  {
#define SLN_SB __builtin_amdgcn_sched_barrier(0)
    auto pieceG = [&](int rt, int buf, int p, auto stage) __attribute__((always_inline)) {       // store pass p of tile rt + 1, reload x1 of tile rt + 3
      constexpr int S = decltype(stage)::value;
      const int rl = ra0 + RPA * p;
      const bool v = (rbeg + (rt + 1) * BK + rl) < rend;
      float4 t = G_X2 ? xform_planar(g1[S][p], g2[S][p], *reinterpret_cast<const float4*>(coefGp + ca),
                                     *reinterpret_cast<const float4*>(coefGp + BM + ca), *reinterpret_cast<const float4*>(coefGp + 2 * BM + ca),
                                     *reinterpret_cast<const float4*>(coefGp + 3 * BM + ca))
                      : xform1(g1[S][p], cG);
      t.x = v ? t.x : 0.f; t.y = v ? t.y : 0.f; t.z = v ? t.z : 0.f; t.w = v ? t.w : 0.f;
      dbacc.x += t.x; dbacc.y += t.y; dbacc.z += t.z; dbacc.w += t.w;
      *reinterpret_cast<float4*>(As + (buf ^ 1) * BK * SA + rl * SA + ca) = t;
      SLN_SB;
      const int row = min(rbeg + min(rt + 3, last) * BK + ra0 + RPA * p, rend - 1);
      g1[S][p] = ld4(gs.x1 + (size_t)row * gs.ld1);
    };
    auto pieceG2 = [&](int rt, int p, auto stage) __attribute__((always_inline)) {
      constexpr int S = decltype(stage)::value;
      const int row = min(rbeg + min(rt + 3, last) * BK + ra0 + RPA * p, rend - 1);
      if (G_X2) g2[S][p] = ld4(g_x2 + (size_t)row * g_ld2);
    };
    auto pieceX = [&](int rt, int buf, int p, auto stage) __attribute__((always_inline)) {
      constexpr int S = decltype(stage)::value;
      *reinterpret_cast<float4*>(Bs + (buf ^ 1) * BK * SB + (rb0 + RPB * p) * SB + cb) = xform1(x1[S][p], cX);
      SLN_SB;
      const int row = min(rbeg + min(rt + 3, last) * BK + rb0 + RPB * p, rend - 1);
      const int r = XG ? (xs.which ? xi[S][p] : row) : row;
      x1[S][p] = ld4(xs.x1 + (size_t)r * xs.ld1);
    };
    auto pieceI = [&](int rt, int p, auto stage) __attribute__((always_inline)) {               // indices of the tile this stage loads next
      constexpr int S = decltype(stage)::value;
      if (XG) xi[S][p] = ldi(x_ip + min(rbeg + min(rt + 5, last) * BK + rb0 + RPB * p, rend - 1));
    };
    // the four MFMAs of a quarter (two k-steps x two accumulators), hook(n) behind the n-th
    auto mma4 = [&](auto set, auto&& hook) __attribute__((always_inline)) {
      constexpr int F = decltype(set)::value;
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[F][0].x, fb[F][0], acc[0], 0, 0, 0); SLN_SB; hook(0); SLN_SB;
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[F][0].y, fb[F][0], acc[1], 0, 0, 0); SLN_SB; hook(1); SLN_SB;
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[F][1].x, fb[F][1], acc[0], 0, 0, 0); SLN_SB; hook(2); SLN_SB;
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[F][1].y, fb[F][1], acc[1], 0, 0, 0); SLN_SB; hook(3); SLN_SB;
    };
    static_assert(PA == 2 && PB == 2, "two staging passes per operand and tile");
    auto sbody = [&](int rt, auto stage_next) __attribute__((always_inline)) {
      const int buf = rt & 1;
      rd(buf, 1, S1{});
      SLN_SB;
      mma4(S0{}, [&](int n) __attribute__((always_inline)) {
        if (n == 0) pieceG(rt, buf, 0, stage_next);
        if (n == 1) pieceG2(rt, 0, stage_next);
        if (n == 2) pieceG(rt, buf, 1, stage_next);
        if (n == 3) { pieceG2(rt, 1, stage_next); rd(buf, 2, S0{}); }
      });
      mma4(S1{}, [&](int n) __attribute__((always_inline)) {
        if (n == 0) pieceX(rt, buf, 0, stage_next);
        if (n == 1) pieceX(rt, buf, 1, stage_next);
        if (n == 2) pieceI(rt, 0, stage_next);
        if (n == 3) { pieceI(rt, 1, stage_next); rd(buf, 3, S1{}); }
      });
      mma4(S0{}, [&](int) __attribute__((always_inline)) {});
      __syncthreads();
      rd(buf ^ 1, 0, S0{});
      SLN_SB;
      mma4(S1{}, [&](int) __attribute__((always_inline)) {});
    };
#undef SLN_SB
    for (int rt = 0; rt < ntiles; rt += 2) { sbody(rt, S1{}); sbody(rt + 1, S0{}); }
  }
#else
  for (int rt = 0; rt < ntiles; rt += 2) { body(rt, S1{}); body(rt + 1, S0{}); }
#endif

  // ---- the two row halves meet: wave (wr, wc) finishes accumulator ja = wr of column half wc ----
  // Accumulators are read out of the AGPRs at their use (v_accvgpr_read through an "a" constraint): left alone hipcc copies all
  // of them to VGPRs in front of the epilogue and that copy sets the kernel's register count.
  __syncthreads();                          // every wave is done with the tile buffers
  float* red = As;                          // [wave 4][16][64] floats = 16 KB of the tile buffers
  {
    float* dst = red + (wave * 16) * 64 + lane;
    if (wr == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[r * 64] = tn_acc_read(acc[1][r]);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[r * 64] = tn_acc_read(acc[0][r]);
    }
  }
  __syncthreads();
  {
    const float* src = red + ((wave ^ 2) * 16) * 64 + lane;      // the partner's copy of the accumulator this wave keeps
    const float sc = a.sgd_step != nullptr ? -a.sgd_step[0] : 1.f;   // GemmTNArgs::sgd_step (x 1 is exact)
    const int k = k0 + 32 * wc + lrow;
    float* dcol = a.dW + min(k, a.Kin - 1);
    const bool kv = k < a.Kin;
    // sum order (row half 0) + (row half 1) whichever wave finishes
    if (wr == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = (tn_acc_read(acc[0][r]) + src[r * 64]) * sc;
        const int n = n0 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * lk);
        if (kv && n < a.Nout) sln_gatomic_add(dcol + (size_t)n * a.lddw, v);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = (src[r * 64] + tn_acc_read(acc[1][r])) * sc;
        const int n = n0 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * lk) + 1;
        if (kv && n < a.Nout) sln_gatomic_add(dcol + (size_t)n * a.lddw, v);
      }
    }
  }

  if (a.db != nullptr && (bx % tiles_k) == 0) {
    // threads with equal (tid % TPRA) hold partial sums of the same 4 columns
#pragma unroll
    for (int off = TPRA; off < 64; off <<= 1) {
      dbacc.x += __shfl_xor(dbacc.x, off, 64); dbacc.y += __shfl_xor(dbacc.y, off, 64);
      dbacc.z += __shfl_xor(dbacc.z, off, 64); dbacc.w += __shfl_xor(dbacc.w, off, 64);
    }
    // the four wavefronts' sums meet in LDS in a fixed order: ONE add per element and block (four atomics - one per wavefront, in
    // arrival order - made the bias gradient the last order-dependent sum of a single-chunk wgrad)
    __syncthreads();                          // the accumulator hand-over above is done with `red`
    float4* dbs = reinterpret_cast<float4*>(red);            // [4 waves][TPRA]
    if (lane < TPRA) dbs[wave * TPRA + lane] = dbacc;
    __syncthreads();
    if (tid < TPRA) {
      const float4 p0 = dbs[tid], p1 = dbs[TPRA + tid], p2 = dbs[2 * TPRA + tid], p3 = dbs[3 * TPRA + tid];
      const int n = n0 + 4 * tid;
      const float sb = a.sgd_step != nullptr ? -a.sgd_step[0] : 1.f;
      if (n + 0 < a.Nout) sln_gatomic_add(a.db + n + 0, (((p0.x + p1.x) + p2.x) + p3.x) * sb);
      if (n + 1 < a.Nout) sln_gatomic_add(a.db + n + 1, (((p0.y + p1.y) + p2.y) + p3.y) * sb);
      if (n + 2 < a.Nout) sln_gatomic_add(a.db + n + 2, (((p0.z + p1.z) + p2.z) + p3.z) * sb);
      if (n + 3 < a.Nout) sln_gatomic_add(a.db + n + 3, (((p0.w + p1.w) + p2.w) + p3.w) * sb);
    }
  }
}


inline bool tn_gathers(const GemmTNArgs& a) {
  bool g = false;
  for (int s = 0; s < a.X.nseg; ++s) g |= a.X.seg[s].which != 0;
  return g;
}

inline bool tn_supported(const GemmTNArgs& a) {     // the gradient operand is addressed by plain rows
  for (int s = 0; s < a.G.nseg; ++s) if (a.G.seg[s].which != 0) return false;
  return true;
}

inline size_t nt_smem_bytes(int K, int BM, int BN, int WM) {
  const int kpad = (K + 31) & ~31;
  return (size_t)sln_crows(kpad + 4) * 16 + (size_t)2 * (BM + BN) * (BK + 4) * 4 + (size_t)BN * 16 + (size_t)WM * BN * 16;
}

inline size_t tn_smem_bytes(int BM, int BN) { return (size_t)(BM + BN) * 16 + (size_t)2 * BK * (BM + TN_PAD + BN + TN_PAD) * 4; }

inline int nt_heuristic_tile(const GemmNTArgs& a) {
  // Round 4: always the 64 x 64 tile.  Rounds 1-3 gave launches with >= 512 tiles of 128 x 128 (>= 384 of 128 x 64) the bigger body
  // - a rule from before the 64 x 64 K loop was scheduled by hand.  The big bodies need 270-380 registers (one
  // wavefront per SIMD: nothing runs under a workgroup's prologue and epilogue, which at K = 256-640 are as long as its K loop)
  // and their loop is hipcc's own order; three 64 x 64 workgroups share a CU.  Same box, ms per step at 128 / 256 / 512 / 1 024 /
  // 4 096 graphs: 3.01 / 5.16 / 9.48 / 18.55 / 72.6 with the old rule, 2.84 / 4.77 / 8.57 / 16.36 / 64.9 with this one
  // (tools/lab/nt_by_shape.py: the four big Linears 0.50-0.54 -> 0.59-0.64 of the MFMA peak).  SLN_NT_TILE = 1 / 2 or the `tile`
  // argument of sln_linear_forward still select the bigger bodies.
  static const int forced = std::getenv("SLN_NT_TILE") ? std::atoi(std::getenv("SLN_NT_TILE")) : -1;      // lab: 0 = 64 x 64, 1 = 128 x 64, 2 = 128 x 128
  (void)a;
  return forced >= 0 ? forced : 0;
}

// launches big enough to fill the chip on their own: the paired (dgrad + wgrad) and grouped launches are for the small ones
inline bool nt_big_shape(const GemmNTArgs& a) {
  return (long)sln_cdiv(a.M, 128) * sln_cdiv(a.N, 64) >= 384;      // (pairing them as well: no difference at 128 .. 4 096 graphs, same box)
}

inline bool nt_unaligned(const GemmNTArgs& a) {        // a segment boundary inside a K tile: needs the MULTI = 2 body
  for (int s = 0; s + 1 < a.A.nseg; ++s) if (a.A.seg[s].len % BK) return true;
  return false;
}

inline int nt_amode(const GemmNTArgs& a) {
  bool x2 = false;
  for (int s = 0; s < a.A.nseg; ++s) x2 |= a.A.seg[s].x2 != nullptr;
  bool ident = !x2;
  for (int s = 0; s < a.A.nseg; ++s) ident = ident && a.A.seg[s].coef == SLN_COEF_IDENT;
  return x2 ? 1 : (ident ? 2 : 0);
}

inline bool tn_prepare(GemmTNArgs& a) {   // fills rows_per_block; returns whether G carries a second source
  bool x2 = false;
  for (int s = 0; s < a.G.nseg; ++s) x2 |= a.G.seg[s].x2 != nullptr;
  if (a.rows_per_block <= 0) {
    // measured on MI355X (tools/gemm_bench.py): ~768 blocks in flight, but never fewer than 256 rows per block -
    // below that the per-block prologue and the dW atomics (64x64 per block) dominate; chunks a multiple of BK rows
    static const int target = std::getenv("SLN_TN_TARGET") ? std::atoi(std::getenv("SLN_TN_TARGET")) : 768;
    static const int minrows = std::getenv("SLN_TN_MINROWS") ? std::atoi(std::getenv("SLN_TN_MINROWS")) : 256;
    const int tiles = sln_cdiv(a.Nout, 64) * sln_cdiv(a.Kin, 64);
    int chunks = sln_cdiv(target, tiles);
    int rpb = sln_cdiv(sln_cdiv(a.R, chunks), BK) * BK;
    a.rows_per_block = rpb < minrows ? minrows : rpb;
  }
  return x2;
}

}  // namespace
