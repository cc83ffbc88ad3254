// Shared device/host declarations for the 3D_SLN hot-path kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>

#define SLN_WAVE 64

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// BatchNorm view.  All hot-path MLPs are Linear -> [BatchNorm1d] -> ReLU (reference
// models/graph.py:10-27).  The kernels never materialise the normalised tensor: producers store the
// Linear output ("pre-activation") and accumulate column sums; consumers rebuild
// h = relu(scale*x + shift) while loading.  mode selects where (mean, 1/std) come from.
// ---------------------------------------------------------------------------------------------
enum { SLN_BN_NONE = 0, SLN_BN_TRAIN = 1, SLN_BN_EVAL = 2 };

struct BnView {
  const double* sums;    // [2][cstride]: sum x, sum x^2 over rows (train mode), offset to first column
  const double* gsums;   // [2][cstride]: sum g, sum g*xhat (backward), offset to first column
  const float* gamma;    // offset to first column (nullptr when mode == NONE)
  const float* beta;
  const float* rmean;
  const float* rvar;
  int cstride;           // distance between the two rows of sums / gsums
  int mode;
  float n_rows;          // rows the sums run over (exact in fp32 below 2^24); the reciprocal is taken in fp64: 1.0f / 12 is 6e-8
                         // off, and var = E[x^2] - mean^2 amplifies that by (mean / sigma)^2
  float eps;
  double rn;             // 1 / n_rows in fp64, taken on the host
};

// The coefficient set-up sits in the prologue of EVERY GEMM block, in front of its first MFMA: independent loads are issued
// together (they used to alternate with the arithmetic that consumed them: 3-4 serialized memory round trips), the row-count
// reciprocal comes from the host, and 1/sqrt is v_rsq_f32 on the fp32 variance (the variance itself - E[x^2] - mean^2, which
// cancels - stays in fp64; round 1 used three fp64 divisions per column here).
// (Round 2, from the ISA of the edge kernels: with the mode tests BETWEEN the loads - gamma / beta, branch, sums, and a second
// call for (mean, 1/std) that loaded the sums again - hipcc put an s_waitcnt vmcnt(0) in front of every group: three dependent
// round trips per coefficient table, six in front of a scatter / gather kernel's first row.  Every mode is now one straight
// block: all loads, then the arithmetic; bn_fwd_coef4 returns the four forward values from ONE set of loads.)
// BatchNorm parameters / statistics are read as GLOBAL loads: when the view itself was read out of device memory (the problem
// table of the per-pass wgrad launch) hipcc does not know the address space of its pointers and would emit flat loads.
__device__ __forceinline__ float sln_ldf(const float* p) { return *(const float __attribute__((address_space(1)))*)p; }
__device__ __forceinline__ double sln_ldd(const double* p) { return *(const double __attribute__((address_space(1)))*)p; }

__device__ __forceinline__ void bn_train_mean_istd(const BnView& b, double s1, double s2, float& mean, float& istd) {
  const double m = s1 * b.rn;
  double v = fma(-m, m, s2 * b.rn);
  v = v < 0.0 ? 0.0 : v;
  mean = (float)m;
  istd = __builtin_amdgcn_rsqf((float)v + b.eps);
}
__device__ __forceinline__ void bn_mean_istd(const BnView& b, int c, float& mean, float& istd) {
  if (b.mode == SLN_BN_TRAIN) {
    const double s1 = sln_ldd(b.sums + c), s2 = sln_ldd(b.sums + b.cstride + c);
    bn_train_mean_istd(b, s1, s2, mean, istd);
  } else if (b.mode == SLN_BN_EVAL) {
    const float rm = sln_ldf(b.rmean + c), rv = sln_ldf(b.rvar + c);
    mean = rm;
    istd = __builtin_amdgcn_rsqf(rv + b.eps);
  } else {
    mean = 0.f;
    istd = 1.f;
  }
}
// bn_fwd_coef4 for a train-mode view in two steps (loads, arithmetic): the masked epilogue's table is requested in FRONT of the
// operand's coefficient table and finished behind it - one memory round trip less in the helper wavefronts of a masked dgrad
struct BnFwdRaw { float gamma, beta; double s1, s2; };
__device__ __forceinline__ BnFwdRaw bn_fwd_train_load(const BnView& b, int c) {
  BnFwdRaw r;
  r.gamma = sln_ldf(b.gamma + c); r.beta = sln_ldf(b.beta + c); r.s1 = sln_ldd(b.sums + c); r.s2 = sln_ldd(b.sums + b.cstride + c);
  return r;
}
// forward coefficients (scale, shift, mean, istd): h = max(scale*x + shift, 0)
__device__ __forceinline__ float4 bn_fwd_coef4(const BnView& b, int c) {
  float mean, istd;
  if (b.mode == SLN_BN_TRAIN) {
    const float gamma = sln_ldf(b.gamma + c), beta = sln_ldf(b.beta + c);
    const double s1 = sln_ldd(b.sums + c), s2 = sln_ldd(b.sums + b.cstride + c);
    bn_train_mean_istd(b, s1, s2, mean, istd);
    const float scale = gamma * istd;
    return make_float4(scale, beta - mean * scale, mean, istd);
  }
  if (b.mode == SLN_BN_EVAL) {
    const float gamma = sln_ldf(b.gamma + c), beta = sln_ldf(b.beta + c), rm = sln_ldf(b.rmean + c), rv = sln_ldf(b.rvar + c);
    mean = rm;
    istd = __builtin_amdgcn_rsqf(rv + b.eps);
    const float scale = gamma * istd;
    return make_float4(scale, beta - mean * scale, mean, istd);
  }
  return make_float4(1.f, 0.f, 0.f, 1.f);
}
__device__ __forceinline__ float4 bn_fwd_train_finish(const BnView& b, const BnFwdRaw& r) {
  float mean, istd;
  bn_train_mean_istd(b, r.s1, r.s2, mean, istd);
  const float scale = r.gamma * istd;
  return make_float4(scale, r.beta - mean * scale, mean, istd);
}
// two columns of the same BatchNorm at once (the subject / object halves of GraphTripleConv's second Linear): one round trip
__device__ __forceinline__ void bn_fwd_coef4x2(const BnView& b, int ca, int cb, float4& va, float4& vb) {
  if (b.mode == SLN_BN_TRAIN) {
    const float ga = sln_ldf(b.gamma + ca), ba = sln_ldf(b.beta + ca), gb = sln_ldf(b.gamma + cb), bb = sln_ldf(b.beta + cb);
    const double a1 = sln_ldd(b.sums + ca), a2 = sln_ldd(b.sums + b.cstride + ca), b1 = sln_ldd(b.sums + cb), b2 = sln_ldd(b.sums + b.cstride + cb);
    float mean, istd;
    bn_train_mean_istd(b, a1, a2, mean, istd);
    float scale = ga * istd;
    va = make_float4(scale, ba - mean * scale, mean, istd);
    bn_train_mean_istd(b, b1, b2, mean, istd);
    scale = gb * istd;
    vb = make_float4(scale, bb - mean * scale, mean, istd);
    return;
  }
  va = bn_fwd_coef4(b, ca);
  vb = bn_fwd_coef4(b, cb);
}
__device__ __forceinline__ void bn_fwd_coef(const BnView& b, int c, float& scale, float& shift) {
  const float4 v = bn_fwd_coef4(b, c);
  scale = v.x; shift = v.y;
}
// backward coefficients: dX = p0*g + p1*x + p2  (g = relu-masked incoming gradient)
//   train: dX = scale*(g - mean(g) - xhat*mean(g*xhat)), xhat = (x-mean)*istd
__device__ __forceinline__ void bn_bwd_coef(const BnView& b, int c, float& p0, float& p1, float& p2) {
  if (b.mode == SLN_BN_TRAIN) {
    const float gamma = sln_ldf(b.gamma + c);
    const double g1 = sln_ldd(b.gsums + c), g2 = sln_ldd(b.gsums + b.cstride + c);
    const double s1 = sln_ldd(b.sums + c), s2 = sln_ldd(b.sums + b.cstride + c);
    float mean, istd;
    bn_train_mean_istd(b, s1, s2, mean, istd);
    const float scale = gamma * istd;
    const float c1 = (float)(g1 * b.rn);
    const float c2 = (float)(g2 * b.rn);
    p0 = scale;
    p1 = -scale * istd * c2;
    p2 = -scale * c1 - p1 * mean;
    return;
  }
  if (b.mode == SLN_BN_EVAL) {
    const float gamma = sln_ldf(b.gamma + c), rv = sln_ldf(b.rvar + c);
    p0 = gamma * __builtin_amdgcn_rsqf(rv + b.eps);
    p1 = 0.f; p2 = 0.f;
    return;
  }
  p0 = 1.f; p1 = 0.f; p2 = 0.f;
}

// ---------------------------------------------------------------------------------------------
// Logical GEMM operand: up to three column segments of row-major sources, each optionally
// row-gathered (GraphTripleConv's [obj[s] | pred | obj[o]] concat, models/graph.py:74-80) and
// transformed on load as   v = max(c0*x1 + c1*x2 + c2, floor)   with per-column coefficients.
// Every segment except the last must have len % 32 == 0; all lens % 4 == 0.
// ---------------------------------------------------------------------------------------------
enum { SLN_COEF_IDENT = 0, SLN_COEF_FWD = 1, SLN_COEF_FWD_NORELU = 2, SLN_COEF_BWD = 3 };

struct Seg {
  const float* x1;
  const float* x2;   // second source (pre-activation for BWD coefficients); nullptr otherwise
  int ld1, ld2;      // row strides in floats
  int c1, c2;        // first source column
  int len;           // logical columns in this segment
  int which;         // 0: identity rows, 1: rows = idx_a[row], 2: rows = idx_b[row]
  int coef;          // SLN_COEF_*
  int pad_;
  BnView bn;         // statistics/parameters aligned with the segment's first column
};

struct Operand {
  Seg seg[3];
  const int* idx_a;
  const int* idx_b;
  int nseg;
  int rows;          // logical rows
  int cols;          // logical columns (sum of seg lens)
  int pad_;
};

__device__ __forceinline__ float4 sln_coef_for(const Seg& s, int c) {
  // returns (c0, c1, c2, floor) for logical column c of this segment
  float4 r;
  const float NEG = -3.0e38f;
  if (s.coef == SLN_COEF_IDENT) { r = make_float4(1.f, 0.f, 0.f, NEG); }
  else if (s.coef == SLN_COEF_BWD) {
    float p0, p1, p2; bn_bwd_coef(s.bn, c, p0, p1, p2);
    r = make_float4(p0, p1, p2, NEG);
  } else {
    float sc, sh; bn_fwd_coef(s.bn, c, sc, sh);
    r = make_float4(sc, 0.f, sh, s.coef == SLN_COEF_FWD ? 0.f : NEG);
  }
  return r;
}

// Fill an LDS table coef[0..cols) for the whole operand (all threads of the block participate).
// Column by column (the generic loop at the end) every further column of a thread is another dependent memory round trip in front
// of the block's first barrier: the helper wavefronts of an NT kernel had their table written 2 200 clocks after the block's start
// with one column per thread and +950 per additional one (tools/lab/gemm_lab.hip, helper-wave stamps).  The three 128-column
// segments of GraphTripleConv's concat were three such trips: that case loads what a thread's first column needs in EVERY segment
// in one basic block (unconditional loads at clamped columns, only the LDS store is predicated), then does the arithmetic (the
// expressions of bn_fwd_coef4, same order, same bits): -0.8 us per launch.  (The same for several columns of ONE segment - K = 640
// is three columns per helper thread - shortened the lab kernel's table by 1 000 clocks and left the product kernel 0.3 us slower:
// not kept.)
__device__ __forceinline__ bool sln_seg_fwd_train(const Seg& s) { return (s.coef == SLN_COEF_FWD || s.coef == SLN_COEF_FWD_NORELU) && s.bn.mode == SLN_BN_TRAIN; }
__device__ __forceinline__ float4 sln_fwd_train_coef(const Seg& g, float gamma, float beta, double s1, double s2) {
  float mean, istd;
  bn_train_mean_istd(g.bn, s1, s2, mean, istd);
  const float scale = gamma * istd;
  return make_float4(scale, 0.f, beta - mean * scale, g.coef == SLN_COEF_FWD ? 0.f : -3.0e38f);
}
// NSEG_MAX: what the caller knows at compile time - 1: a single segment (one plain loop: no segment bookkeeping, no reads of the
// other segments' fields), 3: GraphTripleConv's concat is possible (the three-segment path), 0: nothing (the generic loop only).
// A cold prologue pays for every kernel-argument field it touches (hipcc fetches them stage by stage, ~400 clocks per scalar-cache
// miss, tools/lab/kernarg_lines.hip) and for the code it jumps over: with the fast paths in front of the generic loop of EVERY
// caller the K <= 256 kernels that never take them were 0.5-1.2 us slower per launch; with the single-segment callers on their own
// three-line loop they are 0.1-1.4 us faster than before.
// Layout of an NT kernel's coefficient table in LDS: one float4 per operand column, with one unused row behind every 16 columns.
// A staging lane reads the four rows of its four columns (64 bytes, lanes 64 bytes apart): in a plain array lanes kq and kq + 4 of a
// ds_read_b128 met on the same bank group - the two-way conflict behind the NT kernels' SQ_LDS_BANK_CONFLICT share of 0.14-0.33
// (the identity-operand variants, which have no table, showed 0.000).  With the gap the upper four lanes sit one bank group further.
__host__ __device__ __forceinline__ int sln_cidx(int c) { return c + (c >> 4); }
__host__ __device__ __forceinline__ int sln_crows(int cols) { return cols + ((cols + 15) >> 4); }      // rows of a table of `cols` columns

template <int NSEG_MAX = 0>
__device__ __forceinline__ void sln_fill_coefs(const Operand& op, float4* coef, int tid, int nthreads) {
  if (NSEG_MAX == 1) {
    const Seg& g = op.seg[0];
    for (int c = tid; c < g.len; c += nthreads) coef[sln_cidx(c)] = sln_coef_for(g, c);
    return;
  }
  if (NSEG_MAX == 3 && op.nseg == 3 && sln_seg_fwd_train(op.seg[0]) && sln_seg_fwd_train(op.seg[1]) && sln_seg_fwd_train(op.seg[2]) &&
      op.seg[0].len > 0 && op.seg[1].len > 0 && op.seg[2].len > 0) {
    // GraphTripleConv's concat [obj[s] | pred | obj[o]]: the first column a thread owns in each of the three segments together
    float ga[3], be[3]; double s1[3], s2[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const BnView& b = op.seg[s].bn;
      const int c = min(tid, op.seg[s].len - 1);
      ga[s] = sln_ldf(b.gamma + c); be[s] = sln_ldf(b.beta + c); s1[s] = sln_ldd(b.sums + c); s2[s] = sln_ldd(b.sums + b.cstride + c);
    }
    int base = 0;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int len = op.seg[s].len;
      const float4 v = sln_fwd_train_coef(op.seg[s], ga[s], be[s], s1[s], s2[s]);
      if (tid < len) coef[sln_cidx(base + tid)] = v;
      for (int c = tid + nthreads; c < len; c += nthreads) coef[sln_cidx(base + c)] = sln_coef_for(op.seg[s], c);
      base += len;
    }
    return;
  }
  int base = 0;
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    if (s < op.nseg) {
      for (int c = tid; c < op.seg[s].len; c += nthreads) coef[sln_cidx(base + c)] = sln_coef_for(op.seg[s], c);
      base += op.seg[s].len;
    }
  }
}

// Order-independent fp64 accumulation (round 3).  The column statistics and the loss terms are sums of per-block partials added
// with fp64 atomics, in whatever order the blocks arrive.  A partial rounded to a multiple of 2^-k is added EXACTLY as long as
// the running sum stays below 2^(53-k) - and exact additions commute: the total no longer depends on the arrival order (two
// runs give bit-identical BatchNorm statistics and losses).  The rounding itself moves a partial by at most 2^-(k+1):
//   forward sums (sum y, sum y^2)        k = 20: 5e-7 absolute per block on sums of magnitude >= 1, exact below 8.6e9
//   backward sums (sum g, sum g*xhat)    k = 44: 3e-14 absolute on sums of magnitude 1e-6 .. 1, exact below 512
//   loss terms                           k = 24: exact below 5e8
// Beyond those ranges rint() is the identity or the additions round as before: values stay correct, only the order
// independence is lost.
#define SLN_Q_FWD 1048576.0                   /* 2^20 */
#define SLN_Q_BWD 17592186044416.0            /* 2^44 */
#define SLN_Q_LOSS 16777216.0                 /* 2^24 */
__device__ __forceinline__ double sln_qd(double v, double scale) { return rint(v * scale) * (1.0 / scale); }
// Forward sums (round 5): the quantum follows the ROW COUNT of the launch (every block of a launch sees the same M, so the
// partials still add exactly and commute): 2^-max(20, 33 - ceil(log2 M)).  With the fixed 2^-20 an 8-row BatchNorm (BASELINE
// config c1, the refinement loop's rooms) lost to the quantum what fp32 rounding loses to a sum of ~10 - and var = E[x^2] - mean^2
// amplifies that by mean^2 / var (~100 when a column's mean is 10 x its spread): 4e-4 on the gradients of an 8-object graph
// against the fp64 oracle (tests/test_vae_gpu.py::test_tight_gradients..[batch-True-1]; 4e-7 with this rule).  Exact - hence
// order-independent - while the column's mean of y^2 stays below ~1e6 (2^(53 - k) / M), for every M; M >= 8192 is the old rule.
__device__ __forceinline__ double sln_q_fwd(int M) {
  const int lg = M > 1 ? 32 - __builtin_clz((unsigned)(M - 1)) : 0;          // ceil(log2 M)
  const int k = 33 - lg > 20 ? 33 - lg : 20;
  return (double)(1ull << k);
}

__device__ __forceinline__ float wave_sum_halves(float v) {   // lanes l and l^32 -> both hold the sum
  return v + __shfl_xor(v, 32, 64);
}

static inline int sln_cdiv(int a, int b) { return (a + b - 1) / b; }

// Zero-fill as a kernel, for buffers a caller may have just re-allocated inside a stream capture.  A captured hipMemsetAsync on
// a block the caching allocator handed out again (the gradient tensor of a backward pass taking the place of a just-freed
// workspace) replayed BEFORE the kernels that still wrote the old tenant: the first / last floats of the gradient came back as
// garbage in two replays out of three (tools/lab/dbg_graph.py; a captured memset on its own is fine, tools/lab/dbg_memset.py).
// bytes % 4 == 0.
template <typename T>
__global__ void sln_zero_kernel(T* p, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = T{};
}
static inline int sln_zero_async(void* p, size_t bytes, hipStream_t st) {
  if (bytes < 4) return 0;
  if (reinterpret_cast<uintptr_t>(p) % 16 == 0 && bytes % 16 == 0) {
    const size_t n = bytes / 16;
    hipLaunchKernelGGL(sln_zero_kernel<uint4>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, static_cast<uint4*>(p), n);
  } else {
    const size_t n = bytes / 4;
    hipLaunchKernelGGL(sln_zero_kernel<uint32_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, static_cast<uint32_t*>(p), n);
  }
  return (int)hipGetLastError();
}

#define SLN_CHECK_LAUNCH()                                                  \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) return (int)e__;                                 \
  } while (0)

// A pooled stream that overlaps with `main` (streams.hip: two streams on one hardware queue do not; probed once per caller stream);
// nullptr when none can be had right now (first use inside a stream capture).  Plain engine side streams: sln_side_stream_create.
hipStream_t sln_overlapping_stream(hipStream_t main);
inline hipError_t sln_side_stream_create(hipStream_t* s) { return hipStreamCreateWithFlags(s, hipStreamNonBlocking); }
inline bool sln_capturing(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}
