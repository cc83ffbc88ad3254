// SPADEGenerator4 forward (reference models/SPADE_related.py:70-85,128-149,1404-1605) on gfx950.
//
// 152.6 GMAC per 256x256 image, 98 % of it 3x3 reflect-padded convolutions, 72 % in the per-SPADE
// gamma/beta/shared convs whose outputs only exist to modulate the normalised activation.  The workhorse
// is one implicit-GEMM kernel on v_mfma_f32_32x32x2_f32 (exact fp32, the 1e-4 budget of the north star
// leaves no room for bf16):
//     out[co, pixel] = sum_{ci, tap} Wp[tap][ci][co] * x[ci, reflect(pixel + tap)]
//   * M = output channels (A operand = packed weights), N = pixels (B operand = activations): the MFMA
//     result then has lanes = 32 consecutive pixels, i.e. coalesced NCHW stores and epilogue loads;
//   * a workgroup owns a (16 x 16)- or (8 x 16)-pixel patch x 128 (or 64) channels; per chunk of 4 or 8 input channels the
//     reflect-padded halo is staged once in LDS and reused for all 9 taps (9x fewer global loads than im2col) together with
//     the [9][chunk][channels] weight slab.  Two kernels share the K loop and the epilogue: conv_glds_kernel DMAs the next
//     chunk straight into a second LDS buffer (global_load_lds) while the MFMAs of the current one run; conv_mfma_kernel
//     (64-row blocks on small grids, 1x1 convs, tiny images, Cin % 8 != 0) prefetches it into registers;
//   * epilogues: bias + {none, ReLU, LeakyReLU(s)}; or the SPADE modulation - the gamma and beta channels
//     of one feature land in the same lane/register of two accumulators (weights are packed [32 gamma | 32
//     beta] per 64 rows), so out = (x - mu_b) * inv_b * (1 + gamma) + beta [-> LeakyReLU(0.2)] is computed in
//     registers and gamma/beta never touch HBM.
#include <mutex>
#include <vector>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>

#include "../../include/sln_hip.h"
#include "sln_common.h"
#include "sln_prof.h"

namespace {

constexpr int CK = 8;                  // input channels per LDS chunk
constexpr int TH = 8, TW = 16;         // pixel patch per workgroup (128 pixels)
constexpr int HALO = (TH + 2) * (TW + 2);

enum { CEPI_BIAS_ACT = 0, CEPI_MODULATE = 1 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LEAKY = 2 };

struct ConvArgs {
  const float* x;        // [B, Cin, H, W]
  const float* wp;       // packed weights [KS*KS][Cin][rows_pad]
  const float* bias;     // [rows_pad] packed like the rows (or nullptr)
  float* y;              // BIAS_ACT: [B, Cout, H, W];  MODULATE: [B, C, H, W]
  int B, Cin, H, W, rows, rows_pad;   // rows = logical output rows (Cout, or 2C packed for MODULATE)
  int act; float slope;
  // MODULATE
  const float* xin;      // tensor being normalised [B, C, H, W]
  const float* stats;    // [B, 2] = (mean, 1/(std+eps))
  int C;
  int xin_up;            // MODULATE: xin is [B, C, H/2, W/2] and is read through nn.Upsample(x2, nearest) (never materialised)
  // BIAS_ACT, optional reductions of what the epilogue writes (fp64 atomics, zeroed by the caller)
  double* ln_acc;        // [B][LN_ACC_STRIDE]: sum y, sum y^2 of sample b (LayerNorm2D statistics of the NEXT SPADE layer)
  double* gap_acc;       // [B, rows]: sum over pixels (SEBlock2's global average pool)
  int blocked;           // accumulate in blocks of input channels (see BLK below)
  // launches of a few workgroups (batch-1 calls on the 8 x 8 .. 64 x 64 layers): gridDim.z workgroups share the input channels of
  // an output tile, each stores its raw partial sums to part + blockIdx.z * part_stride ([B, rows, H, W] each) and
  // conv_split_finish_kernel adds them in order, then bias / activation / the reductions (see launch_conv_blocked)
  float* part;
  long part_stride;
};
constexpr int LN_ACC_STRIDE = 16;     // doubles between the accumulators of two samples (one 128-byte line each)

// Read one accumulator element where it is used.  Left to itself the register allocator copies all 64 accumulator registers of
// a wave out of the AGPRs in one place after the K loop, and that copy - not the loop - sets the kernel's VGPR count (182 -> 2
// waves per SIMD); pinned to the use, the epilogue works through 16 values at a time and three workgroups fit a CU.
__device__ __forceinline__ float acc_read(const float& v) {
  float r;
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(r) : "a"(v));
  return r;
}

__device__ __forceinline__ int reflect_idx(int i, int n) {      // ReflectionPad2d(1) (pad < n)
  i = i < 0 ? -i : i;
  return i >= n ? 2 * n - 2 - i : i;
}

// Epilogue shared by the conv kernels.  4 waves: WM x WN over (rows, pixels); acc[i][j] = 32 rows x 32 pixels.
template <int BMC, int EPI, int THT = TH>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x16 (&acc)[2][(BMC / 64) * THT * TW / 4 / 32], float* lds, int b, int r0,
                                              int x0, int y0) {
  constexpr int WM = BMC / 64, WN = 4 / WM, TN = THT * TW / WN / 32, TM = 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = (wave / WN) * 64, wp0 = (wave % WN) * (THT * TW / WN);
  const int li = lane & 31, lk = lane >> 5;
  const size_t plane = (size_t)a.H * a.W;
  // ---- epilogue: lane = pixel (li), register r = row (r&3) + 8 (r>>2) + 4 lk inside the 32-row tile.
  // All loads first (bias per row once, x per output element; unconditional, clamped addresses), then the arithmetic, then the
  // stores: written as "if (valid) { load, load, load, compute, store }" per element, hipcc waited for every element's loads
  // (and the previous element's store) before issuing the next ones - 32 serialized memory round trips per lane, ~10 us per
  // 128 x 128 tile of the modulation convolutions.
  if (EPI == CEPI_BIAS_ACT && a.part != nullptr) {      // input-channel split: raw sums, no bias, no reductions
    float* __restrict__ pz = a.part + (size_t)blockIdx.z * a.part_stride;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int m = wp0 + 32 * j + li;
        const int py = y0 + m / TW, px = x0 + m % TW;
        const bool pv = py < a.H && px < a.W;
        const size_t pix = (size_t)py * a.W + px;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = r0 + wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lk;
          const float v = acc_read(acc[i][j][r]);
          if (pv && row < a.rows) pz[((size_t)b * a.rows + row) * plane + pix] = v;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    return;
  }
  if (EPI == CEPI_BIAS_ACT) {
    // one 32 x 32 tile at a time, the scheduler fenced between tiles: with everything hoisted the epilogue, not the K loop, set
    // the kernel's register count and cost a third of the occupancy
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float bias[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = min(r0 + wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lk, a.rows - 1);
        bias[r] = a.bias ? a.bias[row] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int m = wp0 + 32 * j + li;
        const int py = y0 + m / TW, px = x0 + m % TW;
        const bool pv = py < a.H && px < a.W;
        const size_t pix = (size_t)py * a.W + px;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = r0 + wr + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lk;
          float v = acc_read(acc[i][j][r]) + bias[r];
          if (a.act == ACT_RELU) v = fmaxf(v, 0.f);
          else if (a.act == ACT_LEAKY) v = v > 0.f ? v : v * a.slope;
          const bool ok = pv && row < a.rows;
          if (ok) a.y[((size_t)b * a.rows + row) * plane + pix] = v;
          acc[i][j][r] = ok ? v : 0.f;               // what was written, for the reductions below
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (a.ln_acc) {
      // LayerNorm2D sums of the written values in fp64 (as the stand-alone statistics kernel); 2 atomics per block
      double d1 = 0.0, d2 = 0.0;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) { const double v = acc_read(acc[i][j][r]); d1 += v; d2 = fma(v, v, d2); }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) { d1 += __shfl_xor(d1, off); d2 += __shfl_xor(d2, off); }
      double* red = reinterpret_cast<double*>(lds);      // the slabs are free: the K loop ended with a barrier
      if (lane == 0) { red[2 * wave] = d1; red[2 * wave + 1] = d2; }
      __syncthreads();
      if (tid == 0) {
        atomicAdd(a.ln_acc + LN_ACC_STRIDE * b, red[0] + red[2] + red[4] + red[6]);
        atomicAdd(a.ln_acc + LN_ACC_STRIDE * b + 1, red[1] + red[3] + red[5] + red[7]);
      }
      __syncthreads();
    }
    if (a.gap_acc) {
      // per-row sums over the block's pixels: the lane's pixel tiles in registers, the 32 pixel lanes through a padded LDS
      // transpose (row-major, 33 floats per row), then one fp64 atomic per (wave, row)
      float* tr = lds + wave * (64 * 33);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = 0.f;
#pragma unroll
          for (int j = 0; j < TN; ++j) v += acc_read(acc[i][j][r]);
          tr[(32 * i + (r & 3) + 8 * (r >> 2) + 4 * lk) * 33 + li] = v;
        }
      __syncthreads();
      double rs = 0.0;
#pragma unroll
      for (int q = 0; q < 32; ++q) rs += (double)tr[lane * 33 + q];
      const int row = r0 + wr + lane;
      if (row < a.rows) atomicAdd(a.gap_acc + (size_t)b * a.rows + row, rs);
    }
  } else {
    // rows of this wave: [32 gamma | 32 beta] of channels cbase .. cbase+31
    const int cbase = (r0 + wr) / 2;
    const float mean = a.stats[2 * b], inv = a.stats[2 * b + 1];
    // uniform per-sample bases + 32-bit lane offsets (one VGPR per address instead of two; C * plane < 2^31, checked by the caller)
    const unsigned xplane = a.xin_up ? (unsigned)(plane >> 2) : (unsigned)plane;
    const float* __restrict__ xinb = a.xin + (size_t)b * a.C * xplane;
    float* __restrict__ yb = a.y + (size_t)b * a.C * plane;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int m = wp0 + 32 * j + li;
      const int py = y0 + m / TW, px = x0 + m % TW;
      const bool pv = py < a.H && px < a.W;
      const unsigned pix = (unsigned)py * a.W + px;
      const int pyc = min(py, a.H - 1), pxc = min(px, a.W - 1);
      const unsigned pixc = a.xin_up ? (unsigned)(pyc >> 1) * (a.W >> 1) + (pxc >> 1) : (unsigned)pyc * a.W + pxc;
      float xin[16], gb[16], bb[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cl = (r & 3) + 8 * (r >> 2) + 4 * lk;
        const unsigned c = min(cbase + cl, a.C - 1);
        xin[r] = xinb[c * xplane + pixc];
        gb[r] = a.bias[r0 + wr + cl];
        bb[r] = a.bias[r0 + wr + 32 + cl];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = cbase + (r & 3) + 8 * (r >> 2) + 4 * lk;
        const float gamma = acc_read(acc[0][j][r]) + gb[r];
        const float beta = acc_read(acc[1][j][r]) + bb[r];
        float v = (xin[r] - mean) * inv;
        v = v * (1.f + gamma) + beta;
        if (a.act == ACT_LEAKY) v = v > 0.f ? v : v * a.slope;
        if (pv && c < a.C) yb[(unsigned)c * (unsigned)plane + pix] = v;
      }
      __builtin_amdgcn_sched_barrier(0);             // one pixel tile's 16 loads in flight at a time (register count, see above)
    }
  }
}

// Blocked accumulation: tot += acc; acc = 0 - or, behind the last chunk, acc = the total (the epilogue reads `acc`).
template <int TM, int TN, int TMT, int TNT>
__device__ __forceinline__ void conv_flush(f32x16 (&acc)[TM][TN], f32x16 (&tot)[TMT][TNT], bool last) {
  static_assert(TM == TMT && TN == TNT, "blocked accumulation needs a full second register set");
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float t = tot[i][j][r] + acc[i][j][r];
        tot[i][j][r] = t;
        acc[i][j][r] = last ? t : 0.f;
      }
}

// BMC: channels (rows) per block (64 or 128); KS: 1 or 3.  4 waves: WM x WN over (rows, pixels).
//
// BLK > 0: BLOCKED ACCUMULATION (round 4).  An MFMA accumulator is, bit for bit, a serial fmaf chain over K = 9 Cin
// (tools/lab/mfma_round.hip): its rounding error grows like K (rms 4e-7 of the output scale at K = 9 216, 1.2e-7 at 1 152)
// while torch's CPU convolution - what the reference runs - sums in short blocks and stays at 4-6e-8 whatever K is.  With BLK
// the accumulators hold BLK chunks only (BLK x CK or GK input channels x 9 taps = 144 products), are then added to a second
// fp32 register set and restart from zero: error ~ sqrt(K/2 (n_b + K/n_b)) instead of K/sqrt(2) - 7.9e-8 at K = 9 216.  The
// second set costs 64 registers with 64 accumulators per lane (the 8 x 16-pixel / 64-row variants), which is why the 16 x 16
// x 128-row workgroups (128 accumulators) have no such form; it is used where it matters: the main 3x3 convolutions with
// Cin >= 512 (tools/spade_error_budget.py section 4: 15 % of the MACs).
template <int BMC, int KS, int EPI, int BLK = 0>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvArgs a) {
  constexpr int TAPS = KS * KS;
  constexpr int WM = BMC / 64;                 // waves along rows (each wave: 64 rows = 2 tiles)
  constexpr int WN = 4 / WM;                   // waves along pixels
  constexpr int TN = 128 / WN / 32;            // pixel tiles per wave (128 pixels per block)
  constexpr int TM = 2;
  constexpr int HS = KS == 3 ? HALO : TH * TW; // floats per channel in the LDS patch
  constexpr int WSLAB = TAPS * CK * BMC;
  constexpr int NW4 = (WSLAB / 4 + 255) / 256; // float4 weight loads per thread per chunk (tail slots clamped)
  constexpr int NH = (CK * HS + 255) / 256;    // scalar patch loads per thread per chunk
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* wl = lds;                             // [TAPS][CK][BMC]
  float* xl = lds + WSLAB;                     // [CK][HS]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + TH - 1) / TH;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y; bid /= tiles_y;
  const int b = bid;
  const int r0 = blockIdx.y * BMC;             // first packed row of this block
  const int x0 = tx * TW, y0 = ty * TH;
  const size_t plane = (size_t)a.H * a.W;
  const float* xb = a.x + (size_t)b * a.Cin * plane;

  // per-thread patch positions (constant over chunks): element e -> (channel-in-chunk, pos) -> global offset
  int h_off[NH], h_lds[NH];
#pragma unroll
  for (int j = 0; j < NH; ++j) {
    const int e = tid + 256 * j;
    const int c = e / HS, p = e % HS;
    int gy, gx;
    if (KS == 3) { gy = reflect_idx(y0 + p / (TW + 2) - 1, a.H); gx = reflect_idx(x0 + p % (TW + 2) - 1, a.W); }
    else { gy = min(y0 + p / TW, a.H - 1); gx = min(x0 + p % TW, a.W - 1); }
    gy = min(max(gy, 0), a.H - 1); gx = min(max(gx, 0), a.W - 1);
    h_off[j] = e < CK * HS ? (int)(c * plane + (size_t)gy * a.W + gx) : 0;     // tail slots load a valid address, never stored
    h_lds[j] = e;
  }
  float hreg[NH];
  float4 wreg[NW4];
  const int nchunks = (a.Cin + CK - 1) / CK;

  auto gload = [&](int ch) {
    const int ci0 = ch * CK;
#pragma unroll
    for (int j = 0; j < NH; ++j) {
      const int c = min((tid + 256 * j) / HS, CK - 1);
      // unconditional load (clamped channel); surplus channels / tail slots are dropped at the LDS store
      const int cc = min(ci0 + c, a.Cin - 1) - c;
      hreg[j] = xb[(ptrdiff_t)cc * (ptrdiff_t)plane + h_off[j]];
    }
#pragma unroll
    for (int j = 0; j < NW4; ++j) {
      const int e4 = min(tid + 256 * j, WSLAB / 4 - 1);     // float4 index inside the slab [TAPS][CK][BMC/4]
      const int col4 = e4 % (BMC / 4), rr = e4 / (BMC / 4); // rr = tap*CK + c
      const int tap = rr / CK, c = rr % CK;
      const int ci = min(ci0 + c, a.Cin - 1);
      wreg[j] = *reinterpret_cast<const float4*>(a.wp + ((size_t)tap * a.Cin + ci) * a.rows_pad + r0 + 4 * col4);
    }
  };
  auto lstore = [&](int ch) {
    const int ci0 = ch * CK;
#pragma unroll
    for (int j = 0; j < NH; ++j) {
      const int e = h_lds[j];
      if (e < CK * HS) xl[e] = (ci0 + e / HS) < a.Cin ? hreg[j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NW4; ++j) {
      const int e4 = tid + 256 * j;
      const int c = (e4 / (BMC / 4)) % CK;
      float4 v = wreg[j];
      if (ci0 + c >= a.Cin) v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e4 < WSLAB / 4) *reinterpret_cast<float4*>(wl + 4 * e4) = v;
    }
  };

  const int wr = (wave / WN) * 64;                 // wave's first row inside the block
  const int wp0 = (wave % WN) * (128 / WN);        // wave's first pixel inside the patch
  const int li = lane & 31, lk = lane >> 5;
  f32x16 acc[TM][TN], tot[BLK ? TM : 1][BLK ? TN : 1];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; if (BLK) tot[i][j][r] = 0.f; }
  // pixel of this lane for pixel-tile j: m = wp0 + 32 j + li -> (py, px)
  int pbase[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int m = wp0 + 32 * j + li;
    pbase[j] = KS == 3 ? (m / TW) * (TW + 2) + (m % TW) : m;
  }

  gload(0);
  lstore(0);
  __syncthreads();
  for (int ch = 0; ch < nchunks; ++ch) {
    if (ch + 1 < nchunks) gload(ch + 1);
    // 36 (tap, channel pair) steps of 4 MFMAs; the operands of step s + 1 are read from LDS before the MFMAs of step s issue
    // (hipcc on its own sinks each step's ds_reads to right in front of its first MFMA: the scheduler is fenced)
    // 36 (tap, channel pair) steps of 4 MFMAs; the operands of step s + 1 are read from LDS before the MFMAs of step s issue
    // (hipcc on its own sinks each step's ds_reads to right in front of its first MFMA: the scheduler is fenced).  All 36 steps
    // unrolled: rolled per tap, the 25 scalar / address instructions at the loop end outlast the tap's last MFMA and the K loop
    // alone drops from 154 to 139 TFLOP/s (tools/lab/mfma_peak modes 6 / 4); the two-way bank conflict of the two-row pixel tile
    // costs nothing once unrolled (modes 6 / 7), a conflict-free 48-float row stride measured 0.7 % slower (larger LDS image).
    constexpr int NS = TAPS * (CK / 2);
    auto ld = [&](int st, float (&av)[TM], float (&bv)[TN]) {
      const int tap = st / (CK / 2), kk = (st % (CK / 2)) * 2;
      const int toff = KS == 3 ? (tap / 3) * (TW + 2) + (tap % 3) : 0;
#pragma unroll
      for (int i = 0; i < TM; ++i) av[i] = wl[(tap * CK + kk + lk) * BMC + wr + 32 * i + li];
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[j] = xl[(kk + lk) * HS + pbase[j] + toff];
    };
    auto mma = [&](const float (&av)[TM], const float (&bv)[TN]) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    };
    float av0[TM], bv0[TN], av1[TM], bv1[TN];
    ld(0, av0, bv0);
#pragma unroll
    for (int st = 0; st < NS; st += 2) {
      ld(st + 1, av1, bv1);
      __builtin_amdgcn_sched_barrier(0);
      mma(av0, bv0);
      __builtin_amdgcn_sched_barrier(0);
      if (st + 2 < NS) ld(st + 2, av0, bv0);
      __builtin_amdgcn_sched_barrier(0);
      mma(av1, bv1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (BLK > 0) { if ((ch + 1) % BLK == 0 || ch + 1 == nchunks) conv_flush(acc, tot, ch + 1 == nchunks); }
    __syncthreads();                     // everyone done reading the LDS slab
    if (ch + 1 < nchunks) { lstore(ch + 1); __syncthreads(); }
  }

  conv_epilogue<BMC, EPI>(a, acc, lds, b, r0, x0, y0);
}

// The same convolution with its operands DMA'd straight into LDS (global_load_lds; Cin % GK == 0): no staging registers and no
// ds_write pass between two barriers per chunk.  Two LDS buffers of GK input channels each ([9][GK][rows] weights + the
// GK-channel halo): the loads of chunk c + 1 are issued right after the single barrier that opens chunk c and have the whole
// chunk to land.  THT x 16 pixels per workgroup: 8 x 16 (64 accumulator registers per lane, 72 MFMAs per wave and chunk at
// GK = 4) or 16 x 16 (128 accumulators, 144 MFMAs per wave and chunk, half the weight DMA per MFMA; 116 VGPRs + 128 AGPRs and
// 49 KB of LDS: two workgroups per CU).  Ablation of the staged kernel on the modulation convs of up_3 (B 32, 128 -> 2 x 128
// channels at 256 x 256, 9.8 ms): 8.84 ms without its per-chunk load / ds_write / second barrier; this kernel: 9.06 ms.
template <int BMC, int KS, int EPI, int THT, int GK, int BLK = 0>      // GK input channels per LDS buffer (4 or 8); BLK: see conv_mfma_kernel
__global__ __launch_bounds__(256) void conv_glds_kernel(const ConvArgs a) {
  constexpr int TAPS = KS * KS;
  constexpr int WM = BMC / 64, WN = 4 / WM, TN = THT * TW / WN / 32, TM = 2;   // THT x 16 pixels per workgroup (THT = 8 or 16)
  constexpr int HS = KS == 3 ? (THT + 2) * (TW + 2) : THT * TW;
  constexpr int WSLAB = TAPS * GK * BMC;            // floats; WSLAB / 4 float4 is a multiple of 64: whole waves per round
  constexpr int NWR = (WSLAB / 4 + 255) / 256;      // rounds of 256 x 16 B
  constexpr int NHR = (GK * HS + 255) / 256;        // rounds of 256 x 4 B (the last one runs past the patch into padding)
  constexpr int BUF = WSLAB + NHR * 256;
  static_assert((WSLAB / 4) % 64 == 0, "weight slab is a whole number of wave-wide 16-byte loads");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  typedef __attribute__((address_space(3))) void* lds_ptr;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_x = (a.W + TW - 1) / TW, tiles_y = (a.H + THT - 1) / THT;
  int bid = blockIdx.x;
  const int tx = bid % tiles_x; bid /= tiles_x;
  const int ty = bid % tiles_y; bid /= tiles_y;
  const int b = bid;
  const int r0 = blockIdx.y * BMC;
  const int x0 = tx * TW, y0 = ty * THT;
  const size_t plane = (size_t)a.H * a.W;
  const float* xb = a.x + (size_t)b * a.Cin * plane;

  // per-lane source offsets inside a chunk (constant over chunks)
  unsigned h_off[NHR], w_off[NWR];
#pragma unroll
  for (int j = 0; j < NHR; ++j) {
    const int e = tid + 256 * j;
    const int c = e / HS, p = e % HS;
    int gy, gx;
    if (KS == 3) { gy = reflect_idx(y0 + p / (TW + 2) - 1, a.H); gx = reflect_idx(x0 + p % (TW + 2) - 1, a.W); }
    else { gy = y0 + p / TW; gx = x0 + p % TW; }
    gy = min(max(gy, 0), a.H - 1); gx = min(max(gx, 0), a.W - 1);
    h_off[j] = e < GK * HS ? (unsigned)(c * plane + (size_t)gy * a.W + gx) : 0u;     // padding lanes fetch a valid word
  }
#pragma unroll
  for (int j = 0; j < NWR; ++j) {
    const int e4 = min(tid + 256 * j, WSLAB / 4 - 1);
    const int col4 = e4 % (BMC / 4), rr = e4 / (BMC / 4);   // rr = tap * GK + c
    w_off[j] = (unsigned)(((rr / GK) * a.Cin + rr % GK) * a.rows_pad + 4 * col4);
  }
  auto issue = [&](int ch, int buf) {
#if defined(__HIP_DEVICE_COMPILE__)      // the DMA builtin exists in the device pass only (the host pass must still emit the launch stub)
    float* wb = lds + buf * BUF;
    const float* ws = a.wp + (size_t)ch * GK * a.rows_pad + r0;
    const float* xs = xb + (size_t)ch * GK * plane;
#pragma unroll
    for (int j = 0; j < NWR; ++j)
      if (256 * (j + 1) <= WSLAB / 4 || 256 * j + 64 * wave < WSLAB / 4)
        __builtin_amdgcn_global_load_lds(ws + w_off[j], (lds_ptr)(wb + 4 * (256 * j + 64 * wave)), 16, 0, 0);
#pragma unroll
    for (int j = 0; j < NHR; ++j)
      __builtin_amdgcn_global_load_lds(xs + h_off[j], (lds_ptr)(wb + WSLAB + 256 * j + 64 * wave), 4, 0, 0);
#endif
  };

  const int wr = (wave / WN) * 64, wp0 = (wave % WN) * (THT * TW / WN);
  const int li = lane & 31, lk = lane >> 5;
  f32x16 acc[TM][TN], tot[BLK ? TM : 1][BLK ? TN : 1];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; if (BLK) tot[i][j][r] = 0.f; }
  int pbase[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int m = wp0 + 32 * j + li;
    pbase[j] = KS == 3 ? (m / TW) * (TW + 2) + (m % TW) : m;
  }

  // input-channel split (a.part): workgroup z of gridDim.z takes the chunks [c0, nchunks) of its share, a whole number of
  // accumulation blocks each
  int c0 = 0, nchunks = a.Cin / GK;
  if (a.part != nullptr) {
    constexpr int Q = BLK > 0 ? BLK : 1;
    const int per = ((nchunks + (int)gridDim.z - 1) / (int)gridDim.z + Q - 1) / Q * Q;
    c0 = min((int)blockIdx.z * per, nchunks); nchunks = min(nchunks, c0 + per);
  }
  if (c0 < nchunks) issue(c0, c0 & 1);
  for (int ch = c0; ch < nchunks; ++ch) {
    __syncthreads();                 // chunk ch has landed (the barrier drains the DMA queue); nobody reads the other buffer any more
    if (ch + 1 < nchunks) issue(ch + 1, (ch + 1) & 1);
    const float* wl = lds + (ch & 1) * BUF;
    const float* xl = wl + WSLAB;
    auto ld = [&](int tap, int kk, float (&av)[TM], float (&bv)[TN]) {
      const int toff = KS == 3 ? (tap / 3) * (TW + 2) + (tap % 3) : 0;
#pragma unroll
      for (int i = 0; i < TM; ++i) av[i] = wl[(tap * GK + kk + lk) * BMC + wr + 32 * i + li];
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[j] = xl[(kk + lk) * HS + pbase[j] + toff];
    };
    auto mma = [&](const float (&av)[TM], const float (&bv)[TN]) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    };
    constexpr int SPT = GK / 2, NS = TAPS * SPT;       // channel-pair steps per tap, steps per chunk (even)
    static_assert(NS % 2 == 0, "steps are issued in pairs");
    float av0[TM], bv0[TN], av1[TM], bv1[TN];
    ld(0, 0, av0, bv0);
#pragma unroll
    for (int st = 0; st < NS; st += 2) {             // operands of the next step read before the MFMAs of this one (fenced, see above)
      ld((st + 1) / SPT, ((st + 1) % SPT) * 2, av1, bv1);
      __builtin_amdgcn_sched_barrier(0);
      mma(av0, bv0);
      __builtin_amdgcn_sched_barrier(0);
      if (st + 2 < NS) ld((st + 2) / SPT, ((st + 2) % SPT) * 2, av0, bv0);
      __builtin_amdgcn_sched_barrier(0);
      mma(av1, bv1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (BLK > 0) { if ((ch + 1 - c0) % BLK == 0 || ch + 1 == nchunks) conv_flush(acc, tot, ch + 1 == nchunks); }
  }
  __syncthreads();                   // the epilogue's reductions reuse the buffers
  conv_epilogue<BMC, EPI, THT>(a, acc, lds, b, r0, x0, y0);
}

template <int BMC, int KS, int EPI, int THT, int GK, int BLK = 0>
int launch_conv_dma(const ConvArgs& a, hipStream_t st, int ksplit = 1) {
  constexpr int TAPS = KS * KS;
  constexpr int HS = KS == 3 ? (THT + 2) * (TW + 2) : THT * TW;
  size_t smem = sizeof(float) * 2 * (size_t)(TAPS * GK * BMC + ((GK * HS + 255) / 256) * 256);
  if (a.gap_acc && smem < sizeof(float) * 4 * 64 * 33) smem = sizeof(float) * 4 * 64 * 33;      // the epilogue's row-sum transpose
  static bool raised = false;
  if (!raised) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_glds_kernel<BMC, KS, EPI, THT, GK, BLK>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    raised = true;
  }
  const int tiles = sln_cdiv(a.W, TW) * sln_cdiv(a.H, THT) * a.B;
  hipLaunchKernelGGL((conv_glds_kernel<BMC, KS, EPI, THT, GK, BLK>), dim3(tiles, a.rows_pad / BMC, ksplit), dim3(256), smem, st, a);
  SLN_CHECK_LAUNCH();
  return 0;
}

// Second half of an input-channel split: y = act(sum_z part[z] + bias), the partial sums added in z order (a fixed order: the
// result does not depend on how the workgroups were scheduled), and the optional reductions of conv_epilogue over what was written.
// One workgroup per (sample, output row); NT threads = one wavefront for planes of up to 64 pixels, else four.
template <int NT>
__global__ __launch_bounds__(NT) void conv_split_finish_kernel(const float* __restrict__ part, int S, long stride, const float* __restrict__ bias,
                                                               int act, float slope, float* __restrict__ y, int rows, int plane,
                                                               double* __restrict__ ln_acc, double* __restrict__ gap_acc) {
  const int b = blockIdx.y, row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ double red[4][2];
  const size_t base = ((size_t)b * rows + row) * plane;
  const float bs = bias ? bias[row] : 0.f;
  double l1 = 0.0, l2 = 0.0;
  for (int p = tid; p < plane; p += NT) {
    const float* src = part + base + p;
    // four independent loads at a time, added in z order
    float v = 0.f;
    int z = 0;
    for (; z + 4 <= S; z += 4) {
      const float t0 = src[(size_t)z * stride], t1 = src[(size_t)(z + 1) * stride], t2 = src[(size_t)(z + 2) * stride], t3 = src[(size_t)(z + 3) * stride];
      v += t0; v += t1; v += t2; v += t3;
    }
    for (; z < S; ++z) v += src[(size_t)z * stride];
    v += bs;
    if (act == ACT_RELU) v = fmaxf(v, 0.f);
    else if (act == ACT_LEAKY) v = v > 0.f ? v : v * slope;
    y[base + p] = v;
    l1 += (double)v; l2 = fma((double)v, (double)v, l2);
  }
  if (ln_acc == nullptr && gap_acc == nullptr) return;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { l1 += __shfl_xor(l1, off); l2 += __shfl_xor(l2, off); }
  if (NT > 64) {
    if (lane == 0) { red[wave][0] = l1; red[wave][1] = l2; }
    __syncthreads();
    l1 = red[0][0] + red[1][0] + red[2][0] + red[3][0]; l2 = red[0][1] + red[1][1] + red[2][1] + red[3][1];
  }
  if (tid == 0) {
    if (gap_acc) atomicAdd(gap_acc + (size_t)b * rows + row, l1);
    if (ln_acc) { atomicAdd(ln_acc + LN_ACC_STRIDE * b, l1); atomicAdd(ln_acc + LN_ACC_STRIDE * b + 1, l2); }
  }
}
static int launch_conv_split_finish(const ConvArgs& a, const float* part, int S, long stride, hipStream_t st) {
  const int plane = a.H * a.W;
  if (plane <= 64) hipLaunchKernelGGL(conv_split_finish_kernel<64>, dim3(a.rows, a.B), dim3(64), 0, st, part, S, stride, a.bias, a.act, a.slope, a.y, a.rows, plane, a.ln_acc, a.gap_acc);
  else hipLaunchKernelGGL(conv_split_finish_kernel<256>, dim3(a.rows, a.B), dim3(256), 0, st, part, S, stride, a.bias, a.act, a.slope, a.y, a.rows, plane, a.ln_acc, a.gap_acc);
  SLN_CHECK_LAUNCH();
  return 0;
}

// Scratch of the input-channel split.  A split launch has < 192 workgroups of 8 x 16 pixels x <= 128 rows and at most ~(512 + 192)
// partial workgroups in all: its partial sums are <= 704 x 128 x 128 floats = 46 MB whatever the layer - one slot of 48 MB serves every
// split launch of a stream (round 5 pinned 128 MB per stream, for ever).
// One slot per (device, stream), under a mutex: launches of one stream are ordered, so a stream's slot needs no further guard; two
// streams never share one (their partial sums would interleave).  At most 16 slots (768 MB): the least recently used one is freed
// for a new stream (hipFree waits for the device, so no launch still reads it); sln_spade_release frees a stream's slot (or all).
// A slot cannot be allocated while the caller captures the stream: such a launch FAILS with SLN_E_STATE (it used to fall back to
// the unsplit kernel silently, whose sums round differently - results depended on the process's history): call sln_spade_prepare on
// the stream (or run one eager forward) before capturing.
constexpr size_t CONV_PART_BYTES = (size_t)48 << 20;
constexpr size_t CONV_PART_SLOTS = 16;
struct ConvPartSlot { int dev; hipStream_t st; float* p; uint64_t used; };
static std::mutex g_conv_part_mu;
static std::vector<ConvPartSlot> g_conv_parts;
static uint64_t g_conv_part_clock = 0;
static int conv_part_scratch(hipStream_t st, float** out) {
  *out = nullptr;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return SLN_E_STATE;
  std::lock_guard<std::mutex> lk(g_conv_part_mu);
  for (ConvPartSlot& e : g_conv_parts) if (e.dev == dev && e.st == st) { e.used = ++g_conv_part_clock; *out = e.p; return 0; }
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return SLN_E_STATE;
  if (g_conv_parts.size() >= CONV_PART_SLOTS) {
    size_t lru = 0;
    for (size_t i = 1; i < g_conv_parts.size(); ++i) if (g_conv_parts[i].used < g_conv_parts[lru].used) lru = i;
    int cur = dev;
    if (g_conv_parts[lru].dev != cur) (void)hipSetDevice(g_conv_parts[lru].dev);
    (void)hipFree(g_conv_parts[lru].p);
    if (g_conv_parts[lru].dev != cur) (void)hipSetDevice(cur);
    g_conv_parts.erase(g_conv_parts.begin() + (long)lru);
  }
  float* p = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&p), CONV_PART_BYTES) != hipSuccess) { (void)hipGetLastError(); return SLN_E_NOMEM; }
  g_conv_parts.push_back(ConvPartSlot{dev, st, p, ++g_conv_part_clock});
  *out = p;
  return 0;
}
static int conv_part_release(hipStream_t st, bool all) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return SLN_E_STATE;
  std::lock_guard<std::mutex> lk(g_conv_part_mu);
  int n = 0;
  for (size_t i = 0; i < g_conv_parts.size();) {
    ConvPartSlot& e = g_conv_parts[i];
    if (all || (e.dev == dev && e.st == st)) {
      if (e.dev != dev) (void)hipSetDevice(e.dev);
      (void)hipFree(e.p);
      if (e.dev != dev) (void)hipSetDevice(dev);
      g_conv_parts.erase(g_conv_parts.begin() + (long)i); ++n;
    } else ++i;
  }
  return n;
}

// Batch-1 calls (test_SPADE_shade.py:77-79: one call per z).  The 1 024 / 512-channel layers at 8 x 8 .. 64 x 64 pixels are 8 .. 64
// workgroups that each walk all input channels: 375-760 us per launch on 3-25 % of the CUs - 6 of the 7.7 ms of a call - and the
// 256 / 128-channel layers behind them 64-128 workgroups.  Launches of fewer than 192 workgroups split the input channels over
// gridDim.z workgroups (>= 8 chunks = 32 channels each, ~512 workgroups in all: the 8 x 16-pixel DMA kernel, BLK = accumulation
// block of the blocked form or 0) and conv_split_finish_kernel adds the partial sums in order.  *done: the launch was taken.
template <int BMC, int KS, int BLK>
int launch_conv_split(const ConvArgs& a, hipStream_t st, bool* done) {
  static const int split_max = getenv("SLN_CONV_KSPLIT") ? atoi(getenv("SLN_CONV_KSPLIT")) : 32;      // 1: never
  *done = false;
  if (split_max <= 1 || a.Cin % 4 != 0) return 0;
  const long blocks = (long)sln_cdiv(a.W, TW) * sln_cdiv(a.H, TH) * a.B * (a.rows_pad / BMC);
  static const int few = getenv("SLN_CONV_KSPLIT_BLOCKS") ? atoi(getenv("SLN_CONV_KSPLIT_BLOCKS")) : 192;      // lab
  static const int target = getenv("SLN_CONV_KSPLIT_TARGET") ? atoi(getenv("SLN_CONV_KSPLIT_TARGET")) : 512;
  if (blocks >= few) return 0;
  const int nch = a.Cin / 4;
  const size_t one = (size_t)a.B * a.rows * a.H * a.W;
  int S = (int)std::min<long>(std::min(split_max, nch / 8), (target + blocks - 1) / blocks);
  S = (int)std::min<size_t>((size_t)S, CONV_PART_BYTES / (one * sizeof(float)));
  if (S <= 1) return 0;
  float* part = nullptr;
  *done = true;                                            // from here on the launch is the split one - or an error, never another kernel
  { const int r = conv_part_scratch(st, &part); if (r) return r; }
  ConvArgs p = a; p.part = part; p.part_stride = (long)one;
  const int r = launch_conv_dma<BMC, KS, CEPI_BIAS_ACT, TH, 4, BLK>(p, st, S);
  return r ? r : launch_conv_split_finish(a, part, S, (long)one, st);
}

// Blocked accumulation (see conv_mfma_kernel): variants with 64 accumulators per lane.  Blocks of 16 input channels (144
// products): 2 buffers of 8 channels / 4 buffers of 4 / 2 staged chunks of 8.
template <int BMC>
int launch_conv_blocked(const ConvArgs& a, hipStream_t st) {
  static const int variant = getenv("SLN_CONV_BLOCK_VARIANT") ? atoi(getenv("SLN_CONV_BLOCK_VARIANT")) : 0;   // lab: 1 = 8 x 16 x BMC rows, 2 = staged
  static const bool staged_only = getenv("SLN_CONV_STAGED") != nullptr;      // the A/B switch of launch_conv covers the Cin >= 512 convolutions too
  if (a.Cin % 8 == 0 && variant != 2 && !staged_only) {
    // 64-row workgroups of 16 x 16 pixels (the halo is DMA'd once per 64 rows instead of once per 128: 1.2x the operand traffic
    // of the 128-row workgroups; 8 x 16 x 128 rows re-reads the weights per 128 pixels: 1.8x)
    const long tall_blocks = (long)sln_cdiv(a.W, TW) * sln_cdiv(a.H, 16) * a.B * (a.rows_pad / 64);
    if (variant == 0 && a.H >= 16 && tall_blocks >= 512) return launch_conv_dma<64, 3, CEPI_BIAS_ACT, 16, 8, 2>(a, st);
    { bool done = false; const int r = launch_conv_split<BMC, 3, 4>(a, st, &done); if (done) return r; }
    return launch_conv_dma<BMC, 3, CEPI_BIAS_ACT, TH, 4, 4>(a, st);
  }
  constexpr int HS = HALO;
  size_t smem = sizeof(float) * (size_t)(9 * CK * BMC + CK * HS);
  if (a.gap_acc && smem < sizeof(float) * 4 * 64 * 33) smem = sizeof(float) * 4 * 64 * 33;
  static bool raised = false;
  if (!raised) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<BMC, 3, CEPI_BIAS_ACT, 2>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    raised = true;
  }
  const int tiles = sln_cdiv(a.W, TW) * sln_cdiv(a.H, TH) * a.B;
  hipLaunchKernelGGL((conv_mfma_kernel<BMC, 3, CEPI_BIAS_ACT, 2>), dim3(tiles, a.rows_pad / BMC), dim3(256), smem, st, a);
  SLN_CHECK_LAUNCH();
  return 0;
}

template <int BMC, int KS, int EPI>
int launch_conv(const ConvArgs& a, hipStream_t st) {
  constexpr int TAPS = KS * KS;
  constexpr int HS = KS == 3 ? HALO : TH * TW;
  static const bool staged_only = getenv("SLN_CONV_STAGED") != nullptr;      // A/B runs and tests
  // measured per shape (tools/lab/conv_lab.py, same box): with 8 x 16 pixels per workgroup the DMA kernel wins 2 % on the modulation
  // convs and loses 2-6 % on the bias/activation convs (twice the barriers of the staged kernel); with 16 x 16 pixels it wins on
  // every 128-row conv (modulation 9.8 -> 9.15 ms, 1 024 -> 512 channels at 32 x 32 2.34 -> 2.26 ms)
  static const bool dma_all = getenv("SLN_CONV_DMA") != nullptr;
  static const bool tall = getenv("SLN_CONV_NO_TALL") == nullptr;
  if constexpr (KS == 3 && EPI == CEPI_BIAS_ACT) { if (a.blocked) return launch_conv_blocked<BMC>(a, st); }
  if constexpr (EPI == CEPI_BIAS_ACT) {
    if (!staged_only) { bool done = false; const int r = launch_conv_split<BMC, KS, 0>(a, st, &done); if (done) return r; }
  }
  if (a.Cin % 8 == 0 && !staged_only) {
    // 16 x 16 pixels per workgroup where the image has the rows (half the weight DMA per MFMA, twice the MFMAs per barrier)
    // (not for launches too small to give every CU two of the tall workgroups: batch-1 convs of the one-map-many-z path)
    const long tall_blocks = (long)sln_cdiv(a.W, TW) * sln_cdiv(a.H, 16) * a.B * (a.rows_pad / BMC);
    if (BMC == 128 && KS == 3 && tall && a.H >= 16 && tall_blocks >= 512)
      return launch_conv_dma<BMC, KS, EPI, (BMC == 128 && KS == 3 ? 16 : TH), 4>(a, st);
    // 64-row blocks: buffers of 8 channels (the same 144 MFMAs per wave and barrier; 2 % over the staged kernel at 256 x 256).
    // 8-channel buffers for the 128-row blocks leave room for one workgroup per CU only: 10.1 ms against 9.06 ms.
    if (BMC == 64 && KS == 3 && tall && a.H >= 16 && tall_blocks >= 512)
      return launch_conv_dma<BMC, KS, EPI, (BMC == 64 && KS == 3 ? 16 : TH), 8>(a, st);
    if (EPI == CEPI_MODULATE || dma_all) return launch_conv_dma<BMC, KS, EPI, TH, 4>(a, st);
  }
  size_t smem = sizeof(float) * (size_t)(TAPS * CK * BMC + CK * HS);
  if (a.gap_acc && smem < sizeof(float) * 4 * 64 * 33) smem = sizeof(float) * 4 * 64 * 33;      // the epilogue's row-sum transpose
  static bool raised = false;
  if (!raised) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<BMC, KS, EPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    raised = true;
  }
  const int tiles = sln_cdiv(a.W, TW) * sln_cdiv(a.H, TH) * a.B;
  hipLaunchKernelGGL((conv_mfma_kernel<BMC, KS, EPI>), dim3(tiles, a.rows_pad / BMC), dim3(256), smem, st, a);
  SLN_CHECK_LAUNCH();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------------
// det (deterministic mode): block k of a sample STORES its sums to slot pair k + 1 of the sample's line (no atomics, at most
// LN_ACC_STRIDE / 2 - 1 = 7 blocks per sample); ln_finalize_kernel adds the slots in order and leaves the total in slot 0.
__global__ void ln_stats_kernel(const float* __restrict__ x, long n, double* __restrict__ acc, int det) {
  const int b = blockIdx.y;
  const float* xb = x + (size_t)b * n;
  double s = 0.0, q = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const double v = xb[i];
    s += v; q += v * v;
  }
  __shared__ double rs[256], rq[256];
  rs[threadIdx.x] = s; rq[threadIdx.x] = q;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) { rs[threadIdx.x] += rs[threadIdx.x + off]; rq[threadIdx.x] += rq[threadIdx.x + off]; }
    __syncthreads();
  }
  // one 128-byte line per sample: device atomics to the same line are serialised on the memory side
  if (threadIdx.x == 0) {
    if (det) { acc[LN_ACC_STRIDE * b + 2 * (blockIdx.x + 1)] = rs[0]; acc[LN_ACC_STRIDE * b + 2 * (blockIdx.x + 1) + 1] = rq[0]; }
    else { atomicAdd(acc + LN_ACC_STRIDE * b, rs[0]); atomicAdd(acc + LN_ACC_STRIDE * b + 1, rq[0]); }
  }
}
// rep: every accumulated value stands for `rep` elements of the normalised tensor (4 when the tensor is the nearest x2
// upsampling of what was summed: same mean, n -> 4 n in the unbiased variance)
// nslots > 0: the sums are the slot pairs 1 .. nslots of the sample's line, added here in order (and left in slot 0)
__global__ void ln_finalize_kernel(double* __restrict__ acc, long n_acc, int rep, int B, float eps, float* __restrict__ stats, int nslots) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (nslots > 0) {
    double s = 0.0, q = 0.0;
    for (int k = 1; k <= nslots; ++k) { s += acc[LN_ACC_STRIDE * b + 2 * k]; q += acc[LN_ACC_STRIDE * b + 2 * k + 1]; }
    acc[LN_ACC_STRIDE * b] = s; acc[LN_ACC_STRIDE * b + 1] = q;
  }
  const double n = (double)n_acc * rep;
  const double mean = acc[LN_ACC_STRIDE * b] / (double)n_acc;
  double var = ((double)rep * acc[LN_ACC_STRIDE * b + 1] - n * mean * mean) / (n - 1.0);     // unbiased (torch.std default)
  var = var < 0.0 ? 0.0 : var;
  stats[2 * b] = (float)mean;
  stats[2 * b + 1] = 1.0f / ((float)sqrt(var) + eps);                           // LayerNorm2D adds eps to sigma
}

// F.interpolate(seg, size) : mode 0 nearest (src = floor(dst * in/out)), mode 1 bilinear align_corners=False
__global__ void resize_kernel(const float* __restrict__ src, int C, int Hi, int Wi, int Ho, int Wo, int mode, long n,
                              float* __restrict__ dst) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int xo = (int)(i % Wo), yo = (int)((i / Wo) % Ho);
  const long bc = i / ((long)Wo * Ho);
  const float* s = src + bc * (long)Hi * Wi;
  if (mode == 0) {
    const int ys = min((int)floorf(yo * ((float)Hi / Ho)), Hi - 1), xs = min((int)floorf(xo * ((float)Wi / Wo)), Wi - 1);
    dst[i] = s[(long)ys * Wi + xs];
    return;
  }
  const float sy = (float)Hi / Ho, sx = (float)Wi / Wo;
  float fy = (yo + 0.5f) * sy - 0.5f, fx = (xo + 0.5f) * sx - 0.5f;
  fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
  const int ya = min((int)fy, Hi - 1), xa = min((int)fx, Wi - 1);
  const int yb = min(ya + 1, Hi - 1), xb = min(xa + 1, Wi - 1);
  const float ly = fy - ya, lx = fx - xa;
  dst[i] = (1.f - ly) * ((1.f - lx) * s[(long)ya * Wi + xa] + lx * s[(long)ya * Wi + xb]) +
           ly * ((1.f - lx) * s[(long)yb * Wi + xa] + lx * s[(long)yb * Wi + xb]);
  (void)C;
}

// [leaky_0.01(conv3x3_reflect(seg[:,0], 1->nd)) | seg[:,1:]]  ->  out [B, nd + Cs - 1, H, W]   (SPADE4 :1445-1446)
// cw = channels written per sample: nd + Cs - 1 (everything) or nd (the mask channels of `out` were filled once for this
// resolution and only the depth features change from one SPADE layer to the next)
// grid = (pixel blocks, channel, sample): the flat-index form of round 1 spent its time in three 64-bit divisions per element
// (33 MB in 37 us per launch, 18 launches per forward)
__global__ __launch_bounds__(256) void depth_concat_kernel(const float* __restrict__ seg, int Cs, int H, int W, const float* __restrict__ wpd,
                                                           const float* __restrict__ bpd, int nd, float* __restrict__ out) {
  const int p = blockIdx.x * 256 + threadIdx.x;            // pixel of the plane
  const int plane = W * H;
  if (p >= plane) return;
  const int c = blockIdx.y, b = blockIdx.z;
  const int y = p / W, x = p - y * W;
  const int Co = nd + Cs - 1;
  const float* sb = seg + (size_t)b * Cs * plane;
  float* o = out + ((size_t)b * Co + c) * plane + p;
  if (c >= nd) { *o = sb[(size_t)(c - nd + 1) * plane + p]; return; }
  float v = bpd[c];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
      v = fmaf(wpd[c * 9 + ky * 3 + kx], sb[reflect_idx(y + ky - 1, H) * W + reflect_idx(x + kx - 1, W)], v);
  *o = v > 0.f ? v : 0.01f * v;
}

__global__ void gap_kernel(const float* __restrict__ x, long hw, float* __restrict__ out) {   // one block per (b, c)
  const float* p = x + (size_t)blockIdx.x * hw;
  double s = 0.0;                           // fp64: the pool feeds the squeeze-excite FCs, see se_fc_kernel
  for (long i = threadIdx.x; i < hw; i += blockDim.x) s += (double)p[i];
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) { if (threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off]; __syncthreads(); }
  if (threadIdx.x == 0) out[blockIdx.x] = (float)(red[0] / (double)hw);
}
// scale[b, :] = sigmoid(W2 relu(W0 gap[b, :]))   (SEBlock2, :70-85; reduction 8)
// gap: the averages, or (gsum != nullptr) the fp64 pixel sums a conv epilogue accumulated, divided here by hw.
// Both dot products and the logistic are evaluated in fp64 (round 4).  With the reference's default initialisation the
// spectral-normalised convolutions produce feature maps of magnitude ~1e2, so the logit v = W2 relu(W0 gap) is a sum of ~1e3
// terms of magnitude 1e1..1e2: an fp32 chain leaves an ABSOLUTE error of ~1e-4 in v, and the logistic turns it into a RELATIVE
// error of up to 2.5e-5 in the scale of a whole channel plane - coherent over the plane, so the next convolutions add it up
// instead of averaging it out.  tools/spade_error_budget.py: with this (and the blocked accumulation of the convolutions) the
// full-size image is 5e-5 from an fp64 evaluation; with fp32 chains here it was 3e-4 whatever the convolutions did (torch's CPU
// path, whose blocked dot products are short chains, 1.1e-4).  2 x C x C/8 fp64 FMAs per sample: nothing.
// Both FCs read their weight rows with 16-byte loads that are contiguous across the lanes that share a row (a whole wavefront per
// hidden row, 8 lanes per output row) and keep SEVERAL rows in flight (8 hidden rows / 4 output rows per lane group: their loads
// and their shuffle trees are independent chains) - one block per sample is all the parallelism there is, so the kernel is a
// latency chain per row otherwise.
// phase 0: the whole block in one workgroup per sample.  Few samples (a batch-1 call: 35 us per launch, 0.25 of the 2 ms of a call):
// phase 1 = the hidden rows 32 blockIdx.y .. + 31 to `hid` [B, Cr] (fp64, global), phase 2 = the output rows 128 blockIdx.y .. + 127
// from it - two launches of Cr / 32 and C / 128 workgroups per sample.  Same arithmetic per row in every phase: identical results.
__global__ __launch_bounds__(256) void se_fc_kernel(const float* __restrict__ gap, const double* __restrict__ gsum, double hw,
                                                    const float* __restrict__ w0, const float* __restrict__ w2, int C, int Cr,
                                                    float* __restrict__ scale, int phase, double* __restrict__ hid,
                                                    double* __restrict__ zero_acc, int n_zero) {
  extern __shared__ double smd[];
  double* g = smd; double* hdn = smd + C;
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // the LayerNorm accumulator the tail kernel (next launch) adds into starts at zero: cleared here instead of by a launch of its own
  if (zero_acc != nullptr && blockIdx.x == 0 && blockIdx.y == 0) for (int i = threadIdx.x; i < n_zero; i += blockDim.x) zero_acc[i] = 0.0;
  if (phase != 2) for (int c = threadIdx.x; c < C; c += blockDim.x) g[c] = gsum ? gsum[(size_t)b * C + c] / hw : (double)gap[(size_t)b * C + c];
  else for (int r = threadIdx.x; r < Cr; r += blockDim.x) hdn[r] = hid[(size_t)b * Cr + r];
  __syncthreads();
  // hidden = relu(W0 g): a wavefront per row, lanes take 4 consecutive columns per step (C % 4 == 0: the callers require C % 8 == 0)
  constexpr int RB = 8;
  const int r_first = phase == 1 ? 4 * RB * (int)blockIdx.y + wave * RB : wave * RB, r_end = phase == 1 ? min(Cr, 4 * RB * ((int)blockIdx.y + 1)) : Cr;
  if (phase != 2)
  for (int r0 = r_first; r0 < r_end; r0 += 4 * RB) {
    double s[RB];
#pragma unroll
    for (int k = 0; k < RB; ++k) s[k] = 0.0;
    for (int c = 4 * lane; c < C; c += 256) {
      float4 w[RB];
#pragma unroll
      for (int k = 0; k < RB; ++k) w[k] = *reinterpret_cast<const float4*>(w0 + (size_t)min(r0 + k, Cr - 1) * C + c);
      const double g0 = g[c], g1 = g[c + 1], g2 = g[c + 2], g3 = g[c + 3];
#pragma unroll
      for (int k = 0; k < RB; ++k) { s[k] = fma((double)w[k].x, g0, s[k]); s[k] = fma((double)w[k].y, g1, s[k]); s[k] = fma((double)w[k].z, g2, s[k]); s[k] = fma((double)w[k].w, g3, s[k]); }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int k = 0; k < RB; ++k) s[k] += __shfl_xor(s[k], off);
    if (lane < RB && r0 + lane < Cr) {
      double v = s[0];
#pragma unroll
      for (int k = 1; k < RB; ++k) v = lane == k ? s[k] : v;
      hdn[r0 + lane] = v > 0.0 ? v : 0.0;
      if (phase == 1) hid[(size_t)b * Cr + r0 + lane] = v > 0.0 ? v : 0.0;
    }
  }
  if (phase == 1) return;
  __syncthreads();
  // scale = sigmoid(W2 hidden): 8 lanes per output row, each a contiguous eighth of the row when that is whole float4s
  const int sub = threadIdx.x & 7, grp = threadIdx.x >> 3;
  const bool vec = (Cr & 31) == 0;
  const int seg = Cr >> 3;
  constexpr int OB = 4;
  const int c_first = phase == 2 ? 32 * OB * (int)blockIdx.y + grp * OB : grp * OB, c_end = phase == 2 ? min(C, 32 * OB * ((int)blockIdx.y + 1)) : C;
  for (int c0 = c_first; c0 < c_end; c0 += 32 * OB) {
    double s[OB];
#pragma unroll
    for (int k = 0; k < OB; ++k) s[k] = 0.0;
    if (vec) {
      for (int q = 0; q < seg; q += 4) {
        const int r = sub * seg + q;
        float4 w[OB];
#pragma unroll
        for (int k = 0; k < OB; ++k) w[k] = *reinterpret_cast<const float4*>(w2 + (size_t)min(c0 + k, C - 1) * Cr + r);
        const double h0 = hdn[r], h1 = hdn[r + 1], h2 = hdn[r + 2], h3 = hdn[r + 3];
#pragma unroll
        for (int k = 0; k < OB; ++k) { s[k] = fma((double)w[k].x, h0, s[k]); s[k] = fma((double)w[k].y, h1, s[k]); s[k] = fma((double)w[k].z, h2, s[k]); s[k] = fma((double)w[k].w, h3, s[k]); }
      }
    } else {
      for (int r = sub; r < Cr; r += 8)
#pragma unroll
        for (int k = 0; k < OB; ++k) s[k] = fma((double)w2[(size_t)min(c0 + k, C - 1) * Cr + r], hdn[r], s[k]);
    }
#pragma unroll
    for (int off = 1; off < 8; off <<= 1)
#pragma unroll
      for (int k = 0; k < OB; ++k) s[k] += __shfl_xor(s[k], off);
    if (sub < OB && c0 + sub < C) {
      double v = s[0];
#pragma unroll
      for (int k = 1; k < OB; ++k) v = sub == k ? s[k] : v;
      scale[(size_t)b * C + c0 + sub] = (float)(1.0 / (1.0 + exp(-v)));
    }
  }
}
__global__ void se_scale_add_kernel(const float* __restrict__ xs, const float* __restrict__ dx, const float* __restrict__ scale,
                                    long hw, long n, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = xs[i] + dx[i] * scale[i / hw];
}

// nn.Upsample(scale_factor=2): mode 0 nearest, mode 1 bilinear (align_corners=False)
__global__ void upsample2x_kernel(const float* __restrict__ x, int H, int W, int mode, long n, float* __restrict__ y) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int Wo = 2 * W, Ho = 2 * H;
  const int xo = (int)(i % Wo), yo = (int)((i / Wo) % Ho);
  const float* s = x + (i / ((long)Wo * Ho)) * (long)H * W;
  if (mode == 0) { y[i] = s[(long)(yo >> 1) * W + (xo >> 1)]; return; }
  float fy = (yo + 0.5f) * 0.5f - 0.5f, fx = (xo + 0.5f) * 0.5f - 0.5f;
  fy = fy < 0.f ? 0.f : fy; fx = fx < 0.f ? 0.f : fx;
  const int ya = min((int)fy, H - 1), xa = min((int)fx, W - 1), yb = min(ya + 1, H - 1), xb = min(xa + 1, W - 1);
  const float ly = fy - ya, lx = fx - xa;
  y[i] = (1.f - ly) * ((1.f - lx) * s[(long)ya * W + xa] + lx * s[(long)ya * W + xb]) +
         ly * ((1.f - lx) * s[(long)yb * W + xa] + lx * s[(long)yb * W + xb]);
}

// Tail of a SPADEResnetBlock4 (:1492-1493) and the nn.Upsample that follows it (:1585-1600) in one pass:
//   v = xs + dx * scale[b, c];   out = v | nearest x2 (v) | bilinear x2 (v);   acc[b] += (sum out, sum out^2)
// The residual sum is written once, at the resolution the next block reads, together with the LayerNorm2D sums of the next
// block (separately: scale-add R2 W1, upsample R1 W4, statistics R4 units of HBM traffic; here R2 W4 - or R2 W1 when the
// consumers read the result through the nearest upsampling themselves and only the sums are needed at the higher resolution).
// UP: 0 none, 1 nearest, 2 bilinear (align_corners=False).  xs_up: xs is [B,C,H/2,W/2], read through nearest x2 (the identity
// shortcut of a block whose input was never materialised).  Four consecutive outputs of a row per thread.
template <int UP>
__global__ __launch_bounds__(256) void block_tail_kernel(const float* __restrict__ xs, const float* __restrict__ dx,
                                                         const float* __restrict__ scale, int C, int H, int W, int xs_up,
                                                         float* __restrict__ out, double* __restrict__ acc) {
  const int b = blockIdx.y;
  const int Ho = UP ? 2 * H : H, Wo = UP ? 2 * W : W;
  const long plane = (long)H * W, oplane = (long)Ho * Wo;
  const long n4 = (long)C * oplane / 4;
  const float* dxb = dx + (size_t)b * C * plane;
  const float* xsb = xs + (size_t)b * C * (xs_up ? plane / 4 : plane);
  float* ob = out + (size_t)b * C * oplane;
  const int w4 = Wo / 4;
  double s = 0.0, q = 0.0;
  if (UP == 2) {
    // bilinear x2: a thread writes the output rows 2 p - 1 and 2 p (both interpolate the source rows p - 1 and p) x 4 columns:
    // 8 source values for 8 outputs (one output row per thread read 8 for 4 - the kernel moved 2.1 GB in 0.78 ms at up_2)
    const long npair = (long)C * (H + 1) * w4;
    for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < npair; g += (long)gridDim.x * 256) {
      const int xo = (int)(g % w4) * 4;
      const long t = g / w4;
      const int p = (int)(t % (H + 1)), c = (int)(t / (H + 1));
      const float sc = scale[(size_t)b * C + c];
      const float* dp = dxb + (size_t)c * plane;
      const float* xp = xsb + (size_t)c * (xs_up ? plane / 4 : plane);
      auto val = [&](int y, int x) -> float {
        const float xv = xs_up ? xp[(long)(y >> 1) * (W >> 1) + (x >> 1)] : xp[(long)y * W + x];
        return xv + dp[(long)y * W + x] * sc;
      };
      const int k2 = xo >> 1;
      const int cx[4] = {max(k2 - 1, 0), k2, k2 + 1, min(k2 + 2, W - 1)};
      const int ra_ = max(p - 1, 0), rb_ = min(p, H - 1);
      float ra[4], rb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { ra[i] = val(ra_, cx[i]); rb[i] = val(rb_, cx[i]); }
      const float lx0 = k2 == 0 ? 0.f : 0.75f;
      const int i0 = k2 == 0 ? 1 : 0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int yo = 2 * p - 1 + h;
        if (yo < 0 || yo >= Ho) continue;
        // the row weights of the one-row form, evaluated the same way (yo = 0: weight 0 on the second row, whichever it is)
        float fy = (yo + 0.5f) * 0.5f - 0.5f;
        fy = fy < 0.f ? 0.f : fy;
        const int ya = min((int)fy, H - 1);
        const float ly = fy - ya;
        float4 v;
        v.x = (1.f - ly) * ((1.f - lx0) * ra[i0] + lx0 * ra[i0 + 1]) + ly * ((1.f - lx0) * rb[i0] + lx0 * rb[i0 + 1]);
        v.y = (1.f - ly) * (0.75f * ra[1] + 0.25f * ra[2]) + ly * (0.75f * rb[1] + 0.25f * rb[2]);
        v.z = (1.f - ly) * (0.25f * ra[1] + 0.75f * ra[2]) + ly * (0.25f * rb[1] + 0.75f * rb[2]);
        v.w = (1.f - ly) * (0.75f * ra[2] + 0.25f * ra[3]) + ly * (0.75f * rb[2] + 0.25f * rb[3]);
        *reinterpret_cast<float4*>(ob + (size_t)c * oplane + (long)yo * Wo + xo) = v;
        s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
        q += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
      }
    }
  } else {
  const bool small = n4 < (1L << 31);          // 32-bit index arithmetic (the 64-bit divisions below were most of the kernel's instructions)
  for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < n4; g += (long)gridDim.x * 256) {
    int xo, yo, c;
    if (small) {
      const unsigned gu = (unsigned)g, tu = gu / (unsigned)w4;
      xo = (int)(gu - tu * (unsigned)w4) * 4;
      c = (int)(tu / (unsigned)Ho); yo = (int)(tu - (unsigned)c * (unsigned)Ho);
    } else {
      xo = (int)(g % w4) * 4;
      const long t = g / w4;
      yo = (int)(t % Ho); c = (int)(t / Ho);
    }
    const float sc = scale[(size_t)b * C + c];
    const float* dp = dxb + (size_t)c * plane;
    const float* xp = xsb + (size_t)c * (xs_up ? plane / 4 : plane);
    auto val = [&](int y, int x) -> float {           // v at source pixel (y, x)
      const float xv = xs_up ? xp[(long)(y >> 1) * (W >> 1) + (x >> 1)] : xp[(long)y * W + x];
      return xv + dp[(long)y * W + x] * sc;
    };
    float4 v;
    if (UP == 0) {
      if (!xs_up) {
        const float4 xv = *reinterpret_cast<const float4*>(xp + (long)yo * W + xo);
        const float4 dv = *reinterpret_cast<const float4*>(dp + (long)yo * W + xo);
        v.x = xv.x + dv.x * sc; v.y = xv.y + dv.y * sc; v.z = xv.z + dv.z * sc; v.w = xv.w + dv.w * sc;
      } else {
        v.x = val(yo, xo); v.y = val(yo, xo + 1); v.z = val(yo, xo + 2); v.w = val(yo, xo + 3);
      }
    } else if (UP == 1) {
      const float a0 = val(yo >> 1, xo >> 1), a1 = val(yo >> 1, (xo >> 1) + 1);
      v.x = a0; v.y = a0; v.z = a1; v.w = a1;
    } else {
      float fy = (yo + 0.5f) * 0.5f - 0.5f;
      fy = fy < 0.f ? 0.f : fy;
      const int ya = min((int)fy, H - 1), yb = min(ya + 1, H - 1);
      const float ly = fy - ya;
      const int k2 = xo >> 1;                                       // source columns k2-1 .. k2+2 (clamped)
      const int cx[4] = {max(k2 - 1, 0), k2, k2 + 1, min(k2 + 2, W - 1)};
      float ra[4], rb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { ra[i] = val(ya, cx[i]); rb[i] = val(yb, cx[i]); }
      // output xo + t: fx = k2 - 0.25 + 0.5 t  ->  (xa, lx) = (k2-1, .75), (k2, .25), (k2, .75), (k2+1, .25); fx < 0 clamps to 0
      const float lx0 = k2 == 0 ? 0.f : 0.75f;
      const int i0 = k2 == 0 ? 1 : 0;                                // xa = 0, xb = 1 at the left border
      v.x = (1.f - ly) * ((1.f - lx0) * ra[i0] + lx0 * ra[i0 + 1]) + ly * ((1.f - lx0) * rb[i0] + lx0 * rb[i0 + 1]);
      v.y = (1.f - ly) * (0.75f * ra[1] + 0.25f * ra[2]) + ly * (0.75f * rb[1] + 0.25f * rb[2]);
      v.z = (1.f - ly) * (0.25f * ra[1] + 0.75f * ra[2]) + ly * (0.25f * rb[1] + 0.75f * rb[2]);
      v.w = (1.f - ly) * (0.75f * ra[2] + 0.25f * ra[3]) + ly * (0.75f * rb[2] + 0.25f * rb[3]);
    }
    *reinterpret_cast<float4*>(ob + (size_t)c * oplane + (long)yo * Wo + xo) = v;
    s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
    q += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  }
  if (!acc) return;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s += __shfl_xor(s, off); q += __shfl_xor(q, off); }
  __shared__ double red[8];
  if ((threadIdx.x & 63) == 0) { red[2 * (threadIdx.x >> 6)] = s; red[2 * (threadIdx.x >> 6) + 1] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(acc + LN_ACC_STRIDE * b, red[0] + red[2] + red[4] + red[6]);
    atomicAdd(acc + LN_ACC_STRIDE * b + 1, red[1] + red[3] + red[5] + red[7]);
  }
}

// tanh(conv5x5_zero_pad(leaky_0.2(x)))  (:1602-1603).  Cout is tiny (3): VALU FMAs.  A workgroup owns a 16x16 output
// tile; per chunk of 8 input channels the zero-padded, LeakyReLU'ed (16+4)^2 halo goes to LDS once and is read 25x;
// the weights are addressed with loop counters only, so hipcc fetches them through the scalar cache.
// (Round 4, measured and dropped: a 16 x 64 tile with four pixels per thread - two ds_read_b128 of the halo per (channel, tap row)
// instead of 25 ds_read_b32 per channel - with the weights through the scalar cache, 772 us per batch of 32, and with the
// weights as LDS broadcasts, 834 us, against 626 us for this one-pixel-per-thread form: its eight small workgroups per CU hide
// the halo fill and the scalar loads better than three large ones.)
constexpr int IT = 16, IH = IT + 4, ICK = 8;
template <int COUT>
__global__ __launch_bounds__(256) void conv_img_kernel(const float* __restrict__ x, int Cin, int H, int W,
                                                       const float* __restrict__ w, const float* __restrict__ bias,
                                                       float* __restrict__ y) {
  __shared__ float halo[ICK][IH * IH];
  const int tiles_x = (W + IT - 1) / IT;
  const int tx0 = (blockIdx.x % tiles_x) * IT, ty0 = (blockIdx.x / tiles_x) * IT, b = blockIdx.y;
  const int lx = threadIdx.x & (IT - 1), ly = threadIdx.x >> 4;
  const long plane = (long)H * W;
  float acc[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
  // Where a thread's NF halo elements of a chunk come from does not depend on the chunk: the offsets (relative to the chunk's first
  // channel plane, -1 outside the image) are worked out once.  (Round 5: the two divisions and the bounds tests per element, every
  // chunk, were ~500 of the ~1 300 instructions a thread issues per chunk - the 25 LDS reads and 75 FMAs of a channel are 800 per
  // chunk.)  The element's LDS slot is its running index e.
  constexpr int NF = (ICK * IH * IH + 255) / 256;
  const bool fast = Cin % ICK == 0 && (long)ICK * plane < (1L << 31);
  int goff[NF];
  if (fast) {
#pragma unroll
    for (int k = 0; k < NF; ++k) {
      const int e = threadIdx.x + 256 * k;
      const int c = e / (IH * IH), p = e % (IH * IH);
      const int sy = ty0 + p / IH - 2, sx = tx0 + p % IH - 2;
      goff[k] = (e < ICK * IH * IH && sy >= 0 && sy < H && sx >= 0 && sx < W) ? c * (int)plane + sy * W + sx : -1;
    }
  }
  float* hflat = &halo[0][0];
  for (int c0 = 0; c0 < Cin; c0 += ICK) {
    if (fast) {
      const float* xc = x + ((size_t)b * Cin + c0) * plane;
      float v[NF];
#pragma unroll
      for (int k = 0; k < NF; ++k) v[k] = goff[k] >= 0 ? xc[goff[k]] : 0.f;
#pragma unroll
      for (int k = 0; k < NF; ++k) {
        const int e = threadIdx.x + 256 * k;
        if (e < ICK * IH * IH) hflat[e] = v[k] > 0.f ? v[k] : 0.2f * v[k];
      }
    } else
    for (int e = threadIdx.x; e < ICK * IH * IH; e += 256) {
      const int c = e / (IH * IH), p = e % (IH * IH);
      const int sy = ty0 + p / IH - 2, sx = tx0 + p % IH - 2;
      float v = 0.f;
      if (c0 + c < Cin && sy >= 0 && sy < H && sx >= 0 && sx < W) {
        v = x[((size_t)b * Cin + c0 + c) * plane + (long)sy * W + sx];
        v = v > 0.f ? v : 0.2f * v;
      }
      halo[c][p] = v;
    }
    __syncthreads();
    const int cn = min(ICK, Cin - c0);
    // Three-level sum (round 3): 25 taps of a channel -> the ICK channels of a chunk -> the chunks.  One serial fp32 chain of
    // Cin * 25 = 1 600 cancelling products per output (what this loop used to be) left the full-size image 3.0e-4 of its scale away
    // from an fp64 evaluation, three times the distance of the CPU fp32 path (whose blocked sums are short chains too); with chains
    // of 25 / ICK / (Cin / ICK) terms the rounding error no longer grows with the product count.  Three adds per channel more.
    float chunk[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) chunk[co] = 0.f;
    for (int c = 0; c < cn; ++c) {
      const float* wc = w + (size_t)(c0 + c) * 25;          // + co * Cin * 25 below: uniform -> scalar loads
      float part[COUT];
#pragma unroll
      for (int co = 0; co < COUT; ++co) part[co] = 0.f;
#pragma unroll
      for (int ky = 0; ky < 5; ++ky)
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) {
          const float v = halo[c][(ly + ky) * IH + lx + kx];
#pragma unroll
          for (int co = 0; co < COUT; ++co) part[co] = fmaf(wc[(size_t)co * Cin * 25 + ky * 5 + kx], v, part[co]);
        }
#pragma unroll
      for (int co = 0; co < COUT; ++co) chunk[co] += part[co];
    }
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] += chunk[co];
    __syncthreads();
  }
  const int yy = ty0 + ly, xx = tx0 + lx;
  if (yy < H && xx < W)
#pragma unroll
    for (int co = 0; co < COUT; ++co) y[((size_t)b * COUT + co) * plane + (long)yy * W + xx] = tanhf(acc[co] + bias[co]);
}

// The same convolution for launches of a few hundred 16 x 16 tiles (batch 1: 256 workgroups, one per CU, each a chain of eight
// halo fills and 8 x 25 x COUT dependent FMAs per thread - 109 us for what costs 20 us per image inside a batch of 32): 8 x 8-pixel
// tiles, and the four wavefronts of a workgroup take the channels c = wave (mod 4) of every chunk; their sums meet in LDS.  Sums:
// 25 taps -> a wavefront's two channels of a chunk -> the chunks -> the four wavefronts (short chains, as above).
constexpr int IT2 = 8, IH2 = IT2 + 4;
template <int COUT>
__global__ __launch_bounds__(256) void conv_img_small_kernel(const float* __restrict__ x, int Cin, int H, int W,
                                                             const float* __restrict__ w, const float* __restrict__ bias,
                                                             float* __restrict__ y) {
  __shared__ float halo[ICK][IH2 * IH2];
  __shared__ float red[4][COUT][64];
  const int tiles_x = (W + IT2 - 1) / IT2;
  const int tx0 = (blockIdx.x % tiles_x) * IT2, ty0 = (blockIdx.x / tiles_x) * IT2, b = blockIdx.y;
  const int pix = threadIdx.x & 63, wv = threadIdx.x >> 6, lx = pix & (IT2 - 1), ly = pix >> 3;
  const long plane = (long)H * W;
  float acc[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
  for (int c0 = 0; c0 < Cin; c0 += ICK) {
    for (int e = threadIdx.x; e < ICK * IH2 * IH2; e += 256) {
      const int c = e / (IH2 * IH2), p = e % (IH2 * IH2);
      const int sy = ty0 + p / IH2 - 2, sx = tx0 + p % IH2 - 2;
      float v = 0.f;
      if (c0 + c < Cin && sy >= 0 && sy < H && sx >= 0 && sx < W) {
        v = x[((size_t)b * Cin + c0 + c) * plane + (long)sy * W + sx];
        v = v > 0.f ? v : 0.2f * v;
      }
      halo[c][p] = v;
    }
    __syncthreads();
    const int cn = min(ICK, Cin - c0);
    float chunk[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) chunk[co] = 0.f;
    for (int c = wv; c < cn; c += 4) {                        // wave-uniform: the weights still come through the scalar cache
      const float* wc = w + (size_t)(c0 + c) * 25;
      float part[COUT];
#pragma unroll
      for (int co = 0; co < COUT; ++co) part[co] = 0.f;
#pragma unroll
      for (int ky = 0; ky < 5; ++ky)
#pragma unroll
        for (int kx = 0; kx < 5; ++kx) {
          const float v = halo[c][(ly + ky) * IH2 + lx + kx];
#pragma unroll
          for (int co = 0; co < COUT; ++co) part[co] = fmaf(wc[(size_t)co * Cin * 25 + ky * 5 + kx], v, part[co]);
        }
#pragma unroll
      for (int co = 0; co < COUT; ++co) chunk[co] += part[co];
    }
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] += chunk[co];
    __syncthreads();
  }
#pragma unroll
  for (int co = 0; co < COUT; ++co) red[wv][co][pix] = acc[co];
  __syncthreads();
  const int yy = ty0 + ly, xx = tx0 + lx;
  if (wv == 0 && yy < H && xx < W)
#pragma unroll
    for (int co = 0; co < COUT; ++co)
      y[((size_t)b * COUT + co) * plane + (long)yy * W + xx] = tanhf((((red[0][co][pix] + red[1][co][pix]) + red[2][co][pix]) + red[3][co][pix]) + bias[co]);
}

// SPADE modulation with gamma/beta shared by the whole batch (one semantic map, many z): gb [rows_pad, plane] is the
// output of the packed [32 gamma | 32 beta] conv for that map; out[b,c,p] = ((x - mu_b) * inv_b) * (1 + gamma) + beta.
// HBM-bound: x read once, out written once, gb stays in L2/MALL across the batch (blockIdx.y = sample is the slow index).
// x_up: x is [B, C, H/2, W/2] read through nearest x2 (W given; 4 consecutive outputs of a row = 2 source pixels).
__global__ __launch_bounds__(256) void spade_apply_kernel(const float* __restrict__ x, const float* __restrict__ gb, int C, long plane,
                                                          const float* __restrict__ stats, int act, float slope, float* __restrict__ out,
                                                          int x_up, int W) {
  const long i4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  const long n = (long)C * plane;
  if (i4 >= n) return;
  const int b = blockIdx.y;
  const int c = (int)(i4 / plane);
  const long pix = i4 - (long)c * plane;
  const float mean = stats[2 * b], inv = stats[2 * b + 1];
  const long grow = (long)(c / 32) * 64 + (c % 32);
  float4 xv;
  if (x_up) {
    const int yo = (int)(pix / W), xo = (int)(pix % W);
    const float* xs = x + ((size_t)b * C + c) * (plane >> 2) + (size_t)(yo >> 1) * (W >> 1) + (xo >> 1);
    const float a0 = xs[0], a1 = xs[1];
    xv = make_float4(a0, a0, a1, a1);
  } else {
    xv = *reinterpret_cast<const float4*>(x + (size_t)b * n + i4);
  }
  const float4 g = *reinterpret_cast<const float4*>(gb + grow * plane + pix);
  const float4 be = *reinterpret_cast<const float4*>(gb + (grow + 32) * plane + pix);
  float4 v;
  v.x = (xv.x - mean) * inv; v.x = v.x * (1.f + g.x) + be.x;
  v.y = (xv.y - mean) * inv; v.y = v.y * (1.f + g.y) + be.y;
  v.z = (xv.z - mean) * inv; v.z = v.z * (1.f + g.z) + be.z;
  v.w = (xv.w - mean) * inv; v.w = v.w * (1.f + g.w) + be.w;
  if (act == ACT_LEAKY) {
    v.x = v.x > 0.f ? v.x : v.x * slope; v.y = v.y > 0.f ? v.y : v.y * slope;
    v.z = v.z > 0.f ? v.z : v.z * slope; v.w = v.w > 0.f ? v.w : v.w * slope;
  }
  *reinterpret_cast<float4*>(out + (size_t)b * n + i4) = v;
}

}  // namespace

extern "C" {

// Batch-shared modulation (see spade_apply_kernel): gb = sln_spade_conv(actv of the ONE map, packed gamma|beta weights).
int sln_spade_apply_up(const float* x, int x_up, const float* gb, int B, int C, int H, int W, int rows_pad, const float* stats, int act,
                       float slope, float* out, void* stream) {
  if (!x || !gb || !stats || !out || B <= 0 || C <= 0 || rows_pad < 64 * ((C + 31) / 32)) return SLN_E_BADARG;
  const long plane = (long)H * W;
  if (plane % 4 != 0 || (x_up && (W % 4 != 0 || H % 2 != 0))) return SLN_E_UNSUPPORTED;
  const long n4 = (long)C * plane / 4;
  SlnProfScope prof(SLN_FAM_OTHER, 4.0 * (2.0 * B * C * plane + 2.0 * C * plane), (hipStream_t)stream);
  hipLaunchKernelGGL(spade_apply_kernel, dim3((unsigned)((n4 + 255) / 256), B), dim3(256), 0, (hipStream_t)stream, x, gb, C, plane, stats, act,
                     slope, out, x_up, W);
  SLN_CHECK_LAUNCH();
  return 0;
}
int sln_spade_apply(const float* x, const float* gb, int B, int C, int H, int W, int rows_pad, const float* stats, int act, float slope,
                    float* out, void* stream) {
  return sln_spade_apply_up(x, 0, gb, B, C, H, W, rows_pad, stats, act, slope, out, stream);
}

// conv KSxKS (KS = 3 reflect pad 1, KS = 1) with packed weights wp[KS*KS][Cin][rows_pad]; rows_pad % 64 == 0.
int sln_spade_conv_sums(const float* x, int B, int Cin, int H, int W, const float* wp, const float* bias, int rows, int rows_pad,
                        int ksize, int act, float slope, float* y, double* ln_acc, double* gap_acc, void* stream) {
  if (!x || !wp || !y || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || rows <= 0 || rows_pad % 64 != 0 || rows > rows_pad) return SLN_E_BADARG;
  if (ksize != 1 && ksize != 3) return SLN_E_UNSUPPORTED;
  if (ksize == 3 && (H < 2 || W < 2)) return SLN_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  ConvArgs a; a.x = x; a.wp = wp; a.bias = bias; a.y = y; a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.rows = rows; a.rows_pad = rows_pad;
  a.act = act; a.slope = slope; a.xin = nullptr; a.stats = nullptr; a.C = 0; a.xin_up = 0; a.ln_acc = ln_acc; a.gap_acc = gap_acc;
  // blocked accumulation for the long chains (K = 9 Cin >= 4 608), see conv_mfma_kernel; SLN_CONV_BLOCK_CIN moves the threshold (lab)
  static const int block_cin = getenv("SLN_CONV_BLOCK_CIN") ? atoi(getenv("SLN_CONV_BLOCK_CIN")) : 512;
  a.blocked = ksize == 3 && Cin >= block_cin;
  a.part = nullptr; a.part_stride = 0;
  SlnProfScope prof(SLN_FAM_CONV, 2.0 * B * H * W * (double)Cin * ksize * ksize * rows, st);
  const bool big = rows_pad % 128 == 0;
  if (ksize == 3) return big ? launch_conv<128, 3, CEPI_BIAS_ACT>(a, st) : launch_conv<64, 3, CEPI_BIAS_ACT>(a, st);
  return big ? launch_conv<128, 1, CEPI_BIAS_ACT>(a, st) : launch_conv<64, 1, CEPI_BIAS_ACT>(a, st);
}
int sln_spade_prepare(void* stream) {
  float* p = nullptr;
  return conv_part_scratch((hipStream_t)stream, &p);
}
int sln_spade_release(void* stream, int all) {
  return conv_part_release((hipStream_t)stream, all != 0);
}

int sln_spade_conv(const float* x, int B, int Cin, int H, int W, const float* wp, const float* bias, int rows, int rows_pad,
                   int ksize, int act, float slope, float* y, void* stream) {
  return sln_spade_conv_sums(x, B, Cin, H, W, wp, bias, rows, rows_pad, ksize, act, slope, y, nullptr, nullptr, stream);
}

// out = LN(xin) * (1 + gamma) + beta [-> LeakyReLU(slope) when act == 2], gamma/beta = conv3x3_reflect(actv) with
// weights packed [32 gamma | 32 beta] per 64 rows (rows_pad = 64 * ceil(C / 32)).
int sln_spade_modulate_up(const float* actv, int B, int Cin, int H, int W, const float* wp, const float* bias, int C, int rows_pad,
                          const float* xin, int xin_up, const float* stats, int act, float slope, float* out, void* stream) {
  if (xin_up && ((H | W) & 1)) return SLN_E_BADARG;
  if ((int64_t)C * H * W >= (int64_t)1 << 31) return SLN_E_UNSUPPORTED;      // 32-bit offsets inside a sample
  if (!actv || !wp || !bias || !xin || !stats || !out || B <= 0 || C <= 0 || rows_pad % 64 != 0 || rows_pad < 64 * ((C + 31) / 32))
    return SLN_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  ConvArgs a; a.x = actv; a.wp = wp; a.bias = bias; a.y = out; a.B = B; a.Cin = Cin; a.H = H; a.W = W; a.rows = 2 * C; a.rows_pad = rows_pad;
  a.act = act; a.slope = slope; a.xin = xin; a.stats = stats; a.C = C; a.xin_up = xin_up; a.ln_acc = nullptr; a.gap_acc = nullptr; a.blocked = 0; a.part = nullptr; a.part_stride = 0;
  SlnProfScope prof(SLN_FAM_CONV, 2.0 * B * H * W * (double)Cin * 9 * 2 * C, st);
  return rows_pad % 128 == 0 ? launch_conv<128, 3, CEPI_MODULATE>(a, st) : launch_conv<64, 3, CEPI_MODULATE>(a, st);
}
int sln_spade_modulate(const float* actv, int B, int Cin, int H, int W, const float* wp, const float* bias, int C, int rows_pad,
                       const float* xin, const float* stats, int act, float slope, float* out, void* stream) {
  return sln_spade_modulate_up(actv, B, Cin, H, W, wp, bias, C, rows_pad, xin, 0, stats, act, slope, out, stream);
}

// stats[b] from sums a conv epilogue / block tail accumulated (acc [B][16] doubles: sum, sum of squares over n_acc values,
// each standing for `rep` elements of the normalised tensor)
int sln_layernorm_finalize(const double* acc, int B, int64_t n_acc, int rep, float eps, float* stats, void* stream) {
  if (!acc || !stats || B <= 0 || n_acc < 1 || rep < 1 || n_acc * rep < 2) return SLN_E_BADARG;
  hipLaunchKernelGGL(ln_finalize_kernel, dim3(sln_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, const_cast<double*>(acc), (long)n_acc, rep, B, eps, stats, 0);
  SLN_CHECK_LAUNCH();
  return 0;
}

// out = up(xs + dx * sigmoid(W2 relu(W0 GAP(dx)))) and the LayerNorm2D statistics of the next block's input in one pass.
//   xs [B,C,H,W] (xs_up = 1: [B,C,H/2,W/2] read through nearest x2), dx [B,C,H,W];
//   gap_sums: fp64 pixel sums of dx from sln_spade_conv_sums, or NULL (a reduction kernel computes the averages);
//   up_mode: -1 out [B,C,H,W]; 0 nearest / 1 bilinear: out [B,C,2H,2W];
//   stats_rep: 1, or 4 = the consumers read `out` through nearest x2 (statistics of that tensor);  stats may be NULL.
//   scratch: 2*B*C floats;  acc: 16*B doubles (zeroed here).
int sln_block_tail(const float* xs, int xs_up, const float* dx, int B, int C, int H, int W, const double* gap_sums, const float* w0,
                   const float* w2, float* scratch, int up_mode, float* out, double* acc, int stats_rep, float eps, float* stats,
                   void* stream) {
  if (!xs || !dx || !w0 || !w2 || !scratch || !out || B <= 0 || C <= 0 || C % 8 != 0 || up_mode < -1 || up_mode > 1) return SLN_E_BADARG;
  if (stats && !acc) return SLN_E_BADARG;
  if ((up_mode < 0 && W % 4 != 0) || (W % 2 != 0) || (xs_up && (H % 2 != 0))) return SLN_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const long hw = (long)H * W;
  float* gap = scratch; float* scale = scratch + (size_t)B * C;
  if (!gap_sums) hipLaunchKernelGGL(gap_kernel, dim3(B * C), dim3(256), 0, st, dx, hw, gap);
  double* zacc = stats ? acc : nullptr;                                  // cleared by the (last) FC launch, in front of the tail kernel
  const int nz = LN_ACC_STRIDE * B;
  if (gap_sums && B <= 8) {      // few samples: the two FCs as two launches of many workgroups; the hidden rows go through the unused gap buffer
    double* hid = reinterpret_cast<double*>(gap);                       // B * C / 8 doubles in B * C floats
    hipLaunchKernelGGL(se_fc_kernel, dim3(B, sln_cdiv(C / 8, 32)), dim3(256), sizeof(double) * (C + C / 8), st, gap, gap_sums, (double)hw, w0, w2, C, C / 8, scale, 1, hid,
                       (double*)nullptr, 0);
    hipLaunchKernelGGL(se_fc_kernel, dim3(B, sln_cdiv(C, 128)), dim3(256), sizeof(double) * (C + C / 8), st, gap, gap_sums, (double)hw, w0, w2, C, C / 8, scale, 2, hid,
                       zacc, nz);
  } else
  hipLaunchKernelGGL(se_fc_kernel, dim3(B), dim3(256), sizeof(double) * (C + C / 8), st, gap, gap_sums, (double)hw, w0, w2, C, C / 8, scale, 0, (double*)nullptr,
                     zacc, nz);
  const long n_out = (long)C * hw * (up_mode >= 0 ? 4 : 1);
  SlnProfScope prof(SLN_FAM_OTHER, 4.0 * B * (2.0 * C * hw + n_out), st);
  long gx = (n_out / 4 + 255) / 256;
  // workgroups per sample: with the LayerNorm sums on, every workgroup ends in two fp64 atomics on its sample's one 128-byte line,
  // and the memory side serialises them (~12 ns each): at batch 1 the 4 096 workgroups of the 256 x 256 tail spent 98 of their
  // 103 us there.  512 per sample at most then (the loop below strides over the grid).
  long cap = 4096 / B > 16 ? 4096 / B : 16;
  if (stats && cap > 512) cap = 512;
  gx = gx > cap ? cap : gx;
  const dim3 grid((unsigned)gx, B);
  double* ac = stats ? acc : nullptr;
  if (up_mode < 0) hipLaunchKernelGGL(block_tail_kernel<0>, grid, dim3(256), 0, st, xs, dx, scale, C, H, W, xs_up, out, ac);
  else if (up_mode == 0) hipLaunchKernelGGL(block_tail_kernel<1>, grid, dim3(256), 0, st, xs, dx, scale, C, H, W, xs_up, out, ac);
  else hipLaunchKernelGGL(block_tail_kernel<2>, grid, dim3(256), 0, st, xs, dx, scale, C, H, W, xs_up, out, ac);
  if (stats) hipLaunchKernelGGL(ln_finalize_kernel, dim3(sln_cdiv(B, 64)), dim3(64), 0, st, acc, n_out, stats_rep, B, eps, stats, 0);
  SLN_CHECK_LAUNCH();
  return 0;
}

// stats[b] = (mean, 1/(std_unbiased + eps)) over the n = C*H*W elements of sample b; scratch: 2*B doubles
int sln_layernorm_stats(const float* x, int B, int64_t n, float eps, double* scratch, float* stats, void* stream) {
  if (!x || !scratch || !stats || B <= 0 || n < 2) return SLN_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const int e = sln_zero_async(scratch, sizeof(double) * LN_ACC_STRIDE * B, st);
  if (e != 0) return e;
  int gx = (int)((n + 256 * 16 - 1) / (256 * 16)); gx = gx > 128 ? 128 : (gx < 1 ? 1 : gx);
  // deterministic mode: at most 7 blocks per sample, each storing its sums to its own slot; the finalize adds them in order
  const int det = g_sln_deterministic ? 1 : 0;
  if (det && gx > LN_ACC_STRIDE / 2 - 1) gx = LN_ACC_STRIDE / 2 - 1;
  hipLaunchKernelGGL(ln_stats_kernel, dim3(gx, B), dim3(256), 0, st, x, (long)n, scratch, det);
  hipLaunchKernelGGL(ln_finalize_kernel, dim3(sln_cdiv(B, 64)), dim3(64), 0, st, scratch, (long)n, 1, B, eps, stats, det ? gx : 0);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_resize(const float* src, int BC, int Hi, int Wi, int Ho, int Wo, int mode, float* dst, void* stream) {
  if (!src || !dst || BC <= 0) return SLN_E_BADARG;
  const long n = (long)BC * Ho * Wo;
  hipLaunchKernelGGL(resize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, 0, Hi, Wi, Ho, Wo, mode,
                     n, dst);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_spade_depth_concat(const float* seg, int B, int Cs, int H, int W, const float* wpd, const float* bpd, int nd, float* out,
                           int copy_masks, void* stream) {
  if (!seg || !wpd || !bpd || !out) return SLN_E_BADARG;
  const int cw = copy_masks ? nd + Cs - 1 : nd;
  if (B <= 0 || cw <= 0 || H <= 0 || W <= 0 || (long)H * W > (1L << 30) || B > 65535 || cw > 65535) return SLN_E_BADARG;
  hipLaunchKernelGGL(depth_concat_kernel, dim3((unsigned)(((long)H * W + 255) / 256), cw, B), dim3(256), 0, (hipStream_t)stream, seg, Cs, H, W,
                     wpd, bpd, nd, out);
  SLN_CHECK_LAUNCH();
  return 0;
}

// out = xs + dx * sigmoid(W2 relu(W0 GAP(dx)))      scratch: B*C (gap) + B*C (scale) floats
int sln_se_scale_add(const float* xs, const float* dx, int B, int C, int64_t hw, const float* w0, const float* w2, float* scratch,
                     float* out, void* stream) {
  if (!xs || !dx || !w0 || !w2 || !scratch || !out || C % 8 != 0) return SLN_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  float* gap = scratch; float* scale = scratch + (size_t)B * C;
  hipLaunchKernelGGL(gap_kernel, dim3(B * C), dim3(256), 0, st, dx, (long)hw, gap);
  hipLaunchKernelGGL(se_fc_kernel, dim3(B), dim3(256), sizeof(double) * (C + C / 8), st, gap, (const double*)nullptr, (double)hw, w0, w2,
                     C, C / 8, scale, 0, (double*)nullptr, (double*)nullptr, 0);
  const long n = (long)B * C * hw;
  hipLaunchKernelGGL(se_scale_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, xs, dx, scale, (long)hw, n, out);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_upsample2x(const float* x, int BC, int H, int W, int mode, float* y, void* stream) {
  if (!x || !y) return SLN_E_BADARG;
  const long n = (long)BC * 4 * H * W;
  hipLaunchKernelGGL(upsample2x_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, H, W, mode, n, y);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_conv_img_tanh(const float* x, int B, int Cin, int H, int W, const float* w, const float* bias, int Cout, float* y, void* stream) {
  if (!x || !w || !bias || !y || Cout > 4 || Cout <= 0) return SLN_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  SlnProfScope prof(SLN_FAM_OTHER, 4.0 * B * H * W * (Cin + Cout), st);
  static const bool no_small = getenv("SLN_CONV_IMG_NO_SMALL") != nullptr;      // lab
  if (!no_small && (long)sln_cdiv(W, IT) * sln_cdiv(H, IT) * B < 512) {      // a few hundred 16 x 16 tiles: 8 x 8 tiles, channels over the wavefronts
    const dim3 g2(sln_cdiv(W, IT2) * sln_cdiv(H, IT2), B);
    switch (Cout) {
      case 1: hipLaunchKernelGGL(conv_img_small_kernel<1>, g2, dim3(256), 0, st, x, Cin, H, W, w, bias, y); break;
      case 2: hipLaunchKernelGGL(conv_img_small_kernel<2>, g2, dim3(256), 0, st, x, Cin, H, W, w, bias, y); break;
      case 3: hipLaunchKernelGGL(conv_img_small_kernel<3>, g2, dim3(256), 0, st, x, Cin, H, W, w, bias, y); break;
      default: hipLaunchKernelGGL(conv_img_small_kernel<4>, g2, dim3(256), 0, st, x, Cin, H, W, w, bias, y); break;
    }
    SLN_CHECK_LAUNCH();
    return 0;
  }
  const dim3 grid(sln_cdiv(W, IT) * sln_cdiv(H, IT), B);
  switch (Cout) {
    case 1: hipLaunchKernelGGL(conv_img_kernel<1>, grid, dim3(256), 0, st, x, Cin, H, W, w, bias, y); break;
    case 2: hipLaunchKernelGGL(conv_img_kernel<2>, grid, dim3(256), 0, st, x, Cin, H, W, w, bias, y); break;
    case 3: hipLaunchKernelGGL(conv_img_kernel<3>, grid, dim3(256), 0, st, x, Cin, H, W, w, bias, y); break;
    default: hipLaunchKernelGGL(conv_img_kernel<4>, grid, dim3(256), 0, st, x, Cin, H, W, w, bias, y); break;
  }
  SLN_CHECK_LAUNCH();
  return 0;
}

}  // extern "C"
