// Sustained fp32 MFMA rate of the part: v_mfma_f32_32x32x2_f32 back to back from registers only (no LDS, no HBM; argv[4] = 1: random operands), 4 independent
// accumulators per wave, W waves per SIMD, for ~DUR ms per launch.  Build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void spin(float* out, int iters, int random_data) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
  if (random_data) {                       // operands with all mantissa bits toggling (the power the data path draws depends on them)
    unsigned h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u + 12345u);
    h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
    x = (float)(int)h * (1.0f / 2147483648.0f);
    h *= 0x5bd1e995u; h ^= h >> 13;
    y = (float)(int)h * (1.0f / 2147483648.0f);
  }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// mode 9: TWO accumulators alternating (dependency distance 2); mode 10: the 16x16x4 shape on one accumulator; mode 11: 16x16x4, four
template <int NACC>
__global__ __launch_bounds__(256) void spin_chain_n(float* out, int iters) {
  f32x16 a[NACC] = {};
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 32; ++u) a[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a[u % NACC], 0, 0, 0);
  }
  float s = 0.f;
  for (int k = 0; k < NACC; ++k) for (int r = 0; r < 16; ++r) s += a[k][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void spin_16(float* out, int iters) {        // v_mfma_f32_16x16x4_f32: 2048 flop per instruction
  f32x4 a[NACC] = {};
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 64; ++u) a[u % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a[u % NACC], 0, 0, 0);
  }
  float s = 0.f;
  for (int k = 0; k < NACC; ++k) for (int r = 0; r < 4; ++r) s += a[k][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// mode 8: ONE accumulator per wave - a dependent MFMA chain (the 32 x 32 wave tile of the 64 x 64 GEMM blocks).  Measured: 154.8
// TFLOP/s like every other variant here (2 / 3 accumulators, 16x16x4 on 1 / 2 / 4): a dependent chain costs nothing on gfx950.
__global__ __launch_bounds__(256) void spin_chain(float* out, int iters) {
  f32x16 a0 = {0};
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 32; ++u) a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += a0[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// The conv kernel's K step: 4 ds_read_b32 ahead, 4 MFMAs (mode 2); mode 3 adds a global load per step (L2 hits).
__global__ __launch_bounds__(256) void spin_lds(float* out, int iters, const float* g) {
  __shared__ float sm[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = (float)((i * 2654435761u) >> 8) * (1.0f / 16777216.0f) - 0.5f;
  __syncthreads();
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  const int lane = threadIdx.x & 63;
  float x0 = sm[lane], x1 = sm[lane + 64], y0 = sm[lane + 128], y1 = sm[lane + 192], gsum = 0.f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int o = ((i * 8 + u) * 256) & 4095;
      const float nx0 = sm[o + lane], nx1 = sm[o + lane + 1024], ny0 = sm[o + 2048 + lane], ny1 = sm[o + 3072 + lane];
      if (g) gsum += g[(o + threadIdx.x) & 1023];
      __builtin_amdgcn_sched_barrier(0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y0, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y1, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y0, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y1, a3, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      x0 = nx0; x1 = nx1; y0 = ny0; y1 = ny1;
    }
  }
  float s = gsum;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// mode 4: the conv kernel's chunk loop verbatim (BMC 128, 3x3: [9][8][128] weights + 8 x 10 x 18 halo in LDS, 9 rolled taps x 4
// fenced steps, ds_read2 for the weights, the 2-way bank conflict of the two-row pixel tile), without loads and barriers.
// mode 5: the same with a conflict-free pixel mapping (one 32-pixel row per tile).  modes 6 / 7: 4 / 5 with the taps unrolled.
template <int CONFLICT, int UNROLL>
__global__ __launch_bounds__(256) void spin_conv(float* out, int iters) {
  constexpr int BMC = 128, CK = 8, TW = 16, HS = 180, WSLAB = 9 * CK * BMC;
  extern __shared__ float lds[];
  float* wl = lds; float* xl = lds + WSLAB;
  for (int i = threadIdx.x; i < WSLAB + CK * HS + 64; i += 256) lds[i] = (float)((i * 2654435761u) >> 8) * (1.0f / 16777216.0f) - 0.5f;
  __syncthreads();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = (wave / 2) * 64, wp0 = (wave % 2) * 64, li = lane & 31, lk = lane >> 5;
  int pbase[2];
  for (int j = 0; j < 2; ++j) { const int m = wp0 + 32 * j + li; pbase[j] = CONFLICT ? (m / TW) * (TW + 2) + (m % TW) : (m / 32) * 36 + (m % 32); }
  f32x16 acc[2][2] = {};
  auto ld = [&](int tap, int kk, float (&av)[2], float (&bv)[2]) {
    const int toff = CONFLICT ? (tap / 3) * (TW + 2) + (tap % 3) : 0;
    for (int i = 0; i < 2; ++i) av[i] = wl[(tap * CK + kk + lk) * BMC + wr + 32 * i + li];
    for (int j = 0; j < 2; ++j) bv[j] = xl[(kk + lk) * HS + pbase[j] + toff];
  };
  auto mma = [&](const float (&av)[2], const float (&bv)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
  };
  for (int it = 0; it < iters; ++it) {
    float av0[2], bv0[2], av1[2], bv1[2];
    ld(0, 0, av0, bv0);
#pragma unroll UNROLL
    for (int tap = 0; tap < 9; ++tap) {
      ld(tap, 2, av1, bv1); __builtin_amdgcn_sched_barrier(0);
      mma(av0, bv0); __builtin_amdgcn_sched_barrier(0);
      ld(tap, 4, av0, bv0); __builtin_amdgcn_sched_barrier(0);
      mma(av1, bv1); __builtin_amdgcn_sched_barrier(0);
      ld(tap, 6, av1, bv1); __builtin_amdgcn_sched_barrier(0);
      mma(av0, bv0); __builtin_amdgcn_sched_barrier(0);
      ld(tap + 1 < 9 ? tap + 1 : tap, 0, av0, bv0); __builtin_amdgcn_sched_barrier(0);
      mma(av1, bv1); __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main(int argc, char** argv) {
  const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 2;
  const int iters = argc > 2 ? atoi(argv[2]) : 20000;
  const int reps = argc > 3 ? atoi(argv[3]) : 20;
  const int blocks = 256 * waves_per_simd;           // 256 CUs x (4 waves per block = 1 per SIMD)
  float* out; hipMalloc(&out, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int rnd = argc > 4 ? atoi(argv[4]) : 0;
  float* gbuf; hipMalloc(&gbuf, 4096); hipMemset(gbuf, 0, 4096);
  spin<<<blocks, 256>>>(out, 100, rnd); hipDeviceSynchronize();
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(e0); if (rnd == 8) spin_chain<<<blocks, 256>>>(out, iters); else if (rnd == 9) spin_chain_n<2><<<blocks, 256>>>(out, iters); else if (rnd == 12) spin_chain_n<3><<<blocks, 256>>>(out, iters); else if (rnd == 10) spin_16<1><<<blocks, 256>>>(out, iters); else if (rnd == 11) spin_16<4><<<blocks, 256>>>(out, iters); else if (rnd == 13) spin_16<2><<<blocks, 256>>>(out, iters); else if (rnd == 4) spin_conv<1, 1><<<blocks, 256, 43008>>>(out, iters / 18); else if (rnd == 5) spin_conv<0, 1><<<blocks, 256, 43008>>>(out, iters / 18); else if (rnd == 6) spin_conv<1, 9><<<blocks, 256, 43008>>>(out, iters / 18); else if (rnd == 7) spin_conv<0, 9><<<blocks, 256, 43008>>>(out, iters / 18); else if (rnd >= 2) spin_lds<<<blocks, 256>>>(out, iters, rnd == 3 ? gbuf : nullptr); else spin<<<blocks, 256>>>(out, iters, rnd); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * (rnd >= 4 && rnd <= 7 ? (iters / 18) * 144.0 / 32 : (double)iters) * 32 * 2.0 * 32 * 32 * 2;
    printf("rep %2d: %.2f ms  %.1f TFLOP/s  (implied clock %.0f MHz at 256 flop/clk/CU)\n", r, ms, flop / ms / 1e9, flop / ms / 1e9 * 1e6 / (256.0 * 256.0) );
  }
  return 0;
}
