// Sustained fp32 MFMA rate of the part: v_mfma_f32_32x32x2_f32 back to back from registers only (no LDS, no HBM), 4 independent
// accumulators per wave, W waves per SIMD, for ~DUR ms per launch.  Build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void spin(float* out, int iters) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
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
int main(int argc, char** argv) {
  const int waves_per_simd = argc > 1 ? atoi(argv[1]) : 2;
  const int iters = argc > 2 ? atoi(argv[2]) : 20000;
  const int reps = argc > 3 ? atoi(argv[3]) : 20;
  const int blocks = 256 * waves_per_simd;           // 256 CUs x (4 waves per block = 1 per SIMD)
  float* out; hipMalloc(&out, sizeof(float) * blocks * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  spin<<<blocks, 256>>>(out, 100); hipDeviceSynchronize();
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(e0); spin<<<blocks, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)blocks * 4 * iters * 32 * 2.0 * 32 * 32 * 2;
    printf("rep %2d: %.2f ms  %.1f TFLOP/s  (implied clock %.0f MHz at 256 flop/clk/CU)\n", r, ms, flop / ms / 1e9, flop / ms / 1e9 * 1e6 / (256.0 * 256.0) );
  }
  return 0;
}
