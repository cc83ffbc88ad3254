// What does ONE wavefront per SIMD overlap with its own dependent MFMA chain?  (GPU box only.)
// Every wave runs `iters` x 32 dependent v_mfma_f32_32x32x2_f32 with K other instructions of one kind behind each MFMA
// (sched_barrier keeps the order): independent VALU, ds_write_b128, ds_read_b128, global_load_dwordx4 (L2 hits), SALU.
// Prints clocks per MFMA (64 = the matrix pipe never waits).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/overlap.hip -o /tmp/overlap && /tmp/overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int K, int NACC>
__global__ __launch_bounds__(256) void k(float* out, long long* ticks, const float* g, int iters) {
  __shared__ __attribute__((aligned(16))) float sm[4096];
  f32x16 a0[NACC] = {};
  const int tid = threadIdx.x;
  float x = tid * 1e-3f, y = blockIdx.x * 1e-3f;
  float t[4] = {x, y, x + 1.f, y + 1.f};
  f32x4 w = {x, y, x, y};
  f32x4 rd[4] = {w, w, w, w};
  const unsigned laddr = (unsigned)(tid * 16);          // byte address inside sm (conflict-free b128 pattern)
  const float* gp = g + (tid & 255) * 4;
  int sacc = 0;
  for (int i = tid; i < 4096; i += 256) sm[i] = 0.f;
  __syncthreads();
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      a0[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0[u % NACC], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < K; ++j) {
        if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(t[j & 3]) : "v"(x), "v"(y));
        if (MODE == 1) asm volatile("ds_write_b128 %0, %1" :: "v"(laddr), "v"(w) : "memory");
        if (MODE == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(rd[j & 3]) : "v"(laddr) : "memory");
        if (MODE == 3) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rd[j & 3]) : "v"(gp) : "memory");
        if (MODE == 4) asm volatile("s_add_i32 %0, %0, 1" : "+s"(sacc));
        if (MODE == 5) asm volatile("v_cndmask_b32_e64 %0, 0, %1, vcc" : "=v"(t[j & 3]) : "v"(x));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 1 || MODE == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (MODE == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const long long t1 = clock64();
  float s = t[0] + t[1] + t[2] + t[3] + (float)sacc;
  for (int q = 0; q < NACC; ++q) for (int r = 0; r < 16; ++r) s += a0[q][r];
  for (int q = 0; q < 4; ++q) s += rd[q][0] + rd[q][1] + rd[q][2] + rd[q][3];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE, int K, int NACC = 1>
void run(const char* name, float* out, long long* ticks, const float* g, int blocks, int iters) {
  hipLaunchKernelGGL((k<MODE, K, NACC>), dim3(blocks), dim3(256), 0, 0, out, ticks, g, 4);
  hipLaunchKernelGGL((k<MODE, K, NACC>), dim3(blocks), dim3(256), 0, 0, out, ticks, g, iters);
  hipDeviceSynchronize();
  long long h[256];
  hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
  long long sum = 0;
  for (int i = 0; i < 256; ++i) sum += h[i];
  printf("%-22s K=%2d acc=%d blocks=%d : %6.1f clocks per MFMA of a wave (%.1f per SIMD-MFMA)\n", name, K, NACC, blocks, (double)sum / 256 / iters / 32,
         (double)sum / 256 / iters / 32 / (blocks / 256));
}

#define ROW(MODE, NAME) \
  run<MODE, 0>(NAME, out, ticks, g, blocks, iters); run<MODE, 1>(NAME, out, ticks, g, blocks, iters); run<MODE, 2>(NAME, out, ticks, g, blocks, iters); \
  run<MODE, 4>(NAME, out, ticks, g, blocks, iters); run<MODE, 8>(NAME, out, ticks, g, blocks, iters); run<MODE, 12>(NAME, out, ticks, g, blocks, iters); \
  run<MODE, 16>(NAME, out, ticks, g, blocks, iters);

int main(int argc, char**) {
  const int blocks = 256, iters = 200;
  float* out; hipMalloc(&out, sizeof(float) * 512 * 256);
  long long* ticks; hipMalloc(&ticks, sizeof(long long) * 512);
  float* g; hipMalloc(&g, 65536); hipMemset(g, 0, 65536);
  if (argc > 1) {
    ROW(0, "v_fma_f32")
    ROW(5, "v_cndmask_b32_e64")
    ROW(4, "s_add_i32")
    ROW(1, "ds_write_b128")
    ROW(2, "ds_read_b128")
    ROW(3, "global_load_dwordx4")
  }
  // independent accumulator chains in ONE wave: does VALU issue in the shadow of an MFMA that the next MFMA does not depend on?
  run<0, 0, 2>("v_fma 2 chains", out, ticks, g, blocks, iters); run<0, 4, 2>("v_fma 2 chains", out, ticks, g, blocks, iters);
  run<0, 8, 2>("v_fma 2 chains", out, ticks, g, blocks, iters); run<0, 12, 2>("v_fma 2 chains", out, ticks, g, blocks, iters);
  run<0, 0, 4>("v_fma 4 chains", out, ticks, g, blocks, iters); run<0, 4, 4>("v_fma 4 chains", out, ticks, g, blocks, iters);
  run<0, 8, 4>("v_fma 4 chains", out, ticks, g, blocks, iters); run<0, 12, 4>("v_fma 4 chains", out, ticks, g, blocks, iters);
  // TWO waves per SIMD (512 workgroups), one dependent chain each: does wave B's MFMA run under wave A's VALU?
  run<0, 0, 1>("v_fma 2 waves/SIMD", out, ticks, g, 512, iters); run<0, 4, 1>("v_fma 2 waves/SIMD", out, ticks, g, 512, iters);
  run<0, 8, 1>("v_fma 2 waves/SIMD", out, ticks, g, 512, iters); run<0, 12, 1>("v_fma 2 waves/SIMD", out, ticks, g, 512, iters);
  run<0, 16, 1>("v_fma 2 waves/SIMD", out, ticks, g, 512, iters);
  run<5, 8, 1>("cndmask 2 waves/SIMD", out, ticks, g, 512, iters);
  run<1, 2, 1>("ds_write 2 waves/SIMD", out, ticks, g, 512, iters);
  run<3, 1, 1>("gload 2 waves/SIMD", out, ticks, g, 512, iters);
  return 0;
}
