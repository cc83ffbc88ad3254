// What does a device-wide barrier inside one persistent kernel cost next to the kernel boundary it would replace?  (GPU box only.)
// The c2 step is ~145 dependent launches because every train-mode BatchNorm is a grid-wide reduction between two Linears; a fused
// GraphTripleConv stage would trade each boundary for a barrier.  Both forms below run the SAME dependent phases: every workgroup reads
// PAY floats another workgroup (on another XCD: +17) wrote in the phase before, adds one, writes its own PAY floats.
//   (a) one launch per phase, 200 launches captured in a hipGraph (what the engine does today);
//   (b) one launch, a barrier between phases: arrive = release at agent scope (L2 write-back) + atomic add, wait = spin on an acquire load
//       + acquire fence (L2 / L1 invalidate of lines owned elsewhere) in every wavefront.
// Prints us per phase for both and checks the sums (a stale read shows as a wrong count).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/grid_barrier.hip -o /tmp/grid_barrier && /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void phase(const float* in, float* out, int pay, int G) {
  const int src = (blockIdx.x + 17) % G;
  for (int i = threadIdx.x; i < pay; i += blockDim.x) out[(size_t)blockIdx.x * pay + i] = in[(size_t)src * pay + i] + 1.0f;
}

__global__ __launch_bounds__(256) void one_phase(const float* in, float* out, int pay, int G) { phase(in, out, pay, G); }

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
  __atomic_thread_fence(__ATOMIC_ACQUIRE);       // (hipcc: agent scope for the plain builtin) every wavefront drops its stale lines
}

__global__ __launch_bounds__(256) void all_phases(float* a, float* b, int pay, int G, int phases, unsigned* ctr) {
  for (int p = 0; p < phases; ++p) {
    phase(p & 1 ? b : a, p & 1 ? a : b, pay, G);
    grid_barrier(ctr, (unsigned)(p + 1) * (unsigned)G);
  }
}

int main() {
  const int phases = 200;
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int G : {128, 256, 512, 1024})
    for (int pay : {64, 4096, 16384}) {
      float *a, *b; unsigned* ctr;
      const size_t n = (size_t)G * pay;
      CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&ctr, 4));
      std::vector<float> h(n);
      auto check = [&](const char* what) {
        CK(hipMemcpy(h.data(), a, n * 4, hipMemcpyDeviceToHost));       // (an even number of phases ends in a)
        size_t bad = 0;
        for (size_t i = 0; i < n; ++i) bad += h[i] != (float)phases;
        if (bad) printf("   %s: %zu of %zu values WRONG\n", what, bad, n);
      };
      // (a) graph of launches
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int p = 0; p < phases; ++p) one_phase<<<G, 256, 0, st>>>(p & 1 ? b : a, p & 1 ? a : b, pay, G);
      CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      float ms_a = 1e9f, ms_b = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemsetAsync(a, 0, n * 4, st));
        CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms_a = ms < ms_a ? ms : ms_a;
      }
      check("launches");
      // (b) one launch with barriers (G <= resident capacity: 256 CUs x 8 workgroups of 256 threads)
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemsetAsync(a, 0, n * 4, st)); CK(hipMemsetAsync(ctr, 0, 4, st));
        CK(hipEventRecord(e0, st));
        all_phases<<<G, 256, 0, st>>>(a, b, pay, G, phases, ctr);
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms_b = ms < ms_b ? ms : ms_b;
      }
      check("barriers");
      printf("G %4d workgroups, %6d floats each per phase: launch per phase %6.2f us   barrier per phase %6.2f us\n", G, pay,
             ms_a / phases * 1e3, ms_b / phases * 1e3);
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(ctr));
    }
  return 0;
}
