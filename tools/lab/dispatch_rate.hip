// How fast does the chip retire workgroups that leave after one flag load?  (GPU box only.)  pixel_map_backward launches one
// one-wavefront workgroup per (face, edge x axis) and three of four find that their face owns no pixel; the same ~371 k wavefronts
// are launched here as workgroups of 1, 2, 3, 6 wavefronts.   hipcc --offload-arch=gfx950 -O3 tools/lab/dispatch_rate.hip -o /tmp/dr && /tmp/dr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void leave(const int* __restrict__ flags, int* __restrict__ out, int waves_per_wg) {
  const int w = blockIdx.x * waves_per_wg + (threadIdx.x >> 6);
  if (flags[w] == 0) return;
  out[w] = 1;
}
int main() {
  const int waves = 370944;
  int *flags, *out; CK(hipMalloc(&flags, waves * 4 + 64)); CK(hipMalloc(&out, waves * 4 + 64)); CK(hipMemset(flags, 0, waves * 4 + 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int wpw : {1, 2, 3, 4, 6, 8, 12}) {
    const int grid = (waves + wpw - 1) / wpw;
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipEventRecord(e0, 0));
      leave<<<grid, 64 * wpw>>>(flags, out, wpw);
      CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
    }
    printf("%2d wavefronts per workgroup, %6d workgroups: %7.1f us  (%.2f ns per wavefront)\n", wpw, grid, best * 1e3, best * 1e6 / waves);
  }
  return 0;
}
