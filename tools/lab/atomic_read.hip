// How far away do device-scope fp64 atomics leave their result?  Kernel P adds into an array with atomicAdd (every workgroup, as
// the statistics epilogues do) or writes it with plain stores; kernel C (next launch) times one dependent load of it per workgroup.
//   hipcc --offload-arch=gfx950 -O3 tools/lab/atomic_read.hip -o /tmp/atomic_read && /tmp/atomic_read
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void produce(double* a, int n, int mode) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    if (mode == 0) atomicAdd(a + i, 1.0);
    else if (mode == 1) { if (blockIdx.x == 0) a[i] = 1.0; }
    else if (mode == 2) { if (blockIdx.x == (unsigned)(i & 255)) a[i] = 1.0; }      // written by many workgroups (many XCDs)
  }
}
__global__ void consume(const double* a, int n, long long* ticks, double* sink) {
  const int i = (blockIdx.x * 64 + threadIdx.x) % n;
  const long long t0 = clock64();
  const double v = a[i];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = clock64();
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
  if (v == 123.0) sink[0] = v;
}
int main() {
  const int n = 1280;
  double *a, *sink; long long* ticks;
  hipMalloc(&a, sizeof(double) * n); hipMalloc(&sink, 8); hipMalloc(&ticks, sizeof(long long) * 256);
  const char* names[3] = {"atomicAdd from 256 workgroups", "plain stores from one workgroup", "plain stores from 256 workgroups"};
  for (int mode = 0; mode < 3; ++mode) {
    std::vector<long long> all;
    for (int rep = 0; rep < 20; ++rep) {
      hipMemset(a, 0, sizeof(double) * n);
      hipLaunchKernelGGL(produce, dim3(256), dim3(256), 0, 0, a, n, mode);
      hipLaunchKernelGGL(consume, dim3(256), dim3(64), 0, 0, a, n, ticks, sink);
      hipDeviceSynchronize();
      long long h[256]; hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
      if (rep >= 2) for (int i = 0; i < 256; ++i) all.push_back(h[i]);
    }
    std::sort(all.begin(), all.end());
    printf("%-36s: load latency in the next kernel p10 %lld  p50 %lld  p90 %lld clocks\n", names[mode], all[all.size() / 10], all[all.size() / 2], all[all.size() * 9 / 10]);
  }
  return 0;
}
