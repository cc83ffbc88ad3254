#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>
struct Big { const float* p[20]; int v[40]; };
__global__ void k_struct(const Big a, long long* out) {
  const long long t0 = clock64();
  int m; asm volatile("s_mov_b32 %0, %1" : "=s"(m) : "s"(a.v[3]));
  const long long t1 = clock64();
  const float x = a.p[2][threadIdx.x & 63];
  const long long t2 = clock64();
  if (threadIdx.x == 0) { out[blockIdx.x * 4] = t1 - t0; out[blockIdx.x * 4 + 1] = t2 - t1; out[blockIdx.x * 4 + 2] = m + (int)x; }
}
__global__ void k_scalar(const float* p, int v, long long* out, const Big a) {
  const long long t0 = clock64();
  int m; asm volatile("s_mov_b32 %0, %1" : "=s"(m) : "s"(v));
  const long long t1 = clock64();
  const float x = p[threadIdx.x & 63];
  const long long t2 = clock64();
  if (threadIdx.x == 0) { out[blockIdx.x * 4] = t1 - t0; out[blockIdx.x * 4 + 1] = t2 - t1; out[blockIdx.x * 4 + 2] = m + (int)x + a.v[5]; }
}
int main() {
  long long* out; hipMalloc(&out, 8 * 4 * 256);
  float* buf; hipMalloc(&buf, 4096); hipMemset(buf, 0, 4096);
  Big a; for (auto& q : a.p) q = buf; for (auto& q : a.v) q = 7;
  std::vector<long long> h(4 * 256);
  for (int variant = 0; variant < 2; ++variant) {
    std::vector<long long> d0, d1;
    for (int it = 0; it < 20; ++it) {
      if (variant == 0) hipLaunchKernelGGL(k_struct, dim3(256), dim3(256), 0, 0, a, out);
      else hipLaunchKernelGGL(k_scalar, dim3(256), dim3(256), 0, 0, buf, 7, out, a);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), out, 8 * 4 * 256, hipMemcpyDeviceToHost);
      if (it >= 5) for (int b = 0; b < 256; ++b) { d0.push_back(h[4 * b]); d1.push_back(h[4 * b + 1]); }
    }
    std::sort(d0.begin(), d0.end()); std::sort(d1.begin(), d1.end());
    printf("%s: first kernel argument after %lld ticks (p10 %lld, p90 %lld); first global load through it after another %lld\n", variant ? "scalar args" : "struct arg ",
           d0[d0.size() / 2], d0[d0.size() / 10], d0[d0.size() * 9 / 10], d1[d1.size() / 2]);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int it = 0; it < 200; ++it) {
      if (variant == 0) hipLaunchKernelGGL(k_struct, dim3(256), dim3(256), 0, 0, a, out);
      else hipLaunchKernelGGL(k_scalar, dim3(256), dim3(256), 0, 0, buf, 7, out, a);
    }
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("   back-to-back launches: %.2f us each\n", ms / 200 * 1e3);
  }
  return 0;
}
