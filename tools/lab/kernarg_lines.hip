// kernarg fetch time against the number of 64-byte lines of the argument block a kernel reads (GPU box only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>
struct Big { int v[160]; };   // 640 bytes = 10 lines
template <int L, int STRIDE>
__global__ void k(const Big a, long long* out) {
  const long long t0 = clock64();
  int m = 0;
#pragma unroll
  for (int i = 0; i < L; ++i) { int x; asm volatile("s_mov_b32 %0, %1" : "=s"(x) : "s"(a.v[i * STRIDE])); m += x; }
  const long long t1 = clock64();
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = m; }
}
template <int L, int STRIDE>
void run(const Big& a, long long* out) {
  std::vector<long long> h(512), d;
  for (int it = 0; it < 20; ++it) {
    hipLaunchKernelGGL((k<L, STRIDE>), dim3(256), dim3(256), 0, 0, a, out);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), out, 8 * 512, hipMemcpyDeviceToHost);
    if (it >= 5) for (int b = 0; b < 256; ++b) d.push_back(h[2 * b]);
  }
  std::sort(d.begin(), d.end());
  printf("%2d dwords read, one every %3d bytes (%2d lines): p50 %lld ticks (p10 %lld, p90 %lld)\n", L, STRIDE * 4, (L * STRIDE * 4 + 63) / 64, d[d.size() / 2], d[d.size() / 10],
         d[d.size() * 9 / 10]);
}
int main() {
  long long* out; hipMalloc(&out, 8 * 512);
  Big a; for (auto& q : a.v) q = 1;
  run<1, 1>(a, out); run<2, 16>(a, out); run<4, 16>(a, out); run<6, 16>(a, out); run<8, 16>(a, out); run<10, 16>(a, out);
  run<16, 1>(a, out); run<32, 1>(a, out); run<64, 1>(a, out); run<128, 1>(a, out); run<160, 1>(a, out);
  return 0;
}
