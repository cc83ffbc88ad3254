// Stand-alone GEMM laboratory (GPU box only): phase stamps of the product's NT body on the VAE's shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -I include -I 3d_sln_amd/csrc tools/lab/gemm_lab.hip -o /tmp/gemm_lab && /tmp/gemm_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__device__ long long* g_trace;
#define SLN_TRACE(i) do { if (threadIdx.x == 0) g_trace[(size_t)bid * 8 + (i)] = clock64(); } while (0)
#include "gemm_bodies.h"
#include "../../3d_sln_amd/csrc/prof.hip"

template <int BM, int BN, int AMODE, int EPI>
__global__ __launch_bounds__(256) void lab_nt(const GemmNTArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int bid = blockIdx.x;
  gemm_nt_body<BM, BN, 2, 2, AMODE, EPI, 0>(a, bid, gridDim.x, smem);
}

template <int J, int AMODE, int EPI>
__global__ __launch_bounds__(256) void lab_nt16(const GemmNTArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int bid = blockIdx.x;
  gemm_nt_body16<J, AMODE, EPI>(a, bid, gridDim.x, smem);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
  const int shapes[][3] = {{4096, 256, 384}, {4096, 256, 640}, {4096, 640, 256}, {4096, 384, 256}, {2048, 256, 256}, {2048, 128, 256}, {2048, 256, 128}, {32768, 640, 256}};
  long long* trace; CK(hipMalloc(&trace, sizeof(long long) * 8 * 65536));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &trace, sizeof(trace)));
  for (auto& sh : shapes) {
    const int M = sh[0], N = sh[1], K = sh[2];
    float *x, *W, *b, *y; double* sums;
    CK(hipMalloc(&x, sizeof(float) * M * K)); CK(hipMalloc(&W, sizeof(float) * N * K)); CK(hipMalloc(&b, sizeof(float) * N));
    CK(hipMalloc(&y, sizeof(float) * M * N)); CK(hipMalloc(&sums, sizeof(double) * 2 * N));
    CK(hipMemset(x, 0, sizeof(float) * M * K)); CK(hipMemset(W, 0, sizeof(float) * N * K)); CK(hipMemset(b, 0, sizeof(float) * N));
    CK(hipMemset(sums, 0, sizeof(double) * 2 * N));
    GemmNTArgs a; memset(&a, 0, sizeof(a));
    Seg s; memset(&s, 0, sizeof(s)); s.x1 = x; s.ld1 = K; s.len = K; s.coef = SLN_COEF_IDENT;
    a.A.seg[0] = s; a.A.nseg = 1; a.A.rows = M; a.A.cols = K;
    a.W = W; a.bias = b; a.Y = y; a.ldy = N; a.M = M; a.N = N; a.K = K; a.ldw = K; a.osums = sums; a.ocstride = N;
    const size_t smem = nt_smem_bytes(K, 64, 64, 2);
    const int grid = ((M + 63) / 64) * ((N + 63) / 64);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lab_nt<64, 64, 2, EPI_STATS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((lab_nt<64, 64, 2, EPI_STATS>), dim3(grid), dim3(256), smem, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((lab_nt<64, 64, 2, EPI_STATS>), dim3(grid), dim3(256), smem, 0, a);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    {   // two independent launches of the same problem side by side (two streams): does co-residency hide the per-block latencies?
      hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
      float* y2; CK(hipMalloc(&y2, sizeof(float) * M * N)); double* sums2; CK(hipMalloc(&sums2, sizeof(double) * 2 * N));
      CK(hipMemset(sums2, 0, sizeof(double) * 2 * N));
      GemmNTArgs a2 = a; a2.Y = y2; a2.osums = sums2;
      CK(hipDeviceSynchronize());
      hipEvent_t f0, f1, f2; CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1)); CK(hipEventCreate(&f2));
      CK(hipEventRecord(f0, s1)); CK(hipStreamWaitEvent(s2, f0, 0));
      for (int i = 0; i < 50; ++i) {
        hipLaunchKernelGGL((lab_nt<64, 64, 2, EPI_STATS>), dim3(grid), dim3(256), smem, s1, a);
        hipLaunchKernelGGL((lab_nt<64, 64, 2, EPI_STATS>), dim3(grid), dim3(256), smem, s2, a2);
      }
      CK(hipEventRecord(f1, s1)); CK(hipEventRecord(f2, s2)); CK(hipDeviceSynchronize());
      float m1, m2; CK(hipEventElapsedTime(&m1, f0, f1)); CK(hipEventElapsedTime(&m2, f0, f2));
      printf("   two streams side by side: %.2f us per PAIR of launches\n", (m1 > m2 ? m1 : m2) / 50 * 1e3);
      (void)hipFree(y2); (void)hipFree(sums2);
    }
    std::vector<long long> t(8 * grid);
    CK(hipMemcpy(t.data(), trace, sizeof(long long) * 8 * grid, hipMemcpyDeviceToHost));
    long long tmin = t[0], tmax = 0;
    std::vector<long long> ph[4];
    for (int b2 = 0; b2 < grid; ++b2) {
      tmin = std::min(tmin, t[8 * b2]); tmax = std::max(tmax, t[8 * b2 + 4]);
      for (int p = 0; p < 4; ++p) ph[p].push_back(t[8 * b2 + p + 1] - t[8 * b2 + p]);
    }
    for (auto& v : ph) std::sort(v.begin(), v.end());
    auto med = [&](std::vector<long long>& v) { return v[v.size() / 2]; };
    std::vector<long long> starts; for (int b2 = 0; b2 < grid; ++b2) starts.push_back(t[8 * b2] - tmin);
    std::sort(starts.begin(), starts.end());
    printf("M=%5d N=%4d K=%4d grid %4d: %6.2f us/launch (%.1f TF) | span %lld ticks; median ticks: coef/prologue %lld, first tile %lld, main loop %lld (%d k-tiles), epilogue %lld; block start p50 %lld p100 %lld\n",
           M, N, K, grid, ms / 50 * 1e3, 2.0 * M * N * K / (ms / 50 * 1e-3) / 1e12, tmax - tmin, med(ph[0]), med(ph[1]), med(ph[2]), (K + 31) / 32, med(ph[3]),
           starts[starts.size() / 2], starts.back());
    {   // the same problem with a train-mode BatchNorm + ReLU operand (AMODE 0): what does the coefficient set-up add to the prologue?
      float *gamma, *beta; double* bsums;
      CK(hipMalloc(&gamma, sizeof(float) * K)); CK(hipMalloc(&beta, sizeof(float) * K)); CK(hipMalloc(&bsums, sizeof(double) * 2 * K));
      std::vector<float> hg(K, 1.0f), hb(K, 0.1f); std::vector<double> hs(2 * K);
      for (int c = 0; c < K; ++c) { hs[c] = 0.5 * M; hs[K + c] = 1.25 * M; }
      CK(hipMemcpy(gamma, hg.data(), sizeof(float) * K, hipMemcpyHostToDevice)); CK(hipMemcpy(beta, hb.data(), sizeof(float) * K, hipMemcpyHostToDevice));
      CK(hipMemcpy(bsums, hs.data(), sizeof(double) * 2 * K, hipMemcpyHostToDevice));
      GemmNTArgs ab = a;
      ab.A.seg[0].coef = SLN_COEF_FWD;
      BnView v; memset(&v, 0, sizeof(v));
      v.sums = bsums; v.gamma = gamma; v.beta = beta; v.cstride = K; v.mode = SLN_BN_TRAIN; v.n_rows = (float)M; v.eps = 1e-5f; v.rn = 1.0 / M;
      ab.A.seg[0].bn = v;
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lab_nt<64, 64, 0, EPI_STATS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((lab_nt<64, 64, 0, EPI_STATS>), dim3(grid), dim3(256), smem, 0, ab);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((lab_nt<64, 64, 0, EPI_STATS>), dim3(grid), dim3(256), smem, 0, ab);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float msb; CK(hipEventElapsedTime(&msb, e0, e1));
      std::vector<long long> tb(8 * grid);
      CK(hipMemcpy(tb.data(), trace, sizeof(long long) * 8 * grid, hipMemcpyDeviceToHost));
      std::vector<long long> phb[4];
      for (int b2 = 0; b2 < grid; ++b2) for (int p = 0; p < 4; ++p) phb[p].push_back(tb[8 * b2 + p + 1] - tb[8 * b2 + p]);
      for (auto& v2 : phb) std::sort(v2.begin(), v2.end());
      printf("   BatchNorm operand: %6.2f us/launch; median ticks: prologue %lld, first tile %lld, main loop %lld, epilogue %lld\n", msb / 50 * 1e3,
             med(phb[0]), med(phb[1]), med(phb[2]), med(phb[3]));
      (void)hipFree(gamma); (void)hipFree(beta); (void)hipFree(bsums);
    }
    if (N % 160 == 0 || N % 96 == 0) {
      const int J = N % 160 == 0 ? 5 : 3;
      const size_t sm16 = nt16_smem_bytes(K, J);
      const int g16 = ((M + 63) / 64) * (N / (32 * J));
      auto launch = [&]() {
        if (J == 5) hipLaunchKernelGGL((lab_nt16<5, 2, EPI_STATS>), dim3(g16), dim3(256), sm16, 0, a);
        else hipLaunchKernelGGL((lab_nt16<3, 2, EPI_STATS>), dim3(g16), dim3(256), sm16, 0, a);
      };
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lab_nt16<5, 2, EPI_STATS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lab_nt16<3, 2, EPI_STATS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      for (int i = 0; i < 5; ++i) launch();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int i = 0; i < 50; ++i) launch();
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float ms16; CK(hipEventElapsedTime(&ms16, e0, e1));
      std::vector<long long> t16(8 * g16);
      CK(hipMemcpy(t16.data(), trace, sizeof(long long) * 8 * g16, hipMemcpyDeviceToHost));
      std::vector<long long> ph16[4];
      for (int b2 = 0; b2 < g16; ++b2) for (int p = 0; p < 4; ++p) ph16[p].push_back(t16[8 * b2 + p + 1] - t16[8 * b2 + p]);
      for (auto& v : ph16) std::sort(v.begin(), v.end());
      printf("   16x16 body J=%d grid %d: %6.2f us/launch (%.1f TF); median ticks: prologue %lld, first tile %lld, main loop %lld, epilogue %lld\n", J, g16,
             ms16 / 50 * 1e3, 2.0 * M * N * K / (ms16 / 50 * 1e-3) / 1e12, med(ph16[0]), med(ph16[1]), med(ph16[2]), med(ph16[3]));
    }
    hipFree(x); hipFree(W); hipFree(b); hipFree(y); hipFree(sums);
  }
  return 0;
}
