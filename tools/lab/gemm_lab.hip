// Stand-alone GEMM laboratory (GPU box only): phase stamps of the product's NT body on the VAE's shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -I include -I 3d_sln_amd/csrc tools/lab/gemm_lab.hip -o /tmp/gemm_lab && /tmp/gemm_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__device__ long long* g_trace;
#define SLN_TRACE(i) do { if (threadIdx.x == 0) g_trace[(size_t)bid * 16 + (i)] = clock64(); } while (0)
#define SLN_TRACEH(i) do { if (threadIdx.x == 256) g_trace[(size_t)bid * 16 + (i)] = clock64(); } while (0)
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

template <int BM, int BN, int AMODE, int EPI>
__global__ __launch_bounds__(512) void lab_nt_help(const GemmNTArgs a) {        // with the four helper wavefronts
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int bid = blockIdx.x;
  gemm_nt_body<BM, BN, 2, 2, AMODE, EPI, 0, true>(a, bid, gridDim.x, smem);
}
// hot kernel arguments as leading scalars: with -mllvm -amdgpu-kernarg-preload-count=16 they arrive in SGPRs with the wave
template <int BM, int BN, int AMODE, int EPI>
__global__ __launch_bounds__(256) void lab_nt_pre(const float* x1, const float* W, const int* idx_a, const int* idx_b, int M, int N, int K, int ld1, int ldw,
                                                  int c1, int which, int nwg, const GemmNTArgs a0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  GemmNTArgs a = a0;
  a.A.seg[0].x1 = x1; a.W = W; a.A.idx_a = idx_a; a.A.idx_b = idx_b; a.M = M; a.N = N; a.K = K; a.A.seg[0].ld1 = ld1; a.ldw = ldw;
  a.A.seg[0].c1 = c1; a.A.seg[0].which = which;
  const int bid = blockIdx.x;
  gemm_nt_body<BM, BN, 2, 2, AMODE, EPI, 0>(a, bid, nwg, smem);
}
// helper-wavefront kernel with a BatchNorm operand: SET 0 preloads the operand chain's heads, SET 1 the coefficient chain's too
template <int BM, int BN, int AMODE, int EPI, int SET>
__global__ __launch_bounds__(512) void lab_nt_help_pre(const float* x1, const float* W, const void* p2, const void* p3, const void* p4, int M, int N, int K, int ld1,
                                                       const GemmNTArgs a0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  GemmNTArgs a = a0;
  a.A.seg[0].x1 = x1; a.W = W; a.M = M; a.N = N; a.K = K; a.A.seg[0].ld1 = ld1;
  if (SET == 0) { a.A.idx_a = (const int*)p2; a.A.idx_b = (const int*)p3; }
  else { a.A.seg[0].bn.sums = (const double*)p2; a.A.seg[0].bn.gamma = (const float*)p3; a.A.seg[0].bn.beta = (const float*)p4; }
  const int bid = blockIdx.x;
  gemm_nt_body<BM, BN, 2, 2, AMODE, EPI, 0, true>(a, bid, ((M + BM - 1) / BM) * ((N + BN - 1) / BN), smem);
}
// everything both chains need in front of the first barrier as leading scalars: 14 dwords arrive with the wave, the rest is adjacent
// in the argument block (one or two wide s_loads instead of a dozen narrow ones in three dependent stages)
template <int BM, int BN, int AMODE, int EPI>
__global__ __launch_bounds__(512) void lab_nt_help_hot(const float* x1, const float* W, const int* idx_a, const int* idx_b, const double* sums, const float* gamma, int M, int N,
                                                       const float* beta, const double* gsums, int K, int ld1, int ldw, int c1, int which, int len, int coef, int cstride,
                                                       int mode, float n_rows, float eps, int nseg, double rn, const GemmNTArgs a0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  GemmNTArgs a = a0;
  Seg& g = a.A.seg[0];
  g.x1 = x1; a.W = W; a.A.idx_a = idx_a; a.A.idx_b = idx_b; g.bn.sums = sums; g.bn.gamma = gamma; a.M = M; a.N = N;
  g.bn.beta = beta; g.bn.gsums = gsums; a.K = K; g.ld1 = ld1; a.ldw = ldw; g.c1 = c1; g.which = which; g.len = len; g.coef = coef; g.bn.cstride = cstride;
  g.bn.mode = mode; g.bn.n_rows = n_rows; g.bn.eps = eps; a.A.nseg = nseg; g.bn.rn = rn;
  const int bid = blockIdx.x;
  gemm_nt_body<BM, BN, 2, 2, AMODE, EPI, 0, true>(a, bid, ((M + BM - 1) / BM) * ((N + BN - 1) / BN), smem);
}
__global__ void touch_sums(double* s, int n) {     // what a producer's statistics epilogue does to the sums right before the consumer runs
  for (int i = threadIdx.x; i < n; i += blockDim.x) atomicAdd(s + i, 0.0);
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
  const int shapes[][3] = {{4096, 256, 384}, {4096, 256, 640}, {4096, 640, 256}, {4096, 384, 256}, {2048, 256, 256}, {2048, 128, 256}, {2048, 256, 128}, {32768, 640, 256}};
  long long* trace; CK(hipMalloc(&trace, sizeof(long long) * 16 * 65536));
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
    std::vector<long long> t(16 * grid);
    for (int variant = 0; variant < 2; ++variant) {      // A/B inside one process: struct argument vs leading scalars (preloaded when built with the flag)
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lab_nt_pre<64, 64, 2, EPI_STATS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 50; ++i) {
          if (variant == 0) hipLaunchKernelGGL((lab_nt<64, 64, 2, EPI_STATS>), dim3(grid), dim3(256), smem, 0, a);
          else hipLaunchKernelGGL((lab_nt_pre<64, 64, 2, EPI_STATS>), dim3(grid), dim3(256), smem, 0, a.A.seg[0].x1, a.W, a.A.idx_a, a.A.idx_b, a.M, a.N, a.K,
                                  a.A.seg[0].ld1, a.ldw, a.A.seg[0].c1, a.A.seg[0].which, grid, a);
        }
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float m2; CK(hipEventElapsedTime(&m2, e0, e1)); best = std::min(best, m2);
      }
      CK(hipMemcpy(t.data(), trace, sizeof(long long) * 16 * grid, hipMemcpyDeviceToHost));
      std::vector<long long> q[3];
      for (int b2 = 0; b2 < grid; ++b2) { q[0].push_back(t[16 * b2 + 5] - t[16 * b2]); q[1].push_back(t[16 * b2 + 1] - t[16 * b2]); q[2].push_back(t[16 * b2 + 4] - t[16 * b2]); }
      for (auto& v : q) std::sort(v.begin(), v.end());
      printf("   kernarg %s: %6.2f us/launch; median ticks: indices %lld, whole prologue %lld, block %lld\n", variant ? "leading scalars" : "struct         ", best / 50 * 1e3,
             q[0][grid / 2], q[1][grid / 2], q[2][grid / 2]);
    }
    {   // the same launch without the column statistics (EPI_PLAIN): what do the fp64 atomics of the epilogue cost?
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lab_nt<64, 64, 2, EPI_PLAIN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((lab_nt<64, 64, 2, EPI_PLAIN>), dim3(grid), dim3(256), smem, 0, a);
      CK(hipEventRecord(e0));
      for (int i = 0; i < 50; ++i) hipLaunchKernelGGL((lab_nt<64, 64, 2, EPI_PLAIN>), dim3(grid), dim3(256), smem, 0, a);
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      float m3; CK(hipEventElapsedTime(&m3, e0, e1));
      CK(hipMemcpy(t.data(), trace, sizeof(long long) * 16 * grid, hipMemcpyDeviceToHost));
      std::vector<long long> q;
      for (int b2 = 0; b2 < grid; ++b2) q.push_back(t[16 * b2 + 4] - t[16 * b2 + 3]);
      std::sort(q.begin(), q.end());
      printf("   without column statistics: %6.2f us/launch, epilogue %lld ticks\n", m3 / 50 * 1e3, q[grid / 2]);
    }
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((lab_nt<64, 64, 2, EPI_STATS>), dim3(grid), dim3(256), smem, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(t.data(), trace, sizeof(long long) * 16 * grid, hipMemcpyDeviceToHost));
    long long tmin = t[0], tmax = 0;
    std::vector<long long> ph[4];
    for (int b2 = 0; b2 < grid; ++b2) {
      tmin = std::min(tmin, t[16 * b2]); tmax = std::max(tmax, t[16 * b2 + 4]);
      for (int p = 0; p < 4; ++p) ph[p].push_back(t[16 * b2 + p + 1] - t[16 * b2 + p]);
    }
    for (auto& v : ph) std::sort(v.begin(), v.end());
    auto med = [&](std::vector<long long>& v) { return v[v.size() / 2]; };
    {
      std::vector<long long> q[4];
      for (int b2 = 0; b2 < grid; ++b2) {
        q[0].push_back(t[16 * b2 + 5] - t[16 * b2]); q[1].push_back(t[16 * b2 + 6] - t[16 * b2 + 5]);
        q[2].push_back(t[16 * b2 + 7] - t[16 * b2 + 6]); q[3].push_back(t[16 * b2 + 1] - t[16 * b2 + 7]);
      }
      for (auto& v : q) std::sort(v.begin(), v.end());
      printf("   prologue pieces: indices %lld, issue of two tiles' loads %lld, tables + bias %lld, barrier %lld\n", med(q[0]), med(q[1]), med(q[2]), med(q[3]));
    }
    std::vector<long long> starts; for (int b2 = 0; b2 < grid; ++b2) starts.push_back(t[16 * b2] - tmin);
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
      std::vector<long long> tb(16 * grid);
      CK(hipMemcpy(tb.data(), trace, sizeof(long long) * 16 * grid, hipMemcpyDeviceToHost));
      std::vector<long long> phb[4];
      for (int b2 = 0; b2 < grid; ++b2) for (int p = 0; p < 4; ++p) phb[p].push_back(tb[16 * b2 + p + 1] - tb[16 * b2 + p]);
      for (auto& v2 : phb) std::sort(v2.begin(), v2.end());
      printf("   BatchNorm operand: %6.2f us/launch; median ticks: prologue %lld, first tile %lld, main loop %lld, epilogue %lld\n", msb / 50 * 1e3,
             med(phb[0]), med(phb[1]), med(phb[2]), med(phb[3]));
      for (int variant = 0; variant < 4; ++variant) {      // helper-wavefront kernel: struct / operand heads preloaded / coefficient heads preloaded too
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lab_nt_help<64, 64, 0, EPI_STATS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lab_nt_help_pre<64, 64, 0, EPI_STATS, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lab_nt_help_pre<64, 64, 0, EPI_STATS, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lab_nt_help_hot<64, 64, 0, EPI_STATS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipEventRecord(e0));
          for (int i = 0; i < 50; ++i) {
            const Seg& s0 = ab.A.seg[0];
            if (variant == 0) hipLaunchKernelGGL((lab_nt_help<64, 64, 0, EPI_STATS>), dim3(grid), dim3(512), smem, 0, ab);
            else if (variant == 1) hipLaunchKernelGGL((lab_nt_help_pre<64, 64, 0, EPI_STATS, 0>), dim3(grid), dim3(512), smem, 0, s0.x1, ab.W, (const void*)ab.A.idx_a,
                                                      (const void*)ab.A.idx_b, (const void*)nullptr, ab.M, ab.N, ab.K, s0.ld1, ab);
            else if (variant == 2) hipLaunchKernelGGL((lab_nt_help_pre<64, 64, 0, EPI_STATS, 1>), dim3(grid), dim3(512), smem, 0, s0.x1, ab.W, (const void*)s0.bn.sums, (const void*)s0.bn.gamma,
                                    (const void*)s0.bn.beta, ab.M, ab.N, ab.K, s0.ld1, ab);
            else hipLaunchKernelGGL((lab_nt_help_hot<64, 64, 0, EPI_STATS>), dim3(grid), dim3(512), smem, 0, s0.x1, ab.W, ab.A.idx_a, ab.A.idx_b, s0.bn.sums, s0.bn.gamma, ab.M, ab.N,
                                    s0.bn.beta, s0.bn.gsums, ab.K, s0.ld1, ab.ldw, s0.c1, s0.which, s0.len, s0.coef, s0.bn.cstride, s0.bn.mode, s0.bn.n_rows, s0.bn.eps,
                                    ab.A.nseg, s0.bn.rn, ab);
          }
          CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
          float m2; CK(hipEventElapsedTime(&m2, e0, e1)); best = std::min(best, m2);
        }
        CK(hipMemcpy(tb.data(), trace, sizeof(long long) * 16 * grid, hipMemcpyDeviceToHost));
        std::vector<long long> q[2];
        for (int b2 = 0; b2 < grid; ++b2) { q[0].push_back(tb[16 * b2 + 1] - tb[16 * b2]); q[1].push_back(tb[16 * b2 + 4] - tb[16 * b2]); }
        for (auto& v : q) std::sort(v.begin(), v.end());
        printf("   kernarg, BatchNorm operand + helpers, %s: %6.2f us/launch; median ticks: whole prologue %lld, block %lld\n",
               variant == 0 ? "struct             " : (variant == 1 ? "operand heads      " : (variant == 2 ? "coefficient heads  " : "all hot fields     ")), best / 50 * 1e3, q[0][grid / 2], q[1][grid / 2]);
      }
      for (int variant = 0; variant < 2; ++variant) {      // sums freshly updated by atomics of the previous launch; then the same with helper wavefronts
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lab_nt_help<64, 64, 0, EPI_STATS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        for (int i = 0; i < 20; ++i) {
          hipLaunchKernelGGL(touch_sums, dim3(256), dim3(256), 0, 0, bsums, 2 * K);
          if (variant == 0) hipLaunchKernelGGL((lab_nt<64, 64, 0, EPI_STATS>), dim3(grid), dim3(256), smem, 0, ab);
          else hipLaunchKernelGGL((lab_nt_help<64, 64, 0, EPI_STATS>), dim3(grid), dim3(512), smem, 0, ab);
        }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(tb.data(), trace, sizeof(long long) * 16 * grid, hipMemcpyDeviceToHost));
        for (auto& v2 : phb) v2.clear();
        for (int b2 = 0; b2 < grid; ++b2) for (int p = 0; p < 4; ++p) phb[p].push_back(tb[16 * b2 + p + 1] - tb[16 * b2 + p]);
        for (auto& v2 : phb) std::sort(v2.begin(), v2.end());
        {
          std::vector<long long> q[4];
          for (int b2 = 0; b2 < grid; ++b2) {
            q[0].push_back(tb[16 * b2 + 5] - tb[16 * b2]); q[1].push_back(tb[16 * b2 + 6] - tb[16 * b2 + 5]);
            q[2].push_back(tb[16 * b2 + 7] - tb[16 * b2 + 6]); q[3].push_back(tb[16 * b2 + 1] - tb[16 * b2 + 7]);
          }
          for (auto& v : q) std::sort(v.begin(), v.end());
          printf("   prologue pieces: indices %lld, issue of two tiles' loads %lld, tables + bias %lld, barrier %lld\n", med(q[0]), med(q[1]), med(q[2]), med(q[3]));
        }
        if (variant == 1) {
          std::vector<long long> q[5];
          for (int b2 = 0; b2 < grid; ++b2) {
            const long long* r = &tb[16 * b2];
            q[0].push_back(r[8] - r[0]); q[1].push_back(r[9] - r[0]); q[2].push_back(r[10] - r[0]); q[3].push_back(r[7] - r[0]); q[4].push_back(r[1] - r[0]);
          }
          for (auto& v : q) std::sort(v.begin(), v.end());
          printf("   since the block's first stamp: helper wave starts %lld, tables written %lld, helper past the barrier %lld | staging waves reach the barrier %lld, pass it %lld\n",
                 med(q[0]), med(q[1]), med(q[2]), med(q[3]), med(q[4]));
        }
        printf("   %s: median ticks: prologue %lld, first tile %lld, main loop %lld, epilogue %lld\n",
               variant == 0 ? "sums touched by atomics just before" : "the same with helper wavefronts    ", med(phb[0]), med(phb[1]), med(phb[2]), med(phb[3]));
      }
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
      std::vector<long long> t16(16 * g16);
      CK(hipMemcpy(t16.data(), trace, sizeof(long long) * 16 * g16, hipMemcpyDeviceToHost));
      std::vector<long long> ph16[4];
      for (int b2 = 0; b2 < g16; ++b2) for (int p = 0; p < 4; ++p) ph16[p].push_back(t16[16 * b2 + p + 1] - t16[16 * b2 + p]);
      for (auto& v : ph16) std::sort(v.begin(), v.end());
      printf("   16x16 body J=%d grid %d: %6.2f us/launch (%.1f TF); median ticks: prologue %lld, first tile %lld, main loop %lld, epilogue %lld\n", J, g16,
             ms16 / 50 * 1e3, 2.0 * M * N * K / (ms16 / 50 * 1e-3) / 1e12, med(ph16[0]), med(ph16[1]), med(ph16[2]), med(ph16[3]));
    }
    hipFree(x); hipFree(W); hipFree(b); hipFree(y); hipFree(sums);
  }
  return 0;
}
