// What does v_mfma_f32_32x32x2_f32 round, and how?  (lab; GPU box)
//   hipcc --offload-arch=gfx950 -O2 tools/lab/mfma_round.hip -o /tmp/mfma_round && /tmp/mfma_round
// One wavefront computes C[32 x 32] = A[32 x K] . B[K x 32] as a chain of K/2 MFMAs and the host compares the bits with
// candidate models of the accumulation evaluated on the CPU in the same k order:
//   fma_rne   c = fmaf(a_k, b_k, c), round to nearest even, one k at a time          (an FMA chain)
//   fma_rtz   the same with round toward zero
//   pair_rne  c = RN(c + (a_k b_k + a_k+1 b_k+1)) with the pair exact (one rounding per MFMA)
//   pair_rtz  the same toward zero
//   mul_add   c = RN(c + RN(a_k b_k)), products rounded first
// and reports, against an fp64 evaluation, the rms / max error of the MFMA chain and of every model.
#include <hip/hip_runtime.h>
#include <cfenv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void mfma_chain(const float* __restrict__ A, const float* __restrict__ B, int K, float* __restrict__ C, int mode) {
  const int lane = threadIdx.x, li = lane & 31, lk = lane >> 5;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (mode == 0) {
    for (int k = 0; k < K; k += 2) {
      const float a = A[li * K + k + lk], b = B[(k + lk) * 32 + li];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  } else {
    // blocked: chunks of `mode` k values into a fresh accumulator, totals added in fp32 by the vector ALU
    f32x16 tot;
    for (int r = 0; r < 16; ++r) tot[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += mode) {
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      for (int k = k0; k < k0 + mode && k < K; k += 2) {
        const float a = A[li * K + k + lk], b = B[(k + lk) * 32 + li];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
      }
      for (int r = 0; r < 16; ++r) tot[r] += acc[r];
    }
    acc = tot;
  }
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * lk) * 32 + li] = acc[r];
}

// VALU chain for comparison: one thread per output element, fmaf in k order
__global__ void fma_chain(const float* __restrict__ A, const float* __restrict__ B, int K, float* __restrict__ C) {
  const int i = blockIdx.x, j = threadIdx.x;
  float c = 0.f;
  for (int k = 0; k < K; ++k) c = fmaf(A[i * K + k], B[k * 32 + j], c);
  C[i * 32 + j] = c;
}

static float rz_from_double(double x) {           // round a double toward zero to float
  float f = (float)x;                             // RNE
  if (std::fabs((double)f) > std::fabs(x)) f = std::nextafterf(f, 0.f);
  return f;
}

int main() {
  std::mt19937 rng(1234);
  std::normal_distribution<float> nd(0.f, 1.f);
  const int Ks[] = {2, 8, 64, 1152, 9216};
  for (int K : Ks) {
    std::vector<float> A(32 * K), B(K * 32), C(1024), Cb(1024), Cv(1024);
    for (auto& v : A) v = nd(rng);
    for (auto& v : B) v = nd(rng);
    float *dA, *dB, *dC;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mfma_chain, dim3(1), dim3(64), 0, 0, dA, dB, K, dC, 0);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(mfma_chain, dim3(1), dim3(64), 0, 0, dA, dB, K, dC, 72);
    hipMemcpy(Cb.data(), dC, 4096, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(fma_chain, dim3(32), dim3(32), 0, 0, dA, dB, K, dC);
    hipMemcpy(Cv.data(), dC, 4096, hipMemcpyDeviceToHost);
    if (hipDeviceSynchronize() != hipSuccess) { printf("hip error\n"); return 1; }

    const char* names[] = {"fma_rne", "fma_rtz", "pair_rne", "pair_rtz", "mul_add", "pair_rne_prodrn"};
    const int NM = 6;
    int match[NM] = {0};
    double se[NM + 3] = {0}, mx[NM + 3] = {0}, scale = 0;
    int match_valu = 0;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)A[i * K + k] * (double)B[k * 32 + j];
        scale = std::fmax(scale, std::fabs(ref));
        float m[NM];
        { float c = 0; for (int k = 0; k < K; ++k) c = fmaf(A[i * K + k], B[k * 32 + j], c); m[0] = c; }
        { float c = 0; for (int k = 0; k < K; ++k) c = rz_from_double((double)A[i * K + k] * (double)B[k * 32 + j] + (double)c); m[1] = c; }
        // (the product of two floats is exact in double; adding c (24 bits) may round in double for far-apart exponents: negligible here)
        { float c = 0; for (int k = 0; k < K; k += 2) c = (float)((double)A[i * K + k] * B[k * 32 + j] + (double)A[i * K + k + 1] * B[(k + 1) * 32 + j] + (double)c); m[2] = c; }
        { float c = 0; for (int k = 0; k < K; k += 2) c = rz_from_double((double)A[i * K + k] * B[k * 32 + j] + (double)A[i * K + k + 1] * B[(k + 1) * 32 + j] + (double)c); m[3] = c; }
        { float c = 0; for (int k = 0; k < K; ++k) { volatile float p = A[i * K + k] * B[k * 32 + j]; c = c + p; } m[4] = c; }
        { float c = 0; for (int k = 0; k < K; k += 2) { volatile float p0 = A[i * K + k] * B[k * 32 + j]; volatile float p1 = A[i * K + k + 1] * B[(k + 1) * 32 + j];
            c = (float)((double)p0 + (double)p1 + (double)c); } m[5] = c; }
        const float g = C[i * 32 + j];
        for (int q = 0; q < NM; ++q) {
          if (m[q] == g) match[q]++;
          const double e = (double)m[q] - ref; se[q] += e * e; mx[q] = std::fmax(mx[q], std::fabs(e));
        }
        const double e = (double)g - ref; se[NM] += e * e; mx[NM] = std::fmax(mx[NM], std::fabs(e));
        const double eb = (double)Cb[i * 32 + j] - ref; se[NM + 1] += eb * eb; mx[NM + 1] = std::fmax(mx[NM + 1], std::fabs(eb));
        const double ev = (double)Cv[i * 32 + j] - ref; se[NM + 2] += ev * ev; mx[NM + 2] = std::fmax(mx[NM + 2], std::fabs(ev));
        if (Cv[i * 32 + j] == m[0]) match_valu++;
      }
    printf("K=%d  scale %.3e   (errors relative to the scale; bit matches of the MFMA chain out of 1024)\n", K, scale);
    printf("  %-18s rms %.3e max %.3e\n", "MFMA chain", std::sqrt(se[NM] / 1024) / scale, mx[NM] / scale);
    printf("  %-18s rms %.3e max %.3e\n", "MFMA blocked/72", std::sqrt(se[NM + 1] / 1024) / scale, mx[NM + 1] / scale);
    printf("  %-18s rms %.3e max %.3e   (== host fmaf chain: %d)\n", "VALU fmaf chain", std::sqrt(se[NM + 2] / 1024) / scale, mx[NM + 2] / scale, match_valu);
    for (int q = 0; q < NM; ++q)
      printf("  %-18s rms %.3e max %.3e   bit-equal to MFMA: %4d\n", names[q], std::sqrt(se[q] / 1024) / scale, mx[q] / scale, match[q]);
    hipFree(dA); hipFree(dB); hipFree(dC);
  }
  return 0;
}
