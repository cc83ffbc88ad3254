// Streaming variants of the Adam update (GPU box only): 4 reads + 3 writes of fp32 per element, n = 3.7 M (the VAE's parameter count).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lab/adam_lab.hip -o /tmp/adam_lab && /tmp/adam_lab
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void upd(float& p, float g, float& m, float& v, float b1, float b2, float ss, float rs2, float eps) {
  m = b1 * m + (1.f - b1) * g; v = b2 * v + (1.f - b2) * g * g; p -= ss * m / (sqrtf(v) * rs2 + eps);
}
__global__ void k_scalar(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n, float ss) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float pi = p[i], mi = m[i], vi = v[i];
    upd(pi, g[i], mi, vi, 0.9f, 0.999f, ss, 1.01f, 1e-8f);
    m[i] = mi; v[i] = vi; p[i] = pi;
  }
}
template <bool NT>
__global__ void k_vec4(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n4, float ss) {
  v4f* p4 = (v4f*)p; const v4f* g4 = (const v4f*)g; v4f* m4 = (v4f*)m; v4f* v4 = (v4f*)v;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    v4f pi = NT ? __builtin_nontemporal_load(p4 + i) : p4[i], gi = NT ? __builtin_nontemporal_load(g4 + i) : g4[i];
    v4f mi = NT ? __builtin_nontemporal_load(m4 + i) : m4[i], vi = NT ? __builtin_nontemporal_load(v4 + i) : v4[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) { float a = pi[k], b = mi[k], c = vi[k]; upd(a, gi[k], b, c, 0.9f, 0.999f, ss, 1.01f, 1e-8f); pi[k] = a; mi[k] = b; vi[k] = c; }
    if (NT) { __builtin_nontemporal_store(mi, m4 + i); __builtin_nontemporal_store(vi, v4 + i); __builtin_nontemporal_store(pi, p4 + i); }
    else { m4[i] = mi; v4[i] = vi; p4[i] = pi; }
  }
}
// fast reciprocal / rsqrt instead of the IEEE division and square root (is the kernel bound by its arithmetic?)
__global__ void k_vec4_fast(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n4, float ss) {
  v4f* p4 = (v4f*)p; const v4f* g4 = (const v4f*)g; v4f* m4 = (v4f*)m; v4f* v4 = (v4f*)v;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    v4f pi = p4[i], gi = g4[i], mi = m4[i], vi = v4[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      mi[k] = 0.9f * mi[k] + 0.1f * gi[k]; vi[k] = 0.999f * vi[k] + 0.001f * gi[k] * gi[k];
      pi[k] -= ss * mi[k] * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(vi[k]) * 1.01f + 1e-8f);
    }
    m4[i] = mi; v4[i] = vi; p4[i] = pi;
  }
}
__global__ void k_copy(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n4, float ss) {
  v4f* p4 = (v4f*)p; const v4f* g4 = (const v4f*)g; v4f* m4 = (v4f*)m; v4f* v4 = (v4f*)v;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    v4f pi = p4[i], gi = g4[i], mi = m4[i], vi = v4[i];
    m4[i] = mi + gi; v4[i] = vi + gi; p4[i] = pi + gi * ss;
  }
}
int main() {
  const long n = 3700000 / 4 * 4;
  float *p, *g, *m, *v;
  hipMalloc(&p, 4 * n); hipMalloc(&g, 4 * n); hipMalloc(&m, 4 * n); hipMalloc(&v, 4 * n);
  hipMemset(p, 0, 4 * n); hipMemset(g, 0, 4 * n); hipMemset(m, 0, 4 * n); hipMemset(v, 0, 4 * n);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %7.2f us  %6.2f TB/s\n", name, ms / 20 * 1e3, 28.0 * n / (ms / 20 * 1e-3) / 1e12);
  };
  for (int blocks : {1024, 2048, 4096, 8192}) {
    char nm[96];
    snprintf(nm, sizeof nm, "scalar, %d x 256", blocks); time(nm, [&] { hipLaunchKernelGGL(k_scalar, dim3(blocks), dim3(256), 0, 0, p, g, m, v, n, 1e-3f); });
    snprintf(nm, sizeof nm, "float4, %d x 256", blocks); time(nm, [&] { hipLaunchKernelGGL((k_vec4<false>), dim3(blocks), dim3(256), 0, 0, p, g, m, v, n / 4, 1e-3f); });
    snprintf(nm, sizeof nm, "float4 nontemporal, %d x 256", blocks); time(nm, [&] { hipLaunchKernelGGL((k_vec4<true>), dim3(blocks), dim3(256), 0, 0, p, g, m, v, n / 4, 1e-3f); });
    snprintf(nm, sizeof nm, "float4 rcp / sqrt instructions, %d x 256", blocks); time(nm, [&] { hipLaunchKernelGGL(k_vec4_fast, dim3(blocks), dim3(256), 0, 0, p, g, m, v, n / 4, 1e-3f); });
    snprintf(nm, sizeof nm, "float4 adds only (traffic floor), %d x 256", blocks); time(nm, [&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, p, g, m, v, n / 4, 1e-3f); });
  }
  return 0;
}
