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
// the product's kernel (vae_kernels.hip::adam_kernel), argument for argument
struct AdamScalars { int64_t step; float lr, beta1, beta2, eps; float kl_weight; float bc1, bc2; int skip, pad_;
                     unsigned long long rng_seed, rng_offset; unsigned int rng_done, adam_done; };
template <int VARIANT>        // 0: as shipped, 1: without the pow() calls, 2: without the arrival ticket, 3: neither
__global__ void k_product(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                          float* __restrict__ v, long n, AdamScalars* __restrict__ sc, const float* __restrict__ total_loss) {
  const bool skip = total_loss != nullptr && !isfinite(total_loss[0]);
  const int64_t step = sc->step + 1;
  const float b1 = sc->beta1, b2 = sc->beta2, eps = sc->eps;
  float bc1, bc2;
  if (VARIANT & 1) { bc1 = 1.0f - b1 * (float)step; bc2 = 1.0f - b2 * (float)step; }
  else { bc1 = (float)(1.0 - pow((double)b1, (double)step)); bc2 = (float)(1.0 - pow((double)b2, (double)step)); }
  const float step_size = sc->lr / bc1, rs2 = 1.0f / sqrtf(bc2);
  if (!skip)
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
      const float gi = g[i];
      const float mi = b1 * m[i] + (1.f - b1) * gi;
      const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
      m[i] = mi; v[i] = vi;
      p[i] -= step_size * mi / (sqrtf(vi) * rs2 + eps);
    }
  if (VARIANT & 2) return;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned int done = atomicAdd(&sc->adam_done, 1u);
    if (done == gridDim.x - 1) {
      sc->adam_done = 0; sc->skip = skip ? 1 : 0;
      if (!skip) { sc->step = step; sc->bc1 = bc1; sc->bc2 = bc2; }
    }
  }
}

int main() {
  const long n = 3700000 / 4 * 4;
  // COLD operands: twelve sets (710 MB, more than the 256 MB infinity cache) taken in turn - in the training step two milliseconds
  // of other traffic pass between two updates
  constexpr int SETS = 12;
  float* base; hipMalloc(&base, 4 * n * 4 * SETS); hipMemset(base, 0, 4 * n * 4 * SETS);
  int turn = 0;
  float *p = base, *g = base + n, *m = base + 2 * n, *v = base + 3 * n;
  auto next = [&] { turn = (turn + 1) % SETS; p = base + (long)turn * 4 * n; g = p + n; m = p + 2 * n; v = p + 3 * n; };
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) { next(); launch(); }
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) { next(); launch(); }
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %7.2f us  %6.2f TB/s\n", name, ms / 20 * 1e3, 28.0 * n / (ms / 20 * 1e-3) / 1e12);
  };
  AdamScalars hs{}; hs.step = 10; hs.lr = 1e-4f; hs.beta1 = 0.9f; hs.beta2 = 0.999f; hs.eps = 1e-8f;
  AdamScalars* sc; hipMalloc(&sc, sizeof(hs)); hipMemcpy(sc, &hs, sizeof(hs), hipMemcpyHostToDevice);
  float* tl; hipMalloc(&tl, 4); hipMemset(tl, 0, 4);
  time("product kernel, 2048 x 256", [&] { hipLaunchKernelGGL(k_product<0>, dim3(2048), dim3(256), 0, 0, p, g, m, v, n, sc, tl); });
  time("  without pow()", [&] { hipLaunchKernelGGL(k_product<1>, dim3(2048), dim3(256), 0, 0, p, g, m, v, n, sc, tl); });
  time("  without the ticket", [&] { hipLaunchKernelGGL(k_product<2>, dim3(2048), dim3(256), 0, 0, p, g, m, v, n, sc, tl); });
  time("  without both", [&] { hipLaunchKernelGGL(k_product<3>, dim3(2048), dim3(256), 0, 0, p, g, m, v, n, sc, tl); });
  for (int blocks : {2048, 8192}) {
    char nm[96];
    snprintf(nm, sizeof nm, "scalar, %d x 256", blocks); time(nm, [&] { hipLaunchKernelGGL(k_scalar, dim3(blocks), dim3(256), 0, 0, p, g, m, v, n, 1e-3f); });
    snprintf(nm, sizeof nm, "float4, %d x 256", blocks); time(nm, [&] { hipLaunchKernelGGL((k_vec4<false>), dim3(blocks), dim3(256), 0, 0, p, g, m, v, n / 4, 1e-3f); });
    snprintf(nm, sizeof nm, "float4 nontemporal, %d x 256", blocks); time(nm, [&] { hipLaunchKernelGGL((k_vec4<true>), dim3(blocks), dim3(256), 0, 0, p, g, m, v, n / 4, 1e-3f); });
    snprintf(nm, sizeof nm, "float4 rcp / sqrt instructions, %d x 256", blocks); time(nm, [&] { hipLaunchKernelGGL(k_vec4_fast, dim3(blocks), dim3(256), 0, 0, p, g, m, v, n / 4, 1e-3f); });
    snprintf(nm, sizeof nm, "float4 adds only (traffic floor), %d x 256", blocks); time(nm, [&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, p, g, m, v, n / 4, 1e-3f); });
  }
  return 0;
}
