// Optional per-launch timing with HIP events (bench.py roofline figures).  Off by default; when on,
// every launcher brackets its kernel(s) with two events on the launch stream.
#pragma once
#include <hip/hip_runtime.h>

enum { SLN_FAM_GEMM_NT = 0, SLN_FAM_GEMM_TN = 1, SLN_FAM_EDGE = 2, SLN_FAM_OTHER = 3, SLN_FAM_RASTER = 4,
       SLN_FAM_RASTER_BWD = 5, SLN_FAM_CONV = 6, SLN_FAM_GEMM_DUAL = 7, SLN_FAM_COUNT = 8 };

extern bool g_sln_prof_on;
// SLN_DETERMINISTIC=1 / sln_set_deterministic(1): launchers pick the order-independent variants (see DESIGN.md "Deterministic mode")
extern int g_sln_deterministic;
void sln_prof_begin(int family, double work, hipStream_t st);
void sln_prof_end(hipStream_t st);

struct SlnProfScope {
  hipStream_t st; bool on;
  SlnProfScope(int family, double work, hipStream_t s) : st(s), on(g_sln_prof_on) { if (on) sln_prof_begin(family, work, s); }
  ~SlnProfScope() { if (on) sln_prof_end(st); }
};
