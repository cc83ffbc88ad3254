// HIP-event profiler behind sln_prof_enable / sln_prof_read (include/sln_hip.h).
#include <cstdlib>
#include <vector>
#include "../../include/sln_hip.h"
#include "sln_prof.h"

bool g_sln_prof_on = false;
int g_sln_deterministic = [] { const char* v = std::getenv("SLN_DETERMINISTIC"); return (v && v[0] == '1') ? 1 : 0; }();

namespace {
struct Rec { int family; double work; hipEvent_t a, b; };
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
hipEvent_t take_event() {
  if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
}  // namespace

void sln_prof_begin(int family, double work, hipStream_t st) {
  Rec r; r.family = family; r.work = work; r.a = take_event(); r.b = take_event();
  (void)hipEventRecord(r.a, st);
  g_recs.push_back(r);
}
void sln_prof_end(hipStream_t st) {
  if (!g_recs.empty()) (void)hipEventRecord(g_recs.back().b, st);
}

extern "C" int sln_set_deterministic(int on) { g_sln_deterministic = on != 0; return 0; }
extern "C" int sln_get_deterministic(void) { return g_sln_deterministic; }
extern "C" int sln_prof_enable(int enable) {
  g_sln_prof_on = enable != 0;
  return 0;
}

extern "C" int sln_prof_read(double* ms, double* work, int64_t* launches, int n) {
  if (!ms || !work || !launches || n <= 0) return SLN_E_BADARG;
  for (int i = 0; i < n; ++i) { ms[i] = 0.0; work[i] = 0.0; launches[i] = 0; }
  for (auto& r : g_recs) {
    float t = 0.f;
    hipError_t e = hipEventSynchronize(r.b);
    if (e == hipSuccess) e = hipEventElapsedTime(&t, r.a, r.b);
    if (e == hipSuccess && r.family >= 0 && r.family < n) { ms[r.family] += t; work[r.family] += r.work; launches[r.family] += 1; }
    g_pool.push_back(r.a); g_pool.push_back(r.b);
  }
  g_recs.clear();
  return 0;
}
