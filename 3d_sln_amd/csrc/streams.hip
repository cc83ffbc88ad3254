// Side streams that really run next to the caller's stream (round 5).
//
// The runtime deals a process's streams to a few hardware queues round-robin (GPU_MAX_HW_QUEUES, 4 by default), and two streams
// on one hardware queue do not overlap: their packets are ordered.  A side stream created as the n-th stream of a process can
// therefore be worthless - the 16-room refinement ran 1.63 ms per iteration in a fresh process and 1.84 behind another leg of
// bench.py that had created streams of its own (1.63 again with GPU_MAX_HW_QUEUES=8, 1.84 in the fresh process with =2).
// High-priority side streams (another priority level has queues of its own) fixed that case and broke others: a hipGraph replay
// of the same iteration went from 1.76 to 4.2 ms and eager launches of the process stayed slow afterwards.
//
// So the library keeps a small pool of plain streams per device and, per caller stream, PROBES which of them overlaps with it:
// a kernel on the caller's stream waits (at most ~150 us) for a flag that a kernel on the candidate sets - if the candidate
// shares the caller's hardware queue its kernel sits behind the waiting one and the wait times out.  One probe per (caller
// stream, candidate), a few hundred microseconds once; never inside a stream capture (a capture of a stream not probed before
// takes the pool's first stream).  No candidate overlaps (one hardware queue): no side stream at all - the callers run their
// side work on their own stream.  SLN_SIDE_PROBE=0: no probing, first stream of the pool.
//
// Round 6 hardening: (1) a pick ages - after REPROBE_AFTER look-ups the stream is probed again (a destroyed and re-created stream can
// come back under the same handle on another hardware queue); a handle the runtime no longer knows (hipStreamQuery fails) loses its
// entry; sln_side_stream_forget drops one explicitly.  (2) A probe that finds NO overlapping candidate is repeated once with ten
// times the budget before "one hardware queue" is concluded (on a busy GPU the candidate's kernel may simply be late).  (3) The
// probe is a host synchronisation of the caller's stream (hipStreamSynchronize + a ~150 us spin kernel): sln_side_stream_prepare does
// it at a time of the caller's choosing, so that the first sln_scene_backward / sln_vae_group_* call on a stream stays asynchronous.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <map>
#include <mutex>
#include <utility>

#include "sln_common.h"

namespace {

constexpr int POOL = 4;
constexpr long long PROBE_TICKS = 15000;          // wall_clock64 runs at 100 MHz: 150 us
constexpr unsigned REPROBE_AFTER = 4096;          // look-ups after which a cached pick is probed again (~0.5 ms once per ~1 000 iterations)

__global__ void probe_wait_kernel(int* flag, int* result, long long max_ticks) {
  const long long t0 = wall_clock64();
  int seen = 0;
  while (!(seen = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) && wall_clock64() - t0 < max_ticks) __builtin_amdgcn_s_sleep(16);
  result[0] = seen ? 1 : 0;
}
__global__ void probe_set_kernel(int* flag) { __hip_atomic_store(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct DevPool {
  hipStream_t s[POOL] = {};
  int n = 0;
  int* words = nullptr;                            // flag, result
  struct Pick { int idx; unsigned age; };
  std::map<hipStream_t, Pick> pick;                // caller stream -> index into s (-1: no stream of the pool overlaps with it), look-ups since the probe
  bool any_ok = false;                             // some probe of this process has seen two streams overlap
};
std::mutex g_mu;
DevPool g_pools[64];

bool overlaps(DevPool& p, hipStream_t main, hipStream_t cand, long long ticks = PROBE_TICKS) {
  if (hipMemsetAsync(p.words, 0, 2 * sizeof(int), main) != hipSuccess) return false;
  if (hipStreamSynchronize(main) != hipSuccess) return false;
  hipLaunchKernelGGL(probe_wait_kernel, dim3(1), dim3(1), 0, main, p.words, p.words + 1, ticks);
  hipLaunchKernelGGL(probe_set_kernel, dim3(1), dim3(1), 0, cand, p.words);
  int res = 0;
  if (hipStreamSynchronize(main) != hipSuccess || hipStreamSynchronize(cand) != hipSuccess) return false;
  if (hipMemcpy(&res, p.words + 1, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return false;
  (void)hipGetLastError();
  return res == 1;
}

}  // namespace

hipStream_t sln_overlapping_stream(hipStream_t main) {
  static const bool no_probe = std::getenv("SLN_SIDE_PROBE") != nullptr && std::getenv("SLN_SIDE_PROBE")[0] == '0';
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(g_mu);
  DevPool& p = g_pools[dev];
  const bool capturing = sln_capturing(main);
  if (p.n == 0) {
    if (capturing) return nullptr;                 // nothing may be created while the caller records: no side stream this time
    for (int i = 0; i < POOL; ++i) {
      hipStream_t s = nullptr;
      if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) break;
      p.s[p.n++] = s;
    }
    if (p.n == 0) return nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&p.words), 2 * sizeof(int)) != hipSuccess) p.words = nullptr;
  }
  auto it = p.pick.find(main);
  if (it != p.pick.end()) {
    if (capturing || ++it->second.age < REPROBE_AFTER) return it->second.idx >= 0 ? p.s[it->second.idx] : nullptr;
    p.pick.erase(it);                              // aged: probed again below
  }
  if (no_probe || p.words == nullptr) return p.s[0];
  if (!capturing && main != nullptr) {             // a handle the runtime does not know any more: no entry, no side stream
    const hipError_t q = hipStreamQuery(main);
    if (q != hipSuccess && q != hipErrorNotReady) { (void)hipGetLastError(); return nullptr; }
  }
  // a stream met for the first time while it is being captured cannot be probed: it gets the pool's first stream - unless no probe
  // of this process has ever seen an overlap (a single hardware queue: GPU_MAX_HW_QUEUES=1; a fork then buys nothing, and the
  // runtime of this image crashes on a captured cross-stream fork in that configuration)
  if (capturing) return p.any_ok ? p.s[0] : nullptr;
  int chosen = -1;
  for (int pass = 0; pass < 2 && chosen < 0; ++pass)           // second pass: ten times the budget (a busy GPU starts the candidate's kernel late)
    for (int i = 0; i < p.n; ++i)
      if (overlaps(p, main, p.s[i], pass ? 10 * PROBE_TICKS : PROBE_TICKS)) { chosen = i; break; }
  p.pick[main] = DevPool::Pick{chosen, 0u};        // -1: none overlaps - the caller keeps everything on its own stream
  p.any_ok = p.any_ok || chosen >= 0;
  return chosen >= 0 ? p.s[chosen] : nullptr;
}

// diagnostics (tests): the side stream the library uses next to `stream`, its index in the pool, and whether a fresh probe sees the
// two overlap.  Returns 0, or SLN_E_STATE when no side stream can be had (first use inside a capture).
extern "C" int sln_debug_side_stream(void* stream, int* index, int* overlapped) {
  hipStream_t main = (hipStream_t)stream;
  hipStream_t side = sln_overlapping_stream(main);
  if (side == nullptr) return -3;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(g_mu);
  DevPool& p = g_pools[dev];
  int idx = -1;
  for (int i = 0; i < p.n; ++i) if (p.s[i] == side) idx = i;
  if (index) *index = idx;
  if (overlapped) *overlapped = (p.words != nullptr && !sln_capturing(main) && overlaps(p, main, side)) ? 1 : 0;
  return 0;
}

// Probe now (see the header of this file): 1 = a side stream overlaps with `stream`, 0 = none does (side work will run on `stream`
// itself), < 0 = error.  Synchronises `stream`.  Not inside a capture (SLN_E_STATE).
extern "C" int sln_side_stream_prepare(void* stream) {
  hipStream_t main = (hipStream_t)stream;
  if (sln_capturing(main)) return -3;
  {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -3;
    std::lock_guard<std::mutex> lk(g_mu);
    g_pools[dev].pick.erase(main);                 // a fresh probe even when an entry exists
  }
  return sln_overlapping_stream(main) != nullptr ? 1 : 0;
}

// Drops what the library remembers about `stream` (call before destroying a stream that ran sln_scene_backward / sln_vae_group_*):
// a later stream that gets the same handle is probed afresh.  Returns the number of entries dropped.
extern "C" int sln_side_stream_forget(void* stream) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -3;
  std::lock_guard<std::mutex> lk(g_mu);
  return (int)g_pools[dev].pick.erase((hipStream_t)stream);
}
