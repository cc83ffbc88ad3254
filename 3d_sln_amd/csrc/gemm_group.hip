// Grouped launches of the fused GEMM family: up to two NT problems (forward Linears or dgrads) and up to two TN problems
// (wgrads) that are mutually independent run as ONE grid.  The scene-graph VAE has pairs of identical-shape branches
// (box / angle posterior heads of the encoder, box_net / angle_net of the decoder) whose GEMMs carry 5-15 us of work each
// under a ~5 us launch floor; grouping them halves the number of dispatches of those stages.
// The kernel bodies are the ones of gemm_f32.hip (gemm_bodies.h).
#include "gemm_bodies.h"

namespace {

struct GroupArgs {
  GemmNTArgs nt[2];
  GemmTNArgs tn[2];
  int nt_blocks[2];
  int tn_gx[2], tn_blocks[2];
};

struct GroupDims { int nt_blocks[2]; int tn_gx[2], tn_blocks[2]; };

// the four problem descriptions are SEPARATE kernel parameters: as members of one struct parameter hipcc copied them to
// scratch memory (2.6 KB per lane) and the two-source variants ran 4x slower
template <int AMODE, int EPI, bool MULTI, bool XG, bool HAS_TN>
// The block counts that decide which problem a workgroup belongs to are LEADING scalar arguments: they arrive in SGPRs with the
// wavefront (kernarg preload, build.py) - as the last member of the 2.8 KB argument block they were a scalar-cache miss of their own
// in front of the first field of the chosen problem.
__global__ __launch_bounds__(256) void gemm_group_kernel(const int nt_blocks0, const int nt_blocks1, const int tn_gx0, const int tn_gx1, const int tn_blocks0,
                                                         const GemmNTArgs a0, const GemmNTArgs a1, const GemmTNArgs t0, const GemmTNArgs t1) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int b = blockIdx.x;
  if (b < nt_blocks0) { gemm_nt_body<64, 64, 2, 2, AMODE, EPI, MULTI>(a0, b, nt_blocks0, smem); return; }
  b -= nt_blocks0;
  if (b < nt_blocks1) { gemm_nt_body<64, 64, 2, 2, AMODE, EPI, MULTI>(a1, b, nt_blocks1, smem); return; }
  b -= nt_blocks1;
  if (HAS_TN) {
    if (b < tn_blocks0) { gemm_tn_body<64, 64, 2, 2, AMODE == 1, XG>(t0, b % tn_gx0, b / tn_gx0, smem); return; }
    b -= tn_blocks0;
    gemm_tn_body<64, 64, 2, 2, AMODE == 1, XG>(t1, b % tn_gx1, b / tn_gx1, smem);
  }
}

template <int AMODE, int EPI, bool MULTI, bool XG, bool HAS_TN>
int launch_group(const GroupArgs& g, size_t smem, int blocks, hipStream_t st) {
  if (smem > 48 * 1024) { int r = sln_gemm_init(); if (r) return r; }
  GroupDims d;
  for (int i = 0; i < 2; ++i) { d.nt_blocks[i] = g.nt_blocks[i]; d.tn_gx[i] = g.tn_gx[i] > 0 ? g.tn_gx[i] : 1; d.tn_blocks[i] = g.tn_blocks[i]; }
  hipLaunchKernelGGL((gemm_group_kernel<AMODE, EPI, MULTI, XG, HAS_TN>), dim3(blocks), dim3(256), smem, st, d.nt_blocks[0], d.nt_blocks[1], d.tn_gx[0], d.tn_gx[1],
                     d.tn_blocks[0], g.nt[0], g.nt[1], g.tn[0], g.tn[1]);
  SLN_CHECK_LAUNCH();
  return 0;
}

inline bool nt_multi(const GemmNTArgs& a) { return a.A.nseg > 1; }

template <int AMODE, int EPI, bool MULTI, bool XG, bool HAS_TN>
int raise_lds_limit() {
  return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_group_kernel<AMODE, EPI, MULTI, XG, HAS_TN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace

// dynamic-LDS limits of every grouped instantiation (called by sln_gemm_init, outside any stream capture)
int sln_gemm_group_init() {
  int r = 0;
#define SLN_G_FWD(AM, EP) if (!r) r = raise_lds_limit<AM, EP, false, false, false>(); if (!r) r = raise_lds_limit<AM, EP, true, false, false>();
#define SLN_G_BWD(AM, EP, XGV) if (!r) r = raise_lds_limit<AM, EP, false, XGV, true>();
  SLN_G_FWD(0, EPI_PLAIN) SLN_G_FWD(0, EPI_STATS) SLN_G_FWD(2, EPI_PLAIN) SLN_G_FWD(2, EPI_STATS)
  // dgrad-only groups (round 3: the wgrads of a pass run in their own launch, sln_launch_gemm_tn_multi)
  SLN_G_FWD(0, EPI_MASK) SLN_G_FWD(1, EPI_PLAIN) SLN_G_FWD(1, EPI_MASK) SLN_G_FWD(2, EPI_MASK)
  SLN_G_BWD(0, EPI_PLAIN, false) SLN_G_BWD(0, EPI_MASK, false) SLN_G_BWD(1, EPI_PLAIN, false) SLN_G_BWD(1, EPI_MASK, false)
  SLN_G_BWD(2, EPI_PLAIN, false) SLN_G_BWD(2, EPI_MASK, false) SLN_G_BWD(1, EPI_PLAIN, true)
#undef SLN_G_FWD
#undef SLN_G_BWD
  return r;
}

// nt[n_nt] with epilogues epi[n_nt], tn[n_tn]; n_nt, n_tn <= 2.  All problems must be independent of each other.
// Returns SLN_GROUP_FALLBACK (1) without launching when the problems do not fit one grouped kernel (different operand modes
// or epilogues, shapes that want a bigger tile, gathered gradient operands): the caller then launches them one by one.
int sln_launch_gemm_group(const GemmNTArgs* nt, const int* epi, int n_nt, const GemmTNArgs* tn_in, int n_tn, hipStream_t st) {
  if (n_nt < 1 || n_nt > 2 || n_tn < 0 || n_tn > 2) return 1;
  GroupArgs g; std::memset(&g, 0, sizeof(g));
  const int amode = nt_amode(nt[0]);
  bool multi = false;
  double work = 0.0;
  size_t smem = 0;
  int blocks = 0;
  for (int i = 0; i < n_nt; ++i) {
    if (nt[i].M <= 0 || nt[i].N <= 0 || nt_big_shape(nt[i]) || nt_amode(nt[i]) != amode || epi[i] != epi[0]) return 1;
    multi |= nt_multi(nt[i]);
    if (nt_multi(nt[i]) && nt_unaligned(nt[i])) return 1;
    g.nt[i] = nt[i];
    g.nt_blocks[i] = sln_cdiv(nt[i].M, 64) * sln_cdiv(nt[i].N, 64);
    blocks += g.nt_blocks[i];
    const size_t s = nt_smem_bytes(nt[i].K, 64, 64, 2);
    smem = s > smem ? s : smem;
    work += 2.0 * nt[i].M * nt[i].N * nt[i].K;
  }
  bool xg = false;
  for (int i = 0; i < n_tn; ++i) {
    GemmTNArgs t = tn_in[i];
    const bool x2 = tn_prepare(t);
    if (t.R <= 0 || t.Nout <= 0 || t.Kin <= 0 || !tn_supported(t) || x2 != (amode == 1) || epi[0] == EPI_STATS) return 1;
    xg |= tn_gathers(t);
    g.tn[i] = t;
    g.tn_gx[i] = sln_cdiv(t.Nout, 64) * sln_cdiv(t.Kin, 64);
    g.tn_blocks[i] = g.tn_gx[i] * sln_cdiv(t.R, t.rows_per_block);
    blocks += g.tn_blocks[i];
    const size_t s = tn_smem_bytes(64, 64);
    smem = s > smem ? s : smem;
    work += 2.0 * t.R * t.Nout * t.Kin;
  }
  if (xg) {                                  // the index pipeline loads indices unconditionally: every X needs a valid index array
    const int* any = nullptr;
    for (int i = 0; i < n_tn; ++i) { if (g.tn[i].X.idx_a) any = g.tn[i].X.idx_a; else if (g.tn[i].X.idx_b) any = g.tn[i].X.idx_b; }
    for (int i = 0; i < n_tn; ++i) if (!g.tn[i].X.idx_a && !g.tn[i].X.idx_b) g.tn[i].X.idx_a = any;
  }
  if (n_tn == 0 && multi && epi[0] == EPI_MASK) return 1;
  SlnProfScope prof(SLN_FAM_GEMM_DUAL, work, st);
  const int e0 = epi[0];
#define SLN_GROUP_FWD(AM, EP)                                                                     \
  if (amode == AM && e0 == EP)                                                                    \
    return multi ? launch_group<AM, EP, true, false, false>(g, smem, blocks, st) : launch_group<AM, EP, false, false, false>(g, smem, blocks, st);
#define SLN_GROUP_BWD(AM, EP, XGV)                                                                \
  if (amode == AM && e0 == EP && xg == XGV && !multi) return launch_group<AM, EP, false, XGV, true>(g, smem, blocks, st);
  if (n_tn == 0) {
    SLN_GROUP_FWD(0, EPI_PLAIN) SLN_GROUP_FWD(0, EPI_STATS) SLN_GROUP_FWD(2, EPI_PLAIN) SLN_GROUP_FWD(2, EPI_STATS)
    SLN_GROUP_FWD(0, EPI_MASK) SLN_GROUP_FWD(1, EPI_PLAIN) SLN_GROUP_FWD(1, EPI_MASK) SLN_GROUP_FWD(2, EPI_MASK)
  } else {
    SLN_GROUP_BWD(0, EPI_PLAIN, false) SLN_GROUP_BWD(0, EPI_MASK, false) SLN_GROUP_BWD(1, EPI_PLAIN, false) SLN_GROUP_BWD(1, EPI_MASK, false)
    SLN_GROUP_BWD(2, EPI_PLAIN, false) SLN_GROUP_BWD(2, EPI_MASK, false) SLN_GROUP_BWD(1, EPI_PLAIN, true)
  }
#undef SLN_GROUP_FWD
#undef SLN_GROUP_BWD
  return 1;
}
