// Fused fp32 MFMA GEMM family for the scene-graph VAE (gfx950).
//
// All matmuls of the hot path are tiny-K (<= 640) fp32 products whose operands are *views* of
// stored pre-activations: BatchNorm+ReLU (forward) or the BatchNorm backward formula is applied
// while the tile is staged, the GraphTripleConv gather/concat is a row-indexed segment, and the
// epilogue produces the column statistics the next BatchNorm needs.  v_mfma_f32_32x32x2_f32 is
// exact fp32 (an fmaf chain) so results stay within 1e-4 of the reference without tricks.
//
//   gemm_nt : Y[M,N]  = op(A)[M,K] * W[N,K]^T (+bias)        forward Linear and dgrad (with W^T)
//   gemm_tn : dW[N,K] += op(G)[R,N]^T * op(X)[R,K]            wgrad, split over row chunks
//   gemm_tn_multi : every wgrad of a backward pass in ONE launch (problem table + one item per workgroup in device memory,
//                   XCD-grouped, ~768-row chunks; sln_tn_multi_plan builds the table on the host)
//
//   gemm_nt16 : the same product on 64 x 96 / 64 x 160 tiles of v_mfma_f32_16x16x4_f32 for the widths (N = 384, 640) that leave
//               64 x 64 tiles with a ragged last round of workgroups (nt16_pick)
//   gemm_nt_small : 32 x 32 tiles, K split over the four wavefronts (the object-side Linears: 64-128 tiles of 64 x 64 for 256 CUs)
//
// NT: k-contiguous LDS tiles ([rows][BK + 4]: one ds_read_b128 feeds four MFMAs); TN: row-major tiles as the rows arrive, the
// waves split a tile's rows (gemm_bodies.h).  Global->LDS staging goes through registers (the BatchNorm / ReLU / gather
// transform happens on the way), double-buffered so one barrier per K tile.  The K loops of the 64 x 64 NT tile, of the 16 x 16
// body and of the wgrad body are written out instruction by instruction (one piece of staging work behind each MFMA): one
// wavefront per SIMD hides nothing behind its own MFMAs (tools/lab/overlap.hip).  NT kernels with a
// coefficient table run 512 threads: wavefronts 4-7 build the table and leave.
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#include "gemm_bodies.h"
#include "vae_multi.h"
namespace {
// 64 x 64 tiles with a BatchNorm operand or a masked epilogue run with four helper wavefronts (512 threads, see gemm_nt_body)
template <int BM, int BN, int AMODE, int EPI>
constexpr bool nt_helpers() { return BM == 64 && BN == 64 && (AMODE != 2 || EPI == EPI_MASK); }
template <int BM, int BN, int WM, int WN, int AMODE, int EPI, int MULTI>
__global__ __launch_bounds__((nt_helpers<BM, BN, AMODE, EPI>() ? 512 : 256)) void gemm_nt_kernel(const GemmNTArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // the grid is exactly the tile count (launch_nt): computed from M and N, which the body needs anyway, instead of read from the
  // dispatch packet's block count - one kernel-argument cache line less in front of the first address
  gemm_nt_body<BM, BN, WM, WN, AMODE, EPI, MULTI, nt_helpers<BM, BN, AMODE, EPI>()>(a, blockIdx.x, ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN), smem);
}

template <int AMODE, int EPI>
constexpr bool nt_helpers2() { return AMODE != 2 || EPI == EPI_MASK; }      // there is a coefficient table to build
template <int J, int AMODE, int EPI>
__global__ __launch_bounds__((nt_helpers2<AMODE, EPI>() ? 512 : 256)) void gemm_nt16_kernel(const GemmNTArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemm_nt_body16<J, AMODE, EPI, nt_helpers2<AMODE, EPI>()>(a, blockIdx.x, ((a.M + 63) / 64) * ((a.N + 32 * J - 1) / (32 * J)), smem);
}

template <int AMODE, int EPI, int NSEG = 1>
__global__ __launch_bounds__((nt_helpers2<AMODE, EPI>() ? 512 : 256)) void gemm_nt_small_kernel(const GemmNTArgs a, const int xcd) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemm_nt_small_body<AMODE, EPI, nt_helpers2<AMODE, EPI>(), NSEG>(a, blockIdx.x, smem, xcd != 0);
}

// Forward Linears / dgrads of R rooms (R parameter copies) in one grid: blockIdx.y = room, the room's problem comes out of a device
// table, workgroups beyond its tile count leave (vae_multi.h).  Plain tile order inside a room (a room is a handful of tiles).
template <int AMODE, int EPI, int NSEG>
__global__ __launch_bounds__((nt_helpers2<AMODE, EPI>() ? 512 : 256)) void gemm_nt_small_multi_kernel(const GemmNTArgs* __restrict__ tab, const int* __restrict__ tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int r = blockIdx.y;
  if ((int)blockIdx.x >= tiles[r]) return;
  gemm_nt_small_body<AMODE, EPI, nt_helpers2<AMODE, EPI>(), NSEG>(tab[r], blockIdx.x, smem, false);
}
// ... and of the 64 x 64 body, for the operands the small body does not take (box_net's two-segment input [obj_vecs | attr_emb[attrs]])
template <int AMODE, int EPI, int MULTI>
__global__ __launch_bounds__((nt_helpers<64, 64, AMODE, EPI>() ? 512 : 256)) void gemm_nt64_multi_kernel(const GemmNTArgs* __restrict__ tab, const int* __restrict__ tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int r = blockIdx.y;
  const int nt = tiles[r];
  if ((int)blockIdx.x >= nt) return;
  gemm_nt_body<64, 64, 2, 2, AMODE, EPI, MULTI, nt_helpers<64, 64, AMODE, EPI>()>(tab[r], blockIdx.x, nt, smem);
}

}  // namespace
int sln_gemm_init();
namespace {

template <int AMODE, int EPI, int NSEG = 1>
int launch_nt_small(const GemmNTArgs& a, hipStream_t st) {
  const size_t smem = nt_small_smem_bytes(a.K);
  const int grid = sln_cdiv(a.M, 32) * sln_cdiv(a.N, 32);
  if (grid <= 0) return 0;
  if (smem > 48 * 1024) { int r = sln_gemm_init(); if (r) return r; }
  static const int xcd = std::getenv("SLN_NT_SMALL_NO_XCD") ? 0 : 1;      // lab switch: plain workgroup order
  hipLaunchKernelGGL((gemm_nt_small_kernel<AMODE, EPI, NSEG>), dim3(grid), dim3(nt_helpers2<AMODE, EPI>() ? 512 : 256), smem, st, a, xcd);
  SLN_CHECK_LAUNCH();
  return 0;
}

template <int J, int AMODE, int EPI>
int launch_nt16(const GemmNTArgs& a, hipStream_t st) {
  const size_t smem = nt16_smem_bytes(a.K, J);
  const int grid = sln_cdiv(a.M, 64) * sln_cdiv(a.N, 32 * J);
  if (grid <= 0) return 0;
  if (smem > 48 * 1024) { int r = sln_gemm_init(); if (r) return r; }
  hipLaunchKernelGGL((gemm_nt16_kernel<J, AMODE, EPI>), dim3(grid), dim3(nt_helpers2<AMODE, EPI>() ? 512 : 256), smem, st, a);
  SLN_CHECK_LAUNCH();
  return 0;
}

template <int BM, int BN, int WM, int WN, int AMODE, int EPI>
int launch_nt(const GemmNTArgs& a, hipStream_t st) {
  const int kpad = (a.K + 31) & ~31;
  const size_t smem = nt_smem_bytes(a.K, BM, BN, WM);
  const int grid = sln_cdiv(a.M, BM) * sln_cdiv(a.N, BN);
  if (grid <= 0) return 0;
  if (smem > 48 * 1024) { int r = sln_gemm_init(); if (r) return r; }
  constexpr int NTHR = nt_helpers<BM, BN, AMODE, EPI>() ? 512 : 256;
  if (a.A.nseg > 1 && nt_unaligned(a)) {
    if constexpr (BM == 64 && BN == 64) hipLaunchKernelGGL((gemm_nt_kernel<64, 64, WM, WN, AMODE, EPI, 2>), dim3(grid), dim3(NTHR), smem, st, a);
    else return -2;   // SLN_E_UNSUPPORTED (sln_launch_gemm_nt forces tile 0 for such operands)
  }
  else if (a.A.nseg > 1) hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WM, WN, AMODE, EPI, 1>), dim3(grid), dim3(NTHR), smem, st, a);
  else hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WM, WN, AMODE, EPI, 0>), dim3(grid), dim3(NTHR), smem, st, a);
  SLN_CHECK_LAUNCH();
  return 0;
}

template <int AMODE, int EPI>
int dispatch_nt_tile(const GemmNTArgs& a, hipStream_t st, int tile) {
  switch (tile) {
    case 1: return launch_nt<128, 64, 2, 2, AMODE, EPI>(a, st);
    case 2: return launch_nt<128, 128, 2, 2, AMODE, EPI>(a, st);
    default: return launch_nt<64, 64, 2, 2, AMODE, EPI>(a, st);
  }
}

template <int BM, int BN, int WM, int WN, bool G_X2, bool XG>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const GemmTNArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  gemm_tn_body<BM, BN, WM, WN, G_X2, XG>(a, blockIdx.x, blockIdx.y, smem);
}

// One launch for the two GEMMs that consume the same output gradient G of a Linear: the dgrad (NT, 64x64 tiles, blocks
// [0, nt_blocks)) and the wgrad (TN, the remaining tn_gx * tn_gy blocks).  At batch 64 either one fills less than half of
// the chip and a launch costs about as much as its work, so the pair shares one dispatch.
template <int AMODE, int EPI, bool XG>
__global__ __launch_bounds__(256) void gemm_dual_kernel(const GemmNTArgs a, const GemmTNArgs b, const int nt_blocks, const int tn_gx) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)blockIdx.x < nt_blocks) {
    gemm_nt_body<64, 64, 2, 2, AMODE, EPI, 0>(a, blockIdx.x, nt_blocks, smem);
  } else {
    const int id = blockIdx.x - nt_blocks;
    gemm_tn_body<64, 64, 2, 2, AMODE == 1, XG>(b, id % tn_gx, id / tn_gx, smem);
  }
}

// Every wgrad of a backward pass in one grid: workgroup b runs entry b of the item table (problem, output tile, row chunk; see
// TnMultiMeta for the XCD-aware layout) on that problem's description in device memory (uniform address: scalar loads).
template <bool G_X2, bool XG>
__global__ __launch_bounds__(256) void gemm_tn_multi_kernel(const GemmTNArgs* __restrict__ probs, const TnMultiMeta* __restrict__ meta) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const TnMultiItem it = meta->item[blockIdx.x];
  if (it.prob < 0) return;
  gemm_tn_body<64, 64, 2, 2, G_X2, XG>(probs[it.prob], it.tile, it.chunk, smem);
}


template <int AMODE, int EPI>
int launch_dual(const GemmNTArgs& a, const GemmTNArgs& b, hipStream_t st) {
  const size_t s1 = nt_smem_bytes(a.K, 64, 64, 2), s2 = tn_smem_bytes(64, 64);
  const size_t smem = s1 > s2 ? s1 : s2;
  const int nt_blocks = sln_cdiv(a.M, 64) * sln_cdiv(a.N, 64);
  const int gx = sln_cdiv(b.Nout, 64) * sln_cdiv(b.Kin, 64), gy = sln_cdiv(b.R, b.rows_per_block);
  if (smem > 48 * 1024) { int r = sln_gemm_init(); if (r) return r; }
  if (tn_gathers(b)) hipLaunchKernelGGL((gemm_dual_kernel<AMODE, EPI, true>), dim3(nt_blocks + gx * gy), dim3(256), smem, st, a, b, nt_blocks, gx);
  else hipLaunchKernelGGL((gemm_dual_kernel<AMODE, EPI, false>), dim3(nt_blocks + gx * gy), dim3(256), smem, st, a, b, nt_blocks, gx);
  SLN_CHECK_LAUNCH();
  return 0;
}

template <int BM, int BN, int WM, int WN, bool G_X2>
int launch_tn(const GemmTNArgs& a, hipStream_t st) {
  const size_t smem = tn_smem_bytes(BM, BN);
  const int gx = sln_cdiv(a.Nout, BM) * sln_cdiv(a.Kin, BN);
  const int gy = sln_cdiv(a.R, a.rows_per_block);
  if (gx <= 0 || gy <= 0) return 0;
  if (!tn_supported(a)) return -1;
  if (tn_gathers(a)) hipLaunchKernelGGL((gemm_tn_kernel<BM, BN, WM, WN, G_X2, true>), dim3(gx, gy), dim3(256), smem, st, a);
  else hipLaunchKernelGGL((gemm_tn_kernel<BM, BN, WM, WN, G_X2, false>), dim3(gx, gy), dim3(256), smem, st, a);
  SLN_CHECK_LAUNCH();
  return 0;
}

}  // namespace

// Raise the dynamic-LDS limit of every instantiation once (must not happen inside a stream capture).
template <int BM, int BN, int WM, int WN>
static int init_nt_tile() {
  hipError_t e = hipSuccess;
#define SLN_SET(X2, EPI)                                                                                          \
  if (e == hipSuccess)                                                                                            \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<BM, BN, WM, WN, X2, EPI, 0>),           \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                              \
  if (e == hipSuccess)                                                                                            \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<BM, BN, WM, WN, X2, EPI, 1>),           \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                              \
  if (e == hipSuccess && BM == 64 && BN == 64)                                                                    \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<64, 64, WM, WN, X2, EPI, 2>),           \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  SLN_SET(0, EPI_PLAIN) SLN_SET(0, EPI_STATS) SLN_SET(0, EPI_MASK)
  SLN_SET(1, EPI_PLAIN) SLN_SET(1, EPI_STATS) SLN_SET(1, EPI_MASK)
  SLN_SET(2, EPI_PLAIN) SLN_SET(2, EPI_STATS) SLN_SET(2, EPI_MASK)
#undef SLN_SET
  return (int)e;
}

int sln_gemm_group_init();
int sln_gemm_init() {
  static bool done = false;
  if (done) return 0;
  int r = init_nt_tile<64, 64, 2, 2>();
#define SLN_SET_S(AM, EPI)                                                                                        \
  if (!r) r = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_small_kernel<AM, EPI, 1>),         \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                  \
  if (!r) r = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_small_kernel<AM, EPI, 3>),         \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  SLN_SET_S(0, EPI_PLAIN) SLN_SET_S(0, EPI_STATS) SLN_SET_S(0, EPI_MASK) SLN_SET_S(1, EPI_PLAIN) SLN_SET_S(1, EPI_STATS)
  SLN_SET_S(1, EPI_MASK) SLN_SET_S(2, EPI_PLAIN) SLN_SET_S(2, EPI_STATS) SLN_SET_S(2, EPI_MASK)
#undef SLN_SET_S
#define SLN_SET_16(J, AM, EPI)                                                                                    \
  if (!r) r = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt16_kernel<J, AM, EPI>),             \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#define SLN_SET_16J(J)                                                                                            \
  SLN_SET_16(J, 0, EPI_PLAIN) SLN_SET_16(J, 0, EPI_STATS) SLN_SET_16(J, 0, EPI_MASK) SLN_SET_16(J, 1, EPI_PLAIN) SLN_SET_16(J, 1, EPI_STATS) \
  SLN_SET_16(J, 1, EPI_MASK) SLN_SET_16(J, 2, EPI_PLAIN) SLN_SET_16(J, 2, EPI_STATS) SLN_SET_16(J, 2, EPI_MASK)
  SLN_SET_16J(3) SLN_SET_16J(5)
#undef SLN_SET_16J
#undef SLN_SET_16
  if (!r) r = init_nt_tile<128, 64, 2, 2>();
  if (!r) r = init_nt_tile<128, 128, 2, 2>();
#define SLN_SET_TN(X2, XG)                                                                                        \
  if (!r) r = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_kernel<64, 64, 2, 2, X2, XG>),     \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  SLN_SET_TN(true, true) SLN_SET_TN(true, false) SLN_SET_TN(false, true) SLN_SET_TN(false, false)
#undef SLN_SET_TN
#define SLN_SET_TNM(X2, XG)                                                                                       \
  if (!r) r = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_multi_kernel<X2, XG>),             \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  SLN_SET_TNM(true, true) SLN_SET_TNM(true, false) SLN_SET_TNM(false, true) SLN_SET_TNM(false, false)
#undef SLN_SET_TNM
#define SLN_SET_DUAL(AM, EPI)                                                                                     \
  if (!r) r = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_dual_kernel<AM, EPI, false>),         \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                  \
  if (!r) r = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_dual_kernel<AM, EPI, true>),          \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  SLN_SET_DUAL(0, EPI_PLAIN) SLN_SET_DUAL(0, EPI_MASK) SLN_SET_DUAL(1, EPI_PLAIN) SLN_SET_DUAL(1, EPI_MASK)
  SLN_SET_DUAL(2, EPI_PLAIN) SLN_SET_DUAL(2, EPI_MASK)
#undef SLN_SET_DUAL
  if (!r) r = sln_gemm_group_init();
  done = r == 0;
  return r;
}




int sln_launch_gemm_nt(const GemmNTArgs& a, int epi, int tile, hipStream_t st) {
  SlnProfScope prof(SLN_FAM_GEMM_NT, 2.0 * a.M * a.N * a.K, st);
  static const bool no_small = std::getenv("SLN_NO_SMALL_NT") != nullptr;
  const int amode = nt_amode(a);
  // SLN_NT_LOG=1: one line per launch on stderr (tools/lab/nt_by_shape.sh joins them, in order, with a kernel trace)
  static const bool nt_log = std::getenv("SLN_NT_LOG") != nullptr;
  if (nt_log) std::fprintf(stderr, "NTLOG M=%d N=%d K=%d amode=%d epi=%d nseg=%d\n", a.M, a.N, a.K, amode, epi, a.A.nseg);
  // stand-alone launches only: inside a dual launch (dgrad blocks next to wgrad blocks on every CU) the small body measured
  // slower than the 64 x 64 one (pairs 28.8 -> 30.3 us on average): its 4x more workgroups pay 4x the prologues on a busy chip
  if (tile < 0 && !no_small && nt_wants_small(a)) {
#define SLN_DISPATCH_S(AM)                                                            \
    if (amode == AM) {                                                                \
      if (epi == EPI_MASK) return launch_nt_small<AM, EPI_MASK>(a, st);               \
      if (epi == EPI_STATS) return launch_nt_small<AM, EPI_STATS>(a, st);             \
      return launch_nt_small<AM, EPI_PLAIN>(a, st);                                   \
    }
    SLN_DISPATCH_S(0) SLN_DISPATCH_S(1) SLN_DISPATCH_S(2)
#undef SLN_DISPATCH_S
  }
  if (tile < 0 && !no_small && nt_wants_small3(a)) {          // the gathered concat of a graph of a few rows
#define SLN_DISPATCH_S3(AM)                                                           \
    if (amode == AM) {                                                                \
      if (epi == EPI_MASK) return launch_nt_small<AM, EPI_MASK, 3>(a, st);            \
      if (epi == EPI_STATS) return launch_nt_small<AM, EPI_STATS, 3>(a, st);          \
      return launch_nt_small<AM, EPI_PLAIN, 3>(a, st);                                \
    }
    SLN_DISPATCH_S3(0) SLN_DISPATCH_S3(1) SLN_DISPATCH_S3(2)
#undef SLN_DISPATCH_S3
  }
  if (tile < 0) {
    // widths that leave 64 x 64 tiles with a ragged last round (N = 640: 2.5 tiles per CU at 64 graphs, N = 384: 1.5) run on
    // 64 x 160 / 64 x 96 tiles of 16 x 16 MFMAs - one workgroup per CU, see gemm_nt_body16
    const int J16 = nt16_pick(a);
#define SLN_DISPATCH_16(J, AM)                                                        \
    if (J16 == J && amode == AM) {                                                    \
      if (epi == EPI_MASK) return launch_nt16<J, AM, EPI_MASK>(a, st);                \
      if (epi == EPI_STATS) return launch_nt16<J, AM, EPI_STATS>(a, st);              \
      return launch_nt16<J, AM, EPI_PLAIN>(a, st);                                    \
    }
    SLN_DISPATCH_16(3, 0) SLN_DISPATCH_16(3, 1) SLN_DISPATCH_16(3, 2) SLN_DISPATCH_16(5, 0) SLN_DISPATCH_16(5, 1) SLN_DISPATCH_16(5, 2)
#undef SLN_DISPATCH_16
  }
  if (tile < 0) tile = nt_heuristic_tile(a);
  if (a.A.nseg > 1 && nt_unaligned(a)) tile = 0;        // the per-thread segment choice exists for the 64x64 tile only
#define SLN_DISPATCH(AM)                                                              \
  if (amode == AM) {                                                                  \
    if (epi == EPI_MASK) return dispatch_nt_tile<AM, EPI_MASK>(a, st, tile);          \
    if (epi == EPI_STATS) return dispatch_nt_tile<AM, EPI_STATS>(a, st, tile);        \
    return dispatch_nt_tile<AM, EPI_PLAIN>(a, st, tile);                              \
  }
  SLN_DISPATCH(0) SLN_DISPATCH(1) SLN_DISPATCH(2)
#undef SLN_DISPATCH
  return -1;
}

// ---- multi-room form of the 32 x 32 split-K body (vae_multi.h) -------------------------------------------------------------
int sln_plan_nt_small(const GemmNTArgs& a, int epi, int* tiles) {
  static const bool no_small = std::getenv("SLN_NO_SMALL_NT") != nullptr;
  if (no_small || a.M <= 0 || a.N <= 0) return -1;
  int nseg = 0;
  const int amode = nt_amode(a);
  if (amode == 1 || epi == EPI_STATS) return -1;    // train-mode BatchNorm: not instantiated (the refinement loop runs in eval mode)
  if (nt_wants_small(a)) nseg = 1;                  // the single-room dispatcher's own rules, in its order (sln_launch_gemm_nt)
  else if (nt_wants_small3(a)) nseg = 3;
  if (nseg) { *tiles = sln_cdiv(a.M, 32) * sln_cdiv(a.N, 32); return amode * 16 + epi * 4 + nseg; }
  if (nt_big_shape(a)) return -1;
  // everything else of a room's size: the 64 x 64 body (key 1000 + ...; last digit: 0 one segment, 1 segments on k-tile
  // boundaries, 2 a boundary inside a k-tile)
  *tiles = sln_cdiv(a.M, 64) * sln_cdiv(a.N, 64);
  return 1000 + amode * 16 + epi * 4 + (a.A.nseg > 1 ? (nt_unaligned(a) ? 2 : 1) : 0);
}

template <int AMODE, int EPI, int NSEG>
static int launch_nt_small_multi(const GemmNTArgs* tab, const int* tiles, int R, int max_tiles, size_t smem, hipStream_t st) {
  static bool raised = false;
  if (smem > 48 * 1024 && !raised) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_small_multi_kernel<AMODE, EPI, NSEG>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    raised = true;
  }
  hipLaunchKernelGGL((gemm_nt_small_multi_kernel<AMODE, EPI, NSEG>), dim3(max_tiles, R), dim3(nt_helpers2<AMODE, EPI>() ? 512 : 256), smem, st, tab, tiles);
  SLN_CHECK_LAUNCH();
  return 0;
}

template <int AMODE, int EPI, int MULTI>
static int launch_nt64_multi(const GemmNTArgs* tab, const int* tiles, int R, int max_tiles, size_t smem, hipStream_t st) {
  static bool raised = false;
  if (smem > 48 * 1024 && !raised) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt64_multi_kernel<AMODE, EPI, MULTI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    raised = true;
  }
  hipLaunchKernelGGL((gemm_nt64_multi_kernel<AMODE, EPI, MULTI>), dim3(max_tiles, R), dim3(nt_helpers<64, 64, AMODE, EPI>() ? 512 : 256), smem, st, tab, tiles);
  SLN_CHECK_LAUNCH();
  return 0;
}

int sln_launch_gemm_nt_small_multi(const GemmNTArgs* tab, const int* tiles_dev, int R, int key, int max_tiles, int max_K, double flops,
                                   hipStream_t st) {
  if (R <= 0 || max_tiles <= 0) return 0;
  SlnProfScope prof(SLN_FAM_GEMM_NT, flops, st);
  if (key >= 1000) {
    const size_t smem64 = nt_smem_bytes(max_K, 64, 64, 2);
    const int am = (key - 1000) / 16, ep = ((key - 1000) / 4) % 4, mu = (key - 1000) % 4;
#define SLN_NT64M(AM, EP, MU) if (am == AM && ep == EP && mu == MU) return launch_nt64_multi<AM, EP, MU>(tab, tiles_dev, R, max_tiles, smem64, st);
    SLN_NT64M(0, EPI_PLAIN, 0) SLN_NT64M(0, EPI_PLAIN, 1) SLN_NT64M(0, EPI_PLAIN, 2) SLN_NT64M(0, EPI_MASK, 0) SLN_NT64M(0, EPI_MASK, 1) SLN_NT64M(0, EPI_MASK, 2)
    SLN_NT64M(2, EPI_PLAIN, 0) SLN_NT64M(2, EPI_PLAIN, 1) SLN_NT64M(2, EPI_PLAIN, 2) SLN_NT64M(2, EPI_MASK, 0) SLN_NT64M(2, EPI_MASK, 1) SLN_NT64M(2, EPI_MASK, 2)
#undef SLN_NT64M
    return -1;
  }
  const size_t smem = nt_small_smem_bytes(max_K);
  const int amode = key / 16, epi = (key / 4) % 4, nseg = key % 4;
#define SLN_NTM(AM, EP, NS) if (amode == AM && epi == EP && nseg == NS) return launch_nt_small_multi<AM, EP, NS>(tab, tiles_dev, R, max_tiles, smem, st);
  SLN_NTM(0, EPI_PLAIN, 1) SLN_NTM(0, EPI_PLAIN, 3) SLN_NTM(0, EPI_MASK, 1) SLN_NTM(0, EPI_MASK, 3)
  SLN_NTM(2, EPI_PLAIN, 1) SLN_NTM(2, EPI_PLAIN, 3) SLN_NTM(2, EPI_MASK, 1) SLN_NTM(2, EPI_MASK, 3)
#undef SLN_NTM
  return -1;
}

int sln_launch_gemm_tn(const GemmTNArgs& a0, int tile, hipStream_t st) {
  SlnProfScope prof(SLN_FAM_GEMM_TN, 2.0 * a0.R * a0.Nout * a0.Kin, st);
  GemmTNArgs a = a0;
  const bool x2 = tn_prepare(a);
  (void)tile;
  if (x2) return launch_tn<64, 64, 2, 2, true>(a, st);
  return launch_tn<64, 64, 2, 2, false>(a, st);
}

int sln_launch_gemm_dual(const GemmNTArgs& a, int epi, const GemmTNArgs& b0, hipStream_t st) {
  GemmTNArgs b = b0;
  const bool x2 = tn_prepare(b);
  const int amode = nt_amode(a);
  const bool nonempty = a.M > 0 && a.N > 0 && b.R > 0 && b.Nout > 0 && b.Kin > 0;
  if (!nonempty || nt_big_shape(a) || epi == EPI_STATS || x2 != (amode == 1) || a.A.nseg > 1 || !tn_supported(b)) {   // big or odd shapes: separate launches
    int r = sln_launch_gemm_tn(b0, -1, st);
    return r ? r : sln_launch_gemm_nt(a, epi, -1, st);
  }
  SlnProfScope prof(SLN_FAM_GEMM_DUAL, 2.0 * a.M * a.N * a.K + 2.0 * b.R * b.Nout * b.Kin, st);
#define SLN_DISPATCH(AM)                                                              \
  if (amode == AM) return epi == EPI_MASK ? launch_dual<AM, EPI_MASK>(a, b, st) : launch_dual<AM, EPI_PLAIN>(a, b, st);
  SLN_DISPATCH(0) SLN_DISPATCH(1) SLN_DISPATCH(2)
#undef SLN_DISPATCH
  return -1;
}

int sln_tn_multi_plan(GemmTNArgs* probs, int n, TnMultiMeta* meta, int* blocks, bool* x2, bool* xg, double* flops) {
  if (n < 1 || n > SLN_TN_MULTI_MAX) return -1;
  // rows per block: long chunks (tools/gemm_bench.py: a 256-row block spends as long in its prologue, its first-tile latency and
  // its 64 x 64 atomics as in its 8 tiles - 46 TF at R = 4096 against 79 TF with 1.6 k-row chunks)
  static const int target0 = std::getenv("SLN_TN_MULTI_ROWS") ? std::atoi(std::getenv("SLN_TN_MULTI_ROWS")) : 768;
  static const bool no_xcd = std::getenv("SLN_TN_NO_XCD") != nullptr;           // lab: plain order, no XCD grouping
  bool any_x2 = false, any_xg = false;
  double work = 0.0;
  for (int i = 0; i < n; ++i) {
    const GemmTNArgs& a = probs[i];
    if (a.R <= 0 || a.Nout <= 0 || a.Kin <= 0 || !tn_supported(a)) return -1;
    for (int s = 0; s < a.G.nseg; ++s) any_x2 |= a.G.seg[s].x2 != nullptr;
    any_xg |= tn_gathers(a);
    work += 2.0 * a.R * a.Nout * a.Kin;
  }
  // chunking; the launch must fit the item table (8 XCDs x the longest XCD list): longer chunks for big batches
  struct Group { int prob, chunk, tiles; long cost; };
  std::vector<Group> groups;
  int per_xcd[8];
  std::vector<int> owner;
  // deterministic mode: ONE chunk per problem - every dW / db element then receives exactly one add per launch (onto the zeroed
  // gradient, or onto the previous launch's result in stream order): no sum depends on the arrival order of workgroups
  for (long target = g_sln_deterministic ? (1L << 30) : (target0 > 0 ? target0 : 1024);; target *= 2) {
    groups.clear();
    for (int i = 0; i < n; ++i) {
      GemmTNArgs& a = probs[i];
      const int chunks = sln_cdiv(a.R, (int)(target > a.R ? a.R : target));
      a.rows_per_block = sln_cdiv(sln_cdiv(a.R, chunks), BK) * BK;
      const int tiles = sln_cdiv(a.Nout, 64) * sln_cdiv(a.Kin, 64), nch = sln_cdiv(a.R, a.rows_per_block);
      for (int c = 0; c < nch; ++c) {
        const int rows = (c + 1) * a.rows_per_block <= a.R ? a.rows_per_block : a.R - c * a.rows_per_block;
        groups.push_back(Group{i, c, tiles, (long)tiles * rows});
      }
    }
    // longest processing time first: sort the groups by cost (stable), give each to the XCD with the least work so far
    std::stable_sort(groups.begin(), groups.end(), [](const Group& x, const Group& y) { return x.cost > y.cost; });
    long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int x = 0; x < 8; ++x) per_xcd[x] = 0;
    owner.assign(groups.size(), 0);
    for (size_t g = 0; g < groups.size(); ++g) {
      int best = 0;
      if (no_xcd) best = (int)(g % 8);
      else for (int x = 1; x < 8; ++x) if (load[x] < load[best]) best = x;
      owner[g] = best; load[best] += groups[g].cost; per_xcd[best] += groups[g].tiles;
    }
    int mx = 0;
    for (int x = 0; x < 8; ++x) mx = per_xcd[x] > mx ? per_xcd[x] : mx;
    if (8 * mx <= SLN_TN_MULTI_ITEMS) break;
    if (target > (1L << 30)) return -1;
  }
  if (any_xg) {        // the index pipeline of the gathering body loads indices unconditionally: every X needs a valid array
    const int* any = nullptr;
    for (int i = 0; i < n; ++i) { if (probs[i].X.idx_a) any = probs[i].X.idx_a; else if (probs[i].X.idx_b) any = probs[i].X.idx_b; }
    for (int i = 0; i < n; ++i) if (!probs[i].X.idx_a && !probs[i].X.idx_b) probs[i].X.idx_a = any;
  }
  int mx = 0;
  for (int x = 0; x < 8; ++x) mx = per_xcd[x] > mx ? per_xcd[x] : mx;
  std::memset(meta, 0, sizeof(*meta));
  meta->nprob = n; meta->nblocks = 8 * mx;
  for (int b = 0; b < 8 * mx; ++b) meta->item[b].prob = -1;
  int fill[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (size_t g = 0; g < groups.size(); ++g) {          // cost order: an XCD starts with its longest groups
    const int x = owner[g];
    for (int t = 0; t < groups[g].tiles; ++t) {
      TnMultiItem& it = meta->item[8 * (fill[x]++) + x];  // the j-th workgroup of XCD x is workgroup 8 j + x of the grid
      it.prob = groups[g].prob; it.tile = t; it.chunk = groups[g].chunk;
    }
  }
  *blocks = 8 * mx; *x2 = any_x2; *xg = any_xg; *flops = work;
  return 0;
}

int sln_launch_gemm_tn_multi(const GemmTNArgs* dev_probs, const TnMultiMeta* dev_meta, int blocks, bool x2, bool xg, double flops,
                             hipStream_t st) {
  if (blocks <= 0) return 0;
  SlnProfScope prof(SLN_FAM_GEMM_TN, flops, st);
  const size_t smem = tn_smem_bytes(64, 64);
  if (smem > 48 * 1024) { int r = sln_gemm_init(); if (r) return r; }
  if (x2 && xg) hipLaunchKernelGGL((gemm_tn_multi_kernel<true, true>), dim3(blocks), dim3(256), smem, st, dev_probs, dev_meta);
  else if (x2) hipLaunchKernelGGL((gemm_tn_multi_kernel<true, false>), dim3(blocks), dim3(256), smem, st, dev_probs, dev_meta);
  else if (xg) hipLaunchKernelGGL((gemm_tn_multi_kernel<false, true>), dim3(blocks), dim3(256), smem, st, dev_probs, dev_meta);
  else hipLaunchKernelGGL((gemm_tn_multi_kernel<false, false>), dim3(blocks), dim3(256), smem, st, dev_probs, dev_meta);
  SLN_CHECK_LAUNCH();
  return 0;
}

// Host-only view of the planner for tests (no device work): problems given by their (rows, outputs, inputs); returns the number
// of workgroups, writes rows_per_block[n] and one (problem, tile, chunk) triple per workgroup (problem < 0: padding).
extern "C" int sln_debug_tn_plan(const int* R, const int* Nout, const int* Kin, int n, int* rows_per_block, int* items, int max_items) {
  if (!R || !Nout || !Kin || !rows_per_block || !items || n < 1 || n > SLN_TN_MULTI_MAX) return -1;
  static thread_local GemmTNArgs probs[SLN_TN_MULTI_MAX];
  static thread_local TnMultiMeta meta;
  std::memset(probs, 0, sizeof(GemmTNArgs) * (size_t)n);
  for (int i = 0; i < n; ++i) {
    probs[i].R = R[i]; probs[i].Nout = Nout[i]; probs[i].Kin = Kin[i];
    probs[i].G.nseg = 1; probs[i].X.nseg = 1; probs[i].lddw = i;          // lddw carries the caller's index through the planner
  }
  int blocks = 0; bool x2 = false, xg = false; double flops = 0.0;
  const int r = sln_tn_multi_plan(probs, n, &meta, &blocks, &x2, &xg, &flops);
  if (r) return r;
  if (blocks > max_items) return -2;
  for (int i = 0; i < n; ++i) rows_per_block[probs[i].lddw] = probs[i].rows_per_block;
  for (int b = 0; b < blocks; ++b) {
    const TnMultiItem& it = meta.item[b];
    items[3 * b] = it.prob < 0 ? -1 : probs[it.prob].lddw; items[3 * b + 1] = it.tile; items[3 * b + 2] = it.chunk;
  }
  return blocks;
}
