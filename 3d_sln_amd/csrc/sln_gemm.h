// Argument blocks of the fused fp32 GEMM kernels (see gemm_f32.hip).
#pragma once
#include "sln_common.h"

enum { EPI_PLAIN = 0, EPI_STATS = 1, EPI_MASK = 2 };

struct GemmNTArgs {
  Operand A;            // logical [M, K]
  const float* W;       // [N, K] row-major, leading dimension ldw
  const float* bias;    // [N] or nullptr
  float* Y;             // output rows of length ldy, written at column ycol0
  int ldy, ycol0;
  int M, N, K, ldw;
  // optional addend (another gradient contribution to the same tensor): y += addend[row, addcol0 + col]
  const float* addend;
  int ldadd, addcol0;
  // EPI_STATS: column sums of y / y^2 (train-mode BatchNorm statistics of the produced tensor)
  double* osums;
  int ocstride;
  // EPI_MASK: y is the gradient w.r.t. h = relu(bn(xprev)); writes g = y * [h > 0] and accumulates
  // sum g, sum g*xhat per column into ogsums (uses ocstride)
  int ldx, xcol0;
  const float* xprev;
  double* ogsums;
  BnView obn;           // aligned with output column 0
};

struct GemmTNArgs {
  Operand G;            // logical [R, Nout]: gradient w.r.t. the Linear output, rebuilt on load
  Operand X;            // logical [R, Kin]: the Linear input, rebuilt on load
  float* dW;            // [Nout, Kin] (+=)
  float* db;            // [Nout] (+=) or nullptr
  int lddw;
  int R, Nout, Kin;
  int rows_per_block;   // <= 0: choose
  // SGD in the epilogue (the refinement loop's R rooms in flight, vae_engine.hip SlnVaeGroup): when set (DEVICE pointer to the step),
  // dW / db point at the PARAMETERS and receive -step * (the gradient) - the same atomic add, one pass over the weights instead of
  // three (gradient +=, then the optimizer's read of both and write)
  const float* sgd_step;
};

// epi: EPI_*; tile: -1 = heuristic, 0 = 64x64, 1 = 128x64, 2 = 128x128
int sln_launch_gemm_nt(const GemmNTArgs& a, int epi, int tile, hipStream_t st);
int sln_launch_gemm_tn(const GemmTNArgs& a, int tile, hipStream_t st);
// dgrad (NT) and wgrad (TN) of the same Linear in one launch when both are small; falls back to two launches otherwise
int sln_launch_gemm_dual(const GemmNTArgs& nt, int epi, const GemmTNArgs& tn, hipStream_t st);
// up to two independent NT problems + up to two independent TN problems in one launch (gemm_group.hip); returns 1 without
// launching anything when the problems cannot share a kernel - the caller launches them separately then
int sln_launch_gemm_group(const GemmNTArgs* nt, const int* epi, int n_nt, const GemmTNArgs* tn, int n_tn, hipStream_t st);
// ---- every wgrad of a backward pass as ONE launch (round 3) --------------------------------------------------------------
// A wgrad has no consumer before the optimizer, so the engine no longer pairs it with the next dgrad: it records the problems
// and launches them together when the pass is over.  The problem table lives in device memory (a hipGraph replays the launch
// with the same pointers); each block finds its problem from the prefix of block counts.
enum { SLN_TN_MULTI_MAX = 64, SLN_TN_MULTI_ITEMS = 4096 };
// One entry per workgroup of the launch: problem, output tile, row chunk.  The table is laid out for the hardware's round-robin
// placement (workgroup b runs on XCD b % 8): all tiles of one (problem, row chunk) share an XCD - they read the same G / X rows,
// which then come out of THAT XCD's L2 once instead of from the fabric once per XCD (profiles/r03: 645 MB fetched per launch for
// 100 MB of operands before) - and the (problem, chunk) groups are dealt to the eight XCDs longest first, so that every XCD gets
// the same amount of work.  XCDs with fewer items are padded with empty entries (prob < 0).
struct TnMultiItem { int prob, tile, chunk, pad_; };
struct TnMultiMeta { int nprob, nblocks; TnMultiItem item[SLN_TN_MULTI_ITEMS]; };
// host side: fills rows_per_block of every problem (long chunks: the per-block prologue / 64x64 atomics are paid once per
// ~1 k rows instead of once per 256; longer still when the launch would not fit the table) and builds the item table.
// Returns 0, or a negative value when a problem cannot run on the TN body.
int sln_tn_multi_plan(GemmTNArgs* probs, int n, TnMultiMeta* meta, int* blocks, bool* x2, bool* xg, double* flops);
int sln_launch_gemm_tn_multi(const GemmTNArgs* dev_probs, const TnMultiMeta* dev_meta, int blocks, bool x2, bool xg, double flops,
                             hipStream_t st);
int sln_gemm_init();   // raises dynamic-LDS limits; call once outside any stream capture
