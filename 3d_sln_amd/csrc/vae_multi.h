// Multi-room launches (round 5): the layout-refinement loop fine-tunes one COPY of the model per room
// (testing/test_render_refine.py:250-263 reloads the checkpoint for every trial, :286-292 steps its parameters), so R rooms in
// flight are R independent problems with R parameter sets.  Every kernel of a decoder forward / backward pass gets a
// "one launch for all rooms" form: a table of per-room argument blocks in device memory (pointers into that room's engine
// workspace and parameter copy) and a grid of (largest per-room grid) x R; blockIdx.z picks the room, workgroups beyond the
// room's own grid leave.  The kernel bodies are the single-room ones, untouched: a room's arithmetic does not depend on how many
// rooms share the launch.
#pragma once
#include "sln_gemm.h"
#include "vae_kernels.h"

struct MScatterFwd { const float* A2; int ld, H, D; BnView bn; GraphCsr g; int O; float* pooled; int gx, gy; };
struct MScatterBwd {
  const float* dM; const float* dP; int lddp, dpcol0; const float* A2; int ld, H, D; BnView bn; GraphCsr g; int T;
  float* g2; double* gsums; int cstride; int gx, gy;
};
struct MGatherBwd {
  const float* dG; int ldg, D; GraphCsr g; int O; const float* add1; int ldadd1; const float* xprev; int ldx; BnView bn; int masked;
  float* out; int ldo; double* gsums; int cstride; int gx, gy;
};
struct MMaskGstats {
  const float* d1; int ld1; const float* d2; int ld2; const float* xprev; int ldx; BnView bn; int rows, cols; float* out; int ldo;
  double* gsums; int cstride; int gx, gy;
};
struct MDecAssemble { DecAssemble a; int gx, pad_; };
struct MDecAssembleBwd { DecAssembleBwd a; int gx, pad_; };
struct MEmbedGather { const int* idx; const float* emb; int rows, n; float* out; int gx, pad_; };
struct MEmbedBwd { const void* idx; const float* d; int ld, col0, rows, n, table_rows, rows_per_block; float* d_emb; int gx, gy; };
struct MAdd2 { const float* a; int lda; const float* b; int ldb; int rows, cols; float* out; int ldo; int gx, pad_; };
struct MZero { void* p; long n16; };          // n16 16-byte words

// variants (every room of a launch must carry the same one; the planner splits a step otherwise)
enum { MV_SCATTER_FWD_64x4 = 0, MV_SCATTER_FWD_32x8 = 1 };
enum { MV_GATHER_32x16 = 0, MV_GATHER_16x16 = 1 };
enum { MV_EMBED_DET = 0, MV_EMBED_LDS = 1, MV_EMBED_PLAIN = 2 };     // + 4: int64 indices
enum { MV_ASM_BWD_LDS = 0, MV_ASM_BWD_PLAIN = 1 };

// plan_*: fill gx / gy (and derived fields) of a room's block on the host, return its variant (< 0: no multi form for these
// arguments - the caller falls back to the room's own launch).  launch_*: `tab` is the DEVICE table of R blocks, (gx, gy) the
// largest per-room grid.
int sln_plan_scatter_avg_fwd(MScatterFwd& a);
int sln_launch_scatter_avg_fwd_multi(const MScatterFwd* tab, int R, int variant, int gx, int gy, hipStream_t st);
int sln_plan_scatter_avg_bwd(MScatterBwd& a);
int sln_launch_scatter_avg_bwd_multi(const MScatterBwd* tab, int R, int variant, int gx, int gy, hipStream_t st);
int sln_plan_gather_bwd(MGatherBwd& a);
int sln_launch_gather_bwd_multi(const MGatherBwd* tab, int R, int variant, int gx, int gy, hipStream_t st);
int sln_plan_mask_gstats(MMaskGstats& a);
int sln_launch_mask_gstats_multi(const MMaskGstats* tab, int R, int gx, int gy, hipStream_t st);
int sln_plan_dec_assemble(MDecAssemble& a);
int sln_launch_dec_assemble_multi(const MDecAssemble* tab, int R, int gx, hipStream_t st);
// the embedding gradients of the decoder's assembled input: LDS-table form (AssembleBwdLds is private to vae_kernels.hip: the
// planner writes the room's block into `blob`, at most SLN_ASM_BLOB bytes) or the plain-atomics form; deterministic mode has no
// multi form here (the engine issues the per-table launches, which have one)
enum { SLN_ASM_BLOB = 256 };
int sln_plan_dec_assemble_bwd(const DecAssembleBwd& a, void* blob, int* gx, int* smem_floats);
int sln_launch_dec_assemble_bwd_multi(const void* tab, int R, int variant, int gx, int smem_floats, hipStream_t st);
int sln_plan_embed_gather(MEmbedGather& a);
int sln_launch_embed_gather_multi(const MEmbedGather* tab, int R, int gx, hipStream_t st);
int sln_plan_embed_bwd(MEmbedBwd& a, int idx64);
int sln_launch_embed_bwd_multi(const MEmbedBwd* tab, int R, int variant, int gx, int gy, int smem_floats, hipStream_t st);
int sln_plan_add2(MAdd2& a);
int sln_launch_add2_multi(const MAdd2* tab, int R, int gx, hipStream_t st);
int sln_launch_zero_multi(const MZero* tab, int R, long max_n16, hipStream_t st);

int sln_launch_copy2d_multi(const MAdd2* tab, int R, int gx, hipStream_t st);       // out[r, :cols] = a[r, :cols] (b unused)
// forward Linears / dgrads of all rooms (gemm_f32.hip).  sln_plan_nt_small: the template key of the body the single-room dispatcher
// would pick for the problem - amode * 16 + epi * 4 + nseg for the 32 x 32 split-K body, 1000 + amode * 16 + epi * 4 + segment class
// for the 64 x 64 body, -1 when there is no multi form (train-mode BatchNorm, chip-filling shapes); `tiles` = its grid.
int sln_plan_nt_small(const GemmNTArgs& a, int epi, int* tiles);
int sln_launch_gemm_nt_small_multi(const GemmNTArgs* tab, const int* tiles_dev, int R, int key, int max_tiles, int max_K, double flops,
                                   hipStream_t st);
