// Launchers of the non-GEMM kernels of the scene-graph VAE path (see vae_kernels.hip).
#pragma once
#include "sln_common.h"

// ---- graph structure (once per batch) -------------------------------------------------------
struct GraphCsr {          // device pointers, all int32 unless noted
  int* s;                  // [T] subject row of each triple
  int* p;                  // [T] predicate id
  int* o;                  // [T] object row
  int* deg;                // [O] #incident (s or o) triples
  float* invdeg;           // [O] 1 / max(deg, 1)            (models/graph.py:102-108)
  int* rowptr;             // [O+1]
  int* cursor;             // [O] scratch
  int* ent;                // [2T] incident entries: e < T -> triple e as subject, e >= T -> triple e-T as object
  int T, O;
};
int sln_launch_graph_prep(const int64_t* triples, int T, int O, int num_preds, GraphCsr g, int* err_flag, hipStream_t st,
                          int edges_only = 0, int deg_is_zero = 0);
// bn view must be aligned with column col0 of x
int sln_launch_bn_relu_apply(const float* x, int ld, int col0, int cols, int rows, BnView bn, float* out, int ldo, hipStream_t st);
// err_flag bits: 1 triple ids, 2 object class, 4 attribute, 8 angle bin out of range
struct StageBatch {
  const int64_t* objs; const int64_t* attrs; const int64_t* angles; const float* boxes;
  int64_t* st_objs; int64_t* st_attrs; int64_t* st_angles; float* st_boxes;
  int* attrs32; int* deg; int* err;
  int O, box_dim, n_objs, n_attrs, n_angle;
};
int sln_launch_stage_batch(const StageBatch& a, hipStream_t st);
int sln_launch_validate_ids(const int64_t* objs, const int64_t* attrs, const int64_t* angles, int O, int n_objs, int n_attrs,
                            int n_angle, int* err_flag, hipStream_t st);

// pooled[i, c] = invdeg[i] * sum_{e in inc(i)} relu(bn(A2))[t_e, role ? H+D+c : c]   (graph.py:86-109)
int sln_launch_scatter_avg_fwd(const float* A2, int ld, int H, int D, BnView bn2, GraphCsr g, int O,
                               float* pooled, hipStream_t st);

// g2[t, :] = mask_relu(bn2(A2)) * [dM[s_t]*invdeg | dP[t] | dM[o_t]*invdeg]; column sums -> gsums
int sln_launch_scatter_avg_bwd(const float* dM, const float* dP, int lddp, int dpcol0, const float* A2, int ld,
                               int H, int D, BnView bn2, GraphCsr g, int T, float* g2, double* gsums,
                               int cstride, hipStream_t st);

// dX[i, c] = sum_{e in inc(i)} dG[t_e, role ? 2D+c : c] (+ add1 + add2), then relu/bn mask of xprev
// (mask_mode 0: no mask, plain gradient).  Column sums of g and g*xhat -> gsums when masked.
int sln_launch_gather_bwd(const float* dG, int ldg, int D, GraphCsr g, int O, const float* add1, int ldadd1,
                          const float* xprev, int ldx, BnView bn, int masked, float* out, int ldo,
                          double* gsums, int cstride, hipStream_t st);

// out[r, c] = a[r, c] + b[r, c]  (c < cols; plain strided sum, e.g. the two head gradients w.r.t. z when decoder_cat is off)
int sln_launch_add2(const float* a, int lda, const float* b, int ldb, int rows, int cols, float* out, int ldo, hipStream_t st);
// generic junction: g = mask(d1 + d2) with column statistics
int sln_launch_mask_gstats(const float* d1, int ld1, const float* d2, int ld2, const float* xprev, int ldx,
                           BnView bn, int rows, int cols, float* out, int ldo, double* gsums, int cstride,
                           hipStream_t st);

// ---- embeddings / assembly --------------------------------------------------------------------
struct EncAssemble {       // X0 = [obj_emb[objs] | attr_emb[attrs] | boxes*Wb^T+bb | angle_emb[angles]]
  const int64_t* objs; const int64_t* attrs; const int64_t* angles; const float* boxes;
  const float* obj_emb; const float* attr_emb; const float* angle_emb; const float* wb; const float* bb;
  int O, n_obj, n_attr, n_box, n_angle, box_dim;
  float* x0;               // [O, n_obj+n_attr+n_box+n_angle]
};
int sln_launch_enc_assemble(EncAssemble a, hipStream_t st);
struct EncAssembleBwd {
  const int64_t* objs; const int64_t* attrs; const int64_t* angles; const float* boxes;
  const float* dx0; int O, n_obj, n_attr, n_box, n_angle, box_dim;
  float* d_obj_emb; float* d_attr_emb; float* d_angle_emb; float* d_wb; float* d_bb;
  int rows_obj, rows_attr, rows_angle;      // table rows (> 0: the tables are accumulated in LDS per block of rows first)
};
int sln_launch_enc_assemble_bwd(EncAssembleBwd a, hipStream_t st);

struct DecAssemble {       // z = eps*exp(.5*logvar)+mu (or mu); X0 = [obj_emb[objs] | attr_emb[attrs] | z]
  const int64_t* objs; const int64_t* attrs;
  const float* obj_emb; const float* attr_emb; const float* mu; const float* logvar; const float* eps;
  const float* z_in;       // when non-null: use this z (decoder() call surface), mu/logvar/eps ignored
  int O, n_obj, n_attr, n_z, use_ae;
  float* z; float* x0;
  int z_in_x0;             // decoder_cat: 1 = X0 = [obj | attr | z] (Sg2ScVAE_model.py:157-159), 0 = X0 = [obj | attr], z joins after the gconv net (:162-164)
};
int sln_launch_dec_assemble(DecAssemble a, hipStream_t st);
struct DecAssembleBwd {    // scatter dX0 into the two embedding grads, pass dz through
  const int64_t* objs; const int64_t* attrs; const float* dx0; int O, n_obj, n_attr, n_z;
  float* d_obj_emb; float* d_attr_emb; float* dz;
  int z_in_x0;             // 0: dx0 is [O, n_obj + n_attr] and dz is not touched
  int rows_obj, rows_attr; // table rows (> 0: LDS accumulation, see EncAssembleBwd)
};
int sln_launch_dec_assemble_bwd(DecAssembleBwd a, hipStream_t st);

// d_emb[idx[r], :] += d[r, col0 : col0+n]   (embedding backward for predicate / attribute tables)
// out[r, :] = emb[idx[r], :]
int sln_launch_embed_gather_i32(const int* idx, const float* emb, int rows, int n, float* out, hipStream_t st);
int sln_launch_embed_bwd_i32(const int* idx, const float* d, int ld, int col0, int rows, int n, int table_rows,
                             float* d_emb, hipStream_t st);
int sln_launch_embed_bwd_i64(const int64_t* idx, const float* d, int ld, int col0, int rows, int n, int table_rows,
                             float* d_emb, hipStream_t st);

// ---- loss (utils.py:12-33) --------------------------------------------------------------------
struct LossArgs {
  const float* boxes; const float* boxes_pred; int box_dim;
  const int64_t* angles; const float* logits; float* angles_pred; int n_angle;   // log_softmax written here
  const float* mu; const float* logvar; int n_z; int use_ae;
  const float* kl_weight;  // device scalar
  int O;
  double* acc;             // [4] zeroed: sum|db|, sum nll, sum kl-term, unused
  float* losses;           // [4]: bbox, angle, kl*w, total  (written by finalize)
  float* d_boxes_pred; float* d_logits;   // may be nullptr (forward only)
  int ld_dbp;              // row stride of d_boxes_pred (padded to a multiple of 4)
  int acc_prezeroed;       // the caller already cleared acc (one bulk memset per training iteration)
  int from_logits;         // angles_pred = log_softmax(logits) is computed HERE (the fused iteration: no log_softmax launch)
};
// d_logits = d_logprob - softmax * rowsum(d_logprob)   (backward of log_softmax given grad of its output)
int sln_launch_log_softmax_bwd(const float* logprob, const float* d_logprob, float* d_logits, int O, int n, hipStream_t st);
// convert int64 ids to int32 (attributes are used as a GEMM row-gather index)
int sln_launch_i64_to_i32(const int64_t* src, int* dst, int n, hipStream_t st);
int sln_launch_log_softmax(const float* logits, float* out, int O, int n, hipStream_t st);
int sln_launch_loss(LossArgs a, hipStream_t st);
// dmu = w*mu/O + dz ; dlogvar = w*0.5*(exp(lv)-1)/O + dz*eps*0.5*exp(0.5 lv)
int sln_launch_latent_bwd(const float* mu, const float* logvar, const float* eps, const float* dz,
                          const float* kl_weight, int O, int n_z, int use_ae, float* dmu, float* dlogvar,
                          hipStream_t st);

// ---- parameters -------------------------------------------------------------------------------
struct BnTableEntry {      // one BatchNorm application (module may repeat in 'recurrent' mode)
  const double* sums; const double* gsums; int cstride; int C; int rows;
  float* rmean; float* rvar; int64_t* nbt; float* dgamma; float* dbeta;
};
// independent != 0: no BatchNorm module occurs twice in the table (feedforward mode) -> one block row per entry
// table rows == -1 / -2: the batch's triple / object count (rows_t / rows_o), rows > 0: as given
int sln_launch_bn_running_update(const BnTableEntry* table, int n, int max_c, float momentum, int independent,
                                 hipStream_t st, int rows_t = 0, int rows_o = 0);
int sln_launch_bn_param_grads(const BnTableEntry* table, int n, int max_c, int independent, hipStream_t st);

struct TransposeEntry { const float* src; float* dst; int rows; int cols; int dst_ld; int pad_; };   // dst[c*dst_ld + r] = src[r*cols + c]
int sln_launch_transpose_table(const TransposeEntry* table, int n, int max_tiles, hipStream_t st);

struct AdamScalars {
  int64_t step; float lr, beta1, beta2, eps; float kl_weight; float bc1, bc2; int skip, pad_;
  // Philox stream of the reparameterisation draw (Sg2ScVAE_model.py:182): key = seed, counter = (element / 4, offset);
  // the draw kernel itself advances `rng_offset`, so a replayed hipGraph takes a fresh draw every iteration
  unsigned long long rng_seed, rng_offset; unsigned int rng_done, adam_done;      // arrival tickets of the draw / the update
};
// total_loss (device, may be NULL): a non-finite value skips the update and the step count (train.py:79-81 'not backpropping')
int sln_launch_adam(float* params, const float* grads, float* m, float* v, long n, AdamScalars* scalars, const float* total_loss,
                    hipStream_t st);

// eps[i] ~ N(0, 1), i < n: Philox-4x32-10 + Box-Muller, 4 values per counter; the last block to finish advances
// scalars->rng_offset by one (one offset per draw: streams of different iterations never overlap)
int sln_launch_randn(float* eps, long n, AdamScalars* scalars, hipStream_t st);

// The head of a fused training iteration as ONE launch (round 3; it was four): the N(0,1) draw (eps == NULL: skipped), the
// encoder's input assembly and BOTH predicate-embedding gathers (the decoder's does not depend on the encoder).
struct StepPrologue {
  float* eps; long n_eps; AdamScalars* scalars;
  EncAssemble enc;
  const int* pidx; int T;
  const float* pemb_ec; int n_ec; float* p0e;
  const float* pemb_dc; int n_dc; float* p0d;
  void* zero_ptr; long zero_bytes;      // optional: a region to clear (16-byte aligned, a multiple of 16 bytes): the iteration's loss accumulators and BatchNorm sums
};
int sln_launch_step_prologue(const StepPrologue& a, hipStream_t st);
// counts[obj][floor(cz (cs - 1))][floor(cx (cs - 1))] += 1 for every (trial, object != room row): testing/test_heatmap.py:80-99
int sln_launch_layout_heatmap(const float* boxes, long n_trials, int O, int cs, int clip, float* counts, hipStream_t st);
