/* sln_hip.h - C ABI of libsln_hip.so, the MI355X (gfx950) implementation of the 3D_SLN hot path.
 *
 * The reference (aluo-x/3D_SLN) has no FFI layer: its boundary for this path is a set of Python
 * call surfaces.  This header is what those surfaces bind to; the Python mirror
 * in 3d_sln_amd/host/ loads the library with ctypes and passes raw device pointers
 * (torch.Tensor.data_ptr()) plus the current HIP stream.  No torch types cross this boundary.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; tensors are dense
 *     row-major fp32 unless stated; index tensors are int64 exactly as the reference's
 *     suncg_collate_fn produces them (data/suncg_dataset.py:295-337);
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued, nothing synchronises;
 *   - every function returns 0 on success, a hipError_t value (>0) for a HIP failure, or a
 *     negative SLN_E_* code for a contract violation (nothing is launched in that case).
 *
 * Each entry point cites the reference interface it stands behind (paths relative to the
 * reference repository root).
 */
#ifndef SLN_HIP_H
#define SLN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLN_E_BADARG (-1)      /* null pointer / size that violates the contract below            */
#define SLN_E_UNSUPPORTED (-2) /* configuration outside the HIP path (e.g. gconv_num_layers == 0) */
#define SLN_E_STATE (-3)       /* call order violated (e.g. backward before forward)             */
#define SLN_E_NOGPU (-4)       /* no gfx950 device visible                                       */
#define SLN_E_NOMEM (-5)       /* a host staging buffer / event could not be allocated           */
#define SLN_E_CAPTURE (-6)     /* the caller's stream is being captured and the call needs an eager upload (new batch shape) */

int sln_version(void);                 /* ABI version, bumped on any signature change */
const char* sln_build_arch(void);      /* "gfx950" */
int sln_device_ok(void);               /* 0 when a gfx950 device is usable, SLN_E_NOGPU otherwise */

/* =============================================================================================
 * A. Scene-graph VAE  (models/graph.py:10-143, models/Sg2ScVAE_model.py:7-188, utils.py:12-33,
 *    train.py:62-84)
 * ============================================================================================= */

typedef struct SlnVaeConfig {
  int embedding_dim;     /* Sg2ScVAE_model.py:8; must be a multiple of 16                         */
  int gconv_num_layers;  /* >= 1                                                                   */
  int recurrent;         /* gconv_mode == 'recurrent' (weights shared by all layers)               */
  int batch_norm;        /* mlp_normalization == 'batch' (1) or 'none' (0)                         */
  int decoder_cat;       /* 1: z is concatenated in front of the decoder's gconv net (train.py default,
                          * options/options.py:55); 0: behind it (the class default, Sg2ScVAE_model.py:9,157-164) */
  int use_ae;            /* use_AE: z = mu, no KL term                                             */
  int box_dim;           /* 6 (train_3d) or 4                                                      */
  int n_angle;           /* Nangle = 24                                                            */
  int num_objs;          /* rows of obj_embeddings_* (len(object_idx_to_name) + 1)                 */
  int num_preds;         /* rows of pred_embeddings_*                                              */
  int num_attrs;         /* rows of attr_embedding_* (0 with no_attr)                              */
  int no_attr;           /* 1 = use_attr=False (Sg2ScVAE_model.py:17,35-37,48-50,97-98): no attribute
                          * embeddings, the class embedding is E wide, box_net reads 2E columns     */
} SlnVaeConfig;

/* Number of Linear(+BatchNorm) units and their canonical order:
 *   box_mean_var.{0,1}, box_mean.0, box_var.0, angle_mean_var.{0,1}, angle_mean.0, angle_var.0,
 *   gconv_net_ec.gconvs.i.{net1.0, net1.1, net2.0, net2.1} for each module i,
 *   gconv_net_dc.gconvs.i.{...}, box_net.{0,1}, angle_net.{0,1}
 * ("X.1" is the second Linear of the MLP, state_dict index 3 with BatchNorm, 2 without). */
int sln_vae_num_units(const SlnVaeConfig* cfg);

typedef struct SlnVaeUnit {        /* one Linear [+ BatchNorm1d] */
  float* weight;  float* bias;                 /* [out, in], [out]                               */
  float* bn_weight; float* bn_bias;            /* [out] or NULL                                  */
  float* bn_running_mean; float* bn_running_var; int64_t* bn_num_batches_tracked;
  float* d_weight; float* d_bias; float* d_bn_weight; float* d_bn_bias;   /* gradients (+=)      */
} SlnVaeUnit;

typedef struct SlnVaeTensors {
  /* embeddings, Sg2ScVAE_model.py:44-58 (parameter, gradient) */
  float* obj_emb_ec;  float* d_obj_emb_ec;
  float* pred_emb_ec; float* d_pred_emb_ec;
  float* obj_emb_dc;  float* d_obj_emb_dc;
  float* pred_emb_dc; float* d_pred_emb_dc;
  float* attr_emb_ec; float* d_attr_emb_ec;
  float* attr_emb_dc; float* d_attr_emb_dc;
  float* box_emb_w;   float* d_box_emb_w;
  float* box_emb_b;   float* d_box_emb_b;
  float* angle_emb;   float* d_angle_emb;
  const SlnVaeUnit* units_host;    /* HOST array of sln_vae_num_units() entries                  */
  /* flat views used by zero_grad / Adam / the data-parallel all-reduce: every parameter above
   * must live inside [flat_params, flat_params + n_flat) at the same offset as its gradient in
   * flat_grads.  adam_m / adam_v may be NULL when sln_vae_train_step is never called.           */
  float* flat_params; float* flat_grads; float* adam_m; float* adam_v; int64_t n_flat;
} SlnVaeTensors;

typedef struct SlnVaeBatch {       /* suncg_collate_fn's tuple, data/suncg_dataset.py:327-337     */
  const int64_t* objs;       /* [O]                                                              */
  const int64_t* triples;    /* [T,3] (s, p, o) with global row ids                              */
  const float*   boxes;      /* [O, box_dim]                                                     */
  const int64_t* angles;     /* [O]                                                              */
  const int64_t* attributes; /* [O]                                                              */
  int O, T;
} SlnVaeBatch;

typedef struct SlnVae SlnVae;      /* opaque engine: kernel plan + workspace carve-up             */

int sln_vae_create(const SlnVaeConfig* cfg, SlnVae** out);
void sln_vae_destroy(SlnVae* h);
/* bytes of device workspace needed for batches up to (max_objs, max_triples) */
int64_t sln_vae_workspace_bytes(const SlnVae* h, int max_objs, int max_triples);
/* workspace must be zero-filled once by the caller - the fill must have COMPLETED: bind writes into it with blocking copies
 * that are not ordered behind work on other streams - and stay alive while the engine is used */
int sln_vae_bind(SlnVae* h, const SlnVaeTensors* t, void* workspace, int64_t workspace_bytes,
                 int max_objs, int max_triples);

/* Build the per-batch graph structure (int32 ids, degrees, CSR of incident triples).  Must
 * precede encoder/decoder calls for a new batch.  Replaces the index/scatter_add bookkeeping of
 * GraphTripleConv.forward (models/graph.py:70-72,89-108), hoisted out of the 10 layers.  All five inputs are copied
 * (stream-ordered) into the workspace: the caller's tensors are not read after this call, and a captured iteration
 * (sln_vae_train_step with use_graph) replays on whatever batch was bound last. */
int sln_vae_set_batch(SlnVae* h, const SlnVaeBatch* b, void* stream);
/* Synchronising check of the bound batch: SLN_E_BADARG when a triple / class / attribute / angle id is out of range
 * (the reference's embedding and index ops raise IndexError there; the kernels neutralise such rows). */
int sln_vae_check_batch(SlnVae* h, void* stream);

/* Sg2ScVAEModel.encoder (Sg2ScVAE_model.py:115-143): writes mu, logvar [O, embedding_dim].
 * training != 0: BatchNorm uses batch statistics and updates running stats (train()). */
int sln_vae_encoder(SlnVae* h, float* mu, float* logvar, int training, void* stream);
/* Sg2ScVAEModel.decoder (:145-172).  z [O, embedding_dim]; boxes_pred [O, box_dim];
 * angles_pred [O, n_angle] = log_softmax. */
int sln_vae_decoder(SlnVae* h, const float* z, float* boxes_pred, float* angles_pred, int training, void* stream);
/* Sg2ScVAEModel.forward (:174-188) with the N(0,1) draw supplied by the caller (eps [O, E];
 * torch.randn_like in the reference).  z_out may be NULL. */
int sln_vae_forward(SlnVae* h, const float* eps, float* mu, float* logvar, float* z_out, float* boxes_pred,
                    float* angles_pred, int training, void* stream);

/* calculate_model_losses (utils.py:12-33): losses_out[4] = {bbox_pred, angle_pred, KLD*w, total}.
 * When with_grads != 0 the gradients w.r.t. boxes_pred / logits are kept for sln_vae_backward. */
int sln_vae_loss(SlnVae* h, const float* boxes_pred, const float* angles_pred, const float* mu,
                 const float* logvar, float kl_weight, float* losses_out, int with_grads, void* stream);

/* Backward of the decoder given d(boxes_pred) [O, box_dim] and d(angles_pred) [O, n_angle]
 * (gradient w.r.t. the log-softmax OUTPUT).  Accumulates parameter gradients, writes dz [O, E]. */
int sln_vae_decoder_backward(SlnVae* h, const float* d_boxes_pred, const float* d_angles_pred, float* dz, void* stream);
/* Backward of the encoder given d(mu), d(logvar) [O, E]; accumulates parameter gradients. */
int sln_vae_encoder_backward(SlnVae* h, const float* d_mu, const float* d_logvar, void* stream);
/* total_loss.backward() of train.py:83 after sln_vae_forward + sln_vae_loss(with_grads=1). */
int sln_vae_backward(SlnVae* h, void* stream);

/* Tell the engine that parameters were modified outside of it (load_state_dict, a torch optimizer):
 * cached transposed weights are rebuilt before the next backward. */
int sln_vae_params_changed(SlnVae* h);
int sln_vae_zero_grad(SlnVae* h, void* stream);                       /* optimizer.zero_grad(), train.py:82 */
/* torch.optim.Adam(lr).step(), train.py:15,84 (betas .9/.999, eps 1e-8, no weight decay) */
int sln_vae_adam_step(SlnVae* h, float lr, void* stream);
int sln_vae_adam_reset(SlnVae* h, int64_t step, void* stream);       /* restore the step counter (checkpoint resume) */
/* The device's own step counter (synchronises): after an iteration skipped for a non-finite loss it is one behind
 * the number of sln_vae_adam_step calls (train.py:79-81 skips the optimizer step too). */
int sln_vae_adam_get_step(SlnVae* h, int64_t* step_out, void* stream);
/* Data-parallel NaN guard (the reference is single-GPU: build_dataset_model.py:54-55 asserts).  `slot` = one device float
 * that the trainer all-reduces together with flat_grads (in practice flat_grads[n_flat]): SLN_TRAIN_BACKWARD /
 * SLN_TRAIN_UPTO_DECODER store the rank's total loss there, sln_vae_adam_step skips the update when the slot - by then
 * the average over the ranks - is not finite, so all replicas skip or step together.  NULL restores the per-rank guard.
 * Call while the engine is idle (captured iterations are dropped). */
int sln_vae_set_grad_guard(SlnVae* h, float* slot);
/* The reparameterisation draw, Sg2ScVAE_model.py:180-183 (eps = randn_like(std)).  sln_vae_forward / sln_vae_train_step
 * called with eps == NULL draw eps on the device: Philox-4x32-10 keyed by `seed`, one `offset` per draw (advanced by
 * the draw kernel itself, so a replayed hipGraph sees a new draw every iteration).  sln_vae_last_eps copies the eps of
 * the last forward / iteration (the injected one or the drawn one) to eps_out[O, embedding_dim]. */
int sln_vae_seed(SlnVae* h, uint64_t seed, uint64_t offset, void* stream);
int sln_vae_last_eps(SlnVae* h, float* eps_out, void* stream);
/* n values ~ N(0,1) from the engine's Philox stream (one offset per call, as the reparameterisation draws): the z draws of
 * posterior sampling (testing/test_VAE.py:83-84, testing/test_heatmap.py:57-58 call np.random.multivariate_normal on the host and
 * copy the sample to the device once per decode) */
int sln_vae_randn(SlnVae* h, float* out, int64_t n, void* stream);
/* The accumulation of testing/test_heatmap.py:80-99 for all trials in one launch: boxes_pred [n_trials, O, 6] (room row last in
 * every trial) -> counts [O - 1, container_size, container_size] += 1 at (floor(cz (cs - 1)), floor(cx (cs - 1))) of every object
 * centre in the room's own frame (clip_coor: centres clamped to [0, 1]; otherwise trials with a centre outside (0, 1) are dropped). */
int sln_layout_heatmap(const float* boxes_pred, int64_t n_trials, int O, int box_dim, int container_size, int clip_coor, float* counts,
                       void* stream);

/* One iteration of the train.py:62-84 loop on the bound batch: zero_grad, forward, loss, backward and
 * (with_adam == SLN_TRAIN_FULL) the Adam update.  `use_graph`: replay a captured hipGraph while shapes are unchanged
 * (needs a non-default stream).  The data-parallel trainer either calls it with SLN_TRAIN_BACKWARD, all-reduces
 * flat_grads and calls sln_vae_adam_step, or - to overlap the collective with the backward pass - runs the iteration
 * in two halves: SLN_TRAIN_UPTO_DECODER returns (enqueues) everything up to the point where the gradients of the
 * decoder-side parameters are final, SLN_TRAIN_ENCODER_BWD the rest; the all-reduce of the decoder half of flat_grads
 * runs on another stream in between.  `losses_out` is written by every mode except SLN_TRAIN_ENCODER_BWD. */
enum { SLN_TRAIN_BACKWARD = 0, SLN_TRAIN_FULL = 1, SLN_TRAIN_UPTO_DECODER = 2, SLN_TRAIN_ENCODER_BWD = 3 };
/* BatchNorm mode of sln_vae_train_step: 1 (default) = batch statistics + running-stat update, 0 = running statistics as a fixed
 * affine map (train.py:63-65: after --eval_mode_after the reference calls model.eval() and keeps training). */
int sln_vae_set_training(SlnVae* h, int training);
int sln_vae_train_step(SlnVae* h, const float* eps, float kl_weight, float lr, float* losses_out, int use_graph,
                       int with_adam, void* stream);

/* R rooms in flight: the decoder of R engines as one launch sequence (layout refinement, testing/test_render_refine.py:250-263:
 * every trial - room - reloads the checkpoint, model.eval(), and :279-359 runs 60 dependent iterations of
 * decoder -> render -> loss -> backward -> SGD on z AND on its own copy of the parameters; trials are independent).
 * `engines`: R handles created and bound by the caller on R parameter copies (sln_vae_create / sln_vae_bind) with their rooms'
 * graphs set (sln_vae_set_batch; the boxes / angles of a decoder-only room are not read); same SlnVaeConfig; eval-mode BatchNorm.
 * The group takes the engines over: their decoder outputs and gradient inputs become slices [row0[r], row0[r] + O_r) of the
 * row-concatenated arrays below, and sln_vae_group_decoder / _backward run sln_vae_decoder(training = 0) /
 * sln_vae_decoder_backward of all rooms - one launch per step for all rooms instead of one per step and room.  A room's results
 * do not depend on R (same kernel bodies, one room per grid slice).  The engines must outlive the group; destroy the group first. */
typedef struct SlnVaeGroup SlnVaeGroup;
typedef struct SlnVaeGroupIO {
  int rows_total;                 /* sum over the rooms of their object rows                                            */
  const int* row0_host;           /* HOST [R]: first row of room r                                                       */
  const float* z;                 /* [rows_total, E]        decoder input, read by every forward                         */
  float* boxes_pred;              /* [rows_total, box_dim]  written by forward                                           */
  float* angles_pred;             /* [rows_total, n_angle]  log-softmax, written by forward, read by backward            */
  float* d_boxes_pred;            /* [rows_total, 8]        gradient w.r.t. boxes_pred, row stride 8 (columns >= box_dim 0) */
  float* d_angles_pred;           /* [rows_total, n_angle]  gradient w.r.t. the log-softmax output                       */
  float* dz;                      /* [rows_total, E]        written by backward                                          */
  const float* sgd_step;          /* optional DEVICE scalar: when set, the wgrads of backward apply  W -= step * dW,  b -= step * db
                                   * to every room's Linear parameters in their epilogue instead of accumulating into the
                                   * gradient buffers (sln_vae_group_fused_params lists the tensors; the caller's optimizer
                                   * steps the others: sln_refine_sgd_rooms over the remaining ranges)                     */
} SlnVaeGroupIO;
/* While a group exists its engines' decoder outputs / gradient inputs are slices of the group's arrays; a failed create and
 * sln_vae_group_destroy put the engines' own buffers back (destroy a group BEFORE its engines). */
int sln_vae_group_create(SlnVae* const* engines, int R, const SlnVaeGroupIO* io, SlnVaeGroup** out);
/* forward also rebuilds W^T of the decoder's weights for the following backward; a backward that is not preceded by a forward of
 * the same group since the last backward (the parameters have been stepped in between) rebuilds it itself.  Parameters changed by
 * the caller BETWEEN a forward and its backward are not noticed. */
int sln_vae_group_decoder(SlnVaeGroup* g, void* stream);
/* parameter gradients are accumulated (+=) into every room's gradient buffer (but see SlnVaeGroupIO::sgd_step), dz is written */
int sln_vae_group_decoder_backward(SlnVaeGroup* g, void* stream);
/* the tensors of ROOM 0's parameter copy that backward steps itself (SlnVaeGroupIO::sgd_step; the same tensors of every room):
 * returns their number, writes at most `max` (pointer, element count) pairs */
int sln_vae_group_fused_params(const SlnVaeGroup* g, const float** params, int64_t* numel, int max);
/* launches per forward / backward pass and how many of them are single-room fallbacks (diagnostics) */
int sln_vae_group_launches(const SlnVaeGroup* g, int* fwd, int* bwd, int* single_room_fallbacks);
/* how many times the group has built W^T of the decoder's weights so far (one per forward, plus one per backward that found none valid) */
int64_t sln_vae_group_transposes(const SlnVaeGroup* g);
void sln_vae_group_destroy(SlnVaeGroup* g);

/* Diagnostics: the pooled side stream the library runs next to `stream` (wgrads of a room group, the depth chain of the scene
 * backward): its index in the per-device pool and whether a probe sees the two streams overlap.  Two streams that share a hardware
 * queue do not - the runtime deals streams to a few queues round-robin - so the library probes once per caller stream and keeps a
 * pooled stream that does (csrc/streams.hip). */
int sln_debug_side_stream(void* stream, int* index, int* overlapped);
/* The probe is a HOST SYNCHRONISATION of `stream` (hipStreamSynchronize + a ~150 us spin kernel per candidate).  Without a prepare call
 * it happens inside the first sln_scene_backward / sln_vae_group_decoder[_backward] on a stream (never inside a capture: a stream first
 * met while captured gets the pool's first stream unprobed).  sln_side_stream_prepare probes now: 1 = a pooled stream overlaps with
 * `stream`, 0 = none (side work runs on `stream`), < 0 error (SLN_E_STATE inside a capture).  A pick is probed again after 4 096
 * look-ups; sln_side_stream_forget drops it (call it before destroying the stream: the next stream with the same handle may sit on
 * another hardware queue); returns the number of entries dropped. */
int sln_side_stream_prepare(void* stream);
int sln_side_stream_forget(void* stream);

/* Per-kernel-family timing with HIP events on the launch stream (bench.py roofline figures).
 * Families: 0 gemm_nt (forward/dgrad), 1 gemm_tn (wgrad), 2 edge scatter/gather, 3 other.
 * enable=1 starts recording (eager launches only, not under graph capture); sln_prof_read
 * synchronises, sums and clears the recorded intervals. work = FLOPs for GEMM families, bytes else. */
int sln_prof_enable(int enable);
/* Deterministic mode (also: environment SLN_DETERMINISTIC=1 at load time).  The reference's CPU path is run-to-run deterministic;
 * the default HIP path adds partial sums with atomics in arrival order (wgrad row chunks, embedding tables, the renderer's face
 * gradients).  With the switch on, every such sum is taken in a fixed order: two runs on the same inputs are bit-identical
 * (scene-graph VAE step and scene backward; cost in DESIGN.md).  Takes effect at the next launch; captured iterations are
 * re-captured. */
int sln_set_deterministic(int on);
int sln_get_deterministic(void);
/* Test hook (host only, no device work): the workgroup table of the per-pass wgrad launch (csrc/sln_gemm.h: TnMultiMeta) for n
 * problems given as (rows, outputs, inputs).  Returns the number of workgroups (< 0: error), fills rows_per_block[n] and
 * items[3 * workgroup] = (problem or -1 for padding, output tile, row chunk). */
int sln_debug_tn_plan(const int* R, const int* Nout, const int* Kin, int n, int* rows_per_block, int* items, int max_items);
int sln_prof_read(double* ms_by_family, double* work_by_family, int64_t* launches_by_family, int n_families);

/* Debug/test tap: copy an internal activation to `dst` (device).  what: 0 A1,1 A2,2 M,3 A3,4 A4 of
 * gconv instance `layer` (0..L-1 encoder, L..2L-1 decoder).  Returns the element count or <0. */
int64_t sln_vae_tap(SlnVae* h, int layer, int what, float* dst, void* stream);

/* GraphTripleConv.forward / GraphTripleConvNet.forward (models/graph.py:57-111,136-143) on their own (inference /
 * feature extraction; training goes through the VAE engine).  units_host: 4 SlnVaeUnit per module in the order
 * net1.0, net1.1, net2.0, net2.1 (gradient fields unused); layer l uses module (n_modules == 1 ? 0 : l).
 * edges [T,2] int64 (s, o).  D % 32 == 0; num_layers > 1 requires Dout == D. */
int64_t sln_gconv_workspace_bytes(int D, int H, int Dout, int O, int T, int num_layers);
int sln_gconv_forward(int D, int H, int Dout, int num_layers, int n_modules, int batch_norm, const SlnVaeUnit* units_host,
                      const float* obj_vecs, const float* pred_vecs, const int64_t* edges, int O, int T, int training, void* workspace,
                      int64_t workspace_bytes, float* new_obj, float* new_pred, void* stream);

/* GraphTripleConvNet on its own WITH autograd (models/graph.py:57-143 is differentiable): an engine handle whose units are
 * the modules' four Linears (net1.0, net1.1, net2.0, net2.1 per module; `recurrent` = one shared module).  Dout != D (a bare
 * GraphTripleConv(input_dim, output_dim), models/graph.py:36-56) needs num_layers == 1.
 * Life cycle: sln_gconv_net_create -> sln_vae_workspace_bytes / sln_vae_bind (SlnVaeTensors: only units_host, with the
 * gradient pointers filled) -> per graph sln_gconv_net_set_edges -> sln_gconv_net_forward -> sln_gconv_net_backward (parameter
 * gradients are accumulated into the units' d_* pointers; input gradients written) -> sln_vae_destroy. */
int sln_gconv_net_create(int D, int H, int Dout, int num_layers, int recurrent, int batch_norm, SlnVae** out);
int sln_gconv_net_set_edges(SlnVae* h, const int64_t* edges, int O, int T, void* stream);
int sln_gconv_net_forward(SlnVae* h, const float* obj_vecs, const float* pred_vecs, float* new_obj, float* new_pred, int training,
                          void* stream);
int sln_gconv_net_backward(SlnVae* h, const float* d_new_obj, const float* d_new_pred, float* d_obj_vecs, float* d_pred_vecs,
                           void* stream);

/* Standalone Linear kernels (parity tests of the GEMM family):
 *   y[M,N] = x[M,K] W[N,K]^T + bias, optional fp64 column sums of y and y^2 in sums[2][N] */
int sln_linear_forward(const float* x, int M, int K, const float* W, const float* bias, float* y, int N, double* sums,
                       int tile, void* stream);
/*   dW[N,K] += g[R,N]^T x[R,K];  db[N] += colsum(g) */
int sln_linear_wgrad(const float* g, const float* x, int R, int N, int K, float* dW, float* db, void* stream);


/* =============================================================================================
 * B. Differentiable rasterizer  (third-party `neural_renderer`, un-vendored; reference call sites
 *    models/misc.py:7, models/diff_render.py:359-361 (ctor), :366 (mode='depth'), :398 (mode="rgb"))
 *
 * The host mirror does the camera projection / fill_back / vertices_to_faces with torch ops; the C ABI
 * starts at faces[B, F, 3, 3] fp32 = (x_ndc, y_ndc, z_cam) of the three vertices of every face.
 * All maps are [B, is, is] in RASTER order (row 0 = bottom; the package flips rows afterwards).
 * ============================================================================================= */
/* neural_renderer.projection (K [B,3,3], R [B,3,3], t [B,1,3], orig_size; dist_coeffs dropped as the reference README asks,
 * README.md:12-18) followed by vertices_to_faces: faces_xyz [B,F,3,3] = (x_ndc, y_ndc, z_cam) per corner; and its adjoint. */
int sln_project_faces(const float* vertices, const int32_t* faces, const float* K, const float* R, const float* t, int B, int V, int F,
                      float orig_size, float eps, float* faces_xyz, void* stream);
int sln_project_faces_backward(const float* vertices, const int32_t* faces, const float* K, const float* R, const float* t, int B, int V,
                               int F, float orig_size, float eps, const float* grad_faces_xyz, float* grad_vertices, void* stream);
int64_t sln_raster_workspace_bytes(int B, int F);
/* rasterize / rasterize_depth: face_index int32 (-1 = background), weight [B,is,is,3], depth (far where empty) */
int sln_raster_forward(const float* faces, int B, int F, int image_size, float near, float far, void* workspace,
                       int32_t* face_index, float* weight, float* depth, void* stream);
/* one geometry pass, two z-buffers with different near planes (depth pass 0.1 / rgb passes ctor value) */
int sln_raster_forward_dual(const float* faces, int B, int F, int image_size, float near_a, float near_b, float far,
                            void* workspace, int32_t* fi_a, float* w_a, float* d_a, int32_t* fi_b, float* w_b,
                            float* d_b, void* stream);
/* forward_texture_sampling: textures [B,F,ts,ts,ts,3] -> rgb [B,is,is,3] (background 0) */
int sln_raster_texture_sample(const float* faces, const float* textures, const int32_t* face_index, const float* weight,
                              const float* depth, int B, int F, int image_size, int texture_size, float eps, float* rgb,
                              void* stream);
/* backward_depth_map: grad_faces[B,F,3,3] += ... (atomics) */
int sln_raster_backward_depth(const float* faces, const int32_t* face_index, const float* weight, const float* depth,
                              const float* grad_depth, int B, int F, int image_size, float* grad_faces, void* stream);
/* backward_pixel_map for a `channels`-channel image rgb/grad_rgb [B,is,is,channels]: grad_faces += ... */
int sln_raster_backward_rgb(const float* faces, const int32_t* face_index, const float* rgb, const float* grad_rgb, int B,
                            int F, int image_size, int channels, float eps, float* grad_faces, void* stream);
/* Renderer.__call__(mode="rgb") (models/diff_render.py:398) behind cached maps, one launch: forward_texture_sampling + ambient
 * light (factor `scale`) + the package's permute / row flip -> rgb_chw [B,3,is,is].  textures [B,F_tex,ts,ts,ts,3] with
 * F_tex == F, or F_tex == F / 2 for fill_back: face f >= F_tex samples cube f - F_tex with texture axes 0 and 2 swapped
 * (the package's torch.cat((textures, textures.permute((0, 1, 4, 3, 2, 5))), dim=1)). */
int sln_raster_texture_sample_chw(const float* faces, const float* textures, const int32_t* face_index, const float* weight,
                                  const float* depth, int B, int F, int F_tex, int image_size, int texture_size, float eps, float scale,
                                  float* rgb_chw, void* stream);
/* backward_pixel_map of P <= 64 rgb passes rendered over the SAME face-index map (the 32 class passes of mesh_render_func,
 * models/diff_render.py:381-398, back-propagated together): grad_faces += sum over the passes of the package's per-pass
 * gradient, in one edge walk.  rgb_chw / grad_chw: device arrays of P device pointers to [B,3,is,is] images as returned by
 * sln_raster_texture_sample_chw / handed back by autograd; mask_ws: 8 * B * is * is bytes of scratch. */
int sln_raster_backward_rgb_multi(const float* faces, const int32_t* face_index, const float* const* rgb_chw,
                                  const float* const* grad_chw, int P, int B, int F, int image_size, float eps, void* mask_ws,
                                  float* grad_faces, void* stream);

/* Fused scene pass = models/diff_render.py:359-434 (1 depth + one rgb pass per class, masks, per-class mean
 * depth, wall_max normalisation, 70-channel layout) in ONE rasterisation.
 *   face_class [B,F] int32: class id (0 = wall, ids follow the reference's sorted class list) or -1
 *   class_channel [num_classes]: NYU-40 index of each class (final channel 1 + index)
 *   class_depth_channel [num_classes]: index into the 29 depth channels (final channel 41 + index) or -1
 *   final_out [B,70,is,is] image order (rows flipped), grad_faces [B,F,3,3] (overwritten). */
int64_t sln_scene_workspace_bytes(int B, int F, int image_size);
int sln_scene_forward(const float* faces, const int32_t* face_class, int B, int F, int image_size, int num_classes,
                      const int32_t* class_channel, const int32_t* class_depth_channel, float near_depth, float near_rgb,
                      float far, float tex_eps, void* workspace, float* final_out, void* stream);
/* sln_scene_forward for a consumer that reads the image through its flags (SlnRefineLoss::live_planes): `live` [B, 70] receives
 * the flags described below, and only the planes flagged 3 of final_out are written (a plane flagged 0 would hold zeros, a plane
 * flagged 1 the constant 1: their memory is left as it was). */
int sln_scene_forward_live(const float* faces, const int32_t* face_class, int B, int F, int image_size, int num_classes,
                           const int32_t* class_channel, const int32_t* class_depth_channel, float near_depth, float near_rgb,
                           float far, float tex_eps, void* workspace, float* final_out, unsigned char* live,
                           unsigned char* null_mask /* optional [B, S, S]: SlnRefineLoss::null_mask */, void* stream);
/* live [B, 70] (bytes) of the last sln_scene_forward on `workspace`, two bits per channel: bit 0 clear = the plane is all zeros
 * (semantic channels of classes without a visible pixel in that image), bit 1 clear = sln_scene_backward never reads the
 * plane's gradient (those, and the depth-hot planes of such classes, which hold the constant 1).  The refinement loss skips
 * what the bits allow (SlnRefineLoss::live_planes). */
int sln_scene_live_channels(void* workspace, int B, int F, int image_size, int num_classes, const int32_t* class_channel,
                            const int32_t* class_depth_channel, unsigned char* live, void* stream);
int sln_scene_backward(const float* faces, const int32_t* face_class, int B, int F, int image_size, int num_classes,
                       const int32_t* class_channel, const int32_t* class_depth_channel, float pix_eps, void* workspace,
                       const float* grad_final, float* grad_faces, void* stream);


/* Object placement of mesh_render_func for one room with device-resident meshes (models/diff_render.py:76-159), fused with
 * the projection (as sln_project_faces), the near-plane cull (:346-356, culled faces are degenerated to a point) and
 * fill_back: boxes [n,6] (room-normalised rows, visible object k is row vis[k]), angles [n] (bins, fractional) ->
 * faces_out [2F,3,3] = (x_ndc, y_ndc, z_cam) (faces F..2F-1: corners (2,1,0)), sizes [n_vis,3], size_loss [1] =
 * sum_k mean_j (size - size_target)^2 (0 when size_target is NULL).  The face list is grouped by object (obj_face_ptr
 * [n_vis+1], the room shell's faces last); object faces index model_v [n_vis*Vm,3], shell faces n_vis*Vm + shell_v [Vs,3].
 * backward: grad_faces [2F,3,3], grad_size_loss [1] (device, may be NULL) -> grad_boxes [n,6], grad_angles [n]. */
typedef struct {
  int n, n_vis, Vm, Vs, F, reserved;
  const int32_t* vis; const float* model_v; const float* msize; const float* mcenter; const float* shell_v;
  const int32_t* faces; const int32_t* obj_face_ptr;
  float ext[3];                    /* room extent (boxes[-1][3:]) */
  float K[9], R[9], t[3];          /* camera of get_cam_mat (diff_render.py:13-46) */
  float orig_size, proj_eps, cull_eps;
} SlnPlacement;
int sln_place_forward(const SlnPlacement* P /* host struct */, const float* boxes, const float* angles, const float* size_target,
                      float* faces_out, float* sizes, float* size_loss, void* stream);
int sln_place_backward(const SlnPlacement* P, const float* boxes, const float* angles, const float* size_target, const float* grad_faces,
                       const float* grad_size_loss, float* grad_boxes, float* grad_angles, void* stream);

/* The tensor glue between the decoder and the placement in one launch each way (testing/test_render_refine.py:296-306):
 *   boxes_full [n,6] = [boxes_pred[:-1] ; box_last],  idx [n] = [softargmax(angles_pred, beta)[:-1] + noise[:-1] / 10 ; angle_last[0]]
 * with softargmax(x) = sum_j softmax(beta * x)_j * (j + 1) - 1 (:20-25); noise may be NULL.  backward folds the reference's two
 * gradient hooks in: quad_grad (d idx * 4, :226-228) and fix_grad (both halves of a box row get the mean of their two
 * gradients, :217-224); the frozen last row gets zero gradients.  grad_boxes_pred [n,6], grad_angles_pred [n,n_angle]. */
int sln_refine_head_forward(int n, int n_angle, const float* boxes_pred, const float* angles_pred, const float* noise,
                            const float* box_last, const float* angle_last, float beta, float* boxes_full, float* idx, void* stream);
int sln_refine_head_backward(int n, int n_angle, const float* angles_pred, const float* grad_boxes_full, const float* grad_idx,
                             float beta, float* grad_boxes_pred, float* grad_angles_pred, void* stream);
/* params -= step * grads; grads = 0; z -= step_z * grad_z  (one launch).  The reference builds a NEW SGD(momentum=0.1,
 * nesterov=True) every iteration (:286-292), whose single step is p -= lr * 1.1 * grad: pass step = lr * 1.1.
 * params / grads 16-byte aligned. */
int sln_refine_sgd(float* params, float* grads, int64_t n, float step, float* z, const float* grad_z, int64_t nz, float step_z,
                   void* stream);

/* ---- R rooms per launch (the rooms of testing/test_render_refine.py:250-263 are independent: one iteration of all of them) ----
 * Rows of all rooms concatenated (rows_total); room_of_row [rows_total] int32; last_row [R] = the room's frozen last row.
 * head forward / backward as sln_refine_head_forward / _backward per room: box_last [R,6], angle_last [R]; the backward writes
 * grad_boxes_pred with row stride ld_gb (8 = the engine's padded gradient layout, the padding columns are zeroed). */
int sln_refine_head_forward_rooms(int rows_total, int n_angle, const int32_t* room_of_row, const int32_t* last_row, const float* boxes_pred,
                                  const float* angles_pred, const float* noise, const float* box_last, const float* angle_last, float beta,
                                  float* boxes_full, float* idx, void* stream);
int sln_refine_head_backward_rooms(int rows_total, int n_angle, const int32_t* room_of_row, const int32_t* last_row, const float* angles_pred,
                                   const float* grad_boxes_full, const float* grad_idx, float beta, float* grad_boxes_pred, int ld_gb,
                                   float* grad_angles_pred, void* stream);
/* sln_place_forward / _backward of R rooms in one launch each: `rooms` is a DEVICE array of R blocks (the room's descriptor and its
 * tensors); F_max / n_max the largest face / row count (grid sizing). */
typedef struct {
  SlnPlacement P;
  const float* boxes; const float* angles; const float* size_target;      /* [n,6], [n], [n_vis,3] or NULL */
  float* faces_out; float* sizes; float* size_loss;                        /* [2F,3,3], [n_vis,3], [1] */
  const float* grad_faces; const float* grad_size_loss; float* grad_boxes; float* grad_angles;   /* backward */
  /* optional head (boxes_pred != NULL; needs P.n <= 128): the two placement launches then do sln_refine_head_forward / _backward of
   * the room themselves - forward derives `boxes` / `angles` from the decoder's outputs (and stores them there), backward turns
   * grad_boxes / grad_angles into the gradients of the decoder's outputs - with the stand-alone kernels' arithmetic */
  const float* boxes_pred; const float* angles_pred; const float* noise;     /* [n,6], [n,n_angle] log-softmax, [n] or NULL */
  int* noise_step; int64_t noise_stride;                                      /* optional device counter k: forward reads noise + k * noise_stride,
                                                                               * backward (room 0 of the launch) advances it by one */
  const float* box_last; const float* angle_last;                             /* [6], [1]: the frozen last row */
  float* grad_boxes_pred; float* grad_angles_pred;                            /* [n,ld_gb], [n,n_angle] */
  int n_angle, ld_gb; float beta; int pad_;
} SlnPlacementRoom;
int sln_place_forward_rooms(const SlnPlacementRoom* rooms /* device */, int R, int F_max, void* stream);
int sln_place_backward_rooms(const SlnPlacementRoom* rooms /* device */, int R, int n_max, void* stream);
/* sln_refine_sgd over R parameter copies: for every room r and every range k: p[r * stride + off_k + i] -= step * g[...], g = 0
 * (i < len_k; off_k, len_k multiples of 4 floats; n_ranges <= 96: the decoder's parameters are the three *_dc embedding tables in
 * front and the trailing gconv_net_dc / box_net / angle_net run - encoder-only parameters have zero gradients in this loop and are
 * not touched - or, with SlnVaeGroupIO::sgd_step, what is left of those runs between the Linear tensors the wgrads step
 * themselves); z [nz] -= step_z * grad_z. */
int sln_refine_sgd_rooms(float* params, float* grads, int R, int64_t stride, const int64_t* off_host, const int64_t* len_host, int n_ranges,
                         float step, float* z, const float* grad_z, int64_t nz, float step_z, void* stream);

/* Refinement loss of the layout-refinement loop (testing/test_render_refine.py:192-215 PSP_pool_new, :332-356):
 * null-fill of the last depth channel, bilinear(align_corners=True) resampling of the 40 semantic and 29 depth channels of
 * the [B,70,S,S] scene tensor to each scale and bilinear resampling to pooled_size, L1 against the pooled target depth * 0.5,
 * sum over the scales of cross-entropy against the target's labels / 800;  loss_out = {100 * depth + 100 * sem, depth, sem}
 * (the caller adds 2 * size_loss; per_room: one such triple per image).  All table pointers are DEVICE arrays the caller builds once per geometry:
 *   stage 2 (scale -> pooled, align_corners=False), per scale and pooled index: source rows k0, k1 and the weight of k1;
 *   stage 1 (image -> scale, align_corners=True), per scale and scale index (row stride stage1_stride): i0, i1, weight of i1;
 *   transposed composite operator for backward, CSR per scale over the image index: col_ptr [n_scales][S+1] (global offsets),
 *   col_out (pooled index), col_w.
 * forward leaves d loss / d pooled in the workspace, backward gathers it into grad_image [B,70,S,S] (overwritten) times
 * grad_scale[0] (device scalar: the incoming gradient of loss_out[0]). */
typedef struct {
  int B, image_size, pooled_size, channels;     /* 70 channels: diff_render.py:400-434 */
  int sem0, n_sem, dep0, n_dep;                 /* 1, 40, 41, 29 */
  int n_scales, stage1_stride;                  /* <= 4 scales (32, 48, 64, 96) */
  const int32_t* s2_k0; const int32_t* s2_k1; const float* s2_l1;     /* [n_scales][pooled_size] */
  const int32_t* s1_i0; const int32_t* s1_i1; const float* s1_l1;     /* [n_scales][stage1_stride] */
  const int32_t* col_ptr; const int32_t* col_out; const float* col_w;
  int max_col_entries;                          /* longest CSR row (selects the register-list backward kernel when <= 5); 0 = unknown */
  int per_room;                                 /* != 0: the B images are B independent rooms (one refinement loop each): loss_out is
                                                 * [B][3], every room's L1 mean runs over its own elements, inv_count is [B][n_scales] */
  const unsigned char* live_planes;             /* optional [B, channels] (device), as written by sln_scene_live_channels for `image`:
                                                 * bit 0 clear - the plane is all zeros: forward does not read it (a semantic plane
                                                 * enters the cross-entropy as zeros, unpooled; a depth-hot plane gets a zero pooled
                                                 * plane); value 1 - the plane is the constant 1: pooled from 1.f without loads (or not at
                                                 * all, see pooled_ones); bit 1 clear - nobody reads the plane's gradient: backward
                                                 * leaves that plane of grad_image as it is.  NULL: every plane is processed.  Honoured
                                                 * for the shapes of the refinement loop (P <= 96, 40 semantic channels), ignored else. */
  const unsigned char* null_mask;               /* optional [B, S, S] (device), with live_planes: 1 where the 29 depth-hot values of the pixel sum
                                                 * to < 0.5 (test_render_refine.py:332), as sln_scene_forward_live writes it - the loss then
                                                 * does not compute it */
  const float* pooled_ones;                     /* optional [n_scales, P, P] (device): sln_refine_pool of an all-ones plane.  With live_planes,
                                                 * the planes flagged "constant 1" are then not pooled per image: the loss reads this table */
} SlnRefineLoss;
int64_t sln_refine_loss_workspace_bytes(int B, int image_size, int pooled_size, int n_scales, int n_sem, int n_dep);
/* 1 when live_planes / null_mask / pooled_ones of L would be honoured (the refinement loop's shapes), 0 when every plane is processed
 * whatever they say: a producer that leaves dead planes unwritten (sln_scene_forward_live) must only be paired with a loss that
 * does not read them */
int sln_refine_loss_live_ok(const SlnRefineLoss* L);
int sln_refine_loss_init(const SlnRefineLoss* L /* host struct */, void* workspace, void* stream);   /* validates L; once per workspace */
/* The resampling alone: pooled_out [B][n_scales][n_sem + n_dep][P][P] of `image` (null_fill != 0: with the null-fill of the
 * last depth channel).  The caller derives the target's pooled depth and labels from it, so that regions where the iterate
 * equals the target give a difference of exactly 0 (and sign(0) = 0 in the L1 gradient), as in the reference where both go
 * through the same resampling code. */
int sln_refine_pool(const SlnRefineLoss* L, const float* image, int null_fill, void* workspace, float* pooled_out, void* stream);
/* target_depth_pooled [B][n_scales*n_dep][P][P], labels [B][n_scales][P][P] int32 (-100 = ignored), inv_count [n_scales]
 * (1 / number of non-ignored labels of the scale over the batch) */
int sln_refine_loss_forward(const SlnRefineLoss* L, const float* image, const float* target_depth_pooled, const int32_t* labels,
                            const float* inv_count, void* workspace, float* loss_out, void* stream);
int sln_refine_loss_backward(const SlnRefineLoss* L, const void* workspace, const float* grad_scale, float* grad_image, void* stream);

/* =============================================================================================
 * C. SPADE generator (models/SPADE_related.py: SPADEGenerator4 :1507-1605, SPADEResnetBlock4 :1457-1505,
 *    SPADE4 :1404-1454, LayerNorm2D :128-149, SEBlock2 :70-85; driver testing/test_SPADE_shade.py:9-13,77-79)
 * Tensors are NCHW fp32.  Conv weights are PRE-PACKED by the host once per checkpoint:
 *   wp[ks*ks][Cin][rows_pad] (output rows contiguous, rows_pad % 64 == 0), spectral norm already folded
 *   (W = W_orig / (u . (W_mat v))); for the modulation conv the rows of gamma and beta are interleaved in
 *   groups of 32: rows [64 g, 64 g + 32) = gamma of channels [32 g, 32 g + 32), the next 32 rows their beta.
 * ============================================================================================= */
/* Small convolution launches (< 192 workgroups: the batch-1 calls of testing/test_SPADE_shade.py:77-79) split their input channels
 * over several workgroups and add the partial sums in a fixed order; the partial sums live in a 48 MB scratch slot per (device,
 * stream) - at most 16 slots, the least recently used one is freed for a new stream.
 * sln_spade_prepare allocates `stream`'s slot now (optional: the first eager split launch on a stream does the same).  A slot cannot
 * be allocated while the stream is being captured: a split launch on a stream first seen during its capture fails with SLN_E_STATE
 * (-3) - call sln_spade_prepare (or run one eager forward) on the stream before capturing it.  An allocation failure is SLN_E_NOMEM.
 * sln_spade_release frees the slot of `stream` on the current device (all != 0: every slot of the process); it waits for the device.
 * Returns the number of slots freed.  Call it before destroying a stream that ran SPADE launches (a recycled handle would otherwise
 * inherit the slot, which is harmless, or keep 48 MB alive). */
int sln_spade_prepare(void* stream);
int sln_spade_release(void* stream, int all);
/* y = act(conv_ks(x) + bias): ks = 3 (ReflectionPad2d(1)) or 1; act 0 none, 1 ReLU, 2 LeakyReLU(slope) */
int sln_spade_conv(const float* x, int B, int Cin, int H, int W, const float* wp, const float* bias, int rows, int rows_pad,
                   int ksize, int act, float slope, float* y, void* stream);
/* SPADE4 tail fused: out = LayerNorm2D(xin) * (1 + gamma) + beta [LeakyReLU(slope) when act == 2] with
 * [gamma | beta] = conv3x3_reflect(actv); stats[b] = (mean, 1 / (std + eps)) from sln_layernorm_stats */
int sln_spade_modulate(const float* actv, int B, int Cin, int H, int W, const float* wp, const float* bias, int C, int rows_pad,
                       const float* xin, const float* stats, int act, float slope, float* out, void* stream);
/* sln_spade_conv that also accumulates, from the values its epilogue writes, the sums the next layers need (fp64 atomics into
 * buffers the caller zeroed; either may be NULL): ln_acc [B][16] doubles = (sum y, sum y^2) of sample b -> LayerNorm2D
 * statistics of the following SPADE layer (:128-149) through sln_layernorm_finalize; gap_acc [B, rows] doubles = sum over
 * pixels -> SEBlock2's average pool (:70-85) through sln_block_tail. */
int sln_spade_conv_sums(const float* x, int B, int Cin, int H, int W, const float* wp, const float* bias, int rows, int rows_pad,
                        int ksize, int act, float slope, float* y, double* ln_acc, double* gap_acc, void* stream);
/* sln_spade_modulate with xin_up = 1: xin is [B, C, H/2, W/2] and stands for nn.Upsample(scale_factor=2) (nearest) of itself
 * (SPADEGenerator4.forward :1585-1597) - the upsampled tensor is never written; stats must be those of the upsampled tensor. */
int sln_spade_modulate_up(const float* actv, int B, int Cin, int H, int W, const float* wp, const float* bias, int C, int rows_pad,
                          const float* xin, int xin_up, const float* stats, int act, float slope, float* out, void* stream);
/* stats[b] = (mean, 1 / (unbiased std + eps)) from accumulated sums: acc [B][16] doubles (sum, sum of squares of n_acc values),
 * every value standing for `rep` elements of the normalised tensor (4 = its nearest x2 upsampling) */
int sln_layernorm_finalize(const double* acc, int B, int64_t n_acc, int rep, float eps, float* stats, void* stream);
/* Tail of SPADEResnetBlock4.forward (:1492-1493) and the nn.Upsample after it (:1585-1600) in one pass:
 *   out = up(xs + dx * sigmoid(W2 relu(W0 GAP(dx)))),  stats = LayerNorm2D statistics of the next block's input.
 * xs [B,C,H,W] (xs_up = 1: [B,C,H/2,W/2] read through nearest x2); gap_sums = the gap_acc of sln_spade_conv_sums or NULL;
 * up_mode -1: out [B,C,H,W], 0 nearest / 1 bilinear: out [B,C,2H,2W]; stats_rep 4 when the consumers read `out` through
 * nearest x2 themselves; stats may be NULL; scratch 2*B*C floats; acc 16*B doubles. */
int sln_block_tail(const float* xs, int xs_up, const float* dx, int B, int C, int H, int W, const double* gap_sums, const float* w0,
                   const float* w2, float* scratch, int up_mode, float* out, double* acc, int stats_rep, float eps, float* stats,
                   void* stream);
/* The same modulation when ONE semantic map drives the whole batch (colorize_with_spade, testing/test_SPADE_shade.py:30-79:
 * 50 z vectors per room): gb [rows_pad, H, W] = sln_spade_conv(actv of that map, the packed gamma|beta weights, act 0) is
 * computed once, then out[b] = LayerNorm2D(xin[b]) * (1 + gamma) + beta for every sample.  H*W % 4 == 0. */
int sln_spade_apply(const float* xin, const float* gb, int B, int C, int H, int W, int rows_pad, const float* stats, int act,
                    float slope, float* out, void* stream);
/* sln_spade_apply with xin_up = 1: xin is [B, C, H/2, W/2] and stands for its nearest x2 upsampling (W % 4 == 0, H % 2 == 0) */
int sln_spade_apply_up(const float* xin, int xin_up, const float* gb, int B, int C, int H, int W, int rows_pad, const float* stats,
                       int act, float slope, float* out, void* stream);
/* stats[b] = (mean, 1 / (unbiased std + eps)) over the n elements of sample b; scratch: 16 * B doubles */
int sln_layernorm_stats(const float* x, int B, int64_t n, float eps, double* scratch, float* stats, void* stream);
/* F.interpolate(size=...): mode 0 nearest, 1 bilinear(align_corners=False) over BC planes */
int sln_resize(const float* src, int BC, int Hi, int Wi, int Ho, int Wo, int mode, float* dst, void* stream);
/* [LeakyReLU_0.01(conv3x3_reflect(seg[:,0:1])) | seg[:,1:]] (SPADE4.mlp_preshared_depth + cat, :1445-1446) into
 * out [B, nd + Cs - 1, H, W].  copy_masks = 0 writes the nd depth features only: the mask channels of a per-resolution
 * buffer are filled once (copy_masks = 1) and shared by every SPADE layer of that resolution. */
int sln_spade_depth_concat(const float* seg, int B, int Cs, int H, int W, const float* wpd, const float* bpd, int nd, float* out,
                           int copy_masks, void* stream);
/* x_s + SEBlock2(dx) (:1492-1493); scratch 2*B*C floats */
int sln_se_scale_add(const float* xs, const float* dx, int B, int C, int64_t hw, const float* w0, const float* w2, float* scratch,
                     float* out, void* stream);
/* nn.Upsample(scale_factor=2): mode 0 nearest, 1 bilinear */
int sln_upsample2x(const float* x, int BC, int H, int W, int mode, float* y, void* stream);
/* tanh(conv5x5_zero_pad(LeakyReLU_0.2(x))) (:1602-1603); w [Cout,Cin,5,5], Cout <= 4 */
int sln_conv_img_tanh(const float* x, int B, int Cin, int H, int W, const float* w, const float* bias, int Cout, float* y, void* stream);

/* =============================================================================================
 * Scene-graph builder: SuncgDataset.__getitem__ + suncg_collate_fn for a batch of rooms
 * (reference data/suncg_dataset.py:110-353; compute_rel utils.py:36-80).  The reference builds one room at a
 * time in python; here the room table is resident in HBM and a batch of B rooms is two launches + one emit.
 * All pointers are device pointers unless noted.
 * ============================================================================================= */
typedef struct {
  const int* room_off;            /* [n_rooms+1] first object of each room in the flat arrays */
  const int* cls;                 /* [N] vocabulary index of every object (>= 1; 0 is '__room__') */
  const float* bbox;              /* [N,6] raw x0,y0,z0,x1,y1,z1 ('new_bbox', suncg_dataset.py:116-123) */
  const int* rot;                 /* [N] 'rotation' bin */
  const float* room_bbox;         /* [n_rooms,3] (suncg_dataset.py:131-137) */
  const int64_t* room_id;         /* [n_rooms] ids returned as all_ids, or NULL: the table index */
  const float* size_thr;          /* [n_classes,4]: use_attr_30 = 0: (height, volume, -, -) of size_info_many.json;
                                     = 1: (height_7, height_3, volume_7, volume_3) of 30_size_info_many.json */
  const unsigned char* has_size;  /* [n_classes] class present in that json */
  int n_rooms, n_classes, use_attr_30, reserved;
} SlnRoomTable;

/* The random decisions of __getitem__, one entry per NON-room object of the batch, batch order
 * (suncg_dataset.py:189-196: random.choice / random.random; :236-282: attribute draws). */
typedef struct {
  const int* other;               /* partner, as an object index inside its room (!= the object itself) */
  const unsigned char* swap;      /* 1: (s, o) = (cur, other)  [random.random() > 0.5],  0: (other, cur) */
  const unsigned char* attr_mode; /* 0: 'none' [first draw > 0.5 or class without statistics], 1: height test, 2: volume test */
} SlnGraphDraws;

/* Outputs with the dtypes of suncg_collate_fn (suncg_dataset.py:310-353). */
typedef struct {
  int64_t* ids;                   /* [B] or NULL */
  int64_t* objs;                  /* [O] */
  float* boxes;                   /* [O,6] room-normalised, the room row last in every graph */
  int64_t* triples;               /* [T,3] (s, p, o) with batch-global rows */
  int64_t* angles;                /* [O] */
  int64_t* attributes;            /* [O] */
  int64_t* obj_to_img;            /* [O] */
  int64_t* triple_to_img;         /* [T] */
} SlnGraphBatch;

/* counts [2B] (rows, triples per room; rows < 0 marks a room index outside the table); offsets [2B+3]:
 * [0..B] exclusive scan of rows (offsets[B] = O), [B+1..2B+1] of triples (offsets[2B+1] = T), [2B+2] = bad room indices.
 * room_idx [B] indexes the table.  The caller reads O and T back (3 ints) to size the outputs. */
int sln_graph_plan(const SlnRoomTable* tab /* host struct */, const int* room_idx, int B, int* counts, int* offsets, void* stream);
int sln_graph_emit(const SlnRoomTable* tab /* host struct */, const int* room_idx, int B, const int* offsets,
                   const SlnGraphDraws* draws /* host struct */, const SlnGraphBatch* out /* host struct */, void* stream);
/* The decisions of SlnGraphDraws drawn on the device (the reference draws them with python's `random`, suncg_dataset.py:189-196,
 * 236-282): Philox keyed by key[0..1] (two 64-bit words in DEVICE memory - e.g. taken from a torch generator, which keeps its own
 * stream position), one counter per non-room object of the batch (offsets as written by sln_graph_plan / planned on the host). */
int sln_graph_draw(const SlnRoomTable* tab /* host struct */, const int* room_idx, int B, const int* offsets, const int64_t* key,
                   int* other, unsigned char* swap, unsigned char* attr_mode, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SLN_HIP_H */
