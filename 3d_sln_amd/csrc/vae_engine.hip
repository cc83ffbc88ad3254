// Scene-graph VAE engine: owns the kernel plan for Sg2ScVAEModel (reference
// models/Sg2ScVAE_model.py:115-188, models/graph.py:57-143, utils.py:12-33, train.py:62-84) and
// exposes it through the C ABI of include/sln_hip.h.  One engine per process/GPU; every call
// enqueues kernels on the caller's stream.  A whole training iteration (zero_grad, forward, loss,
// backward, Adam) is 145 launches at train.py's defaults (one per forward Linear / dgrad, 30 edge
// launches, FOUR wgrad launches - every wgrad of a pass runs in one multi-problem grid, see
// flush_deferred -, a dozen bookkeeping launches), captured once into a hipGraph and replayed.
#include <cstddef>
#include <new>
#include <type_traits>
#include <utility>
#include <vector>
#include <cstring>
#include <cstdlib>

#include "../../include/sln_hip.h"
#include "sln_gemm.h"
#include "vae_kernels.h"
#include "vae_multi.h"
#include "sln_prof.h"

namespace {

// ---- recording (round 5: R rooms in flight, see SlnVaeGroup at the end of this file) ----------------------------------------
// With `rec` set, the decoder's forward / backward pass of an engine does not launch anything: every launch site appends its
// fully resolved argument block to the list.  The group walks the lists of its R engines in lock step and turns step s of all
// rooms into ONE multi-room launch (vae_multi.h).
enum { SK_DEC_ASSEMBLE = 0, SK_EMBED_GATHER, SK_NT, SK_SCATTER_FWD, SK_TN, SK_TN_FLUSH, SK_MASK_GSTATS, SK_ADD2, SK_EMBED_BWD,
       SK_SCATTER_BWD, SK_GATHER_BWD, SK_DEC_ASSEMBLE_BWD, SK_COPY2D };
struct RecStep {
  int kind = -1, epi = 0, idx64 = 0;
  GemmNTArgs nt; GemmTNArgs tn;
  MScatterFwd sf; MScatterBwd sb; MGatherBwd gb; MMaskGstats mg; MDecAssemble da; DecAssembleBwd dab; MEmbedGather eg; MEmbedBwd eb; MAdd2 a2;
};
struct Recorder { std::vector<RecStep> steps; };

constexpr float kBnEps = 1e-5f;
constexpr float kBnMomentum = 0.1f;

inline int rup(int x, int a) { return (x + a - 1) / a * a; }

struct Unit {
  SlnVaeUnit p;
  int out = 0, in = 0;
  bool bn = false;
  float* wt = nullptr;   // [in][wt_ld] transposed copy for dgrad
  int wt_ld = 0;
};

struct BnInst {
  int unit = -1;
  int C = 0, rows = 0;
  int rows_code = 0;        // -1: runs over the triples, -2: over the objects (what the device table holds: it does not change with the batch)
  double* sums = nullptr;   // [2][C]
  double* gsums = nullptr;  // [2][C]
};

struct Layer {              // one GraphTripleConv application
  float *A1 = nullptr, *A2 = nullptr, *M = nullptr, *A3 = nullptr, *A4 = nullptr;
  // relu-masked gradients w.r.t. the four Linear outputs.  One set PER LAYER since round 3: the wgrads that read them run at the
  // end of the backward pass (flush_deferred), not next to the dgrad that produced them.
  float *g1 = nullptr, *g2 = nullptr, *g3 = nullptr, *g4 = nullptr;
  int bn[4] = {-1, -1, -1, -1};
  int u0 = 0;               // unit index of net1.0
  int D = 0;                // input width of the object / predicate vectors
  int Do = 0;               // output width (models/graph.py:36-56: output_dim; = D everywhere except a bare GraphTripleConv)
  bool first = false, last = false;
  int net = 0;              // 0 encoder, 1 decoder
};

struct Bump {
  char* base; size_t off = 0; bool dry;
  explicit Bump(void* b) : base(static_cast<char*>(b)), dry(b == nullptr) {}
  template <typename T> T* take(size_t n) {
    off = (off + 255) & ~size_t(255);
    T* p = dry ? nullptr : reinterpret_cast<T*>(base + off);
    off += n * sizeof(T);
    return p;
  }
};

}  // namespace

struct SlnVae {
  SlnVaeConfig cfg;
  int E, H, L, nmod, n_obj_e, n_attr_e, n_box_e, n_angle_e, Dec, Ddc;
  std::vector<Unit> units;
  SlnVaeTensors t;
  bool bound = false;
  int maxO = 0, maxT = 0, O = 0, T = 0;
  SlnVaeBatch batch;
  bool batch_set = false;

  std::vector<Layer> layers;      // 2L
  std::vector<BnInst> bns;
  int bn_head[6] = {-1, -1, -1, -1, -1, -1};   // bmv0,bmv1,amv0,amv1, boxnet0, anglenet0
  int n_bn_enc = 0;

  // workspace
  GraphCsr g;
  int* attrs32 = nullptr; int* err_flag = nullptr;
  double* stats_base = nullptr; size_t stats_doubles = 0, enc_stats_doubles = 0;   // [enc sums | dec sums]
  double* gstats_base = nullptr;                                                   // [enc gsums | dec gsums]
  char* zero_begin = nullptr; size_t zero_bytes = 0; bool bulk_zeroed = false;     // see carve() / train_iteration()
  bool step_training = true;     // BatchNorm mode of the fused training iteration (train.py:63-65: model.eval() keeps training)
  int64_t* st_objs = nullptr; int64_t* st_angles = nullptr; int64_t* st_attrs = nullptr; float* st_boxes = nullptr;   // staged batch inputs
  BnTableEntry* bn_table_dev = nullptr;
  TransposeEntry* tr_table_dev = nullptr; int n_tr = 0, tr_max_tiles = 0;
  AdamScalars* scalars = nullptr; double* loss_acc = nullptr; float* losses = nullptr;
  float *X0e = nullptr, *P0e = nullptr, *X0d = nullptr, *P0d = nullptr;
  float *hbA1 = nullptr, *hbA2 = nullptr, *haA1 = nullptr, *haA2 = nullptr;
  float *mu = nullptr, *logvar = nullptr, *z = nullptr, *eps_buf = nullptr;
  float *bnA1 = nullptr, *anA1 = nullptr, *boxes_pred = nullptr, *logits = nullptr, *angles_pred = nullptr;
  // backward temporaries
  float *dbp = nullptr, *dlogits = nullptr, *g_bn = nullptr, *g_an = nullptr, *d_bx = nullptr, *d_ax = nullptr;
  float *dM = nullptr, *dG[2] = {nullptr, nullptr};
  float *dX0 = nullptr, *dz = nullptr, *dmu = nullptr, *dlv = nullptr;
  float *g_h2 = nullptr, *g_h1 = nullptr, *d_xb = nullptr, *d_xa = nullptr, *tmp_d = nullptr;
  float *g_h2b = nullptr, *g_h1b = nullptr, *tmp_db = nullptr;     // the angle branch's copies (both branches run grouped)
  int dbp_ld = 8;

  bool enc_training = false, dec_training = false, have_enc = false, have_dec = false, have_loss_grads = false;
  const float* z_src = nullptr;   // z given to decoder() (external) or nullptr when computed from mu/logvar/eps
  bool z_from_latent = false;
  bool wt_fresh = false;          // transposed weights valid for the current parameters
  float host_kl = 0.f, host_lr = 0.f; int64_t host_step = 0; bool host_scalars_valid = false;
  float* grad_guard = nullptr;    // sln_vae_set_grad_guard: where the iteration leaves its total loss for the collective NaN guard
  bool draw_eps = false;          // the iteration draws its own N(0,1) (no eps from the caller)
  unsigned long long host_seed[2] = {0, 0};

  // wgrad GEMMs run on a side stream, concurrent with the dgrad chain (both only half-fill the chip at
  // batch 64); fork after the producer of the wgrad's gradient operand, join before that buffer is reused.
  hipStream_t side = nullptr;
  std::vector<hipEvent_t> events; size_t ev_next = 0; bool use_side = true; bool side_busy = false;
  hipEvent_t next_event() {
    if (ev_next == events.size()) { hipEvent_t e = nullptr; (void)hipEventCreateWithFlags(&e, hipEventDisableTiming); events.push_back(e); }
    return events[ev_next++];
  }
  int fork_side(hipStream_t st) {
    hipEvent_t e = next_event();
    hipError_t r = hipEventRecord(e, st);
    if (r == hipSuccess) r = hipStreamWaitEvent(side, e, 0);
    side_busy = true;
    return (int)r;
  }
  // Pairing: a wgrad is held back and shares the launch of the next dgrad (sln_launch_gemm_dual); this replaced the
  // side stream as the default (SLN_NO_DUAL=1 goes back to it) - one dispatch instead of two, and no event edges.
  bool use_dual = true;
  std::vector<GemmTNArgs> pending;
  int flush_pending(hipStream_t st) {
    for (const GemmTNArgs& t : pending) { int r = sln_launch_gemm_tn(t, -1, st); if (r) { pending.clear(); return r; } }
    pending.clear();
    return 0;
  }
  // Round 3: deferral.  A wgrad has no consumer before the optimizer, so nothing forces it to run next to the dgrad of the same
  // Linear.  The problems of a backward pass are recorded and run as ONE launch per pass (two when some gather their X rows)
  // through sln_launch_gemm_tn_multi: chunks of ~1 k rows instead of 256 (the prologue, the first-tile latency and the 64 x 64
  // atomics of a block are paid a quarter as often), no launch boundary per wgrad, and the dgrad chain - now alone in its
  // launches - gets the chip to itself.  SLN_NO_DEFER=1 restores the pairing.
  // The problem tables live in device memory.  Set 0 belongs to the captured iterations (filled after the capture ends, before
  // the graph is launched: a capture records, nothing runs), set 1 to eager calls; [which] 0 = decoder pass, 1 = encoder pass;
  // [k] 0 = plain rows, 1 = gathered X rows.
  struct TnGroup {
    GemmTNArgs probs[SLN_TN_MULTI_MAX]; TnMultiMeta meta; int n = 0, blocks = 0; bool x2 = false, xg = false; double flops = 0.0;
    bool dirty = true; GemmTNArgs* dev_probs = nullptr; TnMultiMeta* dev_meta = nullptr;
  };
  // table slots of the per-pass wgrad launches: three regions of tn_slots / 3 - decoder pass, encoder pass, and (round 6) the ONE flush of
  // a full iteration, whose tables must not share slots with the two-half graphs of the same engine.  Sized from the layer count
  // at creation (per-layer flushing - SLN_TN_PER_LAYER, deterministic mode with shared 'recurrent' weights - takes up to two per
  // layer; a launch that finds none runs its problems one by one)
  int tn_slots = 32;
  std::vector<GemmTNArgs> tn_kind_scratch;
  bool defer = true, capturing = false, tn_upload_pending = false;
  int det_seen = 0;                              // g_sln_deterministic the captured iterations were recorded with
  bool tn_per_layer = false;                     // flush after every GraphTripleConv instead of once per pass
  bool tn_side = false, tn_side_busy = false;    // run the wgrad launches on the side stream, next to the dgrad chain
  int tn_slot_next[3] = {0, 0, 0};
  std::vector<GemmTNArgs> deferred;
  TnGroup* tn_groups_store = nullptr;            // [2 sets][TN_SLOTS][2], heap (a TnGroup is 50 KB)
  TnGroup& tn_group(int set, int slot, int k) { return tn_groups_store[((size_t)set * tn_slots + slot) * 2 + k]; }
  int upload_group(TnGroup& g) {
    hipError_t e = hipMemcpy(g.dev_probs, g.probs, sizeof(GemmTNArgs) * (size_t)g.n, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(g.dev_meta, &g.meta, sizeof(TnMultiMeta), hipMemcpyHostToDevice);
    g.dirty = e != hipSuccess;
    return (int)e;
  }
  // Eager calls whose problem set changed - a batch of another size, i.e. EVERY step of train.py on real rooms - upload their tables
  // in STREAM ORDER from a ring of pinned staging buffers: the copy runs behind the previous step's launches (which read the old
  // table) and in front of this step's, the host never waits (the blocking upload behind a hipStreamSynchronize cost 0.35 ms per
  // step with varying shapes, tools/varshape_time.py).  A slot is reused TN_STAGE_SLOTS uploads later, after its event.
  bool bn_table_live = false;      // the BatchNorm table of this binding is on the device (row counts are codes: one upload per binding)
  struct TnStage { char* host = nullptr; hipEvent_t done = nullptr; };
  enum { TN_STAGE_SLOTS = 16 };
  TnStage tn_stage[TN_STAGE_SLOTS];
  int tn_stage_next = 0;
  // stream-ordered host -> device copy of up to two pieces through the next ring slot (slot capacity: a wgrad table)
  int stage_upload(void* dev0, const void* src0, size_t n0, void* dev1, const void* src1, size_t n1, hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return SLN_E_CAPTURE;   // a caller's capture would record a copy out of a ring slot
    const size_t cap0 = sizeof(GemmTNArgs) * (size_t)SLN_TN_MULTI_MAX, cap = cap0 + sizeof(TnMultiMeta);
    if (n0 > cap0 || n1 > cap - cap0) return SLN_E_BADARG;
    TnStage& sl = tn_stage[tn_stage_next];
    tn_stage_next = (tn_stage_next + 1) % TN_STAGE_SLOTS;
    hipError_t e = hipSuccess;
    if (!sl.host) {
      void* hp = nullptr;
      if (hipHostMalloc(&hp, cap, hipHostMallocDefault) != hipSuccess) return SLN_E_NOMEM;
      sl.host = static_cast<char*>(hp);
      if (hipEventCreateWithFlags(&sl.done, hipEventDisableTiming) != hipSuccess) { (void)hipHostFree(hp); sl.host = nullptr; return SLN_E_NOMEM; }
    } else {
      e = hipEventSynchronize(sl.done);
    }
    if (e == hipSuccess && n0) { std::memcpy(sl.host, src0, n0); e = hipMemcpyAsync(dev0, sl.host, n0, hipMemcpyHostToDevice, st); }
    if (e == hipSuccess && n1) { std::memcpy(sl.host + cap0, src1, n1); e = hipMemcpyAsync(dev1, sl.host + cap0, n1, hipMemcpyHostToDevice, st); }
    if (e == hipSuccess) e = hipEventRecord(sl.done, st);
    return (int)e;
  }
  int upload_group_async(TnGroup& g, hipStream_t st) {
    const int r = stage_upload(g.dev_probs, g.probs, sizeof(GemmTNArgs) * (size_t)g.n, g.dev_meta, &g.meta,
                               offsetof(TnMultiMeta, item) + sizeof(TnMultiItem) * (size_t)g.meta.nblocks, st);
    if (r == 0) g.dirty = false;
    return r;
  }
  int upload_pending_tables() {                   // after a capture: everything the captured launches will read
    for (int w = 0; w < tn_slots; ++w)
      for (int k = 0; k < 2; ++k) {
        TnGroup& g = tn_group(0, w, k);
        if (g.n > 0 && g.dirty) { int r = upload_group(g); if (r) return r; }
      }
    tn_upload_pending = false;
    return 0;
  }
  // the wgrad launches may run on the side stream: fork behind the producer of their operands, join before anything reads the
  // parameter gradients (end of an iteration / of an eager backward call)
  hipStream_t tn_side_stream = nullptr;          // the stream the current iteration's wgrad launches went to (join_tn_side)
  int join_tn_side(hipStream_t st) {
    if (!tn_side_busy) return 0;
    hipEvent_t e = next_event();
    hipError_t r = hipEventRecord(e, tn_side_stream ? tn_side_stream : side);
    if (r == hipSuccess) r = hipStreamWaitEvent(st, e, 0);
    tn_side_busy = false;
    return (int)r;
  }
  hipStream_t tn_eager_stream = nullptr;         // the stream whose launches read the eager table set (set 1) last
  int flush_deferred(int which, hipStream_t st) {
    if (rec) { rec_push(SK_TN_FLUSH); return 0; }
    if (deferred.empty()) return 0;
    const int set = capturing ? 0 : 1;
    if (set == 1 && tn_eager_stream != st) {
      // eager steps alternating between two caller streams on ONE engine: an upload on the new stream must not overtake the
      // launches of the old one that still read the tables (uploads are ordered per stream only).  Rare: drain the old stream.
      // (not while the CALLER captures `st` - torch.cuda.graph around an eager-mode call: a synchronisation would invalidate the
      // capture, and such a call cannot upload anyway: its tables were made resident by the warm-up run the capture follows)
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone, cs_old = hipStreamCaptureStatusNone;
      const bool st_capt = hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
      if (!st_capt) {
        if (tn_eager_stream != nullptr && hipStreamIsCapturing(tn_eager_stream, &cs_old) == hipSuccess && cs_old == hipStreamCaptureStatusNone)
          (void)hipStreamSynchronize(tn_eager_stream);
        tn_eager_stream = st;
      }
    }
    hipStream_t lst = st;
    if (tn_side && side) {
      // (lab, SLN_TN_SIDE=1) eager launches: a pooled stream that is PROBED to overlap with the caller's (csrc/streams.hip: the engine's
      // own side stream may share the caller's hardware queue); captures keep the engine's stream
      hipStream_t sd = sln_capturing(st) ? side : sln_overlapping_stream(st);
      if (sd == nullptr) sd = side;
      hipEvent_t e = next_event();
      hipError_t r = hipEventRecord(e, st);
      if (r == hipSuccess) r = hipStreamWaitEvent(sd, e, 0);
      if (r != hipSuccess) { deferred.clear(); return (int)r; }
      lst = sd; tn_side_stream = sd; tn_side_busy = true;
    }
    static thread_local TnGroup tmp;
    // two kinds of problems (X rows gathered: every net1.0; plain rows: the rest).  A launch group is closed by problem count
    // (SLN_TN_MULTI_MAX) AND by planned tile count: all tiles of one (problem, row chunk) sit on one XCD, so the item table
    // (SLN_TN_MULTI_ITEMS) holds a group only while 8 x the longest XCD list fits - longer row chunks cannot shrink a problem's
    // (Nout / 64) x (Kin / 64) tiles.  A group the planner still refuses is halved; a single problem it refuses (hidden widths
    // >= 1024: more than ITEMS / 8 tiles) or one that finds no free table slot runs as a plain per-problem launch.
    std::vector<GemmTNArgs>& kind = tn_kind_scratch;
    for (int k = 0; k < 2; ++k) {
      kind.clear();
      for (const GemmTNArgs& t : deferred) {
        bool gathers = false;
        for (int s2 = 0; s2 < t.X.nseg; ++s2) gathers |= t.X.seg[s2].which != 0;
        if ((int)gathers == k) kind.push_back(t);
      }
      size_t next = 0;
      while (next < kind.size()) {
        int want = 0; long tiles = 0;
        for (size_t i = next; i < kind.size() && want < SLN_TN_MULTI_MAX; ++i) {
          const long tl = (long)sln_cdiv(kind[i].Nout, 64) * sln_cdiv(kind[i].Kin, 64);
          if (want > 0 && tiles + tl > SLN_TN_MULTI_ITEMS / 2) break;
          tiles += tl; ++want;
        }
        int r = -1;
        const bool have_slot = tn_slot_next[which] < tn_slots / 3;
        for (; have_slot && want >= 1; want /= 2) {
          tmp.n = want;
          for (int i = 0; i < want; ++i) tmp.probs[i] = kind[next + i];
          r = sln_tn_multi_plan(tmp.probs, tmp.n, &tmp.meta, &tmp.blocks, &tmp.x2, &tmp.xg, &tmp.flops);
          if (r == 0) break;
        }
        if (r != 0) {                 // one problem, launched on its own (2-D grid: no table)
          GemmTNArgs one = kind[next++];
          if (g_sln_deterministic) one.rows_per_block = sln_cdiv(one.R, 32) * 32;         // one add per dW element, as in the planned launches
          r = sln_launch_gemm_tn(one, -1, lst);
          if (r) { deferred.clear(); return r; }
          continue;
        }
        next += (size_t)tmp.n;
        const int slot = which * (tn_slots / 3) + tn_slot_next[which]++;
        TnGroup& g = tn_group(set, slot, k);
        if (g.n != tmp.n || std::memcmp(g.probs, tmp.probs, sizeof(GemmTNArgs) * (size_t)tmp.n) != 0 ||
            std::memcmp(&g.meta, &tmp.meta, sizeof(TnMultiMeta)) != 0) {
          std::memcpy(g.probs, tmp.probs, sizeof(GemmTNArgs) * (size_t)tmp.n);
          g.meta = tmp.meta; g.n = tmp.n; g.blocks = tmp.blocks; g.x2 = tmp.x2; g.xg = tmp.xg; g.flops = tmp.flops;
          g.dirty = true;
        }
        if (g.dirty) {
          if (capturing) tn_upload_pending = true;
          else if (lst == st) {       // eager call with a new problem set (new shape, other BatchNorm mode): stream-ordered upload
            r = upload_group_async(g, st);
            if (r) { deferred.clear(); return r; }
          } else {                    // wgrads on the side stream (SLN_TN_SIDE=1): blocking, as before
            hipError_t e = hipStreamSynchronize(st);          // an earlier launch may still read the old table
            if (e == hipSuccess && side) e = hipStreamSynchronize(side);
            if (e == hipSuccess && lst != side) e = hipStreamSynchronize(lst);
            if (e != hipSuccess) { deferred.clear(); return (int)e; }
            r = upload_group(g);
            if (r) { deferred.clear(); return r; }
          }
        }
        r = sln_launch_gemm_tn_multi(g.dev_probs, g.dev_meta, g.blocks, g.x2, g.xg, g.flops, lst);
        if (r) { deferred.clear(); return r; }
      }
    }
    deferred.clear();
    return 0;
  }
  int join_side(hipStream_t st) {
    { int r = flush_pending(st); if (r) return r; }
    if (!side_busy) return 0;
    hipEvent_t e = next_event();
    hipError_t r = hipEventRecord(e, side);
    if (r == hipSuccess) r = hipStreamWaitEvent(st, e, 0);
    side_busy = false;
    return (int)r;
  }

  // hipGraph of one training iteration
  // one per mode of train_iteration (TRAIN_*)
  hipGraphExec_t graph_exec[4] = {nullptr, nullptr, nullptr, nullptr}; int graph_O[4] = {-1, -1, -1, -1}, graph_T[4] = {-1, -1, -1, -1};

  // ------------------------------------------------------------------------------------------
  // Round 3: bookkeeping launches of the fused iteration merged (train_iteration sets these around its calls; the stand-alone
  // entry points - sln_vae_encoder / _decoder / _loss / *_backward - keep their own launches):
  bool it_zero_in_prologue = false;   // ... and that launch also clears the iteration's accumulators
  bool it_prologue = false;       // enc_assemble + both predicate gathers (+ the N(0,1) draw) already issued as ONE launch
  bool it_fused_loss = false;     // log_softmax is taken inside the loss kernel
  bool it_one_flush = false;      // full iterations: the decoder pass's wgrads ride with the encoder pass's launches (SLN_TN_ONE_FLUSH=0: off)
  bool it_merge_bn = false;       // ONE running-statistics launch per iteration (after the decoder) and ONE parameter-gradient launch
  bool no_merge = false;          // SLN_NO_MERGE=1 at creation: none of the three
  bool gconv_only = false;        // a bare GraphTripleConvNet (sln_gconv_net_*): units = the modules' four Linears, one net, no heads
  int unit_of(int net, int l, int k) const { return (gconv_only ? 0 : 8) + (net * nmod + (cfg.recurrent ? 0 : l)) * 4 + k; }
  int unit_boxnet(int k) const { return 8 + 2 * nmod * 4 + k; }
  int unit_anglenet(int k) const { return 8 + 2 * nmod * 4 + 2 + k; }

  int bn_mode(const BnInst& b, bool training) const {
    if (b.unit < 0 || !units[b.unit].bn) return SLN_BN_NONE;
    return training ? SLN_BN_TRAIN : SLN_BN_EVAL;
  }
  BnView view(int inst, int col0, bool training) const {
    BnView v; std::memset(&v, 0, sizeof(v));
    v.eps = kBnEps; v.n_rows = 1.f; v.rn = 1.0; v.mode = SLN_BN_NONE;
    if (inst < 0) return v;
    const BnInst& b = bns[inst];
    const Unit& u = units[b.unit];
    v.mode = bn_mode(b, training);
    if (v.mode == SLN_BN_NONE) return v;
    v.sums = b.sums + col0; v.gsums = b.gsums + col0; v.cstride = b.C;
    v.gamma = u.p.bn_weight + col0; v.beta = u.p.bn_bias + col0;
    v.rmean = u.p.bn_running_mean + col0; v.rvar = u.p.bn_running_var + col0;
    v.n_rows = (float)(b.rows > 0 ? b.rows : 1); v.rn = 1.0 / (double)(b.rows > 0 ? b.rows : 1);
    return v;
  }
  static Seg seg_ident(const float* x, int ld, int col0, int len, int which) {
    Seg s; std::memset(&s, 0, sizeof(s));
    s.x1 = x; s.ld1 = ld; s.c1 = col0; s.len = len; s.which = which; s.coef = SLN_COEF_IDENT;
    return s;
  }
  Seg seg_act(const float* x, int ld, int col0, int len, int inst, int which, bool training) const {
    // relu(bn(x)) of a stored pre-activation
    Seg s = seg_ident(x, ld, col0, len, which);
    s.coef = SLN_COEF_FWD; s.bn = view(inst, col0, training);
    return s;
  }
  Seg seg_bwd(const float* gmask, int ldg, const float* x, int ldx, int len, int inst, bool training) const {
    // gradient w.r.t. the Linear output, rebuilt from the relu-masked gradient and the pre-activation
    Seg s = seg_ident(gmask, ldg, 0, len, 0);
    s.bn = view(inst, 0, training);
    s.coef = SLN_COEF_BWD;
    if (s.bn.mode == SLN_BN_TRAIN) { s.x2 = x; s.ld2 = ldx; s.c2 = 0; }
    return s;
  }
  static Operand op1(const Seg& s, int rows) {
    Operand o; std::memset(&o, 0, sizeof(o));
    o.seg[0] = s; o.nseg = 1; o.rows = rows; o.cols = s.len;
    return o;
  }
  Operand layer_input(int gi, bool training) const {
    const Layer& ly = layers[gi];
    Operand o; std::memset(&o, 0, sizeof(o));
    const int D = ly.D;
    if (ly.first) {
      const float* X0 = ly.net == 0 ? X0e : X0d;
      const float* P0 = ly.net == 0 ? P0e : P0d;
      o.seg[0] = seg_ident(X0, D, 0, D, 1);
      o.seg[1] = seg_ident(P0, D, 0, D, 0);
      o.seg[2] = seg_ident(X0, D, 0, D, 2);
    } else {
      const Layer& pv = layers[gi - 1];
      o.seg[0] = seg_act(pv.A4, D, 0, D, pv.bn[3], 1, training);
      o.seg[1] = seg_act(pv.A2, 2 * H + D, H, D, pv.bn[1], 0, training);
      o.seg[2] = seg_act(pv.A4, D, 0, D, pv.bn[3], 2, training);
    }
    o.nseg = 3; o.rows = T; o.cols = 3 * D; o.idx_a = g.s; o.idx_b = g.o;
    return o;
  }
  Operand layer_output(int gi, bool training) const {
    const Layer& ly = layers[gi];
    return op1(seg_act(ly.A4, ly.Do, 0, ly.Do, ly.bn[3], 0, training), O);
  }

  // Group mode: between begin_group() and end_group() the Linear launches are recorded instead of issued; end_group()
  // issues them two at a time through sln_launch_gemm_group (the recorded problems must be independent of each other:
  // the two posterior heads of the encoder, box_net / angle_net of the decoder).
  struct GroupItem { GemmNTArgs nt; int epi; bool has_tn; GemmTNArgs tn; };
  bool grouping = false, use_group = true;
  std::vector<GroupItem> group_items;
  void begin_group() { grouping = use_group; group_items.clear(); }
  int launch_item(const GroupItem& it, hipStream_t st) {
    return it.has_tn ? sln_launch_gemm_dual(it.nt, it.epi, it.tn, st) : sln_launch_gemm_nt(it.nt, it.epi, -1, st);
  }
  int end_group(hipStream_t st) {
    grouping = false;
    for (size_t i = 0; i < group_items.size(); i += 2) {
      const GroupItem& a = group_items[i];
      if (i + 1 == group_items.size()) { int r = launch_item(a, st); if (r) return r; break; }
      const GroupItem& b2 = group_items[i + 1];
      GemmNTArgs nt[2] = {a.nt, b2.nt}; int epi[2] = {a.epi, b2.epi};
      GemmTNArgs tn[2]; int ntn = 0;
      if (a.has_tn) tn[ntn++] = a.tn;
      if (b2.has_tn) tn[ntn++] = b2.tn;
      int r = sln_launch_gemm_group(nt, epi, 2, tn, ntn, st);
      if (r == 1) { r = launch_item(a, st); if (!r) r = launch_item(b2, st); }
      if (r) return r;
    }
    group_items.clear();
    return 0;
  }

  int linear_fwd(const Operand& A, int ui, float* Y, int ldy, int ycol0, int M, int inst, bool training, hipStream_t st) {
    const Unit& u = units[ui];
    GemmNTArgs a; std::memset(&a, 0, sizeof(a));
    a.A = A; a.W = u.p.weight; a.bias = u.p.bias; a.Y = Y; a.ldy = ldy; a.ycol0 = ycol0;
    a.M = M; a.N = u.out; a.K = u.in; a.ldw = u.in;
    int epi = EPI_PLAIN;
    if (inst >= 0 && bn_mode(bns[inst], training) == SLN_BN_TRAIN) { epi = EPI_STATS; a.osums = bns[inst].sums; a.ocstride = bns[inst].C; }
    if (rec) { RecStep& r = rec_push(SK_NT); r.nt = a; r.epi = epi; return 0; }
    if (grouping) { GroupItem it; it.nt = a; it.epi = epi; it.has_tn = false; group_items.push_back(it); return 0; }
    return sln_launch_gemm_nt(a, epi, -1, st);
  }
  // dIn[M, in] = G[M, out] * W ; optional relu/BN mask of the producing stage (xprev, inst) and addend
  int linear_dgrad(const Operand& G, int ui, float* Y, int ldy, int M, const float* xprev, int ldx, int mask_inst,
                   bool masked, const float* addend, int ldadd, bool training, hipStream_t st) {
    const Unit& u = units[ui];
    GemmNTArgs a; std::memset(&a, 0, sizeof(a));
    a.A = G; a.W = u.wt; a.bias = nullptr; a.Y = Y; a.ldy = ldy; a.ycol0 = 0;
    a.M = M; a.N = u.in; a.K = G.cols; a.ldw = u.wt_ld;
    a.addend = addend; a.ldadd = ldadd; a.addcol0 = 0;
    int epi = EPI_PLAIN;
    if (masked) {
      epi = EPI_MASK; a.xprev = xprev; a.ldx = ldx; a.xcol0 = 0;
      a.obn = view(mask_inst, 0, training);
      if (a.obn.mode != SLN_BN_NONE) { a.ogsums = bns[mask_inst].gsums; a.ocstride = bns[mask_inst].C; }
    }
    if (rec) { RecStep& r = rec_push(SK_NT); r.nt = a; r.epi = epi; return 0; }
    if (grouping) {
      GroupItem it; it.nt = a; it.epi = epi; it.has_tn = !pending.empty();
      if (it.has_tn) { it.tn = pending.front(); pending.erase(pending.begin()); }
      group_items.push_back(it);
      return 0;
    }
    if (!pending.empty()) {
      const GemmTNArgs t = pending.front();
      pending.erase(pending.begin());
      return sln_launch_gemm_dual(a, epi, t, st);
    }
    return sln_launch_gemm_nt(a, epi, -1, st);
  }
  int linear_wgrad(const Operand& G, const Operand& X, int ui, int R, hipStream_t st) {
    const Unit& u = units[ui];
    GemmTNArgs a; std::memset(&a, 0, sizeof(a));
    a.G = G; a.X = X; a.dW = u.p.d_weight; a.db = u.p.d_bias; a.lddw = u.in;
    a.R = R; a.Nout = u.out; a.Kin = u.in; a.rows_per_block = 0;
    if (rec) { rec_push(SK_TN).tn = a; return 0; }
    if (defer) { deferred.push_back(a); return 0; }
    if (use_dual) { pending.push_back(a); return 0; }
    if (use_side && side) {
      int r = fork_side(st);
      if (r) return r;
      return sln_launch_gemm_tn(a, -1, side);
    }
    return sln_launch_gemm_tn(a, -1, st);
  }

  // ---- launch sites of the decoder path: launch, or record (see Recorder) ----
  Recorder* rec = nullptr;
  RecStep& rec_push(int kind) { rec->steps.emplace_back(); RecStep& r = rec->steps.back(); r.kind = kind; return r; }
  int k_scatter_fwd(const float* A2, int ld, int Hh, int D, BnView bn, GraphCsr gg, int Oo, float* pooled, hipStream_t st) {
    if (!rec) return sln_launch_scatter_avg_fwd(A2, ld, Hh, D, bn, gg, Oo, pooled, st);
    MScatterFwd& m = rec_push(SK_SCATTER_FWD).sf; std::memset(&m, 0, sizeof(m));
    m.A2 = A2; m.ld = ld; m.H = Hh; m.D = D; m.bn = bn; m.g = gg; m.O = Oo; m.pooled = pooled;
    return 0;
  }
  int k_scatter_bwd(const float* dM_, const float* dP, int lddp, int dpcol0, const float* A2, int ld, int Hh, int D, BnView bn, GraphCsr gg, int Tt,
                    float* g2, double* gsums, int cstride, hipStream_t st) {
    if (!rec) return sln_launch_scatter_avg_bwd(dM_, dP, lddp, dpcol0, A2, ld, Hh, D, bn, gg, Tt, g2, gsums, cstride, st);
    MScatterBwd& m = rec_push(SK_SCATTER_BWD).sb; std::memset(&m, 0, sizeof(m));
    m.dM = dM_; m.dP = dP; m.lddp = lddp; m.dpcol0 = dpcol0; m.A2 = A2; m.ld = ld; m.H = Hh; m.D = D; m.bn = bn; m.g = gg; m.T = Tt; m.g2 = g2;
    m.gsums = gsums; m.cstride = cstride;
    return 0;
  }
  int k_gather_bwd(const float* dGp, int ldg, int D, GraphCsr gg, int Oo, const float* add1, int ldadd1, const float* xprev, int ldx, BnView bn,
                   int masked, float* out, int ldo, double* gsums, int cstride, hipStream_t st) {
    if (!rec) return sln_launch_gather_bwd(dGp, ldg, D, gg, Oo, add1, ldadd1, xprev, ldx, bn, masked, out, ldo, gsums, cstride, st);
    MGatherBwd& m = rec_push(SK_GATHER_BWD).gb; std::memset(&m, 0, sizeof(m));
    m.dG = dGp; m.ldg = ldg; m.D = D; m.g = gg; m.O = Oo; m.add1 = add1; m.ldadd1 = ldadd1; m.xprev = xprev; m.ldx = ldx; m.bn = bn; m.masked = masked;
    m.out = out; m.ldo = ldo; m.gsums = gsums; m.cstride = cstride;
    return 0;
  }
  int k_mask_gstats(const float* d1, int ld1, const float* d2, int ld2, const float* xprev, int ldx, BnView bn, int rows, int cols, float* out, int ldo,
                    double* gsums, int cstride, hipStream_t st) {
    if (!rec) return sln_launch_mask_gstats(d1, ld1, d2, ld2, xprev, ldx, bn, rows, cols, out, ldo, gsums, cstride, st);
    MMaskGstats& m = rec_push(SK_MASK_GSTATS).mg; std::memset(&m, 0, sizeof(m));
    m.d1 = d1; m.ld1 = ld1; m.d2 = d2; m.ld2 = ld2; m.xprev = xprev; m.ldx = ldx; m.bn = bn; m.rows = rows; m.cols = cols; m.out = out; m.ldo = ldo;
    m.gsums = gsums; m.cstride = cstride;
    return 0;
  }
  int k_embed_gather(const int* idx, const float* emb, int rows, int n, float* out, hipStream_t st) {
    if (!rec) return sln_launch_embed_gather_i32(idx, emb, rows, n, out, st);
    MEmbedGather& m = rec_push(SK_EMBED_GATHER).eg; std::memset(&m, 0, sizeof(m));
    m.idx = idx; m.emb = emb; m.rows = rows; m.n = n; m.out = out;
    return 0;
  }
  int k_embed_bwd(const void* idx, int idx64, const float* d, int ld, int col0, int rows, int n, int table_rows, float* d_emb, hipStream_t st) {
    if (!rec) return idx64 ? sln_launch_embed_bwd_i64(static_cast<const int64_t*>(idx), d, ld, col0, rows, n, table_rows, d_emb, st)
                           : sln_launch_embed_bwd_i32(static_cast<const int*>(idx), d, ld, col0, rows, n, table_rows, d_emb, st);
    RecStep& r = rec_push(SK_EMBED_BWD); r.idx64 = idx64;
    MEmbedBwd& m = r.eb; std::memset(&m, 0, sizeof(m));
    m.idx = idx; m.d = d; m.ld = ld; m.col0 = col0; m.rows = rows; m.n = n; m.table_rows = table_rows; m.d_emb = d_emb;
    return 0;
  }
  int k_add2(const float* a, int lda, const float* b, int ldb, int rows, int cols, float* out, int ldo, hipStream_t st) {
    if (!rec) return sln_launch_add2(a, lda, b, ldb, rows, cols, out, ldo, st);
    MAdd2& m = rec_push(SK_ADD2).a2; std::memset(&m, 0, sizeof(m));
    m.a = a; m.lda = lda; m.b = b; m.ldb = ldb; m.rows = rows; m.cols = cols; m.out = out; m.ldo = ldo;
    return 0;
  }
  int k_dec_assemble(const DecAssemble& da, hipStream_t st) {
    if (!rec) return sln_launch_dec_assemble(da, st);
    MDecAssemble& m = rec_push(SK_DEC_ASSEMBLE).da; std::memset(&m, 0, sizeof(m));
    m.a = da;
    return 0;
  }
  int k_dec_assemble_bwd(const DecAssembleBwd& db, hipStream_t st) {
    if (!rec) return sln_launch_dec_assemble_bwd(db, st);
    if (g_sln_deterministic && db.rows_obj > 0 && (db.n_attr == 0 || db.rows_attr > 0)) {
      // the deterministic form of sln_launch_dec_assemble_bwd, piece by piece: one table at a time in row order, then the dz columns
      const int ld = db.n_obj + db.n_attr + (db.z_in_x0 ? db.n_z : 0);
      int r = k_embed_bwd(db.objs, 1, db.dx0, ld, 0, db.O, db.n_obj, db.rows_obj, db.d_obj_emb, st);
      if (!r && db.n_attr > 0) r = k_embed_bwd(db.attrs, 1, db.dx0, ld, db.n_obj, db.O, db.n_attr, db.rows_attr, db.d_attr_emb, st);
      if (r) return r;
      if (db.z_in_x0 && db.dz) {
        MAdd2& m = rec_push(SK_COPY2D).a2; std::memset(&m, 0, sizeof(m));
        m.a = db.dx0 + db.n_obj + db.n_attr; m.lda = ld; m.b = nullptr; m.ldb = 0; m.rows = db.O; m.cols = db.n_z; m.out = db.dz; m.ldo = db.n_z;
      }
      return 0;
    }
    rec_push(SK_DEC_ASSEMBLE_BWD).dab = db;
    return 0;
  }

  size_t carve(void* base, int mo, int mt);
  int refresh_transposes(hipStream_t st);
  int gconv_forward(int gi, bool training, hipStream_t st);
  int gconv_backward(int gi, const float* dP, int lddp, int dpcol0, int slot, bool training, hipStream_t st);
  int encoder_forward(bool training, hipStream_t st);
  int decoder_forward(const float* z_ext, const float* eps, bool training, hipStream_t st);
  int decoder_backward(hipStream_t st);
  int encoder_backward(hipStream_t st);
  int loss(const float* bp, const float* ap, const float* mu_, const float* lv_, bool with_grads, hipStream_t st);
  int run_bn_updates(int first, int count, hipStream_t st);
  enum { TRAIN_BACKWARD = 0, TRAIN_FULL = 1, TRAIN_UPTO_DECODER = 2, TRAIN_ENCODER_BWD = 3 };   // = SLN_TRAIN_* of sln_hip.h
  int train_iteration(const float* eps, int mode, hipStream_t st);
  // input of box_net (with_attr) / angle_net: [obj_vecs | attr_vecs] resp. obj_vecs, where obj_vecs is the decoder gconv
  // output - with decoder_cat off followed by z (Sg2ScVAE_model.py:162-171)
  Operand head_input(bool with_attr, bool training) const {
    const Layer& ll = layers[2 * L - 1];
    Operand o; std::memset(&o, 0, sizeof(o));
    int n = 0, cols = 0;
    o.seg[n++] = seg_act(ll.A4, Ddc, 0, Ddc, ll.bn[3], 0, training); cols += Ddc;
    if (!cfg.decoder_cat) { o.seg[n++] = seg_ident(z, E, 0, E, 0); cols += E; }
    if (with_attr && n_attr_e > 0) { o.seg[n++] = seg_ident(t.attr_emb_dc, n_attr_e, 0, n_attr_e, 1); cols += n_attr_e; o.idx_a = attrs32; }
    o.nseg = n; o.rows = O; o.cols = cols;
    return o;
  }
  void drop_graphs() {
    for (int i = 0; i < 4; ++i)
      if (graph_exec[i]) { (void)hipGraphExecDestroy(graph_exec[i]); graph_exec[i] = nullptr; }
  }
};

#define RET_IF(x) do { int r__ = (x); if (r__ != 0) return r__; } while (0)
#define HIP_RET(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) return (int)e__; } while (0)

size_t SlnVae::carve(void* base, int mo, int mt) {
  Bump b(base);
  const size_t Om = (size_t)mo, Tm = (size_t)mt;
  g.s = b.take<int>(Tm); g.p = b.take<int>(Tm); g.o = b.take<int>(Tm);
  g.deg = b.take<int>(Om); g.invdeg = b.take<float>(Om); g.rowptr = b.take<int>(Om + 1);
  g.cursor = b.take<int>(Om); g.ent = b.take<int>(2 * Tm);
  attrs32 = b.take<int>(Om); err_flag = b.take<int>(4);
  // engine-owned copies of the batch inputs (see sln_vae_set_batch)
  st_objs = b.take<int64_t>(Om); st_angles = b.take<int64_t>(Om); st_attrs = b.take<int64_t>(Om); st_boxes = b.take<float>(Om * 6);
  scalars = b.take<AdamScalars>(1); losses = b.take<float>(4);
  // BatchNorm statistics arena
  size_t nd = 0, nd_enc = 0;
  for (size_t i = 0; i < bns.size(); ++i) { if ((int)i == n_bn_enc) nd_enc = nd; nd += 2 * (size_t)bns[i].C; }
  if ((int)bns.size() == n_bn_enc) nd_enc = nd;
  stats_doubles = nd; enc_stats_doubles = nd_enc;
  // [loss accumulators | forward sums | backward sums]: one region, cleared by ONE memset per training iteration
  loss_acc = b.take<double>(4); stats_base = b.take<double>(nd); gstats_base = b.take<double>(nd);
  zero_begin = reinterpret_cast<char*>(loss_acc);
  zero_bytes = (size_t)(reinterpret_cast<char*>(gstats_base + nd) - zero_begin);
  size_t o = 0;
  for (auto& bi : bns) { bi.sums = stats_base ? stats_base + o : nullptr; bi.gsums = gstats_base ? gstats_base + o : nullptr; o += 2 * (size_t)bi.C; }
  bn_table_dev = b.take<BnTableEntry>(bns.size() + 1);
  tr_table_dev = b.take<TransposeEntry>(units.size() + 1);
  for (auto& u : units) { u.wt_ld = rup(u.out, 4); u.wt = b.take<float>((size_t)u.in * u.wt_ld); }
  const size_t W = 2 * (size_t)E;
  X0e = b.take<float>(Om * Dec); P0e = b.take<float>(Tm * Dec);
  X0d = b.take<float>(Om * Ddc); P0d = b.take<float>(Tm * Ddc);
  for (auto& ly : layers) {
    const size_t D = ly.Do;
    ly.A1 = b.take<float>(Tm * H); ly.A2 = b.take<float>(Tm * (2 * H + D)); ly.M = b.take<float>(Om * H);
    ly.A3 = b.take<float>(Om * H); ly.A4 = b.take<float>(Om * D);
  }
  hbA1 = b.take<float>(Om * H); hbA2 = b.take<float>(Om * W); haA1 = b.take<float>(Om * H); haA2 = b.take<float>(Om * W);
  mu = b.take<float>(Om * E); logvar = b.take<float>(Om * E); z = b.take<float>(Om * E); eps_buf = b.take<float>(Om * E);
  bnA1 = b.take<float>(Om * H); anA1 = b.take<float>(Om * H);
  boxes_pred = b.take<float>(Om * cfg.box_dim); logits = b.take<float>(Om * cfg.n_angle);
  angles_pred = b.take<float>(Om * cfg.n_angle);
  dbp_ld = rup(cfg.box_dim, 4);
  dbp = b.take<float>(Om * dbp_ld); dlogits = b.take<float>(Om * cfg.n_angle);
  g_bn = b.take<float>(Om * H); g_an = b.take<float>(Om * H);
  d_bx = b.take<float>(Om * (W + n_attr_e)); d_ax = b.take<float>(Om * W);
  const size_t Dm = (size_t)(Dec > Ddc ? Dec : Ddc);      // = 2E for the VAE
  dM = b.take<float>(Om * H);
  for (auto& ly : layers) {
    const size_t D = ly.Do;
    ly.g4 = b.take<float>(Om * D); ly.g3 = b.take<float>(Om * H); ly.g2 = b.take<float>(Tm * (2 * H + D)); ly.g1 = b.take<float>(Tm * H);
  }
  for (int set = 0; set < 2; ++set)
    for (int w = 0; w < tn_slots; ++w)
      for (int k = 0; k < 2; ++k) {
        // a per-layer slot holds 4 problems, a per-pass slot all of a pass
        GemmTNArgs* dp = b.take<GemmTNArgs>(SLN_TN_MULTI_MAX);
        TnMultiMeta* dm = b.take<TnMultiMeta>(1);
        if (!b.dry && tn_groups_store) {  // (a dry run works on a COPY of the handle that shares the tables: leave them alone)
          TnGroup& tg = tn_group(set, w, k);
          tg.dev_probs = dp; tg.dev_meta = dm; tg.dirty = true; tg.n = 0;
        }
      }
  dG[0] = b.take<float>(Tm * 3 * Dm); dG[1] = b.take<float>(Tm * 3 * Dm);
  dX0 = b.take<float>(Om * Dm); dz = b.take<float>(Om * E); dmu = b.take<float>(Om * E); dlv = b.take<float>(Om * E);
  g_h2 = b.take<float>(Om * W); g_h1 = b.take<float>(Om * H); d_xb = b.take<float>(Om * W); d_xa = b.take<float>(Om * W);
  tmp_d = b.take<float>(Om * W);
  g_h2b = b.take<float>(Om * W); g_h1b = b.take<float>(Om * H); tmp_db = b.take<float>(Om * W);
  return (b.off + 255) & ~size_t(255);
}

int SlnVae::refresh_transposes(hipStream_t st) {
  if (wt_fresh) return 0;
  RET_IF(sln_launch_transpose_table(tr_table_dev, n_tr, tr_max_tiles, st));
  wt_fresh = true;
  return 0;
}

int SlnVae::run_bn_updates(int first, int count, hipStream_t st) {
  if (count <= 0) return 0;
  int maxc = 0;
  for (int i = first; i < first + count; ++i) maxc = bns[i].C > maxc ? bns[i].C : maxc;
  return sln_launch_bn_running_update(bn_table_dev + first, count, maxc, kBnMomentum, cfg.recurrent ? 0 : 1, st, T, O);
}

// One GraphTripleConv forward (models/graph.py:57-111) as 5 launches.
int SlnVae::gconv_forward(int gi, bool training, hipStream_t st) {
  Layer& ly = layers[gi];
  const int Do = ly.Do;
  RET_IF(linear_fwd(layer_input(gi, training), ly.u0 + 0, ly.A1, H, 0, T, ly.bn[0], training, st));
  RET_IF(linear_fwd(op1(seg_act(ly.A1, H, 0, H, ly.bn[0], 0, training), T), ly.u0 + 1, ly.A2, 2 * H + Do, 0, T, ly.bn[1],
                    training, st));
  RET_IF(k_scatter_fwd(ly.A2, 2 * H + Do, H, Do, view(ly.bn[1], 0, training), g, O, ly.M, st));
  RET_IF(linear_fwd(op1(seg_ident(ly.M, H, 0, H, 0), O), ly.u0 + 2, ly.A3, H, 0, O, ly.bn[2], training, st));
  RET_IF(linear_fwd(op1(seg_act(ly.A3, H, 0, H, ly.bn[2], 0, training), O), ly.u0 + 3, ly.A4, Do, 0, O, ly.bn[3], training,
                    st));
  return 0;
}

// Backward of one GraphTripleConv.  On entry g4 holds the relu-masked gradient w.r.t. this layer's
// object output (and bn[3].gsums its column sums); dP the gradient w.r.t. its predicate output
// (a column slice of the next layer's dG) or nullptr.  Writes dG[slot] = gradient w.r.t. the
// gathered [obj[s] | pred | obj[o]] input.
int SlnVae::gconv_backward(int gi, const float* dP, int lddp, int dpcol0, int slot, bool tr, hipStream_t st) {
  Layer& ly = layers[gi];
  const int D = ly.D, Do = ly.Do, C2 = 2 * H + Do;
  // net2.1 : h3 -> A4
  float *g1 = ly.g1, *g2 = ly.g2, *g3 = ly.g3, *g4 = ly.g4;
  Operand G4 = op1(seg_bwd(g4, Do, ly.A4, Do, Do, ly.bn[3], tr), O);
  RET_IF(linear_wgrad(G4, op1(seg_act(ly.A3, H, 0, H, ly.bn[2], 0, tr), O), ly.u0 + 3, O, st));
  RET_IF(linear_dgrad(G4, ly.u0 + 3, g3, H, O, ly.A3, H, ly.bn[2], true, nullptr, 0, tr, st));
  // net2.0 : pooled -> A3
  Operand G3 = op1(seg_bwd(g3, H, ly.A3, H, H, ly.bn[2], tr), O);
  RET_IF(linear_wgrad(G3, op1(seg_ident(ly.M, H, 0, H, 0), O), ly.u0 + 2, O, st));
  RET_IF(linear_dgrad(G3, ly.u0 + 2, dM, H, O, nullptr, 0, -1, false, nullptr, 0, tr, st));
  // avg-pool backward (a gather) + relu/BN mask of A2
  BnView v2 = view(ly.bn[1], 0, tr);
  RET_IF(k_scatter_bwd(dM, dP, lddp, dpcol0, ly.A2, C2, H, Do, v2, g, T, g2,
                       v2.mode != SLN_BN_NONE ? bns[ly.bn[1]].gsums : nullptr, C2, st));
  // net1.1 : h1 -> A2
  Operand G2 = op1(seg_bwd(g2, C2, ly.A2, C2, C2, ly.bn[1], tr), T);
  RET_IF(linear_wgrad(G2, op1(seg_act(ly.A1, H, 0, H, ly.bn[0], 0, tr), T), ly.u0 + 1, T, st));
  RET_IF(linear_dgrad(G2, ly.u0 + 1, g1, H, T, ly.A1, H, ly.bn[0], true, nullptr, 0, tr, st));
  // net1.0 : gathered concat -> A1
  Operand G1 = op1(seg_bwd(g1, H, ly.A1, H, H, ly.bn[0], tr), T);
  RET_IF(linear_wgrad(G1, layer_input(gi, tr), ly.u0 + 0, T, st));
  RET_IF(linear_dgrad(G1, ly.u0 + 0, dG[slot], 3 * D, T, nullptr, 0, -1, false, nullptr, 0, tr, st));
  RET_IF(join_side(st));      // (pairing / side-stream modes: their wgrads are launched before the next layer starts)
  // deterministic mode with shared (recurrent) weights: the layers' wgrads add into the SAME dW, so they run as separate
  // launches in stream order instead of side by side in one
  if (tn_per_layer || (g_sln_deterministic && cfg.recurrent) || rec) RET_IF(flush_deferred(ly.net == 0 ? 1 : 0, st));     // (a group runs a layer's wgrads on its side stream, next to the following layers' dgrad chain)
  return 0;
}

int SlnVae::encoder_forward(bool training, hipStream_t st) {
  if (training && enc_stats_doubles && !bulk_zeroed) RET_IF(sln_zero_async(stats_base, enc_stats_doubles * sizeof(double), st));
  EncAssemble ea; std::memset(&ea, 0, sizeof(ea));
  ea.objs = batch.objs; ea.attrs = batch.attributes; ea.angles = batch.angles; ea.boxes = batch.boxes;
  ea.obj_emb = t.obj_emb_ec; ea.attr_emb = t.attr_emb_ec; ea.angle_emb = t.angle_emb; ea.wb = t.box_emb_w; ea.bb = t.box_emb_b;
  ea.O = O; ea.n_obj = n_obj_e; ea.n_attr = n_attr_e; ea.n_box = n_box_e; ea.n_angle = n_angle_e; ea.box_dim = cfg.box_dim;
  ea.x0 = X0e;
  if (it_prologue) {
    StepPrologue sp; std::memset(&sp, 0, sizeof(sp));
    sp.eps = draw_eps ? eps_buf : nullptr; sp.n_eps = (long)O * E; sp.scalars = scalars;        // Sg2ScVAE_model.py:182
    sp.enc = ea; sp.pidx = g.p; sp.T = T;
    sp.pemb_ec = t.pred_emb_ec; sp.n_ec = Dec; sp.p0e = P0e;
    sp.pemb_dc = t.pred_emb_dc; sp.n_dc = Ddc; sp.p0d = P0d;
    if (it_zero_in_prologue) { sp.zero_ptr = zero_begin; sp.zero_bytes = (long)zero_bytes; }
    RET_IF(sln_launch_step_prologue(sp, st));
  } else {
    RET_IF(sln_launch_enc_assemble(ea, st));
    RET_IF(sln_launch_embed_gather_i32(g.p, t.pred_emb_ec, T, Dec, P0e, st));
  }
  for (int l = 0; l < L; ++l) RET_IF(gconv_forward(l, training, st));
  const Operand XL = layer_output(L - 1, training);
  const int W = 2 * E;
  // box_mean_var / box_mean / box_var (Sg2ScVAE_model.py:134-136) and angle_mean_var / angle_mean / angle_var (:138-140):
  // two identical-shape branches, issued stage by stage as grouped launches
  begin_group();
  RET_IF(linear_fwd(XL, 0, hbA1, H, 0, O, bn_head[0], training, st));
  RET_IF(linear_fwd(XL, 4, haA1, H, 0, O, bn_head[2], training, st));
  RET_IF(end_group(st));
  begin_group();
  RET_IF(linear_fwd(op1(seg_act(hbA1, H, 0, H, bn_head[0], 0, training), O), 1, hbA2, W, 0, O, bn_head[1], training, st));
  RET_IF(linear_fwd(op1(seg_act(haA1, H, 0, H, bn_head[2], 0, training), O), 5, haA2, W, 0, O, bn_head[3], training, st));
  RET_IF(end_group(st));
  const Operand HB = op1(seg_act(hbA2, W, 0, W, bn_head[1], 0, training), O);
  const Operand HA = op1(seg_act(haA2, W, 0, W, bn_head[3], 0, training), O);
  begin_group();
  RET_IF(linear_fwd(HB, 2, mu, E, 0, O, -1, training, st));
  RET_IF(linear_fwd(HA, 6, mu, E, n_box_e, O, -1, training, st));
  RET_IF(linear_fwd(HB, 3, logvar, E, 0, O, -1, training, st));
  RET_IF(linear_fwd(HA, 7, logvar, E, n_box_e, O, -1, training, st));
  RET_IF(end_group(st));
  if (training && !it_merge_bn) RET_IF(run_bn_updates(0, n_bn_enc, st));
  enc_training = training; have_enc = true;
  return 0;
}

int SlnVae::decoder_forward(const float* z_ext, const float* eps, bool training, hipStream_t st) {
  const size_t dec_doubles = stats_doubles - enc_stats_doubles;
  if (training && dec_doubles && !bulk_zeroed) RET_IF(sln_zero_async(stats_base + enc_stats_doubles, dec_doubles * sizeof(double), st));
  DecAssemble da; std::memset(&da, 0, sizeof(da));
  da.objs = batch.objs; da.attrs = batch.attributes; da.obj_emb = t.obj_emb_dc; da.attr_emb = t.attr_emb_dc;
  da.mu = mu; da.logvar = logvar; da.eps = eps; da.z_in = z_ext;
  da.O = O; da.n_obj = n_obj_e; da.n_attr = n_attr_e; da.n_z = E; da.use_ae = cfg.use_ae;
  da.z = z; da.x0 = X0d; da.z_in_x0 = cfg.decoder_cat ? 1 : 0;
  RET_IF(k_dec_assemble(da, st));
  if (!it_prologue) RET_IF(k_embed_gather(g.p, t.pred_emb_dc, T, Ddc, P0d, st));
  for (int l = 0; l < L; ++l) RET_IF(gconv_forward(L + l, training, st));
  // box_net([obj_vecs | attr_vecs]) and angle_net(obj_vecs)  (Sg2ScVAE_model.py:166-171)
  begin_group();
  RET_IF(linear_fwd(head_input(true, training), unit_boxnet(0), bnA1, H, 0, O, bn_head[4], training, st));
  RET_IF(linear_fwd(head_input(false, training), unit_anglenet(0), anA1, H, 0, O, bn_head[5], training, st));
  RET_IF(end_group(st));
  begin_group();
  RET_IF(linear_fwd(op1(seg_act(bnA1, H, 0, H, bn_head[4], 0, training), O), unit_boxnet(1), boxes_pred, cfg.box_dim, 0, O, -1,
                    training, st));
  RET_IF(linear_fwd(op1(seg_act(anA1, H, 0, H, bn_head[5], 0, training), O), unit_anglenet(1), logits, cfg.n_angle, 0, O, -1,
                    training, st));
  RET_IF(end_group(st));
  if (!it_fused_loss && !rec) RET_IF(sln_launch_log_softmax(logits, angles_pred, O, cfg.n_angle, st));     // (a group takes it over all rooms' rows at once)
  if (training) {
    if (it_merge_bn) RET_IF(run_bn_updates(0, (int)bns.size(), st));      // encoder's and decoder's tables: one launch, application order
    else RET_IF(run_bn_updates(n_bn_enc, (int)bns.size() - n_bn_enc, st));
  }
  dec_training = training; have_dec = true; z_from_latent = (z_ext == nullptr);
  return 0;
}

int SlnVae::loss(const float* bp, const float* ap, const float* mu_, const float* lv_, bool with_grads, hipStream_t st) {
  LossArgs a; std::memset(&a, 0, sizeof(a));
  a.boxes = batch.boxes; a.boxes_pred = bp; a.box_dim = cfg.box_dim;
  a.angles = batch.angles; a.logits = logits; a.angles_pred = const_cast<float*>(ap); a.n_angle = cfg.n_angle;
  a.mu = mu_; a.logvar = lv_; a.n_z = E; a.use_ae = cfg.use_ae; a.kl_weight = &scalars->kl_weight;
  a.O = O; a.acc = loss_acc; a.losses = losses; a.acc_prezeroed = bulk_zeroed ? 1 : 0;
  a.d_boxes_pred = with_grads ? dbp : nullptr; a.d_logits = with_grads ? dlogits : nullptr; a.ld_dbp = dbp_ld;
  a.from_logits = (it_fused_loss && ap == angles_pred) ? 1 : 0;
  RET_IF(sln_launch_loss(a, st));
  have_loss_grads = with_grads;
  return 0;
}

// Backward of decoder(): expects dbp (padded) and dlogits filled.
int SlnVae::decoder_backward(hipStream_t st) {
  const bool tr = dec_training;
  ev_next = 0;
  tn_slot_next[0] = 0;
  const size_t dec_doubles = stats_doubles - enc_stats_doubles;
  if (dec_doubles && !bulk_zeroed && !rec) RET_IF(sln_zero_async(gstats_base + enc_stats_doubles, dec_doubles * sizeof(double), st));
  if (!rec) RET_IF(refresh_transposes(st));        // (a group clears the sums and rebuilds the decoder's W^T of all rooms in one launch each)
  const int last = 2 * L - 1;
  const Layer& ll = layers[last];
  // W: width of the gconv vectors; Wh: width of the heads' input without the attribute columns (= W, plus z when decoder_cat is off)
  const int W = Ddc, Wh = 2 * E, WA = Wh + n_attr_e;
  // box_net.1 and angle_net.1 (grouped), then box_net.0 and angle_net.0 (grouped)
  Operand Gb = op1(seg_ident(dbp, dbp_ld, 0, dbp_ld, 0), O); Gb.cols = dbp_ld;
  Operand Ga = op1(seg_ident(dlogits, cfg.n_angle, 0, cfg.n_angle, 0), O);
  begin_group();
  {
    Operand Gw = Gb; Gw.seg[0].len = cfg.box_dim; Gw.cols = cfg.box_dim;   // wgrad masks the padded columns itself
    RET_IF(linear_wgrad(Gw, op1(seg_act(bnA1, H, 0, H, bn_head[4], 0, tr), O), unit_boxnet(1), O, st));
  }
  RET_IF(linear_dgrad(Gb, unit_boxnet(1), g_bn, H, O, bnA1, H, bn_head[4], true, nullptr, 0, tr, st));
  RET_IF(linear_wgrad(Ga, op1(seg_act(anA1, H, 0, H, bn_head[5], 0, tr), O), unit_anglenet(1), O, st));
  RET_IF(linear_dgrad(Ga, unit_anglenet(1), g_an, H, O, anA1, H, bn_head[5], true, nullptr, 0, tr, st));
  RET_IF(end_group(st));
  Operand G0 = op1(seg_bwd(g_bn, H, bnA1, H, H, bn_head[4], tr), O);
  const Operand XA = head_input(true, tr);
  Operand Ga0 = op1(seg_bwd(g_an, H, anA1, H, H, bn_head[5], tr), O);
  begin_group();
  RET_IF(linear_wgrad(G0, XA, unit_boxnet(0), O, st));
  RET_IF(linear_dgrad(G0, unit_boxnet(0), d_bx, WA, O, nullptr, 0, -1, false, nullptr, 0, tr, st));
  RET_IF(linear_wgrad(Ga0, head_input(false, tr), unit_anglenet(0), O, st));
  RET_IF(linear_dgrad(Ga0, unit_anglenet(0), d_ax, Wh, O, nullptr, 0, -1, false, nullptr, 0, tr, st));
  RET_IF(end_group(st));
  RET_IF(join_side(st));      // g_bn / g_an / dbp / dlogits consumers done before g4 is produced
  // junction: obj_vecs feeds box_net (first W columns of d_bx) and angle_net
  {
    BnView v = view(ll.bn[3], 0, tr);
    RET_IF(k_mask_gstats(d_bx, WA, d_ax, Wh, ll.A4, W, v, O, W, ll.g4, W,
                         v.mode != SLN_BN_NONE ? bns[ll.bn[3]].gsums : nullptr, W, st));
  }
  // decoder_cat off: z entered behind the gconv net, its gradient is the sum of the two heads' (Sg2ScVAE_model.py:164)
  if (!cfg.decoder_cat) RET_IF(k_add2(d_bx + W, WA, d_ax + W, Wh, O, E, dz, E, st));
  if (n_attr_e > 0) RET_IF(k_embed_bwd(batch.attributes, 1, d_bx, WA, Wh, O, n_attr_e, cfg.num_attrs, t.d_attr_emb_dc, st));
  // gconv layers, last to first
  for (int l = L - 1; l >= 0; --l) {
    const int gi = L + l, slot = l & 1;
    const float* dP = (l == L - 1) ? nullptr : dG[slot ^ 1];
    RET_IF(gconv_backward(gi, dP, 3 * W, W, slot, tr, st));
    if (l > 0) {
      const Layer& pv = layers[gi - 1];
      BnView v = view(pv.bn[3], 0, tr);
      RET_IF(k_gather_bwd(dG[slot], 3 * W, W, g, O, nullptr, 0, pv.A4, W, v, 1, pv.g4, W,
                          v.mode != SLN_BN_NONE ? bns[pv.bn[3]].gsums : nullptr, W, st));
    } else {
      BnView none = view(-1, 0, tr);
      RET_IF(k_gather_bwd(dG[slot], 3 * W, W, g, O, nullptr, 0, nullptr, 0, none, 0, dX0, W, nullptr, 0, st));
      RET_IF(k_embed_bwd(g.p, 0, dG[slot], 3 * W, W, T, W, cfg.num_preds, t.d_pred_emb_dc, st));
    }
  }
  DecAssembleBwd db; std::memset(&db, 0, sizeof(db));
  db.objs = batch.objs; db.attrs = batch.attributes; db.dx0 = dX0; db.O = O; db.n_obj = n_obj_e; db.n_attr = n_attr_e; db.n_z = E;
  db.d_obj_emb = t.d_obj_emb_dc; db.d_attr_emb = t.d_attr_emb_dc; db.dz = dz; db.z_in_x0 = cfg.decoder_cat ? 1 : 0;
  db.rows_obj = cfg.num_objs; db.rows_attr = cfg.num_attrs;
  RET_IF(k_dec_assemble_bwd(db, st));
  const int nb = (int)bns.size() - n_bn_enc;
  if (nb > 0 && !it_merge_bn && !rec) {
    int maxc = 0;
    for (int i = n_bn_enc; i < (int)bns.size(); ++i) maxc = bns[i].C > maxc ? bns[i].C : maxc;
    RET_IF(sln_launch_bn_param_grads(bn_table_dev + n_bn_enc, nb, maxc, cfg.recurrent ? 0 : 1, st));
  }
  // (round 6, full iterations only: the decoder pass's wgrads wait and share the encoder pass's launches - two multi-problem launches
  //  per step instead of four, 1.871 -> 1.845 ms per 64-graph step; their operands are per-layer buffers that nothing overwrites
  //  before the end of the iteration.  SLN_TN_ONE_FLUSH=0 restores the flush per pass; the two-half data-parallel form always
  //  flushes here: the decoder's half of the gradient must be final for its all-reduce)
  if (!it_one_flush) RET_IF(flush_deferred(0, st));       // every decoder-side wgrad: after this launch the upper half of the flat gradient is final
  return 0;
}

// Backward of encoder(): expects dmu / dlv filled.
int SlnVae::encoder_backward(hipStream_t st) {
  const bool tr = enc_training;
  if (ev_next > 4096) ev_next = 0;
  tn_slot_next[1] = tn_slot_next[2] = 0;
  if (enc_stats_doubles && !bulk_zeroed) RET_IF(sln_zero_async(gstats_base, enc_stats_doubles * sizeof(double), st));
  RET_IF(refresh_transposes(st));
  const int last = L - 1, W = 2 * E;
  const Layer& ll = layers[last];
  const Operand XL = layer_output(last, tr);
  float* d_x[2] = {d_xb, d_xa};
  // 0: box branch (units 0..3), 1: angle branch (units 4..7); the branches are independent and identical in shape, so every
  // stage issues both of them as one grouped launch (each branch has its own temporaries)
  float* hA1v[2] = {hbA1, haA1}; float* hA2v[2] = {hbA2, haA2};
  float* tmpv[2] = {tmp_d, tmp_db}; float* gh2v[2] = {g_h2, g_h2b}; float* gh1v[2] = {g_h1, g_h1b};
  const int c0v[2] = {0, n_box_e}, nv[2] = {n_box_e, n_angle_e};
  begin_group();
  for (int br = 0; br < 2; ++br) {
    const Operand Hh = op1(seg_act(hA2v[br], W, 0, W, bn_head[br * 2 + 1], 0, tr), O);
    Operand Gm = op1(seg_ident(dmu, E, c0v[br], nv[br], 0), O);
    RET_IF(linear_wgrad(Gm, Hh, br * 4 + 2, O, st));
    RET_IF(linear_dgrad(Gm, br * 4 + 2, tmpv[br], W, O, nullptr, 0, -1, false, nullptr, 0, tr, st));
  }
  RET_IF(end_group(st));
  begin_group();
  for (int br = 0; br < 2; ++br) {
    const int b1 = bn_head[br * 2 + 1];
    const Operand Hh = op1(seg_act(hA2v[br], W, 0, W, b1, 0, tr), O);
    Operand Gv = op1(seg_ident(dlv, E, c0v[br], nv[br], 0), O);
    RET_IF(linear_wgrad(Gv, Hh, br * 4 + 3, O, st));
    RET_IF(linear_dgrad(Gv, br * 4 + 3, gh2v[br], W, O, hA2v[br], W, b1, true, tmpv[br], W, tr, st));
  }
  RET_IF(end_group(st));
  begin_group();
  for (int br = 0; br < 2; ++br) {
    const int b0 = bn_head[br * 2], b1 = bn_head[br * 2 + 1];
    Operand G2 = op1(seg_bwd(gh2v[br], W, hA2v[br], W, W, b1, tr), O);
    RET_IF(linear_wgrad(G2, op1(seg_act(hA1v[br], H, 0, H, b0, 0, tr), O), br * 4 + 1, O, st));
    RET_IF(linear_dgrad(G2, br * 4 + 1, gh1v[br], H, O, hA1v[br], H, b0, true, nullptr, 0, tr, st));
  }
  RET_IF(end_group(st));
  begin_group();
  for (int br = 0; br < 2; ++br) {
    const int b0 = bn_head[br * 2];
    Operand G1 = op1(seg_bwd(gh1v[br], H, hA1v[br], H, H, b0, tr), O);
    RET_IF(linear_wgrad(G1, XL, br * 4 + 0, O, st));
    RET_IF(linear_dgrad(G1, br * 4 + 0, d_x[br], W, O, nullptr, 0, -1, false, nullptr, 0, tr, st));
  }
  RET_IF(end_group(st));
  RET_IF(join_side(st));
  {
    BnView v = view(ll.bn[3], 0, tr);
    RET_IF(sln_launch_mask_gstats(d_xb, W, d_xa, W, ll.A4, W, v, O, W, ll.g4, W,
                                  v.mode != SLN_BN_NONE ? bns[ll.bn[3]].gsums : nullptr, W, st));
  }
  for (int l = L - 1; l >= 0; --l) {
    const int gi = l, slot = l & 1;
    const float* dP = (l == L - 1) ? nullptr : dG[slot ^ 1];
    RET_IF(gconv_backward(gi, dP, 3 * W, W, slot, tr, st));
    if (l > 0) {
      const Layer& pv = layers[gi - 1];
      BnView v = view(pv.bn[3], 0, tr);
      RET_IF(sln_launch_gather_bwd(dG[slot], 3 * W, W, g, O, nullptr, 0, pv.A4, W, v, 1, pv.g4, W,
                                   v.mode != SLN_BN_NONE ? bns[pv.bn[3]].gsums : nullptr, W, st));
    } else {
      BnView none = view(-1, 0, tr);
      RET_IF(sln_launch_gather_bwd(dG[slot], 3 * W, W, g, O, nullptr, 0, nullptr, 0, none, 0, dX0, W, nullptr, 0, st));
      RET_IF(sln_launch_embed_bwd_i32(g.p, dG[slot], 3 * W, W, T, W, cfg.num_preds, t.d_pred_emb_ec, st));
    }
  }
  EncAssembleBwd eb; std::memset(&eb, 0, sizeof(eb));
  eb.objs = batch.objs; eb.attrs = batch.attributes; eb.angles = batch.angles; eb.boxes = batch.boxes; eb.dx0 = dX0;
  eb.O = O; eb.n_obj = n_obj_e; eb.n_attr = n_attr_e; eb.n_box = n_box_e; eb.n_angle = n_angle_e; eb.box_dim = cfg.box_dim;
  eb.d_obj_emb = t.d_obj_emb_ec; eb.d_attr_emb = t.d_attr_emb_ec; eb.d_angle_emb = t.d_angle_emb;
  eb.d_wb = t.d_box_emb_w; eb.d_bb = t.d_box_emb_b;
  eb.rows_obj = cfg.num_objs; eb.rows_attr = cfg.num_attrs; eb.rows_angle = cfg.n_angle;
  RET_IF(sln_launch_enc_assemble_bwd(eb, st));
  if (it_merge_bn && !bns.empty()) {            // both passes' BatchNorm parameter gradients: one launch
    int maxc = 0;
    for (auto& bi : bns) maxc = bi.C > maxc ? bi.C : maxc;
    RET_IF(sln_launch_bn_param_grads(bn_table_dev, (int)bns.size(), maxc, cfg.recurrent ? 0 : 1, st));
  } else if (n_bn_enc > 0) {
    int maxc = 0;
    for (int i = 0; i < n_bn_enc; ++i) maxc = bns[i].C > maxc ? bns[i].C : maxc;
    RET_IF(sln_launch_bn_param_grads(bn_table_dev, n_bn_enc, maxc, cfg.recurrent ? 0 : 1, st));
  }
  RET_IF(flush_deferred(it_one_flush ? 2 : 1, st));        // (region 2: the whole iteration's wgrads in one flush, see decoder_backward)
  return 0;
}

// mode: TRAIN_BACKWARD (zero_grad .. backward), TRAIN_FULL (+ Adam), or the same iteration in two halves for the
// data-parallel trainer: TRAIN_UPTO_DECODER ends when every decoder-side gradient is final (the all-reduce of that half of the
// flat buffer can start), TRAIN_ENCODER_BWD is the rest of the backward pass.
int SlnVae::train_iteration(const float* eps, int mode, hipStream_t st) {
  int r = 0;
  if (mode != TRAIN_ENCODER_BWD) {
    HIP_RET(hipMemsetAsync(t.flat_grads, 0, sizeof(float) * (size_t)t.n_flat, st));
    const bool no_merge = this->no_merge;                                         // SLN_NO_MERGE=1 (read at creation): the round-2 launch sequence
    it_prologue = it_fused_loss = !no_merge;
    // loss accumulators + every BatchNorm sum of the iteration: cleared by the prologue launch (the first kernel of the iteration,
    // nothing in front of it touches them) when the region can be written as 16-byte words, by a memset node otherwise
    it_zero_in_prologue = it_prologue && (reinterpret_cast<uintptr_t>(zero_begin) & 15) == 0 && (zero_bytes & 15) == 0;
    if (!it_zero_in_prologue) HIP_RET(hipMemsetAsync(zero_begin, 0, zero_bytes, st));
    bulk_zeroed = true;
    it_merge_bn = !no_merge && (mode == TRAIN_BACKWARD || mode == TRAIN_FULL);    // the two-half form hands the decoder's gradients out early
    { static const bool one = !(std::getenv("SLN_TN_ONE_FLUSH") && std::getenv("SLN_TN_ONE_FLUSH")[0] == '0'); it_one_flush = one && defer && !tn_per_layer && (mode == TRAIN_BACKWARD || mode == TRAIN_FULL); }
    if (!it_prologue && draw_eps) r = sln_launch_randn(eps_buf, (long)O * E, scalars, st);       // Sg2ScVAE_model.py:182
    if (!r) r = encoder_forward(step_training, st);
    if (!r) r = decoder_forward(nullptr, eps, step_training, st);
    if (!r) r = loss(boxes_pred, angles_pred, mu, logvar, true, st);
    // data-parallel guard: the total loss travels with the gradients (one more element of the all-reduced bucket); a
    // non-finite loss on ANY rank makes the reduced slot non-finite on EVERY rank, and sln_vae_adam_step skips on all of them
    if (!r && grad_guard && mode != TRAIN_FULL)
      r = (int)hipMemcpyAsync(grad_guard, losses + 3, sizeof(float), hipMemcpyDeviceToDevice, st);
    if (!r) r = decoder_backward(st);
    if (!r) r = sln_launch_latent_bwd(mu, logvar, eps, dz, &scalars->kl_weight, O, E, cfg.use_ae, dmu, dlv, st);
  }
  if (mode != TRAIN_UPTO_DECODER) {
    bulk_zeroed = true;                // zeroed by the first half of this iteration
    if (!r) r = encoder_backward(st);
  }
  bulk_zeroed = false;
  it_prologue = it_fused_loss = it_merge_bn = it_zero_in_prologue = it_one_flush = false;
  RET_IF(r);
  RET_IF(join_tn_side(st));            // the parameter gradients are complete behind this point (all-reduce, optimizer)
  if (mode == TRAIN_FULL) {
    RET_IF(sln_launch_adam(t.flat_params, t.flat_grads, t.adam_m, t.adam_v, (long)t.n_flat, scalars, losses + 3, st));
    wt_fresh = false;                  // transposed copies are rebuilt at the start of the next backward
  }
  return 0;
}

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int sln_version(void) { return 1; }
const char* sln_build_arch(void) { return "gfx950"; }
int sln_device_ok(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return SLN_E_NOGPU;
  hipDeviceProp_t p;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return SLN_E_NOGPU;
  return std::strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 0 : SLN_E_NOGPU;
}

static int cfg_check(const SlnVaeConfig* c) {
  if (!c) return SLN_E_BADARG;
  if (c->embedding_dim <= 0 || c->embedding_dim % 16 != 0) return SLN_E_UNSUPPORTED;
  if (c->gconv_num_layers < 1) return SLN_E_UNSUPPORTED;
  if (c->box_dim != 6 && c->box_dim != 4) return SLN_E_UNSUPPORTED;
  if (c->n_angle % 4 != 0 || c->n_angle <= 0) return SLN_E_UNSUPPORTED;
  return 0;
}

int sln_vae_num_units(const SlnVaeConfig* c) {
  if (cfg_check(c) != 0) return cfg_check(c);
  const int nmod = c->recurrent ? 1 : c->gconv_num_layers;
  return 8 + 2 * nmod * 4 + 4;
}

int sln_vae_create(const SlnVaeConfig* c, SlnVae** out) {
  if (!out) return SLN_E_BADARG;
  int r = cfg_check(c);
  if (r != 0) return r;
  SlnVae* h = new (std::nothrow) SlnVae();
  if (!h) return SLN_E_BADARG;
  h->cfg = *c;
  const int E = c->embedding_dim;
  h->E = E; h->H = 4 * E; h->L = c->gconv_num_layers; h->nmod = c->recurrent ? 1 : h->L;
  // Sg2ScVAE_model.py:21-37: without attributes the class embedding takes the whole E columns
  h->n_obj_e = c->no_attr ? E : E * 3 / 4; h->n_attr_e = c->no_attr ? 0 : E / 4; h->n_box_e = E * 3 / 4; h->n_angle_e = E / 4;
  h->Dec = 2 * E; h->Ddc = c->decoder_cat ? 2 * E : E;          // :79-88: the decoder's gconv net runs on E columns when z joins afterwards
  const int H = h->H, W = 2 * E, n = sln_vae_num_units(c);
  h->units.resize(n);
  auto set = [&](int i, int out_, int in_, bool bn) { h->units[i].out = out_; h->units[i].in = in_; h->units[i].bn = bn && c->batch_norm; };
  set(0, H, W, true); set(1, W, H, true); set(2, h->n_box_e, W, false); set(3, h->n_box_e, W, false);
  set(4, H, W, true); set(5, W, H, true); set(6, h->n_angle_e, W, false); set(7, h->n_angle_e, W, false);
  for (int net = 0; net < 2; ++net)
    for (int m = 0; m < h->nmod; ++m) {
      const int u = 8 + (net * h->nmod + m) * 4, D = net == 0 ? h->Dec : h->Ddc;
      set(u + 0, H, 3 * D, true); set(u + 1, 2 * H + D, H, true); set(u + 2, H, H, true); set(u + 3, D, H, true);
    }
  set(h->unit_boxnet(0), H, W + h->n_attr_e, true); set(h->unit_boxnet(1), c->box_dim, H, false);
  set(h->unit_anglenet(0), H, W, true); set(h->unit_anglenet(1), c->n_angle, H, false);
  // layers and BatchNorm applications, in execution order
  h->layers.resize(2 * h->L);
  for (int net = 0; net < 2; ++net) {
    for (int l = 0; l < h->L; ++l) {
      Layer& ly = h->layers[net * h->L + l];
      ly.net = net; ly.first = l == 0; ly.last = l == h->L - 1; ly.D = net == 0 ? h->Dec : h->Ddc; ly.Do = ly.D; ly.u0 = h->unit_of(net, l, 0);
      const int Cs[4] = {H, 2 * H + ly.D, H, ly.D};
      for (int k = 0; k < 4; ++k) {
        if (!h->units[ly.u0 + k].bn) continue;
        BnInst b; b.unit = ly.u0 + k; b.C = Cs[k]; b.rows = b.rows_code = k < 2 ? -1 : -2;   // -1: T rows, -2: O rows (set per batch)
        ly.bn[k] = (int)h->bns.size(); h->bns.push_back(b);
      }
    }
    auto head = [&](int slot, int unit, int C) {
      if (!h->units[unit].bn) return;
      BnInst b; b.unit = unit; b.C = C; b.rows = b.rows_code = -2;
      h->bn_head[slot] = (int)h->bns.size(); h->bns.push_back(b);
    };
    if (net == 0) { head(0, 0, H); head(1, 1, W); head(2, 4, H); head(3, 5, W); h->n_bn_enc = (int)h->bns.size(); }
    else { head(4, h->unit_boxnet(0), H); head(5, h->unit_anglenet(0), H); }
  }
  if (!c->batch_norm) h->n_bn_enc = 0;
  {
    const char* ns = std::getenv("SLN_NO_SIDE_STREAM");
    const char* nd = std::getenv("SLN_NO_DUAL");
    h->use_dual = !(nd && nd[0] == '1');
    const char* nf = std::getenv("SLN_NO_DEFER");
    h->defer = !(nf && nf[0] == '1');
    { const char* v = std::getenv("SLN_NO_MERGE"); h->no_merge = v && v[0] == '1'; }
    h->tn_slots = 3 * (2 * h->L + 8 > 16 ? 2 * h->L + 8 : 16);
    h->tn_groups_store = new (std::nothrow) SlnVae::TnGroup[2 * (size_t)h->tn_slots * 2];
    if (!h->tn_groups_store) { delete h; return SLN_E_BADARG; }
    { const char* v = std::getenv("SLN_TN_PER_LAYER"); h->tn_per_layer = v && v[0] == '1'; }
    { const char* v = std::getenv("SLN_TN_SIDE"); h->tn_side = h->defer && v && v[0] == '1'; }
    if (h->tn_side && !h->side && sln_side_stream_create(&h->side) != hipSuccess) { h->side = nullptr; h->tn_side = false; }
    const char* ng = std::getenv("SLN_NO_GROUP");
    h->use_group = h->use_dual && !(ng && ng[0] == '1');
    h->use_side = !h->use_dual && !(ns && ns[0] == '1');
    if (h->use_side && sln_side_stream_create(&h->side) != hipSuccess) { h->side = nullptr; h->use_side = false; }
  }
  *out = h;
  return 0;
}

void sln_vae_destroy(SlnVae* h) {
  if (!h) return;
  h->drop_graphs();
  for (auto e : h->events) (void)hipEventDestroy(e);
  if (h->side) (void)hipStreamDestroy(h->side);
  delete[] h->tn_groups_store;
  for (auto& sl : h->tn_stage) {
    if (sl.done) { (void)hipEventSynchronize(sl.done); (void)hipEventDestroy(sl.done); }
    if (sl.host) (void)hipHostFree(sl.host);
  }
  delete h;
}

int64_t sln_vae_workspace_bytes(const SlnVae* h, int max_objs, int max_triples) {
  if (!h || max_objs <= 0 || max_triples < 0) return SLN_E_BADARG;
  SlnVae tmp = *h;                 // carve() on a copy in dry-run mode
  for (int i = 0; i < 4; ++i) tmp.graph_exec[i] = nullptr;
  return (int64_t)tmp.carve(nullptr, max_objs, max_triples < 1 ? 1 : max_triples);
}

int sln_vae_bind(SlnVae* h, const SlnVaeTensors* t, void* workspace, int64_t workspace_bytes, int max_objs, int max_triples) {
  if (!h || !t || !workspace || !t->units_host) return SLN_E_BADARG;
  if (max_triples < 1) max_triples = 1;
  const int64_t need = sln_vae_workspace_bytes(h, max_objs, max_triples);
  if (need < 0 || workspace_bytes < need) return SLN_E_BADARG;
  h->t = *t;
  for (size_t i = 0; i < h->units.size(); ++i) {
    h->units[i].p = t->units_host[i];
    if (!h->units[i].p.weight || !h->units[i].p.bias) return SLN_E_BADARG;
    if (h->units[i].bn && (!h->units[i].p.bn_weight || !h->units[i].p.bn_bias || !h->units[i].p.bn_running_mean ||
                           !h->units[i].p.bn_running_var)) return SLN_E_BADARG;
  }
  h->carve(workspace, max_objs, max_triples);
  h->maxO = max_objs; h->maxT = max_triples;
  // transposition table (weights needed by dgrad)
  std::vector<TransposeEntry> tr;
  int maxt = 0;
  for (auto& u : h->units) {
    TransposeEntry e; e.src = u.p.weight; e.dst = u.wt; e.rows = u.out; e.cols = u.in; e.dst_ld = u.wt_ld; e.pad_ = 0;
    tr.push_back(e);
    const int tiles = sln_cdiv(u.out, 32) * sln_cdiv(u.in, 32);
    maxt = tiles > maxt ? tiles : maxt;
  }
  h->n_tr = (int)tr.size(); h->tr_max_tiles = maxt;
  HIP_RET(hipMemcpy(h->tr_table_dev, tr.data(), sizeof(TransposeEntry) * tr.size(), hipMemcpyHostToDevice));
  AdamScalars sc; std::memset(&sc, 0, sizeof(sc));
  sc.step = 0; sc.lr = 1e-4f; sc.beta1 = 0.9f; sc.beta2 = 0.999f; sc.eps = 1e-8f; sc.kl_weight = 0.1f; sc.bc1 = 1.f; sc.bc2 = 1.f;
  sc.rng_seed = h->host_seed[0]; sc.rng_offset = h->host_seed[1];
  HIP_RET(hipMemcpy(h->scalars, &sc, sizeof(sc), hipMemcpyHostToDevice));
  RET_IF(sln_gemm_init());
  h->host_scalars_valid = false;
  h->bound = true; h->batch_set = false; h->wt_fresh = false; h->have_enc = h->have_dec = false; h->bn_table_live = false;
  h->drop_graphs();
  return 0;
}

static int upload_bn_table(SlnVae* h) {
  std::vector<BnTableEntry> tab(h->bns.size());
  for (size_t i = 0; i < h->bns.size(); ++i) {
    BnInst& b = h->bns[i];
    const Unit& u = h->units[b.unit];
    BnTableEntry e; std::memset(&e, 0, sizeof(e));
    e.sums = b.sums; e.gsums = b.gsums; e.cstride = b.C; e.C = b.C; e.rows = b.rows_code;     // the kernel gets (T, O) with its launch
    e.rmean = u.p.bn_running_mean; e.rvar = u.p.bn_running_var; e.nbt = u.p.bn_num_batches_tracked;
    e.dgamma = u.p.d_bn_weight; e.dbeta = u.p.d_bn_bias;
    tab[i] = e;
  }
  if (!tab.empty()) HIP_RET(hipMemcpy(h->bn_table_dev, tab.data(), sizeof(BnTableEntry) * tab.size(), hipMemcpyHostToDevice));
  return 0;
}

int sln_vae_set_batch(SlnVae* h, const SlnVaeBatch* b, void* stream) {
  if (!h || !b || !h->bound) return SLN_E_BADARG;
  if (b->O <= 0 || b->T < 0 || b->O > h->maxO || b->T > h->maxT) return SLN_E_BADARG;
  if (!b->objs || !b->boxes || !b->angles || !b->attributes || (b->T > 0 && !b->triples)) return SLN_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const bool shape_changed = (b->O != h->O) || (b->T != h->T) || !h->batch_set;
  h->batch = *b; h->O = b->O; h->T = b->T;
  h->g.T = b->T; h->g.O = b->O;
  if (shape_changed) {
    // BatchNorm row counts: instances over triples (net1) use T rows, the others O rows
    for (auto& ly : h->layers)
      for (int k = 0; k < 4; ++k)
        if (ly.bn[k] >= 0) h->bns[ly.bn[k]].rows = k < 2 ? b->T : b->O;
    for (int s = 0; s < 6; ++s)
      if (h->bn_head[s] >= 0) h->bns[h->bn_head[s]].rows = b->O;
    // (the device table holds row-count CODES, the running-statistics kernel gets (T, O) with its launch: uploaded once per binding)
    if (!h->bn_table_live) {
      HIP_RET(hipStreamSynchronize(st));
      RET_IF(upload_bn_table(h));
      h->bn_table_live = true;
    }
    bool has_graph = false;
    for (int i = 0; i < 4; ++i) has_graph |= h->graph_exec[i] != nullptr;
    if (has_graph) HIP_RET(hipStreamSynchronize(st));        // a replay may be in flight: finish it before its hipGraphExec goes away
    h->drop_graphs();
  }
  // The kernels of an iteration read the batch through h->batch.  A captured iteration (hipGraph) has those addresses baked in,
  // and the caller's tensors are new ones for every batch (DataLoader, synthetic generator): stage the inputs in engine-owned
  // buffers whose addresses never change, so that a replay sees the CURRENT batch (it used to read the tensors of the batch it was
  // captured with whenever two consecutive batches had the same shape).  One small kernel outside the graph (stage_batch_kernel).
  RET_IF(sln_zero_async(h->err_flag, sizeof(int), st));
  StageBatch sb; std::memset(&sb, 0, sizeof(sb));
  sb.objs = b->objs; sb.attrs = b->attributes; sb.angles = b->angles; sb.boxes = b->boxes;
  sb.st_objs = h->st_objs; sb.st_attrs = h->st_attrs; sb.st_angles = h->st_angles; sb.st_boxes = h->st_boxes;
  sb.attrs32 = h->attrs32; sb.deg = h->g.deg; sb.err = h->err_flag;
  sb.O = b->O; sb.box_dim = h->cfg.box_dim; sb.n_objs = h->cfg.num_objs; sb.n_attrs = h->cfg.num_attrs; sb.n_angle = h->cfg.n_angle;
  RET_IF(sln_launch_stage_batch(sb, st));        // copies + id checks + int32 attributes + cleared degree counters: one launch
  h->batch.objs = h->st_objs; h->batch.angles = h->st_angles; h->batch.attributes = h->st_attrs; h->batch.boxes = h->st_boxes;
  RET_IF(sln_launch_graph_prep(b->triples, b->T, b->O, h->cfg.num_preds, h->g, h->err_flag, st, 0, 1));
  h->batch_set = true; h->have_enc = h->have_dec = false;
  return 0;
}

// Blocking check of the ids of the bound batch (the reference's embedding / index ops raise IndexError for these).
int sln_vae_check_batch(SlnVae* h, void* stream) {
  if (!h || !h->batch_set) return SLN_E_STATE;
  int flag = 0;
  HIP_RET(hipMemcpyAsync(&flag, h->err_flag, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_RET(hipStreamSynchronize((hipStream_t)stream));
  return flag ? SLN_E_BADARG : 0;
}

static int copy_out(float* dst, const float* src, size_t n, hipStream_t st) {
  if (!dst || dst == src) return 0;
  return (int)hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st);
}

int sln_vae_encoder(SlnVae* h, float* mu, float* logvar, int training, void* stream) {
  if (!h || !h->batch_set) return SLN_E_STATE;
  hipStream_t st = (hipStream_t)stream;
  RET_IF(h->encoder_forward(training != 0, st));
  RET_IF(copy_out(mu, h->mu, (size_t)h->O * h->E, st));
  RET_IF(copy_out(logvar, h->logvar, (size_t)h->O * h->E, st));
  return 0;
}

int sln_vae_decoder(SlnVae* h, const float* z, float* boxes_pred, float* angles_pred, int training, void* stream) {
  if (!h || !h->batch_set) return SLN_E_STATE;
  if (!z) return SLN_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  RET_IF(h->decoder_forward(z, nullptr, training != 0, st));
  RET_IF(copy_out(boxes_pred, h->boxes_pred, (size_t)h->O * h->cfg.box_dim, st));
  RET_IF(copy_out(angles_pred, h->angles_pred, (size_t)h->O * h->cfg.n_angle, st));
  return 0;
}

int sln_vae_forward(SlnVae* h, const float* eps, float* mu, float* logvar, float* z_out, float* boxes_pred,
                    float* angles_pred, int training, void* stream) {
  if (!h || !h->batch_set) return SLN_E_STATE;
  hipStream_t st = (hipStream_t)stream;
  if (eps) RET_IF(copy_out(h->eps_buf, eps, (size_t)h->O * h->E, st));
  else if (!h->cfg.use_ae) RET_IF(sln_launch_randn(h->eps_buf, (long)h->O * h->E, h->scalars, st));    // Sg2ScVAE_model.py:182
  RET_IF(h->encoder_forward(training != 0, st));
  RET_IF(h->decoder_forward(nullptr, h->eps_buf, training != 0, st));
  RET_IF(copy_out(mu, h->mu, (size_t)h->O * h->E, st));
  RET_IF(copy_out(logvar, h->logvar, (size_t)h->O * h->E, st));
  RET_IF(copy_out(z_out, h->z, (size_t)h->O * h->E, st));
  RET_IF(copy_out(boxes_pred, h->boxes_pred, (size_t)h->O * h->cfg.box_dim, st));
  RET_IF(copy_out(angles_pred, h->angles_pred, (size_t)h->O * h->cfg.n_angle, st));
  return 0;
}

static int set_kl(SlnVae* h, float kl_weight, float lr, hipStream_t st) {
  // two floats inside the device scalar block (kept out of kernel arguments so a captured graph sees updates)
  if (kl_weight != h->host_kl || !h->host_scalars_valid) {
    h->host_kl = kl_weight;
    HIP_RET(hipMemcpyAsync(&h->scalars->kl_weight, &h->host_kl, sizeof(float), hipMemcpyHostToDevice, st));
  }
  if (lr > 0.f && (lr != h->host_lr || !h->host_scalars_valid)) {
    h->host_lr = lr;
    HIP_RET(hipMemcpyAsync(&h->scalars->lr, &h->host_lr, sizeof(float), hipMemcpyHostToDevice, st));
  }
  h->host_scalars_valid = true;
  return 0;
}

int sln_vae_loss(SlnVae* h, const float* boxes_pred, const float* angles_pred, const float* mu, const float* logvar,
                 float kl_weight, float* losses_out, int with_grads, void* stream) {
  if (!h || !h->batch_set) return SLN_E_STATE;
  hipStream_t st = (hipStream_t)stream;
  RET_IF(set_kl(h, kl_weight, -1.f, st));
  RET_IF(h->loss(boxes_pred ? boxes_pred : h->boxes_pred, angles_pred ? angles_pred : h->angles_pred,
                 mu ? mu : h->mu, logvar ? logvar : h->logvar, with_grads != 0, st));
  // the guard slot of the optimizer step (sln_vae_set_grad_guard) follows the loss of THIS path too: only train_iteration wrote
  // it, so a forward() / loss / backward() / adam_step() sequence kept whatever an earlier data-parallel iteration had left
  // there - a stale NaN skipped every later update
  if (with_grads && h->grad_guard) HIP_RET(hipMemcpyAsync(h->grad_guard, h->losses + 3, sizeof(float), hipMemcpyDeviceToDevice, st));
  RET_IF(copy_out(losses_out, h->losses, 4, st));
  return 0;
}

int sln_vae_decoder_backward(SlnVae* h, const float* d_boxes_pred, const float* d_angles_pred, float* dz, void* stream) {
  if (!h || !h->have_dec) return SLN_E_STATE;
  hipStream_t st = (hipStream_t)stream;
  // gradients from the caller's own loss (torch autograd): the engine has no loss value to guard the update with - the caller
  // checks it, as train.py:79-81 does - so the slot must not keep an older iteration's value
  if ((d_boxes_pred || d_angles_pred) && h->grad_guard) RET_IF(sln_zero_async(h->grad_guard, sizeof(float), st));
  if (d_boxes_pred)
    HIP_RET(hipMemcpy2DAsync(h->dbp, sizeof(float) * h->dbp_ld, d_boxes_pred, sizeof(float) * h->cfg.box_dim,
                             sizeof(float) * h->cfg.box_dim, (size_t)h->O, hipMemcpyDeviceToDevice, st));
  else RET_IF(sln_zero_async(h->dbp, sizeof(float) * (size_t)h->O * h->dbp_ld, st));
  if (d_angles_pred) RET_IF(sln_launch_log_softmax_bwd(h->angles_pred, d_angles_pred, h->dlogits, h->O, h->cfg.n_angle, st));
  else RET_IF(sln_zero_async(h->dlogits, sizeof(float) * (size_t)h->O * h->cfg.n_angle, st));
  RET_IF(h->decoder_backward(st));
  RET_IF(h->join_tn_side(st));
  RET_IF(copy_out(dz, h->dz, (size_t)h->O * h->E, st));
  return 0;
}

int sln_vae_encoder_backward(SlnVae* h, const float* d_mu, const float* d_logvar, void* stream) {
  if (!h || !h->have_enc) return SLN_E_STATE;
  hipStream_t st = (hipStream_t)stream;
  const size_t n = (size_t)h->O * h->E;
  if (d_mu) RET_IF(copy_out(h->dmu, d_mu, n, st)); else RET_IF(sln_zero_async(h->dmu, n * sizeof(float), st));
  if (d_logvar) RET_IF(copy_out(h->dlv, d_logvar, n, st)); else RET_IF(sln_zero_async(h->dlv, n * sizeof(float), st));
  RET_IF(h->encoder_backward(st));
  RET_IF(h->join_tn_side(st));
  return 0;
}

int sln_vae_backward(SlnVae* h, void* stream) {
  if (!h || !h->have_enc || !h->have_dec || !h->have_loss_grads || !h->z_from_latent) return SLN_E_STATE;
  hipStream_t st = (hipStream_t)stream;
  RET_IF(h->decoder_backward(st));
  RET_IF(sln_launch_latent_bwd(h->mu, h->logvar, h->eps_buf, h->dz, &h->scalars->kl_weight, h->O, h->E, h->cfg.use_ae,
                               h->dmu, h->dlv, st));
  RET_IF(h->encoder_backward(st));
  RET_IF(h->join_tn_side(st));
  return 0;
}

int sln_vae_zero_grad(SlnVae* h, void* stream) {
  if (!h || !h->bound) return SLN_E_STATE;
  return sln_zero_async(h->t.flat_grads, sizeof(float) * (size_t)h->t.n_flat, (hipStream_t)stream);
}

int sln_vae_adam_step(SlnVae* h, float lr, void* stream) {
  if (!h || !h->bound || !h->t.adam_m || !h->t.adam_v) return SLN_E_STATE;
  hipStream_t st = (hipStream_t)stream;
  RET_IF(set_kl(h, h->host_scalars_valid ? h->host_kl : 0.1f, lr, st));
  // guard of the update: the slot registered with sln_vae_set_grad_guard (after the trainer's all-reduce it holds the
  // rank-averaged total loss: every rank skips, or none), else the loss of the last iteration on this rank
  RET_IF(sln_launch_adam(h->t.flat_params, h->t.flat_grads, h->t.adam_m, h->t.adam_v, (long)h->t.n_flat, h->scalars,
                         h->grad_guard ? h->grad_guard : h->losses + 3, st));
  h->wt_fresh = false;
  return 0;
}

int sln_vae_adam_reset(SlnVae* h, int64_t step, void* stream) {
  if (!h || !h->bound) return SLN_E_STATE;
  h->host_step = step;
  return (int)hipMemcpyAsync(&h->scalars->step, &h->host_step, sizeof(int64_t), hipMemcpyHostToDevice, (hipStream_t)stream);
}

int sln_vae_adam_get_step(SlnVae* h, int64_t* step_out, void* stream) {
  if (!h || !h->bound || !step_out) return SLN_E_STATE;
  HIP_RET(hipMemcpyAsync(step_out, &h->scalars->step, sizeof(int64_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_RET(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int sln_vae_set_grad_guard(SlnVae* h, float* slot) {
  if (!h) return SLN_E_BADARG;
  if (h->grad_guard != slot) { h->grad_guard = slot; h->drop_graphs(); }
  return 0;
}

int sln_vae_seed(SlnVae* h, uint64_t seed, uint64_t offset, void* stream) {
  if (!h || !h->bound) return SLN_E_STATE;
  h->host_seed[0] = seed; h->host_seed[1] = offset;
  HIP_RET(hipMemcpyAsync(&h->scalars->rng_seed, h->host_seed, 2 * sizeof(unsigned long long), hipMemcpyHostToDevice, (hipStream_t)stream));
  return 0;
}

int sln_vae_randn(SlnVae* h, float* out, int64_t n, void* stream) {
  if (!h || !h->bound || !out || n < 0) return SLN_E_BADARG;
  if (n == 0) return 0;
  return sln_launch_randn(out, (long)n, h->scalars, (hipStream_t)stream);
}

int sln_layout_heatmap(const float* boxes_pred, int64_t n_trials, int O, int box_dim, int container_size, int clip_coor, float* counts,
                       void* stream) {
  if (!boxes_pred || !counts || n_trials < 0 || O < 2 || container_size < 2) return SLN_E_BADARG;
  if (box_dim != 6) return SLN_E_UNSUPPORTED;
  return sln_launch_layout_heatmap(boxes_pred, (long)n_trials, O, container_size, clip_coor, counts, (hipStream_t)stream);
}

int sln_vae_last_eps(SlnVae* h, float* eps_out, void* stream) {
  if (!h || !h->batch_set || !eps_out) return SLN_E_STATE;
  return copy_out(eps_out, h->eps_buf, (size_t)h->O * h->E, (hipStream_t)stream);
}

int sln_vae_set_training(SlnVae* h, int training) {
  if (!h) return SLN_E_BADARG;
  if (h->step_training != (training != 0)) { h->step_training = training != 0; h->drop_graphs(); }
  return 0;
}

int sln_vae_params_changed(SlnVae* h) {       // parameters were modified outside the engine (load_state_dict, SGD)
  if (!h) return SLN_E_BADARG;
  h->wt_fresh = false;
  return 0;
}

int sln_vae_train_step(SlnVae* h, const float* eps, float kl_weight, float lr, float* losses_out, int use_graph,
                       int with_adam, void* stream) {
  if (!h || !h->batch_set || !h->t.adam_m || !h->t.adam_v) return SLN_E_STATE;
  const int mode = with_adam;
  if (mode < 0 || mode > 3) return SLN_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if (mode == SlnVae::TRAIN_ENCODER_BWD) {
    if (!(h->have_dec && h->dec_training == h->step_training && h->z_from_latent)) return SLN_E_STATE;       // needs the first half
  } else {
    RET_IF(set_kl(h, kl_weight, lr, st));
    if (eps) RET_IF(copy_out(h->eps_buf, eps, (size_t)h->O * h->E, st));
    const bool draw = !eps && !h->cfg.use_ae;            // no eps from the caller: the iteration draws it on the device
    if (draw != h->draw_eps) { HIP_RET(hipStreamSynchronize(st)); h->draw_eps = draw; h->drop_graphs(); }
  }
  if (h->det_seen != g_sln_deterministic) { h->det_seen = g_sln_deterministic; h->drop_graphs(); }   // other launch sequence: re-capture
  if (use_graph && st != nullptr) {
    hipGraphExec_t& ge = h->graph_exec[mode];
    if (!ge || h->graph_O[mode] != h->O || h->graph_T[mode] != h->T) {
      if (ge) { (void)hipGraphExecDestroy(ge); ge = nullptr; }
      HIP_RET(hipStreamSynchronize(st));
      // the captured iteration always rebuilds the transposed weights itself (the second half relies on the first one's)
      if (mode != SlnVae::TRAIN_ENCODER_BWD) h->wt_fresh = false;
      hipGraph_t graph = nullptr;
      HIP_RET(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      h->capturing = true;
      const int r = h->train_iteration(h->eps_buf, mode, st);
      h->capturing = false;
      hipError_t e = hipStreamEndCapture(st, &graph);
      if (r != 0) { if (graph) (void)hipGraphDestroy(graph); return r; }
      if (e != hipSuccess) return (int)e;
      if (h->tn_upload_pending) RET_IF(h->upload_pending_tables());     // the wgrad problem tables the captured launches read
      e = hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (e != hipSuccess) { ge = nullptr; return (int)e; }
      h->graph_O[mode] = h->O; h->graph_T[mode] = h->T;
    }
    HIP_RET(hipGraphLaunch(ge, st));
    h->enc_training = h->dec_training = h->step_training; h->have_enc = h->have_dec = true; h->z_from_latent = true;
    h->wt_fresh = mode == SlnVae::TRAIN_UPTO_DECODER || mode == SlnVae::TRAIN_ENCODER_BWD;   // rebuilt by the graph, no Adam yet
  } else {
    RET_IF(h->train_iteration(h->eps_buf, mode, st));
  }
  if (mode == SlnVae::TRAIN_ENCODER_BWD) return 0;       // the losses were handed out by the first half
  RET_IF(copy_out(losses_out, h->losses, 4, st));
  return 0;
}

// ---------------------------------------------------------------------------------------------
// A bare GraphTripleConvNet with autograd (models/graph.py:57-143 used on its own): the engine's layer kernels without the
// VAE around them.  Same handle type; destroy / workspace_bytes / bind are the sln_vae_* ones (SlnVaeTensors carries only
// units_host: 4 units per module - net1.0, net1.1, net2.0, net2.1 - with weights AND gradient pointers).
// ---------------------------------------------------------------------------------------------
int sln_gconv_net_create(int D, int H, int Dout, int num_layers, int recurrent, int batch_norm, SlnVae** out) {
  if (!out || D <= 0 || H <= 0 || Dout <= 0 || num_layers < 1 || D % 4 || H % 4 || Dout % 4) return SLN_E_UNSUPPORTED;
  if (Dout != D && num_layers != 1) return SLN_E_UNSUPPORTED;       // a stack feeds its output back in (models/graph.py:121-131)
  SlnVae* h = new (std::nothrow) SlnVae();
  if (!h) return SLN_E_BADARG;
  std::memset(&h->cfg, 0, sizeof(h->cfg));
  h->cfg.gconv_num_layers = num_layers; h->cfg.recurrent = recurrent; h->cfg.batch_norm = batch_norm; h->cfg.decoder_cat = 1;
  h->cfg.box_dim = 4; h->cfg.n_angle = 4; h->cfg.num_preds = 1 << 30;
  h->gconv_only = true;
  h->E = 0; h->H = H; h->L = num_layers; h->nmod = recurrent ? 1 : num_layers;
  h->n_obj_e = h->n_attr_e = h->n_box_e = h->n_angle_e = 0;
  h->Dec = D; h->Ddc = D;
  h->units.resize((size_t)h->nmod * 4);
  for (int m = 0; m < h->nmod; ++m) {
    Unit* u = &h->units[(size_t)m * 4];
    u[0].out = H; u[0].in = 3 * D; u[1].out = 2 * H + Dout; u[1].in = H; u[2].out = H; u[2].in = H; u[3].out = Dout; u[3].in = H;
    for (int k = 0; k < 4; ++k) u[k].bn = batch_norm != 0;
  }
  h->layers.resize(num_layers);
  for (int l = 0; l < num_layers; ++l) {
    Layer& ly = h->layers[l];
    ly.net = 0; ly.first = l == 0; ly.last = l == num_layers - 1; ly.D = D; ly.Do = Dout; ly.u0 = h->unit_of(0, l, 0);
    const int Cs[4] = {H, 2 * H + Dout, H, Dout};
    for (int k = 0; k < 4; ++k) {
      if (!h->units[ly.u0 + k].bn) continue;
      BnInst b; b.unit = ly.u0 + k; b.C = Cs[k]; b.rows = b.rows_code = k < 2 ? -1 : -2;
      ly.bn[k] = (int)h->bns.size(); h->bns.push_back(b);
    }
  }
  h->n_bn_enc = (int)h->bns.size();
  h->use_dual = true; h->use_group = false; h->use_side = false;
  {
    const char* nf = std::getenv("SLN_NO_DEFER");
    h->defer = !(nf && nf[0] == '1');
    { const char* v = std::getenv("SLN_NO_MERGE"); h->no_merge = v && v[0] == '1'; }
    h->tn_slots = 3 * (2 * h->L + 8 > 16 ? 2 * h->L + 8 : 16);
    h->tn_groups_store = new (std::nothrow) SlnVae::TnGroup[2 * (size_t)h->tn_slots * 2];
    if (!h->tn_groups_store) { delete h; return SLN_E_BADARG; }
    { const char* v = std::getenv("SLN_TN_PER_LAYER"); h->tn_per_layer = v && v[0] == '1'; }
    { const char* v = std::getenv("SLN_TN_SIDE"); h->tn_side = h->defer && v && v[0] == '1'; }
    if (h->tn_side && !h->side && sln_side_stream_create(&h->side) != hipSuccess) { h->side = nullptr; h->tn_side = false; }
  }
  *out = h;
  return 0;
}

// edges [T,2] int64 (s, o) of the graph the next forward / backward run on
int sln_gconv_net_set_edges(SlnVae* h, const int64_t* edges, int O, int T, void* stream) {
  if (!h || !h->bound || !h->gconv_only) return SLN_E_STATE;
  if (O <= 0 || T < 0 || O > h->maxO || T > h->maxT || (T > 0 && !edges)) return SLN_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const bool shape_changed = O != h->O || T != h->T || !h->batch_set;
  h->O = O; h->T = T; h->g.T = T; h->g.O = O;
  if (shape_changed) {
    for (auto& ly : h->layers)
      for (int k = 0; k < 4; ++k)
        if (ly.bn[k] >= 0) h->bns[ly.bn[k]].rows = k < 2 ? T : O;
    // (the device table holds row-count codes, see sln_vae_set_batch: uploaded once per binding)
    if (!h->bn_table_live) {
      HIP_RET(hipStreamSynchronize(st));
      RET_IF(upload_bn_table(h));
      h->bn_table_live = true;
    }
  }
  RET_IF(sln_zero_async(h->err_flag, sizeof(int), st));
  RET_IF(sln_launch_graph_prep(edges, T, O, 1 << 30, h->g, h->err_flag, st, 1, 0));
  h->batch_set = true; h->have_enc = false;
  return 0;
}

// (new_obj [O,Dout], new_pred [T,Dout]) = GraphTripleConvNet(obj_vecs [O,D], pred_vecs [T,D], edges); pre-activations stay in the
// workspace for sln_gconv_net_backward.  training: batch statistics + running-statistics update.
int sln_gconv_net_forward(SlnVae* h, const float* obj_vecs, const float* pred_vecs, float* new_obj, float* new_pred, int training,
                          void* stream) {
  if (!h || !h->gconv_only || !h->batch_set) return SLN_E_STATE;
  if (!obj_vecs || !pred_vecs || !new_obj || !new_pred) return SLN_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const int D = h->Dec, H = h->H, L = h->L;
  const bool tr = training != 0;
  if (tr && h->stats_doubles) RET_IF(sln_zero_async(h->stats_base, h->stats_doubles * sizeof(double), st));
  RET_IF(copy_out(h->X0e, obj_vecs, (size_t)h->O * D, st));
  RET_IF(copy_out(h->P0e, pred_vecs, (size_t)h->T * D, st));
  for (int l = 0; l < L; ++l) RET_IF(h->gconv_forward(l, tr, st));
  const Layer& ll = h->layers[L - 1];
  const int Do = ll.Do;
  RET_IF(sln_launch_bn_relu_apply(ll.A4, Do, 0, Do, h->O, h->view(ll.bn[3], 0, tr), new_obj, Do, st));
  RET_IF(sln_launch_bn_relu_apply(ll.A2, 2 * H + Do, H, Do, h->T, h->view(ll.bn[1], H, tr), new_pred, Do, st));
  if (tr) RET_IF(h->run_bn_updates(0, (int)h->bns.size(), st));
  h->enc_training = tr; h->have_enc = true;
  return 0;
}

// Backward of the last sln_gconv_net_forward (d_new_obj [O,Dout], d_new_pred [T,Dout]): d_obj_vecs [O,D], d_pred_vecs [T,D]
// (either may be NULL), parameter gradients
// accumulated (+=) into the d_* pointers of the bound units.
int sln_gconv_net_backward(SlnVae* h, const float* d_new_obj, const float* d_new_pred, float* d_obj_vecs, float* d_pred_vecs,
                           void* stream) {
  if (!h || !h->gconv_only || !h->have_enc) return SLN_E_STATE;
  if (!d_new_obj || !d_new_pred) return SLN_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  const int D = h->Dec, L = h->L;
  const bool tr = h->enc_training;
  h->ev_next = 0;
  h->tn_slot_next[0] = h->tn_slot_next[1] = h->tn_slot_next[2] = 0;
  if (h->stats_doubles) RET_IF(sln_zero_async(h->gstats_base, h->stats_doubles * sizeof(double), st));
  h->wt_fresh = false;                      // the caller's optimizer owns the parameters: rebuild W^T every backward
  RET_IF(h->refresh_transposes(st));
  {
    const Layer& ll = h->layers[L - 1];
    BnView v = h->view(ll.bn[3], 0, tr);
    const int Do = ll.Do;
    RET_IF(sln_launch_mask_gstats(d_new_obj, Do, nullptr, 0, ll.A4, Do, v, h->O, Do, ll.g4, Do,
                                  v.mode != SLN_BN_NONE ? h->bns[ll.bn[3]].gsums : nullptr, Do, st));
  }
  for (int l = L - 1; l >= 0; --l) {
    const int slot = l & 1;
    const float* dP = (l == L - 1) ? d_new_pred : h->dG[slot ^ 1];
    RET_IF(h->gconv_backward(l, dP, l == L - 1 ? h->layers[l].Do : 3 * D, l == L - 1 ? 0 : D, slot, tr, st));
    if (l > 0) {
      const Layer& pv = h->layers[l - 1];
      BnView v = h->view(pv.bn[3], 0, tr);
      RET_IF(sln_launch_gather_bwd(h->dG[slot], 3 * D, D, h->g, h->O, nullptr, 0, pv.A4, D, v, 1, pv.g4, D,
                                   v.mode != SLN_BN_NONE ? h->bns[pv.bn[3]].gsums : nullptr, D, st));
    } else {
      BnView none = h->view(-1, 0, tr);
      RET_IF(sln_launch_gather_bwd(h->dG[slot], 3 * D, D, h->g, h->O, nullptr, 0, nullptr, 0, none, 0, h->dX0, D, nullptr, 0, st));
      if (d_obj_vecs) RET_IF(copy_out(d_obj_vecs, h->dX0, (size_t)h->O * D, st));
      if (d_pred_vecs && h->T > 0)
        HIP_RET(hipMemcpy2DAsync(d_pred_vecs, sizeof(float) * D, h->dG[slot] + D, sizeof(float) * 3 * D, sizeof(float) * D, (size_t)h->T,
                                 hipMemcpyDeviceToDevice, st));
    }
  }
  if (!h->bns.empty()) {
    int maxc = 0;
    for (auto& b : h->bns) maxc = b.C > maxc ? b.C : maxc;
    RET_IF(sln_launch_bn_param_grads(h->bn_table_dev, (int)h->bns.size(), maxc, h->cfg.recurrent ? 0 : 1, st));
  }
  RET_IF(h->flush_deferred(1, st));
  RET_IF(h->join_tn_side(st));
  return 0;
}

// ---------------------------------------------------------------------------------------------
// R rooms in flight (round 5): the decoder forward / backward of R engines - R rooms, R parameter copies, eval-mode BatchNorm
// (testing/test_render_refine.py:250-263,279-359: every trial reloads the checkpoint and steps its own copy) - as ONE launch
// sequence.  At creation every engine RECORDS its two passes (SlnVae::rec); step s of all rooms becomes one multi-room launch
// whose per-room argument blocks sit in device memory (vae_multi.h).  Nothing depends on the iteration: the program is built and
// uploaded once and then only replayed - by plain launches, or from inside a caller's hipGraph capture.
// A room's arithmetic is the single-room kernels' own (same bodies, same dispatch rules, one room per blockIdx.z slice): results
// do not depend on how many rooms share the launches.
// ---------------------------------------------------------------------------------------------
}  // extern "C"

struct SlnVaeGroup {
  std::vector<SlnVae*> eng;
  int R = 0, rows_total = 0, n_angle = 0;
  SlnVaeGroupIO io;
  float* logits = nullptr; float* dlogits = nullptr;
  std::vector<void*> allocs;
  enum { L_NT = 100, L_NT_SINGLE, L_TN_MULTI, L_TN_SINGLE, L_SINGLE_STEP, L_ZERO, L_TRANSPOSE, L_BN_GRADS, L_LOG_SOFTMAX, L_LOG_SOFTMAX_BWD,
         L_FORK, L_JOIN, L_JOIN_TR };
  struct Launch {
    int kind = -1, variant = 0, count = 0, gx = 0, gy = 0, smem_floats = 0, maxK = 0; double flops = 0.0;
    const void* tab = nullptr; const int* tiles = nullptr;
    const GemmTNArgs* tn_probs = nullptr; const TnMultiMeta* tn_meta = nullptr; int blocks = 0; bool x2 = false, xg = false;
    long max_n16 = 0; int single = -1;
    bool on_side = false;
  };
  std::vector<Launch> fwd, bwd;
  // Side stream: the decoder chain is ~80 dependent launches of a few microseconds each - the chip is mostly waiting for the next
  // launch - so work that nothing in the chain waits for runs next to it: W^T of the decoder's weights (needed by the first
  // dgrad only) under the forward pass and everything between the two calls (render, loss), a layer's wgrads under the following
  // layers' dgrads.  Fork / join are events (legal inside a caller's stream capture: the side stream joins the capture).
  hipStream_t side = nullptr;                      // a pooled stream that overlaps with the caller's (sln_overlapping_stream), per run()
  hipEvent_t ev_fork = nullptr, ev_tr = nullptr, ev_join = nullptr;
  bool tr_pending = false, use_side = true;
  // W^T of the decoder's weights is built by the forward pass (side stream) and consumed by the backward pass's dgrads.  wt_valid says
  // that the copy in the engines' workspaces matches the parameters: set by a forward's transposition, cleared when a backward has
  // run (its fused wgrads - or the caller's optimizer right behind it - step W).  A backward that finds it clear (no forward in
  // front of it, or a second backward after one forward) transposes first.
  bool wt_valid = false;
  int64_t n_transposes = 0;        // W^T builds so far (diagnostics: sln_vae_group_transposes)
  // what create() redirected in every engine (outputs / gradient inputs of its decoder): put back by fail() and the destructor, so
  // that an engine used on its own afterwards does not write into the group's freed arrays
  struct EngineIO { float *boxes_pred, *logits, *angles_pred, *dbp, *dlogits, *dz; };
  std::vector<EngineIO> saved_io;
  void restore_engines() {
    for (size_t r = 0; r < saved_io.size() && r < eng.size(); ++r) {
      SlnVae* h = eng[r]; const EngineIO& s = saved_io[r];
      h->boxes_pred = s.boxes_pred; h->logits = s.logits; h->angles_pred = s.angles_pred; h->dbp = s.dbp; h->dlogits = s.dlogits; h->dz = s.dz;
      h->rec = nullptr;
      h->drop_graphs();
    }
    saved_io.clear();
  }
  std::vector<std::pair<const float*, int64_t>> fused;      // room 0's parameter tensors stepped by the wgrad launches (io.sgd_step)
  std::vector<RecStep> singles;          // steps without a multi form: replayed through the single-room launchers
  std::vector<int> single_room;

  // The program's tables (a hundred-odd, a few hundred bytes to 64 KB each) are staged on the host and go to the device as ONE
  // allocation and ONE copy when the program is complete (commit_tables): a hipMalloc + blocking hipMemcpy per table was 3 ms of a
  // 16-room batch's set-up.  Until then a table's pointer holds 1 + its byte offset in the stage.
  std::vector<char> stage;
  template <typename T> int upload(const std::vector<T>& v, const T** out) {
    const size_t off = (stage.size() + 255) / 256 * 256;
    const size_t bytes = sizeof(T) * (v.empty() ? 1 : v.size());
    stage.resize(off + bytes);
    if (!v.empty()) std::memcpy(stage.data() + off, v.data(), sizeof(T) * v.size());
    *out = reinterpret_cast<const T*>(static_cast<uintptr_t>(off + 1));
    return 0;
  }
  int commit_tables() {
    void* d = nullptr;
    hipError_t e = hipMalloc(&d, stage.empty() ? 256 : stage.size());
    if (e != hipSuccess) return SLN_E_NOMEM;
    allocs.push_back(d);
    if (!stage.empty()) { e = hipMemcpy(d, stage.data(), stage.size(), hipMemcpyHostToDevice); if (e != hipSuccess) return (int)e; }
    const char* base = static_cast<const char*>(d);
    auto fix = [&](auto& p) { if (p != nullptr) p = reinterpret_cast<std::remove_reference_t<decltype(p)>>(base + (reinterpret_cast<uintptr_t>(p) - 1)); };
    for (std::vector<Launch>* prog : {&fwd, &bwd})
      for (Launch& l : *prog) { fix(l.tab); fix(l.tiles); fix(l.tn_probs); fix(l.tn_meta); }
    stage.clear(); stage.shrink_to_fit();
    return 0;
  }
  ~SlnVaeGroup() {
    for (void* p : allocs) (void)hipFree(p);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_tr) (void)hipEventDestroy(ev_tr);
    if (ev_join) (void)hipEventDestroy(ev_join);
  }
  int fork_side(hipStream_t st) {
    hipError_t e = hipEventRecord(ev_fork, st);
    if (e == hipSuccess) e = hipStreamWaitEvent(side, ev_fork, 0);
    return (int)e;
  }

  int add_single(std::vector<Launch>& prog, const RecStep& st, int room) {
    Launch l; l.kind = L_SINGLE_STEP; l.single = (int)singles.size();
    singles.push_back(st); single_room.push_back(room);
    prog.push_back(l);
    return 0;
  }
  // the TN problems gathered since the last flush marker -> per-pass wgrad launches over all rooms (flush_deferred's grouping)
  int flush_tn(std::vector<Launch>& prog, std::vector<GemmTNArgs>& pend) {
    static thread_local SlnVae::TnGroup tmp;
    if (!pend.empty()) { Launch f; f.kind = L_FORK; prog.push_back(f); }      // the side stream picks up behind the producers of these gradients
    for (int k = 0; k < 2; ++k) {
      std::vector<GemmTNArgs> kind;
      for (const GemmTNArgs& t : pend) if ((int)tn_gathers_host(t) == k) kind.push_back(t);
      size_t next = 0;
      while (next < kind.size()) {
        int want = 0; long tiles = 0;
        for (size_t i = next; i < kind.size() && want < SLN_TN_MULTI_MAX; ++i) {
          const long tl = (long)sln_cdiv(kind[i].Nout, 64) * sln_cdiv(kind[i].Kin, 64);
          if (want > 0 && tiles + tl > SLN_TN_MULTI_ITEMS / 2) break;
          tiles += tl; ++want;
        }
        int r = -1;
        for (; want >= 1; want /= 2) {
          tmp.n = want;
          for (int i = 0; i < want; ++i) tmp.probs[i] = kind[next + i];
          r = sln_tn_multi_plan(tmp.probs, tmp.n, &tmp.meta, &tmp.blocks, &tmp.x2, &tmp.xg, &tmp.flops);
          if (r == 0) break;
        }
        if (r != 0) {                                     // one problem the planner refuses: its own launch
          RecStep st; st.kind = SK_TN; st.tn = kind[next++];
          if (g_sln_deterministic) st.tn.rows_per_block = sln_cdiv(st.tn.R, 32) * 32;
          Launch l; l.kind = L_TN_SINGLE; l.single = (int)singles.size(); l.on_side = true;
          singles.push_back(st); single_room.push_back(-1);
          prog.push_back(l);
          continue;
        }
        next += (size_t)tmp.n;
        Launch l; l.kind = L_TN_MULTI; l.blocks = tmp.blocks; l.x2 = tmp.x2; l.xg = tmp.xg; l.flops = tmp.flops; l.on_side = true;
        std::vector<GemmTNArgs> pv(tmp.probs, tmp.probs + tmp.n);
        RET_IF(upload(pv, &l.tn_probs));
        std::vector<TnMultiMeta> mv(1, tmp.meta);
        RET_IF(upload(mv, &l.tn_meta));
        prog.push_back(l);
      }
    }
    pend.clear();
    return 0;
  }
  static bool tn_gathers_host(const GemmTNArgs& t) {
    bool g = false;
    for (int s2 = 0; s2 < t.X.nseg; ++s2) g |= t.X.seg[s2].which != 0;
    return g;
  }

  // step s of every room -> launches
  int merge(std::vector<Launch>& prog, const std::vector<Recorder>& recs) {
    const size_t n = recs[0].steps.size();
    for (int r = 1; r < R; ++r) {
      if (recs[r].steps.size() != n) return SLN_E_UNSUPPORTED;
      for (size_t s = 0; s < n; ++s) if (recs[r].steps[s].kind != recs[0].steps[s].kind) return SLN_E_UNSUPPORTED;
    }
    std::vector<GemmTNArgs> pend;
    int n_flush = 0;
    for (size_t s = 0; s < n; ++s) {
      const int kind = recs[0].steps[s].kind;
      if (kind == SK_TN) {
        for (int r = 0; r < R; ++r) {
          GemmTNArgs t = recs[r].steps[s].tn;
          if (io.sgd_step != nullptr && t.lddw == t.Kin) {      // SGD in the wgrad's epilogue: the add lands in the parameter itself
            for (const Unit& u : eng[r]->units)
              if (u.p.d_weight == t.dW && (t.db == nullptr || u.p.d_bias == t.db)) {
                t.dW = u.p.weight; if (t.db != nullptr) t.db = u.p.bias;
                t.sgd_step = io.sgd_step;
                if (r == 0) {                              // (a recurrent stack steps the same weight from several wgrads: listed once)
                  bool listed = false;
                  for (const auto& f : fused) listed = listed || f.first == (const float*)u.p.weight;
                  if (!listed) {
                    fused.push_back(std::make_pair((const float*)u.p.weight, (int64_t)u.out * u.in));
                    if (t.db != nullptr) fused.push_back(std::make_pair((const float*)u.p.bias, (int64_t)u.out));
                  }
                }
                break;
              }
          }
          pend.push_back(t);
        }
        continue;
      }
      if (kind == SK_TN_FLUSH) {
        // lab: SLN_GROUP_FLUSH_EVERY=n hands the wgrads to the side stream at every n-th marker only (fewer forks, later wgrads)
        static const int every = std::getenv("SLN_GROUP_FLUSH_EVERY") ? std::max(1, std::atoi(std::getenv("SLN_GROUP_FLUSH_EVERY"))) : 1;
        bool last = true;
        for (size_t s2 = s + 1; s2 < n; ++s2) if (recs[0].steps[s2].kind == SK_TN) { last = false; break; }
        if (last || ++n_flush % every == 0) RET_IF(flush_tn(prog, pend));
        continue;
      }
      // rooms whose block plans to the same variant share a launch
      std::vector<int> var(R, -1), gxs(R, 0), gys(R, 0), smf(R, 0);
      std::vector<RecStep> blk(R);
      std::vector<char> asm_blob((size_t)R * SLN_ASM_BLOB);
      for (int r = 0; r < R; ++r) {
        blk[r] = recs[r].steps[s];
        RecStep& b = blk[r];
        switch (kind) {
          case SK_NT: { int t = 0; var[r] = sln_plan_nt_small(b.nt, b.epi, &t); if (var[r] >= 0) var[r] = var[r] * 4 + b.epi * 0; gxs[r] = t; break; }
          case SK_SCATTER_FWD: var[r] = sln_plan_scatter_avg_fwd(b.sf); gxs[r] = b.sf.gx; gys[r] = b.sf.gy; break;
          case SK_SCATTER_BWD: var[r] = sln_plan_scatter_avg_bwd(b.sb); gxs[r] = b.sb.gx; gys[r] = b.sb.gy; break;
          case SK_GATHER_BWD: var[r] = sln_plan_gather_bwd(b.gb); gxs[r] = b.gb.gx; gys[r] = b.gb.gy; break;
          case SK_MASK_GSTATS: var[r] = sln_plan_mask_gstats(b.mg); gxs[r] = b.mg.gx; gys[r] = b.mg.gy; break;
          case SK_DEC_ASSEMBLE: var[r] = sln_plan_dec_assemble(b.da); gxs[r] = b.da.gx; gys[r] = 1; break;
          case SK_EMBED_GATHER: var[r] = sln_plan_embed_gather(b.eg); gxs[r] = b.eg.gx; gys[r] = 1; break;
          case SK_EMBED_BWD: var[r] = sln_plan_embed_bwd(b.eb, b.idx64); gxs[r] = b.eb.gx; gys[r] = b.eb.gy; smf[r] = b.eb.table_rows * b.eb.n; break;
          case SK_ADD2: case SK_COPY2D: var[r] = sln_plan_add2(b.a2); gxs[r] = b.a2.gx; gys[r] = 1; break;
          case SK_DEC_ASSEMBLE_BWD: var[r] = sln_plan_dec_assemble_bwd(b.dab, asm_blob.data() + (size_t)r * SLN_ASM_BLOB, &gxs[r], &smf[r]); gys[r] = 1; break;
          default: return SLN_E_UNSUPPORTED;
        }
      }
      std::vector<bool> done(R, false);
      for (int r0 = 0; r0 < R; ++r0) {
        if (done[r0]) continue;
        if (var[r0] < 0) { done[r0] = true; RET_IF(add_single(prog, recs[r0].steps[s], r0)); continue; }
        std::vector<int> rooms;
        for (int r = r0; r < R; ++r) if (!done[r] && var[r] == var[r0]) { rooms.push_back(r); done[r] = true; }
        Launch l; l.kind = kind; l.variant = var[r0]; l.count = (int)rooms.size();
        for (int r : rooms) { l.gx = gxs[r] > l.gx ? gxs[r] : l.gx; l.gy = gys[r] > l.gy ? gys[r] : l.gy; l.smem_floats = smf[r] > l.smem_floats ? smf[r] : l.smem_floats; }
#define SLN_UP(FIELD, TYPE) { std::vector<TYPE> v; for (int r : rooms) v.push_back(blk[r].FIELD); const TYPE* d = nullptr; RET_IF(upload(v, &d)); l.tab = d; }
        switch (kind) {
          case SK_NT: {
            l.kind = L_NT; l.variant = var[r0] / 4;
            std::vector<GemmNTArgs> v; std::vector<int> tl;
            for (int r : rooms) { v.push_back(blk[r].nt); tl.push_back(gxs[r]); l.maxK = blk[r].nt.K > l.maxK ? blk[r].nt.K : l.maxK; l.flops += 2.0 * blk[r].nt.M * blk[r].nt.N * blk[r].nt.K; }
            const GemmNTArgs* d = nullptr; RET_IF(upload(v, &d)); l.tab = d;
            RET_IF(upload(tl, &l.tiles));
            break;
          }
          case SK_SCATTER_FWD: SLN_UP(sf, MScatterFwd) break;
          case SK_SCATTER_BWD: SLN_UP(sb, MScatterBwd) break;
          case SK_GATHER_BWD: SLN_UP(gb, MGatherBwd) break;
          case SK_MASK_GSTATS: SLN_UP(mg, MMaskGstats) break;
          case SK_DEC_ASSEMBLE: SLN_UP(da, MDecAssemble) break;
          case SK_EMBED_GATHER: SLN_UP(eg, MEmbedGather) break;
          case SK_EMBED_BWD: SLN_UP(eb, MEmbedBwd) break;
          case SK_ADD2: case SK_COPY2D: SLN_UP(a2, MAdd2) break;
          case SK_DEC_ASSEMBLE_BWD: {
            std::vector<char> v;
            for (int r : rooms) v.insert(v.end(), asm_blob.begin() + (size_t)r * SLN_ASM_BLOB, asm_blob.begin() + (size_t)(r + 1) * SLN_ASM_BLOB);
            const char* d = nullptr; RET_IF(upload(v, &d)); l.tab = d;
            break;
          }
        }
#undef SLN_UP
        prog.push_back(l);
      }
    }
    if (!pend.empty()) RET_IF(flush_tn(prog, pend));
    return 0;
  }

  int run(const std::vector<Launch>& prog, hipStream_t st) {
    const bool want_side = use_side;
    // (no side stream inside a capture: this runtime replays a forked hipGraph node by node from the host and does not overlap its
    //  branches - see sln_scene_backward; SLN_CAPTURE_SIDE=1 keeps the forks, lab)
    static const bool capture_side = std::getenv("SLN_CAPTURE_SIDE") != nullptr;
    side = (want_side && (capture_side || !sln_capturing(st))) ? sln_overlapping_stream(st) : nullptr;
    struct Restore { bool& flag; bool v; ~Restore() { flag = v; } } restore{use_side, want_side};
    if (side == nullptr) use_side = false;          // none to be had (first use inside a capture): this pass on the caller's stream
    for (const Launch& l : prog) {
      int r = 0;
      hipStream_t main_st = st;
      hipStream_t st = (l.on_side && use_side) ? side : main_st;       // (shadows the parameter for the launches below)
      switch (l.kind) {
        case L_FORK: if (use_side) r = fork_side(main_st); break;
        case L_JOIN:
          if (use_side) { hipError_t e = hipEventRecord(ev_join, side); if (e == hipSuccess) e = hipStreamWaitEvent(main_st, ev_join, 0); r = (int)e; }
          break;
        case L_JOIN_TR:
          if (tr_pending) { r = (int)hipStreamWaitEvent(main_st, ev_tr, 0); tr_pending = false; }
          break;
        case L_NT: r = sln_launch_gemm_nt_small_multi(static_cast<const GemmNTArgs*>(l.tab), l.tiles, l.count, l.variant, l.gx, l.maxK, l.flops, st); break;
        case L_TN_MULTI: r = sln_launch_gemm_tn_multi(l.tn_probs, l.tn_meta, l.blocks, l.x2, l.xg, l.flops, st); break;
        case L_TN_SINGLE: r = sln_launch_gemm_tn(singles[l.single].tn, -1, st); break;
        case L_SINGLE_STEP: r = run_single(singles[l.single], st); break;
        case SK_SCATTER_FWD: r = sln_launch_scatter_avg_fwd_multi(static_cast<const MScatterFwd*>(l.tab), l.count, l.variant, l.gx, l.gy, st); break;
        case SK_SCATTER_BWD: r = sln_launch_scatter_avg_bwd_multi(static_cast<const MScatterBwd*>(l.tab), l.count, l.variant, l.gx, l.gy, st); break;
        case SK_GATHER_BWD: r = sln_launch_gather_bwd_multi(static_cast<const MGatherBwd*>(l.tab), l.count, l.variant, l.gx, l.gy, st); break;
        case SK_MASK_GSTATS: r = sln_launch_mask_gstats_multi(static_cast<const MMaskGstats*>(l.tab), l.count, l.gx, l.gy, st); break;
        case SK_DEC_ASSEMBLE: r = sln_launch_dec_assemble_multi(static_cast<const MDecAssemble*>(l.tab), l.count, l.gx, st); break;
        case SK_EMBED_GATHER: r = sln_launch_embed_gather_multi(static_cast<const MEmbedGather*>(l.tab), l.count, l.gx, st); break;
        case SK_EMBED_BWD: r = sln_launch_embed_bwd_multi(static_cast<const MEmbedBwd*>(l.tab), l.count, l.variant, l.gx, l.gy, l.smem_floats, st); break;
        case SK_ADD2: r = sln_launch_add2_multi(static_cast<const MAdd2*>(l.tab), l.count, l.gx, st); break;
        case SK_COPY2D: r = sln_launch_copy2d_multi(static_cast<const MAdd2*>(l.tab), l.count, l.gx, st); break;
        case SK_DEC_ASSEMBLE_BWD: r = sln_launch_dec_assemble_bwd_multi(l.tab, l.count, l.variant, l.gx, l.smem_floats, st); break;
        case L_ZERO: r = sln_launch_zero_multi(static_cast<const MZero*>(l.tab), l.count, l.max_n16, st); break;
        case L_TRANSPOSE:
          if (l.variant == 1 && wt_valid) break;          // backward's own transposition: only when no forward left a valid W^T
          r = sln_launch_transpose_table(static_cast<const TransposeEntry*>(l.tab), l.count, l.gx, st);
          if (!r && l.on_side && use_side) { r = (int)hipEventRecord(ev_tr, side); tr_pending = true; }
          if (!r) { wt_valid = true; ++n_transposes; }
          break;
        case L_BN_GRADS: r = sln_launch_bn_param_grads(static_cast<const BnTableEntry*>(l.tab), l.count, l.gx, eng[0]->cfg.recurrent ? 0 : 1, st); break;
        case L_LOG_SOFTMAX: r = sln_launch_log_softmax(logits, io.angles_pred, rows_total, n_angle, st); break;
        case L_LOG_SOFTMAX_BWD: r = sln_launch_log_softmax_bwd(io.angles_pred, io.d_angles_pred, dlogits, rows_total, n_angle, st); break;
        default: r = SLN_E_UNSUPPORTED;
      }
      if (r) return r;
    }
    if (&prog == &bwd) wt_valid = false;             // the parameters move behind a backward pass (fused wgrads, or the caller's step)
    return 0;
  }
  // a recorded step through the single-room launcher (a room whose block has no multi form)
  int run_single(const RecStep& b, hipStream_t st) {
    switch (b.kind) {
      case SK_NT: return sln_launch_gemm_nt(b.nt, b.epi, -1, st);
      case SK_SCATTER_FWD: return sln_launch_scatter_avg_fwd(b.sf.A2, b.sf.ld, b.sf.H, b.sf.D, b.sf.bn, b.sf.g, b.sf.O, b.sf.pooled, st);
      case SK_SCATTER_BWD: return sln_launch_scatter_avg_bwd(b.sb.dM, b.sb.dP, b.sb.lddp, b.sb.dpcol0, b.sb.A2, b.sb.ld, b.sb.H, b.sb.D, b.sb.bn, b.sb.g, b.sb.T,
                                                             b.sb.g2, b.sb.gsums, b.sb.cstride, st);
      case SK_GATHER_BWD: return sln_launch_gather_bwd(b.gb.dG, b.gb.ldg, b.gb.D, b.gb.g, b.gb.O, b.gb.add1, b.gb.ldadd1, b.gb.xprev, b.gb.ldx, b.gb.bn,
                                                       b.gb.masked, b.gb.out, b.gb.ldo, b.gb.gsums, b.gb.cstride, st);
      case SK_MASK_GSTATS: return sln_launch_mask_gstats(b.mg.d1, b.mg.ld1, b.mg.d2, b.mg.ld2, b.mg.xprev, b.mg.ldx, b.mg.bn, b.mg.rows, b.mg.cols, b.mg.out,
                                                         b.mg.ldo, b.mg.gsums, b.mg.cstride, st);
      case SK_DEC_ASSEMBLE: return sln_launch_dec_assemble(b.da.a, st);
      case SK_EMBED_GATHER: return sln_launch_embed_gather_i32(b.eg.idx, b.eg.emb, b.eg.rows, b.eg.n, b.eg.out, st);
      case SK_EMBED_BWD: return b.idx64 ? sln_launch_embed_bwd_i64(static_cast<const int64_t*>(b.eb.idx), b.eb.d, b.eb.ld, b.eb.col0, b.eb.rows, b.eb.n, b.eb.table_rows, b.eb.d_emb, st)
                                        : sln_launch_embed_bwd_i32(static_cast<const int*>(b.eb.idx), b.eb.d, b.eb.ld, b.eb.col0, b.eb.rows, b.eb.n, b.eb.table_rows, b.eb.d_emb, st);
      case SK_ADD2: return sln_launch_add2(b.a2.a, b.a2.lda, b.a2.b, b.a2.ldb, b.a2.rows, b.a2.cols, b.a2.out, b.a2.ldo, st);
      case SK_COPY2D: return (int)hipMemcpy2DAsync(b.a2.out, sizeof(float) * b.a2.ldo, b.a2.a, sizeof(float) * b.a2.lda, sizeof(float) * b.a2.cols, (size_t)b.a2.rows,
                                                   hipMemcpyDeviceToDevice, st);
      case SK_DEC_ASSEMBLE_BWD: return sln_launch_dec_assemble_bwd(b.dab, st);
      default: return SLN_E_UNSUPPORTED;
    }
  }
};

extern "C" {

int sln_vae_group_create(SlnVae* const* engines, int R, const SlnVaeGroupIO* io, SlnVaeGroup** out) {
  if (!engines || R < 1 || !io || !out || !io->row0_host || !io->z || !io->boxes_pred || !io->angles_pred || !io->d_boxes_pred ||
      !io->d_angles_pred || !io->dz || io->rows_total < 1) return SLN_E_BADARG;
  for (int r = 0; r < R; ++r) {
    SlnVae* h = engines[r];
    if (!h || !h->bound || !h->batch_set || h->gconv_only) return SLN_E_STATE;
    if (std::memcmp(&h->cfg, &engines[0]->cfg, sizeof(SlnVaeConfig)) != 0) return SLN_E_BADARG;
    if (io->row0_host[r] < 0 || io->row0_host[r] + h->O > io->rows_total) return SLN_E_BADARG;
  }
  SlnVaeGroup* g = new (std::nothrow) SlnVaeGroup();
  if (!g) return SLN_E_NOMEM;
  g->R = R; g->io = *io; g->io.row0_host = nullptr; g->rows_total = io->rows_total; g->n_angle = engines[0]->cfg.n_angle;
  g->eng.assign(engines, engines + R);
  int rc = sln_gemm_init();
  auto fail = [&](int code) { for (SlnVae* h : g->eng) h->rec = nullptr; g->restore_engines(); delete g; return code; };
  if (rc) return fail(rc);
  {
    void* p = nullptr;
    if (hipMalloc(&p, sizeof(float) * (size_t)g->rows_total * g->n_angle * 2) != hipSuccess) return fail(SLN_E_NOMEM);
    g->allocs.push_back(p);
    g->logits = static_cast<float*>(p); g->dlogits = g->logits + (size_t)g->rows_total * g->n_angle;
  }
  std::vector<Recorder> rf(R), rb(R);
  std::vector<MZero> zeros; std::vector<TransposeEntry> trs; std::vector<BnTableEntry> bnt;
  long max_n16 = 0; int tr_tiles = 0, bn_maxc = 0;
  for (int r = 0; r < R; ++r) {
    SlnVae* h = g->eng[r];
    const size_t row0 = (size_t)io->row0_host[r];
    const int E = h->E, na = h->cfg.n_angle;
    // the engine's outputs / gradient inputs of the decoder become slices of the group's row-concatenated arrays
    g->saved_io.push_back(SlnVaeGroup::EngineIO{h->boxes_pred, h->logits, h->angles_pred, h->dbp, h->dlogits, h->dz});
    h->boxes_pred = io->boxes_pred + row0 * h->cfg.box_dim; h->logits = g->logits + row0 * na; h->angles_pred = io->angles_pred + row0 * na;
    if (h->dbp_ld != 8 && h->dbp_ld != h->cfg.box_dim) return fail(SLN_E_UNSUPPORTED);
    h->dbp = io->d_boxes_pred + row0 * h->dbp_ld; h->dlogits = g->dlogits + row0 * na; h->dz = io->dz + row0 * E;
    h->drop_graphs();
    h->rec = &rf[r];
    rc = h->decoder_forward(io->z + row0 * E, nullptr, false, nullptr);
    if (!rc) { h->rec = &rb[r]; rc = h->decoder_backward(nullptr); }
    h->rec = nullptr;
    if (rc) return fail(rc);
    // per-room pieces of the three table-driven launches: the BatchNorm backward sums to clear, W^T of the decoder's units,
    // the BatchNorm parameter gradients of the decoder's applications
    const size_t dec_doubles = h->stats_doubles - h->enc_stats_doubles;
    if (dec_doubles) {
      MZero z; z.p = h->gstats_base + h->enc_stats_doubles; z.n16 = (long)(dec_doubles * sizeof(double) / 16);
      if ((reinterpret_cast<uintptr_t>(z.p) & 15) || (dec_doubles * sizeof(double)) % 16) return fail(SLN_E_UNSUPPORTED);
      zeros.push_back(z); max_n16 = z.n16 > max_n16 ? z.n16 : max_n16;
    }
    std::vector<int> dec_units;
    for (int l = 0; l < h->L; ++l) for (int k = 0; k < 4; ++k) {
      const int u = h->unit_of(1, l, k);
      bool seen = false; for (int q : dec_units) seen |= q == u;
      if (!seen) dec_units.push_back(u);
    }
    for (int k = 0; k < 2; ++k) { dec_units.push_back(h->unit_boxnet(k)); dec_units.push_back(h->unit_anglenet(k)); }
    for (int ui : dec_units) {
      const Unit& u = h->units[ui];
      TransposeEntry e; e.src = u.p.weight; e.dst = u.wt; e.rows = u.out; e.cols = u.in; e.dst_ld = u.wt_ld; e.pad_ = 0;
      trs.push_back(e);
      const int tiles = sln_cdiv(u.out, 32) * sln_cdiv(u.in, 32);
      tr_tiles = tiles > tr_tiles ? tiles : tr_tiles;
    }
    // (shared 'recurrent' modules: the per-entry form of the parameter-gradient kernel would let two applications of one module add
    //  to the same dgamma from different workgroups - that launch then takes the serial form, see L_BN_GRADS in run())
    for (size_t i = (size_t)h->n_bn_enc; i < h->bns.size(); ++i) {
      const BnInst& b = h->bns[i];
      const Unit& u = h->units[b.unit];
      BnTableEntry e; std::memset(&e, 0, sizeof(e));
      e.sums = b.sums; e.gsums = b.gsums; e.cstride = b.C; e.C = b.C; e.rows = b.rows_code;
      e.rmean = u.p.bn_running_mean; e.rvar = u.p.bn_running_var; e.nbt = u.p.bn_num_batches_tracked;
      e.dgamma = u.p.d_bn_weight; e.dbeta = u.p.d_bn_bias;
      bnt.push_back(e); bn_maxc = b.C > bn_maxc ? b.C : bn_maxc;
    }
  }
  if (hipEventCreateWithFlags(&g->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&g->ev_tr, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&g->ev_join, hipEventDisableTiming) != hipSuccess)
    return fail(SLN_E_NOMEM);
  // the side stream pays from four rooms on (one / two rooms: 0.86 / 0.93 ms per iteration without it, 0.90 / 0.96 with; four: equal);
  // SLN_GROUP_NO_SIDE=1 / =0 forces it off / on (lab)
  { const char* v = std::getenv("SLN_GROUP_NO_SIDE"); g->use_side = v ? v[0] != '1' : R >= 4; }
  // forward program: W^T of the decoder's weights on the side stream (the parameters are final: the previous iteration's update
  // is in front of the fork), the recorded steps, then ONE log-softmax over every room's rows
  { SlnVaeGroup::Launch l; l.kind = SlnVaeGroup::L_FORK; g->fwd.push_back(l); }
  {
    SlnVaeGroup::Launch l; l.kind = SlnVaeGroup::L_TRANSPOSE; l.count = (int)trs.size(); l.gx = tr_tiles; l.on_side = true;
    const TransposeEntry* d = nullptr; rc = g->upload(trs, &d); if (rc) return fail(rc); l.tab = d;
    g->fwd.push_back(l);
  }
  rc = g->merge(g->fwd, rf);
  if (rc) return fail(rc);
  { SlnVaeGroup::Launch l; l.kind = SlnVaeGroup::L_LOG_SOFTMAX; g->fwd.push_back(l); }
  // backward program: log-softmax backward, the cleared sums, W^T, the recorded steps, the BatchNorm parameter gradients
  { SlnVaeGroup::Launch l; l.kind = SlnVaeGroup::L_LOG_SOFTMAX_BWD; g->bwd.push_back(l); }
  if (!zeros.empty()) {
    SlnVaeGroup::Launch l; l.kind = SlnVaeGroup::L_ZERO; l.count = (int)zeros.size(); l.max_n16 = max_n16;
    const MZero* d = nullptr; rc = g->upload(zeros, &d); if (rc) return fail(rc); l.tab = d;
    g->bwd.push_back(l);
  }
  {
    // W^T: built by the forward call's launch (side stream: join).  A backward that does not find a valid one - no forward in
    // front of it, or a second backward after one forward, whose first has stepped W - builds it here (variant 1: skipped when
    // SlnVaeGroup::wt_valid)
    SlnVaeGroup::Launch j; j.kind = SlnVaeGroup::L_JOIN_TR; g->bwd.push_back(j);
    SlnVaeGroup::Launch l; l.kind = SlnVaeGroup::L_TRANSPOSE; l.count = (int)trs.size(); l.gx = tr_tiles; l.variant = 1;
    l.tab = g->fwd[1].tab;
    g->bwd.push_back(l);
  }
  {
    std::vector<SlnVaeGroup::Launch> rest;
    rc = g->merge(rest, rb);
    if (rc) return fail(rc);
    // the parameter-gradient launch of the BatchNorm applications reads the sums every masked dgrad / edge kernel has finished:
    // behind the last recorded step (the wgrads run on the side stream and do not touch them)
    for (size_t i = 0; i < rest.size(); ++i) g->bwd.push_back(rest[i]);
    if (!bnt.empty()) {
      SlnVaeGroup::Launch l; l.kind = SlnVaeGroup::L_BN_GRADS; l.count = (int)bnt.size(); l.gx = bn_maxc;
      const BnTableEntry* d = nullptr; rc = g->upload(bnt, &d); if (rc) return fail(rc); l.tab = d;
      g->bwd.push_back(l);
    }
    { SlnVaeGroup::Launch l; l.kind = SlnVaeGroup::L_JOIN; g->bwd.push_back(l); }     // every wgrad has landed behind this point
  }
  rc = g->commit_tables();
  if (rc) return fail(rc);
  *out = g;
  return 0;
}

int sln_vae_group_decoder(SlnVaeGroup* g, void* stream) {
  if (!g) return SLN_E_BADARG;
  return g->run(g->fwd, (hipStream_t)stream);
}

int sln_vae_group_decoder_backward(SlnVaeGroup* g, void* stream) {
  if (!g) return SLN_E_BADARG;
  return g->run(g->bwd, (hipStream_t)stream);
}

int sln_vae_group_launches(const SlnVaeGroup* g, int* fwd, int* bwd, int* single_room_fallbacks) {
  if (!g) return SLN_E_BADARG;
  if (fwd) *fwd = (int)g->fwd.size();
  if (bwd) *bwd = (int)g->bwd.size();
  if (single_room_fallbacks) *single_room_fallbacks = (int)g->singles.size();
  return 0;
}

int64_t sln_vae_group_transposes(const SlnVaeGroup* g) { return g ? g->n_transposes : -1; }

int sln_vae_group_fused_params(const SlnVaeGroup* g, const float** params, int64_t* numel, int max) {
  if (!g || max < 0 || (max > 0 && (!params || !numel))) return SLN_E_BADARG;
  const int n = (int)g->fused.size();
  for (int i = 0; i < n && i < max; ++i) { params[i] = g->fused[i].first; numel[i] = g->fused[i].second; }
  return n;
}

void sln_vae_group_destroy(SlnVaeGroup* g) {
  if (!g) return;
  g->restore_engines();          // (the engines must still be alive: destroy a group before its engines)
  delete g;
}

int64_t sln_vae_tap(SlnVae* h, int layer, int what, float* dst, void* stream) {
  if (!h || layer < 0 || layer >= (int)h->layers.size() || !dst) return SLN_E_BADARG;
  const Layer& ly = h->layers[layer];
  const int H = h->H, D = ly.Do;
  const float* src; size_t n;
  switch (what) {
    case 0: src = ly.A1; n = (size_t)h->T * H; break;
    case 1: src = ly.A2; n = (size_t)h->T * (2 * H + D); break;
    case 2: src = ly.M; n = (size_t)h->O * H; break;
    case 3: src = ly.A3; n = (size_t)h->O * H; break;
    case 4: src = ly.A4; n = (size_t)h->O * D; break;
    default: return SLN_E_BADARG;
  }
  hipError_t e = hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream);
  return e == hipSuccess ? (int64_t)n : -(int64_t)e;
}

int sln_linear_forward(const float* x, int M, int K, const float* W, const float* bias, float* y, int N, double* sums,
                       int tile, void* stream) {
  if (!x || !W || !y || M <= 0 || N <= 0 || K <= 0 || (K & 3)) return SLN_E_BADARG;
  GemmNTArgs a; std::memset(&a, 0, sizeof(a));
  Seg s; std::memset(&s, 0, sizeof(s));
  s.x1 = x; s.ld1 = K; s.len = K; s.coef = SLN_COEF_IDENT;
  a.A.seg[0] = s; a.A.nseg = 1; a.A.rows = M; a.A.cols = K;
  a.W = W; a.bias = bias; a.Y = y; a.ldy = N; a.M = M; a.N = N; a.K = K; a.ldw = K;
  a.osums = sums; a.ocstride = N;
  return sln_launch_gemm_nt(a, sums ? EPI_STATS : EPI_PLAIN, tile, (hipStream_t)stream);
}

int sln_linear_wgrad(const float* gq, const float* x, int R, int N, int K, float* dW, float* db, void* stream) {
  if (!gq || !x || !dW || R <= 0 || N <= 0 || K <= 0 || (K & 3) || (N & 3)) return SLN_E_BADARG;
  GemmTNArgs a; std::memset(&a, 0, sizeof(a));
  Seg sg; std::memset(&sg, 0, sizeof(sg));
  sg.x1 = gq; sg.ld1 = N; sg.len = N; sg.coef = SLN_COEF_IDENT;
  Seg sx = sg; sx.x1 = x; sx.ld1 = K; sx.len = K;
  a.G.seg[0] = sg; a.G.nseg = 1; a.G.rows = R; a.G.cols = N;
  a.X.seg[0] = sx; a.X.nseg = 1; a.X.rows = R; a.X.cols = K;
  a.dW = dW; a.db = db; a.lddw = K; a.R = R; a.Nout = N; a.Kin = K;
  return sln_launch_gemm_tn(a, -1, (hipStream_t)stream);
}

}  // extern "C"
