// Standalone GraphTripleConv / GraphTripleConvNet forward (reference models/graph.py:57-111,136-143) through
// the same kernels the VAE engine uses.  Inference / feature extraction only: the training path (with its
// fused backward) is the VAE engine (vae_engine.hip); the Python mirror raises if gradients are requested.
#include <cstring>
#include <vector>

#include "../../include/sln_hip.h"
#include "sln_gemm.h"
#include "vae_kernels.h"

namespace {
struct Ws {
  GraphCsr g; int* err; double* sums; float *A1, *A2, *M, *A3, *A4, *X, *P;
};
size_t carve(void* base, int D, int H, int Do, int O, int T, int layers, Ws* w) {
  char* p = static_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t bytes) { off = (off + 255) & ~size_t(255); char* r = base ? p + off : nullptr; off += bytes; return r; };
  const size_t Os = O, Ts = T > 0 ? T : 1;
  Ws z; std::memset(&z, 0, sizeof(z));
  z.g.s = (int*)take(4 * Ts); z.g.p = (int*)take(4 * Ts); z.g.o = (int*)take(4 * Ts); z.g.deg = (int*)take(4 * Os);
  z.g.invdeg = (float*)take(4 * Os); z.g.rowptr = (int*)take(4 * (Os + 1)); z.g.cursor = (int*)take(4 * Os); z.g.ent = (int*)take(8 * Ts);
  z.err = (int*)take(16);
  z.sums = (double*)take(sizeof(double) * 2 * (size_t)(3 * H + 2 * H + 2 * Do) * layers);
  z.A1 = (float*)take(4 * Ts * H); z.A2 = (float*)take(4 * Ts * (2 * H + Do)); z.M = (float*)take(4 * Os * H);
  z.A3 = (float*)take(4 * Os * H); z.A4 = (float*)take(4 * Os * Do);
  z.X = (float*)take(4 * Os * (D > Do ? D : Do)); z.P = (float*)take(4 * Ts * (D > Do ? D : Do));
  if (w) *w = z;
  return (off + 255) & ~size_t(255);
}
Seg ident(const float* x, int ld, int col0, int len, int which) {
  Seg s; std::memset(&s, 0, sizeof(s));
  s.x1 = x; s.ld1 = ld; s.c1 = col0; s.len = len; s.which = which; s.coef = SLN_COEF_IDENT;
  return s;
}
BnView bnview(const SlnVaeUnit& u, double* sums, int C, int rows, int col0, int mode) {
  BnView v; std::memset(&v, 0, sizeof(v));
  v.mode = mode; v.eps = 1e-5f; v.n_rows = (float)(rows > 0 ? rows : 1); v.rn = 1.0 / (double)(rows > 0 ? rows : 1);
  if (mode == SLN_BN_NONE) return v;
  v.sums = sums + col0; v.gsums = nullptr; v.cstride = C;
  v.gamma = u.bn_weight + col0; v.beta = u.bn_bias + col0; v.rmean = u.bn_running_mean + col0; v.rvar = u.bn_running_var + col0;
  return v;
}
Operand one(const Seg& s, int rows) {
  Operand o; std::memset(&o, 0, sizeof(o));
  o.seg[0] = s; o.nseg = 1; o.rows = rows; o.cols = s.len;
  return o;
}
int linear(const Operand& A, const SlnVaeUnit& u, int out, int in, float* Y, int M, double* sums, int mode, hipStream_t st) {
  GemmNTArgs a; std::memset(&a, 0, sizeof(a));
  a.A = A; a.W = u.weight; a.bias = u.bias; a.Y = Y; a.ldy = out; a.M = M; a.N = out; a.K = in; a.ldw = in;
  int epi = EPI_PLAIN;
  if (mode == SLN_BN_TRAIN) { epi = EPI_STATS; a.osums = sums; a.ocstride = out; }
  return sln_launch_gemm_nt(a, epi, -1, st);
}
}  // namespace

#define RET_IF(x) do { int r__ = (x); if (r__ != 0) return r__; } while (0)

extern "C" {

int64_t sln_gconv_workspace_bytes(int D, int H, int Dout, int O, int T, int num_layers) {
  if (D <= 0 || H <= 0 || Dout <= 0 || O <= 0 || T < 0 || num_layers <= 0) return SLN_E_BADARG;
  return (int64_t)carve(nullptr, D, H, Dout, O, T, num_layers, nullptr);
}

// units_host: 4 entries per module (net1.0, net1.1, net2.0, net2.1); module of layer l = n_modules == 1 ? 0 : l.
// obj_vecs [O,D], pred_vecs [T,D], edges [T,2] int64 -> new_obj [O,Dout], new_pred [T,Dout].
// num_layers > 1 requires Dout == D (GraphTripleConvNet).
int sln_gconv_forward(int D, int H, int Dout, int num_layers, int n_modules, int batch_norm, const SlnVaeUnit* units_host,
                      const float* obj_vecs, const float* pred_vecs, const int64_t* edges, int O, int T, int training, void* workspace,
                      int64_t workspace_bytes, float* new_obj, float* new_pred, void* stream) {
  if (!units_host || !obj_vecs || !pred_vecs || !workspace || !new_obj || !new_pred || (T > 0 && !edges)) return SLN_E_BADARG;
  if (D % 32 || H % 4 || Dout % 4 || (num_layers > 1 && D != Dout) || n_modules < 1) return SLN_E_UNSUPPORTED;
  if (workspace_bytes < sln_gconv_workspace_bytes(D, H, Dout, O, T, num_layers)) return SLN_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  RET_IF(sln_gemm_init());
  Ws w; carve(workspace, D, H, Dout, O, T, num_layers, &w);
  w.g.T = T; w.g.O = O;
  const int mode = batch_norm ? (training ? SLN_BN_TRAIN : SLN_BN_EVAL) : SLN_BN_NONE;
  const int C2 = 2 * H + Dout;
  const size_t per_layer = 2 * (size_t)(3 * H + 2 * H + 2 * Dout);       // doubles: [bn1 2H | bn2 2C2 | bn3 2H | bn4 2Do]
  int e = sln_zero_async(w.err, sizeof(int), st);                  // caller-owned workspace: kernel fills (sln_common.h)
  if (e == 0 && mode == SLN_BN_TRAIN) e = sln_zero_async(w.sums, sizeof(double) * per_layer * num_layers, st);
  if (e != hipSuccess) return (int)e;
  RET_IF(sln_launch_graph_prep(edges, T, O, 1 << 30, w.g, w.err, st, 1));
  const float* X = obj_vecs; const float* P = pred_vecs;
  std::vector<BnTableEntry> table;
  for (int l = 0; l < num_layers; ++l) {
    const SlnVaeUnit* u = units_host + 4 * (n_modules == 1 ? 0 : l);
    double* s1 = w.sums + per_layer * l; double* s2 = s1 + 2 * H; double* s3 = s2 + 2 * C2; double* s4 = s3 + 2 * H;
    Operand in; std::memset(&in, 0, sizeof(in));
    in.seg[0] = ident(X, D, 0, D, 1); in.seg[1] = ident(P, D, 0, D, 0); in.seg[2] = ident(X, D, 0, D, 2);
    in.nseg = 3; in.rows = T; in.cols = 3 * D; in.idx_a = w.g.s; in.idx_b = w.g.o;
    RET_IF(linear(in, u[0], H, 3 * D, w.A1, T, s1, mode, st));
    Seg a1 = ident(w.A1, H, 0, H, 0); a1.coef = SLN_COEF_FWD; a1.bn = bnview(u[0], s1, H, T, 0, mode);
    RET_IF(linear(one(a1, T), u[1], C2, H, w.A2, T, s2, mode, st));
    RET_IF(sln_launch_scatter_avg_fwd(w.A2, C2, H, Dout, bnview(u[1], s2, C2, T, 0, mode), w.g, O, w.M, st));
    RET_IF(linear(one(ident(w.M, H, 0, H, 0), O), u[2], H, H, w.A3, O, s3, mode, st));
    Seg a3 = ident(w.A3, H, 0, H, 0); a3.coef = SLN_COEF_FWD; a3.bn = bnview(u[2], s3, H, O, 0, mode);
    RET_IF(linear(one(a3, O), u[3], Dout, H, w.A4, O, s4, mode, st));
    const bool last = l == num_layers - 1;
    float* xo = last ? new_obj : w.X; float* po = last ? new_pred : w.P;
    RET_IF(sln_launch_bn_relu_apply(w.A4, Dout, 0, Dout, O, bnview(u[3], s4, Dout, O, 0, mode), xo, Dout, st));
    RET_IF(sln_launch_bn_relu_apply(w.A2, C2, H, Dout, T, bnview(u[1], s2, C2, T, H, mode), po, Dout, st));
    X = xo; P = po;
    if (mode == SLN_BN_TRAIN) {
      const int Cs[4] = {H, C2, H, Dout}; double* ss[4] = {s1, s2, s3, s4}; const int rows[4] = {T, T, O, O};
      for (int k = 0; k < 4; ++k) {
        BnTableEntry t; std::memset(&t, 0, sizeof(t));
        t.sums = ss[k]; t.cstride = Cs[k]; t.C = Cs[k]; t.rows = rows[k];
        t.rmean = u[k].bn_running_mean; t.rvar = u[k].bn_running_var; t.nbt = u[k].bn_num_batches_tracked;
        table.push_back(t);
      }
    }
  }
  if (!table.empty()) {      // running statistics (momentum 0.1), in application order; small blocking upload
    BnTableEntry* dev = nullptr;
    e = hipMalloc(&dev, sizeof(BnTableEntry) * table.size());
    if (e != hipSuccess) return (int)e;
    e = hipMemcpy(dev, table.data(), sizeof(BnTableEntry) * table.size(), hipMemcpyHostToDevice);
    int r = e == hipSuccess ? sln_launch_bn_running_update(dev, (int)table.size(), 2 * H + Dout, 0.1f, n_modules == 1 && num_layers > 1 ? 0 : 1, st)
                            : (int)e;
    if (r == 0) e = hipStreamSynchronize(st);
    (void)hipFree(dev);
    if (r) return r;
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

}  // extern "C"
