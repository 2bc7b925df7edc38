// extern "C" entry points of the training path (include/sepr.h, "Training path"): per block a train-mode forward that keeps
// what the backward needs in a caller-provided context buffer, and the backward.  Each is a fixed sequence of launches
// on the caller's stream: projections on the forward projection cores (sepr_gemm.h / sepr_gemm_x3.h) with transposed,
// affine-folded weights for the input gradients, the TN contraction (sepr_gemm_tn.hip) for the weight gradients, and
// the row-wise pieces of sepr_train_pw.hip / sepr_train_attn.hip.
//
// Layout of a block's context and scratch is defined ONCE, by the block's own implementation running in "dry" mode
// (Carve::dry): sepr_train_ctx_bytes / sepr_train_ws_bytes replay the same carving without launching anything.
#include <stdlib.h>
#include <string.h>

#include "sepr_gcfn_fused.h"
#include "sepr_pointwise.h"
#include "sepr_train.h"

namespace sepr {
namespace {

const float LN_EPS_T = 1e-5f, BN_EPS_T = 1e-5f, BN_MOM = 0.1f;
// The sizing entry points (sepr_train_ctx_bytes / sepr_train_ws_bytes) carry no encoder geometry: the training path is
// sized for - and its waveform-end entry points only accept - the encoder / decoder filter every shipped configuration uses
// (configs.yaml: kernel_size 16, stride 4).  Any other geometry is a clear SEPR_EINVAL instead of an undersized workspace.
constexpr int TRAIN_ENC_K = 16, TRAIN_ENC_STRIDE = 4;

struct Carve {
  char* base;
  size_t size, off;
  bool dry;
  Carve(void* p, size_t n, bool d) : base(static_cast<char*>(p)), size(n), off(0), dry(d) {}
  void* take(size_t bytes) {
    const size_t o = align_up(off);
    off = o + bytes;
    if (dry) return reinterpret_cast<void*>(0x1000);   // never dereferenced
    if (!base || off > size) return nullptr;
    return base + o;
  }
  float* f32(size_t count) { return static_cast<float*>(take(count * sizeof(float))); }
  bool ok() const { return dry || (base != nullptr && off <= size); }
  size_t need() const { return align_up(off); }
};

// Scratch of ONE finished projection - the reduced contraction G [N][K] and its column sums s [N] (sepr_train.h finishers).  With a
// deferred-finisher window open on this thread the pair lives in the window's arena until the flush (every projection its OWN slot: the
// finisher runs later, so the buffers of a block cannot be shared between its projections as the workspace copies are); otherwise, and
// in sizing runs, it is carved from the workspace like everything else (the carve itself is identical in both cases).
struct FinBuf { float* G; float* s; };
FinBuf fin_buf(Carve& ws, long long nk, int n) {
  FinBuf b;
  b.G = ws.f32((size_t)nk);
  b.s = ws.f32((size_t)n);
  if (!ws.dry) {
    float* g = fin_alloc((size_t)nk);
    float* s = g ? fin_alloc((size_t)n) : nullptr;
    if (g && s) { b.G = g; b.s = s; }
  }
  return b;
}

int lin(int pro, int epi, GemmArgs& a, const sepr_lin& l, int site, hipStream_t st) {
  a.W = l.w;
  a.Wp = l.wp;
  a.bias = l.b;
  a.bf1 = (l.wp && l.planes == 1) ? 1 : 0;
  if (l.wp) return launch_gemm_x3(pro, epi, a, site, st);
  if (!l.w) return SEPR_EINVAL;
  return launch_gemm(pro, epi, a, site, st);
}
// y[M][N] = a[M][K] . W^T (+ bias); residual add when R != null (y = . + R)
int plain(const float* A, int lda, float* Y, int ldc, long long M, int N, int K, const sepr_lin& l, const float* R, hipStream_t st) {
  GemmArgs a = gemm_args_zero();
  a.M = (int)M; a.N = N; a.K = K;
  a.A = A; a.lda = lda; a.Y = Y; a.ldc = ldc; a.R = R;
  return lin(PRO_PLAIN, R ? EPI_RES : EPI_STORE, a, l, SEPR_SITE_NONE, st);
}
// y = xh . W^T + bias with xh = (x - mean) * rstd from per-row stats (affine folded into W / bias)
int normed(const float* X, int lda, const float* stats, float* Y, int ldc, long long M, int N, int K, const sepr_lin& l,
           hipStream_t st) {
  GemmArgs a = gemm_args_zero();
  a.M = (int)M; a.N = N; a.K = K;
  a.A = X; a.lda = lda; a.stats = stats; a.Y = Y; a.ldc = ldc;
  return lin(PRO_NORM, EPI_STORE, a, l, SEPR_SITE_NONE, st);
}
// dx = (dres ? dres : 0) + LayerNorm'(dA . W^T) (+ pooled gradient): the input-gradient projection behind a LayerNorm with the
// LayerNorm backward in its epilogue (EPI_LNBWD: an F-wide output is one column tile, so a tile holds whole rows) - one launch and
// no [M, F] round trip where there were a projection and launch_ln_bwd.  F > 128 (Large) keeps the two-launch form.
int dgrad_ln(const float* dA, int lda, int K, const sepr_lin& l, int a16, const float* x, const float* stats, const float* dres,
             const float* padd, int T, int Tp, int fac, float* dx, float* scratch, long long M, int F, hipStream_t st) {
  if (F <= GEMM_BN) {
    GemmArgs a = gemm_args_zero();
    a.M = (int)M; a.N = F; a.K = K;
    a.A = dA; a.lda = lda; a.a16 = a16; a.Y = dx; a.ldc = F;
    a.aux = x; a.stats = stats; a.R = dres; a.aux2 = padd; a.T = T; a.Tp = Tp; a.fac = fac;
    return lin(PRO_PLAIN, EPI_LNBWD, a, l, SEPR_SITE_NONE, st);
  }
  GemmArgs a = gemm_args_zero();
  a.M = (int)M; a.N = F; a.K = K;
  a.A = dA; a.lda = lda; a.a16 = a16; a.Y = scratch; a.ldc = F;
  SEPR_TRY(lin(PRO_PLAIN, EPI_STORE, a, l, SEPR_SITE_NONE, st));
  return launch_ln_bwd(scratch, x, stats, dres, padd, T, Tp, fac, dx, M, F, st);
}
// G[N][K] = sum_m A[m][n] B[m][k] (B optionally normalised with per-row stats), colsum[N]
int wgrad(const float* A, int lda, const float* B, int ldb, const float* stats, float* G, float* colsum, long long M, int N, int K,
          int accumulate, int x3, void* ws, size_t wsb, hipStream_t st) {
  TnArgs t = tn_args_zero();
  t.M = (int)M; t.N = N; t.K = K;
  t.A = A; t.lda = lda; t.B = B; t.ldb = ldb; t.stats = stats;
  t.G = G; t.ldg = K; t.accumulate = accumulate;
  t.colsum = colsum; t.colsum_accumulate = accumulate;
  return launch_gemm_tn(t, x3, ws, wsb, st);
}
// arithmetic of the weight-gradient contraction that goes with a projection: 0 exact f32, 1 bf16x3, 2 plain bf16
int tn_mode(const sepr_lin& l) { return !l.wp ? 0 : (l.planes == 1 ? 2 : 1); }
bool rows_ok(long long M) { return M > 0 && M <= 0x7fffffffLL / 8; }
// distinct generator streams per dropout site of one block call
sepr_u64 site_off(int site) { return (sepr_u64)site << 44; }

// =====================================================================================================================
// GCFN  (network.py:46-66)
// =====================================================================================================================
struct GcfnCtx { float *stats, *h1, *g; };
GcfnCtx gcfn_ctx(Carve& c, long long M, int F) {
  GcfnCtx k;
  k.stats = c.f32(2 * M);
  k.h1 = c.f32(6LL * F * M);
  k.g = c.f32(3LL * F * M);
  return k;
}
// Fused pair (w->fused_w1p set; F in {64, 128}): the forward is ONE launch of the fused inference kernel's TRAIN instantiation
// and keeps only the LayerNorm statistics; the backward recomputes the hidden tensor in its middle kernel
// (sepr_gcfn_bwd_fused.hip), then runs the two weight-gradient contractions, their finishers, the F-wide input-gradient
// projection and the LayerNorm backward of the unfused form.
bool gcfn_is_fused(const sepr_gcfn_tw* w, int F) { return w && w->fused_w1p && w->fused_w2p && (F == 64 || F == 128); }
// pl16 (the plain-bf16 precision, w->up.planes == 1; sizing op SEPR_TOP_GCFN_FUSED16): the forward additionally keeps the normalised
// rows as bf16 [M][F] (256 B per row at F = 128) - the backward's middle kernel stages them (and bf16(dropout1(dy)), written by a
// pre-pass) by LDS-DMA with no VALU, and both weight-gradient contractions read bf16 operands only (round 4).  pl_dry: sizing runs.
bool gcfn_pl16(const sepr_gcfn_tw* w, int pl_dry) {
  if (pl_dry >= 0) return pl_dry != 0;
  // (SEPR_TRAIN_GCFN_PLANES=0 selects the register-staged form; latched once per process: the forward, the backward and the Python
  //  mirror train_engine._gcfn_op all read the SAME value - the A/B test calls sepr_knobs_reload() between whole runs)
  return knob(SEPR_KNOB_TRAIN_GCFN_PLANES) != 0 && w && w->up.planes == 1;
}
// EGA attention of the plain-bf16 precision (sepr_lin.planes == 1): ONE bf16 MFMA per product in the forward and both backward
// kernels instead of the bf16x3 triple (round 4; SEPR_TRAIN_ATTN_ONE=0 keeps the triple - the A/B test flips it inside one process)
int attn_one(const sepr_lin& qkv) {
  return (knob(SEPR_KNOB_TRAIN_ATTN_ONE) != 0 && qkv.planes == 1) ? 1 : 0;
}
int gcfn_fused_fwd(const float* x, float* y, int n, int T, int F, const sepr_gcfn_tw* w, Carve& cx, float p, sepr_u64 seed,
                   hipStream_t st, int pl_dry = -1) {
  const long long M = (long long)n * T;
  float* stats = cx.f32(2 * M);
  const bool pl = gcfn_pl16(w, pl_dry);
  void* xh16 = pl ? cx.take((size_t)M * F * 2) : nullptr;
  if (cx.dry) return SEPR_OK;
  if (!cx.ok()) return SEPR_EWORKSPACE;
  if (!w->up.wp || !w->down_t.wp) return SEPR_EINVAL;   // the backward's middle kernel runs on the packed-bf16 cores
  GcfnFusedArgs f = {};
  f.x = x; f.y = y; f.M = (int)M; f.T = T;
  f.w1p = w->fused_w1p; f.w2p = w->fused_w2p;
  f.b2 = w->b2; f.ls = w->ls; f.eps = LN_EPS_T;
  f.train = 1; f.stats = stats;
  f.planes = w->up.planes == 1 ? 1 : 3;      // precision "bf16": the single-plane instantiation (round 4)
  f.xhat16 = xh16;
  f.drop_thr = p > 0.f ? sepr_drop_thr16(p) : 0u;
  f.drop_scale = p > 0.f ? sepr_drop_scale16(p) : 1.0f;
  f.seed = seed; f.salt = drop_salt();
  return launch_gcfn_fused(f, F, SEPR_SITE_GCFN_UP, st);
}
int gcfn_fused_bwd(const float* x, const float* dy, float* dx, int n, int T, int F, const sepr_gcfn_tw* w, const sepr_gcfn_grad* g,
                   Carve& cx, Carve& ws, float p, sepr_u64 seed, hipStream_t st, int pl_dry = -1) {
  const long long M = (long long)n * T;
  float* stats = cx.f32(2 * M);
  const bool pl = gcfn_pl16(w, pl_dry);
  void* xh16 = pl ? cx.take((size_t)M * F * 2) : nullptr;
  void* dy16 = pl ? ws.take((size_t)M * F * 2) : nullptr;
  // plain-bf16 precision: the two big intermediates are STORED as bf16 (their consumers round them to bf16 anyway)
  const bool o16 = ws.dry ? false : (w->up.planes == 1);
  void* gd = ws.take((size_t)3 * F * M * sizeof(float));       // (sized for fp32 in both cases: one workspace plan)
  void* dh1 = ws.take((size_t)6 * F * M * sizeof(float));
  float* dyp = p > 0.f ? ws.f32((long long)F * M) : nullptr;
  const FinBuf fb2 = fin_buf(ws, 3LL * F * F, F);
  float* Gr = fb2.G;
  float* s2 = fb2.s;
  const FinBuf fb1 = fin_buf(ws, 6LL * F * F, 6 * F);
  float* dWh = fb1.G;
  float* s1 = fb1.s;
  float* dxh = ws.f32((long long)F * M);
  const size_t tnb = tn_workspace_bytes((int)M, 6 * F, F) > tn_workspace_bytes((int)M, F, 3 * F) ? tn_workspace_bytes((int)M, 6 * F, F)
                                                                                                   : tn_workspace_bytes((int)M, F, 3 * F);
  void* tnw = ws.take(tnb);
  const size_t midb = gcfn_bwd_fused_ws(M, F);
  void* midw = ws.take(midb);
  if (ws.dry) return SEPR_OK;
  if (!cx.ok() || !ws.ok()) return SEPR_EWORKSPACE;
  const int x3 = tn_mode(w->up);
  if (pl) SEPR_TRY(launch_gcfn_dyplane(dy, dy16, M, F, p, seed, drop_salt(), st));
  SEPR_TRY(launch_gcfn_bwd_fused(x, stats, dy, n, T, F, w, gd, dh1, o16 ? 1 : 0, pl ? nullptr : dyp, g->dw_w, g->dw_b, p, seed, drop_salt(), midw,
                                 midb, st, xh16, dy16));
  const float* dyq = p > 0.f ? dyp : dy;
  // net2.2 + LayerScale
  {
    TnArgs t = tn_args_zero();
    t.M = (int)M; t.N = F; t.K = 3 * F;
    t.A = pl ? static_cast<const float*>(dy16) : dyq; t.lda = F; t.a16 = pl ? 1 : 0;
    t.B = static_cast<const float*>(gd); t.ldb = 3 * F; t.b16 = o16 ? 1 : 0;
    t.G = Gr; t.ldg = 3 * F; t.colsum = s2;
    SEPR_TRY(launch_gemm_tn(t, x3, tnw, tnb, st));
  }
  SEPR_TRY(launch_finish_linear_ls(Gr, s2, w->w2, w->b2, w->ls, g->w2, g->b2, g->ls, F, 3 * F, st));
  // net1 behind the LayerNorm
  {
    TnArgs t = tn_args_zero();
    t.M = (int)M; t.N = 6 * F; t.K = F;
    t.A = static_cast<const float*>(dh1); t.lda = 6 * F; t.a16 = o16 ? 1 : 0;
    if (pl) { t.B = static_cast<const float*>(xh16); t.ldb = F; t.b16 = 1; }     // the saved bf16 rows: no statistics prologue, half the bytes
    else { t.B = x; t.ldb = F; t.stats = stats; }
    t.G = dWh; t.ldg = F; t.colsum = s1;
    SEPR_TRY(launch_gemm_tn(t, x3, tnw, tnb, st));
  }
  SEPR_TRY(launch_finish_norm_linear(dWh, s1, w->w1, w->ln_g, w->ln_b, g->w1, g->b1, g->ln_g, g->ln_b, 6 * F, F, st));
  return dgrad_ln(static_cast<const float*>(dh1), 6 * F, 6 * F, w->up_t, o16 ? 1 : 0, x, stats, dy, nullptr, 0, 0, 0, dx, dxh, M, F, st);
}
int gcfn_fwd(const float* x, float* y, int n, int T, int F, const sepr_gcfn_tw* w, Carve& cx, Carve& ws, float p, sepr_u64 seed,
             hipStream_t st) {
  const long long M = (long long)n * T;
  GcfnCtx k = gcfn_ctx(cx, M, F);
  float* out = p > 0.f ? ws.f32((long long)F * M) : nullptr;
  if (cx.dry) return SEPR_OK;
  if (!cx.ok()) return SEPR_EWORKSPACE;
  if (!ws.ok()) return SEPR_EWORKSPACE;
  SEPR_TRY(launch_rowstats(x, k.stats, M, F, LN_EPS_T, st));
  SEPR_TRY(normed(x, F, k.stats, k.h1, 6 * F, M, 6 * F, F, w->up, st));                       // network.py:61
  SEPR_TRY(launch_dwglu(k.h1, k.g, n, T, F, w->dw_w, w->dw_b, st));                            // :62-65 (conv, GLU)
  if (p > 0.f) {
    SEPR_TRY(launch_dropout(k.g, k.g, 3LL * F * M, p, seed, site_off(0), st));                 // net2[1]
    SEPR_TRY(plain(k.g, 3 * F, out, F, M, F, 3 * F, w->down, nullptr, st));                    // net2[2]
    SEPR_TRY(launch_res_ls(x, out, w->ls, y, M, F, p, seed, site_off(1), st));                 // net2[3] dropout, :66
  } else {
    GemmArgs a = gemm_args_zero();
    a.M = (int)M; a.N = F; a.K = 3 * F;
    a.A = k.g; a.lda = 3 * F; a.Y = y; a.ldc = F; a.R = x; a.ls = w->ls;
    SEPR_TRY(lin(PRO_PLAIN, EPI_RES, a, w->down, SEPR_SITE_NONE, st));
  }
  return SEPR_OK;
}
int gcfn_bwd(const float* x, const float* dy, float* dx, int n, int T, int F, const sepr_gcfn_tw* w, const sepr_gcfn_grad* g, Carve& cx,
             Carve& ws, float p, sepr_u64 seed, hipStream_t st) {
  const long long M = (long long)n * T;
  GcfnCtx k = gcfn_ctx(cx, M, F);
  float* dyp = p > 0.f ? ws.f32((long long)F * M) : nullptr;
  const FinBuf fb2 = fin_buf(ws, 3LL * F * F, F);
  float* Gr = fb2.G;
  float* s2 = fb2.s;
  float* dg = ws.f32(3LL * F * M);
  float* dh1 = ws.f32(6LL * F * M);
  const FinBuf fb1 = fin_buf(ws, 6LL * F * F, 6 * F);
  float* dWh = fb1.G;
  float* s1 = fb1.s;
  float* dxh = ws.f32((long long)F * M);
  const size_t tnb = tn_workspace_bytes((int)M, 6 * F, F) > tn_workspace_bytes((int)M, F, 3 * F) ? tn_workspace_bytes((int)M, 6 * F, F)
                                                                                                   : tn_workspace_bytes((int)M, F, 3 * F);
  void* tnw = ws.take(tnb);
  const size_t midb = gcfn_mid_bwd_ws(n, T, 3 * F);
  void* midw = ws.take(midb);
  if (ws.dry) return SEPR_OK;
  if (!cx.ok() || !ws.ok()) return SEPR_EWORKSPACE;
  const int x3 = tn_mode(w->up);
  const float* dyq = dy;
  if (p > 0.f) {
    SEPR_TRY(launch_dropout(dy, dyp, (long long)F * M, p, seed, site_off(1), st));
    dyq = dyp;
  }
  // net2.2 + LayerScale: raw contraction, then dW2 / db2 / dls
  SEPR_TRY(wgrad(dyq, F, k.g, 3 * F, nullptr, Gr, s2, M, F, 3 * F, 0, x3, tnw, tnb, st));
  SEPR_TRY(launch_finish_linear_ls(Gr, s2, w->w2, w->b2, w->ls, g->w2, g->b2, g->ls, F, 3 * F, st));
  SEPR_TRY(plain(dyq, F, dg, 3 * F, M, 3 * F, F, w->down_t, nullptr, st));
  // GLU + depthwise conv (the dropout mask of the gated tensor is applied while dg is read)
  SEPR_TRY(launch_gcfn_mid_bwd(k.h1, dg, dh1, n, T, 3 * F, w->dw_w, w->dw_b, g->dw_w, g->dw_b, p, seed, site_off(0), midw, midb, st));
  // net1: LayerNorm-folded projection
  SEPR_TRY(wgrad(dh1, 6 * F, x, F, k.stats, dWh, s1, M, 6 * F, F, 0, x3, tnw, tnb, st));
  SEPR_TRY(launch_finish_norm_linear(dWh, s1, w->w1, w->ln_g, w->ln_b, g->w1, g->b1, g->ln_g, g->ln_b, 6 * F, F, st));
  return dgrad_ln(dh1, 6 * F, 6 * F, w->up_t, 0, x, k.stats, dy, nullptr, 0, 0, 0, dx, dxh, M, F, st);
}

// =====================================================================================================================
// CLA  (network.py:159-187), train-mode BatchNorm
// =====================================================================================================================
// Plain-bf16 precision (round 4): four of the block's intermediates are STORED as bf16 - c (conv output), d = gelu(bn(z)), dz and da
// (the gradients w.r.t. linear2's / linear1's outputs).  Every reader of those four is an MFMA operand loader that rounds to bf16
// anyway (linear2 / linear3 and their weight gradients, the two input-gradient projections), so results are bit-identical to the
// fp32-stored form (tested) while ~5.4 KB of the block's ~27 KB of HBM traffic per row disappear.  SEPR_TRAIN_CLA16=0: fp32 (A/B, test).
bool cla_h16(const sepr_cla_tw* w, int F, int K) {
  return knob(SEPR_KNOB_TRAIN_CLA16) != 0 && w && w->l1.wp && w->l1.planes == 1 && F % 128 == 0 && K == 65;
}
constexpr unsigned CLA_DROP_SITE = 3u;   // 16-bit generator site of the CLA output dropout (0, 1: fused GCFN; 2: attention probabilities)
struct ClaCtx { float *stats, *a, *u, *c, *z, *bn, *d; };
ClaCtx cla_ctx(Carve& cx, long long M, int F) {
  ClaCtx k;
  k.stats = cx.f32(2 * M);
  k.a = cx.f32(2LL * F * M);    // linear1 output (pre-GLU)
  k.u = cx.f32((long long)F * M);
  k.c = cx.f32((long long)F * M);
  k.z = cx.f32(2LL * F * M);    // linear2 output (pre-BatchNorm)
  k.bn = cx.f32(4 * F);         // batch mean [2F], rstd [2F]
  k.d = cx.f32(2LL * F * M);    // gelu(bn(z)) (post-dropout input of linear3 when p > 0 ... dropout sits after linear3)
  return k;
}
int cla_fwd(const float* x, float* y, int n, int T, int F, int K, const sepr_cla_tw* w, Carve& cx, Carve& ws, float p, sepr_u64 seed,
            hipStream_t st) {
  const long long M = (long long)n * T;
  ClaCtx k = cla_ctx(cx, M, F);
  const size_t csb = colstats_ws(M, 2 * F);
  void* csw = ws.take(csb);
  if (cx.dry) return SEPR_OK;
  if (!cx.ok() || !ws.ok()) return SEPR_EWORKSPACE;
  SEPR_TRY(launch_rowstats(x, k.stats, M, F, LN_EPS_T, st));
  {   // network.py:175-177 in ONE launch (round 4): LayerNorm prologue, F -> 2F, GLU in the epilogue, which stores BOTH the
      // pre-activation rows (the GLU backward needs value and gate) and the gated rows
    GemmArgs a = gemm_args_zero();
    a.M = (int)M; a.N = 2 * F; a.K = F;
    a.A = x; a.lda = F; a.stats = k.stats; a.Y = k.u; a.ldc = F; a.Ysave = k.a;
    SEPR_TRY(lin(PRO_NORM, EPI_GLUSAVE, a, w->l1, SEPR_SITE_NONE, st));
  }
  const bool h16 = cla_h16(w, F, K);
  if (h16) {
    SEPR_TRY(launch_dwconv_same16(k.u, k.c, n, T, F, K, w->dw_w, w->dw_b, st));               // :178-180, c as bf16
    GemmArgs a = gemm_args_zero();
    a.M = (int)M; a.N = 2 * F; a.K = F;
    a.A = k.c; a.lda = F; a.a16 = 1; a.Y = k.z; a.ldc = 2 * F;
    SEPR_TRY(lin(PRO_PLAIN, EPI_STORE, a, w->l2, SEPR_SITE_NONE, st));                        // :181
  } else {
    SEPR_TRY(launch_dwconv_same(k.u, k.c, n, T, F, K, w->dw_w, w->dw_b, st));                 // :178-180
    SEPR_TRY(plain(k.c, F, k.z, 2 * F, M, 2 * F, F, w->l2, nullptr, st));                     // :181
  }
  SEPR_TRY(launch_colstats(k.z, M, 2 * F, BN_EPS_T, BN_MOM, k.bn, w->bn_rm, w->bn_rv, csw, csb, st));   // :183 (batch statistics)
  SEPR_TRY(launch_bn_gelu_fwd(k.z, k.bn, w->bn_g, w->bn_b, k.d, M, 2 * F, st, h16 ? 1 : 0));  // :183,185 (GELU); d as bf16 when h16
  if (p > 0.f) {   // :185-187 with the dropout of linear3[2] in the projection's epilogue (round 4; 16-bit generator, site CLA_DROP_SITE)
    GemmArgs a = gemm_args_zero();
    a.M = (int)M; a.N = F; a.K = 2 * F;
    a.A = k.d; a.lda = 2 * F; a.a16 = h16 ? 1 : 0; a.Y = y; a.ldc = F; a.R = x; a.ls = w->ls;
    a.drop_thr = sepr_drop_thr16(p); a.drop_scale = sepr_drop_scale16(p); a.drop_seed = seed; a.drop_salt = drop_salt(); a.drop_site = CLA_DROP_SITE;
    SEPR_TRY(lin(PRO_PLAIN, EPI_RESDROP, a, w->l3, SEPR_SITE_NONE, st));
  } else {
    GemmArgs a = gemm_args_zero();
    a.M = (int)M; a.N = F; a.K = 2 * F;
    a.A = k.d; a.lda = 2 * F; a.a16 = h16 ? 1 : 0; a.Y = y; a.ldc = F; a.R = x; a.ls = w->ls;
    SEPR_TRY(lin(PRO_PLAIN, EPI_RES, a, w->l3, SEPR_SITE_NONE, st));                          // :185-187
  }
  return SEPR_OK;
}
int cla_bwd(const float* x, const float* dy, float* dx, int n, int T, int F, int K, const sepr_cla_tw* w, const sepr_cla_grad* g,
            Carve& cx, Carve& ws, float p, sepr_u64 seed, hipStream_t st) {
  const long long M = (long long)n * T;
  ClaCtx k = cla_ctx(cx, M, F);
  float* dyp = p > 0.f ? ws.f32((long long)F * M) : nullptr;
  const FinBuf fb3 = fin_buf(ws, 2LL * F * F, 2 * F), fb1 = fin_buf(ws, 2LL * F * F, 2 * F);      // linear3 (+ LayerScale), linear1 (behind the LayerNorm)
  float* Gr = fb3.G;
  float* s = fb3.s;
  float* dd = ws.f32(2LL * F * M);      // d(gelu out), later dz in place, later da
  float* dc = ws.f32((long long)F * M);
  float* du = ws.f32((long long)F * M);
  float* h16buf = ws.f32(2LL * F * M);  // plain-bf16 precision: dz [M][2F] bf16 in its first half, da [M][2F] bf16 in its second half
  float* dxh = dc;                      // dc is dead once du / the conv weight gradient are formed
  const size_t tnb = tn_workspace_bytes((int)M, 2 * F, F);
  void* tnw = ws.take(tnb);
  const size_t csb = colstats_ws(M, 2 * F);
  void* csw = ws.take(csb);
  const size_t wgb = dwconv_wgrad_ws(n, T, F, K);
  void* wgw = ws.take(wgb);
  if (ws.dry) return SEPR_OK;
  if (!cx.ok() || !ws.ok()) return SEPR_EWORKSPACE;
  const int x3 = tn_mode(w->l1);
  const float* dyq = dy;
  if (p > 0.f) {
    SEPR_TRY(launch_dropout16(dy, dyp, M, F, p, seed, CLA_DROP_SITE, st));                    // the mask of cla_fwd's EPI_RESDROP
    dyq = dyp;
  }
  if (cla_h16(w, F, K)) {   // the bf16-stored form: c, d from the forward; dz, da here (see cla_h16)
    float* dz16 = h16buf;
    float* da16 = h16buf + (long long)F * M;                    // (2F bf16 per row = F floats per row)
    {
      TnArgs t = tn_args_zero();
      t.M = (int)M; t.N = F; t.K = 2 * F;
      t.A = dyq; t.lda = F; t.B = k.d; t.ldb = 2 * F; t.b16 = 1;
      t.G = Gr; t.ldg = 2 * F; t.colsum = s;
      SEPR_TRY(launch_gemm_tn(t, x3, tnw, tnb, st));
    }
    SEPR_TRY(launch_finish_linear_ls(Gr, s, w->w3, w->b3, w->ls, g->w3, g->b3, g->ls, F, 2 * F, st));
    SEPR_TRY(plain(dyq, F, dd, 2 * F, M, 2 * F, F, w->l3_t, nullptr, st));
    SEPR_TRY(launch_bn_gelu_bwd(dd, k.z, k.bn, w->bn_g, w->bn_b, dz16, g->bn_g, g->bn_b, M, 2 * F, csw, csb, st, 1));
    {
      TnArgs t = tn_args_zero();
      t.M = (int)M; t.N = 2 * F; t.K = F;
      t.A = dz16; t.lda = 2 * F; t.a16 = 1; t.B = k.c; t.ldb = F; t.b16 = 1;
      t.G = g->w2; t.ldg = F; t.accumulate = 1; t.colsum = g->b2; t.colsum_accumulate = 1;
      SEPR_TRY(launch_gemm_tn(t, x3, tnw, tnb, st));                                                              // linear2 (direct)
    }
    {
      GemmArgs a = gemm_args_zero();
      a.M = (int)M; a.N = F; a.K = 2 * F;
      a.A = dz16; a.lda = 2 * F; a.a16 = 1; a.Y = dc; a.ldc = F;
      SEPR_TRY(lin(PRO_PLAIN, EPI_STORE, a, w->l2_t, SEPR_SITE_NONE, st));
    }
    SEPR_TRY(launch_dwconv_wgrad(k.u, dc, n, T, F, K, g->dw_w, g->dw_b, wgw, wgb, st));
    SEPR_TRY(launch_dwconv_same_glu_bwd(dc, k.a, da16, n, T, F, K, w->dw_wf, w->zeros, st, 1));
    {
      TnArgs t = tn_args_zero();
      t.M = (int)M; t.N = 2 * F; t.K = F;
      t.A = da16; t.lda = 2 * F; t.a16 = 1; t.B = x; t.ldb = F; t.stats = k.stats;
      t.G = fb1.G; t.ldg = F; t.colsum = fb1.s;
      SEPR_TRY(launch_gemm_tn(t, x3, tnw, tnb, st));
    }
    SEPR_TRY(launch_finish_norm_linear(fb1.G, fb1.s, w->w1, w->ln_g, w->ln_b, g->w1, g->b1, g->ln_g, g->ln_b, 2 * F, F, st));
    return dgrad_ln(da16, 2 * F, 2 * F, w->l1_t, 1, x, k.stats, dy, nullptr, 0, 0, 0, dx, dxh, M, F, st);
  }
  SEPR_TRY(wgrad(dyq, F, k.d, 2 * F, nullptr, Gr, s, M, F, 2 * F, 0, x3, tnw, tnb, st));
  SEPR_TRY(launch_finish_linear_ls(Gr, s, w->w3, w->b3, w->ls, g->w3, g->b3, g->ls, F, 2 * F, st));
  SEPR_TRY(plain(dyq, F, dd, 2 * F, M, 2 * F, F, w->l3_t, nullptr, st));
  SEPR_TRY(launch_bn_gelu_bwd(dd, k.z, k.bn, w->bn_g, w->bn_b, dd, g->bn_g, g->bn_b, M, 2 * F, csw, csb, st));   // dd := dz
  SEPR_TRY(wgrad(dd, 2 * F, k.c, F, nullptr, g->w2, g->b2, M, 2 * F, F, 1, x3, tnw, tnb, st));                   // linear2 (direct)
  SEPR_TRY(plain(dd, 2 * F, dc, F, M, F, 2 * F, w->l2_t, nullptr, st));
  SEPR_TRY(launch_dwconv_wgrad(k.u, dc, n, T, F, K, g->dw_w, g->dw_b, wgw, wgb, st));
  wgrad_join(st);   // dd (dz) is re-used for da below while linear2's contraction - on the weight-gradient side stream, if one is registered - may still read it
  if (F % 128 == 0 && K == 65) {   // correlation with reversed taps, GLU backward in its epilogue (round 4): dd := da [M][2F]
    SEPR_TRY(launch_dwconv_same_glu_bwd(dc, k.a, dd, n, T, F, K, w->dw_wf, w->zeros, st));
  } else {
    SEPR_TRY(launch_dwconv_same(dc, du, n, T, F, K, w->dw_wf, w->zeros, st));                 // correlation with reversed taps
    SEPR_TRY(launch_glu_bwd(du, k.a, dd, M, F, st));                                           // dd := da [M][2F]
  }
  SEPR_TRY(wgrad(dd, 2 * F, x, F, k.stats, fb1.G, fb1.s, M, 2 * F, F, 0, x3, tnw, tnb, st));
  SEPR_TRY(launch_finish_norm_linear(fb1.G, fb1.s, w->w1, w->ln_g, w->ln_b, g->w1, g->b1, g->ln_g, g->ln_b, 2 * F, F, st));
  return dgrad_ln(dd, 2 * F, 2 * F, w->l1_t, 0, x, k.stats, dy, nullptr, 0, 0, 0, dx, dxh, M, F, st);
}

// =====================================================================================================================
// MultiHeadAttention pieces shared by EGA and SpkAttention
// =====================================================================================================================
// d(linear_out * LayerScale): Graw / finish / dO = dy' . (ls * Wo)
int mha_out_bwd(const float* dyq, const float* o, float* dO, long long M, int F, const sepr_mha_tw* w, const sepr_mha_grad* g, float* Gr,
                float* s, int x3, void* tnw, size_t tnb, hipStream_t st) {
  SEPR_TRY(wgrad(dyq, F, o, F, nullptr, Gr, s, M, F, F, 0, x3, tnw, tnb, st));
  SEPR_TRY(launch_finish_linear_ls(Gr, s, w->wo, w->bo, w->ls, g->wo, g->bo, g->ls, F, F, st));
  return plain(dyq, F, dO, F, M, F, F, w->out_t, nullptr, st);
}
// d(q/k/v projection behind the LayerNorm): dWh [3F][F] -> three finishers (shared dgamma / dbeta), dxh = dqkv . (Wqkv * gamma)
int mha_qkv_bwd(const float* dqkv, const float* xin, const float* stats, float* dxh, long long M, int F, const sepr_mha_tw* w,
                const sepr_mha_grad* g, float* dWh, float* s, int x3, void* tnw, size_t tnb, const float* dres, float* dx, hipStream_t st) {
  SEPR_TRY(wgrad(dqkv, 3 * F, xin, F, stats, dWh, s, M, 3 * F, F, 0, x3, tnw, tnb, st));
  float* gw[3] = {g->wq, g->wk, g->wv};
  float* gb[3] = {g->bq, g->bk, g->bv};
  for (int i = 0; i < 3; ++i)
    SEPR_TRY(launch_finish_norm_linear(dWh + (long long)i * F * F, s + i * F, w->wqkv + (long long)i * F * F, w->ln_g, w->ln_b, gw[i],
                                       gb[i], g->ln_g, g->ln_b, F, F, st));
  // dx = dres + LayerNorm'(dqkv . (Wqkv gamma))   (dxh: scratch of the two-launch form for F > 128)
  return dgrad_ln(dqkv, 3 * F, 3 * F, w->qkv_t, 0, xin, stats, dres, nullptr, 0, 0, 0, dx, dxh, M, F, st);
}

// =====================================================================================================================
// EGA  (network.py:126-155)
// =====================================================================================================================
// The attention of the packed-bf16 precisions runs flash-style on the bf16 MFMA (sepr_attention.hip TRAIN instantiation,
// sepr_train_attn_x3.hip): its context keeps one log-sum-exp per query row where the exact-f32 VALU kernels keep the [Tp, Tp]
// probabilities (k.P below is then [n*H, Tp]).  SEPR_TRAIN_ATTN_VALU=1 forces the VALU kernels (A/B, tests).
bool ega_mfma(const sepr_ega_tw* w, int F, int H) {
  static const bool force_valu = [] {
    const char* e = getenv("SEPR_TRAIN_ATTN_VALU");
    return e && e[0] == '1';
  }();
  const int dk = H > 0 ? F / H : 0;
  return w && w->attn.qkv.wp && !force_valu && (dk == 16 || dk == 32);
}
struct EgaCtx { float *stats, *stats_p, *xd, *qkv, *P, *o, *att, *zg; };
EgaCtx ega_ctx(Carve& cx, int n, int T, int Tp, int F, int H, bool mfma = false) {
  const long long M = (long long)n * T, Mp = (long long)n * Tp;
  EgaCtx k;
  k.stats = cx.f32(2 * M);
  k.stats_p = cx.f32(2 * Mp);
  k.xd = cx.f32((long long)F * Mp);
  k.qkv = cx.f32(3LL * F * Mp);
  k.P = cx.f32(mfma ? (long long)n * H * Tp : (long long)n * H * Tp * Tp);
  k.o = cx.f32((long long)F * Mp);
  k.att = cx.f32((long long)F * Mp);
  k.zg = cx.f32((long long)F * M);
  return k;
}
int ega_fwd(const float* x, float* y, int n, int T, int Tp, int F, int H, const sepr_ega_tw* w, Carve& cx, Carve& ws, float p,
            sepr_u64 seed, hipStream_t st, int mfma_dry = -1) {
  const long long M = (long long)n * T, Mp = (long long)n * Tp;
  const int fac = T / Tp;
  const bool mfma = mfma_dry >= 0 ? mfma_dry != 0 : ega_mfma(w, F, H);
  EgaCtx k = ega_ctx(cx, n, T, Tp, F, H, mfma);
  float* tmp = p > 0.f ? ws.f32((long long)F * Mp) : nullptr;
  if (cx.dry) return SEPR_OK;
  if (!cx.ok() || !ws.ok()) return SEPR_EWORKSPACE;
  const float* xp = x;
  if (fac > 1) {
    SEPR_TRY(launch_pool(x, k.xd, n, Tp, fac, F, st));                                          // network.py:146
    xp = k.xd;
  }
  SEPR_TRY(launch_rowstats(xp, k.stats_p, Mp, F, LN_EPS_T, st));
  SEPR_TRY(normed(xp, F, k.stats_p, k.qkv, 3 * F, Mp, 3 * F, F, w->attn.qkv, st));             // :99-102
  if (mfma) SEPR_TRY(launch_relattn_x3_train_fwd(k.qkv, k.o, k.P, n, Tp, F, H, w->pe_k, w->maxlen, p, seed, drop_salt(), st, attn_one(w->attn.qkv)));
  else SEPR_TRY(launch_relattn_train_fwd(k.qkv, k.o, k.P, n, Tp, F, H, w->pe_k, w->maxlen, p, seed, site_off(0), st));   // :106-122
  if (p > 0.f) {
    SEPR_TRY(plain(k.o, F, tmp, F, Mp, F, F, w->attn.out, nullptr, st));                        // :124 linear_out
    SEPR_TRY(launch_res_ls(nullptr, tmp, w->attn.ls, k.att, Mp, F, p, seed, site_off(1), st)); //      dropout, LayerScale
  } else {
    GemmArgs a = gemm_args_zero();
    a.M = (int)Mp; a.N = F; a.K = F;
    a.A = k.o; a.lda = F; a.Y = k.att; a.ldc = F; a.R = nullptr; a.ls = w->attn.ls;
    SEPR_TRY(lin(PRO_PLAIN, EPI_RES, a, w->attn.out, SEPR_SITE_NONE, st));                      // :124
  }
  SEPR_TRY(launch_rowstats(x, k.stats, M, F, LN_EPS_T, st));
  SEPR_TRY(normed(x, F, k.stats, k.zg, F, M, F, F, w->gate, st));                               // :132-134
  SEPR_TRY(launch_gate_fwd(x, k.zg, k.att, y, n, T, Tp, F, st));                                // :135,151-153
  return SEPR_OK;
}
int ega_bwd(const float* x, const float* dy, float* dx, int n, int T, int Tp, int F, int H, const sepr_ega_tw* w, const sepr_ega_grad* g,
            Carve& cx, Carve& ws, float p, sepr_u64 seed, hipStream_t st, int mfma_dry = -1) {
  const long long M = (long long)n * T, Mp = (long long)n * Tp;
  const int fac = T / Tp;
  const bool mfma = mfma_dry >= 0 ? mfma_dry != 0 : ega_mfma(w, F, H);
  EgaCtx k = ega_ctx(cx, n, T, Tp, F, H, mfma);
  float* dzg = ws.f32((long long)F * M);
  float* datt = ws.f32((long long)F * Mp);
  float* dO = ws.f32((long long)F * Mp);
  float* dqkv = ws.f32(3LL * F * Mp);
  float* dxh_p = ws.f32((long long)F * Mp);
  float* dxd = ws.f32((long long)F * Mp);
  const FinBuf fbo = fin_buf(ws, (long long)F * F, F), fbq = fin_buf(ws, 3LL * F * F, 3 * F), fbg = fin_buf(ws, (long long)F * F, F);   // out, q/k/v, gate
  float* dxh = ws.f32((long long)F * M);
  size_t tnb = tn_workspace_bytes((int)M, F, F);
  if (tn_workspace_bytes((int)Mp, 3 * F, F) > tnb) tnb = tn_workspace_bytes((int)Mp, 3 * F, F);
  void* tnw = ws.take(tnb);
  const size_t atb = mfma ? relattn_x3_bwd_ws(n, Tp, F, H) : relattn_train_ws(n, Tp, F, H);
  void* atw = ws.take(atb);
  if (ws.dry) return SEPR_OK;
  if (!cx.ok() || !ws.ok()) return SEPR_EWORKSPACE;
  const int x3 = tn_mode(w->gate);
  const float* xp = fac > 1 ? k.xd : x;
  // gate: y = x + sigmoid(zg) * up(att)
  SEPR_TRY(launch_gate_bwd(dy, k.zg, k.att, dzg, datt, n, T, Tp, F, st));
  if (p > 0.f) SEPR_TRY(launch_dropout(datt, datt, (long long)F * Mp, p, seed, site_off(1), st));   // attention-output dropout mask
  // attention branch first (its input gradient is added while the gate branch's LayerNorm backward writes dx)
  SEPR_TRY(mha_out_bwd(datt, k.o, dO, Mp, F, &w->attn, &g->attn, fbo.G, fbo.s, x3, tnw, tnb, st));
  if (mfma) SEPR_TRY(launch_relattn_x3_bwd(k.qkv, k.P, k.o, dO, dqkv, g->pe_k, n, Tp, F, H, w->pe_k, w->maxlen, p, seed, drop_salt(), atw, atb, st,
                                           w->attn.qkv.planes == 1 ? 1 : 0, attn_one(w->attn.qkv)));
  else SEPR_TRY(launch_relattn_bwd(k.qkv, k.P, k.o, dO, dqkv, g->pe_k, n, Tp, F, H, w->pe_k, w->maxlen, p, seed, site_off(0), atw, atb, st));
  SEPR_TRY(mha_qkv_bwd(dqkv, xp, k.stats_p, dxh_p, Mp, F, &w->attn, &g->attn, fbq.G, fbq.s, x3, tnw, tnb, nullptr, dxd, st));
  // gate projection behind its own LayerNorm
  SEPR_TRY(wgrad(dzg, F, x, F, k.stats, fbg.G, fbg.s, M, F, F, 0, x3, tnw, tnb, st));
  SEPR_TRY(launch_finish_norm_linear(fbg.G, fbg.s, w->gate_w, w->gate_ln_g, w->gate_ln_b, g->gate_w, g->gate_b, g->gate_ln_g, g->gate_ln_b, F, F,
                                     st));
  // dx = dy + LN'(dzg . (Wgate gamma)) + avg-pool backward of dxd
  return dgrad_ln(dzg, F, F, w->gate_t, 0, x, k.stats, dy, dxd, T, Tp, fac, dx, dxh, M, F, st);
}

// =====================================================================================================================
// SpkAttention up to its feed-forward  (network.py:233-247)
// =====================================================================================================================
struct SpkCtx { float *stats, *qkv, *o; };
SpkCtx spk_ctx(Carve& cx, long long M, int F) {
  SpkCtx k;
  k.stats = cx.f32(2 * M);
  k.qkv = cx.f32(3LL * F * M);
  k.o = cx.f32((long long)F * M);
  return k;
}
int spk_fwd(const float* x, float* y, int nS, int S, int T, int F, int H, const sepr_mha_tw* w, Carve& cx, Carve& ws, float p,
            sepr_u64 seed, hipStream_t st) {
  const long long M = (long long)nS * T;
  SpkCtx k = spk_ctx(cx, M, F);
  float* out = p > 0.f ? ws.f32((long long)F * M) : nullptr;
  if (cx.dry) return SEPR_OK;
  if (!cx.ok() || !ws.ok()) return SEPR_EWORKSPACE;
  SEPR_TRY(launch_rowstats(x, k.stats, M, F, LN_EPS_T, st));
  SEPR_TRY(normed(x, F, k.stats, k.qkv, 3 * F, M, 3 * F, F, w->qkv, st));
  if (p > 0.f) {
    SEPR_TRY(launch_spkmix_train_fwd(k.qkv, k.o, nS / S, S, T, F, H, p, seed, site_off(0), st));
    SEPR_TRY(plain(k.o, F, out, F, M, F, F, w->out, nullptr, st));
    return launch_res_ls(x, out, w->ls, y, M, F, p, seed, site_off(1), st);
  }
  SEPR_TRY(launch_spkmix(k.qkv, k.o, nS / S, S, T, F, H, st));
  GemmArgs a = gemm_args_zero();
  a.M = (int)M; a.N = F; a.K = F;
  a.A = k.o; a.lda = F; a.Y = y; a.ldc = F; a.R = x; a.ls = w->ls;
  return lin(PRO_PLAIN, EPI_RES, a, w->out, SEPR_SITE_NONE, st);
}
int spk_bwd(const float* x, const float* dy, float* dx, int nS, int S, int T, int F, int H, const sepr_mha_tw* w, const sepr_mha_grad* g,
            Carve& cx, Carve& ws, float p, sepr_u64 seed, hipStream_t st) {
  const long long M = (long long)nS * T;
  SpkCtx k = spk_ctx(cx, M, F);
  float* dO = ws.f32((long long)F * M);
  float* dqkv = ws.f32(3LL * F * M);
  const FinBuf fbo = fin_buf(ws, (long long)F * F, F), fbq = fin_buf(ws, 3LL * F * F, 3 * F);      // out, q/k/v
  float* dxh = dO;                      // dO is dead after the speaker-mix backward
  float* dyp = p > 0.f ? ws.f32((long long)F * M) : nullptr;
  const size_t tnb = tn_workspace_bytes((int)M, 3 * F, F);
  void* tnw = ws.take(tnb);
  if (ws.dry) return SEPR_OK;
  if (!cx.ok() || !ws.ok()) return SEPR_EWORKSPACE;
  const int x3 = tn_mode(w->qkv);
  const float* dyq = dy;
  if (p > 0.f) {
    SEPR_TRY(launch_dropout(dy, dyp, (long long)F * M, p, seed, site_off(1), st));
    dyq = dyp;
  }
  SEPR_TRY(mha_out_bwd(dyq, k.o, dO, M, F, w, g, fbo.G, fbo.s, x3, tnw, tnb, st));
  SEPR_TRY(launch_spkmix_bwd(k.qkv, dO, dqkv, nS / S, S, T, F, H, p, seed, site_off(0), st));
  return mha_qkv_bwd(dqkv, x, k.stats, dxh, M, F, w, g, fbq.G, fbq.s, x3, tnw, tnb, dy, dx, st);
}

// =====================================================================================================================
// DownConvLayer (module.py:63-78), train-mode BatchNorm
// =====================================================================================================================
struct DownCtx { float *c, *bn; };
int down_To(int T, int K) { return (T + 2 * ((K - 1) / 2) - K) / 2 + 1; }
DownCtx down_ctx(Carve& cx, int n, int To, int F) {
  DownCtx k;
  k.c = cx.f32((long long)n * To * F);
  k.bn = cx.f32(2 * F);
  return k;
}
int down_fwd(const float* x, float* y, int n, int T, int F, int K, const sepr_down_tw* w, Carve& cx, Carve& ws, hipStream_t st) {
  const int To = down_To(T, K);
  DownCtx k = down_ctx(cx, n, To, F);
  const size_t csb = colstats_ws((long long)n * To, F);
  void* csw = ws.take(csb);
  if (cx.dry) return SEPR_OK;
  if (!cx.ok() || !ws.ok()) return SEPR_EWORKSPACE;
  SEPR_TRY(launch_downconv_pre(x, k.c, n, T, To, F, K, w->w, w->b, st));                                        // module.py:74
  SEPR_TRY(launch_colstats(k.c, (long long)n * To, F, BN_EPS_T, BN_MOM, k.bn, w->bn_rm, w->bn_rv, csw, csb, st));   // :75
  return launch_bn_gelu_fwd(k.c, k.bn, w->bn_g, w->bn_b, y, (long long)n * To, F, st);                          // :75-76
}
int down_bwd(const float* x, const float* dy, float* dx, int n, int T, int F, int K, const sepr_down_tw* w, const sepr_down_grad* g,
             Carve& cx, Carve& ws, hipStream_t st) {
  const int To = down_To(T, K);
  DownCtx k = down_ctx(cx, n, To, F);
  float* dc = ws.f32((long long)n * To * F);
  const size_t csb = colstats_ws((long long)n * To, F);
  void* csw = ws.take(csb);
  const size_t wgb = downconv_bwd_ws(n, T, F, K);
  void* wgw = ws.take(wgb);
  if (ws.dry) return SEPR_OK;
  if (!cx.ok() || !ws.ok()) return SEPR_EWORKSPACE;
  SEPR_TRY(launch_bn_gelu_bwd(dy, k.c, k.bn, w->bn_g, w->bn_b, dc, g->bn_g, g->bn_b, (long long)n * To, F, csw, csb, st));
  return launch_downconv_bwd(x, dc, dx, n, T, To, F, K, w->w, g->w, g->b, wgw, wgb, st);
}

// =====================================================================================================================
// SpkSplitStage (module.py:110-125)
// =====================================================================================================================
struct SplitCtx { float *a, *z, *v, *stats; };
SplitCtx split_ctx(Carve& cx, int B, int S, int T, int F) {
  const long long M = (long long)B * T;
  SplitCtx k;
  k.a = cx.f32(4LL * F * S * M);     // linear.0 output (pre-GLU)
  k.z = cx.f32(2LL * F * S * M);
  k.v = cx.f32((long long)F * S * M);   // linear.2 output in [B*S, T, F] layout (pre-GroupNorm)
  k.stats = cx.f32(2LL * B * S);
  return k;
}
int split_fwd(const float* x, float* y, int B, int S, int T, int F, float eps, const sepr_split_tw* w, Carve& cx, Carve& ws,
              hipStream_t st) {
  const long long M = (long long)B * T;
  SplitCtx k = split_ctx(cx, B, S, T, F);
  const int n_out = B * S;
  const long long count = (long long)T * F;
  const int nchunk = gn_chunks(count);
  double* part = static_cast<double*>(ws.take((size_t)n_out * nchunk * 2 * sizeof(double)));
  if (cx.dry) return SEPR_OK;
  if (!cx.ok() || !ws.ok()) return SEPR_EWORKSPACE;
  SEPR_TRY(plain(x, F, k.a, 4 * F * S, M, 4 * F * S, F, w->l1, nullptr, st));                   // module.py:114
  SEPR_TRY(launch_glu_fwd(k.a, k.z, M, 2 * F * S, st));                                          // :115 (channel dim == last dim here)
  {
    GemmArgs a = gemm_args_zero();
    a.M = (int)M; a.N = F * S; a.K = 2 * F * S;
    a.A = k.z; a.lda = 2 * F * S; a.Y = k.v; a.ldc = F; a.T = T; a.S = S; a.Fs = F;
    SEPR_TRY(lin(PRO_PLAIN, EPI_SPLIT, a, w->l2, SEPR_SITE_NONE, st));                           // :116,123
  }
  SEPR_TRY(launch_gn_partial(k.v, part, n_out, count, nchunk, st));
  SEPR_TRY(launch_gn_finalize(part, n_out, nchunk, count, eps, k.stats, st));
  return launch_gn_apply_oop(k.v, k.stats, w->gn_g, w->gn_b, y, n_out, T, F, st);                // :124
}
int split_bwd(const float* x, const float* dy, float* dx, int dx_acc, int B, int S, int T, int F, const sepr_split_tw* w,
              const sepr_split_grad* g, Carve& cx, Carve& ws, hipStream_t st) {
  const long long M = (long long)B * T;
  SplitCtx k = split_ctx(cx, B, S, T, F);
  float* dv = ws.f32((long long)F * S * M);      // [M][S*F] (channel-split inverted)
  float* dz = ws.f32(2LL * F * S * M);
  float* da = ws.f32(4LL * F * S * M);
  const size_t gnb = gn_bwd_ws(B * S, T, F);
  void* gnw = ws.take(gnb);
  size_t tnb = tn_workspace_bytes((int)M, F * S, 2 * F * S);
  if (tn_workspace_bytes((int)M, 4 * F * S, F) > tnb) tnb = tn_workspace_bytes((int)M, 4 * F * S, F);
  void* tnw = ws.take(tnb);
  if (ws.dry) return SEPR_OK;
  if (!cx.ok() || !ws.ok()) return SEPR_EWORKSPACE;
  const int x3 = tn_mode(w->l1);
  SEPR_TRY(launch_gn_bwd(dy, k.v, k.stats, w->gn_g, dv, g->gn_g, g->gn_b, B * S, T, F, S, gnw, gnb, st));
  SEPR_TRY(wgrad(dv, F * S, k.z, 2 * F * S, nullptr, g->w2, g->b2, M, F * S, 2 * F * S, 1, x3, tnw, tnb, st));
  SEPR_TRY(plain(dv, F * S, dz, 2 * F * S, M, 2 * F * S, F * S, w->l2_t, nullptr, st));
  SEPR_TRY(launch_glu_bwd(dz, k.a, da, M, 2 * F * S, st));
  SEPR_TRY(wgrad(da, 4 * F * S, x, F, nullptr, g->w1, g->b1, M, 4 * F * S, F, 1, x3, tnw, tnb, st));
  return plain(da, 4 * F * S, dx, F, M, F, 4 * F * S, w->l1_t, dx_acc ? dx : nullptr, st);
}

}  // namespace
}  // namespace sepr

using namespace sepr;

// ---------------------------------------------------------------------------------------------------------------------
// extern "C" wrappers
// ---------------------------------------------------------------------------------------------------------------------
#define SEPR_ST static_cast<hipStream_t>(stream)

extern "C" int sepr_gcfn_train_fwd(const float* x, float* y, int n, int T, int F, const sepr_gcfn_tw* w, void* ctx, size_t ctx_bytes,
                                   void* ws, size_t ws_bytes, float p_drop, sepr_u64 seed, sepr_stream_t stream) {
  if (!x || !y || !w || n <= 0 || T <= 0 || F <= 0 || F % 32 || !rows_ok((long long)n * T) || x == y) return SEPR_EINVAL;
  DropSaltScope salt_scope(w->seed_salt);
  Carve cx(ctx, ctx_bytes, false), wk(ws, ws_bytes, false);
  if (gcfn_is_fused(w, F)) return gcfn_fused_fwd(x, y, n, T, F, w, cx, p_drop, seed, SEPR_ST);
  return gcfn_fwd(x, y, n, T, F, w, cx, wk, p_drop, seed, SEPR_ST);
}
extern "C" int sepr_gcfn_bwd(const float* x, const float* dy, float* dx, int n, int T, int F, const sepr_gcfn_tw* w,
                             const sepr_gcfn_grad* g, const void* ctx, size_t ctx_bytes, void* ws, size_t ws_bytes, float p_drop,
                             sepr_u64 seed, sepr_stream_t stream) {
  if (!x || !dy || !dx || !w || !g || n <= 0 || T <= 0 || F <= 0 || F % 32 || !rows_ok((long long)n * T)) return SEPR_EINVAL;
  DropSaltScope salt_scope(w->seed_salt);
  Carve cx(const_cast<void*>(ctx), ctx_bytes, false), wk(ws, ws_bytes, false);
  if (gcfn_is_fused(w, F)) return gcfn_fused_bwd(x, dy, dx, n, T, F, w, g, cx, wk, p_drop, seed, SEPR_ST);
  return gcfn_bwd(x, dy, dx, n, T, F, w, g, cx, wk, p_drop, seed, SEPR_ST);
}
extern "C" int sepr_cla_train_fwd(const float* x, float* y, int n, int T, int F, int K, const sepr_cla_tw* w, void* ctx, size_t ctx_bytes,
                                  void* ws, size_t ws_bytes, float p_drop, sepr_u64 seed, sepr_stream_t stream) {
  if (!x || !y || !w || n <= 0 || T <= 0 || F <= 0 || F % 64 || !rows_ok((long long)n * T) || x == y) return SEPR_EINVAL;
  DropSaltScope salt_scope(w->seed_salt);
  Carve cx(ctx, ctx_bytes, false), wk(ws, ws_bytes, false);
  return cla_fwd(x, y, n, T, F, K, w, cx, wk, p_drop, seed, SEPR_ST);
}
extern "C" int sepr_cla_bwd(const float* x, const float* dy, float* dx, int n, int T, int F, int K, const sepr_cla_tw* w,
                            const sepr_cla_grad* g, const void* ctx, size_t ctx_bytes, void* ws, size_t ws_bytes, float p_drop,
                            sepr_u64 seed, sepr_stream_t stream) {
  if (!x || !dy || !dx || !w || !g || n <= 0 || T <= 0 || F <= 0 || F % 64 || !rows_ok((long long)n * T)) return SEPR_EINVAL;
  DropSaltScope salt_scope(w->seed_salt);
  Carve cx(const_cast<void*>(ctx), ctx_bytes, false), wk(ws, ws_bytes, false);
  return cla_bwd(x, dy, dx, n, T, F, K, w, g, cx, wk, p_drop, seed, SEPR_ST);
}
extern "C" int sepr_ega_train_fwd(const float* x, float* y, int n, int T, int Tp, int F, int H, const sepr_ega_tw* w, void* ctx,
                                  size_t ctx_bytes, void* ws, size_t ws_bytes, float p_drop, sepr_u64 seed, sepr_stream_t stream) {
  if (!x || !y || !w || x == y || n <= 0 || T <= 0 || Tp <= 0 || T % Tp || F <= 0 || F % 32 || H <= 0 || F % H ||
      !rows_ok((long long)n * T))
    return SEPR_EINVAL;
  DropSaltScope salt_scope(w->attn.seed_salt);
  Carve cx(ctx, ctx_bytes, false), wk(ws, ws_bytes, false);
  return ega_fwd(x, y, n, T, Tp, F, H, w, cx, wk, p_drop, seed, SEPR_ST);
}
extern "C" int sepr_ega_bwd(const float* x, const float* dy, float* dx, int n, int T, int Tp, int F, int H, const sepr_ega_tw* w,
                            const sepr_ega_grad* g, const void* ctx, size_t ctx_bytes, void* ws, size_t ws_bytes, float p_drop,
                            sepr_u64 seed, sepr_stream_t stream) {
  if (!x || !dy || !dx || !w || !g || n <= 0 || T <= 0 || Tp <= 0 || T % Tp || F <= 0 || F % 32 || H <= 0 || F % H ||
      !rows_ok((long long)n * T))
    return SEPR_EINVAL;
  DropSaltScope salt_scope(w->attn.seed_salt);
  Carve cx(const_cast<void*>(ctx), ctx_bytes, false), wk(ws, ws_bytes, false);
  return ega_bwd(x, dy, dx, n, T, Tp, F, H, w, g, cx, wk, p_drop, seed, SEPR_ST);
}
extern "C" int sepr_spkattn_train_fwd(const float* x, float* y, int nS, int S, int T, int F, int H, const sepr_mha_tw* w, void* ctx,
                                      size_t ctx_bytes, void* ws, size_t ws_bytes, float p_drop, sepr_u64 seed, sepr_stream_t stream) {
  if (!x || !y || !w || nS <= 0 || S <= 0 || nS % S || T <= 0 || F <= 0 || F % 32 || H <= 0 || F % H || !rows_ok((long long)nS * T))
    return SEPR_EINVAL;
  DropSaltScope salt_scope(w->seed_salt);
  Carve cx(ctx, ctx_bytes, false), wk(ws, ws_bytes, false);
  return spk_fwd(x, y, nS, S, T, F, H, w, cx, wk, p_drop, seed, SEPR_ST);
}
extern "C" int sepr_spkattn_bwd(const float* x, const float* dy, float* dx, int nS, int S, int T, int F, int H, const sepr_mha_tw* w,
                                const sepr_mha_grad* g, const void* ctx, size_t ctx_bytes, void* ws, size_t ws_bytes, float p_drop,
                                sepr_u64 seed, sepr_stream_t stream) {
  if (!x || !dy || !dx || !w || !g || nS <= 0 || S <= 0 || nS % S || T <= 0 || F <= 0 || F % 32 || H <= 0 || F % H ||
      !rows_ok((long long)nS * T))
    return SEPR_EINVAL;
  DropSaltScope salt_scope(w->seed_salt);
  Carve cx(const_cast<void*>(ctx), ctx_bytes, false), wk(ws, ws_bytes, false);
  return spk_bwd(x, dy, dx, nS, S, T, F, H, w, g, cx, wk, p_drop, seed, SEPR_ST);
}
extern "C" int sepr_downconv_train_fwd(const float* x, float* y, int n, int T, int F, int K, const sepr_down_tw* w, void* ctx,
                                       size_t ctx_bytes, void* ws, size_t ws_bytes, sepr_stream_t stream) {
  if (!x || !y || !w || n <= 0 || T <= 0 || F <= 0 || F % 4 || K <= 0 || !(K & 1)) return SEPR_EINVAL;
  Carve cx(ctx, ctx_bytes, false), wk(ws, ws_bytes, false);
  return down_fwd(x, y, n, T, F, K, w, cx, wk, SEPR_ST);
}
extern "C" int sepr_downconv_bwd(const float* x, const float* dy, float* dx, int n, int T, int F, int K, const sepr_down_tw* w,
                                 const sepr_down_grad* g, const void* ctx, size_t ctx_bytes, void* ws, size_t ws_bytes,
                                 sepr_stream_t stream) {
  if (!x || !dy || !dx || !w || !g || n <= 0 || T <= 0 || F <= 0 || F % 4 || K <= 0 || !(K & 1)) return SEPR_EINVAL;
  Carve cx(const_cast<void*>(ctx), ctx_bytes, false), wk(ws, ws_bytes, false);
  return down_bwd(x, dy, dx, n, T, F, K, w, g, cx, wk, SEPR_ST);
}
extern "C" int sepr_spksplit_train_fwd(const float* x, float* y, int B, int S, int T, int F, float gn_eps, const sepr_split_tw* w,
                                       void* ctx, size_t ctx_bytes, void* ws, size_t ws_bytes, sepr_stream_t stream) {
  if (!x || !y || !w || B <= 0 || S <= 0 || T <= 0 || F <= 0 || F % 32 || !rows_ok((long long)B * T * S)) return SEPR_EINVAL;
  Carve cx(ctx, ctx_bytes, false), wk(ws, ws_bytes, false);
  return split_fwd(x, y, B, S, T, F, gn_eps, w, cx, wk, SEPR_ST);
}
extern "C" int sepr_spksplit_bwd(const float* x, const float* dy, float* dx, int dx_accumulate, int B, int S, int T, int F,
                                 const sepr_split_tw* w, const sepr_split_grad* g, const void* ctx, size_t ctx_bytes, void* ws,
                                 size_t ws_bytes, sepr_stream_t stream) {
  if (!x || !dy || !dx || !w || !g || B <= 0 || S <= 0 || T <= 0 || F <= 0 || F % 32 || !rows_ok((long long)B * T * S)) return SEPR_EINVAL;
  Carve cx(const_cast<void*>(ctx), ctx_bytes, false), wk(ws, ws_bytes, false);
  return split_bwd(x, dy, dx, dx_accumulate, B, S, T, F, w, g, cx, wk, SEPR_ST);
}

extern "C" size_t sepr_linear_wgrad_workspace(int M, int N, int K) { return tn_workspace_bytes(M, N, K); }
extern "C" int sepr_linear_wgrad(const float* A, const float* B, float* G, float* colsum, int M, int N, int K, int accumulate, int x3,
                                 void* ws, size_t ws_bytes, sepr_stream_t stream) {
  if (!A || !B || !G || M <= 0) return SEPR_EINVAL;
  return wgrad(A, N, B, K, nullptr, G, colsum, M, N, K, accumulate, x3 < 0 || x3 > 2 ? 1 : x3, ws, ws_bytes, SEPR_ST);
}

extern "C" int sepr_linear_wgrad_bf16(const void* A16, int lda, const void* B16, int ldb, float* G, float* colsum, int M, int N, int K,
                                      int accumulate, void* ws, size_t ws_bytes, sepr_stream_t stream) {
  if (!A16 || !B16 || !G || M <= 0 || lda < N || ldb < K) return SEPR_EINVAL;
  TnArgs t = tn_args_zero();
  t.M = M; t.N = N; t.K = K;
  t.A = static_cast<const float*>(A16); t.lda = lda; t.a16 = 1;
  t.B = static_cast<const float*>(B16); t.ldb = ldb; t.b16 = 1;
  t.G = G; t.ldg = K; t.accumulate = accumulate;
  t.colsum = colsum; t.colsum_accumulate = accumulate;
  return launch_gemm_tn(t, 2, ws, ws_bytes, SEPR_ST);
}

extern "C" int sepr_linear_wgrad_norm(const float* A, const float* B, const float* stats, float* G, float* colsum, int M, int N, int K,
                                      int accumulate, int x3, void* ws, size_t ws_bytes, sepr_stream_t stream) {
  if (!A || !B || !stats || !G || M <= 0) return SEPR_EINVAL;
  return wgrad(A, N, B, K, stats, G, colsum, M, N, K, accumulate, x3 < 0 || x3 > 2 ? 1 : x3, ws, ws_bytes, SEPR_ST);
}

// =====================================================================================================================
// fusion conv, OutputLayer + decoder, encoder + projector, sizing
// =====================================================================================================================
namespace sepr {
namespace {

int fuse_bwd(const float* lo, const float* skip, const float* dy, float* dlo, float* dskip, int n, int T, int F, const sepr_fuse_tw* w,
             const sepr_fuse_grad* g, Carve& ws, hipStream_t st) {
  const long long M = (long long)n * T;
  float* dcat = ws.f32(2LL * F * M);
  const size_t tnb = tn_workspace_bytes((int)M, F, 2 * F);
  void* tnw = ws.take(tnb);
  if (ws.dry) return SEPR_OK;
  if (!ws.ok()) return SEPR_EWORKSPACE;
  TnArgs t = tn_args_zero();                       // dW [F][2F] += sum dy [up2(lo) | skip]   (module.py:212-214)
  t.M = (int)M; t.N = F; t.K = 2 * F;
  t.A = dy; t.lda = F;
  t.B = lo; t.ldb = F; t.rows_out = T; t.rows_valid = T; t.b_shift = 1; t.seq_stride = (long long)(T / 2) * F;
  t.B2 = skip; t.ldb2 = F; t.ksplit = F;
  t.G = g->w; t.ldg = 2 * F; t.accumulate = 1; t.colsum = g->b; t.colsum_accumulate = 1;
  SEPR_TRY(launch_gemm_tn(t, tn_mode(w->l), tnw, tnb, st));
  SEPR_TRY(plain(dy, F, dcat, 2 * F, M, 2 * F, F, w->l_t, nullptr, st));
  return launch_unfuse(dcat, dlo, dskip, n, T, F, st);
}

struct OutCtx { float *a1, *o1, *o2; };
OutCtx out_ctx(Carve& cx, long long Mp, int F, int N) {
  OutCtx k;
  k.a1 = cx.f32(4LL * F * Mp);
  k.o1 = cx.f32(2LL * F * Mp);
  k.o2 = cx.f32((long long)N * Mp);
  return k;
}
int out_fwd(const float* x, int nS, int S, int Tsrc, int L, const int* idx, const float* enc, int F, int N, int K, int stride,
            const sepr_out_tw* w, float* wav, Carve& cx, hipStream_t st) {
  const long long Mp = idx ? (long long)nS * Tsrc : (long long)nS * L;
  OutCtx k = out_ctx(cx, Mp, F, N);
  if (cx.dry) return SEPR_OK;
  if (!cx.ok()) return SEPR_EWORKSPACE;
  {  // Linear F->4F on the frames the head reads (main: the crop of module.py:250; aux: the source frames)
    GemmArgs a = gemm_args_zero();
    a.M = (int)Mp; a.N = 4 * F; a.K = F;
    a.A = x; a.lda = F; a.Y = k.a1; a.ldc = 4 * F;
    if (!idx) { a.rows_out = L; a.rows_src = Tsrc; a.rows_valid = L; }
    SEPR_TRY(lin(PRO_PLAIN, EPI_STORE, a, w->l1, SEPR_SITE_NONE, st));
  }
  SEPR_TRY(launch_glu_fwd(k.a1, k.o1, Mp, 2 * F, st));                                       // module.py:246 (GLU)
  SEPR_TRY(plain(k.o1, 2 * F, k.o2, N, Mp, N, 2 * F, w->l2, nullptr, st));                   // :247
  const int Tout = (L - 1) * stride + K;
  return launch_decoder(k.o2, nS, S, L, N, K, stride, w->wdec, wav, Tout, idx, Tsrc, enc, st);   // :257-260, :278-283
}
int out_bwd(const float* x, const float* dwav, float* dx, int dx_acc, float* denc, int nS, int S, int Tsrc, int L, const int* idx,
            const int* idx_start, const float* enc, int F, int N, int K, int stride, const sepr_out_tw* w, const sepr_out_grad* g,
            Carve& cx, Carve& ws, hipStream_t st) {
  const bool aux = idx != nullptr;
  const long long Mp = aux ? (long long)nS * Tsrc : (long long)nS * L, ML = (long long)nS * L;
  const int Tout = (L - 1) * stride + K;
  OutCtx k = out_ctx(cx, Mp, F, N);
  float* dwp = ws.f32((long long)nS * Tout);
  float* dm = ws.f32((long long)N * ML);
  float* mt = ws.f32(aux ? (long long)N * ML : 4);
  float* do2 = aux ? ws.f32((long long)N * Mp) : dm;
  float* do1 = ws.f32(2LL * F * Mp);
  float* da1 = ws.f32(4LL * F * Mp);
  size_t tnb = tn_workspace_bytes((int)ML, N, K);
  if (tn_workspace_bytes((int)Mp, N, 2 * F) > tnb) tnb = tn_workspace_bytes((int)Mp, N, 2 * F);
  if (tn_workspace_bytes((int)Mp, 4 * F, F) > tnb) tnb = tn_workspace_bytes((int)Mp, 4 * F, F);
  void* tnw = ws.take(tnb);
  if (ws.dry) return SEPR_OK;
  if (!cx.ok() || !ws.ok()) return SEPR_EWORKSPACE;
  if (aux && (!idx_start || !enc || !denc)) return SEPR_EINVAL;
  const int x3 = tn_mode(w->l1);
  SEPR_TRY(launch_permute_sb(dwav, dwp, S, nS / S, Tout, st));
  SEPR_TRY(launch_dec_bwd_dm(dwp, w->wdec, dm, nS, L, N, K, stride, Tout, st));
  const float* m = k.o2;
  if (aux) {
    SEPR_TRY(launch_aux_m(k.o2, enc, idx, mt, nS, S, Tsrc, L, N, st));
    m = mt;
  }
  {  // decoder weight [N][1][K] += sum_rows m[row][n] * dwav_frame[row][k]
    TnArgs t = tn_args_zero();
    t.M = (int)ML; t.N = N; t.K = K;
    t.A = m; t.lda = N;
    t.B = dwp; t.ldb = stride; t.rows_out = L; t.rows_valid = L; t.seq_stride = Tout;
    t.G = g->wdec; t.ldg = K; t.accumulate = 1;
    SEPR_TRY(launch_gemm_tn(t, 0, tnw, tnb, st));      // exact core: K = 16, negligible work
  }
  if (aux) {
    SEPR_TRY(launch_aux_mask_bwd(dm, k.o2, enc, idx_start, do2, nS, S, Tsrc, L, N, st));
    SEPR_TRY(launch_aux_denc(dm, k.o2, idx, denc, nS, S, Tsrc, L, N, st));
  }
  SEPR_TRY(wgrad(do2, N, k.o1, 2 * F, nullptr, g->w2, g->b2, Mp, N, 2 * F, 1, x3, tnw, tnb, st));
  SEPR_TRY(plain(do2, N, do1, 2 * F, Mp, 2 * F, N, w->l2_t, nullptr, st));
  SEPR_TRY(launch_glu_bwd(do1, k.a1, da1, Mp, 2 * F, st));
  {  // first projection: input rows = the frames the head read
    TnArgs t = tn_args_zero();
    t.M = (int)Mp; t.N = 4 * F; t.K = F;
    t.A = da1; t.lda = 4 * F; t.B = x; t.ldb = F;
    if (!aux) { t.rows_out = L; t.rows_valid = L; t.seq_stride = (long long)Tsrc * F; }
    t.G = g->w1; t.ldg = F; t.accumulate = 1; t.colsum = g->b1; t.colsum_accumulate = 1;
    SEPR_TRY(launch_gemm_tn(t, x3, tnw, tnb, st));
  }
  {  // dx over all Tsrc frames of every sequence (main head: frames >= L read nothing -> zero gradient)
    GemmArgs a = gemm_args_zero();
    a.M = (int)((long long)nS * Tsrc); a.N = F; a.K = 4 * F;
    a.A = da1; a.lda = 4 * F; a.Y = dx; a.ldc = F; a.R = dx_acc ? dx : nullptr;
    if (!aux) { a.rows_out = Tsrc; a.rows_src = L; a.rows_valid = L; }
    SEPR_TRY(lin(PRO_PLAIN, dx_acc ? EPI_RES : EPI_STORE, a, w->l1_t, SEPR_SITE_NONE, st));
  }
  return SEPR_OK;
}

int front_fwd(const float* wav, int B, int T, int N, int K, int stride, int F, int Lp, float eps, const sepr_front_tw* w, float* enc,
              float* out, Carve& cx, Carve& ws, hipStream_t st) {
  const int L = (T - K) / stride + 1;
  float* stats = cx.f32(2 * B);
  const int ntile = encoder_tiles(L);
  double* part = static_cast<double*>(ws.take((size_t)B * ntile * 2 * sizeof(double)));
  if (cx.dry) return SEPR_OK;
  if (!cx.ok() || !ws.ok()) return SEPR_EWORKSPACE;
  SEPR_TRY(launch_encoder(wav, B, T, L, w->w_enc, N, K, stride, enc, part, st));              // module.py:19-23
  SEPR_TRY(launch_gn_finalize(part, B, ntile, (long long)L * N, eps, stats, st));
  GemmArgs a = gemm_args_zero();                                                             // module.py:32-35, 220-234
  a.M = B * Lp; a.N = F; a.K = N;
  a.A = enc; a.lda = N; a.rows_out = Lp; a.rows_src = L; a.rows_valid = L;
  a.stats = stats; a.stat_seq = 1; a.gamma = w->gn_g; a.beta = w->gn_b;
  a.W = w->proj_w; a.bias = nullptr; a.Y = out; a.ldc = F;
  return launch_gemm(PRO_NORM, EPI_STORE, a, SEPR_SITE_PROJECTOR, st);
}
int front_bwd(const float* wav, const float* enc, const float* dout, float* denc_aux, int B, int T, int N, int K, int stride, int F,
              int Lp, const sepr_front_tw* w, const sepr_front_grad* g, Carve& cx, Carve& ws, hipStream_t st) {
  const int L = (T - K) / stride + 1;
  const long long ML = (long long)B * L;
  float* stats = cx.f32(2 * B);
  float* dWh = ws.f32((long long)F * N);
  float* s = ws.f32(F);
  float* deh = ws.f32((long long)N * ML);
  float* de = ws.f32((long long)N * ML);
  float* dump = ws.f32(2 * N);
  size_t tnb = tn_workspace_bytes(B * Lp, F, N);
  if (tn_workspace_bytes((int)ML, N, K) > tnb) tnb = tn_workspace_bytes((int)ML, N, K);
  void* tnw = ws.take(tnb);
  const size_t gnb = gn_bwd_ws(B, L, N);
  void* gnw = ws.take(gnb);
  if (ws.dry) return SEPR_OK;
  if (!cx.ok() || !ws.ok()) return SEPR_EWORKSPACE;
  {  // projector weight: dWh [F][N] = sum over the valid frames of dout[m][f] * enc_hat[m][n]
    TnArgs t = tn_args_zero();
    t.M = B * Lp; t.N = F; t.K = N;
    t.A = dout; t.lda = F; t.mask_a = 1;
    t.B = enc; t.ldb = N; t.rows_out = Lp; t.rows_valid = L; t.seq_stride = (long long)L * N;
    t.stats = stats; t.stat_seq = 1;
    t.G = dWh; t.ldg = N; t.colsum = s;
    SEPR_TRY(launch_gemm_tn(t, tn_mode(w->proj_t), tnw, tnb, st));
  }
  SEPR_TRY(launch_finish_norm_linear(dWh, s, w->proj_w, w->gn_g, w->gn_b, g->proj_w, nullptr, g->gn_g, g->gn_b, F, N, st));
  {  // d enc_hat [B*L][N] = dout(valid rows) . (W * gamma)
    GemmArgs a = gemm_args_zero();
    a.M = (int)ML; a.N = N; a.K = F;
    a.A = dout; a.lda = F; a.rows_out = L; a.rows_src = Lp; a.rows_valid = L;
    a.Y = deh; a.ldc = N;
    SEPR_TRY(lin(PRO_PLAIN, EPI_STORE, a, w->proj_t, SEPR_SITE_NONE, st));
  }
  // GroupNorm(1, N) backward over (L, N) of each sample (affine already folded: gamma = 1 here)
  SEPR_TRY(hipMemsetAsync(dump, 0, 2 * N * sizeof(float), st) == hipSuccess ? SEPR_OK : SEPR_EHIP);
  SEPR_TRY(launch_gn_bwd(deh, enc, stats, w->ones, de, dump, dump + N, B, L, N, 0, gnw, gnb, st));
  // + gradient from the auxiliary heads' masks, through the GELU; then the encoder filters
  SEPR_TRY(launch_enc_bwd_pre(de, denc_aux, wav, w->w_enc, B, T, L, N, K, stride, st));
  TnArgs t = tn_args_zero();                       // w_enc [N][1][K] += sum dpre[m][n] * wav[b][stride*l + k]
  t.M = (int)ML; t.N = N; t.K = K;
  t.A = de; t.lda = N;
  t.B = wav; t.ldb = stride; t.rows_out = L; t.rows_valid = L; t.seq_stride = T;
  t.G = g->w_enc; t.ldg = K; t.accumulate = 1;
  return launch_gemm_tn(t, 0, tnw, tnb, st);
}

}  // namespace
}  // namespace sepr

extern "C" int sepr_fuse_bwd(const float* lo, const float* skip, const float* dy, float* dlo, float* dskip, int n, int T, int F,
                             const sepr_fuse_tw* w, const sepr_fuse_grad* g, void* ws, size_t ws_bytes, sepr_stream_t stream) {
  if (!lo || !skip || !dy || !dlo || !dskip || !w || !g || n <= 0 || T <= 0 || (T & 1) || F <= 0 || F % 32) return SEPR_EINVAL;
  Carve wk(ws, ws_bytes, false);
  return fuse_bwd(lo, skip, dy, dlo, dskip, n, T, F, w, g, wk, SEPR_ST);
}
extern "C" int sepr_outlayer_decoder_train_fwd(const float* x, int nS, int S, int Tsrc, int L, const int* idx, const float* enc, int F,
                                               int N, int K, int stride, const sepr_out_tw* w, float* wav, void* ctx, size_t ctx_bytes,
                                               void* ws, size_t ws_bytes, sepr_stream_t stream) {
  (void)ws; (void)ws_bytes;
  if (!x || !w || !wav || nS <= 0 || S <= 0 || nS % S || Tsrc <= 0 || L <= 0 || F % 32 || N % 16) return SEPR_EINVAL;
  if (K != TRAIN_ENC_K || stride != TRAIN_ENC_STRIDE) return SEPR_EINVAL;   // sepr_train_ctx_bytes / ws_bytes size for this geometry
  if (!idx && L > Tsrc) return SEPR_EINVAL;
  if (idx && (Tsrc > L || !enc)) return SEPR_EINVAL;
  Carve cx(ctx, ctx_bytes, false);
  return out_fwd(x, nS, S, Tsrc, L, idx, enc, F, N, K, stride, w, wav, cx, SEPR_ST);
}
extern "C" int sepr_outlayer_decoder_bwd(const float* x, const float* dwav, float* dx, int dx_accumulate, float* denc, int nS, int S,
                                         int Tsrc, int L, const int* idx, const int* idx_start, const float* enc, int F, int N, int K,
                                         int stride, const sepr_out_tw* w, const sepr_out_grad* g, const void* ctx, size_t ctx_bytes,
                                         void* ws, size_t ws_bytes, sepr_stream_t stream) {
  if (!x || !dwav || !dx || !w || !g || nS <= 0 || S <= 0 || nS % S || Tsrc <= 0 || L <= 0 || F % 32 || N % 16) return SEPR_EINVAL;
  if (K != TRAIN_ENC_K || stride != TRAIN_ENC_STRIDE) return SEPR_EINVAL;
  if (!idx && L > Tsrc) return SEPR_EINVAL;
  if (idx && Tsrc > L) return SEPR_EINVAL;
  Carve cx(const_cast<void*>(ctx), ctx_bytes, false), wk(ws, ws_bytes, false);
  return out_bwd(x, dwav, dx, dx_accumulate, denc, nS, S, Tsrc, L, idx, idx_start, enc, F, N, K, stride, w, g, cx, wk, SEPR_ST);
}
extern "C" int sepr_front_train_fwd(const float* wav, int B, int T, int N, int K, int stride, int F, int Lp, float gn_eps,
                                    const sepr_front_tw* w, float* enc, float* out, void* ctx, size_t ctx_bytes, void* ws,
                                    size_t ws_bytes, sepr_stream_t stream) {
  if (!wav || !w || !enc || !out || B <= 0 || T < K || stride <= 0 || Lp < (T - K) / stride + 1) return SEPR_EINVAL;
  if (K != TRAIN_ENC_K || stride != TRAIN_ENC_STRIDE) return SEPR_EINVAL;
  Carve cx(ctx, ctx_bytes, false), wk(ws, ws_bytes, false);
  return front_fwd(wav, B, T, N, K, stride, F, Lp, gn_eps, w, enc, out, cx, wk, SEPR_ST);
}
extern "C" int sepr_front_bwd(const float* wav, const float* enc, const float* dout, float* denc_aux, int B, int T, int N, int K,
                              int stride, int F, int Lp, const sepr_front_tw* w, const sepr_front_grad* g, const void* ctx,
                              size_t ctx_bytes, void* ws, size_t ws_bytes, sepr_stream_t stream) {
  if (!wav || !enc || !dout || !w || !g || B <= 0 || T < K || stride <= 0) return SEPR_EINVAL;
  if (K != TRAIN_ENC_K || stride != TRAIN_ENC_STRIDE) return SEPR_EINVAL;
  Carve cx(const_cast<void*>(ctx), ctx_bytes, false), wk(ws, ws_bytes, false);
  return front_bwd(wav, enc, dout, denc_aux, B, T, N, K, stride, F, Lp, w, g, cx, wk, SEPR_ST);
}

// ---- sizing: replay each block's carving in dry mode --------------------------------------------------------------
static void train_sizes(int op, int n, int T, int Tp, int F, int N, int S, int H, int K, size_t* ctx_b, size_t* ws_b) {
  Carve cf(nullptr, 0, true), wf(nullptr, 0, true), cb(nullptr, 0, true), wb(nullptr, 0, true);
  const float p1 = 0.5f;      // size for the dropout-enabled layout (superset)
  switch (op) {
    case SEPR_TOP_GCFN:
      gcfn_fwd(nullptr, nullptr, n, T, F, nullptr, cf, wf, p1, 0, nullptr);
      gcfn_bwd(nullptr, nullptr, nullptr, n, T, F, nullptr, nullptr, cb, wb, p1, 0, nullptr);
      break;
    case SEPR_TOP_GCFN_FUSED:
      gcfn_fused_fwd(nullptr, nullptr, n, T, F, nullptr, cf, p1, 0, nullptr, 0);
      gcfn_fused_bwd(nullptr, nullptr, nullptr, n, T, F, nullptr, nullptr, cb, wb, p1, 0, nullptr, 0);
      break;
    case SEPR_TOP_GCFN_FUSED16:
      gcfn_fused_fwd(nullptr, nullptr, n, T, F, nullptr, cf, p1, 0, nullptr, 1);
      gcfn_fused_bwd(nullptr, nullptr, nullptr, n, T, F, nullptr, nullptr, cb, wb, p1, 0, nullptr, 1);
      break;
    case SEPR_TOP_CLA:
      cla_fwd(nullptr, nullptr, n, T, F, K, nullptr, cf, wf, p1, 0, nullptr);
      cla_bwd(nullptr, nullptr, nullptr, n, T, F, K, nullptr, nullptr, cb, wb, p1, 0, nullptr);
      break;
    case SEPR_TOP_EGA:
      ega_fwd(nullptr, nullptr, n, T, Tp, F, H, nullptr, cf, wf, p1, 0, nullptr, 0);
      ega_bwd(nullptr, nullptr, nullptr, n, T, Tp, F, H, nullptr, nullptr, cb, wb, p1, 0, nullptr, 0);
      break;
    case SEPR_TOP_EGA_X3:
      ega_fwd(nullptr, nullptr, n, T, Tp, F, H, nullptr, cf, wf, p1, 0, nullptr, 1);
      ega_bwd(nullptr, nullptr, nullptr, n, T, Tp, F, H, nullptr, nullptr, cb, wb, p1, 0, nullptr, 1);
      break;
    case SEPR_TOP_SPKATTN:
      spk_fwd(nullptr, nullptr, n, S, T, F, H, nullptr, cf, wf, p1, 0, nullptr);
      spk_bwd(nullptr, nullptr, nullptr, n, S, T, F, H, nullptr, nullptr, cb, wb, p1, 0, nullptr);
      break;
    case SEPR_TOP_DOWN:
      down_fwd(nullptr, nullptr, n, T, F, K, nullptr, cf, wf, nullptr);
      down_bwd(nullptr, nullptr, nullptr, n, T, F, K, nullptr, nullptr, cb, wb, nullptr);
      break;
    case SEPR_TOP_SPLIT:
      split_fwd(nullptr, nullptr, n, S, T, F, 0.f, nullptr, cf, wf, nullptr);
      split_bwd(nullptr, nullptr, nullptr, 0, n, S, T, F, nullptr, nullptr, cb, wb, nullptr);
      break;
    case SEPR_TOP_FUSE:
      fuse_bwd(nullptr, nullptr, nullptr, nullptr, nullptr, n, T, F, nullptr, nullptr, wb, nullptr);
      break;
    case SEPR_TOP_OUT: {   // n = sequences (B*S), T = output frames L, Tp = source frames; aux layout when Tp < T... both sized
      static int dummy_idx = 0;
      for (int aux = 0; aux < 2; ++aux) {
        if (aux && Tp > T) continue;
        if (!aux && T > Tp) continue;
        Carve c1(nullptr, 0, true), c2(nullptr, 0, true), w2(nullptr, 0, true);
        out_fwd(nullptr, n, S, Tp, T, aux ? &dummy_idx : nullptr, nullptr, F, N, TRAIN_ENC_K, TRAIN_ENC_STRIDE, nullptr, nullptr, c1, nullptr);
        out_bwd(nullptr, nullptr, nullptr, 0, nullptr, n, S, Tp, T, aux ? &dummy_idx : nullptr, nullptr, nullptr, F, N, TRAIN_ENC_K, TRAIN_ENC_STRIDE, nullptr,
                nullptr, c2, w2, nullptr);
        if (c1.need() > cf.off) cf.off = c1.need();
        if (w2.need() > wb.off) wb.off = w2.need();
      }
      break;
    }
    case SEPR_TOP_FRONT: {   // n = B, T = samples, Tp = Lp
      front_fwd(nullptr, n, T, N, TRAIN_ENC_K, TRAIN_ENC_STRIDE, F, Tp, 0.f, nullptr, nullptr, nullptr, cf, wf, nullptr);
      front_bwd(nullptr, nullptr, nullptr, nullptr, n, T, N, TRAIN_ENC_K, TRAIN_ENC_STRIDE, F, Tp, nullptr, nullptr, cb, wb, nullptr);
      break;
    }
    default: break;
  }
  *ctx_b = cf.need() > cb.need() ? cf.need() : cb.need();
  *ws_b = (wf.need() > wb.need() ? wf.need() : wb.need()) + 1024;
}
extern "C" size_t sepr_train_ctx_bytes(int op, int n, int T, int Tp, int F, int N, int S, int H) {
  if (n <= 0 || T <= 0 || F <= 0) return 0;
  size_t c = 0, w = 0;
  train_sizes(op, n, T, Tp, F, N, S, H > 0 ? H : 1, 65, &c, &w);
  return c + 256;
}
extern "C" size_t sepr_train_ws_bytes(int op, int n, int T, int Tp, int F, int N, int S, int H, int K) {
  if (n <= 0 || T <= 0 || F <= 0) return 0;
  size_t c = 0, w = 0;
  train_sizes(op, n, T, Tp, F, N, S, H > 0 ? H : 1, K > 0 ? K : 65, &c, &w);
  return w;
}
