// extern "C" entry points of libsepr_hip.so (declared in include/sepr.h): each fused block of the
// separator is a short, fixed sequence of launches on the caller's stream.
#include <string.h>
#include <atomic>
#include <mutex>

#include "sepr_gemm_epi.h"
#include "sepr_gcfn_fused.h"
#include "sepr_pointwise.h"
#include <stdlib.h>

namespace sepr {

static thread_local char g_err[256] = "";
void set_hip_error(hipError_t e, const char* where) {
  snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
}

static const float LN_EPS = 1e-5f;  // torch.nn.LayerNorm default (network.py:50,81,133,162)

struct SpkFusedArgs {
  const float* x;
  float* y;
  int NF, T;
  const void* w1p;
  const void* w2p;
  const float* bo;
  const float* ls;
  float eps, inv_sqrt_dk;
};
int launch_spk_fused(const SpkFusedArgs& a, int F, int site, hipStream_t stream);     // sepr_spk_fused.hip
struct ClaFusedArgs {
  const float* x;
  const float* res;
  float* y;
  int M;
  const void* w1p;
  const void* w2p;
  const float* b3;
  const float* ls;
  float eps;
  const float* att;
  int T, Tp, fac;
  int ldy, pool;
  const float* o;
  float* att_w;
  const void* wop;
  const float* bo;
  const float* lso;
  int Mp;
};
size_t pit_workspace_bytes(int S, int B, int T);                                      // sepr_criterion.hip
int launch_ega_gate(const ClaFusedArgs& a, int F, int site, hipStream_t stream);      // sepr_cla_fused.hip
int launch_ega_qkv(const ClaFusedArgs& a, int F, int site, hipStream_t stream);       // sepr_cla_fused.hip
int launch_cla_head(const ClaFusedArgs& a, int F, int site, hipStream_t stream);      // sepr_cla_fused.hip
int launch_cla_tail(const ClaFusedArgs& a, int F, int site, hipStream_t stream);

// one projection on whichever core its weights were packed for
static int project(int pro, int epi, GemmArgs& a, const sepr_x3_w& x3, int site, hipStream_t st) {
  if (x3.wp) {
    a.Wp = x3.wp;
    a.bias = x3.bias;
    return launch_gemm_x3(pro, epi, a, site, st);
  }
  return launch_gemm(pro, epi, a, site, st);
}

// the same, with the LayerNorm statistics of the output rows delivered to y_stats (optional): the bf16x3 core takes them from its tile tail
// or launches rowstats itself (GemmArgs.stats_out); the exact-f32 core is followed by a rowstats launch here
static int project_stats(int pro, int epi, GemmArgs& a, const sepr_x3_w& x3, int site, hipStream_t st, float* y_stats) {
  if (y_stats && x3.wp) {
    a.stats_out = y_stats;
    a.stats_eps = LN_EPS;
    return project(pro, epi, a, x3, site, st);
  }
  SEPR_TRY(project(pro, epi, a, x3, site, st));
  return y_stats ? launch_rowstats(a.Y, y_stats, a.M, a.N, LN_EPS, st) : SEPR_OK;
}

// ---- workspace plans (floats unless noted) --------------------------------------------------------
static size_t ws_gcfn(long long M, int F) {
  return align_up(2 * M * 4) + align_up(3LL * F * M * 4) + 1024;
}
static size_t ws_cla(long long M, int F) {
  return align_up(2 * M * 4) + 2 * align_up((long long)F * M * 4) + align_up(2LL * F * M * 4) + 1024;
}
static size_t ws_ega(long long M, long long Mp, int F) {
  return align_up(2 * M * 4) + align_up(2 * Mp * 4) + 3 * align_up((long long)F * Mp * 4) + align_up(3LL * F * Mp * 4) + 2048;
}
static size_t ws_spk(long long M, int F) {
  return align_up(2 * M * 4) + align_up(3LL * F * M * 4) + align_up((long long)F * M * 4) + 1024;
}
static size_t ws_split(long long M, int n_out, int T, int F, int S) {
  return align_up(2LL * F * S * M * 4) + align_up((size_t)n_out * gn_chunks((long long)T * F) * 2 * 8) + align_up((size_t)n_out * 2 * 4) + 1024;
}
static size_t ws_out(long long M, int F, int N) {
  return align_up(2LL * F * M * 4) + align_up((long long)N * M * 4) + 1024;
}
static size_t ws_enc(int B, int L) { return align_up((size_t)B * encoder_tiles(L) * 2 * 8) + 512; }

}  // namespace sepr

using namespace sepr;

namespace sepr {
namespace {
std::atomic<int> g_knobs_loaded{0};
int g_knobs[SEPR_KNOB_COUNT];
std::mutex g_knobs_mu;
void knobs_read() {
  static const struct { const char* name; int dflt; } tab[SEPR_KNOB_COUNT] = {
      {"SEPR_X3_WIDE", 1}, {"SEPR_TRAIN_GCFN_PLANES", 1}, {"SEPR_TRAIN_ATTN_ONE", 1}, {"SEPR_TRAIN_CLA16", 1}, {"SEPR_FOLD_HEAD", 1}, {"SEPR_TN16", 1}};
  for (int i = 0; i < SEPR_KNOB_COUNT; ++i) {
    const char* e = getenv(tab[i].name);
    g_knobs[i] = (e && e[0]) ? atoi(e) : tab[i].dflt;
  }
}
}  // namespace
int knob(int id) {
  if (id < 0 || id >= SEPR_KNOB_COUNT) return 0;
  if (!g_knobs_loaded.load(std::memory_order_acquire)) {
    std::lock_guard<std::mutex> lk(g_knobs_mu);
    if (!g_knobs_loaded.load(std::memory_order_relaxed)) {
      knobs_read();
      g_knobs_loaded.store(1, std::memory_order_release);
    }
  }
  return g_knobs[id];
}
}  // namespace sepr
extern "C" int sepr_knob(int id) { return sepr::knob(id); }
extern "C" void sepr_knobs_reload(void) {
  std::lock_guard<std::mutex> lk(sepr::g_knobs_mu);
  sepr::knobs_read();
  sepr::g_knobs_loaded.store(1, std::memory_order_release);
}

extern "C" int sepr_version(void) { return SEPR_VERSION; }
extern "C" const char* sepr_build_info(void) {
  return "libsepr_hip gfx950 bf16x3+f32 MFMA, fused GCFN/CLA/SpkAttn/EGA-gate " __DATE__ " " __TIME__;
}
extern "C" const char* sepr_last_hip_error(void) { return g_err; }

extern "C" size_t sepr_workspace_bytes(int op, int n, int T, int Tp, int F, int N, int S) {
  if (op == SEPR_OP_PIT) return pit_workspace_bytes(S, n, T);
  if (n <= 0 || T <= 0 || F <= 0) return 0;
  const long long M = (long long)n * T;
  switch (op) {
    case SEPR_OP_ENCODER: return ws_enc(n, T);  // T = encoder frames L
    case SEPR_OP_GCFN: return ws_gcfn(M, F);
    case SEPR_OP_CLA: return ws_cla(M, F);
    case SEPR_OP_EGA: return Tp > 0 ? ws_ega(M, (long long)n * Tp, F) : 0;
    case SEPR_OP_SPKATTN: return ws_spk(M, F);
    case SEPR_OP_SPKSPLIT: return S > 0 ? ws_split(M, n * S, T, F, S) : 0;
    case SEPR_OP_OUTLAYER: return N > 0 ? ws_out(M, F, N) : 0;  // T = output frames L
    default: return 0;
  }
}

// ---------------------------------------------------------------------------------------------------
extern "C" int sepr_encoder_fwd(const float* wav, int B, int T, const float* w_enc, int N, int K, int stride, float gn_eps,
                                float* enc, float* gn_stats, void* ws, size_t ws_bytes, sepr_stream_t stream) {
  if (!wav || !w_enc || !enc || !gn_stats || B <= 0 || T < K || stride <= 0) return SEPR_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int L = (T - K) / stride + 1;
  Arena ar(ws, ws_bytes);
  const int ntile = encoder_tiles(L);
  double* part = ar.f64((size_t)B * ntile * 2);
  if (!part) return SEPR_EWORKSPACE;
  SEPR_TRY(launch_encoder(wav, B, T, L, w_enc, N, K, stride, enc, part, st));
  SEPR_TRY(launch_gn_finalize(part, B, ntile, (long long)L * N, gn_eps, gn_stats, st));
  return SEPR_OK;
}

extern "C" int sepr_projector_fwd(const float* enc, int B, int L, int Lp, int N, int F, const float* gn_stats,
                                  const float* gn_g, const float* gn_b, const float* w, float* out, sepr_stream_t stream) {
  if (!enc || !gn_stats || !gn_g || !gn_b || !w || !out || B <= 0 || L <= 0 || Lp < L) return SEPR_EINVAL;
  GemmArgs a = gemm_args_zero();
  a.M = B * Lp; a.N = F; a.K = N;
  a.A = enc; a.lda = N;
  a.rows_out = Lp; a.rows_src = L; a.rows_valid = L;
  a.stats = gn_stats; a.stat_seq = 1; a.gamma = gn_g; a.beta = gn_b;
  a.W = w; a.bias = nullptr;
  a.Y = out; a.ldc = F;
  return launch_gemm(PRO_NORM, EPI_STORE, a, SEPR_SITE_PROJECTOR, static_cast<hipStream_t>(stream));
}

// The *_st entry points (ABI 4.10) thread the LayerNorm row statistics along the residual stream: x_stats (optional) = (mean, rstd) of x's rows
// as the previous block's *_st call returned them - the block's own statistics pass is skipped; y_stats (optional) = where the statistics of
// y's rows go (taken from the last projection's tile tail when that tile holds whole rows, else by one rowstats launch).  The values are
// those of rowstats_kernel either way (sepr_common.h rowstats_one), so a chained walk and an unchained one give identical results.
static int gcfn_fwd_impl(const float* x, const float* x_stats, float* y, float* y_stats, int n, int T, int F, const sepr_gcfn_w* w, void* ws,
                         size_t ws_bytes, sepr_stream_t stream) {
  if (!x || !y || !w || n <= 0 || T <= 0 || F <= 0 || F % 32 != 0) return SEPR_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long M = (long long)n * T;
  if (M > 0x7fffffffLL / 8) return SEPR_EINVAL;
  if (w->fused_w1p && w->fused_w2p && x != y && (F == 64 || F == 128 || F == 256)) {
    // one kernel: LayerNorm + both projections + depthwise conv + GLU + LayerScale + residual
    GcfnFusedArgs f = {};
    f.x = x; f.y = y; f.M = (int)M; f.T = T;
    f.w1p = w->fused_w1p; f.w2p = w->fused_w2p;
    f.b2 = w->b2; f.ls = w->ls; f.eps = LN_EPS; f.stagger = 0;
    SEPR_TRY(launch_gcfn_fused(f, F, SEPR_SITE_GCFN_UP, st));
    return y_stats ? launch_rowstats(y, y_stats, M, F, LN_EPS, st) : SEPR_OK;
  }
  Arena ar(ws, ws_bytes);
  float* stats_ws = ar.f32(2 * M);
  float* g = ar.f32(3LL * F * M);
  if (!ar.ok()) return SEPR_EWORKSPACE;
  const float* stats = x_stats ? x_stats : stats_ws;
  if (!x_stats) SEPR_TRY(launch_rowstats(x, stats_ws, M, F, LN_EPS, st));
  {  // net1: LayerNorm -> Linear F->6F, then depthwise k=3 + GLU on the tile while it is still in LDS:
     // the [rows, 6F] hidden tensor (3 KB per row) never goes to HBM      (network.py:61-65)
    GemmArgs a = gemm_args_zero();
    a.M = (int)M; a.N = 6 * F; a.K = F;
    a.A = x; a.lda = F; a.stats = stats; a.gamma = w->ln_g; a.beta = w->ln_b;
    a.W = w->w1; a.bias = w->b1; a.Y = g; a.ldc = 3 * F;
    a.dw_w = w->dw_w; a.dw_b = w->dw_b; a.T = T;
    SEPR_TRY(project(PRO_NORM, EPI_DWGLU, a, w->x3_up, SEPR_SITE_GCFN_UP, st));
  }
  {  // net2 Linear 3F->F, LayerScale, residual                            (network.py:65-66)
    GemmArgs a = gemm_args_zero();
    a.M = (int)M; a.N = F; a.K = 3 * F;
    a.A = g; a.lda = 3 * F; a.W = w->w2; a.bias = w->b2;
    a.Y = y; a.ldc = F; a.R = x; a.ls = w->ls;
    SEPR_TRY(project_stats(PRO_PLAIN, EPI_RES, a, w->x3_down, SEPR_SITE_GCFN_DOWN, st, y_stats));
  }
  return SEPR_OK;
}
extern "C" int sepr_gcfn_fwd(const float* x, float* y, int n, int T, int F, const sepr_gcfn_w* w, void* ws, size_t ws_bytes,
                             sepr_stream_t stream) {
  return gcfn_fwd_impl(x, nullptr, y, nullptr, n, T, F, w, ws, ws_bytes, stream);
}
extern "C" int sepr_gcfn_fwd_st(const float* x, const float* x_stats, float* y, float* y_stats, int n, int T, int F, const sepr_gcfn_w* w,
                                void* ws, size_t ws_bytes, sepr_stream_t stream) {
  return gcfn_fwd_impl(x, x_stats, y, y_stats, n, T, F, w, ws, ws_bytes, stream);
}

static int cla_fwd_impl(const float* x, const float* x_stats, float* y, float* y_stats, int n, int T, int F, int K, const sepr_cla_w* w, void* ws,
                        size_t ws_bytes, sepr_stream_t stream) {
  if (!x || !y || !w || n <= 0 || T <= 0 || F <= 0 || F % 64 != 0) return SEPR_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long M = (long long)n * T;
  if (M > 0x7fffffffLL / 8) return SEPR_EINVAL;
  Arena ar(ws, ws_bytes);
  float* stats_ws = ar.f32(2 * M);
  float* u = ar.f32((long long)F * M);
  float* c = ar.f32((long long)F * M);
  float* d = ar.f32(2LL * F * M);
  if (!ar.ok()) return SEPR_EWORKSPACE;
  if (w->fused_w1p && w->fused_w2p && w->fused_w3p && (F == 128 || F == 256) && x != y) {
    // three launches: LayerNorm+linear1+GLU | depthwise conv | linear2+BN+GELU+linear3+LayerScale+residual
    ClaFusedArgs h;
    h.x = x; h.res = nullptr; h.y = u; h.M = (int)M;
    h.w1p = w->fused_w1p; h.w2p = nullptr; h.b3 = nullptr; h.ls = nullptr; h.eps = LN_EPS;
    h.att = nullptr; h.T = 0; h.Tp = 0; h.fac = 0; h.ldy = 0; h.pool = 0; h.o = nullptr; h.att_w = nullptr; h.wop = nullptr; h.bo = nullptr; h.lso = nullptr; h.Mp = 0;
    SEPR_TRY(launch_cla_head(h, F, SEPR_SITE_CLA, st));
    SEPR_TRY(launch_dwconv_same(u, c, n, T, F, K, w->dw_w, w->dw_b, st));
    ClaFusedArgs t;
    t.x = c; t.res = x; t.y = y; t.M = (int)M;
    t.w1p = w->fused_w2p; t.w2p = w->fused_w3p; t.b3 = w->b3; t.ls = w->ls; t.eps = 0.f;
    t.att = nullptr; t.T = 0; t.Tp = 0; t.fac = 0; t.ldy = 0; t.pool = 0; t.o = nullptr; t.att_w = nullptr; t.wop = nullptr; t.bo = nullptr; t.lso = nullptr; t.Mp = 0;
    SEPR_TRY(launch_cla_tail(t, F, SEPR_SITE_CLA, st));
    return y_stats ? launch_rowstats(y, y_stats, M, F, LN_EPS, st) : SEPR_OK;
  }
  const float* stats = x_stats ? x_stats : stats_ws;
  if (!x_stats) SEPR_TRY(launch_rowstats(x, stats_ws, M, F, LN_EPS, st));
  {  // LayerNorm -> linear1 F->2F -> GLU                                  (network.py:175-177)
    GemmArgs a = gemm_args_zero();
    a.M = (int)M; a.N = 2 * F; a.K = F;
    a.A = x; a.lda = F; a.stats = stats; a.gamma = w->ln_g; a.beta = w->ln_b;
    a.W = w->w1; a.bias = w->b1; a.Y = u; a.ldc = F;
    SEPR_TRY(project(PRO_NORM, EPI_GLU, a, w->x3_1, SEPR_SITE_CLA, st));
  }
  SEPR_TRY(launch_dwconv_same(u, c, n, T, F, K, w->dw_w, w->dw_b, st));   // (network.py:178-180)
  {  // linear2 F->2F, eval BatchNorm (folded), GELU                       (network.py:181-185)
    GemmArgs a = gemm_args_zero();
    a.M = (int)M; a.N = 2 * F; a.K = F;
    a.A = c; a.lda = F; a.W = w->w2; a.bias = w->b2; a.Y = d; a.ldc = 2 * F;
    SEPR_TRY(project(PRO_PLAIN, EPI_GELU, a, w->x3_2, SEPR_SITE_CLA, st));
  }
  {  // linear3 2F->F, LayerScale, residual                                (network.py:185-187)
    GemmArgs a = gemm_args_zero();
    a.M = (int)M; a.N = F; a.K = 2 * F;
    a.A = d; a.lda = 2 * F; a.W = w->w3; a.bias = w->b3;
    a.Y = y; a.ldc = F; a.R = x; a.ls = w->ls;
    SEPR_TRY(project_stats(PRO_PLAIN, EPI_RES, a, w->x3_3, SEPR_SITE_CLA, st, y_stats));
  }
  return SEPR_OK;
}
extern "C" int sepr_cla_fwd(const float* x, float* y, int n, int T, int F, int K, const sepr_cla_w* w, void* ws,
                            size_t ws_bytes, sepr_stream_t stream) {
  return cla_fwd_impl(x, nullptr, y, nullptr, n, T, F, K, w, ws, ws_bytes, stream);
}
extern "C" int sepr_cla_fwd_st(const float* x, const float* x_stats, float* y, float* y_stats, int n, int T, int F, int K, const sepr_cla_w* w,
                               void* ws, size_t ws_bytes, sepr_stream_t stream) {
  return cla_fwd_impl(x, x_stats, y, y_stats, n, T, F, K, w, ws, ws_bytes, stream);
}

// the bf16x3 attention kernel serves the bf16x3 arithmetic mode (packed q/k/v present); SEPR_ATTN_F32=1 keeps the f32 one
static int relattn_x3(const sepr_ega_w* w) {
  static const bool f32 = [] {
    const char* e = getenv("SEPR_ATTN_F32");
    return e && e[0] == '1';
  }();
  return (w->attn.x3_qkv.wp != nullptr && !f32) ? 1 : 0;
}

static int ega_fwd_impl(const float* x, const float* x_stats, float* y, float* y_stats, int n, int T, int Tp, int F, int H, const sepr_ega_w* w,
                        void* ws, size_t ws_bytes, sepr_stream_t stream) {
  if (!x || !y || !w || x == y || n <= 0 || T <= 0 || Tp <= 0 || F <= 0 || F % 32 != 0 || T % Tp != 0) return SEPR_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int fac = T / Tp;
  const long long M = (long long)n * T, Mp = (long long)n * Tp;
  if (M > 0x7fffffffLL / 8) return SEPR_EINVAL;
  Arena ar(ws, ws_bytes);
  float* stats_ws = ar.f32(2 * M);
  float* stats_pws = ar.f32(2 * Mp);
  float* xd = ar.f32((long long)F * Mp);
  float* qkv = ar.f32(3LL * F * Mp);
  float* o = ar.f32((long long)F * Mp);
  float* att = ar.f32((long long)F * Mp);
  if (!ar.ok()) return SEPR_EWORKSPACE;
  const float* xpool = x;
  const float* stats_p = stats_pws;
  const bool qkv_fused = w->fused_qkv_p && F == 128 && !x_stats;
  if (qkv_fused) {
    // ONE launch: adaptive_avg_pool1d, LayerNorm, q / k / v (network.py:146, :99-102) - no pooled copy of x, no statistics pass
    ClaFusedArgs q;
    q.x = x; q.res = nullptr; q.y = qkv; q.M = (int)Mp;
    q.w1p = w->fused_qkv_p; q.w2p = nullptr; q.b3 = nullptr; q.ls = nullptr; q.eps = LN_EPS;
    q.att = nullptr; q.T = 0; q.Tp = 0; q.fac = 0; q.ldy = 3 * F; q.pool = fac; q.o = nullptr; q.att_w = nullptr; q.wop = nullptr; q.bo = nullptr; q.lso = nullptr; q.Mp = 0;
    SEPR_TRY(launch_ega_qkv(q, F, SEPR_SITE_ATTN_PROJ, st));
  } else
  if (fac > 1) {  // adaptive_avg_pool1d + the LayerNorm statistics of the pooled rows  (network.py:146, :99)
    SEPR_TRY(launch_pool_stats(x, xd, stats_pws, n, Tp, fac, F, LN_EPS, st));
    xpool = xd;
  } else if (x_stats) {
    stats_p = x_stats;          // no pooling: the attention's LayerNorm sees x's own rows
  } else {
    SEPR_TRY(launch_rowstats(xpool, stats_pws, Mp, F, LN_EPS, st));
  }
  if (!qkv_fused) {  // MHA: LayerNorm -> q,k,v                              (network.py:99-102)
    GemmArgs a = gemm_args_zero();
    a.M = (int)Mp; a.N = 3 * F; a.K = F;
    a.A = xpool; a.lda = F; a.stats = stats_p; a.gamma = w->attn.ln_g; a.beta = w->attn.ln_b;
    a.W = w->attn.wqkv; a.bias = w->attn.bqkv; a.Y = qkv; a.ldc = 3 * F;
    SEPR_TRY(project(PRO_NORM, EPI_STORE, a, w->attn.x3_qkv, SEPR_SITE_ATTN_PROJ, st));
  }
  SEPR_TRY(launch_relattn(qkv, o, n, Tp, F, H, w->pe_k, w->maxlen, relattn_x3(w), st, w->pe_k_planes));   // (network.py:106-122)
  const bool out_fused = w->fused_gate_p && w->fused_out_p && F == 128 && 64 % fac == 0;
  if (!out_fused) {  // linear_out * LayerScale (no residual inside MHA)      (network.py:124)
    GemmArgs a = gemm_args_zero();
    a.M = (int)Mp; a.N = F; a.K = F;
    a.A = o; a.lda = F; a.W = w->attn.wo; a.bias = w->attn.bo;
    a.Y = att; a.ldc = F; a.R = nullptr; a.ls = w->attn.ls;
    SEPR_TRY(project(PRO_PLAIN, EPI_RES, a, w->attn.x3_out, SEPR_SITE_ATTN_PROJ, st));
  }
  if (w->fused_gate_p && (F == 128 || F == 256)) {
    ClaFusedArgs g;
    g.x = x; g.res = x; g.y = y; g.M = (int)M;
    g.w1p = w->fused_gate_p; g.w2p = nullptr; g.b3 = nullptr; g.ls = nullptr; g.eps = LN_EPS;
    g.att = att; g.T = T; g.Tp = Tp; g.fac = fac; g.ldy = 0; g.pool = 0;
    g.o = nullptr; g.att_w = nullptr; g.wop = nullptr; g.bo = nullptr; g.lso = nullptr; g.Mp = 0;
    if (out_fused) {   // linear_out + LayerScale of the attention inside the gate launch (the tile's own pooled rows)
      g.o = o; g.att_w = att; g.wop = w->fused_out_p; g.bo = w->attn.bo; g.lso = w->attn.ls; g.Mp = (int)Mp;
    }
    SEPR_TRY(launch_ega_gate(g, F, SEPR_SITE_EGA_GATE, st));
    return y_stats ? launch_rowstats(y, y_stats, M, F, LN_EPS, st) : SEPR_OK;
  }
  const float* stats = x_stats ? x_stats : stats_ws;
  if (!x_stats) SEPR_TRY(launch_rowstats(x, stats_ws, M, F, LN_EPS, st));
  {  // x + sigmoid(Linear(LayerNorm(x))) * upsample(att)                   (network.py:132-135,151-153)
    GemmArgs a = gemm_args_zero();
    a.M = (int)M; a.N = F; a.K = F;
    a.A = x; a.lda = F; a.stats = stats; a.gamma = w->gate_ln_g; a.beta = w->gate_ln_b;
    a.W = w->gate_w; a.bias = w->gate_b;
    a.Y = y; a.ldc = F; a.R = x; a.aux = att; a.T = T; a.Tp = Tp; a.fac = fac;
    SEPR_TRY(project_stats(PRO_NORM, EPI_GATE, a, w->x3_gate, SEPR_SITE_EGA_GATE, st, y_stats));
  }
  return SEPR_OK;
}
extern "C" int sepr_ega_fwd(const float* x, float* y, int n, int T, int Tp, int F, int H, const sepr_ega_w* w, void* ws,
                            size_t ws_bytes, sepr_stream_t stream) {
  return ega_fwd_impl(x, nullptr, y, nullptr, n, T, Tp, F, H, w, ws, ws_bytes, stream);
}
extern "C" int sepr_ega_fwd_st(const float* x, const float* x_stats, float* y, float* y_stats, int n, int T, int Tp, int F, int H,
                               const sepr_ega_w* w, void* ws, size_t ws_bytes, sepr_stream_t stream) {
  return ega_fwd_impl(x, x_stats, y, y_stats, n, T, Tp, F, H, w, ws, ws_bytes, stream);
}

static int spkattn_fwd_impl(const float* x, const float* x_stats, float* y, float* y_stats, int nS, int S, int T, int F, int H, const sepr_mha_w* w,
                            void* ws, size_t ws_bytes, sepr_stream_t stream) {
  if (!x || !y || !w || nS <= 0 || S <= 0 || nS % S != 0 || T <= 0 || F <= 0 || F % 32 != 0) return SEPR_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long M = (long long)nS * T;
  if (M > 0x7fffffffLL / 8) return SEPR_EINVAL;
  if (w->fused_qkv_p && w->fused_out_p && S == 2 && ((F == 128 && H == 8) || (F == 256 && H == 8)) && x != y) {   // 16- / 32-channel heads
    // one kernel: LayerNorm + q/k/v + attention across the two speakers + output projection + LayerScale + residual
    SpkFusedArgs f;
    f.x = x; f.y = y; f.NF = (int)(M / 2); f.T = T;
    f.w1p = w->fused_qkv_p; f.w2p = w->fused_out_p;
    f.bo = w->bo; f.ls = w->ls; f.eps = LN_EPS; f.inv_sqrt_dk = 1.0f / sqrtf((float)(F / H));
    SEPR_TRY(launch_spk_fused(f, F, SEPR_SITE_ATTN_PROJ, st));
    return y_stats ? launch_rowstats(y, y_stats, M, F, LN_EPS, st) : SEPR_OK;
  }
  Arena ar(ws, ws_bytes);
  float* stats_ws = ar.f32(2 * M);
  float* qkv = ar.f32(3LL * F * M);
  float* o = ar.f32((long long)F * M);
  if (!ar.ok()) return SEPR_EWORKSPACE;
  if (x == y && x_stats && y_stats == x_stats) return SEPR_EINVAL;   // (in-place call: the input statistics are still read while the output's are written)
  const float* stats = x_stats ? x_stats : stats_ws;
  if (!x_stats) SEPR_TRY(launch_rowstats(x, stats_ws, M, F, LN_EPS, st));
  {
    GemmArgs a = gemm_args_zero();
    a.M = (int)M; a.N = 3 * F; a.K = F;
    a.A = x; a.lda = F; a.stats = stats; a.gamma = w->ln_g; a.beta = w->ln_b;
    a.W = w->wqkv; a.bias = w->bqkv; a.Y = qkv; a.ldc = 3 * F;
    SEPR_TRY(project(PRO_NORM, EPI_STORE, a, w->x3_qkv, SEPR_SITE_ATTN_PROJ, st));
  }
  SEPR_TRY(launch_spkmix(qkv, o, nS / S, S, T, F, H, st));
  {  // x + LayerScale(linear_out(.))                                       (network.py:124,244)
    GemmArgs a = gemm_args_zero();
    a.M = (int)M; a.N = F; a.K = F;
    a.A = o; a.lda = F; a.W = w->wo; a.bias = w->bo;
    a.Y = y; a.ldc = F; a.R = x; a.ls = w->ls;
    SEPR_TRY(project_stats(PRO_PLAIN, EPI_RES, a, w->x3_out, SEPR_SITE_ATTN_PROJ, st, y_stats));
  }
  return SEPR_OK;
}
extern "C" int sepr_spkattn_fwd(const float* x, float* y, int nS, int S, int T, int F, int H, const sepr_mha_w* w, void* ws,
                                size_t ws_bytes, sepr_stream_t stream) {
  return spkattn_fwd_impl(x, nullptr, y, nullptr, nS, S, T, F, H, w, ws, ws_bytes, stream);
}
extern "C" int sepr_spkattn_fwd_st(const float* x, const float* x_stats, float* y, float* y_stats, int nS, int S, int T, int F, int H,
                                   const sepr_mha_w* w, void* ws, size_t ws_bytes, sepr_stream_t stream) {
  return spkattn_fwd_impl(x, x_stats, y, y_stats, nS, S, T, F, H, w, ws, ws_bytes, stream);
}

extern "C" int sepr_downconv_fwd(const float* x, float* y, int n, int T, int F, int K, const sepr_down_w* w,
                                 sepr_stream_t stream) {
  if (!x || !y || !w || n <= 0 || T <= 0) return SEPR_EINVAL;
  const int pad = (K - 1) / 2;
  const int To = (T + 2 * pad - K) / 2 + 1;
  return launch_downconv(x, y, n, T, To, F, K, w->w, w->scale, w->shift, static_cast<hipStream_t>(stream));
}

extern "C" int sepr_spksplit_fwd(const float* x, float* y, int B, int S, int T, int F, float gn_eps, const sepr_split_w* w,
                                 void* ws, size_t ws_bytes, sepr_stream_t stream) {
  if (!x || !y || !w || B <= 0 || S <= 0 || T <= 0 || F <= 0 || F % 32 != 0) return SEPR_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long M = (long long)B * T;
  if (M > 0x7fffffffLL / 8) return SEPR_EINVAL;
  const int n_out = B * S;
  const long long count = (long long)T * F;
  const int nchunk = gn_chunks(count);
  Arena ar(ws, ws_bytes);
  float* z = ar.f32(2LL * F * S * M);
  double* part = ar.f64((size_t)n_out * nchunk * 2);
  float* stats = ar.f32((size_t)n_out * 2);
  if (!ar.ok()) return SEPR_EWORKSPACE;
  if (w->fused_w1p && w->fused_w2p && (F == 128 || F == 256) && x != y) {
    // one kernel per speaker: Conv1d F->4FS + GLU + the speaker's F rows of Conv1d 2FS->FS, written straight to sequence
    // b*S+s (module.py:114-116,123); the [rows, 2FS] gated tensor stays in registers
    const int nch = 2 * F * S / 32;
    const size_t half_bytes = (size_t)nch * (F / 16) * 2 * 64 * 16;
    for (int sp = 0; sp < S; ++sp) {
      GcfnFusedArgs f = {};
      f.x = x; f.y = y; f.M = (int)M; f.T = T;
      f.w1p = w->fused_w1p; f.w2p = static_cast<const char*>(w->fused_w2p) + sp * half_bytes;
      f.b2 = w->b2 + sp * F; f.ls = nullptr; f.eps = 0.f;
      f.nch = nch; f.ldy = F; f.col_off = 0; f.out_T = T; f.out_S = S; f.out_s = sp;
      SEPR_TRY(launch_glumlp_fused(f, F, SEPR_SITE_SPLIT, st));
    }
  } else {
  {  // Conv1d F->4FS (k=1) + GLU over the channel axis                     (module.py:114-115)
    GemmArgs a = gemm_args_zero();
    a.M = (int)M; a.N = 4 * F * S; a.K = F;
    a.A = x; a.lda = F; a.W = w->w1; a.bias = w->b1; a.Y = z; a.ldc = 2 * F * S;
    SEPR_TRY(project(PRO_PLAIN, EPI_GLU, a, w->x3_1, SEPR_SITE_SPLIT, st));
  }
  {  // Conv1d 2FS->FS (k=1), then view(B*S, F, T): channel s*F+f -> sequence b*S+s  (module.py:116,123)
    GemmArgs a = gemm_args_zero();
    a.M = (int)M; a.N = F * S; a.K = 2 * F * S;
    a.A = z; a.lda = 2 * F * S; a.W = w->w2; a.bias = w->b2; a.Y = y; a.ldc = F;
    a.T = T; a.S = S; a.Fs = F;
    SEPR_TRY(project(PRO_PLAIN, EPI_SPLIT, a, w->x3_2, SEPR_SITE_SPLIT, st));
  }
  }
  // GroupNorm(1, F) over (F, T) of every (b, s)                            (module.py:124)
  SEPR_TRY(launch_gn_partial(y, part, n_out, count, nchunk, st));
  SEPR_TRY(launch_gn_finalize(part, n_out, nchunk, count, gn_eps, stats, st));
  SEPR_TRY(launch_gn_apply(y, stats, w->gn_g, w->gn_b, n_out, T, F, st));
  return SEPR_OK;
}

extern "C" int sepr_fuse_fwd(const float* lo, const float* skip, float* y, int n, int T, int F, const sepr_fuse_w* w,
                             sepr_stream_t stream) {
  if (!lo || !skip || !y || !w || (!w->w && !w->x3.wp) || n <= 0 || T <= 0 || (T & 1) || F <= 0 || F % 32 != 0) return SEPR_EINVAL;
  const long long M = (long long)n * T;
  if (M > 0x7fffffffLL / 8) return SEPR_EINVAL;
  GemmArgs a = gemm_args_zero();
  a.M = (int)M; a.N = F; a.K = 2 * F;
  a.A = lo; a.lda = F; a.A2 = skip; a.lda2 = F; a.ksplit = F;
  a.rows_out = T; a.rows_src = T / 2; a.rows_valid = T; a.a_shift = 1;   // nearest x2: source frame t>>1
  a.W = w->w; a.bias = w->b; a.Y = y; a.ldc = F;
  return project(PRO_CAT2, EPI_STORE, a, w->x3, SEPR_SITE_FUSE, static_cast<hipStream_t>(stream));
}

extern "C" int sepr_outlayer_decoder_fwd(const float* x, int nS, int S, int Tsrc, int L, const int* idx, const float* enc,
                                         int F, int N, int K, int stride, const sepr_out_w* w, float* wav, void* ws,
                                         size_t ws_bytes, sepr_stream_t stream) {
  if (!x || !w || !wav || nS <= 0 || S <= 0 || nS % S != 0 || Tsrc <= 0 || L <= 0 || F % 32 != 0 || N % 4 != 0) return SEPR_EINVAL;
  if (!idx && L > Tsrc) return SEPR_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const long long M = (long long)nS * L;
  if (M > 0x7fffffffLL / 8) return SEPR_EINVAL;
  if (w->fold_w2p && w->fold_b && w->fused_w1p && (F == 128 || F == 256) && !idx && !enc && K == 16 && stride == 4 && knob(SEPR_KNOB_FOLD_HEAD)) {
    // main head (masking = False, model.py:28): end_conv1x1.2 and the ConvTranspose1d are one linear map - one launch, no [rows, N]
    // basis tensor, no workspace (launch_glumlp_fold)
    GcfnFusedArgs f = {};
    f.x = x; f.y = wav; f.T = L; f.in_src = Tsrc;
    f.w1p = w->fused_w1p; f.w2p = w->fold_w2p; f.b2 = w->fold_b;
    f.nch = 2 * F / 32; f.out_S = S; f.fold_nseq = nS; f.fold_N = N;
    return launch_glumlp_fold(f, F, SEPR_SITE_OUT, st);
  }
  Arena ar(ws, ws_bytes);
  float* o1 = ar.f32(2LL * F * M);
  float* o2 = ar.f32((long long)N * M);
  if (!ar.ok()) return SEPR_EWORKSPACE;
  // Both projections are per-frame maps, so for an upsampled head (idx != NULL) they run on the Tsrc source frames of
  // every sequence, not on the L repeated ones; the decoder kernel applies idx, the ReLU mask and the encoder product
  // while it reads the rows.  Without idx the row map is the crop of module.py:250.
  const bool unique = idx != nullptr && Tsrc <= L;   // (a down-sampling map keeps the gather in the first projection)
  const long long Mp = unique ? (long long)nS * Tsrc : M;
  if (w->fused_w1p && w->fused_w2p && (F == 128 || F == 256) && N % F == 0 && (unique || !idx)) {
    // one kernel per F basis columns: Linear F->4F + GLU + Linear 2F->N; the crop of module.py:250 is the kernel's row map
    const int nch = 2 * F / 32;
    const size_t half_bytes = (size_t)nch * (F / 16) * 2 * 64 * 16;
    for (int h = 0; h < N / F; ++h) {
      GcfnFusedArgs f = {};
      f.x = x; f.y = o2; f.M = (int)Mp; f.T = L;
      f.w1p = w->fused_w1p; f.w2p = static_cast<const char*>(w->fused_w2p) + h * half_bytes;
      f.b2 = w->b2 + F * h; f.ls = nullptr; f.eps = 0.f;
      f.nch = nch; f.ldy = N; f.col_off = F * h;
      if (!unique) { f.in_rows = L; f.in_src = Tsrc; }
      SEPR_TRY(launch_glumlp_fused(f, F, SEPR_SITE_OUT, st));
    }
  } else {
  {  // Linear F->4F, GLU                                                   (module.py:250-252, model.py:49)
    GemmArgs a = gemm_args_zero();
    a.M = (int)Mp; a.N = 4 * F; a.K = F;
    a.A = x; a.lda = F;
    if (!unique) { a.rows_out = L; a.rows_src = Tsrc; a.rows_valid = L; a.idx = idx; }
    a.W = w->w1; a.bias = w->b1; a.Y = o1; a.ldc = 2 * F;
    SEPR_TRY(project(PRO_PLAIN, EPI_GLU, a, w->x3_1, SEPR_SITE_OUT, st));
  }
  {  // Linear 2F->N                                                        (module.py:252-256)
    GemmArgs a = gemm_args_zero();
    a.M = (int)Mp; a.N = N; a.K = 2 * F;
    a.A = o1; a.lda = 2 * F; a.W = w->w2; a.bias = w->b2; a.Y = o2; a.ldc = N;
    SEPR_TRY(project(PRO_PLAIN, EPI_STORE, a, w->x3_2, SEPR_SITE_OUT, st));
  }
  }
  // ReLU(.) * encoder_output for the auxiliary heads (module.py:257-260, network.py:41) + ConvTranspose1d (:278-283)
  const int Tout = (L - 1) * stride + K;
  SEPR_TRY(launch_decoder(o2, nS, S, L, N, K, stride, w->wdec, wav, Tout, unique ? idx : nullptr, Tsrc, enc, st));
  return SEPR_OK;
}

extern "C" int sepr_groupnorm_stats(const float* x, int n, long long count, float eps, float* stats, void* ws, size_t ws_bytes,
                                    sepr_stream_t stream) {
  if (!x || !stats || n <= 0 || count <= 0 || count % 4 != 0) return SEPR_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nchunk = gn_chunks(count);
  Arena ar(ws, ws_bytes);
  double* part = ar.f64((size_t)n * nchunk * 2);
  if (!part) return SEPR_EWORKSPACE;
  SEPR_TRY(launch_gn_partial(x, part, n, count, nchunk, st));
  SEPR_TRY(launch_gn_finalize(part, n, nchunk, count, eps, stats, st));
  return SEPR_OK;
}

extern "C" int sepr_linear_fwd(const float* x, const float* w, const float* bias, float* y, int M, int N, int K,
                               sepr_stream_t stream) {
  if (!x || !w || !y || M <= 0) return SEPR_EINVAL;
  GemmArgs a = gemm_args_zero();
  a.M = M; a.N = N; a.K = K;
  a.A = x; a.lda = K; a.W = w; a.bias = bias; a.Y = y; a.ldc = N;
  return launch_gemm(PRO_PLAIN, EPI_STORE, a, SEPR_SITE_LINEAR, static_cast<hipStream_t>(stream));
}

extern "C" int sepr_linear_x3_fwd(const float* x, const void* wp, const float* bias, float* y, int M, int N, int K,
                                  sepr_stream_t stream) {
  if (!x || !wp || !y || M <= 0) return SEPR_EINVAL;
  GemmArgs a = gemm_args_zero();
  a.M = M; a.N = N; a.K = K;
  a.A = x; a.lda = K; a.Wp = wp; a.bias = bias; a.Y = y; a.ldc = N;
  return launch_gemm_x3(PRO_PLAIN, EPI_STORE, a, SEPR_SITE_LINEAR, static_cast<hipStream_t>(stream));
}
