// Training path: launchers shared by the block-level train / backward entry points of
// sepr_train_api.hip.  Kernels live in sepr_gemm_tn.hip (weight-gradient contraction), sepr_train_pw.hip (row-wise and
// element-wise forward/backward pieces, statistics, finishers) and sepr_train_attn.hip (attention forward with stored
// probabilities and its backward).
#pragma once
#include "sepr_gemm_epi.h"

namespace sepr {

// ---- dropout generator (shared by every dropout site of the training path) ---------------------------------------------
// Counter-based: one 64-bit mix (splitmix64 finaliser) of (seed, element index) per element - stateless, so the backward
// regenerates the mask of any element from the same (seed, index) without storing it.
__device__ __forceinline__ bool sepr_keep(unsigned long long seed, unsigned long long index, unsigned int thr) {
  return (unsigned int)(sepr_mix64(seed ^ sepr_mix64(index)) >> 32) >= thr;
}
// Optional per-step salt of every dropout site: a device-resident 64-bit word XOR-ed into the by-value seed at kernel start.
// It is what lets a captured hipGraph of a training step draw fresh masks on every replay (the by-value seeds are frozen into
// the graph; the host rewrites the word before each replay).  The block entry points set it from their weight struct
// (sepr_*_tw.seed_salt, may be NULL) for the duration of the call - thread-local, so concurrent callers do not interfere.
const unsigned long long* drop_salt();
struct DropSaltScope {
  const unsigned long long* prev;
  explicit DropSaltScope(const unsigned long long* s);
  ~DropSaltScope();
};
__device__ __forceinline__ unsigned long long sepr_salted(unsigned long long seed, const unsigned long long* salt) {
  return salt ? (seed ^ *salt) : seed;
}
inline unsigned int sepr_drop_threshold(float p) {
  const double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 4294967295u : (unsigned int)t;
}

// ---- cheap generator of the fused GCFN pair (sepr_gcfn_fused.hip train instantiation, sepr_gcfn_bwd_fused.hip) ----------
// The fused forward evaluates the gated-tensor mask inside its chunk loop, which is VALU-issue co-bound: a 64-bit mix per
// element (~30 instructions) would double its VALU stream.  Here one 32-bit hash (lowbias32) of (site key, row, channel
// pair) yields TWO 16-bit draws, ~6 instructions per element; element (row, c) of a site is kept iff its 16-bit draw
// (low half for even c) >= thr16 = round(p * 65536), so p_eff = thr16 / 65536 and the keep scale is 65536 / (65536 - thr16).
// (sepr_hash32, DropKey, sepr_drop_key, sepr_drop_word live in sepr_common.h: the projection epilogues use them too)
inline unsigned sepr_drop_thr16(float p) {
  const double t = (double)p * 65536.0 + 0.5;
  return t <= 0.0 ? 0u : (t >= 65535.0 ? 65535u : (unsigned)t);
}
inline float sepr_drop_scale16(float p) { return (float)(65536.0 / (65536.0 - (double)sepr_drop_thr16(p))); }

// ---------------------------------------------------------------------------------------------------------------------
// G[N][K] (+)= sum_m A[m][n] * B'[m][k]      ("TN" contraction over the M = batch x frames rows: every weight gradient)
//   A  : fp32 rows, leading dimension lda (upstream gradient of the projection's output)
//   B' : prologue(B): optional (b - mean) * rstd normalisation (LayerNorm / GroupNorm input of the projection), optional
//        two-source concat, optional row map (crop / nearest upsample / overlapping frames) - the same maps the forward
//        projection applies to its input
//   colsum[N] (+)= sum_m A[m][n]              (bias gradient, optional)
// Deterministic: the M range is split over workgroups, partial tiles go to the workspace and are summed in a fixed order.
// ---------------------------------------------------------------------------------------------------------------------
struct TnArgs {
  int M, N, K;
  const float* A;
  int lda;
  const float* B;
  int ldb;
  const float* B2;   // concat: k >= ksplit reads B2[m * ldb2 + (k - ksplit)] (never row-mapped)
  int ldb2, ksplit;
  // row map of B.  rows_out == 0: source row = m.  Else m -> (seq = m / rows_out, r = m % rows_out); the row is zero
  // unless r < rows_valid; source offset = seq * seq_stride + ((idx ? idx[r] : r) >> b_shift) * ldb.
  int rows_out, rows_valid, b_shift;
  long long seq_stride;
  const int* idx;
  int mask_a;           // 1: rows of A that the row map declares invalid (r >= rows_valid) count as zero too (colsum)
  const float* stats;   // normalisation: (mean, rstd) per row (stat_seq == 0) or per sequence (stat_seq == 1); null = none
  int stat_seq;
  int a16, b16;         // plain loader only: A / B are bf16 tensors (lda / ldb in elements), widened exactly while staged
  float* G;
  int ldg;
  int accumulate;       // 1: G += result, 0: G = result
  float* colsum;        // [N] or null
  int colsum_accumulate;
};
inline TnArgs tn_args_zero() {
  TnArgs a;
  __builtin_memset(&a, 0, sizeof(a));
  return a;
}
size_t tn_workspace_bytes(int M, int N, int K);
// x3: 1 = bf16x3 split arithmetic (3 bf16 MFMAs per product), 2 = plain bf16 operands (1 MFMA), 0 = exact f32 MFMA
int launch_gemm_tn(const TnArgs& a, int x3, void* ws, size_t ws_bytes, hipStream_t s);

// ---- row-wise / element-wise pieces (sepr_train_pw.hip) -------------------------------------------------------------
// dx[m][f] = (dres ? dres[m][f] : 0) + rstd_m * (dxh[m][f] - mean_f(dxh[m]) - xh[m][f] * mean_f(dxh[m] * xh[m])),
// xh = (x - mean_m) * rstd_m  (LayerNorm backward w.r.t. its input, affine already folded into dxh);
// optional pooled add: dx[m] += padd[(m / T) * Tp + (m % T) / fac] * (1 / fac)   (adaptive_avg_pool1d backward)
int launch_ln_bwd(const float* dxh, const float* x, const float* stats, const float* dres, const float* padd, int T, int Tp,
                  int fac, float* dx, long long M, int F, hipStream_t s);
// GLU over the last dim: a [M][2H] -> y [M][H] = a[:, :H] * sigmoid(a[:, H:]); backward da from dy and a
int launch_glu_fwd(const float* a, float* y, long long M, int H, hipStream_t s);
int launch_glu_bwd(const float* dy, const float* a, float* da, long long M, int H, hipStream_t s);
// GCFN middle backward: h1 [n,T,2C] (pre-conv hidden: C value channels then C gate channels), dg [n,T,C] ->
// dh1 [n,T,2C]; depthwise k=3 weight / bias gradients accumulated into dw_g [2C][3] (parameter layout) and db_g [2C].
// (p > 0: dg is the gradient w.r.t. the DROPPED gated tensor; the mask of generator index offset + element is applied on load)
int launch_gcfn_mid_bwd(const float* h1, const float* dg, float* dh1, int n, int T, int C, const float* dw_w, const float* dw_b,
                        float* dw_g, float* db_g, float p, unsigned long long seed, unsigned long long offset, void* ws, size_t ws_bytes,
                        hipStream_t s);
size_t gcfn_mid_bwd_ws(int n, int T, int C);
// [nblk][C][8] per-tile partials (w0 w1 w2 b of the value channel, then of the gate) -> dw_g [2C][3] / db_g [2C] (accumulated)
int launch_gcfn_mid_reduce(const float* part, int nblk, int C, float* dw_g, float* db_g, float* scratch, hipStream_t s);
size_t gcfn_mid_reduce_ws(int C);
// Fused middle of the GCFN backward (sepr_gcfn_bwd_fused.hip): recomputes the hidden tensor from x and the saved LayerNorm
// statistics, x / dy [n*T][F] -> g [n*T][3F] (dropped gated tensor), dh1 [n*T][6F], dyq [n*T][F] (dropout of dy; p > 0 only),
// depthwise weight / bias gradients accumulated.  Arithmetic follows w->up.planes (bf16x3 or plain bf16).
int launch_gcfn_bwd_fused(const float* x, const float* stats, const float* dy, int n, int T, int F, const sepr_gcfn_tw* w, void* g,
                          void* dh1, int out16 /* g, dh1 stored as bf16 */, float* dyq, float* dw_g, float* db_g, float p, unsigned long long seed,
                          const unsigned long long* salt, void* ws, size_t ws_bytes, hipStream_t st, const void* xh16 = nullptr,
                          const void* dy16 = nullptr);
// dy [M][F] -> bf16(dropout1(dy)) [M][F]: the dy plane of the plane-staged middle kernel (plain-bf16 precision)
int launch_gcfn_dyplane(const float* dy, void* out16, long long M, int F, float p, unsigned long long seed, const unsigned long long* salt,
                        hipStream_t st);
size_t gcfn_bwd_fused_ws(long long M, int F);
// depthwise conv weight gradient, stride 1, 'same' zero padding: dw[c][k] += sum_{seq,t} dy[t][c] * x[t + k - K/2][c];
// db[c] += sum dy.  x, dy [n,T,C]; dw in parameter layout [C][K]
int launch_dwconv_wgrad(const float* x, const float* dy, int n, int T, int C, int K, float* dw_g, float* db_g, void* ws,
                        size_t ws_bytes, hipStream_t s);
size_t dwconv_wgrad_ws(int n, int T, int C, int K);
// per-channel batch statistics of z [M][C] (train-mode BatchNorm1d): stats[0][C] = mean, stats[1][C] = rstd (biased var,
// eps inside); running_mean / running_var (unbiased) updated with `momentum` when non-null
int launch_colstats(const float* z, long long M, int C, float eps, float momentum, float* stats, float* run_mean, float* run_var,
                    void* ws, size_t ws_bytes, hipStream_t s);
size_t colstats_ws(long long M, int C);
// y = gelu(gamma * (z - mean) * rstd + beta)   (BatchNorm apply + exact GELU)
// out16 = 1: y is written as bf16 [M][C] (plain-bf16 precision: its readers are MFMA operand loaders)
int launch_bn_gelu_fwd(const float* z, const float* stats, const float* g, const float* b, float* y, long long M, int C,
                       hipStream_t s, int out16 = 0);
// backward of the above w.r.t. z, plus dgamma / dbeta accumulation
// out16 = 1: dz is written as bf16 [M][C] (dz must then not alias dy)
int launch_bn_gelu_bwd(const float* dy, const float* z, const float* stats, const float* g, const float* b, float* dz, float* dg_g,
                       float* db_g, long long M, int C, void* ws, size_t ws_bytes, hipStream_t s, int out16 = 0);
// EGA gate: y = x + sigmoid(zg) * att[(m / T) * Tp + (m % T) / fac]
int launch_gate_fwd(const float* x, const float* zg, const float* att, float* y, int n, int T, int Tp, int F, hipStream_t s);
// dzg = dy * att_up * sigmoid'(zg) [M][F]; datt[Mp][F] = sum over the fac frames of a pooled frame of dy * sigmoid(zg)
int launch_gate_bwd(const float* dy, const float* zg, const float* att, float* dzg, float* datt, int n, int T, int Tp, int F,
                    hipStream_t s);
// attention across the S speakers of each frame (train): forward with dropout on the probabilities, and backward qkv, dO -> dqkv
int launch_spkmix_train_fwd(const float* QKV, float* O, int B, int S, int T, int F, int H, float p, unsigned long long seed,
                            unsigned long long offset, hipStream_t s);
int launch_spkmix_bwd(const float* QKV, const float* dO, float* dQKV, int B, int S, int T, int F, int H, float p, unsigned long long seed,
                      unsigned long long offset, hipStream_t s);
// fusion conv backward glue: dcat [n,T,2F] -> dlo [n,T/2,F] (sum of the two frames that read it), dskip [n,T,F];
// acc_* != 0: add into the destination
int launch_unfuse(const float* dcat, float* dlo, float* dskip, int n, int T, int F, hipStream_t s);
// GroupNorm(1 group) backward over (T, F) of each sequence.  v = pre-norm input [n,T,F], dy upstream.
// dv written with the channel-split inverted when S > 0: sequence q = b*S + s, frame t -> dv[(b*T + t) * S*F + s*F + f]
int launch_gn_bwd(const float* dy, const float* v, const float* stats, const float* g, float* dv, float* dg_g, float* db_g, int n,
                  int T, int F, int S, void* ws, size_t ws_bytes, hipStream_t s);
size_t gn_bwd_ws(int n, int T, int F);
// y = gn(v): out-of-place GroupNorm apply (the train path keeps v)
int launch_gn_apply_oop(const float* v, const float* stats, const float* g, const float* b, float* y, int n, int T, int F,
                        hipStream_t s);
// DownConv (train): c = depthwise K stride-2 conv + bias [n,To,F]; backward dx from dc, weight / bias gradients
int launch_downconv_pre(const float* x, float* c, int n, int T, int To, int F, int K, const float* w, const float* b, hipStream_t s);
int launch_downconv_bwd(const float* x, const float* dc, float* dx, int n, int T, int To, int F, int K, const float* w, float* dw_g,
                        float* db_g, void* ws, size_t ws_bytes, hipStream_t s);
size_t downconv_bwd_ws(int n, int T, int F, int K);
// y (+)= a   (gradient accumulation where two consumers read one tensor)
int launch_add_inplace(float* y, const float* a, long long count, hipStream_t s);
// y = (x ? x : 0) + ls[f] * dropout(v)   (p = 0: no dropout): residual + LayerScale tail of the dropout-enabled paths
int launch_res_ls(const float* x, const float* v, const float* ls, float* y, long long M, int F, float p, unsigned long long seed,
                  unsigned long long offset, hipStream_t s);
// inverted dropout with a counter-based generator: y[i] = keep(seed, offset + i) ? x[i] / (1 - p) : 0 (in place allowed)
int launch_dropout(const float* x, float* y, long long count, float p, unsigned long long seed, unsigned long long offset,
                   hipStream_t s);

// the same with the 16-bit generator of the fused epilogues (EPI_RESDROP): y[m][c] = keep16(seed, site, m, c) ? x[m][c] * scale16 : 0
int launch_dropout16(const float* x, float* y, long long M, int F, float p, unsigned long long seed, unsigned site, hipStream_t s);

// ---- waveform ends (encoder / decoder) -------------------------------------------------------------------------------
// dwav [S,B,Tout] -> dwp [B*S,Tout] (sequence order b*S + s, the order of every decoder-side tensor)
int launch_permute_sb(const float* dwav, float* dwp, int S, int B, int Tout, hipStream_t s);
// ConvTranspose1d backward w.r.t. its input: dm[(seq,l)][n] = sum_k dwp[seq][stride*l + k] * wdec[k][n]
int launch_dec_bwd_dm(const float* dwp, const float* wdec, float* dm, int nS, int L, int N, int K, int stride, int Tout, hipStream_t s);
// auxiliary heads (module.py:257-260): m[(seq,l)][n] = relu(o2[(seq, idx[l])][n]) * enc[(seq / S, l)][n]
int launch_aux_m(const float* o2, const float* enc, const int* idx, float* m, int nS, int S, int Tsrc, int L, int N, hipStream_t s);
// do2[(seq,src)][n] = (o2 > 0) * sum_{l in [start[src], start[src+1])} dm[(seq,l)][n] * enc[(seq / S, l)][n]
int launch_aux_mask_bwd(const float* dm, const float* o2, const float* enc, const int* idx_start, float* do2, int nS, int S, int Tsrc,
                        int L, int N, hipStream_t s);
// denc[(b,l)][n] += sum_s dm[(b*S+s, l)][n] * relu(o2[(b*S+s, idx[l])][n])
int launch_aux_denc(const float* dm, const float* o2, const int* idx, float* denc, int nS, int S, int Tsrc, int L, int N, hipStream_t s);
// encoder: de[(b,l)][n] = (de + add) * gelu'(sum_k w[k][n] wav[b][stride*l + k])   (in place; add may be null)
int launch_enc_bwd_pre(float* de, const float* add, const float* wav, const float* w, int B, int T, int L, int N, int K, int stride,
                       hipStream_t s);

// ---- parameter-gradient finishers -------------------------------------------------------------------------------------
// projection behind a normalisation whose affine was folded into it:  y = ((xh * g + b) . W^T + bias)
//   dWh [N][K] = sum dy xh, s [N] = sum dy   ->   dW += dWh * g_k + s_n * b_k;  dbias += s;
//   dg_k += sum_n W[n][k] dWh[n][k];  db_k += sum_n s[n] W[n][k]
int launch_finish_norm_linear(const float* dWh, const float* s, const float* W, const float* g, const float* b, float* dW_g,
                              float* dbias_g, float* dg_g, float* db_g, int N, int K, hipStream_t st);
// projection followed by LayerScale:  y = ls * (v . W^T + bias)
//   Gr [N][K] = sum dy v, s [N] = sum dy  ->  dW += ls_n Gr;  dbias += ls_n s_n;  dls_n += sum_k W[n][k] Gr[n][k] + bias_n s_n
int launch_finish_linear_ls(const float* Gr, const float* s, const float* W, const float* bias, const float* ls, float* dW_g,
                            float* dbias_g, float* dls_g, int N, int K, hipStream_t st);

// Deferred finishers (round 5; include/sepr.h sepr_train_defer_begin / _flush).  The ~320 finisher launches of a training step are 5-7 us
// each - launch latency, not work (a weight-sized pass) - and nothing in the backward reads what they write (parameter gradients only).
// While a deferral window is open on the calling thread, a finisher whose inputs (the reduced contraction G and its column sums s) live in
// the window's ARENA is queued instead of launched, and the window's flush runs the queue as a few batched launches (8 jobs each, the job
// table travels by value in the kernel arguments).  fin_alloc hands out arena slots for G / s; inputs anywhere else keep the immediate launch.
float* fin_alloc(size_t count);      // nullptr: no window open on this thread, or the arena is full
int fin_flush(hipStream_t st);       // launches and empties the queue (the window stays open)
bool fin_defers(const void* p);      // p lives in the open window's arena: nothing reads it before the flush
// Riding reduction (sepr_gemm_tn.hip): the split-M reduction of a contraction whose outputs live in the window's arena waits for the next
// contraction's launch; tn_flush_pending runs a waiting one on its own (every flush does), tn_parts_set registers / clears the double buffer
// of partial tiles (include/sepr.h sepr_train_defer_parts)
int tn_flush_pending();
void tn_parts_set(void* parts, size_t bytes);

// Weight-gradient side stream (round 6; include/sepr.h sepr_train_wgrad_stream).  Nothing in a backward walk reads what the weight-gradient
// contractions write (parameter gradients, or arena slots of the deferred finishers), so while a side stream is registered on the calling
// thread they - gemm_tn + its split-M reduction, and finishers that are not deferred - are launched THERE, behind an event recorded on the
// caller's stream (their inputs are complete), and run concurrently with the input-gradient chain (HBM-bound contractions beside VALU- /
// MFMA-bound kernels).  wgrad_stream(main) returns the stream to launch on.  Joins: sepr_train_defer_flush and sepr_train_wgrad_join make the
// caller's stream wait for everything issued on the side stream; sepr_train_wgrad_mark / _wait order the re-use of a workspace.
hipStream_t wgrad_stream(hipStream_t main);
void wgrad_join(hipStream_t main);

// ---- attention with stored probabilities (sepr_train_attn.hip) ----------------------------------------------------
size_t relattn_train_ws(int n, int Tp, int F, int H);
// QKV [n,Tp,3F] -> O [n,Tp,F]; P [n,H,Tp,Tp] = softmax probabilities (saved for the backward)
// p > 0: inverted dropout on the probabilities that multiply V (P itself is stored undropped); element index of the
// generator = offset + ((n*H + h) * Tp + i) * Tp + j
int launch_relattn_train_fwd(const float* QKV, float* O, float* P, int n, int Tp, int F, int H, const float* pe_k, int maxlen, float p,
                             unsigned long long seed, unsigned long long offset, hipStream_t s);
// dO [n,Tp,F] -> dQKV [n,Tp,3F]; dpe [2*maxlen][dk] accumulated
int launch_relattn_bwd(const float* QKV, const float* P, const float* O, const float* dO, float* dQKV, float* dpe_g, int n, int Tp,
                       int F, int H, const float* pe_k, int maxlen, float p, unsigned long long seed, unsigned long long offset, void* ws,
                       size_t ws_bytes, hipStream_t s);

// ---- EGA attention on the bf16 MFMA, flash style (sepr_attention.hip TRAIN instantiation + sepr_train_attn_x3.hip) ------------
// forward: QKV [n,Tp,3F] -> O [n,Tp,F], lse [n*H,Tp] (all the backward keeps of the probabilities)
int launch_relattn_x3_train_fwd(const float* QKV, float* O, float* lse, int n, int Tp, int F, int H, const float* pe_k, int maxlen,
                                float p, unsigned long long seed, const unsigned long long* salt, hipStream_t s, int one = 0);
size_t relattn_x3_bwd_ws(int n, int Tp, int F, int H);
// dO [n,Tp,F] -> dQKV [n,Tp,3F]; dpe [2*maxlen][dk] accumulated
int launch_relattn_x3_bwd(const float* QKV, const float* lse, const float* O, const float* dO, float* dQKV, float* dpe_g, int n, int Tp,
                          int F, int H, const float* pe_k, int maxlen, float p, unsigned long long seed, const unsigned long long* salt,
                          void* ws, size_t ws_bytes, hipStream_t s, int ds16 = 0 /* dS rows as bf16: the plain-bf16 precision */,
                          int one = 0 /* ONE bf16 MFMA per product instead of the bf16x3 triple (plain-bf16 precision) */);

}  // namespace sepr
