// Row-wise and element-wise pieces of the training path: the backward of every non-projection
// op of the separator, the train-mode statistics, and the parameter-gradient finishers.  All tensors are channel-last
// fp32 rows like the forward path, so lanes sweep contiguous channels.  Every cross-row reduction (column sums for bias /
// affine / depthwise-weight gradients, BatchNorm batch statistics) is two-stage with a fixed summation order: partials per
// row chunk into the caller's workspace, then one reducing kernel - no float atomics, bit-reproducible gradients.
// Reference semantics are those of torch.autograd applied to modules/network.py / modules/module.py; each kernel cites
// the forward lines it differentiates.
#include "sepr_train.h"
#include <unordered_map>
#include <vector>

namespace sepr {

static thread_local const unsigned long long* tl_drop_salt = nullptr;
const unsigned long long* drop_salt() { return tl_drop_salt; }
DropSaltScope::DropSaltScope(const unsigned long long* s) : prev(tl_drop_salt) { tl_drop_salt = s; }
DropSaltScope::~DropSaltScope() { tl_drop_salt = prev; }


namespace {
constexpr int TPB = 256;

// (reduce16 / sum4: sepr_common.h)
// d/dx of the exact-erf GELU: Phi(x) + x * phi(x)
__device__ __forceinline__ float gelu_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
  return fmaf(x, pdf, cdf);
}
inline int grid_for(long long items, int per_block, int cap = 1 << 20) {
  const long long b = (items + per_block - 1) / per_block;
  return (int)(b < 1 ? 1 : (b < cap ? b : cap));
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm backward w.r.t. its input (torch.nn.LayerNorm, network.py:50,81,133,162), 16 lanes per row.
// The affine is folded into the projection behind it, so dxh is the gradient w.r.t. the normalised value.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void ln_bwd_kernel(const float* __restrict__ dxh, const float* __restrict__ x,
                                                    const float* __restrict__ stats, const float* __restrict__ dres,
                                                    const float* __restrict__ padd, int T, int Tp, int fac, float* __restrict__ dx,
                                                    long long M, int F) {
  const int sub = threadIdx.x & 15;
  const int nf4 = F >> 2;
  const float invF = 1.0f / (float)F;
  const float invfac = fac > 0 ? 1.0f / (float)fac : 0.f;
  for (long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4); row < M; row += (long long)gridDim.x * 16) {
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    float4 g[8], xh[8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = sub + 16 * i;
      g[i] = zero4();
      xh[i] = zero4();
      if (c < nf4) {
        g[i] = ld4(dxh + row * F + 4 * c);
        const float4 v = ld4(x + row * F + 4 * c);
        xh[i] = make_float4((v.x - mean) * rstd, (v.y - mean) * rstd, (v.z - mean) * rstd, (v.w - mean) * rstd);
        s1 += sum4(g[i]);
        s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
      }
    }
    const float m1 = reduce16(s1) * invF, m2 = reduce16(s2) * invF;
    long long prow = 0;
    if (padd) {
      const long long seq = row / T;
      prow = seq * Tp + (row - seq * T) / fac;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = sub + 16 * i;
      if (c < nf4) {
        float4 o = make_float4(rstd * (g[i].x - m1 - xh[i].x * m2), rstd * (g[i].y - m1 - xh[i].y * m2),
                               rstd * (g[i].z - m1 - xh[i].z * m2), rstd * (g[i].w - m1 - xh[i].w * m2));
        if (dres) {
          const float4 r = ld4(dres + row * F + 4 * c);
          o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        if (padd) {
          const float4 p = ld4(padd + prow * F + 4 * c);
          o.x = fmaf(p.x, invfac, o.x); o.y = fmaf(p.y, invfac, o.y); o.z = fmaf(p.z, invfac, o.z); o.w = fmaf(p.w, invfac, o.w);
        }
        st4(dx + row * F + 4 * c, o);
      }
    }
  }
}

int launch_ln_bwd(const float* dxh, const float* x, const float* stats, const float* dres, const float* padd, int T, int Tp, int fac,
                  float* dx, long long M, int F, hipStream_t s) {
  if (M <= 0) return SEPR_OK;
  if (!dxh || !x || !stats || !dx || F % 4 != 0 || F > 512 || F <= 0) return SEPR_EINVAL;
  if (padd && (T <= 0 || Tp <= 0 || fac <= 0)) return SEPR_EINVAL;
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(grid_for(M, 16)), dim3(TPB), 0, s, dxh, x, stats, dres, padd, T, Tp, fac, dx, M, F);
  SEPR_CHECK_LAUNCH("ln_bwd_kernel");
  return SEPR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// GLU over the last dim (torch.nn.GLU: network.py:54,164; module.py:115 on channels == last dim here; module.py:246)
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void glu_fwd_kernel(const float* __restrict__ a, float* __restrict__ y, long long M, int H) {
  const long long total = M * (H >> 2);
  const int h4 = H >> 2;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const long long m = i / h4;
    const int c = (int)(i - m * h4) * 4;
    const float4 v = ld4(a + m * 2 * H + c), g = ld4(a + m * 2 * H + H + c);
    st4(y + m * H + c, make_float4(v.x * sigmoid_exact(g.x), v.y * sigmoid_exact(g.y), v.z * sigmoid_exact(g.z), v.w * sigmoid_exact(g.w)));
  }
}
__global__ __launch_bounds__(TPB) void glu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ a, float* __restrict__ da,
                                                     long long M, int H) {
  const long long total = M * (H >> 2);
  const int h4 = H >> 2;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const long long m = i / h4;
    const int c = (int)(i - m * h4) * 4;
    const float4 v = ld4(a + m * 2 * H + c), g = ld4(a + m * 2 * H + H + c), d = ld4(dy + m * H + c);
    const float4 sg = make_float4(sigmoid_exact(g.x), sigmoid_exact(g.y), sigmoid_exact(g.z), sigmoid_exact(g.w));
    st4(da + m * 2 * H + c, make_float4(d.x * sg.x, d.y * sg.y, d.z * sg.z, d.w * sg.w));
    st4(da + m * 2 * H + H + c, make_float4(d.x * v.x * sg.x * (1.f - sg.x), d.y * v.y * sg.y * (1.f - sg.y),
                                            d.z * v.z * sg.z * (1.f - sg.z), d.w * v.w * sg.w * (1.f - sg.w)));
  }
}
int launch_glu_fwd(const float* a, float* y, long long M, int H, hipStream_t s) {
  if (M <= 0) return SEPR_OK;
  if (!a || !y || H <= 0 || H % 4) return SEPR_EINVAL;
  hipLaunchKernelGGL(glu_fwd_kernel, dim3(grid_for(M * (H >> 2), TPB, 1 << 16)), dim3(TPB), 0, s, a, y, M, H);
  SEPR_CHECK_LAUNCH("glu_fwd_kernel");
  return SEPR_OK;
}
int launch_glu_bwd(const float* dy, const float* a, float* da, long long M, int H, hipStream_t s) {
  if (M <= 0) return SEPR_OK;
  if (!dy || !a || !da || H <= 0 || H % 4) return SEPR_EINVAL;
  hipLaunchKernelGGL(glu_bwd_kernel, dim3(grid_for(M * (H >> 2), TPB, 1 << 16)), dim3(TPB), 0, s, dy, a, da, M, H);
  SEPR_CHECK_LAUNCH("glu_bwd_kernel");
  return SEPR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Pre-reduction of per-block partials: part [nblk][W] -> out [G][W], out[g] = sum of the rows of group g (contiguous row
// ranges, fixed order).  The parameter-gradient reducers below then only walk G <= PR_GROUPS rows per output instead of
// thousands (a thread per output looping over every row chunk of a 500 000-row launch was 5 % of a training step).
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int PR_GROUPS = 32;
__global__ __launch_bounds__(TPB) void prereduce_kernel(const float* __restrict__ part, int nblk, int W, int per_group,
                                                       float* __restrict__ out) {
  __shared__ float sh[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cl;
  const int g = blockIdx.y;
  const int b0 = g * per_group, b1 = min(nblk, b0 + per_group);
  float a = 0.f;
  if (col < W)
    for (int bk = b0 + rl; bk < b1; bk += 4) a += part[(long long)bk * W + col];
  sh[rl][cl] = a;
  __syncthreads();
  if (rl == 0 && col < W) out[(long long)g * W + col] = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
}
// returns the number of rows the caller's reducer has to walk and the buffer holding them
int prereduce(const float* part, int nblk, int W, float* scratch, const float** rows, hipStream_t s) {
  if (nblk <= 2 * PR_GROUPS) {
    *rows = part;
    return nblk;
  }
  const int per_group = (nblk + PR_GROUPS - 1) / PR_GROUPS;
  const int G = (nblk + per_group - 1) / per_group;
  hipLaunchKernelGGL(prereduce_kernel, dim3((W + 63) / 64, G), dim3(TPB), 0, s, part, nblk, W, per_group, scratch);
  *rows = scratch;
  return G;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// GCFN middle backward: depthwise Conv1d(k=3, pad=1) + GLU (network.py:62-65).
//   c[t] = b + w0 h[t-1] + w1 h[t] + w2 h[t+1] (zero padding per sequence), g = c_v * sigmoid(c_g)
//   dc_v = dg sig(c_g), dc_g = dg c_v sig (1 - sig);  dh[t] = w0 dc[t+1] + w1 dc[t] + w2 dc[t-1];
//   dw_k = sum_t dc[t] h[t+k-1];  db = sum_t dc[t]
// One lane = one (value, gate) channel pair walking GM_TC frames with the 3-frame windows in registers; weight / bias
// sums leave as per-block partials [8 values per pair].
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int GM_TC = 64;   // frames per block

__global__ __launch_bounds__(TPB) void gcfn_mid_bwd_kernel(const float* __restrict__ h1, const float* __restrict__ dg,
                                                          float* __restrict__ dh1, int T, int C, int nchunk,
                                                          const float* __restrict__ w /*[3][2C] tap-major*/,
                                                          const float* __restrict__ b /*[2C]*/, float* __restrict__ part,
                                                          unsigned int thr, float dscale, unsigned long long seed,
                                                          unsigned long long offset, const unsigned long long* __restrict__ salt) {
  seed = sepr_salted(seed, salt);
  const int c = blockIdx.y * TPB + threadIdx.x;
  if (c >= C) return;
  const int seq = blockIdx.x / nchunk, chunk = blockIdx.x - seq * nchunk;
  const int t0 = chunk * GM_TC, t1 = min(T, t0 + GM_TC);
  const int C2 = 2 * C;
  const float wv0 = w[c], wv1 = w[C2 + c], wv2 = w[2 * C2 + c];
  const float wg0 = w[C + c], wg1 = w[C2 + C + c], wg2 = w[2 * C2 + C + c];
  const float bv = b[c], bg = b[C + c];
  const float* hs = h1 + (long long)seq * T * C2;
  const float* ds = dg + (long long)seq * T * C;
  float* os = dh1 + (long long)seq * T * C2;
  auto hv = [&](int t) { return (t >= 0 && t < T) ? hs[(long long)t * C2 + c] : 0.f; };
  auto hg = [&](int t) { return (t >= 0 && t < T) ? hs[(long long)t * C2 + C + c] : 0.f; };
  // dc at frame t (zero outside the sequence)
  float a_wv[3] = {0.f, 0.f, 0.f}, a_wg[3] = {0.f, 0.f, 0.f}, a_bv = 0.f, a_bg = 0.f;
  float dcv_m = 0.f, dcg_m = 0.f, dcv_c = 0.f, dcg_c = 0.f;   // dc[t-1], dc[t]
  // sliding windows of h: h[t'-1], h[t'], h[t'+1] for the frame t' whose dc is being formed
  float hv_m = hv(t0 - 2), hv_c = hv(t0 - 1), hv_p = hv(t0);
  float hg_m = hg(t0 - 2), hg_c = hg(t0 - 1), hg_p = hg(t0);
  auto form_dc = [&](int tp, float& dcv, float& dcg, bool own) {
    // on entry the windows hold h[tp-1], h[tp], h[tp+1]
    dcv = 0.f;
    dcg = 0.f;
    if (tp >= 0 && tp < T) {
      const float cv = fmaf(wv2, hv_p, fmaf(wv1, hv_c, fmaf(wv0, hv_m, bv)));
      const float cgt = fmaf(wg2, hg_p, fmaf(wg1, hg_c, fmaf(wg0, hg_m, bg)));
      const float sg = sigmoid_exact(cgt);
      float d = ds[(long long)tp * C + c];
      if (thr)   // the forward's dropout on the gated tensor (network.py:55): same generator index = element of [M][C]
        d = sepr_keep(seed, offset + (unsigned long long)(((long long)seq * T + tp) * C + c), thr) ? d * dscale : 0.f;
      dcv = d * sg;
      dcg = d * cv * sg * (1.f - sg);
      if (own) {
        a_wv[0] = fmaf(dcv, hv_m, a_wv[0]); a_wv[1] = fmaf(dcv, hv_c, a_wv[1]); a_wv[2] = fmaf(dcv, hv_p, a_wv[2]);
        a_wg[0] = fmaf(dcg, hg_m, a_wg[0]); a_wg[1] = fmaf(dcg, hg_c, a_wg[1]); a_wg[2] = fmaf(dcg, hg_p, a_wg[2]);
        a_bv += dcv;
        a_bg += dcg;
      }
    }
  };
  auto advance = [&](int tp_next) {   // slide the h windows so that they are centred on tp_next
    hv_m = hv_c; hv_c = hv_p; hv_p = hv(tp_next + 1);
    hg_m = hg_c; hg_c = hg_p; hg_p = hg(tp_next + 1);
  };
  // prologue: dc[t0-1] (halo, not owned), dc[t0]
  form_dc(t0 - 1, dcv_m, dcg_m, false);
  advance(t0);
  form_dc(t0, dcv_c, dcg_c, true);
  for (int t = t0; t < t1; ++t) {
    float dcv_p, dcg_p;
    advance(t + 1);
    form_dc(t + 1, dcv_p, dcg_p, t + 1 < t1);
    // dh[t] = w0 dc[t+1] + w1 dc[t] + w2 dc[t-1]
    os[(long long)t * C2 + c] = fmaf(wv0, dcv_p, fmaf(wv1, dcv_c, wv2 * dcv_m));
    os[(long long)t * C2 + C + c] = fmaf(wg0, dcg_p, fmaf(wg1, dcg_c, wg2 * dcg_m));
    dcv_m = dcv_c; dcv_c = dcv_p;
    dcg_m = dcg_c; dcg_c = dcg_p;
  }
  float* p = part + ((long long)blockIdx.x * C + c) * 8;
  p[0] = a_wv[0]; p[1] = a_wv[1]; p[2] = a_wv[2]; p[3] = a_bv;
  p[4] = a_wg[0]; p[5] = a_wg[1]; p[6] = a_wg[2]; p[7] = a_bg;
}

__global__ __launch_bounds__(TPB) void gcfn_mid_reduce_kernel(const float* __restrict__ part, int nblk, int C, float* __restrict__ dw_g,
                                                             float* __restrict__ db_g) {
  const int i = blockIdx.x * TPB + threadIdx.x;     // (pair c, slot 0..7)
  if (i >= C * 8) return;
  const int c = i >> 3, slot = i & 7;
  float s = 0.f;
  for (int bk = 0; bk < nblk; ++bk) s += part[((long long)bk * C + c) * 8 + slot];
  const int ch = (slot < 4) ? c : C + c;
  const int k = slot & 3;
  if (k < 3) dw_g[ch * 3 + k] += s;
  else db_g[ch] += s;
}
}  // namespace

size_t gcfn_mid_bwd_ws(int n, int T, int C) {
  const int nchunk = (T + GM_TC - 1) / GM_TC;
  return align_up((size_t)n * nchunk * C * 8 * sizeof(float)) + align_up((size_t)PR_GROUPS * C * 8 * sizeof(float));
}
// reduction of [nblk][C][8] depthwise partials (the fused backward's tiles, sepr_gcfn_bwd_fused.hip) into the parameter gradients
size_t gcfn_mid_reduce_ws(int C) { return align_up((size_t)PR_GROUPS * C * 8 * sizeof(float)); }
int launch_gcfn_mid_reduce(const float* part, int nblk, int C, float* dw_g, float* db_g, float* scratch, hipStream_t s) {
  if (nblk <= 0) return SEPR_OK;
  if (!part || !dw_g || !db_g || !scratch || C <= 0) return SEPR_EINVAL;
  const float* rows = nullptr;
  const int nrows = prereduce(part, nblk, C * 8, scratch, &rows, s);
  hipLaunchKernelGGL(gcfn_mid_reduce_kernel, dim3((C * 8 + TPB - 1) / TPB), dim3(TPB), 0, s, rows, nrows, C, dw_g, db_g);
  SEPR_CHECK_LAUNCH("gcfn_mid_reduce_kernel");
  return SEPR_OK;
}
int launch_gcfn_mid_bwd(const float* h1, const float* dg, float* dh1, int n, int T, int C, const float* dw_w, const float* dw_b,
                        float* dw_g, float* db_g, float p, unsigned long long seed, unsigned long long offset, void* ws, size_t ws_bytes,
                        hipStream_t s) {
  if (n <= 0 || T <= 0) return SEPR_OK;
  if (!h1 || !dg || !dh1 || !dw_w || !dw_b || !dw_g || !db_g || C <= 0) return SEPR_EINVAL;
  if (!ws || ws_bytes < gcfn_mid_bwd_ws(n, T, C)) return SEPR_EWORKSPACE;
  const int nchunk = (T + GM_TC - 1) / GM_TC;
  float* part = static_cast<float*>(ws);
  hipLaunchKernelGGL(gcfn_mid_bwd_kernel, dim3(n * nchunk, (C + TPB - 1) / TPB), dim3(TPB), 0, s, h1, dg, dh1, T, C, nchunk, dw_w,
                     dw_b, part, p > 0.f ? sepr_drop_threshold(p) : 0u, p > 0.f ? 1.0f / (1.0f - p) : 1.0f, seed, offset, drop_salt());
  float* scratch = reinterpret_cast<float*>(static_cast<char*>(ws) + align_up((size_t)n * nchunk * C * 8 * sizeof(float)));
  const float* rows = nullptr;
  const int nrows = prereduce(part, n * nchunk, C * 8, scratch, &rows, s);
  hipLaunchKernelGGL(gcfn_mid_reduce_kernel, dim3((C * 8 + TPB - 1) / TPB), dim3(TPB), 0, s, rows, nrows, C, dw_g, db_g);
  SEPR_CHECK_LAUNCH("gcfn_mid_bwd_kernel");
  return SEPR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Depthwise 'same' conv weight gradient (CLA dw_conv_1d, network.py:165,179): dw[c][k] = sum dy[t][c] x[t + k - K/2][c].
// Block = (sequence, 64-frame chunk, 64 channels); x chunk (+halo) and dy chunk staged in LDS; lane = channel, wave = tap group.
// A wave owns 16 consecutive taps per pass (taps 64 P + 16 w ...): their 16 accumulators and a sliding window of x live in
// registers, so a frame costs two LDS reads (dy[r], one new x) for 16 FMAs (8 packed).  The K % 64 left-over taps (one for K = 65) go
// round-robin over the waves with a plain two-read loop; the bias column is the sum of dy (wave 0).  Partials [block][K + 1][C].
// (First version: every FMA read both operands from LDS - 2.1 MB of LDS reads per block, 150 us per launch on average.)
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int WG_TC = 64, WG_CB = 64, WG_KMAX = 129;   // (64 + 128 + 16 + 64) x 64 x 4 B = 68 KB of LDS at most

// nc (round 6): a block walks nc consecutive 64-frame chunks of its sequence with the tap accumulators kept in registers and writes ONE partial
// row - an eighth of the partial rows (and of the pre-reduction behind the kernel) at nc = 8.  The launcher only asks for nc > 1 when every thread
// has at most one left-over tap (K % 64 < 4) and at most two 64-tap passes (K < 192): K = 65, every shipped configuration.
__global__ __launch_bounds__(TPB) void dwconv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, int T, int C,
                                                          int K, int nchunk, float* __restrict__ part, int nc, int ngrp) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int pad = K / 2;
  const int rows_x = WG_TC + K - 1 + 16;   // + 16 zero rows: the window of the last full tap group reads one group ahead
  float* xs = sm;                       // [rows_x][64]
  float* ds = sm + rows_x * WG_CB;      // [WG_TC][64]
  const int seq = blockIdx.x / ngrp, grp = blockIdx.x - seq * ngrp;
  const int c0 = blockIdx.y * WG_CB;
  const int cl = threadIdx.x & 63, kl = threadIdx.x >> 6;
  const float* xq = x + (long long)seq * T * C;
  const float* dq = dy + (long long)seq * T * C;
  const bool cok = c0 + cl < C;
  const int npass = K / 64;
  typedef float f2 __attribute__((ext_vector_type(2)));
  const bool keep = nc > 1;             // accumulate over the block's chunks (at most two passes, one left-over tap per thread)
  f2 accP[2][8];
  float accL = 0.f;
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int m = 0; m < 8; ++m) accP[q][m] = (f2){0.f, 0.f};
  float* p = part + (long long)blockIdx.x * (K + 1) * C;
  const int ch_end = min(nchunk, (grp + 1) * nc);
  for (int chunk = grp * nc; chunk < ch_end; ++chunk) {
    const int t0 = chunk * WG_TC;
    if (chunk != grp * nc) __syncthreads();                 // every wave is done with the previous chunk's tiles
    for (int i = threadIdx.x; i < rows_x * (WG_CB / 4); i += TPB) {   // C % 4 == 0: a float4 is inside or outside the channel range
      const int r = i >> 4, cc = (i & 15) * 4;
      const int t = t0 - pad + r;
      const bool in = r < WG_TC + K - 1 && t >= 0 && t < T && c0 + cc < C;
      st4(xs + r * WG_CB + cc, in ? ld4(xq + (long long)t * C + c0 + cc) : zero4());
    }
    for (int i = threadIdx.x; i < WG_TC * (WG_CB / 4); i += TPB) {
      const int r = i >> 4, cc = (i & 15) * 4;
      const int t = t0 + r;
      st4(ds + r * WG_CB + cc, (t < T && c0 + cc < C) ? ld4(dq + (long long)t * C + c0 + cc) : zero4());
    }
    __syncthreads();
    const bool last = chunk + 1 == ch_end;
    // one 64-tap pass; `acc` is the pass's accumulator set (zero at the start of a chunk unless the block keeps them across its chunks)
    auto pass = [&](int ps, f2 (&acc)[8]) {
      const int k0 = 64 * ps + 16 * kl;
      const float* xk = xs + k0 * WG_CB + cl;               // xk[q * 64] = x~[q]: the x value tap k0 + j meets at frame q - j
      // Packed FMAs want (tap 2m, tap 2m + 1) operand PAIRS in adjacent registers, and the window slides by one value per frame, so
      // two rings of pairs are kept: E[p] = (x~[2p], x~[2p+1]) serves even frames, O[p] = (x~[2p+1], x~[2p+2]) odd frames.
      // (A single 16-value ring compiled to 3 register moves per packed FMA - slower than the two-reads-per-FMA version.)
      f2 E[8], O[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        if (!keep) acc[m] = (f2){0.f, 0.f};
        E[m] = (f2){xk[(2 * m) * WG_CB], xk[(2 * m + 1) * WG_CB]};
        O[m] = (f2){xk[(2 * m + 1) * WG_CB], xk[(2 * m + 2) * WG_CB]};
      }
#pragma unroll 1
      for (int u0 = 0; u0 < WG_TC / 2; u0 += 8) {
#pragma unroll
        for (int uu = 0; uu < 8; ++uu) {                    // frames 2u, 2u + 1; slot uu holds E[u], O[u]
          const int u = u0 + uu;
          const float d0 = ds[(2 * u) * WG_CB + cl], d1 = ds[(2 * u + 1) * WG_CB + cl];
          const f2 dd0 = (f2){d0, d0}, dd1 = (f2){d1, d1};
#pragma unroll
          for (int m = 0; m < 8; ++m) acc[m] = __builtin_elementwise_fma(dd0, E[(uu + m) & 7], acc[m]);
#pragma unroll
          for (int m = 0; m < 8; ++m) acc[m] = __builtin_elementwise_fma(dd1, O[(uu + m) & 7], acc[m]);
          const float n1 = xk[(2 * u + 17) * WG_CB], n2 = xk[(2 * u + 18) * WG_CB];
          E[uu] = (f2){O[(uu + 7) & 7].y, n1};              // E[u + 8] = (x~[2u+16], x~[2u+17])
          O[uu] = (f2){n1, n2};                             // O[u + 8]
        }
      }
      if (cok && (last || !keep))
#pragma unroll
        for (int m = 0; m < 8; ++m) {
          p[(long long)(k0 + 2 * m) * C + c0 + cl] = acc[m].x;
          p[(long long)(k0 + 2 * m + 1) * C + c0 + cl] = acc[m].y;
        }
    };
    if (keep) {
      if (npass > 0) pass(0, accP[0]);
      if (npass > 1) pass(1, accP[1]);
    } else {
      for (int ps = 0; ps < npass; ++ps) pass(ps, accP[0]);
    }
    for (int k = 64 * npass + kl; k <= K; k += 4) {          // left-over taps; k == K: the bias column
      float acc = keep ? accL : 0.f;
      if (k < K) {
#pragma unroll 8
        for (int r = 0; r < WG_TC; ++r) acc = fmaf(ds[r * WG_CB + cl], xs[(r + k) * WG_CB + cl], acc);
      } else {
#pragma unroll 8
        for (int r = 0; r < WG_TC; ++r) acc += ds[r * WG_CB + cl];
      }
      accL = acc;
      if (cok && (last || !keep)) p[(long long)k * C + c0 + cl] = acc;
    }
  }
}
__global__ __launch_bounds__(TPB) void dwconv_wgrad_reduce_kernel(const float* __restrict__ part, int nblk, int C, int K,
                                                                 float* __restrict__ dw_g, float* __restrict__ db_g) {
  const int i = blockIdx.x * TPB + threadIdx.x;     // (k, c)
  if (i >= (K + 1) * C) return;
  const int k = i / C, c = i - k * C;
  float s = 0.f;
  for (int bk = 0; bk < nblk; ++bk) s += part[(long long)bk * (K + 1) * C + i];
  if (k < K) dw_g[c * K + k] += s;
  else db_g[c] += s;
}
}  // namespace

size_t dwconv_wgrad_ws(int n, int T, int C, int K) {
  const int nchunk = (T + WG_TC - 1) / WG_TC;
  return align_up((size_t)n * nchunk * (K + 1) * C * sizeof(float)) + align_up((size_t)PR_GROUPS * (K + 1) * C * sizeof(float));
}
int launch_dwconv_wgrad(const float* x, const float* dy, int n, int T, int C, int K, float* dw_g, float* db_g, void* ws,
                        size_t ws_bytes, hipStream_t s) {
  if (n <= 0 || T <= 0) return SEPR_OK;
  if (!x || !dy || !dw_g || !db_g || C <= 0 || (C % 4) || K <= 0 || (K & 1) == 0 || K > WG_KMAX) return SEPR_EINVAL;
  if (!ws || ws_bytes < dwconv_wgrad_ws(n, T, C, K)) return SEPR_EWORKSPACE;
  const int nchunk = (T + WG_TC - 1) / WG_TC;
  const size_t shm = (size_t)((WG_TC + K - 1 + 16) + WG_TC) * WG_CB * sizeof(float);
  float* part = static_cast<float*>(ws);
  if (shm > 64 * 1024) {   // K > 113: above the default dynamic-LDS limit
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(dwconv_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (attr != hipSuccess) return SEPR_EINVAL;
  }
  // chunks per block (see the kernel): up to 8 while the launch keeps ~4 blocks per CU (4 -> 8: 71.6 -> 71.3 ms per bf16 step in one call,
  // profiles/r06_dwwg_nc_ab.txt); SEPR_DWWG_NC=1: one chunk per block (rounds 2-5)
  static const int nc_max = [] { const char* e = getenv("SEPR_DWWG_NC"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 8; }();
  const int cblk = (C + WG_CB - 1) / WG_CB;
  int nc = 1;
  if (K % 64 < 4 && K / 64 <= 2)
    while (nc < nc_max && (long long)n * ((nchunk + 2 * nc - 1) / (2 * nc)) * cblk >= 1024) nc *= 2;
  const int ngrp = (nchunk + nc - 1) / nc;
  hipLaunchKernelGGL(dwconv_wgrad_kernel, dim3(n * ngrp, cblk), dim3(TPB), shm, s, x, dy, T, C, K, nchunk, part, nc, ngrp);
  float* scratch = reinterpret_cast<float*>(static_cast<char*>(ws) + align_up((size_t)n * nchunk * (K + 1) * C * sizeof(float)));
  const float* rows = nullptr;
  const int nrows = prereduce(part, n * ngrp, (K + 1) * C, scratch, &rows, s);
  hipLaunchKernelGGL(dwconv_wgrad_reduce_kernel, dim3(((K + 1) * C + TPB - 1) / TPB), dim3(TPB), 0, s, rows, nrows, C, K, dw_g, db_g);
  SEPR_CHECK_LAUNCH("dwconv_wgrad_kernel");
  return SEPR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Column reductions over the M rows of [M][C] tensors (fp64 partials per 512-row chunk, fixed-order combine):
//   MODE 0  (z, z^2)                      -> train-mode BatchNorm1d statistics (network.py:167,183; module.py:69,75)
//   MODE 1  (dpre, dpre * zh) with dpre = dy * gelu'(g zh + b)   -> BatchNorm + GELU backward sums
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int CR_ROWS = 512;

template <int MODE>
__global__ __launch_bounds__(TPB) void colred_kernel(const float* __restrict__ z, const float* __restrict__ dy,
                                                    const float* __restrict__ stats, const float* __restrict__ g,
                                                    const float* __restrict__ b, long long M, int C, double* __restrict__ part) {
  __shared__ double sh[16][64][2];
  const int q4 = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.y * 64 + 4 * q4;
  const long long r0 = (long long)blockIdx.x * CR_ROWS;
  const long long r1 = (r0 + CR_ROWS < M) ? r0 + CR_ROWS : M;
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  if (c < C) {
    float4 mean = zero4(), rstd = zero4(), gg = zero4(), bb = zero4();
    if (MODE == 1) {
      mean = ld4(stats + c); rstd = ld4(stats + C + c); gg = ld4(g + c); bb = ld4(b + c);
    }
    for (long long r = r0 + rl; r < r1; r += 16) {
      const float4 v = ld4(z + r * C + c);
      float a[4] = {v.x, v.y, v.z, v.w};
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s1[j] += (double)a[j];
          s2[j] = fma((double)a[j], (double)a[j], s2[j]);
        }
      } else {
        const float4 d = ld4(dy + r * C + c);
        const float dd[4] = {d.x, d.y, d.z, d.w};
        const float mu[4] = {mean.x, mean.y, mean.z, mean.w}, rs[4] = {rstd.x, rstd.y, rstd.z, rstd.w};
        const float ga[4] = {gg.x, gg.y, gg.z, gg.w}, be[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float zh = (a[j] - mu[j]) * rs[j];
          const float dpre = dd[j] * gelu_grad(fmaf(ga[j], zh, be[j]));
          s1[j] += (double)dpre;
          s2[j] = fma((double)dpre, (double)zh, s2[j]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sh[rl][4 * q4 + j][0] = s1[j];
    sh[rl][4 * q4 + j][1] = s2[j];
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    const int cc = threadIdx.x >> 1, w = threadIdx.x & 1;
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += sh[r][cc][w];
    const int ch = blockIdx.y * 64 + cc;
    if (ch < C) part[((long long)blockIdx.x * C + ch) * 2 + w] = s;
  }
}

// Combine of the per-chunk fp64 partials: a workgroup owns 16 columns, 16 threads per column walk the chunks (thread j takes chunks
// j, j + 16, ...), then a fixed-order sum of the 16 lanes' sums - the order depends on nblk only, so the result is run-to-run identical.
// (First version: one thread per column walking all M / 512 chunks serially - 40 us at 128 k rows, as long as the reduction itself.)
__device__ __forceinline__ void colpart_sum(const double* __restrict__ part, int nblk, int C, int c, double& o1, double& o2) {
  __shared__ double shp[16][16][2];
  const int cl = threadIdx.x & 15, j = threadIdx.x >> 4;
  double s1 = 0.0, s2 = 0.0;
  if (c < C)
    for (int bk = j; bk < nblk; bk += 16) {
      const double2 v = *reinterpret_cast<const double2*>(part + ((long long)bk * C + c) * 2);
      s1 += v.x;
      s2 += v.y;
    }
  shp[j][cl][0] = s1;
  shp[j][cl][1] = s2;
  __syncthreads();
  o1 = 0.0;
  o2 = 0.0;
  if (j == 0)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      o1 += shp[r][cl][0];
      o2 += shp[r][cl][1];
    }
}

__global__ __launch_bounds__(TPB) void colstats_final_kernel(const double* __restrict__ part, int nblk, long long M, int C, float eps,
                                                            float momentum, float* __restrict__ stats, float* __restrict__ run_mean,
                                                            float* __restrict__ run_var) {
  const int c = blockIdx.x * 16 + (threadIdx.x & 15);
  double s, ss;
  colpart_sum(part, nblk, C, c, s, ss);
  if (c >= C || (threadIdx.x >> 4) != 0) return;
  const double mean = s / (double)M;
  double var = ss / (double)M - mean * mean;
  var = var > 0.0 ? var : 0.0;
  stats[c] = (float)mean;
  stats[C + c] = (float)(1.0 / sqrt(var + (double)eps));
  if (run_mean && run_var) {   // torch: running_var uses the unbiased estimate
    const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
    run_mean[c] = (float)((1.0 - momentum) * (double)run_mean[c] + momentum * mean);
    run_var[c] = (float)((1.0 - momentum) * (double)run_var[c] + momentum * unb);
  }
}

__global__ __launch_bounds__(TPB) void colred_final2_kernel(const double* __restrict__ part, int nblk, int C, float* __restrict__ sums,
                                                           float* __restrict__ dg_g, float* __restrict__ db_g) {
  const int c = blockIdx.x * 16 + (threadIdx.x & 15);
  double s1, s2;
  colpart_sum(part, nblk, C, c, s1, s2);
  if (c >= C || (threadIdx.x >> 4) != 0) return;
  sums[c] = (float)s1;
  sums[C + c] = (float)s2;
  if (db_g) db_g[c] += (float)s1;
  if (dg_g) dg_g[c] += (float)s2;
}

__global__ __launch_bounds__(TPB) void bn_gelu_fwd_kernel(const float* __restrict__ z, const float* __restrict__ stats,
                                                         const float* __restrict__ g, const float* __restrict__ b, float* __restrict__ y,
                                                         long long M, int C, int out16) {
  const int c4 = C >> 2;
  const long long total = M * c4;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const long long m = i / c4;
    const int c = (int)(i - m * c4) * 4;
    const float4 v = ld4(z + m * C + c), mu = ld4(stats + c), rs = ld4(stats + C + c), ga = ld4(g + c), be = ld4(b + c);
    const float4 o = make_float4(gelu_exact(fmaf(ga.x, (v.x - mu.x) * rs.x, be.x)), gelu_exact(fmaf(ga.y, (v.y - mu.y) * rs.y, be.y)),
                                 gelu_exact(fmaf(ga.z, (v.z - mu.z) * rs.z, be.z)), gelu_exact(fmaf(ga.w, (v.w - mu.w) * rs.w, be.w)));
    if (out16) st4_bf16(y, m * C + c, o);      // plain-bf16 precision: the only readers are MFMA operand loaders that round to bf16
    else st4(y + m * C + c, o);
  }
}
__global__ __launch_bounds__(TPB) void bn_gelu_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                               const float* __restrict__ stats, const float* __restrict__ g,
                                                               const float* __restrict__ b, const float* __restrict__ sums,
                                                               float* __restrict__ dz, long long M, int C, int out16) {
  const int c4 = C >> 2;
  const long long total = M * c4;
  const float invM = 1.0f / (float)M;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const long long m = i / c4;
    const int c = (int)(i - m * c4) * 4;
    const float4 v = ld4(z + m * C + c), d = ld4(dy + m * C + c);
    const float4 mu = ld4(stats + c), rs = ld4(stats + C + c), ga = ld4(g + c), be = ld4(b + c);
    const float4 s1 = ld4(sums + c), s2 = ld4(sums + C + c);
    const float vv[4] = {v.x, v.y, v.z, v.w}, dd[4] = {d.x, d.y, d.z, d.w};
    const float m_[4] = {mu.x, mu.y, mu.z, mu.w}, r_[4] = {rs.x, rs.y, rs.z, rs.w}, g_[4] = {ga.x, ga.y, ga.z, ga.w};
    const float b_[4] = {be.x, be.y, be.z, be.w}, a1[4] = {s1.x, s1.y, s1.z, s1.w}, a2[4] = {s2.x, s2.y, s2.z, s2.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float zh = (vv[j] - m_[j]) * r_[j];
      const float dpre = dd[j] * gelu_grad(fmaf(g_[j], zh, b_[j]));
      o[j] = g_[j] * r_[j] * (dpre - a1[j] * invM - zh * (a2[j] * invM));
    }
    if (out16) st4_bf16(dz, m * C + c, make_float4(o[0], o[1], o[2], o[3]));
    else st4(dz + m * C + c, make_float4(o[0], o[1], o[2], o[3]));
  }
}
}  // namespace

size_t colstats_ws(long long M, int C) {
  const long long nblk = (M + CR_ROWS - 1) / CR_ROWS;
  return align_up((size_t)nblk * C * 2 * sizeof(double)) + align_up((size_t)2 * C * sizeof(float));
}
int launch_colstats(const float* z, long long M, int C, float eps, float momentum, float* stats, float* run_mean, float* run_var,
                    void* ws, size_t ws_bytes, hipStream_t s) {
  if (M <= 0) return SEPR_EINVAL;
  if (!z || !stats || C <= 0 || C % 4) return SEPR_EINVAL;
  if (!ws || ws_bytes < colstats_ws(M, C)) return SEPR_EWORKSPACE;
  const int nblk = (int)((M + CR_ROWS - 1) / CR_ROWS);
  double* part = static_cast<double*>(ws);
  hipLaunchKernelGGL((colred_kernel<0>), dim3(nblk, (C + 63) / 64), dim3(TPB), 0, s, z, (const float*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, (const float*)nullptr, M, C, part);
  hipLaunchKernelGGL(colstats_final_kernel, dim3((C + 15) / 16), dim3(TPB), 0, s, part, nblk, M, C, eps, momentum, stats, run_mean,
                     run_var);
  SEPR_CHECK_LAUNCH("colstats kernels");
  return SEPR_OK;
}
int launch_bn_gelu_fwd(const float* z, const float* stats, const float* g, const float* b, float* y, long long M, int C, hipStream_t s,
                       int out16) {
  if (M <= 0) return SEPR_OK;
  if (!z || !stats || !g || !b || !y || C <= 0 || C % 4) return SEPR_EINVAL;
  hipLaunchKernelGGL(bn_gelu_fwd_kernel, dim3(grid_for(M * (C >> 2), TPB, 1 << 16)), dim3(TPB), 0, s, z, stats, g, b, y, M, C, out16);
  SEPR_CHECK_LAUNCH("bn_gelu_fwd_kernel");
  return SEPR_OK;
}
int launch_bn_gelu_bwd(const float* dy, const float* z, const float* stats, const float* g, const float* b, float* dz, float* dg_g,
                       float* db_g, long long M, int C, void* ws, size_t ws_bytes, hipStream_t s, int out16) {
  if (M <= 0) return SEPR_OK;
  if (!dy || !z || !stats || !g || !b || !dz || C <= 0 || C % 4) return SEPR_EINVAL;
  if (!ws || ws_bytes < colstats_ws(M, C)) return SEPR_EWORKSPACE;
  const int nblk = (int)((M + CR_ROWS - 1) / CR_ROWS);
  double* part = static_cast<double*>(ws);
  float* sums = reinterpret_cast<float*>(static_cast<char*>(ws) + align_up((size_t)nblk * C * 2 * sizeof(double)));
  hipLaunchKernelGGL((colred_kernel<1>), dim3(nblk, (C + 63) / 64), dim3(TPB), 0, s, z, dy, stats, g, b, M, C, part);
  hipLaunchKernelGGL(colred_final2_kernel, dim3((C + 15) / 16), dim3(TPB), 0, s, part, nblk, C, sums, dg_g, db_g);
  if (out16 && static_cast<const void*>(dz) == static_cast<const void*>(dy)) return SEPR_EINVAL;      // a bf16 dz cannot overwrite the fp32 dy in place
  hipLaunchKernelGGL(bn_gelu_bwd_apply_kernel, dim3(grid_for(M * (C >> 2), TPB, 1 << 16)), dim3(TPB), 0, s, dy, z, stats, g, b, sums, dz,
                     M, C, out16);
  SEPR_CHECK_LAUNCH("bn_gelu_bwd kernels");
  return SEPR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// EGA gate (network.py:151-153): y = x + sigmoid(zg) * upsample(att); one thread per (pooled frame, 4 channels)
// ---------------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(TPB) void gate_fwd_kernel(const float* __restrict__ x, const float* __restrict__ zg,
                                                      const float* __restrict__ att, float* __restrict__ y, long long Mp, int fac, int F) {
  const int f4 = F >> 2;
  const long long total = Mp * f4;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const long long mp = i / f4;
    const int c = (int)(i - mp * f4) * 4;
    const float4 a = ld4(att + mp * F + c);
    for (int j = 0; j < fac; ++j) {
      const long long m = mp * fac + j;
      const float4 xv = ld4(x + m * F + c), z = ld4(zg + m * F + c);
      st4(y + m * F + c, make_float4(fmaf(sigmoid_exact(z.x), a.x, xv.x), fmaf(sigmoid_exact(z.y), a.y, xv.y),
                                     fmaf(sigmoid_exact(z.z), a.z, xv.z), fmaf(sigmoid_exact(z.w), a.w, xv.w)));
    }
  }
}
__global__ __launch_bounds__(TPB) void gate_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ zg,
                                                      const float* __restrict__ att, float* __restrict__ dzg, float* __restrict__ datt,
                                                      long long Mp, int fac, int F) {
  const int f4 = F >> 2;
  const long long total = Mp * f4;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const long long mp = i / f4;
    const int c = (int)(i - mp * f4) * 4;
    const float4 a = ld4(att + mp * F + c);
    float4 acc = zero4();
    for (int j = 0; j < fac; ++j) {
      const long long m = mp * fac + j;
      const float4 d = ld4(dy + m * F + c), z = ld4(zg + m * F + c);
      const float4 sg = make_float4(sigmoid_exact(z.x), sigmoid_exact(z.y), sigmoid_exact(z.z), sigmoid_exact(z.w));
      st4(dzg + m * F + c, make_float4(d.x * a.x * sg.x * (1.f - sg.x), d.y * a.y * sg.y * (1.f - sg.y),
                                       d.z * a.z * sg.z * (1.f - sg.z), d.w * a.w * sg.w * (1.f - sg.w)));
      acc.x = fmaf(d.x, sg.x, acc.x); acc.y = fmaf(d.y, sg.y, acc.y); acc.z = fmaf(d.z, sg.z, acc.z); acc.w = fmaf(d.w, sg.w, acc.w);
    }
    st4(datt + mp * F + c, acc);
  }
}
}  // namespace
int launch_gate_fwd(const float* x, const float* zg, const float* att, float* y, int n, int T, int Tp, int F, hipStream_t s) {
  if (n <= 0) return SEPR_OK;
  if (!x || !zg || !att || !y || T <= 0 || Tp <= 0 || T % Tp || F % 4) return SEPR_EINVAL;
  const long long Mp = (long long)n * Tp;
  hipLaunchKernelGGL(gate_fwd_kernel, dim3(grid_for(Mp * (F >> 2), TPB, 1 << 16)), dim3(TPB), 0, s, x, zg, att, y, Mp, T / Tp, F);
  SEPR_CHECK_LAUNCH("gate_fwd_kernel");
  return SEPR_OK;
}
int launch_gate_bwd(const float* dy, const float* zg, const float* att, float* dzg, float* datt, int n, int T, int Tp, int F,
                    hipStream_t s) {
  if (n <= 0) return SEPR_OK;
  if (!dy || !zg || !att || !dzg || !datt || T <= 0 || Tp <= 0 || T % Tp || F % 4) return SEPR_EINVAL;
  const long long Mp = (long long)n * Tp;
  hipLaunchKernelGGL(gate_bwd_kernel, dim3(grid_for(Mp * (F >> 2), TPB, 1 << 16)), dim3(TPB), 0, s, dy, zg, att, dzg, datt, Mp, T / Tp, F);
  SEPR_CHECK_LAUNCH("gate_bwd_kernel");
  return SEPR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Attention across the S speakers of a frame, backward (network.py:241-247 with :99-124, pos_k = None).
// One thread per (frame, head): S <= 4 speakers, dk <= 32.  QKV rows are (b*S + s)*T + t, [q | k | v] of 3F.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
// FWD = true: writes o (into dQKV's place: [rows][F]) instead of the gradients; dropout (thr > 0) applies to the
// probabilities that multiply V: generator index = offset + ((frame * H + h) * S + s) * S + u
template <int S, int DK, bool FWD>
__global__ __launch_bounds__(TPB) void spkmix_bwd_kernel(const float* __restrict__ QKV, const float* __restrict__ dO,
                                                        float* __restrict__ dQKV, long long frames, int T, int F, int H, float isd,
                                                        unsigned int thr, float dscale, unsigned long long seed, unsigned long long offset,
                                                        const unsigned long long* __restrict__ salt) {
  seed = sepr_salted(seed, salt);
  const long long total = frames * H;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const long long fr = i / H;
    const int h = (int)(i - fr * H);
    const long long b = fr / T;
    const int t = (int)(fr - b * T);
    float q[S][DK], k[S][DK], v[S][DK], go[S][DK];
    long long row[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      row[s] = (b * S + s) * T + t;
      const float* p = QKV + row[s] * 3 * F + h * DK;
#pragma unroll
      for (int d = 0; d < DK; d += 4) {
        const float4 a = ld4(p + d), bb = ld4(p + F + d), c = ld4(p + 2 * F + d);
        const float4 g = FWD ? zero4() : ld4(dO + row[s] * F + h * DK + d);
        q[s][d] = a.x; q[s][d + 1] = a.y; q[s][d + 2] = a.z; q[s][d + 3] = a.w;
        k[s][d] = bb.x; k[s][d + 1] = bb.y; k[s][d + 2] = bb.z; k[s][d + 3] = bb.w;
        v[s][d] = c.x; v[s][d + 1] = c.y; v[s][d + 2] = c.z; v[s][d + 3] = c.w;
        go[s][d] = g.x; go[s][d + 1] = g.y; go[s][d + 2] = g.z; go[s][d + 3] = g.w;
      }
    }
    float P[S][S], dS_[S][S], Mk[S][S];     // Mk: dropout multiplier (0 or 1 / (1 - p)) of P[s][u]
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int u = 0; u < S; ++u)
        Mk[s][u] = (thr == 0u || sepr_keep(seed, offset + (unsigned long long)((i * S + s) * S + u), thr)) ? dscale : 0.f;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      float sc[S], mx = -3.0e38f;
#pragma unroll
      for (int u = 0; u < S; ++u) {
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < DK; ++d) a = fmaf(q[s][d], k[u][d], a);
        sc[u] = a * isd;
        mx = fmaxf(mx, sc[u]);
      }
      float den = 0.f;
#pragma unroll
      for (int u = 0; u < S; ++u) { sc[u] = expf(sc[u] - mx); den += sc[u]; }
      float dP[S], dot = 0.f;
#pragma unroll
      for (int u = 0; u < S; ++u) {
        P[s][u] = sc[u] / den;
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < DK; ++d) a = fmaf(go[s][d], v[u][d], a);
        dP[u] = a * Mk[s][u];
        dot = fmaf(dP[u], P[s][u], dot);
      }
#pragma unroll
      for (int u = 0; u < S; ++u) dS_[s][u] = P[s][u] * (dP[u] - dot) * isd;
    }
    if (FWD) {
#pragma unroll
      for (int s = 0; s < S; ++s) {
        float* o = dQKV + row[s] * F + h * DK;
#pragma unroll
        for (int d = 0; d < DK; d += 4) {
          float a[4] = {0, 0, 0, 0};
#pragma unroll
          for (int u = 0; u < S; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = fmaf(P[s][u] * Mk[s][u], v[u][d + j], a[j]);
          st4(o + d, make_float4(a[0], a[1], a[2], a[3]));
        }
      }
      continue;
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
      float* o = dQKV + row[s] * 3 * F + h * DK;
#pragma unroll
      for (int d = 0; d < DK; d += 4) {
        float dq[4] = {0, 0, 0, 0}, dk_[4] = {0, 0, 0, 0}, dv[4] = {0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < S; ++u)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            dq[j] = fmaf(dS_[s][u], k[u][d + j], dq[j]);       // dq_s = sum_u dS[s][u] k_u
            dk_[j] = fmaf(dS_[u][s], q[u][d + j], dk_[j]);     // dk_s = sum_u dS[u][s] q_u
            dv[j] = fmaf(P[u][s] * Mk[u][s], go[u][d + j], dv[j]);   // dv_s = sum_u P_dropped[u][s] dO_u
          }
        st4(o + d, make_float4(dq[0], dq[1], dq[2], dq[3]));
        st4(o + F + d, make_float4(dk_[0], dk_[1], dk_[2], dk_[3]));
        st4(o + 2 * F + d, make_float4(dv[0], dv[1], dv[2], dv[3]));
      }
    }
  }
}
}  // namespace
static int spkmix_train_launch(bool fwd, const float* QKV, const float* dO, float* out, int B, int S, int T, int F, int H, float p,
                               unsigned long long seed, unsigned long long offset, hipStream_t s) {
  if (B <= 0 || T <= 0) return SEPR_OK;
  if (!QKV || !out || (!fwd && !dO) || H <= 0 || F % H || !(p >= 0.f) || !(p < 1.f)) return SEPR_EINVAL;
  const int dk = F / H;
  const long long frames = (long long)B * T;
  const float isd = 1.0f / sqrtf((float)dk);
  const unsigned int thr = p > 0.f ? sepr_drop_threshold(p) : 0u;
  const float dscale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  const dim3 g(grid_for(frames * H, TPB, 1 << 16)), t(TPB);
#define SEPR_SPKB(SS, DD)                                                                                                               \
  do {                                                                                                                                  \
    if (fwd) hipLaunchKernelGGL((spkmix_bwd_kernel<SS, DD, true>), g, t, 0, s, QKV, dO, out, frames, T, F, H, isd, thr, dscale, seed, offset, drop_salt()); \
    else hipLaunchKernelGGL((spkmix_bwd_kernel<SS, DD, false>), g, t, 0, s, QKV, dO, out, frames, T, F, H, isd, thr, dscale, seed, offset, drop_salt());    \
  } while (0)
  if (S == 2 && dk == 16) SEPR_SPKB(2, 16);
  else if (S == 2 && dk == 32) SEPR_SPKB(2, 32);
  else if (S == 3 && dk == 16) SEPR_SPKB(3, 16);
  else if (S == 3 && dk == 32) SEPR_SPKB(3, 32);
  else return SEPR_EINVAL;
#undef SEPR_SPKB
  SEPR_CHECK_LAUNCH("spkmix train kernel");
  return SEPR_OK;
}
int launch_spkmix_train_fwd(const float* QKV, float* O, int B, int S, int T, int F, int H, float p, unsigned long long seed,
                            unsigned long long offset, hipStream_t s) {
  return spkmix_train_launch(true, QKV, nullptr, O, B, S, T, F, H, p, seed, offset, s);
}
int launch_spkmix_bwd(const float* QKV, const float* dO, float* dQKV, int B, int S, int T, int F, int H, float p, unsigned long long seed,
                      unsigned long long offset, hipStream_t s) {
  return spkmix_train_launch(false, QKV, dO, dQKV, B, S, T, F, H, p, seed, offset, s);
}

// ---------------------------------------------------------------------------------------------------------------------
// Fusion conv backward glue (module.py:212-214): dcat [n,T,2F] -> dlo [n,T/2,F] = dcat[2t',:F] + dcat[2t'+1,:F], dskip = dcat[:,F:]
// ---------------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(TPB) void unfuse_kernel(const float* __restrict__ dcat, float* __restrict__ dlo, float* __restrict__ dskip,
                                                    long long Mh, int F) {
  const int f4 = F >> 2;
  const long long total = Mh * f4;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const long long mh = i / f4;
    const int c = (int)(i - mh * f4) * 4;
    const float* p0 = dcat + (2 * mh) * 2 * F;
    const float* p1 = p0 + 2 * F;
    const float4 a = ld4(p0 + c), b = ld4(p1 + c);
    st4(dlo + mh * F + c, make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w));
    st4(dskip + (2 * mh) * F + c, ld4(p0 + F + c));
    st4(dskip + (2 * mh + 1) * F + c, ld4(p1 + F + c));
  }
}
}  // namespace
int launch_unfuse(const float* dcat, float* dlo, float* dskip, int n, int T, int F, hipStream_t s) {
  if (n <= 0) return SEPR_OK;
  if (!dcat || !dlo || !dskip || T <= 0 || (T & 1) || F % 4) return SEPR_EINVAL;
  const long long Mh = (long long)n * (T / 2);
  hipLaunchKernelGGL(unfuse_kernel, dim3(grid_for(Mh * (F >> 2), TPB, 1 << 16)), dim3(TPB), 0, s, dcat, dlo, dskip, Mh, F);
  SEPR_CHECK_LAUNCH("unfuse_kernel");
  return SEPR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// GroupNorm(1 group) over (T, F) of a sequence (module.py:28,117): out-of-place apply and backward.
//   y = g_f vh + b_f, vh = (v - mean) rstd;   dv = rstd (g dy - mean_all(g dy) - vh mean_all(g dy vh));
//   dg_f += sum_{seq,t} dy vh,  db_f += sum dy
// Pass 1: per (sequence, 256-frame chunk) partials: 2 scalars (fp64) and per-channel (dy vh, dy) sums.
// Pass 2: combine per sequence, apply.  Pass 3: per-channel reduce into the parameter gradients.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int GB_TC = 256;

__global__ __launch_bounds__(TPB) void gn_apply_oop_kernel(const float* __restrict__ v, const float* __restrict__ stats,
                                                          const float* __restrict__ g, const float* __restrict__ b, float* __restrict__ y,
                                                          long long per_seq, int F, long long total4) {
  const int f4 = F >> 2;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total4; i += (long long)gridDim.x * TPB) {
    const long long e = i * 4;
    const long long seq = e / per_seq;
    const int c = (int)(i % f4) * 4;
    const float mean = stats[2 * seq], rstd = stats[2 * seq + 1];
    const float4 x = ld4(v + e), ga = ld4(g + c), be = ld4(b + c);
    st4(y + e, make_float4(fmaf((x.x - mean) * rstd, ga.x, be.x), fmaf((x.y - mean) * rstd, ga.y, be.y),
                           fmaf((x.z - mean) * rstd, ga.z, be.z), fmaf((x.w - mean) * rstd, ga.w, be.w)));
  }
}

// block = (chunk, seq); thread = channel quad q4 (F/4 <= 128 quads -> up to 2 row lanes of 128; generic: rl lanes)
__global__ __launch_bounds__(TPB) void gn_bwd_partial_kernel(const float* __restrict__ dy, const float* __restrict__ v,
                                                            const float* __restrict__ stats, const float* __restrict__ g, int T, int F,
                                                            int nchunk, double* __restrict__ sc_part, float* __restrict__ ch_part) {
  __shared__ double sh[TPB][2];
  __shared__ float shc[TPB][8];
  const int seq = blockIdx.y, chunk = blockIdx.x;
  const int f4 = F >> 2;
  const int lanes = TPB / f4 > 0 ? TPB / f4 : 1;       // row lanes per block (F = 128 -> 8, 256 -> 4, 64 -> 16)
  const int q = threadIdx.x % f4, rl = threadIdx.x / f4;
  const int t0 = chunk * GB_TC, t1 = min(T, t0 + GB_TC);
  const float mean = stats[2 * seq], rstd = stats[2 * seq + 1];
  double s1 = 0.0, s2 = 0.0;
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (rl < lanes) {
    const float4 ga = ld4(g + 4 * q);
    for (int t = t0 + rl; t < t1; t += lanes) {
      const long long e = ((long long)seq * T + t) * F + 4 * q;
      const float4 d = ld4(dy + e), x = ld4(v + e);
      const float vh[4] = {(x.x - mean) * rstd, (x.y - mean) * rstd, (x.z - mean) * rstd, (x.w - mean) * rstd};
      const float dd[4] = {d.x, d.y, d.z, d.w}, gg[4] = {ga.x, ga.y, ga.z, ga.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gd = gg[j] * dd[j];
        s1 += (double)gd;
        s2 = fma((double)gd, (double)vh[j], s2);
        a[j] = fmaf(dd[j], vh[j], a[j]);
        a[4 + j] += dd[j];
      }
    }
  }
  sh[threadIdx.x][0] = s1;
  sh[threadIdx.x][1] = s2;
#pragma unroll
  for (int j = 0; j < 8; ++j) shc[threadIdx.x][j] = a[j];
  __syncthreads();
  if (threadIdx.x < 2) {
    double s = 0.0;
    for (int i = 0; i < TPB; ++i) s += sh[i][threadIdx.x];
    sc_part[((long long)seq * nchunk + chunk) * 2 + threadIdx.x] = s;
  }
  if (threadIdx.x < f4) {
    float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = 0; r < lanes; ++r)
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += shc[r * f4 + threadIdx.x][j];
    float* p = ch_part + (((long long)seq * nchunk + chunk) * f4 + threadIdx.x) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) p[j] = o[j];
  }
}
__global__ __launch_bounds__(TPB) void gn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ v,
                                                          const float* __restrict__ stats, const float* __restrict__ g,
                                                          const double* __restrict__ sc_part, int nchunk, float* __restrict__ dv, int T,
                                                          int F, int S, long long total4) {
  const int f4 = F >> 2;
  const double cnt = (double)T * (double)F;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total4; i += (long long)gridDim.x * TPB) {
    const long long row = i / f4;                 // seq * T + t
    const int c = (int)(i - row * f4) * 4;
    const long long seq = row / T;
    const int t = (int)(row - seq * T);
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < nchunk; ++k) {
      s1 += sc_part[(seq * nchunk + k) * 2];
      s2 += sc_part[(seq * nchunk + k) * 2 + 1];
    }
    const float m1 = (float)(s1 / cnt), m2 = (float)(s2 / cnt);
    const float mean = stats[2 * seq], rstd = stats[2 * seq + 1];
    const float4 d = ld4(dy + row * F + c), x = ld4(v + row * F + c), ga = ld4(g + c);
    const float4 o = make_float4(rstd * (ga.x * d.x - m1 - (x.x - mean) * rstd * m2), rstd * (ga.y * d.y - m1 - (x.y - mean) * rstd * m2),
                                 rstd * (ga.z * d.z - m1 - (x.z - mean) * rstd * m2), rstd * (ga.w * d.w - m1 - (x.w - mean) * rstd * m2));
    if (S > 0) {      // undo view(B*S, F, T): sequence b*S + s, channel f  ->  row (b, t), channel s*F + f   (module.py:123)
      const long long b = seq / S;
      const int sp = (int)(seq - b * S);
      st4(dv + ((b * T + t) * S + sp) * F + c, o);
    } else {
      st4(dv + row * F + c, o);
    }
  }
}
// 16 channels per workgroup, 16 threads per channel over the partial rows (thread j: rows j, j + 16, ...), fixed-order combine
__global__ __launch_bounds__(TPB) void gn_bwd_param_kernel(const float* __restrict__ ch_part, int nblk, int F, float* __restrict__ dg_g,
                                                          float* __restrict__ db_g) {
  __shared__ float shp[16][16][2];
  const int cl = threadIdx.x & 15, jl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const int q = c >> 2, j = c & 3;
  const int f4 = F >> 2;
  float s1 = 0.f, s2 = 0.f;
  if (c < F)
    for (int bk = jl; bk < nblk; bk += 16) {
      s1 += ch_part[((long long)bk * f4 + q) * 8 + j];
      s2 += ch_part[((long long)bk * f4 + q) * 8 + 4 + j];
    }
  shp[jl][cl][0] = s1;
  shp[jl][cl][1] = s2;
  __syncthreads();
  if (jl != 0 || c >= F) return;
  s1 = 0.f;
  s2 = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    s1 += shp[r][cl][0];
    s2 += shp[r][cl][1];
  }
  dg_g[c] += s1;
  db_g[c] += s2;
}
}  // namespace

size_t gn_bwd_ws(int n, int T, int F) {
  const int nchunk = (T + GB_TC - 1) / GB_TC;
  return align_up((size_t)n * nchunk * 2 * sizeof(double)) + align_up((size_t)n * nchunk * (F / 4) * 8 * sizeof(float));
}
int launch_gn_apply_oop(const float* v, const float* stats, const float* g, const float* b, float* y, int n, int T, int F,
                        hipStream_t s) {
  if (n <= 0) return SEPR_OK;
  if (!v || !stats || !g || !b || !y || F % 4) return SEPR_EINVAL;
  const long long per_seq = (long long)T * F, total4 = (long long)n * per_seq / 4;
  hipLaunchKernelGGL(gn_apply_oop_kernel, dim3(grid_for(total4, TPB, 1 << 16)), dim3(TPB), 0, s, v, stats, g, b, y, per_seq, F, total4);
  SEPR_CHECK_LAUNCH("gn_apply_oop_kernel");
  return SEPR_OK;
}
int launch_gn_bwd(const float* dy, const float* v, const float* stats, const float* g, float* dv, float* dg_g, float* db_g, int n, int T,
                  int F, int S, void* ws, size_t ws_bytes, hipStream_t s) {
  if (n <= 0) return SEPR_OK;
  if (!dy || !v || !stats || !g || !dv || !dg_g || !db_g || F % 4 || F > 4 * TPB || F < 4) return SEPR_EINVAL;
  if (S > 0 && n % S) return SEPR_EINVAL;
  if (!ws || ws_bytes < gn_bwd_ws(n, T, F)) return SEPR_EWORKSPACE;
  const int nchunk = (T + GB_TC - 1) / GB_TC;
  double* sc_part = static_cast<double*>(ws);
  float* ch_part = reinterpret_cast<float*>(static_cast<char*>(ws) + align_up((size_t)n * nchunk * 2 * sizeof(double)));
  hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(nchunk, n), dim3(TPB), 0, s, dy, v, stats, g, T, F, nchunk, sc_part, ch_part);
  const long long total4 = (long long)n * T * F / 4;
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(grid_for(total4, TPB, 1 << 16)), dim3(TPB), 0, s, dy, v, stats, g, sc_part, nchunk, dv, T, F,
                     S, total4);
  hipLaunchKernelGGL(gn_bwd_param_kernel, dim3((F + 15) / 16), dim3(TPB), 0, s, ch_part, n * nchunk, F, dg_g, db_g);
  SEPR_CHECK_LAUNCH("gn_bwd kernels");
  return SEPR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// DownConv (module.py:67-76) in train mode: the depthwise stride-2 conv with its bias on its own (BatchNorm needs batch
// statistics of exactly this tensor), and its backward.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(TPB) void downconv_pre_kernel(const float* __restrict__ x, float* __restrict__ c, int T, int To, int F, int K,
                                                          const float* __restrict__ w /*[K][F]*/, const float* __restrict__ b,
                                                          long long total4) {
  const int f4 = F >> 2, pad = (K - 1) / 2;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total4; i += (long long)gridDim.x * TPB) {
    const long long row = i / f4;                  // seq * To + to
    const int ch = (int)(i - row * f4) * 4;
    const long long seq = row / To;
    const int to = (int)(row - seq * To);
    float4 acc = ld4(b + ch);
    for (int k = 0; k < K; ++k) {
      const int t = 2 * to + k - pad;
      if (t >= 0 && t < T) {
        const float4 xv = ld4(x + ((long long)seq * T + t) * F + ch), wv = ld4(w + k * F + ch);
        acc.x = fmaf(wv.x, xv.x, acc.x); acc.y = fmaf(wv.y, xv.y, acc.y); acc.z = fmaf(wv.z, xv.z, acc.z); acc.w = fmaf(wv.w, xv.w, acc.w);
      }
    }
    st4(c + row * F + ch, acc);
  }
}
// dx[t] = sum_k w[k] dc[(t + pad - k) / 2] over k with (t + pad - k) even and the frame in range
__global__ __launch_bounds__(TPB) void downconv_bwd_dx_kernel(const float* __restrict__ dc, float* __restrict__ dx, int T, int To, int F,
                                                             int K, const float* __restrict__ w, long long total4) {
  const int f4 = F >> 2, pad = (K - 1) / 2;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total4; i += (long long)gridDim.x * TPB) {
    const long long row = i / f4;                  // seq * T + t
    const int ch = (int)(i - row * f4) * 4;
    const long long seq = row / T;
    const int t = (int)(row - seq * T);
    float4 acc = zero4();
    for (int k = 0; k < K; ++k) {
      const int u = t + pad - k;
      if (u >= 0 && !(u & 1) && (u >> 1) < To) {
        const float4 d = ld4(dc + ((long long)seq * To + (u >> 1)) * F + ch), wv = ld4(w + k * F + ch);
        acc.x = fmaf(wv.x, d.x, acc.x); acc.y = fmaf(wv.y, d.y, acc.y); acc.z = fmaf(wv.z, d.z, acc.z); acc.w = fmaf(wv.w, d.w, acc.w);
      }
    }
    st4(dx + row * F + ch, acc);
  }
}
// weight / bias partials: block = DC_TC output frames of one sequence, thread = channel (looping), slots k = 0..K-1, 16 = bias
constexpr int DC_TC = 32;
__global__ __launch_bounds__(TPB) void downconv_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dc, int T, int To,
                                                            int F, int K, int nchunk, float* __restrict__ part) {
  const int seq = blockIdx.x / nchunk, chunk = blockIdx.x - seq * nchunk;
  const int pad = (K - 1) / 2;
  const int o0 = chunk * DC_TC, o1 = min(To, o0 + DC_TC);
  for (int ch = threadIdx.x; ch < F; ch += blockDim.x) {
    float acc[17];
#pragma unroll
    for (int k = 0; k < 17; ++k) acc[k] = 0.f;
    for (int to = o0; to < o1; ++to) {
      const float d = dc[((long long)seq * To + to) * F + ch];
      acc[16] += d;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        if (k < K) {
          const int t = 2 * to + k - pad;
          if (t >= 0 && t < T) acc[k] = fmaf(d, x[((long long)seq * T + t) * F + ch], acc[k]);
        }
      }
    }
    float* p = part + ((long long)blockIdx.x * F + ch) * 17;
#pragma unroll
    for (int k = 0; k < 17; ++k) p[k] = acc[k];
  }
}
__global__ __launch_bounds__(TPB) void downconv_wgrad_reduce_kernel(const float* __restrict__ part, int nblk, int F, int K,
                                                                   float* __restrict__ dw_g, float* __restrict__ db_g) {
  const int i = blockIdx.x * TPB + threadIdx.x;
  if (i >= F * 17) return;
  const int ch = i / 17, k = i - ch * 17;
  if (k >= K && k != 16) return;
  float s = 0.f;
  for (int bk = 0; bk < nblk; ++bk) s += part[((long long)bk * F + ch) * 17 + k];
  if (k == 16) db_g[ch] += s;
  else dw_g[ch * K + k] += s;
}
}  // namespace

size_t downconv_bwd_ws(int n, int T, int F, int K) {
  const int pad = (K - 1) / 2;
  const int To = (T + 2 * pad - K) / 2 + 1;
  const int nchunk = (To + DC_TC - 1) / DC_TC;
  return align_up((size_t)n * nchunk * F * 17 * sizeof(float)) + align_up((size_t)PR_GROUPS * F * 17 * sizeof(float));
}
int launch_downconv_pre(const float* x, float* c, int n, int T, int To, int F, int K, const float* w, const float* b, hipStream_t s) {
  if (n <= 0) return SEPR_OK;
  if (!x || !c || !w || !b || F % 4 || K <= 0) return SEPR_EINVAL;
  const long long total4 = (long long)n * To * F / 4;
  hipLaunchKernelGGL(downconv_pre_kernel, dim3(grid_for(total4, TPB, 1 << 16)), dim3(TPB), 0, s, x, c, T, To, F, K, w, b, total4);
  SEPR_CHECK_LAUNCH("downconv_pre_kernel");
  return SEPR_OK;
}
int launch_downconv_bwd(const float* x, const float* dc, float* dx, int n, int T, int To, int F, int K, const float* w, float* dw_g,
                        float* db_g, void* ws, size_t ws_bytes, hipStream_t s) {
  if (n <= 0) return SEPR_OK;
  if (!x || !dc || !dx || !w || !dw_g || !db_g || F % 4 || K <= 0 || K > 16) return SEPR_EINVAL;
  if (!ws || ws_bytes < downconv_bwd_ws(n, T, F, K)) return SEPR_EWORKSPACE;
  const long long total4 = (long long)n * T * F / 4;
  hipLaunchKernelGGL(downconv_bwd_dx_kernel, dim3(grid_for(total4, TPB, 1 << 16)), dim3(TPB), 0, s, dc, dx, T, To, F, K, w, total4);
  const int nchunk = (To + DC_TC - 1) / DC_TC;
  float* part = static_cast<float*>(ws);
  hipLaunchKernelGGL(downconv_wgrad_kernel, dim3(n * nchunk), dim3(F < TPB ? F : TPB), 0, s, x, dc, T, To, F, K, nchunk, part);
  float* scratch = reinterpret_cast<float*>(static_cast<char*>(ws) + align_up((size_t)n * nchunk * F * 17 * sizeof(float)));
  const float* rows = nullptr;
  const int nrows = prereduce(part, n * nchunk, F * 17, scratch, &rows, s);
  hipLaunchKernelGGL(downconv_wgrad_reduce_kernel, dim3((F * 17 + TPB - 1) / TPB), dim3(TPB), 0, s, rows, nrows, F, K, dw_g, db_g);
  SEPR_CHECK_LAUNCH("downconv_bwd kernels");
  return SEPR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// small element-wise helpers
// ---------------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(TPB) void add_inplace_kernel(float* __restrict__ y, const float* __restrict__ a, long long n4) {
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n4; i += (long long)gridDim.x * TPB) {
    const float4 u = ld4(y + 4 * i), v = ld4(a + 4 * i);
    st4(y + 4 * i, make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w));
  }
}
// y = (x ? x : 0) + ls[f] * drop(v): the residual + LayerScale tail of a block whose output dropout is live (thr > 0)
__global__ __launch_bounds__(TPB) void res_ls_kernel(const float* __restrict__ x, const float* __restrict__ v, const float* __restrict__ ls,
                                                    float* __restrict__ y, long long M, int F, unsigned int thr, float dscale,
                                                    unsigned long long seed, unsigned long long offset,
                                                    const unsigned long long* __restrict__ salt) {
  seed = sepr_salted(seed, salt);
  const int f4 = F >> 2;
  const long long total = M * f4;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const int c = (int)(i % f4) * 4;
    const float4 a = x ? ld4(x + 4 * i) : zero4();
    float4 b = ld4(v + 4 * i);
    const float4 l = ld4(ls + c);
    if (thr) {
      const unsigned long long e = offset + 4ull * (unsigned long long)i;
      b.x = sepr_keep(seed, e, thr) ? b.x * dscale : 0.f;
      b.y = sepr_keep(seed, e + 1, thr) ? b.y * dscale : 0.f;
      b.z = sepr_keep(seed, e + 2, thr) ? b.z * dscale : 0.f;
      b.w = sepr_keep(seed, e + 3, thr) ? b.w * dscale : 0.f;
    }
    st4(y + 4 * i, make_float4(fmaf(b.x, l.x, a.x), fmaf(b.y, l.y, a.y), fmaf(b.z, l.z, a.z), fmaf(b.w, l.w, a.w)));
  }
}
__global__ __launch_bounds__(TPB) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long long count, unsigned int thr,
                                                     float scale, unsigned long long seed, unsigned long long offset,
                                                     const unsigned long long* __restrict__ salt) {
  seed = sepr_salted(seed, salt);
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < count; i += (long long)gridDim.x * TPB)
    y[i] = sepr_keep(seed, offset + (unsigned long long)i, thr) ? x[i] * scale : 0.f;
}
// the 16-bit generator of the fused epilogues (EPI_RESDROP, sepr_gemm_epi.h) applied to a [M][F] tensor: element (m, c) keeps iff the
// 16-bit half c & 1 of sepr_drop_word(key, m, c >> 1) >= thr; what the backward of such a block applies to dy
__global__ __launch_bounds__(TPB) void dropout16_kernel(const float* __restrict__ x, float* __restrict__ y, long long M, int F, unsigned thr,
                                                       float scale, unsigned long long seed, const unsigned long long* __restrict__ salt,
                                                       unsigned site) {
  const DropKey key = sepr_drop_key(seed, salt, site);
  const int f4 = F >> 2;
  const long long total = M * f4;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const long long m = i / f4;
    const int c = (int)(i - m * f4) * 4;
    float4 v = ld4(x + 4 * i);
    const unsigned d0 = sepr_drop_word(key, (unsigned)m, (unsigned)(c >> 1)), d1 = sepr_drop_word(key, (unsigned)m, (unsigned)(c >> 1) + 1u);
    v.x = (d0 & 0xffffu) >= thr ? v.x * scale : 0.f;
    v.y = (d0 >> 16) >= thr ? v.y * scale : 0.f;
    v.z = (d1 & 0xffffu) >= thr ? v.z * scale : 0.f;
    v.w = (d1 >> 16) >= thr ? v.w * scale : 0.f;
    st4(y + 4 * i, v);
  }
}
}  // namespace
int launch_dropout16(const float* x, float* y, long long M, int F, float p, unsigned long long seed, unsigned site, hipStream_t s) {
  if (M <= 0) return SEPR_OK;
  if (!x || !y || F % 4 || !(p > 0.f) || !(p < 1.f) || M > 0x7fffffffLL) return SEPR_EINVAL;
  hipLaunchKernelGGL(dropout16_kernel, dim3(grid_for(M * (F >> 2), TPB, 1 << 16)), dim3(TPB), 0, s, x, y, M, F, sepr_drop_thr16(p),
                     sepr_drop_scale16(p), seed, drop_salt(), site);
  SEPR_CHECK_LAUNCH("dropout16_kernel");
  return SEPR_OK;
}
int launch_add_inplace(float* y, const float* a, long long count, hipStream_t s) {
  if (count <= 0) return SEPR_OK;
  if (!y || !a || count % 4) return SEPR_EINVAL;
  hipLaunchKernelGGL(add_inplace_kernel, dim3(grid_for(count / 4, TPB, 1 << 16)), dim3(TPB), 0, s, y, a, count / 4);
  SEPR_CHECK_LAUNCH("add_inplace_kernel");
  return SEPR_OK;
}
int launch_res_ls(const float* x, const float* v, const float* ls, float* y, long long M, int F, float p, unsigned long long seed,
                  unsigned long long offset, hipStream_t s) {
  if (M <= 0) return SEPR_OK;
  if (!v || !ls || !y || F % 4 || !(p >= 0.f) || !(p < 1.f)) return SEPR_EINVAL;
  hipLaunchKernelGGL(res_ls_kernel, dim3(grid_for(M * (F >> 2), TPB, 1 << 16)), dim3(TPB), 0, s, x, v, ls, y, M, F,
                     p > 0.f ? sepr_drop_threshold(p) : 0u, p > 0.f ? 1.0f / (1.0f - p) : 1.0f, seed, offset, drop_salt());
  SEPR_CHECK_LAUNCH("res_ls_kernel");
  return SEPR_OK;
}
int launch_dropout(const float* x, float* y, long long count, float p, unsigned long long seed, unsigned long long offset,
                   hipStream_t s) {
  if (count <= 0) return SEPR_OK;
  if (!x || !y || !(p >= 0.f) || !(p < 1.f)) return SEPR_EINVAL;
  hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(count, TPB, 1 << 16)), dim3(TPB), 0, s, x, y, count, sepr_drop_threshold(p),
                     1.0f / (1.0f - p), seed, offset, drop_salt());
  SEPR_CHECK_LAUNCH("dropout_kernel");
  return SEPR_OK;
}
// ---------------------------------------------------------------------------------------------------------------------
// waveform ends: ConvTranspose1d decoder (module.py:268-283), masked auxiliary heads (module.py:257-260, network.py:41),
// Conv1d + GELU encoder (module.py:12-23)
// ---------------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(TPB) void permute_sb_kernel(const float* __restrict__ dwav, float* __restrict__ dwp, int S, int B, int Tout,
                                                        long long total) {
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
    const long long seq = i / Tout;
    const int t = (int)(i - seq * Tout);
    const int b = (int)(seq / S), sp = (int)(seq - (long long)b * S);
    dwp[i] = dwav[((long long)sp * B + b) * Tout + t];
  }
}
__global__ __launch_bounds__(TPB) void dec_bwd_dm_kernel(const float* __restrict__ dwp, const float* __restrict__ wdec, float* __restrict__ dm,
                                                        int L, int N, int K, int stride, int Tout, long long total4) {
  const int n4 = N >> 2;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total4; i += (long long)gridDim.x * TPB) {
    const long long row = i / n4;
    const int c = (int)(i - row * n4) * 4;
    const long long seq = row / L;
    const int l = (int)(row - seq * L);
    const float* f = dwp + seq * Tout + (long long)stride * l;
    float4 acc = zero4();
    for (int k = 0; k < K; ++k) {
      const float d = f[k];
      const float4 wv = ld4(wdec + k * N + c);
      acc.x = fmaf(d, wv.x, acc.x); acc.y = fmaf(d, wv.y, acc.y); acc.z = fmaf(d, wv.z, acc.z); acc.w = fmaf(d, wv.w, acc.w);
    }
    st4(dm + row * N + c, acc);
  }
}
// Same sum with the K taps of a thread's four columns held in registers: a thread keeps its column group and walks rows, so the
// weights are read once per thread instead of once per output (the loop above re-reads 16 float4 of weights per float4 of output)
template <int KK>
__global__ __launch_bounds__(TPB) void dec_bwd_dm_fixed_kernel(const float* __restrict__ dwp, const float* __restrict__ wdec,
                                                              float* __restrict__ dm, int L, int N, int stride, int Tout, long long rows) {
  const int n4 = N >> 2, rpb = TPB / n4;
  const int c = (threadIdx.x % n4) * 4, rl = threadIdx.x / n4;
  float4 w[KK];
#pragma unroll
  for (int k = 0; k < KK; ++k) w[k] = ld4(wdec + k * N + c);
  for (long long row = (long long)blockIdx.x * rpb + rl; row < rows; row += (long long)gridDim.x * rpb) {
    const long long seq = row / L;
    const int l = (int)(row - seq * L);
    const float* f = dwp + seq * Tout + (long long)stride * l;
    float4 acc = zero4();
#pragma unroll
    for (int k = 0; k < KK; ++k) {
      const float d = f[k];
      acc.x = fmaf(d, w[k].x, acc.x); acc.y = fmaf(d, w[k].y, acc.y); acc.z = fmaf(d, w[k].z, acc.z); acc.w = fmaf(d, w[k].w, acc.w);
    }
    st4(dm + row * N + c, acc);
  }
}
__global__ __launch_bounds__(TPB) void aux_m_kernel(const float* __restrict__ o2, const float* __restrict__ enc, const int* __restrict__ idx,
                                                   float* __restrict__ m, int S, int Tsrc, int L, int N, long long total4) {
  const int n4 = N >> 2;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total4; i += (long long)gridDim.x * TPB) {
    const long long row = i / n4;
    const int c = (int)(i - row * n4) * 4;
    const long long seq = row / L;
    const int l = (int)(row - seq * L);
    const float4 o = ld4(o2 + (seq * Tsrc + idx[l]) * N + c), e = ld4(enc + ((seq / S) * L + l) * N + c);
    st4(m + row * N + c, make_float4(fmaxf(o.x, 0.f) * e.x, fmaxf(o.y, 0.f) * e.y, fmaxf(o.z, 0.f) * e.z, fmaxf(o.w, 0.f) * e.w));
  }
}
__global__ __launch_bounds__(TPB) void aux_mask_bwd_kernel(const float* __restrict__ dm, const float* __restrict__ o2,
                                                          const float* __restrict__ enc, const int* __restrict__ start, float* __restrict__ do2,
                                                          int S, int Tsrc, int L, int N, long long total4) {
  const int n4 = N >> 2;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total4; i += (long long)gridDim.x * TPB) {
    const long long row = i / n4;                  // seq * Tsrc + src
    const int c = (int)(i - row * n4) * 4;
    const long long seq = row / Tsrc;
    const int src = (int)(row - seq * Tsrc);
    float4 acc = zero4();
    for (int l = start[src]; l < start[src + 1]; ++l) {
      const float4 d = ld4(dm + (seq * L + l) * N + c), e = ld4(enc + ((seq / S) * L + l) * N + c);
      acc.x = fmaf(d.x, e.x, acc.x); acc.y = fmaf(d.y, e.y, acc.y); acc.z = fmaf(d.z, e.z, acc.z); acc.w = fmaf(d.w, e.w, acc.w);
    }
    const float4 o = ld4(o2 + row * N + c);
    st4(do2 + row * N + c, make_float4(o.x > 0.f ? acc.x : 0.f, o.y > 0.f ? acc.y : 0.f, o.z > 0.f ? acc.z : 0.f, o.w > 0.f ? acc.w : 0.f));
  }
}
__global__ __launch_bounds__(TPB) void aux_denc_kernel(const float* __restrict__ dm, const float* __restrict__ o2, const int* __restrict__ idx,
                                                      float* __restrict__ denc, int S, int Tsrc, int L, int N, long long total4) {
  const int n4 = N >> 2;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total4; i += (long long)gridDim.x * TPB) {
    const long long row = i / n4;                  // b * L + l
    const int c = (int)(i - row * n4) * 4;
    const long long b = row / L;
    const int l = (int)(row - b * L);
    float4 acc = ld4(denc + row * N + c);
    const int src = idx[l];
    for (int sp = 0; sp < S; ++sp) {
      const long long seq = b * S + sp;
      const float4 d = ld4(dm + (seq * L + l) * N + c), o = ld4(o2 + (seq * Tsrc + src) * N + c);
      acc.x = fmaf(d.x, fmaxf(o.x, 0.f), acc.x); acc.y = fmaf(d.y, fmaxf(o.y, 0.f), acc.y);
      acc.z = fmaf(d.z, fmaxf(o.z, 0.f), acc.z); acc.w = fmaf(d.w, fmaxf(o.w, 0.f), acc.w);
    }
    st4(denc + row * N + c, acc);
  }
}
__global__ __launch_bounds__(TPB) void enc_bwd_pre_kernel(float* __restrict__ de, const float* __restrict__ add, const float* __restrict__ wav,
                                                         const float* __restrict__ w, int T, int L, int N, int K, int stride,
                                                         long long total4) {
  const int n4 = N >> 2;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total4; i += (long long)gridDim.x * TPB) {
    const long long row = i / n4;
    const int c = (int)(i - row * n4) * 4;
    const long long b = row / L;
    const int l = (int)(row - b * L);
    const float* f = wav + b * T + (long long)stride * l;
    float4 pre = zero4();
    for (int k = 0; k < K; ++k) {
      const float x = f[k];
      const float4 wv = ld4(w + k * N + c);
      pre.x = fmaf(x, wv.x, pre.x); pre.y = fmaf(x, wv.y, pre.y); pre.z = fmaf(x, wv.z, pre.z); pre.w = fmaf(x, wv.w, pre.w);
    }
    float4 d = ld4(de + row * N + c);
    if (add) {
      const float4 a = ld4(add + row * N + c);
      d.x += a.x; d.y += a.y; d.z += a.z; d.w += a.w;
    }
    st4(de + row * N + c, make_float4(d.x * gelu_grad(pre.x), d.y * gelu_grad(pre.y), d.z * gelu_grad(pre.z), d.w * gelu_grad(pre.w)));
  }
}
}  // namespace
int launch_permute_sb(const float* dwav, float* dwp, int S, int B, int Tout, hipStream_t s) {
  if (!dwav || !dwp || S <= 0 || B <= 0 || Tout <= 0) return SEPR_EINVAL;
  const long long total = (long long)S * B * Tout;
  hipLaunchKernelGGL(permute_sb_kernel, dim3(grid_for(total, TPB, 1 << 16)), dim3(TPB), 0, s, dwav, dwp, S, B, Tout, total);
  SEPR_CHECK_LAUNCH("permute_sb_kernel");
  return SEPR_OK;
}
int launch_dec_bwd_dm(const float* dwp, const float* wdec, float* dm, int nS, int L, int N, int K, int stride, int Tout, hipStream_t s) {
  if (!dwp || !wdec || !dm || nS <= 0 || L <= 0 || N % 4 || (L - 1) * stride + K > Tout) return SEPR_EINVAL;
  const long long total4 = (long long)nS * L * N / 4;
  const int n4 = N >> 2;
  if (K == 16 && n4 <= TPB && TPB % n4 == 0) {
    const long long rows = total4 / n4;
    hipLaunchKernelGGL((dec_bwd_dm_fixed_kernel<16>), dim3(grid_for(rows, TPB / n4, 2048)), dim3(TPB), 0, s, dwp, wdec, dm, L, N, stride, Tout,
                       rows);
  } else {
    hipLaunchKernelGGL(dec_bwd_dm_kernel, dim3(grid_for(total4, TPB, 1 << 16)), dim3(TPB), 0, s, dwp, wdec, dm, L, N, K, stride, Tout, total4);
  }
  SEPR_CHECK_LAUNCH("dec_bwd_dm_kernel");
  return SEPR_OK;
}
int launch_aux_m(const float* o2, const float* enc, const int* idx, float* m, int nS, int S, int Tsrc, int L, int N, hipStream_t s) {
  if (!o2 || !enc || !idx || !m || nS <= 0 || S <= 0 || N % 4) return SEPR_EINVAL;
  const long long total4 = (long long)nS * L * N / 4;
  hipLaunchKernelGGL(aux_m_kernel, dim3(grid_for(total4, TPB, 1 << 16)), dim3(TPB), 0, s, o2, enc, idx, m, S, Tsrc, L, N, total4);
  SEPR_CHECK_LAUNCH("aux_m_kernel");
  return SEPR_OK;
}
int launch_aux_mask_bwd(const float* dm, const float* o2, const float* enc, const int* idx_start, float* do2, int nS, int S, int Tsrc, int L,
                        int N, hipStream_t s) {
  if (!dm || !o2 || !enc || !idx_start || !do2 || nS <= 0 || S <= 0 || N % 4) return SEPR_EINVAL;
  const long long total4 = (long long)nS * Tsrc * N / 4;
  hipLaunchKernelGGL(aux_mask_bwd_kernel, dim3(grid_for(total4, TPB, 1 << 16)), dim3(TPB), 0, s, dm, o2, enc, idx_start, do2, S, Tsrc, L, N,
                     total4);
  SEPR_CHECK_LAUNCH("aux_mask_bwd_kernel");
  return SEPR_OK;
}
int launch_aux_denc(const float* dm, const float* o2, const int* idx, float* denc, int nS, int S, int Tsrc, int L, int N, hipStream_t s) {
  if (!dm || !o2 || !idx || !denc || nS <= 0 || S <= 0 || nS % S || N % 4) return SEPR_EINVAL;
  const long long total4 = (long long)(nS / S) * L * N / 4;
  hipLaunchKernelGGL(aux_denc_kernel, dim3(grid_for(total4, TPB, 1 << 16)), dim3(TPB), 0, s, dm, o2, idx, denc, S, Tsrc, L, N, total4);
  SEPR_CHECK_LAUNCH("aux_denc_kernel");
  return SEPR_OK;
}
int launch_enc_bwd_pre(float* de, const float* add, const float* wav, const float* w, int B, int T, int L, int N, int K, int stride,
                       hipStream_t s) {
  if (!de || !wav || !w || B <= 0 || L <= 0 || N % 4 || (L - 1) * stride + K > T) return SEPR_EINVAL;
  const long long total4 = (long long)B * L * N / 4;
  hipLaunchKernelGGL(enc_bwd_pre_kernel, dim3(grid_for(total4, TPB, 1 << 16)), dim3(TPB), 0, s, de, add, wav, w, T, L, N, K, stride, total4);
  SEPR_CHECK_LAUNCH("enc_bwd_pre_kernel");
  return SEPR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// parameter-gradient finishers (weight-sized tensors; see sepr_train.h for the algebra)
// ---------------------------------------------------------------------------------------------------------------------
namespace {
// ONE launch (round 5; two before - ~200 of each per step, 5-6 us apiece, i.e. launch-latency bound): blocks [0, N) each own one weight
// row n (dW, dbias), blocks [N, N + ceil(K / 64)) each own 64 input channels k x 16 row lanes (dg, db: rows n = lane, lane + 16, ... are
// read as coalesced 256-byte segments).  Both parts only READ dWh / s and write disjoint gradient tensors, so they need no ordering.
__global__ __launch_bounds__(1024) void finish_norm_kernel(const float* __restrict__ dWh, const float* __restrict__ s, const float* __restrict__ W,
                                                          const float* __restrict__ g, const float* __restrict__ b, float* __restrict__ dW_g,
                                                          float* __restrict__ dbias_g, float* __restrict__ dg_g, float* __restrict__ db_g, int N, int K) {
  __shared__ float sh[16][64][2];
  if ((int)blockIdx.x < N) {
    const int n = blockIdx.x;
    const float sn = s[n];
    for (int k = threadIdx.x; k < K; k += 1024) dW_g[(long long)n * K + k] += fmaf(dWh[(long long)n * K + k], g[k], sn * b[k]);
    if (threadIdx.x == 0 && dbias_g) dbias_g[n] += sn;
    return;
  }
  const int kl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int k = ((int)blockIdx.x - N) * 64 + kl;
  float a = 0.f, c = 0.f;
  if (k < K) {
#pragma unroll 4
    for (int n = rl; n < N; n += 16) {
      const float w = W[(long long)n * K + k];
      a = fmaf(w, dWh[(long long)n * K + k], a);
      c = fmaf(s[n], w, c);
    }
  }
  sh[rl][kl][0] = a;
  sh[rl][kl][1] = c;
  __syncthreads();
  if (rl == 0 && k < K) {
    float ta = 0.f, tc = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { ta += sh[r][kl][0]; tc += sh[r][kl][1]; }
    dg_g[k] += ta;
    db_g[k] += tc;
  }
}
__global__ __launch_bounds__(64) void finish_ls_kernel(const float* __restrict__ Gr, const float* __restrict__ s, const float* __restrict__ W,
                                                      const float* __restrict__ bias, const float* __restrict__ ls, float* __restrict__ dW_g,
                                                      float* __restrict__ dbias_g, float* __restrict__ dls_g, int N, int K) {
  const int n = blockIdx.x;
  const float l = ls[n];
  float a = 0.f;
  for (int k = threadIdx.x; k < K; k += 64) {
    const float gr = Gr[(long long)n * K + k];
    dW_g[(long long)n * K + k] += l * gr;
    a = fmaf(W[(long long)n * K + k], gr, a);
  }
  a = wave_sum(a);
  if (threadIdx.x == 0) {
    dbias_g[n] += l * s[n];
    dls_g[n] += fmaf(bias[n], s[n], a);
  }
}
}  // namespace
// ---- deferred / batched form (sepr_train.h "Deferred finishers") --------------------------------------------------------------------
namespace {
struct FinJob {
  int kind, N, K, pad;                 // kind 0: LayerNorm-folded (finish_norm_kernel), 1: LayerScale (finish_ls_kernel)
  const float *G, *s, *W, *p0, *p1;    // kind 0: p0 = gamma, p1 = beta;  kind 1: p0 = bias, p1 = ls
  float *o0, *o1, *o2, *o3;            // kind 0: dW, dbias, dgamma, dbeta;  kind 1: dW, dbias, dls, -
};
constexpr int FIN_BATCH = 8;
struct FinBatch { FinJob j[FIN_BATCH]; };
// grid (max blocks of any job in the batch, jobs): block (x, y) runs block x of job y's own kernel, bit for bit the immediate form
__global__ __launch_bounds__(1024) void finish_batch_kernel(const FinBatch b) {
  const FinJob& j = b.j[blockIdx.y];
  if (j.kind == 0) {
    __shared__ float sh[16][64][2];
    if ((int)blockIdx.x >= j.N + (j.K + 63) / 64) return;
    if ((int)blockIdx.x < j.N) {
      const int n = blockIdx.x;
      const float sn = j.s[n];
      for (int k = threadIdx.x; k < j.K; k += 1024) j.o0[(long long)n * j.K + k] += fmaf(j.G[(long long)n * j.K + k], j.p0[k], sn * j.p1[k]);
      if (threadIdx.x == 0 && j.o1) j.o1[n] += sn;
      return;
    }
    const int kl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int k = ((int)blockIdx.x - j.N) * 64 + kl;
    float a = 0.f, c = 0.f;
    if (k < j.K) {
#pragma unroll 4
      for (int n = rl; n < j.N; n += 16) {
        const float w = j.W[(long long)n * j.K + k];
        a = fmaf(w, j.G[(long long)n * j.K + k], a);
        c = fmaf(j.s[n], w, c);
      }
    }
    sh[rl][kl][0] = a;
    sh[rl][kl][1] = c;
    __syncthreads();
    if (rl == 0 && k < j.K) {
      float ta = 0.f, tc = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { ta += sh[r][kl][0]; tc += sh[r][kl][1]; }
      j.o2[k] += ta;
      j.o3[k] += tc;
    }
  } else {
    if ((int)blockIdx.x >= j.N || threadIdx.x >= 64) return;
    const int n = blockIdx.x;
    const float l = j.p1[n];
    float a = 0.f;
    for (int k = threadIdx.x; k < j.K; k += 64) {
      const float gr = j.G[(long long)n * j.K + k];
      j.o0[(long long)n * j.K + k] += l * gr;
      a = fmaf(j.W[(long long)n * j.K + k], gr, a);
    }
    a = wave_sum(a);
    if (threadIdx.x == 0) {
      j.o1[n] += l * j.s[n];
      j.o2[n] += fmaf(j.p0[n], j.s[n], a);
    }
  }
}
struct FinQueue {
  bool open = false;
  char* arena = nullptr;
  size_t bytes = 0, off = 0;
  std::vector<FinJob> jobs;
};
thread_local FinQueue g_fin;
bool fin_in_arena(const void* p) {
  const char* c = static_cast<const char*>(p);
  return g_fin.open && c >= g_fin.arena && c < g_fin.arena + g_fin.bytes;
}
}  // namespace
bool fin_defers(const void* p) { return fin_in_arena(p); }
float* fin_alloc(size_t count) {
  if (!g_fin.open) return nullptr;
  const size_t o = align_up(g_fin.off), n = count * sizeof(float);
  if (o + n > g_fin.bytes) return nullptr;
  g_fin.off = o + n;
  return reinterpret_cast<float*>(g_fin.arena + o);
}
int fin_flush(hipStream_t st) {
  // Jobs of one launch run CONCURRENTLY, so two jobs that accumulate into the same tensor (the q / k / v finishers share the LayerNorm's
  // dgamma / dbeta) never share a launch, and launches run in order: a job goes into the first batch with room AFTER the batch of the last
  // earlier job it clashes with.  Every destination therefore sees its additions in the immediate form's order - bit-identical gradients.
  // The last batch that writes each destination is tracked over the WHOLE queue (round 6; rounds 5's 16-job look-back would have let two
  // accumulations into one tensor more than 16 jobs apart - a tied weight, a block reused across levels - share a launch).
  SEPR_TRY(tn_flush_pending());        // the last contraction's reduction (riding reduction, sepr_gemm_tn.hip): the finishers read its G
  const size_t nj = g_fin.jobs.size();
  std::unordered_map<const float*, int> last_batch;      // destination -> batch of the last queued job that accumulates into it
  std::vector<int> batch_of(nj, 0), fill;
  for (size_t i = 0; i < nj; ++i) {
    const FinJob& c = g_fin.jobs[i];
    const float* oc[4] = {c.o0, c.o1, c.o2, c.o3};
    int b0 = 0;
    for (int y = 0; y < 4; ++y) {
      if (!oc[y]) continue;
      const auto it = last_batch.find(oc[y]);
      if (it != last_batch.end() && it->second + 1 > b0) b0 = it->second + 1;
    }
    int bsel = b0;
    while (bsel < (int)fill.size() && fill[bsel] >= FIN_BATCH) ++bsel;
    if (bsel >= (int)fill.size()) fill.resize(bsel + 1, 0);
    batch_of[i] = bsel;
    ++fill[bsel];
    for (int y = 0; y < 4; ++y)
      if (oc[y]) last_batch[oc[y]] = bsel;
  }
  for (int bi = 0; bi < (int)fill.size(); ++bi) {
    if (fill[bi] == 0) continue;
    FinBatch b;
    int cnt = 0, bx = 1;
    for (size_t i = 0; i < nj; ++i) {
      if (batch_of[i] != bi) continue;
      const FinJob& c = g_fin.jobs[i];
      b.j[cnt++] = c;
      const int need = c.kind == 0 ? c.N + (c.K + 63) / 64 : c.N;
      if (need > bx) bx = need;
    }
    for (int q = cnt; q < FIN_BATCH; ++q) b.j[q] = b.j[0];
    hipLaunchKernelGGL(finish_batch_kernel, dim3(bx, cnt), dim3(1024), 0, st, b);
  }
  g_fin.jobs.clear();
  g_fin.off = 0;                       // (everything queued has been consumed in stream order: the arena can be carved again)
  SEPR_CHECK_LAUNCH("finish_batch_kernel");
  return SEPR_OK;
}
int launch_finish_norm_linear(const float* dWh, const float* s, const float* W, const float* g, const float* b, float* dW_g,
                              float* dbias_g, float* dg_g, float* db_g, int N, int K, hipStream_t st) {
  if (!dWh || !s || !W || !g || !b || !dW_g || !dg_g || !db_g || N <= 0 || K <= 0) return SEPR_EINVAL;
  if (fin_in_arena(dWh) && fin_in_arena(s)) {
    g_fin.jobs.push_back(FinJob{0, N, K, 0, dWh, s, W, g, b, dW_g, dbias_g, dg_g, db_g});
    return SEPR_OK;
  }
  st = wgrad_stream(st);       // (not deferred: behind the reduction that produced dWh, wherever that ran)
  hipLaunchKernelGGL(finish_norm_kernel, dim3(N + (K + 63) / 64), dim3(1024), 0, st, dWh, s, W, g, b, dW_g, dbias_g, dg_g, db_g, N, K);
  SEPR_CHECK_LAUNCH("finish_norm_linear kernels");
  return SEPR_OK;
}
int launch_finish_linear_ls(const float* Gr, const float* s, const float* W, const float* bias, const float* ls, float* dW_g,
                            float* dbias_g, float* dls_g, int N, int K, hipStream_t st) {
  if (!Gr || !s || !W || !bias || !ls || !dW_g || !dbias_g || !dls_g || N <= 0 || K <= 0) return SEPR_EINVAL;
  if (fin_in_arena(Gr) && fin_in_arena(s)) {
    g_fin.jobs.push_back(FinJob{1, N, K, 0, Gr, s, W, bias, ls, dW_g, dbias_g, dls_g, nullptr});
    return SEPR_OK;
  }
  st = wgrad_stream(st);
  hipLaunchKernelGGL(finish_ls_kernel, dim3(N), dim3(64), 0, st, Gr, s, W, bias, ls, dW_g, dbias_g, dls_g, N, K);
  SEPR_CHECK_LAUNCH("finish_ls_kernel");
  return SEPR_OK;
}

}  // namespace sepr

namespace sepr {
namespace {
struct WgradSide {
  hipStream_t side = nullptr;
  hipEvent_t fork = nullptr, join = nullptr, mark[2] = {nullptr, nullptr};
  bool marked[2] = {false, false};
  bool dirty = false;          // something was launched on the side stream since the last join
  bool forked = false;         // ... since the stream was registered (before that a mark has nothing to wait for - and, inside a hipGraph capture,
                               // the side stream is not part of the capture yet: an event recorded on it could not be waited for)
};
thread_local WgradSide g_wg;
}  // namespace
hipStream_t wgrad_stream(hipStream_t main) {
  if (!g_wg.side) return main;
  (void)hipEventRecord(g_wg.fork, main);                 // everything the caller has issued so far (the contraction's inputs) ...
  (void)hipStreamWaitEvent(g_wg.side, g_wg.fork, 0);     // ... happens before what follows on the side stream
  g_wg.dirty = true;
  g_wg.forked = true;
  return g_wg.side;
}
void wgrad_join(hipStream_t main) {
  if (!g_wg.side || !g_wg.dirty) return;
  (void)hipEventRecord(g_wg.join, g_wg.side);
  (void)hipStreamWaitEvent(main, g_wg.join, 0);
  g_wg.dirty = false;
}
}  // namespace sepr

extern "C" int sepr_train_wgrad_stream(sepr_stream_t side) {
  using namespace sepr;
  if (side && !g_wg.fork) {
    if (hipEventCreateWithFlags(&g_wg.fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&g_wg.join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g_wg.mark[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&g_wg.mark[1], hipEventDisableTiming) != hipSuccess)
      return SEPR_EHIP;
  }
  if (!side && g_wg.side && g_wg.dirty) return SEPR_EINVAL;      // join first (sepr_train_wgrad_join / sepr_train_defer_flush)
  g_wg.side = static_cast<hipStream_t>(side);
  g_wg.marked[0] = g_wg.marked[1] = false;
  g_wg.dirty = false;
  g_wg.forked = false;
  return SEPR_OK;
}
extern "C" int sepr_train_wgrad_join(sepr_stream_t stream) {
  sepr::wgrad_join(static_cast<hipStream_t>(stream));
  return SEPR_OK;
}
extern "C" int sepr_train_wgrad_mark(int slot) {
  using namespace sepr;
  if (slot < 0 || slot > 1) return SEPR_EINVAL;
  if (!g_wg.side) return SEPR_OK;
  if (!g_wg.forked) {          // nothing has run on the side stream in this window
    g_wg.marked[slot] = false;
    return SEPR_OK;
  }
  (void)hipEventRecord(g_wg.mark[slot], g_wg.side);
  g_wg.marked[slot] = true;
  return SEPR_OK;
}
extern "C" int sepr_train_wgrad_wait(int slot, sepr_stream_t stream) {
  using namespace sepr;
  if (slot < 0 || slot > 1) return SEPR_EINVAL;
  if (!g_wg.side || !g_wg.marked[slot]) return SEPR_OK;
  (void)hipStreamWaitEvent(static_cast<hipStream_t>(stream), g_wg.mark[slot], 0);
  return SEPR_OK;
}

extern "C" int sepr_train_defer_begin(void* arena, size_t arena_bytes) {
  using namespace sepr;
  if (!arena || arena_bytes < 4096) return SEPR_EINVAL;
  if (g_fin.open && !g_fin.jobs.empty()) return SEPR_EINVAL;      // a window with queued work must be flushed first
  g_fin.open = true;
  g_fin.arena = static_cast<char*>(arena);
  g_fin.bytes = arena_bytes;
  g_fin.off = 0;
  g_fin.jobs.clear();
  tn_parts_set(nullptr, 0);
  return SEPR_OK;
}
extern "C" int sepr_train_defer_parts(void* parts, size_t parts_bytes) {
  using namespace sepr;
  if (!g_fin.open) return SEPR_EINVAL;                 // (the double buffer belongs to a window: sepr_train_defer_begin first)
  if (parts && (parts_bytes < 4096 || ((uintptr_t)parts & 255))) return SEPR_EINVAL;
  SEPR_TRY(tn_flush_pending());
  tn_parts_set(parts, parts ? parts_bytes : 0);
  return SEPR_OK;
}
extern "C" int sepr_train_defer_flush(int close, sepr_stream_t stream) {
  using namespace sepr;
  wgrad_join(static_cast<hipStream_t>(stream));      // the queued finishers read what the side stream's reductions wrote
  if (!g_fin.open) return SEPR_OK;
  const int rc = fin_flush(static_cast<hipStream_t>(stream));
  if (close) {
    g_fin.open = false;
    tn_parts_set(nullptr, 0);
  }
  return rc;
}
