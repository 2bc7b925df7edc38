// HBM-bound kernels of the separator path: statistics, depthwise convolutions along the frame axis,
// speaker mixing, waveform encoder / decoder.  All activations are channel-last fp32 rows, so a wave's
// 64 lanes always sweep contiguous channels of one frame (16-byte loads where the op allows it).
#include "sepr_pointwise.h"
#include <stdlib.h>

namespace sepr {

int persistent_grid();   // sepr_gemm.hip: 2 workgroups per CU, a multiple of 8

namespace {
constexpr int TPB = 256;

// SEPR_LEGACY_POINTWISE=1 routes the 65-tap depthwise conv and the speaker mix through their first-generation
// kernels (A/B measurements and bisecting a parity failure; read once)
static bool legacy_pointwise() {
  static const bool v = [] {
    const char* e = getenv("SEPR_LEGACY_POINTWISE");
    return e && e[0] == '1';
  }();
  return v;
}

__device__ __forceinline__ float dot4(float4 a, float4 b) {
  return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)));
}
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
  return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}

// sum of (s, ss) over the block -> thread 0 (as doubles)
__device__ __forceinline__ void block_sum2(double& s, double& ss, double* sh /*[2*4]*/) {
  s = wave_sum_d(s);
  ss = wave_sum_d(ss);
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) {
    sh[2 * wid] = s;
    sh[2 * wid + 1] = ss;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    s = 0.0;
    ss = 0.0;
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) {
      s += sh[2 * i];
      ss += sh[2 * i + 1];
    }
  }
}
}  // namespace

// ---------------------------------------------------------------------------------------------------
// LayerNorm statistics: 16 lanes per row, two passes over registers (exact mean, then centred variance)
// reference: torch.nn.LayerNorm(F) in modules/network.py:50,81,133,162 (eps 1e-5)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void rowstats_kernel(const float* __restrict__ X, float* __restrict__ stats,
                                                      long long M, int F, float eps) {
  const int sub = threadIdx.x & 15;
  const int nf4 = F >> 2;
  const float invF = 1.0f / (float)F;
  for (long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4); row < M; row += (long long)gridDim.x * 16) {
    float mean, rstd;
    rowstats_one(X + row * F, nf4, invF, eps, sub, mean, rstd);      // (sepr_common.h: shared with the projection core's statistics tail)
    if (sub == 0) {
      stats[2 * row] = mean;
      stats[2 * row + 1] = rstd;
    }
  }
}

int launch_rowstats(const float* X, float* stats, long long M, int F, float eps, hipStream_t s) {
  if (M <= 0) return SEPR_OK;
  if (F % 4 != 0 || F > 512 || F <= 0) return SEPR_EINVAL;
  const long long blocks = (M + 15) / 16;
  const int grid = (int)(blocks < (1 << 20) ? blocks : (1 << 20));
  hipLaunchKernelGGL(rowstats_kernel, dim3(grid), dim3(TPB), 0, s, X, stats, M, F, eps);
  SEPR_CHECK_LAUNCH("rowstats_kernel");
  return SEPR_OK;
}

// ---------------------------------------------------------------------------------------------------
// GroupNorm(1 group): deterministic two-stage reduction (per-chunk partials in fp64, no atomics)
// reference: torch.nn.GroupNorm(1, C, eps=1e-8), modules/module.py:28,117
// ---------------------------------------------------------------------------------------------------
constexpr long long GN_CHUNK = 32768;
int gn_chunks(long long count) { return (int)((count + GN_CHUNK - 1) / GN_CHUNK); }

__global__ __launch_bounds__(TPB) void gn_partial_kernel(const float* __restrict__ X, double* __restrict__ part,
                                                        long long count, int nchunk) {
  __shared__ double sh[8];
  const int chunk = blockIdx.x, seq = blockIdx.y;
  const long long beg = (long long)chunk * GN_CHUNK;
  const long long end = (beg + GN_CHUNK < count) ? beg + GN_CHUNK : count;
  const float* p = X + (long long)seq * count;
  float s = 0.f, ss = 0.f;
  for (long long i = beg + 4LL * threadIdx.x; i < end; i += 4LL * TPB) {
    const float4 v = ld4(p + i);
    s += sum4(v);
    ss += dot4(v, v);
  }
  double ds = s, dss = ss;
  block_sum2(ds, dss, sh);
  if (threadIdx.x == 0) {
    part[2 * ((long long)seq * nchunk + chunk)] = ds;
    part[2 * ((long long)seq * nchunk + chunk) + 1] = dss;
  }
}

__global__ __launch_bounds__(64) void gn_finalize_kernel(const double* __restrict__ part, int nchunk, long long count,
                                                        float eps, float* __restrict__ stats) {
  const int seq = blockIdx.x;
  double s = 0.0, ss = 0.0;
  for (int i = threadIdx.x; i < nchunk; i += 64) {
    s += part[2 * ((long long)seq * nchunk + i)];
    ss += part[2 * ((long long)seq * nchunk + i) + 1];
  }
  s = wave_sum_d(s);
  ss = wave_sum_d(ss);
  if (threadIdx.x == 0) {
    const double mean = s / (double)count;
    double var = ss / (double)count - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[2 * seq] = (float)mean;
    stats[2 * seq + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

int launch_gn_partial(const float* X, double* part, int n, long long count, int nchunk, hipStream_t s) {
  if (n <= 0) return SEPR_OK;
  if (count % 4 != 0) return SEPR_EINVAL;
  hipLaunchKernelGGL(gn_partial_kernel, dim3(nchunk, n), dim3(TPB), 0, s, X, part, count, nchunk);
  SEPR_CHECK_LAUNCH("gn_partial_kernel");
  return SEPR_OK;
}
int launch_gn_finalize(const double* part, int n, int nchunk, long long count, float eps, float* stats, hipStream_t s) {
  if (n <= 0) return SEPR_OK;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(n), dim3(64), 0, s, part, nchunk, count, eps, stats);
  SEPR_CHECK_LAUNCH("gn_finalize_kernel");
  return SEPR_OK;
}

__global__ __launch_bounds__(TPB) void gn_apply_kernel(float* __restrict__ X, const float* __restrict__ stats,
                                                      const float* __restrict__ g, const float* __restrict__ b,
                                                      long long total4, long long per_seq4, int F4) {
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total4; i += (long long)gridDim.x * TPB) {
    const long long seq = i / per_seq4;
    const int f4 = (int)(i % F4);
    const float mean = stats[2 * seq], rstd = stats[2 * seq + 1];
    const float4 gg = ld4(g + 4 * f4), bb = ld4(b + 4 * f4);
    float4 v = ld4(X + 4 * i);
    v.x = (v.x - mean) * rstd * gg.x + bb.x;
    v.y = (v.y - mean) * rstd * gg.y + bb.y;
    v.z = (v.z - mean) * rstd * gg.z + bb.z;
    v.w = (v.w - mean) * rstd * gg.w + bb.w;
    st4(X + 4 * i, v);
  }
}

int launch_gn_apply(float* X, const float* stats, const float* g, const float* b, int n, int T, int F, hipStream_t s) {
  if (n <= 0 || T <= 0) return SEPR_OK;
  if (F % 4 != 0) return SEPR_EINVAL;
  const long long per_seq4 = (long long)T * F / 4, total4 = per_seq4 * n;
  const long long blocks = (total4 + TPB - 1) / TPB;
  const int grid = (int)(blocks < 65536 ? blocks : 65536);
  hipLaunchKernelGGL(gn_apply_kernel, dim3(grid), dim3(TPB), 0, s, X, stats, g, b, total4, per_seq4, F / 4);
  SEPR_CHECK_LAUNCH("gn_apply_kernel");
  return SEPR_OK;
}

// ---------------------------------------------------------------------------------------------------
// EGA pooling: adaptive_avg_pool1d with T = Tp*fac is an exact mean of fac frames (network.py:146)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void pool_kernel(const float* __restrict__ X, float* __restrict__ Y, long long total4,
                                                  int F4, int fac) {
  const float inv = 1.0f / (float)fac;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total4; i += (long long)gridDim.x * TPB) {
    const long long row = i / F4;  // seq*Tp + tp : the source rows are row*fac .. row*fac+fac-1
    const int f4 = (int)(i % F4);
    const float* p = X + (row * fac * F4 + f4) * 4;
    float4 a = ld4(p);
    for (int j = 1; j < fac; ++j) {
      const float4 v = ld4(p + (long long)j * F4 * 4);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    st4(Y + 4 * i, make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv));
  }
}

int launch_pool(const float* X, float* Y, int n, int Tp, int fac, int F, hipStream_t s) {
  if (n <= 0 || Tp <= 0) return SEPR_OK;
  if (F % 4 != 0 || fac < 1) return SEPR_EINVAL;
  const long long total4 = (long long)n * Tp * F / 4;
  const long long blocks = (total4 + TPB - 1) / TPB;
  const int grid = (int)(blocks < 65536 ? blocks : 65536);
  hipLaunchKernelGGL(pool_kernel, dim3(grid), dim3(TPB), 0, s, X, Y, total4, F / 4, fac);
  SEPR_CHECK_LAUNCH("pool_kernel");
  return SEPR_OK;
}

// The same mean, fused with the LayerNorm statistics of the pooled rows (what the q / k / v projection's prologue needs next):
// 16 lanes per pooled row as in rowstats_kernel, the pooled values stay in registers for both passes.  Summation orders are
// those of pool_kernel and rowstats_kernel, so outputs and statistics are bit-identical to the two-launch form.
template <int FAC>   // FAC > 0: the factor at compile time (all of a column's source rows are requested before the first add)
__global__ __launch_bounds__(TPB) void pool_stats_kernel(const float* __restrict__ X, float* __restrict__ Y, float* __restrict__ stats,
                                                        long long Mp, int F, int fac, float eps) {
  const int sub = threadIdx.x & 15;
  const int nf4 = F >> 2;
  const float invF = 1.0f / (float)F, inv = 1.0f / (float)fac;
  for (long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4); row < Mp; row += (long long)gridDim.x * 16) {
    const float* p = X + row * fac * F;
    float4 v[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = sub + 16 * i;
      v[i] = zero4();
      if (c < nf4) {
        float4 a;
        if (FAC > 0) {
          float4 q[FAC > 0 ? FAC : 1];
#pragma unroll
          for (int j = 0; j < FAC; ++j) q[j] = ld4(p + (long long)j * F + 4 * c);
          a = q[0];
#pragma unroll
          for (int j = 1; j < FAC; ++j) { a.x += q[j].x; a.y += q[j].y; a.z += q[j].z; a.w += q[j].w; }
        } else {
          a = ld4(p + 4 * c);
          for (int j = 1; j < fac; ++j) {
            const float4 q = ld4(p + (long long)j * F + 4 * c);
            a.x += q.x; a.y += q.y; a.z += q.z; a.w += q.w;
          }
        }
        v[i] = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
        st4(Y + row * F + 4 * c, v[i]);
      }
      s += sum4(v[i]);
    }
    const float mean = reduce16(s) * invF;
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = sub + 16 * i;
      if (c < nf4) {
        const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, e = v[i].w - mean;
        d += (a * a + b * b) + (cc * cc + e * e);
      }
    }
    const float var = reduce16(d) * invF;
    if (sub == 0) {
      stats[2 * row] = mean;
      stats[2 * row + 1] = 1.0f / sqrtf(var + eps);
    }
  }
}

int launch_pool_stats(const float* X, float* Y, float* stats, int n, int Tp, int fac, int F, float eps, hipStream_t s) {
  if (n <= 0 || Tp <= 0) return SEPR_OK;
  if (F % 4 != 0 || F > 512 || F <= 0 || fac < 1) return SEPR_EINVAL;
  const long long Mp = (long long)n * Tp;
  const long long blocks = (Mp + 15) / 16;
  const int grid = (int)(blocks < (1 << 20) ? blocks : (1 << 20));
  switch (fac) {
    case 2: hipLaunchKernelGGL(pool_stats_kernel<2>, dim3(grid), dim3(TPB), 0, s, X, Y, stats, Mp, F, fac, eps); break;
    case 4: hipLaunchKernelGGL(pool_stats_kernel<4>, dim3(grid), dim3(TPB), 0, s, X, Y, stats, Mp, F, fac, eps); break;
    case 8: hipLaunchKernelGGL(pool_stats_kernel<8>, dim3(grid), dim3(TPB), 0, s, X, Y, stats, Mp, F, fac, eps); break;
    case 16: hipLaunchKernelGGL(pool_stats_kernel<16>, dim3(grid), dim3(TPB), 0, s, X, Y, stats, Mp, F, fac, eps); break;
    default: hipLaunchKernelGGL(pool_stats_kernel<0>, dim3(grid), dim3(TPB), 0, s, X, Y, stats, Mp, F, fac, eps);
  }
  SEPR_CHECK_LAUNCH("pool_stats_kernel");
  return SEPR_OK;
}

// ---------------------------------------------------------------------------------------------------
// GCFN middle: depthwise Conv1d(6F, k=3, pad=1) along frames + GLU (network.py:52-54,62-65).
// A thread owns 4 value channels + their 4 gate channels and slides down a strip of frames, so every
// hidden element is loaded once (+2 halo frames per strip).
// ---------------------------------------------------------------------------------------------------
constexpr int DWGLU_STRIP = 8;

__global__ __launch_bounds__(TPB) void dwglu_kernel(const float* __restrict__ H, float* __restrict__ G, long long total,
                                                   int T, int F, int strips, const float* __restrict__ w,
                                                   const float* __restrict__ b) {
  const long long gid = (long long)blockIdx.x * TPB + threadIdx.x;
  if (gid >= total) return;
  const int C3 = 3 * F, C6 = 6 * F, ncg = C3 / 4;
  const int cg = (int)(gid % ncg);
  const long long sidx = gid / ncg;
  const int seq = (int)(sidx / strips);
  const int t0 = (int)(sidx % strips) * DWGLU_STRIP;
  const int t1 = (t0 + DWGLU_STRIP < T) ? t0 + DWGLU_STRIP : T;
  const int c = 4 * cg;
  const float4 wv0 = ld4(w + c), wv1 = ld4(w + C6 + c), wv2 = ld4(w + 2 * C6 + c);
  const float4 wg0 = ld4(w + C3 + c), wg1 = ld4(w + C6 + C3 + c), wg2 = ld4(w + 2 * C6 + C3 + c);
  const float4 bv = ld4(b + c), bg = ld4(b + C3 + c);
  const float* base = H + (long long)seq * T * C6 + c;
  float4 pv = zero4(), pg = zero4();
  if (t0 > 0) {
    pv = ld4(base + (long long)(t0 - 1) * C6);
    pg = ld4(base + (long long)(t0 - 1) * C6 + C3);
  }
  float4 cv = ld4(base + (long long)t0 * C6), cgt = ld4(base + (long long)t0 * C6 + C3);
  float* out = G + ((long long)seq * T) * C3 + c;
  for (int t = t0; t < t1; ++t) {
    float4 nv = zero4(), ng = zero4();
    if (t + 1 < T) {
      nv = ld4(base + (long long)(t + 1) * C6);
      ng = ld4(base + (long long)(t + 1) * C6 + C3);
    }
    const float4 val = fma4(wv2, nv, fma4(wv1, cv, fma4(wv0, pv, bv)));
    const float4 gat = fma4(wg2, ng, fma4(wg1, cgt, fma4(wg0, pg, bg)));
    st4(out + (long long)t * C3, make_float4(val.x * sigmoid_f(gat.x), val.y * sigmoid_f(gat.y),
                                             val.z * sigmoid_f(gat.z), val.w * sigmoid_f(gat.w)));
    pv = cv; pg = cgt; cv = nv; cgt = ng;
  }
}

int launch_dwglu(const float* H, float* G, int n, int T, int F, const float* w, const float* b, hipStream_t s) {
  if (n <= 0 || T <= 0) return SEPR_OK;
  if (F % 4 != 0) return SEPR_EINVAL;
  const int strips = (T + DWGLU_STRIP - 1) / DWGLU_STRIP;
  const long long total = (long long)n * strips * (3 * F / 4);
  const long long blocks = (total + TPB - 1) / TPB;
  if (blocks > 0x7fffffffLL) return SEPR_EINVAL;
  hipLaunchKernelGGL(dwglu_kernel, dim3((unsigned)blocks), dim3(TPB), 0, s, H, G, total, T, F, strips, w, b);
  SEPR_CHECK_LAUNCH("dwglu_kernel");
  return SEPR_OK;
}

// ---------------------------------------------------------------------------------------------------
// CLA middle: depthwise Conv1d(F, k=65, padding='same') along frames (network.py:165,179).
// "LDS staging of local chunks" = a 128-frame time tile with a +-32-frame halo per workgroup; each
// thread owns one channel and produces 16 consecutive frames per pass from an 80-value register window.
// ---------------------------------------------------------------------------------------------------
template <int KW, int CH>
__global__ __launch_bounds__(TPB) void dwconv_same_kernel(const float* __restrict__ U, float* __restrict__ C, int T, int F,
                                                         const float* __restrict__ w, const float* __restrict__ b) {
  constexpr int HALO = KW / 2;
  constexpr int TT = 128;
  constexpr int ROWS = TT + KW - 1;
  constexpr int G = TPB / CH;
  constexpr int OPT = 16;
  constexpr int RP = OPT * G;
  __shared__ __attribute__((aligned(16))) float tile[ROWS * CH];
  __shared__ __attribute__((aligned(16))) float ws[KW * CH];
  const int nchunks = F / CH;
  const int tix = blockIdx.x / nchunks, chunk = blockIdx.x % nchunks, seq = blockIdx.y;
  const int t0 = tix * TT, c0 = chunk * CH;
  const float* src = U + (long long)seq * T * F + c0;
  for (int i = threadIdx.x; i < ROWS * (CH / 4); i += TPB) {
    const int r = i / (CH / 4), q = i % (CH / 4);
    const int t = t0 - HALO + r;
    st4(tile + r * CH + 4 * q, (t >= 0 && t < T) ? ld4(src + (long long)t * F + 4 * q) : zero4());
  }
  for (int i = threadIdx.x; i < KW * (CH / 4); i += TPB) {
    const int j = i / (CH / 4), q = i % (CH / 4);
    st4(ws + j * CH + 4 * q, ld4(w + (long long)j * F + c0 + 4 * q));
  }
  __syncthreads();
  const int ch = threadIdx.x % CH, g = threadIdx.x / CH;
  const float bias = b[c0 + ch];
  float* dst = C + (long long)seq * T * F + c0 + ch;
#pragma unroll 1
  for (int pass = 0; pass < TT / RP; ++pass) {
    const int rb = pass * RP + g * OPT;
    if (t0 + rb >= T) continue;
    float in[OPT + KW - 1];
#pragma unroll
    for (int r = 0; r < OPT + KW - 1; ++r) in[r] = tile[(rb + r) * CH + ch];
    float acc[OPT];
#pragma unroll
    for (int o = 0; o < OPT; ++o) acc[o] = bias;
#pragma unroll
    for (int j = 0; j < KW; ++j) {
      const float wj = ws[j * CH + ch];
#pragma unroll
      for (int o = 0; o < OPT; ++o) acc[o] = fmaf(wj, in[o + j], acc[o]);
    }
#pragma unroll
    for (int o = 0; o < OPT; ++o) {
      const int t = t0 + rb + o;
      if (t < T) dst[(long long)t * F] = acc[o];
    }
  }
}

// Same op, built around what bounds it.  65 FMAs per element is VALU work (44 GFMA per B=32 forward, ~0.8 ms of the chip's
// packed-FMA rate) and the HBM pass (u in, c out) is ~0.9 ms: the two must OVERLAP, and the LDS-DMA tile load of a workgroup
// cannot overlap its own arithmetic (single tile buffer).  So:
//  * 64-channel slabs, 128 output frames per tile: 48 KB tile + 16.6 KB taps = 65 KB of LDS, TWO 4-wave workgroups per CU -
//    one waits for its tile while the other multiplies (round 1's one 129 KB workgroup per CU serialised the two: 1.9 ms);
//  * a lane owns TWO adjacent channels and every multiply-add is a v_pk_fma_f32; a wave = 32 frames x 64 channels (lanes 0-31
//    the first 16 frames, lanes 32-63 the next 16), 16 outputs x 2 channels per lane: a tile row is read from LDS once per
//    half-wave (80 + 65 ds_read_b64 per 1040 packed FMAs);
//  * the input tile goes global -> LDS by LDS-DMA (no VGPR round trip, four 256 B rows per wave instruction);
//  * persistent workgroups walk the tiles slab-major, so the taps are staged once per workgroup and slab.
// Accumulation order per output is bias, tap 0, ..., tap 64 with fused multiply-adds: the same chain as
// dwconv_same_kernel, so the two kernels agree bit for bit.
typedef float f32x2 __attribute__((ext_vector_type(2)));

// GLUB (training, CLA backward): the conv's output du [rows][F] (the gradient w.r.t. the GLU output) does not go to HBM - the epilogue is
// the GLU backward: with the saved pre-activation rows A [rows][2F] (value | gate), DA [rows][2F] = (du sig(g), du v sig(g) (1 - sig(g)))
// (glu_bwd_kernel's arithmetic); C is unused.
// OUT16 (training, plain-bf16 precision): the output rows (C, or DA with GLUB) are stored as bf16 - their only readers are MFMA operand
// loaders that round to bf16 anyway (linear2 and its weight gradient; linear1's weight gradient and input-gradient projection).
template <int KW, bool GLUB = false, bool OUT16 = false>
__global__ __launch_bounds__(256, 2) void dwconv_same_pk_kernel(const float* __restrict__ U, float* __restrict__ C, int T, int F,
                                                               int tiles_per_seq, int ntiles, const float* __restrict__ w,
                                                               const float* __restrict__ b, const float* __restrict__ A = nullptr,
                                                               float* __restrict__ DA = nullptr) {
  constexpr int HALO = KW / 2, TT = 128, ROWS = TT + KW - 1, CH = 64, OPT = 16, NT = 256, NWV = NT / 64;
  static_assert(ROWS % (4 * NWV) == 0, "DMA loop: 4 waves x 4 rows per instruction");
  __shared__ __attribute__((aligned(16))) float tile[ROWS * CH];
  __shared__ __attribute__((aligned(16))) float ws[KW * CH];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int per_chunk = ntiles / (F / CH);
  int cur_chunk = -1;
#pragma unroll 1
  for (int tix = blockIdx.x; tix < ntiles; tix += gridDim.x) {
    // slab-major tile order: a workgroup's consecutive tiles share the channel slab and its taps
    const int chunk = tix / per_chunk, rem = tix % per_chunk;
    const int seq = rem / tiles_per_seq, t0 = (rem % tiles_per_seq) * TT;
    const int c0 = chunk * CH;
    if (chunk != cur_chunk) {   // workgroup-uniform
      cur_chunk = chunk;
      for (int i = threadIdx.x; i < KW * (CH / 4); i += NT) {
        const int j = i / (CH / 4), q = i % (CH / 4);
        st4(ws + j * CH + 4 * q, ld4(w + (long long)j * F + c0 + 4 * q));
      }
    }
    const float* src = U + (long long)seq * T * F + c0;
    {  // rows t0-HALO .. t0+TT+HALO-1 -> LDS; out-of-range rows are fetched from a clamped address, zeroed below
      const int sub = lane >> 4, q = lane & 15;
#pragma unroll
      for (int i = 0; i < ROWS / (4 * NWV); ++i) {
        const int r0 = 4 * (i * NWV + wv);
        int t = t0 - HALO + r0 + sub;
        t = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
        const float* g = src + (long long)t * F + 4 * q;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(tile + r0 * CH), 16, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t0 < HALO || t0 + TT + HALO > T) {   // workgroup-uniform: the zero-padding rows of Conv1d(padding=32)
      for (int i = threadIdx.x; i < ROWS * (CH / 4); i += NT) {
        const int t = t0 - HALO + i / (CH / 4);
        if (t < 0 || t >= T) st4(tile + 4 * i, zero4());
      }
      __syncthreads();
    }
    const int rb = (2 * wv + (lane >> 5)) * OPT;   // first output frame of this half-wave
    if (t0 + 2 * wv * OPT < T) {   // wave-uniform (the second half-wave may compute frames past T: never stored)
      f32x2 acc[OPT];
      const int cp = 2 * (lane & 31);
      const f32x2 bias = *reinterpret_cast<const f32x2*>(b + c0 + cp);
#pragma unroll
      for (int o = 0; o < OPT; ++o) acc[o] = bias;
      const float* trow = tile + rb * CH + cp;
      const float* wrow = ws + cp;
      // 16 taps per trip of a real (not unrolled) loop: 256 packed FMAs against 16 new window rows + 16 taps, the
      // 15 rows the next trip re-uses carried in registers.  (One 1040-FMA basic block lets the scheduler hoist
      // all 145 LDS reads to the top and spill; a loop bounds what it can hoist.)
      static_assert(KW % 16 == 1 && OPT == 16, "tap loop: 16 taps per trip + the last tap");
      f32x2 xc[31];
#pragma unroll
      for (int i = 0; i < 15; ++i) xc[i] = *reinterpret_cast<const f32x2*>(trow + i * CH);
#pragma unroll 1
      for (int jb = 0; jb < KW - 1; jb += 16) {
        const float* tr = trow + jb * CH;
        const float* wr = wrow + jb * CH;
        f32x2 wq[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          xc[15 + i] = *reinterpret_cast<const f32x2*>(tr + (15 + i) * CH);
          wq[i] = *reinterpret_cast<const f32x2*>(wr + i * CH);
        }
#pragma unroll
        for (int jj = 0; jj < 16; ++jj)
#pragma unroll
          for (int o = 0; o < OPT; ++o) acc[o] = __builtin_elementwise_fma(wq[jj], xc[o + jj], acc[o]);
#pragma unroll
        for (int i = 0; i < 15; ++i) xc[i] = xc[i + 16];
      }
      {  // tap KW-1: window rows KW-1 .. KW+14 (xc[0..14] + one more row)
        const f32x2 wl = *reinterpret_cast<const f32x2*>(wrow + (KW - 1) * CH);
        const f32x2 xl = *reinterpret_cast<const f32x2*>(trow + (KW - 1 + 15) * CH);
#pragma unroll
        for (int o = 0; o < 15; ++o) acc[o] = __builtin_elementwise_fma(wl, xc[o], acc[o]);
        acc[15] = __builtin_elementwise_fma(wl, xl, acc[15]);
      }
      if constexpr (GLUB) {
        const long long row0 = (long long)seq * T + t0 + rb;
        const float* ap = A + row0 * 2 * F + c0 + cp;
        float* dp = DA + row0 * 2 * F + c0 + cp;
        f32x2 av[OPT], ag[OPT];
#pragma unroll
        for (int o = 0; o < OPT; ++o) {        // clamped rows: all loads issue back to back
          const long long ro = (t0 + rb + o < T) ? o : 0;
          av[o] = *reinterpret_cast<const f32x2*>(ap + ro * 2 * F);
          ag[o] = *reinterpret_cast<const f32x2*>(ap + ro * 2 * F + F);
        }
#pragma unroll
        for (int o = 0; o < OPT; ++o) {
          if (t0 + rb + o < T) {
            const float s0 = sigmoid_exact(ag[o][0]), s1 = sigmoid_exact(ag[o][1]);
            const f32x2 dv = {acc[o][0] * s0, acc[o][1] * s1};
            const f32x2 dg = {acc[o][0] * av[o][0] * s0 * (1.f - s0), acc[o][1] * av[o][1] * s1 * (1.f - s1)};
            if constexpr (OUT16) {
              typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
              __bf16* d16 = reinterpret_cast<__bf16*>(DA) + (row0 + o) * 2 * F + c0 + cp;
              *reinterpret_cast<bf16x2_t*>(d16) = (bf16x2_t){(__bf16)dv[0], (__bf16)dv[1]};
              *reinterpret_cast<bf16x2_t*>(d16 + F) = (bf16x2_t){(__bf16)dg[0], (__bf16)dg[1]};
            } else {
              *reinterpret_cast<f32x2*>(dp + (long long)o * 2 * F) = dv;
              *reinterpret_cast<f32x2*>(dp + (long long)o * 2 * F + F) = dg;
            }
          }
        }
      } else {
        if constexpr (OUT16) {
          typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
          __bf16* d16 = reinterpret_cast<__bf16*>(C) + ((long long)seq * T + t0 + rb) * F + c0 + cp;
#pragma unroll
          for (int o = 0; o < OPT; ++o)
            if (t0 + rb + o < T) *reinterpret_cast<bf16x2_t*>(d16 + (long long)o * F) = (bf16x2_t){(__bf16)acc[o][0], (__bf16)acc[o][1]};
        } else {
          float* dst = C + ((long long)seq * T + t0 + rb) * F + c0 + cp;
#pragma unroll
          for (int o = 0; o < OPT; ++o)
            if (t0 + rb + o < T) *reinterpret_cast<f32x2*>(dst + (long long)o * F) = acc[o];
        }
      }
    }
    __syncthreads();   // tile fully consumed before the next DMA overwrites it
  }
}

int launch_dwconv_same(const float* U, float* C, int n, int T, int F, int K, const float* w, const float* b, hipStream_t s) {
  if (n <= 0 || T <= 0) return SEPR_OK;
  if (K != 65 || F % 64 != 0) return SEPR_EINVAL;
  const int tiles = (T + 127) / 128;
  if (F % 128 == 0 && !legacy_pointwise()) {
    const long long nt = (long long)tiles * n * (F / 64);
    if (nt > 0x7fffffffLL) return SEPR_EINVAL;
    const int cap = persistent_grid();             // two 65 KB workgroups per CU
    const int grid = (int)(nt < cap ? nt : cap);
    hipLaunchKernelGGL((dwconv_same_pk_kernel<65>), dim3(grid), dim3(256), 0, s, U, C, T, F, tiles, (int)nt, w, b);
  } else if (F % 128 == 0) {
    hipLaunchKernelGGL((dwconv_same_kernel<65, 128>), dim3(tiles * (F / 128), n), dim3(TPB), 0, s, U, C, T, F, w, b);
  } else {
    hipLaunchKernelGGL((dwconv_same_kernel<65, 64>), dim3(tiles * (F / 64), n), dim3(TPB), 0, s, U, C, T, F, w, b);
  }
  SEPR_CHECK_LAUNCH("dwconv_same_kernel");
  return SEPR_OK;
}

// CLA backward: da [rows][2F] = GLU'(a) applied to du = depthwise_k65(dc) (taps w: the forward taps reversed; no bias) - the conv's
// output never reaches HBM (round 4; replaces launch_dwconv_same + launch_glu_bwd and the [rows][F] round trip between them)
// C as a bf16 tensor [n*T][F] (F % 128 == 0 only)
int launch_dwconv_same16(const float* U, float* C, int n, int T, int F, int K, const float* w, const float* b, hipStream_t s) {
  if (n <= 0 || T <= 0) return SEPR_OK;
  if (K != 65 || F % 128 != 0 || !U || !C || !w || !b) return SEPR_EINVAL;
  const int tiles = (T + 127) / 128;
  const long long nt = (long long)tiles * n * (F / 64);
  if (nt > 0x7fffffffLL) return SEPR_EINVAL;
  const int cap = persistent_grid();
  hipLaunchKernelGGL((dwconv_same_pk_kernel<65, false, true>), dim3((int)(nt < cap ? nt : cap)), dim3(256), 0, s, U, C, T, F, tiles, (int)nt, w, b);
  SEPR_CHECK_LAUNCH("dwconv_same_pk_kernel<out16>");
  return SEPR_OK;
}

int launch_dwconv_same_glu_bwd(const float* dc, const float* a, float* da, int n, int T, int F, int K, const float* w, const float* zero_bias,
                               hipStream_t s, int out16) {
  if (n <= 0 || T <= 0) return SEPR_OK;
  if (K != 65 || F % 128 != 0 || !dc || !a || !da || !w || !zero_bias) return SEPR_EINVAL;
  const int tiles = (T + 127) / 128;
  const long long nt = (long long)tiles * n * (F / 64);
  if (nt > 0x7fffffffLL) return SEPR_EINVAL;
  const int cap = persistent_grid();
  const int grid = (int)(nt < cap ? nt : cap);
  if (out16) hipLaunchKernelGGL((dwconv_same_pk_kernel<65, true, true>), dim3(grid), dim3(256), 0, s, dc, nullptr, T, F, tiles, (int)nt, w, zero_bias, a, da);
  else hipLaunchKernelGGL((dwconv_same_pk_kernel<65, true>), dim3(grid), dim3(256), 0, s, dc, nullptr, T, F, tiles, (int)nt, w, zero_bias, a, da);
  SEPR_CHECK_LAUNCH("dwconv_same_pk_kernel<glu_bwd>");
  return SEPR_OK;
}

// ---------------------------------------------------------------------------------------------------
// DownConvLayer: depthwise Conv1d(F, k, stride 2, pad (k-1)/2) + eval BatchNorm + GELU (module.py:63-78)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void downconv_kernel(const float* __restrict__ X, float* __restrict__ Y, long long total4,
                                                      int T, int To, int F4, int K, const float* __restrict__ w,
                                                      const float* __restrict__ scale, const float* __restrict__ shift) {
  const int pad = (K - 1) / 2;
  for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total4; i += (long long)gridDim.x * TPB) {
    const int f4 = (int)(i % F4);
    const long long row = i / F4;  // seq*To + to
    const int seq = (int)(row / To), to = (int)(row % To);
    const float* p = X + ((long long)seq * T * F4 + f4) * 4;
    float4 a = zero4();
    for (int j = 0; j < K; ++j) {
      const int t = 2 * to + j - pad;
      if (t >= 0 && t < T) a = fma4(ld4(w + (j * F4 + f4) * 4), ld4(p + (long long)t * F4 * 4), a);
    }
    const float4 sc = ld4(scale + 4 * f4), sh = ld4(shift + 4 * f4);
    st4(Y + 4 * i, make_float4(gelu_exact(fmaf(a.x, sc.x, sh.x)), gelu_exact(fmaf(a.y, sc.y, sh.y)),
                               gelu_exact(fmaf(a.z, sc.z, sh.z)), gelu_exact(fmaf(a.w, sc.w, sh.w))));
  }
}

int launch_downconv(const float* X, float* Y, int n, int T, int To, int F, int K, const float* w, const float* scale,
                    const float* shift, hipStream_t s) {
  if (n <= 0 || To <= 0) return SEPR_OK;
  if (F % 4 != 0 || K < 1 || K > 15 || (K & 1) == 0) return SEPR_EINVAL;
  const long long total4 = (long long)n * To * F / 4;
  const long long blocks = (total4 + TPB - 1) / TPB;
  const int grid = (int)(blocks < 65536 ? blocks : 65536);
  hipLaunchKernelGGL(downconv_kernel, dim3(grid), dim3(TPB), 0, s, X, Y, total4, T, To, F / 4, K, w, scale, shift);
  SEPR_CHECK_LAUNCH("downconv_kernel");
  return SEPR_OK;
}

// ---------------------------------------------------------------------------------------------------
// SpkAttention core: softmax(q_a . k_c / sqrt(dk)) over the S speakers of one frame, per head
// (network.py:241-244 regroups [B*S,F,T] to [B*T,S,F]; in channel-last rows speaker s of frame (b,t)
// is simply row (b*S+s)*T + t, so no regrouping copy exists here).
// ---------------------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(TPB) void spkmix_kernel(const float* __restrict__ QKV, float* __restrict__ O, long long total,
                                                    int T, int F, int H, float inv_sqrt_dk) {
  const long long gid = (long long)blockIdx.x * TPB + threadIdx.x;
  if (gid >= total) return;
  const int h = (int)(gid % H);
  const long long bt = gid / H;
  const int t = (int)(bt % T);
  const long long b = bt / T;
  const int dk = F / H, ld = 3 * F;
  const float* row[S];
#pragma unroll
  for (int s = 0; s < S; ++s) row[s] = QKV + ((b * S + s) * T + t) * ld + h * dk;
  float sc[S][S];
#pragma unroll
  for (int a = 0; a < S; ++a)
#pragma unroll
    for (int c = 0; c < S; ++c) sc[a][c] = 0.f;
  for (int d = 0; d < dk; d += 4) {
    float4 q[S], k[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
      q[s] = ld4(row[s] + d);
      k[s] = ld4(row[s] + F + d);
    }
#pragma unroll
    for (int a = 0; a < S; ++a)
#pragma unroll
      for (int c = 0; c < S; ++c) sc[a][c] += dot4(q[a], k[c]);
  }
#pragma unroll
  for (int a = 0; a < S; ++a) {
    float mx = sc[a][0] * inv_sqrt_dk;
#pragma unroll
    for (int c = 0; c < S; ++c) {
      sc[a][c] *= inv_sqrt_dk;
      mx = fmaxf(mx, sc[a][c]);
    }
    float den = 0.f;
#pragma unroll
    for (int c = 0; c < S; ++c) {
      sc[a][c] = __expf(sc[a][c] - mx);
      den += sc[a][c];
    }
    const float inv = 1.0f / den;
#pragma unroll
    for (int c = 0; c < S; ++c) sc[a][c] *= inv;
  }
  for (int d = 0; d < dk; d += 4) {
    float4 v[S];
#pragma unroll
    for (int s = 0; s < S; ++s) v[s] = ld4(row[s] + 2 * F + d);
#pragma unroll
    for (int a = 0; a < S; ++a) {
      float4 o = zero4();
#pragma unroll
      for (int c = 0; c < S; ++c) {
        o.x = fmaf(sc[a][c], v[c].x, o.x);
        o.y = fmaf(sc[a][c], v[c].y, o.y);
        o.z = fmaf(sc[a][c], v[c].z, o.z);
        o.w = fmaf(sc[a][c], v[c].w, o.w);
      }
      st4(O + ((b * S + a) * T + t) * F + h * dk + d, o);
    }
  }
}

// Coalesced form: a lane owns one float4 of one frame row (F/4 lanes per row, consecutive lanes = consecutive
// addresses, so every q/k/v load and every store is a full 512 B row segment), the dk/4 lanes of a head combine
// their partial q.k sums with an xor butterfly.  spkmix_kernel above (one lane = one whole head, 64 B lane
// stride) remains for head sizes whose lane group is not a power of two.
template <int S, int LPH>
__global__ __launch_bounds__(TPB) void spkmix_rows_kernel(const float* __restrict__ QKV, float* __restrict__ O,
                                                         long long total4, int T, int F4, float inv_sqrt_dk) {
  const long long gid = (long long)blockIdx.x * TPB + threadIdx.x;
  if (gid >= total4) return;   // total4 is a multiple of LPH and groups are LPH-aligned: a head is all in or out
  const int f4 = (int)(gid % F4);
  const long long bt = gid / F4;
  const int t = (int)(bt % T);
  const long long b = bt / T;
  const long long ld = 12LL * F4;
  float4 q[S], k[S], v[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const float* row = QKV + ((b * S + s) * T + t) * ld + 4 * f4;
    q[s] = ld4(row);
    k[s] = ld4(row + 4 * F4);
    v[s] = ld4(row + 8 * F4);
  }
  float sc[S][S];
#pragma unroll
  for (int a = 0; a < S; ++a)
#pragma unroll
    for (int c = 0; c < S; ++c) {
      float p = dot4(q[a], k[c]);
#pragma unroll
      for (int m = 1; m < LPH; m <<= 1) p += __shfl_xor(p, m, 64);
      sc[a][c] = p * inv_sqrt_dk;
    }
#pragma unroll
  for (int a = 0; a < S; ++a) {
    float mx = sc[a][0];
#pragma unroll
    for (int c = 1; c < S; ++c) mx = fmaxf(mx, sc[a][c]);
    float den = 0.f;
#pragma unroll
    for (int c = 0; c < S; ++c) {
      sc[a][c] = __expf(sc[a][c] - mx);
      den += sc[a][c];
    }
    const float inv = 1.0f / den;
    float4 o = zero4();
#pragma unroll
    for (int c = 0; c < S; ++c) {
      const float p = sc[a][c] * inv;
      o.x = fmaf(p, v[c].x, o.x);
      o.y = fmaf(p, v[c].y, o.y);
      o.z = fmaf(p, v[c].z, o.z);
      o.w = fmaf(p, v[c].w, o.w);
    }
    st4(O + (((b * S + a) * T + t) * F4 + f4) * 4, o);
  }
}

template <int S>
static bool launch_spkmix_rows(const float* QKV, float* O, int B, int T, int F, int H, float isd, hipStream_t s) {
  const int lph = F / H / 4;
  const long long total4 = (long long)B * T * (F / 4);
  const long long blocks = (total4 + TPB - 1) / TPB;
  if (blocks > 0x7fffffffLL) return false;
  const dim3 g((unsigned)blocks), t(TPB);
  switch (lph) {
    case 1: hipLaunchKernelGGL((spkmix_rows_kernel<S, 1>), g, t, 0, s, QKV, O, total4, T, F / 4, isd); return true;
    case 2: hipLaunchKernelGGL((spkmix_rows_kernel<S, 2>), g, t, 0, s, QKV, O, total4, T, F / 4, isd); return true;
    case 4: hipLaunchKernelGGL((spkmix_rows_kernel<S, 4>), g, t, 0, s, QKV, O, total4, T, F / 4, isd); return true;
    case 8: hipLaunchKernelGGL((spkmix_rows_kernel<S, 8>), g, t, 0, s, QKV, O, total4, T, F / 4, isd); return true;
    case 16: hipLaunchKernelGGL((spkmix_rows_kernel<S, 16>), g, t, 0, s, QKV, O, total4, T, F / 4, isd); return true;
    default: return false;
  }
}

int launch_spkmix(const float* QKV, float* O, int B, int S, int T, int F, int H, hipStream_t s) {
  if (B <= 0 || T <= 0) return SEPR_OK;
  if (H <= 0 || F % H != 0 || (F / H) % 4 != 0) return SEPR_EINVAL;
  const long long total = (long long)B * T * H;
  const long long blocks = (total + TPB - 1) / TPB;
  if (blocks > 0x7fffffffLL) return SEPR_EINVAL;
  const float isd = 1.0f / sqrtf((float)(F / H));
  const bool rows_form = !legacy_pointwise();
  if (rows_form && S == 2 && launch_spkmix_rows<2>(QKV, O, B, T, F, H, isd, s)) {
  } else if (rows_form && S == 3 && launch_spkmix_rows<3>(QKV, O, B, T, F, H, isd, s)) {
  } else if (S == 2) {
    hipLaunchKernelGGL((spkmix_kernel<2>), dim3((unsigned)blocks), dim3(TPB), 0, s, QKV, O, total, T, F, H, isd);
  } else if (S == 3) {
    hipLaunchKernelGGL((spkmix_kernel<3>), dim3((unsigned)blocks), dim3(TPB), 0, s, QKV, O, total, T, F, H, isd);
  } else {
    return SEPR_EINVAL;
  }
  SEPR_CHECK_LAUNCH("spkmix_kernel");
  return SEPR_OK;
}

// ---------------------------------------------------------------------------------------------------
// AudioEncoder: Conv1d(1 -> N, K taps, stride, no bias) + exact GELU (module.py:12-23); the epilogue also
// emits the per-tile (sum, sum^2) partials FeatureProjector's GroupNorm(1, N) needs (module.py:28,33).
// A workgroup = 64 frames of one utterance; a thread keeps its channel's K taps in registers.
// ---------------------------------------------------------------------------------------------------
constexpr int ENC_FT = 64;
int encoder_tiles(int L) { return (L + ENC_FT - 1) / ENC_FT; }

template <int K>
__global__ __launch_bounds__(TPB) void encoder_kernel(const float* __restrict__ wav, int T, int L, const float* __restrict__ w,
                                                     int N, int stride, float* __restrict__ E, double* __restrict__ part,
                                                     int ntile) {
  __shared__ float xs[ENC_FT * 8 + K];
  __shared__ double sh[8];
  const int tile = blockIdx.x, b = blockIdx.y;
  const int l0 = tile * ENC_FT;
  const int nl = (L - l0 < ENC_FT) ? L - l0 : ENC_FT;
  const int nsamp = (nl - 1) * stride + K;
  const float* src = wav + (long long)b * T + (long long)l0 * stride;
  for (int i = threadIdx.x; i < nsamp; i += TPB) xs[i] = src[i];
  __syncthreads();
  const int cw = N < TPB ? N : TPB;
  const int groups = TPB / cw;
  const int g = threadIdx.x / cw, c0 = threadIdx.x % cw;
  float s = 0.f, ss = 0.f;
  if (g < groups) {
    for (int c = c0; c < N; c += TPB) {
      float wr[K];
#pragma unroll
      for (int k = 0; k < K; ++k) wr[k] = w[k * N + c];
      float* dst = E + ((long long)b * L + l0) * N + c;
      for (int l = g; l < nl; l += groups) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) acc = fmaf(wr[k], xs[l * stride + k], acc);
        const float v = gelu_exact(acc);
        dst[(long long)l * N] = v;
        s += v;
        ss += v * v;
      }
    }
  }
  double ds = s, dss = ss;
  block_sum2(ds, dss, sh);
  if (threadIdx.x == 0) {
    part[2 * ((long long)b * ntile + tile)] = ds;
    part[2 * ((long long)b * ntile + tile) + 1] = dss;
  }
}

int launch_encoder(const float* wav, int B, int T, int L, const float* w, int N, int K, int stride, float* E,
                   double* part, hipStream_t s) {
  if (B <= 0 || L <= 0) return SEPR_EINVAL;
  if (K != 16 || stride < 1 || stride > 8 || N <= 0) return SEPR_EINVAL;
  const int ntile = encoder_tiles(L);
  hipLaunchKernelGGL((encoder_kernel<16>), dim3(ntile, B), dim3(TPB), 0, s, wav, T, L, w, N, stride, E, part, ntile);
  SEPR_CHECK_LAUNCH("encoder_kernel");
  return SEPR_OK;
}

// ---------------------------------------------------------------------------------------------------
// AudioDecoder: ConvTranspose1d(N -> 1, K taps, stride, no bias) (module.py:268-283) as
//   D[l][k] = sum_c O2[l][c] * wdec[k][c]   then   wav[tau] = sum_{l*stride + k = tau} D[l][k]
// A workgroup = 64 frame rows of one (utterance, speaker): 61 new frames + (K-1)/stride = 3 halo frames that
// are recomputed, one 16-frame sub-tile per wave.  D^T[tap][frame] runs on the f32 MFMA with both operands
// read straight from global memory as 16-byte fragments (the [16, N] tap matrix stays in L1/L2; every
// activation row is read once), then the overlap-add happens in LDS so each output sample is written exactly
// once (no atomics).  Output layout [S,B,Tout]: speaker-major, matching model.py:43-44's list of per-speaker
// [B,T] tensors.
// ---------------------------------------------------------------------------------------------------
constexpr int DEC_ROWS = 64, DEC_HB = 3, DEC_FT = DEC_ROWS - DEC_HB;

// Auxiliary heads (model.py:47-52): the stage output is nearest-upsampled to L frames BEFORE the per-frame
// OutputLayer, i.e. every source frame is pushed through both projections 2-16 times.  The projections are per-row
// maps, so they run once per SOURCE frame; the upsampling (idx), the ReLU mask and the product with the encoder
// frame (network.py:41) happen here, where the rows are read: x = max(O2[seq, idx[l]], 0) * enc[b, l].
template <int K>
__global__ __launch_bounds__(TPB) void decoder_kernel(const float* __restrict__ O2, int S, int B, int L, int N, int stride,
                                                     const float* __restrict__ wdec, float* __restrict__ wav, int Tout,
                                                     const int* __restrict__ idx, int Tsrc, const float* __restrict__ enc) {
  static_assert(K == 16, "one 16-tap MFMA tile");
  __shared__ __attribute__((aligned(16))) float Ds[DEC_ROWS * K];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fi = lane & 15, fg = lane >> 4;
  const int hb = (K - 1) / stride;                 // == DEC_HB (checked by the launcher)
  const int seq = blockIdx.y, l0 = blockIdx.x * DEC_FT;
  // tile row r <-> frame l0 - hb + r; this lane's frame
  const int l = l0 - hb + 16 * w + fi;
  const bool valid = (l >= 0 && l < L);
  const int lc = valid ? l : 0;
  const int src = idx ? idx[lc] : lc;
  const float* xrow = O2 + ((long long)seq * (idx ? Tsrc : L) + src) * N + 4 * fg;
  const float* erow = enc ? enc + ((long long)(seq / S) * L + lc) * N + 4 * fg : nullptr;
  const float* wrow = wdec + (long long)fi * N + 4 * fg;     // tap fi
  // Four K steps per trip: their 12 loads are issued back to back BEFORE the first product (a trip pays one memory latency, not four), and the
  // products alternate between two accumulators (the f32 MFMA's dependent-accumulator latency is 40 cycles against a 32-cycle issue).
  // (N % 64 == 0 is checked by the launcher; round 6: 168 -> see profiles/r06_*kernel_stats.csv per launch for the four auxiliary heads)
  f32x4 d0 = (f32x4){0.f, 0.f, 0.f, 0.f}, d1 = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int steps = N / 16;
#pragma unroll 1
  for (int cb = 0; cb < steps; cb += 4) {
    float4 x[4], wv[4], e[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      x[u] = ld4(xrow + 16 * (cb + u));
      wv[u] = ld4(wrow + 16 * (cb + u));
      e[u] = erow ? ld4(erow + 16 * (cb + u)) : make_float4(1.f, 1.f, 1.f, 1.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (erow) x[u] = make_float4(fmaxf(x[u].x, 0.f) * e[u].x, fmaxf(x[u].y, 0.f) * e[u].y, fmaxf(x[u].z, 0.f) * e[u].z, fmaxf(x[u].w, 0.f) * e[u].w);
      if (!valid) x[u] = zero4();
      d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u].x, x[u].x, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u].y, x[u].y, d1, 0, 0, 0);
      d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u].z, x[u].z, d0, 0, 0, 0);
      d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u].w, x[u].w, d1, 0, 0, 0);
    }
  }
  const f32x4 d = d0 + d1;
  // lane holds D[frame = 16w + fi][tap = 4fg + r]
  st4(Ds + (16 * w + fi) * K + 4 * fg, make_float4(d[0], d[1], d[2], d[3]));
  __syncthreads();
  const int s = seq % S, b = seq / S;
  float* dst = wav + ((long long)s * B + b) * Tout;
  for (int tl = tid; tl < DEC_FT * stride; tl += TPB) {
    const long long tau = (long long)l0 * stride + tl;
    if (tau >= Tout) continue;
    const int lr = tl / stride, ph = tl - lr * stride;
    float y = 0.f;
    for (int j = 0; j <= hb; ++j) {
      const int k = ph + j * stride;
      if (k < K) y += Ds[(lr - j + hb) * K + k];
    }
    dst[tau] = y;
  }
}

int launch_decoder(const float* O2, int nS, int S, int L, int N, int K, int stride, const float* wdec, float* wav,
                   int Tout, const int* idx, int Tsrc, const float* enc, hipStream_t s) {
  if (nS <= 0 || L <= 0) return SEPR_EINVAL;
  if (K != 16 || stride < 4 || stride > 16 || N > 4096 || N % 64 != 0 || S <= 0 || nS % S != 0) return SEPR_EINVAL;
  if ((K - 1) / stride > DEC_HB) return SEPR_EINVAL;
  const int hb = (K - 1) / stride;
  const int tiles = (L + hb + DEC_FT - 1) / DEC_FT;
  hipLaunchKernelGGL((decoder_kernel<16>), dim3(tiles, nS), dim3(TPB), 0, s, O2, S, nS / S, L, N, stride, wdec, wav, Tout,
                     idx, Tsrc, enc);
  SEPR_CHECK_LAUNCH("decoder_kernel");
  return SEPR_OK;
}

}  // namespace sepr
