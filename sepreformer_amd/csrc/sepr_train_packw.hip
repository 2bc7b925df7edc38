// Per-step weight re-pack of the training path on the device (round 5; include/sepr.h sepr_train_pack_lin / sepr_train_fold_bias).
//
// Every optimizer step changes every weight, so the training forward rebuilds the kernel-layout forms of all projections (train_pack.py):
// LayerNorm gamma / LayerScale folded in, the transposed form the input gradient needs, bf16 hi / lo planes in MFMA fragment order, and the
// bias with LayerNorm beta folded in (fp64).  Rounds 2-4 did that with batched torch ops - ~870 aten / rocclr launches and 4 ms per
// captured step (5 % of it), the one place torch arithmetic sat inside the timed training step.  Here a STACK of G same-shaped projections
// (all 56 GCFN up-projections, all 22 EGA q/k/v, ...) is one launch that reads the parameters where they live (device pointer tables) and
// writes the final layout: no stacked copies, no fp64 temporaries, no transposed temporaries.
#include "sepr_train.h"

namespace sepr {
namespace {
typedef __bf16 pw_bf16x8 __attribute__((ext_vector_type(8)));

struct PackLinArgs {
  const float* const* src;     // [G * panels] source matrices: panel p of block g holds rows [p * SN / panels, (p + 1) * SN / panels) of the [SN][SK] source
  const float* const* scale;   // [G] or null
  int G, SN, SK, panels;
  int scale_kind;              // 0 none, 1 per source column (LayerNorm gamma), 2 per source row (LayerScale)
  int transpose;               // out[n][k] = src[k][n]
  int planes;                  // 1: bf16 hi / lo fragments [N/16][K/32][2][64][8];  0: fp32 [N][K]
  void* out;                   // G x N x K x 4 bytes
};

// one wave = one 16 x 32 block of the OUTPUT matrix (lane = 16-row index i + 16 * (8-column group g4): the MFMA A / B fragment order)
__global__ __launch_bounds__(256) void pack_lin_kernel(const PackLinArgs a) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int g = blockIdx.y;
  const int N = a.transpose ? a.SK : a.SN, K = a.transpose ? a.SN : a.SK;
  const int kst = K / 32, nblk = (N / 16) * kst;
  const int blk = blockIdx.x * 4 + w;
  if (blk >= nblk) return;
  const int tile = blk / kst, ks = blk - tile * kst;
  const int i = lane & 15, g4 = lane >> 4;
  const int n = 16 * tile + i, k0 = 32 * ks + 8 * g4;
  const int prow = a.SN / a.panels;                         // source rows per panel
  const float* sc = a.scale ? a.scale[g] : nullptr;
  float v[8];
  if (!a.transpose) {                                       // source row n, columns k0 .. k0 + 7: 32 contiguous bytes
    const float* s = a.src[g * a.panels + n / prow] + (long long)(n % prow) * a.SK + k0;
    const float4 p = ld4(s), q = ld4(s + 4);
    v[0] = p.x; v[1] = p.y; v[2] = p.z; v[3] = p.w; v[4] = q.x; v[5] = q.y; v[6] = q.z; v[7] = q.w;
    if (a.scale_kind == 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[k0 + e];
    } else if (a.scale_kind == 2) {
      const float r = sc[n];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * r;
    }
  } else {                                                  // source rows k0 .. k0 + 7, column n (16 lanes read 64 contiguous bytes per row)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int sn = k0 + e;
      v[e] = a.src[g * a.panels + sn / prow][(long long)(sn % prow) * a.SK + n];
    }
    if (a.scale_kind == 1) {
      const float c = sc[n];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * c;
    } else if (a.scale_kind == 2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = v[e] * sc[k0 + e];
    }
  }
  // (the fold is ONE fp32 multiply per element: the fp64 product torch forms, (double)w * (double)s, is exact in 53 bits, so rounding it to
  //  fp32 is the correctly rounded fp32 product - bit-identical to the batched torch packer of rounds 2-4, which the device test checks)
  if (a.planes) {
    pw_bf16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const __bf16 hh = (__bf16)v[e];
      h[e] = hh;
      l[e] = (__bf16)(v[e] - (float)hh);
    }
    __bf16* o = static_cast<__bf16*>(a.out) + (long long)g * 2 * N * K + ((long long)(tile * kst + ks) * 2) * 512 + lane * 8;
    *reinterpret_cast<pw_bf16x8*>(o) = h;
    *reinterpret_cast<pw_bf16x8*>(o + 512) = l;
  } else {
    float* o = static_cast<float*>(a.out) + (long long)g * N * K + (long long)n * K + k0;
    st4(o, make_float4(v[0], v[1], v[2], v[3]));
    st4(o + 4, make_float4(v[4], v[5], v[6], v[7]));
  }
}

// out[g][n] = (float)((double)bias[g][n] + sum_k (double)W[g][n][k] * (double)beta[g][k]);  one wave per output element
__global__ __launch_bounds__(256) void fold_bias_kernel(const float* const* __restrict__ w, const float* const* __restrict__ bias,
                                                        const float* const* __restrict__ beta, int N, int K, int panels, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, g = blockIdx.y;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  const int prow = N / panels, p = n / prow, r = n - p * prow;
  double s = 0.0;
  if (beta && w) {
    const float* wr = w[g * panels + p] + (long long)r * K;
    const float* be = beta[g];
    for (int k = lane; k < K; k += 64) s += (double)wr[k] * (double)be[k];
    s = wave_sum_d(s);
  }
  if (lane == 0) out[(long long)g * N + n] = (float)((bias ? (double)bias[g * panels + p][r] : 0.0) + s);
}

// The fused GCFN kernel's weight forms (sepr_gcfn_fused.h GcfnFusedArgs w1p / w2p; pack.py pack_gcfn_fused) for all G blocks: grid (chunk, block).
//   w1p  per chunk: [4 tiles v0 v1 g0 g1][KS][plane][64][8] bf16 of W1 * gamma, then 4 KB of constants [2 tile pairs][10][16] fp32 =
//        b1 + W1 . beta (fp64), the value taps, the gate taps and gate bias scaled by -log2(e) (fp64 products), zero padding
//   w2p  per chunk: [F/16][plane][64][8] bf16 of W2 in the down-projection's row / k-slot order
struct PackGcfnArgs {
  const float* const* w1; const float* const* b1; const float* const* ln_g; const float* const* ln_b;
  const float* const* w2; const float* const* dw_w; const float* const* dw_b;
  int F;
  unsigned char* w1p; unsigned char* w2p;
};
__global__ __launch_bounds__(256) void pack_gcfn_fused_kernel(const PackGcfnArgs a) {
#pragma clang fp contract(off)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = blockIdx.x, g = blockIdx.y;
  const int F = a.F, H3 = 3 * F, KS = F / 32, FT = F / 16, nch = H3 / 32;
  const int i = lane & 15, g4 = lane >> 4;
  const float* W1 = a.w1[g];
  const float* gam = a.ln_g[g];
  const size_t w1_chunk = (size_t)4 * KS * 2 * 1024 + 4096;
  unsigned char* o1 = a.w1p + ((size_t)g * nch + c) * w1_chunk;
  // ---- up-projection fragments ----
  for (int b = w; b < 4 * KS; b += 4) {
    const int t = b / KS, ks = b - t * KS;
    const int n = (t < 2 ? 0 : H3) + 32 * c + 16 * (t & 1) + i, k0 = 32 * ks + 8 * g4;
    const float* s = W1 + (long long)n * F + k0;
    const float4 p = ld4(s), q = ld4(s + 4), gp = ld4(gam + k0), gq = ld4(gam + k0 + 4);
    const float v[8] = {p.x * gp.x, p.y * gp.y, p.z * gp.z, p.w * gp.w, q.x * gq.x, q.y * gq.y, q.z * gq.z, q.w * gq.w};
    pw_bf16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const __bf16 hh = (__bf16)v[e];
      h[e] = hh;
      l[e] = (__bf16)(v[e] - (float)hh);
    }
    unsigned char* o = o1 + ((size_t)(t * KS + ks) * 2) * 1024 + lane * 16;
    *reinterpret_cast<pw_bf16x8*>(o) = h;
    *reinterpret_cast<pw_bf16x8*>(o + 1024) = l;
  }
  // ---- constants ----
  float* cst = reinterpret_cast<float*>(o1 + (size_t)4 * KS * 2 * 1024);
  for (int e = 320 + tid; e < 1024; e += 256) cst[e] = 0.f;
  {
    const float* b1 = a.b1[g];
    const float* be = a.ln_b[g];
    for (int u = w; u < 64; u += 4) {                       // folded biases: u = (j, value | gate, i)
      const int j = u >> 5, kind = (u >> 4) & 1, ii = u & 15;
      const int n = (kind ? H3 : 0) + 32 * c + 16 * j + ii;
      double sum = 0.0;
      for (int k = lane; k < F; k += 64) sum += (double)W1[(long long)n * F + k] * (double)be[k];
      sum = wave_sum_d(sum);
      if (lane == 0) cst[j * 160 + kind * 16 + ii] = (float)((double)b1[n] + sum);
    }
    const float* dw = a.dw_w[g];
    const float* db = a.dw_b[g];
    if (tid < 256) {                                         // taps (q = 2..7) and conv biases (q = 8, 9): 2 x 8 x 16 values
      const int j = tid >> 7, q = 2 + ((tid >> 4) & 7), ii = tid & 15;
      const int v = 32 * c + 16 * j + ii;
      double val;
      if (q < 5) val = (double)dw[(long long)v * 3 + (q - 2)];
      else if (q < 8) val = (double)dw[(long long)(H3 + v) * 3 + (q - 5)] * -1.4426950408889634;
      else if (q == 8) val = (double)db[v];
      else val = (double)db[H3 + v] * -1.4426950408889634;
      cst[j * 160 + q * 16 + ii] = (float)val;
    }
  }
  // ---- down-projection fragments: row 16 ft + (4 q + r) -> output channel 32 (ft / 2) + 8 q + 4 (ft % 2) + r; k slot 8 g4 + e -> hidden
  //      channel 32 c + (e < 4 ? 4 g4 + e : 16 + 4 g4 + e - 4) ----
  const float* W2 = a.w2[g];
  unsigned char* o2 = a.w2p + ((size_t)g * nch + c) * FT * 2 * 1024;
  for (int ft = w; ft < FT; ft += 4) {
    const int row = 32 * (ft >> 1) + 8 * (i >> 2) + 4 * (ft & 1) + (i & 3);
    const float* s = W2 + (long long)row * H3 + 32 * c + 4 * g4;
    const float4 p = ld4(s), q = ld4(s + 16);
    const float v[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
    pw_bf16x8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const __bf16 hh = (__bf16)v[e];
      h[e] = hh;
      l[e] = (__bf16)(v[e] - (float)hh);
    }
    unsigned char* o = o2 + ((size_t)ft * 2) * 1024 + lane * 16;
    *reinterpret_cast<pw_bf16x8*>(o) = h;
    *reinterpret_cast<pw_bf16x8*>(o + 1024) = l;
  }
}
}  // namespace
}  // namespace sepr

using namespace sepr;

extern "C" int sepr_train_pack_gcfn_fused(const void* const* w1, const void* const* b1, const void* const* ln_g, const void* const* ln_b,
                                          const void* const* w2, const void* const* dw_w, const void* const* dw_b, int G, int F, void* w1p,
                                          void* w2p, sepr_stream_t stream) {
  if (!w1 || !b1 || !ln_g || !ln_b || !w2 || !dw_w || !dw_b || !w1p || !w2p || G <= 0 || (F != 64 && F != 128)) return SEPR_EINVAL;
  PackGcfnArgs a;
  a.w1 = reinterpret_cast<const float* const*>(w1); a.b1 = reinterpret_cast<const float* const*>(b1);
  a.ln_g = reinterpret_cast<const float* const*>(ln_g); a.ln_b = reinterpret_cast<const float* const*>(ln_b);
  a.w2 = reinterpret_cast<const float* const*>(w2); a.dw_w = reinterpret_cast<const float* const*>(dw_w);
  a.dw_b = reinterpret_cast<const float* const*>(dw_b);
  a.F = F; a.w1p = static_cast<unsigned char*>(w1p); a.w2p = static_cast<unsigned char*>(w2p);
  hipLaunchKernelGGL(pack_gcfn_fused_kernel, dim3(3 * F / 32, G), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  SEPR_CHECK_LAUNCH("pack_gcfn_fused_kernel");
  return SEPR_OK;
}

extern "C" int sepr_train_pack_lin(const void* const* src, const void* const* scale, int G, int SN, int SK, int panels, int scale_kind, int transpose,
                                   int planes, void* out, sepr_stream_t stream) {
  if (!src || !out || G <= 0 || SN <= 0 || SK <= 0 || panels <= 0 || SN % panels) return SEPR_EINVAL;
  if (scale_kind < 0 || scale_kind > 2 || (scale_kind != 0 && !scale)) return SEPR_EINVAL;
  const int N = transpose ? SK : SN, K = transpose ? SN : SK;
  if (N % 16 || K % 32) return SEPR_EINVAL;                // whole MFMA fragments (every projection of the path: multiples of 64)
  PackLinArgs a;
  a.src = reinterpret_cast<const float* const*>(src);
  a.scale = scale_kind ? reinterpret_cast<const float* const*>(scale) : nullptr;
  a.G = G; a.SN = SN; a.SK = SK; a.panels = panels; a.scale_kind = scale_kind; a.transpose = transpose ? 1 : 0; a.planes = planes ? 1 : 0;
  a.out = out;
  const int nblk = (N / 16) * (K / 32);
  hipLaunchKernelGGL(pack_lin_kernel, dim3((nblk + 3) / 4, G), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  SEPR_CHECK_LAUNCH("pack_lin_kernel");
  return SEPR_OK;
}

extern "C" int sepr_train_fold_bias(const void* const* w, const void* const* bias, const void* const* beta, int G, int N, int K, int panels, float* out,
                                    sepr_stream_t stream) {
  if (!out || G <= 0 || N <= 0 || K <= 0 || panels <= 0 || N % panels) return SEPR_EINVAL;
  if ((beta != nullptr) != (w != nullptr)) return SEPR_EINVAL;     // the fold needs both; neither = a plain gather of the biases
  if (!bias && !beta) return SEPR_EINVAL;
  hipLaunchKernelGGL(fold_bias_kernel, dim3((N + 3) / 4, G), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const float* const*>(w), reinterpret_cast<const float* const*>(bias),
                     reinterpret_cast<const float* const*>(beta), N, K, panels, out);
  SEPR_CHECK_LAUNCH("fold_bias_kernel");
  return SEPR_OK;
}
