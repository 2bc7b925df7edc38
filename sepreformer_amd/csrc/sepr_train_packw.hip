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
}  // namespace
}  // namespace sepr

using namespace sepr;

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
