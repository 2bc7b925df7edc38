// Permutation-invariant SI-SNR on the device.
//
// reference: utils/implements/criterions.py
//   PIT_SISNR_time.__call__  :191-217   loss  = mean_b min_perm sum_s clamp(-SISNR(est_s, tgt_perm(s)), min=-30)
//   PIT_SISNRi.__call__      :232-260   score = max_perm sum_s (SISNR(est_s, tgt_p(s)) - SISNR(mix, tgt_p(s)))
//   SISNR(a, t) = 20 log10(eps + |alpha t~| / (|a~ - alpha t~| + eps)),  alpha = <a~,t~> / (|t~|^2 + eps),  ~ = zero-mean
//
// Everything above is a function of 2S+1 sums, 2S+1 sums of squares and S*S + S cross sums per utterance, so the
// waveforms are read exactly once (20 B per sample for S = 2; HBM-bound, ~3 us for 32 x 32000 samples):
//   pit_partial_kernel   one workgroup per (4096-sample chunk, utterance): fp64 accumulators, wave butterfly +
//                        LDS combine, fixed order -> bitwise deterministic; partials to the workspace;
//   pit_finalize_kernel  one wave per utterance: sums the chunk partials in chunk order, forms the zero-mean
//                        moments, the S x S SI-SNR matrix and the mixture row, walks the S! permutations.
// fp64 throughout: the residual energy |a~|^2 - 2 alpha <a~,t~> + alpha^2 |t~|^2 cancels ~3 digits at 30 dB, which
// fp32 moments could not afford; with fp64 moments the result is the exact-arithmetic value to ~1e-9 dB (the
// reference's own fp32 evaluation differs from that by up to ~1e-3 dB at high SI-SNR).
#include "sepr_common.h"

namespace sepr {

namespace {
constexpr int PIT_TPB = 256, PIT_CHUNK = 4096, PIT_SMAX = 3;
__host__ __device__ constexpr int pit_nq(int S) { return 2 * (2 * S + 1) + S * S + S; }

template <int S>
__global__ __launch_bounds__(PIT_TPB) void pit_partial_kernel(const float* __restrict__ est, const float* __restrict__ tgt,
                                                             const float* __restrict__ mix, int B, int T, int nchunk,
                                                             double* __restrict__ part) {
  constexpr int NV = 2 * S + 1, NQ = pit_nq(S);
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int t0 = chunk * PIT_CHUNK, t1 = min(T, t0 + PIT_CHUNK);
  double q[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) q[i] = 0.0;
  for (int t = t0 + threadIdx.x; t < t1; t += PIT_TPB) {
    double v[NV];                       // est_0..est_{S-1}, tgt_0..tgt_{S-1}, mix
#pragma unroll
    for (int s = 0; s < S; ++s) {
      v[s] = (double)est[((long long)s * B + b) * T + t];
      v[S + s] = (double)tgt[((long long)s * B + b) * T + t];
    }
    v[2 * S] = mix ? (double)mix[(long long)b * T + t] : 0.0;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      q[i] += v[i];
      q[NV + i] = fma(v[i], v[i], q[NV + i]);
    }
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int k = 0; k < S; ++k) q[2 * NV + s * S + k] = fma(v[s], v[S + k], q[2 * NV + s * S + k]);
#pragma unroll
    for (int k = 0; k < S; ++k) q[2 * NV + S * S + k] = fma(v[2 * S], v[S + k], q[2 * NV + S * S + k]);
  }
  __shared__ double red[PIT_TPB / 64][NQ];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const double s = wave_sum_d(q[i]);
    if (lane == 0) red[w][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < NQ) {
    double s = 0.0;
#pragma unroll
    for (int ww = 0; ww < PIT_TPB / 64; ++ww) s += red[ww][threadIdx.x];
    part[((long long)b * nchunk + chunk) * NQ + threadIdx.x] = s;
  }
}

__device__ __forceinline__ double sisnr_db(double aa, double tt, double at, double eps) {
  // aa = |a~|^2, tt = |t~|^2, at = <a~,t~>
  const double alpha = at / (tt + eps);
  const double num = fabs(alpha) * sqrt(tt);
  double res2 = aa - 2.0 * alpha * at + alpha * alpha * tt;
  res2 = res2 > 0.0 ? res2 : 0.0;
  return 20.0 * log10(eps + num / (sqrt(res2) + eps));
}

template <int S>
__global__ __launch_bounds__(64) void pit_finalize_kernel(const double* __restrict__ part, int T, int nchunk, double eps_loss,
                                                         double eps_i, double clamp_min, int have_mix, float* __restrict__ loss,
                                                         int* __restrict__ loss_perm, float* __restrict__ sisnri,
                                                         int* __restrict__ sisnri_perm) {
  constexpr int NV = 2 * S + 1, NQ = pit_nq(S);
  const int b = blockIdx.x;
  __shared__ double q[NQ];
  if (threadIdx.x < NQ) {
    double s = 0.0;
    for (int c = 0; c < nchunk; ++c) s += part[((long long)b * nchunk + c) * NQ + threadIdx.x];
    q[threadIdx.x] = s;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  const double n = (double)T;
  double zz[NV];                                        // zero-mean energies
#pragma unroll
  for (int i = 0; i < NV; ++i) zz[i] = q[NV + i] - q[i] * q[i] / n;
  double snr_l[S][S], snr_i[S][S], snr_x[S];
#pragma unroll
  for (int k = 0; k < S; ++k) {
    const double xt = q[2 * NV + S * S + k] - q[2 * S] * q[S + k] / n;
    snr_x[k] = have_mix ? sisnr_db(zz[2 * S], zz[S + k], xt, eps_i) : 0.0;
  }
#pragma unroll
  for (int s = 0; s < S; ++s)
#pragma unroll
    for (int k = 0; k < S; ++k) {
      const double et = q[2 * NV + s * S + k] - q[s] * q[S + k] / n;
      const double l = -sisnr_db(zz[s], zz[S + k], et, eps_loss);
      snr_l[s][k] = l < clamp_min ? clamp_min : l;      // torch.clamp(utt_loss, min=-30)
      snr_i[s][k] = sisnr_db(zz[s], zz[S + k], et, eps_i) - snr_x[k];
    }
  // permutations in itertools.permutations(range(S)) order; first optimum wins (torch.min / torch.max semantics)
  constexpr int NPERM = (S == 1) ? 1 : (S == 2 ? 2 : 6);
  const int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
  const int perms2[2][3] = {{0, 1, 0}, {1, 0, 0}};
  double best_l = 0.0, best_i = 0.0;
  int arg_l = 0, arg_i = 0;
  for (int p = 0; p < NPERM; ++p) {
    double sl = 0.0, si = 0.0;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int k = (S == 2) ? perms2[p][s] : perms[p][s];
      sl += snr_l[s][k];
      si += snr_i[s][k];
    }
    if (p == 0 || sl < best_l) { best_l = sl; arg_l = p; }
    if (p == 0 || si > best_i) { best_i = si; arg_i = p; }
  }
  loss[b] = (float)best_l;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int kl = (S == 2) ? perms2[arg_l][s] : perms[arg_l][s];
    const int ki = (S == 2) ? perms2[arg_i][s] : perms[arg_i][s];
    if (loss_perm) loss_perm[b * S + s] = kl;
    if (sisnri) sisnri[b * S + s] = (float)snr_i[s][ki];
    if (sisnri_perm) sisnri_perm[b * S + s] = ki;
  }
}
}  // namespace

size_t pit_workspace_bytes(int S, int B, int T) {
  if (S < 1 || S > PIT_SMAX || B <= 0 || T <= 0) return 0;
  const int nchunk = (T + PIT_CHUNK - 1) / PIT_CHUNK;
  return align_up((size_t)B * nchunk * pit_nq(S) * sizeof(double));
}

}  // namespace sepr

extern "C" int sepr_pit_sisnr_fwd(const float* est, const float* tgt, const float* mix, int S, int B, int T, double eps_loss,
                                  double eps_i, double clamp_min, float* loss, int* loss_perm, float* sisnri,
                                  int* sisnri_perm, void* ws, size_t ws_bytes, sepr_stream_t stream) {
  using namespace sepr;
  if (!est || !tgt || !loss || S < 1 || S > PIT_SMAX || B <= 0 || T <= 0 || B > 65535) return SEPR_EINVAL;
  if ((sisnri || sisnri_perm) && !mix) return SEPR_EINVAL;
  const size_t need = pit_workspace_bytes(S, B, T);
  if (!ws || ws_bytes < need) return SEPR_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  double* part = static_cast<double*>(ws);
  const int nchunk = (T + PIT_CHUNK - 1) / PIT_CHUNK;
  const dim3 grid(nchunk, B);
  const int have_mix = mix ? 1 : 0;
#define SEPR_PIT_CASE(SS)                                                                                                   \
  case SS:                                                                                                                  \
    hipLaunchKernelGGL((pit_partial_kernel<SS>), grid, dim3(PIT_TPB), 0, st, est, tgt, mix, B, T, nchunk, part);           \
    hipLaunchKernelGGL((pit_finalize_kernel<SS>), dim3(B), dim3(64), 0, st, part, T, nchunk, eps_loss, eps_i, clamp_min,   \
                       have_mix, loss, loss_perm, sisnri, sisnri_perm);                                                     \
    break;
  switch (S) {
    SEPR_PIT_CASE(1)
    SEPR_PIT_CASE(2)
    SEPR_PIT_CASE(3)
    default: return SEPR_EINVAL;
  }
#undef SEPR_PIT_CASE
  SEPR_CHECK_LAUNCH("pit_sisnr kernels");
  return SEPR_OK;
}
