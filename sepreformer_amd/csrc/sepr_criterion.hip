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
#include "sepr_gemm_epi.h"

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
  for (int i = 0; i < NV; ++i) {
    // sum(x^2) - sum(x)^2/n of a constant (DC-only) signal can cancel to a slightly negative number; the
    // reference subtracts the mean first, so its energies are >= 0: clamp, or sqrt() would return NaN
    const double e = q[NV + i] - q[i] * q[i] / n;
    zz[i] = e > 0.0 ? e : 0.0;
  }
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

// ---------------------------------------------------------------------------------------------------------------------
// PIT_SISNR_mag (criterions.py:117-176): the same permutation walk on STFT magnitudes
//   loss(est_s, tgt_k) = -20 log10(eps + |M_k'| / (|M_s - M_k'| + eps)),  M = sqrt(re^2 + im^2 + 1e-10) of the conv-STFT
//   (:74-113, kernel :43-61) of the zero-mean signal, tgt scaled by clamp(<e~,t~> / (|t~|^2 + eps), min = 1e-2) (:155-159),
//   norms over (bins, frames) (:164).
// Four launches: the moment kernel above (means and the S x S scales), stft_prep_kernel (zero-mean, zero-padded copies of
// the 2S waveforms), the STFT itself as ONE f32-MFMA projection (row f of the A operand is x[hop*f : hop*f + N]: an
// overlapping row map with leading dimension = hop, the DFT kernel is the weight) and stft_pair_kernel (fp64 sums of
// M_k'^2 and (M_s - M_k')^2 per (utterance, estimate, target), then the permutation walk).
// ---------------------------------------------------------------------------------------------------------------------
namespace sepr {
int launch_gemm(int pro, int epi, const GemmArgs& a, int site, hipStream_t stream);   // sepr_gemm.hip
namespace {

template <int S>
__global__ __launch_bounds__(64) void mag_scales_kernel(const double* __restrict__ part, int T, int nchunk, double eps,
                                                       float* __restrict__ means, float* __restrict__ scales, int B) {
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
#pragma unroll
  for (int i = 0; i < 2 * S; ++i) means[i * B + b] = (float)(q[i] / n);       // waveform order: est_0.. est_{S-1}, tgt_0..
#pragma unroll
  for (int s = 0; s < S; ++s)
#pragma unroll
    for (int k = 0; k < S; ++k) {
      const double et = q[2 * NV + s * S + k] - q[s] * q[S + k] / n;
      double tt = q[NV + S + k] - q[S + k] * q[S + k] / n;
      tt = tt > 0.0 ? tt : 0.0;                        // cancellation on a DC-only target (see pit_finalize_kernel)
      const double sc = et / (tt + eps);
      scales[(b * S + s) * S + k] = (float)(sc < 1e-2 ? 1e-2 : sc);             // torch.clamp(scale, min=1e-2)
    }
}

__global__ __launch_bounds__(256) void stft_prep_kernel(const float* __restrict__ est, const float* __restrict__ tgt,
                                                       const float* __restrict__ means, int SB, int T, int Tpad,
                                                       float* __restrict__ xz) {
  const int wv = blockIdx.y;                                     // 0 .. 2*S*B - 1
  const float* src = (wv < SB ? est + (long long)wv * T : tgt + (long long)(wv - SB) * T);
  const float mu = means[wv];
  for (int t = blockIdx.x * 256 + threadIdx.x; t < Tpad; t += gridDim.x * 256)
    xz[(long long)wv * Tpad + t] = t < T ? src[t] - mu : 0.f;
}

template <int S>
__global__ __launch_bounds__(256) void stft_pair_kernel(const float* __restrict__ C, int ldc, int B, int nfr, int nbin,
                                                       const float* __restrict__ scales, double* __restrict__ sums) {
  // block = (utterance b, estimate s, target k): sums[0] = sum M_k'^2, sums[1] = sum (M_s - M_k')^2
  const int b = blockIdx.x, s = blockIdx.y / S, k = blockIdx.y % S;
  const float sc = scales[(b * S + s) * S + k];
  const float* Ce = C + (long long)(s * B + b) * nfr * ldc;
  const float* Ct = C + (long long)((S + k) * B + b) * nfr * ldc;
  double a0 = 0.0, a1 = 0.0;
  const int total = nfr * nbin;
  for (int i = threadIdx.x; i < total; i += 256) {
    const int f = i / nbin, w = i - f * nbin;
    const float er = Ce[(long long)f * ldc + w], ei = Ce[(long long)f * ldc + nbin + w];
    const float tr = sc * Ct[(long long)f * ldc + w], ti = sc * Ct[(long long)f * ldc + nbin + w];
    const float me = sqrtf(er * er + ei * ei + 1.0e-10f);
    const float ms = sqrtf(tr * tr + ti * ti + 1.0e-10f);
    const double d = (double)me - (double)ms;
    a0 = fma((double)ms, (double)ms, a0);
    a1 = fma(d, d, a1);
  }
  __shared__ double red[2][4];
  a0 = wave_sum_d(a0);
  a1 = wave_sum_d(a1);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { red[0][w] = a0; red[1][w] = a1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* o = sums + ((long long)(b * S + s) * S + k) * 2;
    o[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    o[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

template <int S>
__global__ __launch_bounds__(64) void mag_finalize_kernel(const double* __restrict__ sums, int B, double eps, float* __restrict__ loss,
                                                         int* __restrict__ perm) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  double l[S][S];
#pragma unroll
  for (int s = 0; s < S; ++s)
#pragma unroll
    for (int k = 0; k < S; ++k) {
      const double* o = sums + ((long long)(b * S + s) * S + k) * 2;
      l[s][k] = -20.0 * log10(eps + sqrt(o[0]) / (sqrt(o[1]) + eps));
    }
  constexpr int NPERM = (S == 1) ? 1 : (S == 2 ? 2 : 6);
  const int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
  const int perms2[2][3] = {{0, 1, 0}, {1, 0, 0}};
  double best = 0.0;
  int arg = 0;
  for (int p = 0; p < NPERM; ++p) {
    double sl = 0.0;
#pragma unroll
    for (int s = 0; s < S; ++s) sl += l[s][(S == 2) ? perms2[p][s] : perms[p][s]];
    if (p == 0 || sl < best) { best = sl; arg = p; }
  }
  loss[b] = (float)best;
  if (perm) {
#pragma unroll
    for (int s = 0; s < S; ++s) perm[b * S + s] = (S == 2) ? perms2[arg][s] : perms[arg][s];
  }
}

struct MagPlan {
  int nchunk, NF, Tpad, nfr, ldc;
  size_t o_part, o_means, o_scales, o_xz, o_c, o_sums, total;
};
MagPlan mag_plan(int S, int B, int T, int N, int hop) {
  MagPlan p;
  p.nchunk = (T + PIT_CHUNK - 1) / PIT_CHUNK;
  p.NF = (T + hop - 1) / hop;
  p.Tpad = p.NF * hop;
  p.nfr = p.Tpad >= N ? (p.Tpad - N) / hop + 1 : 0;
  p.ldc = ((N + 2) + 3) / 4 * 4;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return o; };
  p.o_part = take((size_t)B * p.nchunk * pit_nq(S) * sizeof(double));
  p.o_means = take((size_t)2 * S * B * sizeof(float));
  p.o_scales = take((size_t)B * S * S * sizeof(float));
  p.o_xz = take((size_t)2 * S * B * p.Tpad * sizeof(float));
  p.o_c = take((size_t)2 * S * B * p.nfr * p.ldc * sizeof(float));
  p.o_sums = take((size_t)B * S * S * 2 * sizeof(double));
  p.total = off;
  return p;
}
}  // namespace

size_t pit_mag_workspace_bytes(int S, int B, int T, int N, int hop) {
  if (S < 1 || S > PIT_SMAX || B <= 0 || T <= 0 || N <= 0 || hop <= 0 || N % 32 != 0 || hop % 4 != 0) return 0;
  return mag_plan(S, B, T, N, hop).total;
}
}  // namespace sepr

extern "C" size_t sepr_pit_sisnr_mag_workspace(int S, int B, int T, int frame_len, int frame_shift) {
  return sepr::pit_mag_workspace_bytes(S, B, T, frame_len, frame_shift);
}

extern "C" int sepr_pit_sisnr_mag_fwd(const float* est, const float* tgt, int S, int B, int T, const float* dft, int frame_len,
                                      int frame_shift, double eps, float* loss, int* perm, void* ws, size_t ws_bytes,
                                      sepr_stream_t stream) {
  using namespace sepr;
  if (!est || !tgt || !dft || !loss || S < 1 || S > PIT_SMAX || B <= 0 || T <= 0 || B > 65535) return SEPR_EINVAL;
  if (frame_len <= 0 || frame_len % 32 != 0 || frame_shift <= 0 || frame_shift % 4 != 0) return SEPR_EINVAL;
  const MagPlan p = mag_plan(S, B, T, frame_len, frame_shift);
  if (p.nfr <= 0) return SEPR_EINVAL;                                 // shorter than one frame
  if (!ws || ws_bytes < p.total) return SEPR_EWORKSPACE;
  if ((long long)2 * S * B * p.nfr > 0x7fffffffLL / 8 || (long long)2 * S * B * p.Tpad >= (1LL << 32)) return SEPR_EINVAL;
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* base = static_cast<char*>(ws);
  double* part = reinterpret_cast<double*>(base + p.o_part);
  float* means = reinterpret_cast<float*>(base + p.o_means);
  float* scales = reinterpret_cast<float*>(base + p.o_scales);
  float* xz = reinterpret_cast<float*>(base + p.o_xz);
  float* C = reinterpret_cast<float*>(base + p.o_c);
  double* sums = reinterpret_cast<double*>(base + p.o_sums);
  const int nbin = frame_len / 2 + 1;
#define SEPR_MAG_CASE(SS)                                                                                                    \
  case SS:                                                                                                                   \
    hipLaunchKernelGGL((pit_partial_kernel<SS>), dim3(p.nchunk, B), dim3(PIT_TPB), 0, st, est, tgt, (const float*)nullptr,  \
                       B, T, p.nchunk, part);                                                                                \
    hipLaunchKernelGGL((mag_scales_kernel<SS>), dim3(B), dim3(64), 0, st, part, T, p.nchunk, eps, means, scales, B);        \
    break;
  switch (S) {
    SEPR_MAG_CASE(1)
    SEPR_MAG_CASE(2)
    SEPR_MAG_CASE(3)
    default: return SEPR_EINVAL;
  }
#undef SEPR_MAG_CASE
  {
    const int gx = (p.Tpad + 255) / 256 < 64 ? (p.Tpad + 255) / 256 : 64;
    hipLaunchKernelGGL(stft_prep_kernel, dim3(gx, 2 * S * B), dim3(256), 0, st, est, tgt, means, S * B, T, p.Tpad, xz);
  }
  SEPR_CHECK_LAUNCH("pit mag: moments / prep");
  {  // STFT: row (waveform, frame f) of A = xz[waveform][hop*f : hop*f + N]; weight = DFT kernel [N+2 (padded to ldc), N]
    GemmArgs a = gemm_args_zero();
    a.M = 2 * S * B * p.nfr; a.N = p.ldc; a.K = frame_len;
    a.A = xz; a.lda = frame_shift;
    a.rows_out = p.nfr; a.rows_src = p.NF; a.rows_valid = p.nfr;
    a.W = dft; a.bias = nullptr; a.Y = C; a.ldc = p.ldc;
    SEPR_TRY(launch_gemm(PRO_PLAIN, EPI_STORE, a, SEPR_SITE_NONE, st));
  }
#define SEPR_MAG_CASE2(SS)                                                                                                   \
  case SS:                                                                                                                   \
    hipLaunchKernelGGL((stft_pair_kernel<SS>), dim3(B, SS * SS), dim3(256), 0, st, C, p.ldc, B, p.nfr, nbin, scales, sums); \
    hipLaunchKernelGGL((mag_finalize_kernel<SS>), dim3((B + 63) / 64), dim3(64), 0, st, sums, B, eps, loss, perm);          \
    break;
  switch (S) {
    SEPR_MAG_CASE2(1)
    SEPR_MAG_CASE2(2)
    SEPR_MAG_CASE2(3)
    default: return SEPR_EINVAL;
  }
#undef SEPR_MAG_CASE2
  SEPR_CHECK_LAUNCH("pit mag: pair sums");
  return SEPR_OK;
}
