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

// =====================================================================================================================
// Backward of the two training criteria (reference: torch.autograd through criterions.py:148-217)
// =====================================================================================================================
// PIT_SISNR_time.  For estimate a and its permuted target t (zero-mean versions a~, t~):
//   alpha = <a~,t~> / c, c = |t~|^2 + eps;  e = a~ - alpha t~;  P = |alpha| |t~|;  R = |e|;  u = P / (R + eps)
//   loss = clamp(-20 log10(eps + u), min = clamp_min)
//   d loss / d a = g_u [ sign(alpha) |t~| / (c (R + eps)) t~  -  P / ((R + eps)^2 R) (e - (e.t~)/c t~) ],  g_u = -(20 / ln 10) / (eps + u)
// (a linear combination of the zero-mean vectors a~ and t~, so removing the mean is already accounted for); zero where
// the clamp is active.  The moments come from the same one-pass kernel as the forward; the gradient is then one
// element-wise pass: dest = ca (a - mean_a) + ct (t - mean_t).
namespace sepr {
namespace {
template <int S>
__global__ __launch_bounds__(64) void pit_bwd_coef_kernel(const double* __restrict__ part, const int* __restrict__ perm,
                                                         const float* __restrict__ gl, int T, int nchunk, double eps, double clamp_min,
                                                         float* __restrict__ coef /*[B][S][4]: ca, ct, mean_a, mean_t*/) {
  constexpr int NV = 2 * S + 1, NQ = pit_nq(S);
  const int b = blockIdx.x;
  __shared__ double q[NQ];
  if (threadIdx.x < NQ) {
    double s = 0.0;
    for (int c = 0; c < nchunk; ++c) s += part[((long long)b * nchunk + c) * NQ + threadIdx.x];
    q[threadIdx.x] = s;
  }
  __syncthreads();
  if (threadIdx.x >= S) return;
  const int s = threadIdx.x, k = perm[b * S + s];
  const double n = (double)T;
  double aa = q[NV + s] - q[s] * q[s] / n, tt = q[NV + S + k] - q[S + k] * q[S + k] / n;
  aa = aa > 0.0 ? aa : 0.0;
  tt = tt > 0.0 ? tt : 0.0;
  const double at = q[2 * NV + s * S + k] - q[s] * q[S + k] / n;
  const double c = tt + eps, alpha = at / c;
  const double P = fabs(alpha) * sqrt(tt);
  double R2 = aa - 2.0 * alpha * at + alpha * alpha * tt;
  R2 = R2 > 0.0 ? R2 : 0.0;
  const double R = sqrt(R2), u = P / (R + eps);
  const double loss = -20.0 * log10(eps + u);
  double ca = 0.0, ct = 0.0;
  if (loss > clamp_min && R > 0.0) {
    const double gu = -(20.0 / log(10.0)) / (eps + u) * (double)gl[b];
    const double et = at - alpha * tt;                                   // e . t~
    const double kt = gu * (alpha >= 0.0 ? 1.0 : -1.0) * sqrt(tt) / (c * (R + eps));   // coefficient of t~ from dP
    const double ke = -gu * P / ((R + eps) * (R + eps) * R);             // coefficient of (e - (e.t~)/c t~) from dR
    ca = ke;                                                             // e = a~ - alpha t~
    ct = kt - ke * alpha - ke * et / c;
  }
  float* o = coef + ((long long)b * S + s) * 4;
  o[0] = (float)ca; o[1] = (float)ct; o[2] = (float)(q[s] / n); o[3] = (float)(q[S + k] / n);
}
__global__ __launch_bounds__(256) void pit_bwd_apply_kernel(const float* __restrict__ est, const float* __restrict__ tgt,
                                                           const int* __restrict__ perm, const float* __restrict__ coef, int S, int B, int T,
                                                           float* __restrict__ dest) {
  const int sb = blockIdx.y;                 // s * B + b
  const int s = sb / B, b = sb - s * B;
  const float* cf = coef + ((long long)b * S + s) * 4;
  const float ca = cf[0], ct = cf[1], ma = cf[2], mt = cf[3];
  const float* a = est + (long long)sb * T;
  const float* t = tgt + ((long long)perm[b * S + s] * B + b) * T;
  float* o = dest + (long long)sb * T;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < T; i += gridDim.x * 256) o[i] = fmaf(ca, a[i] - ma, ct * (t[i] - mt));
}
}  // namespace
}  // namespace sepr

extern "C" int sepr_pit_sisnr_bwd(const float* est, const float* tgt, const int* perm, const float* gl, int S, int B, int T, double eps,
                                  double clamp_min, float* dest, void* ws, size_t ws_bytes, sepr_stream_t stream) {
  using namespace sepr;
  if (!est || !tgt || !perm || !gl || !dest || S < 1 || S > PIT_SMAX || B <= 0 || T <= 0 || B > 65535) return SEPR_EINVAL;
  const size_t need = pit_workspace_bytes(S, B, T) + align_up((size_t)B * S * 4 * sizeof(float));
  if (!ws || ws_bytes < need) return SEPR_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  double* part = static_cast<double*>(ws);
  float* coef = reinterpret_cast<float*>(static_cast<char*>(ws) + pit_workspace_bytes(S, B, T));
  const int nchunk = (T + PIT_CHUNK - 1) / PIT_CHUNK;
#define SEPR_PITB(SS)                                                                                                             \
  case SS:                                                                                                                        \
    hipLaunchKernelGGL((pit_partial_kernel<SS>), dim3(nchunk, B), dim3(PIT_TPB), 0, st, est, tgt, (const float*)nullptr, B, T,   \
                       nchunk, part);                                                                                             \
    hipLaunchKernelGGL((pit_bwd_coef_kernel<SS>), dim3(B), dim3(64), 0, st, part, perm, gl, T, nchunk, eps, clamp_min, coef);    \
    break;
  switch (S) {
    SEPR_PITB(1)
    SEPR_PITB(2)
    SEPR_PITB(3)
    default: return SEPR_EINVAL;
  }
#undef SEPR_PITB
  const int gx = (T + 255) / 256 < 64 ? (T + 255) / 256 : 64;
  hipLaunchKernelGGL(pit_bwd_apply_kernel, dim3(gx, S * B), dim3(256), 0, st, est, tgt, perm, coef, S, B, T, dest);
  SEPR_CHECK_LAUNCH("pit_sisnr_bwd kernels");
  return SEPR_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// PIT_SISNR_mag backward.  With M_e = |STFT(a~)|, M_s = |STFT(c t~)| (both sqrt(re^2 + im^2 + 1e-10)), c = max(<a~,t~> / (|t~|^2 + eps), 1e-2),
// P = |M_s|_F, R = |M_e - M_s|_F, u = P / (R + eps), loss = -20 log10(eps + u):
//   dL/dM_e = g_u (-P / ((R + eps)^2 R)) (M_e - M_s);   dL/dM_s = g_u [ M_s / (P (R + eps)) + P / ((R + eps)^2 R) (M_e - M_s) ]
//   d re_e = dL/dM_e re_e / M_e (same for im);  dL/dc = sum dL/dM_s c (re_t^2 + im_t^2) / M_s;  dc/da~ = t~ / (|t~|^2 + eps) unless clamped
//   d a~ = STFT^T(d re_e, d im_e) + dL/dc dc/da~;   d a = d a~ - mean(d a~)
// The STFT adjoint is the same projection core with the transposed DFT kernel followed by an overlap-add gather.
// ---------------------------------------------------------------------------------------------------------------------
namespace sepr {
namespace {
struct MagBwdPlan {
  MagPlan f;
  int ldd;      // leading dimension of dC: frame_len + 2 rounded up to a multiple of 32 (K of the adjoint projection)
  size_t o_dc, o_pc, o_fr, o_da, o_coef, total;
};
MagBwdPlan mag_bwd_plan(int S, int B, int T, int N, int hop) {
  MagBwdPlan p;
  p.f = mag_plan(S, B, T, N, hop);
  p.ldd = ((N + 2) + 31) / 32 * 32;
  size_t off = p.f.total;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return o; };
  p.o_dc = take((size_t)S * B * p.f.nfr * p.ldd * sizeof(float));      // d(re, im) of the estimates' STFT
  p.o_pc = take((size_t)S * B * 2 * sizeof(double));                      // dL/dc partial (one per (s,b)), spare
  p.o_fr = take((size_t)S * B * p.f.nfr * N * sizeof(float));             // adjoint frames before overlap-add
  p.o_da = take((size_t)S * B * p.f.Tpad * sizeof(float));                // d a~ (padded length)
  p.o_coef = take((size_t)S * B * 4 * sizeof(float));
  p.total = off;
  return p;
}

// block = (b, s): pass 1 sums are already in `sums` (P^2, R^2 for every (b, s, k)); this kernel writes dC for the
// estimate s of utterance b against its permuted target and reduces dL/dc
template <int S>
__global__ __launch_bounds__(256) void stft_pair_bwd_kernel(const float* __restrict__ C, int ldc, int B, int nfr, int nbin,
                                                           const float* __restrict__ scales, const double* __restrict__ sums,
                                                           const int* __restrict__ perm, const float* __restrict__ gl, double eps,
                                                           float* __restrict__ dC, int ldd, double* __restrict__ dLdc) {
  const int b = blockIdx.x, s = blockIdx.y;
  const int k = perm[b * S + s];
  const float sc = scales[(b * S + s) * S + k];
  const double* o = sums + ((long long)(b * S + s) * S + k) * 2;
  const double P = sqrt(o[0]), R = sqrt(o[1]);
  const double u = P / (R + eps);
  const double gu = -(20.0 / log(10.0)) / (eps + u) * (double)gl[b];
  const double kR = (R > 0.0) ? P / ((R + eps) * (R + eps) * R) : 0.0;
  const float ge = (float)(-gu * kR);                       // dL/dM_e = ge (M_e - M_s)
  const float gs1 = (float)(P > 0.0 ? gu / (P * (R + eps)) : 0.0);   // dL/dM_s = gs1 M_s + gs2 (M_e - M_s)
  const float gs2 = (float)(gu * kR);
  const float* Ce = C + (long long)(s * B + b) * nfr * ldc;
  const float* Ct = C + (long long)((S + k) * B + b) * nfr * ldc;
  float* De = dC + (long long)(s * B + b) * nfr * ldd;
  double acc = 0.0;
  const int total = nfr * nbin;
  for (int i = threadIdx.x; i < total; i += 256) {
    const int f = i / nbin, w = i - f * nbin;
    const float er = Ce[(long long)f * ldc + w], ei = Ce[(long long)f * ldc + nbin + w];
    const float tr = Ct[(long long)f * ldc + w], ti = Ct[(long long)f * ldc + nbin + w];
    const float me = sqrtf(er * er + ei * ei + 1.0e-10f);
    const float t2 = tr * tr + ti * ti;
    const float ms = sqrtf(sc * sc * t2 + 1.0e-10f);
    const float d = me - ms;
    const float gme = ge * d;
    De[(long long)f * ldd + w] = gme * er / me;
    De[(long long)f * ldd + nbin + w] = gme * ei / me;
    const float gms = gs1 * ms + gs2 * d;
    acc += (double)(gms * sc * t2 / ms);
  }
  // zero the padding columns of this estimate's rows (the adjoint projection reads ldc columns)
  for (int i = threadIdx.x; i < nfr * (ldd - 2 * nbin); i += 256) {
    const int f = i / (ldd - 2 * nbin), w = i - f * (ldd - 2 * nbin);
    De[(long long)f * ldd + 2 * nbin + w] = 0.f;
  }
  __shared__ double red[4];
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) dLdc[b * S + s] = (red[0] + red[1]) + (red[2] + red[3]);
}

// overlap-add gather of the adjoint frames + the scale path: da[t] = sum_{f: 0 <= t - hop f < N} fr[f][t - hop f] + cs t~[t]
template <int S>
__global__ __launch_bounds__(256) void stft_ola_kernel(const float* __restrict__ fr, const float* __restrict__ xz, int B, int nfr, int N,
                                                      int hop, int Tpad, const double* __restrict__ dLdc, const double* __restrict__ part,
                                                      int nchunk, int T, const int* __restrict__ perm, double eps,
                                                      float* __restrict__ da) {
  constexpr int NV = 2 * S + 1, NQ = pit_nq(S);
  const int sb = blockIdx.y;
  const int s = sb / B, b = sb - s * B;
  const int k = perm[b * S + s];
  // scale path coefficient: dL/dc * [raw scale >= 1e-2] / (|t~|^2 + eps)
  __shared__ float cs_s;
  if (threadIdx.x == 0) {
    double q[NQ];
    for (int i = 0; i < NQ; ++i) {
      double a = 0.0;
      for (int c = 0; c < nchunk; ++c) a += part[((long long)b * nchunk + c) * NQ + i];
      q[i] = a;
    }
    const double n = (double)T;
    double tt = q[NV + S + k] - q[S + k] * q[S + k] / n;
    tt = tt > 0.0 ? tt : 0.0;
    const double et = q[2 * NV + s * S + k] - q[s] * q[S + k] / n;
    const double raw = et / (tt + eps);
    cs_s = raw >= 1e-2 ? (float)(dLdc[b * S + s] / (tt + eps)) : 0.f;
  }
  __syncthreads();
  const float cs = cs_s;
  const float* f0 = fr + (long long)sb * nfr * N;
  const float* tz = xz + (long long)((S + k) * B + b) * Tpad;
  for (int t = blockIdx.x * 256 + threadIdx.x; t < Tpad; t += gridDim.x * 256) {
    float a = cs * tz[t];
    const int fhi = t / hop < nfr - 1 ? t / hop : nfr - 1;
    for (int f = fhi; f >= 0 && t - hop * f < N; --f) a += f0[(long long)f * N + (t - hop * f)];
    da[(long long)sb * Tpad + t] = a;
  }
}
// dest = da[:T] - mean(da[:T])   (a~ = a - mean(a)); one block per (s, b)
__global__ __launch_bounds__(256) void demean_kernel(const float* __restrict__ da, int Tpad, int T, float* __restrict__ dest) {
  const int sb = blockIdx.x;
  const float* p = da + (long long)sb * Tpad;
  double s = 0.0;
  for (int i = threadIdx.x; i < T; i += 256) s += (double)p[i];
  __shared__ double red[4];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const float mean = (float)(((red[0] + red[1]) + (red[2] + red[3])) / (double)T);
  float* o = dest + (long long)sb * T;
  for (int i = threadIdx.x; i < T; i += 256) o[i] = p[i] - mean;
}
}  // namespace
}  // namespace sepr

extern "C" size_t sepr_pit_sisnr_mag_bwd_workspace(int S, int B, int T, int frame_len, int frame_shift) {
  using namespace sepr;
  if (S < 1 || S > PIT_SMAX || B <= 0 || T <= 0 || frame_len <= 0 || frame_shift <= 0 || frame_len % 32 || frame_shift % 4) return 0;
  return mag_bwd_plan(S, B, T, frame_len, frame_shift).total;
}

extern "C" int sepr_pit_sisnr_mag_bwd(const float* est, const float* tgt, const int* perm, const float* gl, int S, int B, int T,
                                      const float* dft, const float* dft_t, int frame_len, int frame_shift, double eps, float* dest,
                                      void* ws, size_t ws_bytes, sepr_stream_t stream) {
  using namespace sepr;
  if (!est || !tgt || !perm || !gl || !dft || !dft_t || !dest || S < 1 || S > PIT_SMAX || B <= 0 || T <= 0 || B > 65535) return SEPR_EINVAL;
  if (frame_len <= 0 || frame_len % 32 != 0 || frame_shift <= 0 || frame_shift % 4 != 0) return SEPR_EINVAL;
  const MagBwdPlan bp = mag_bwd_plan(S, B, T, frame_len, frame_shift);
  const MagPlan& p = bp.f;
  if (p.nfr <= 0) return SEPR_EINVAL;
  if (!ws || ws_bytes < bp.total) return SEPR_EWORKSPACE;
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* base = static_cast<char*>(ws);
  double* part = reinterpret_cast<double*>(base + p.o_part);
  float* means = reinterpret_cast<float*>(base + p.o_means);
  float* scales = reinterpret_cast<float*>(base + p.o_scales);
  float* xz = reinterpret_cast<float*>(base + p.o_xz);
  float* C = reinterpret_cast<float*>(base + p.o_c);
  double* sums = reinterpret_cast<double*>(base + p.o_sums);
  float* dC = reinterpret_cast<float*>(base + bp.o_dc);
  double* dLdc = reinterpret_cast<double*>(base + bp.o_pc);
  float* fr = reinterpret_cast<float*>(base + bp.o_fr);
  float* da = reinterpret_cast<float*>(base + bp.o_da);
  const int nbin = frame_len / 2 + 1;
  // ---- recompute the forward quantities (moments, scales, zero-mean padded waveforms, STFT of all 2S waveforms, pair sums)
#define SEPR_MAGB1(SS)                                                                                                              \
  case SS:                                                                                                                          \
    hipLaunchKernelGGL((pit_partial_kernel<SS>), dim3(p.nchunk, B), dim3(PIT_TPB), 0, st, est, tgt, (const float*)nullptr, B, T,    \
                       p.nchunk, part);                                                                                             \
    hipLaunchKernelGGL((mag_scales_kernel<SS>), dim3(B), dim3(64), 0, st, part, T, p.nchunk, eps, means, scales, B);               \
    break;
  switch (S) {
    SEPR_MAGB1(1)
    SEPR_MAGB1(2)
    SEPR_MAGB1(3)
    default: return SEPR_EINVAL;
  }
#undef SEPR_MAGB1
  {
    const int gx = (p.Tpad + 255) / 256 < 64 ? (p.Tpad + 255) / 256 : 64;
    hipLaunchKernelGGL(stft_prep_kernel, dim3(gx, 2 * S * B), dim3(256), 0, st, est, tgt, means, S * B, T, p.Tpad, xz);
  }
  SEPR_CHECK_LAUNCH("pit mag bwd: moments / prep");
  {
    GemmArgs a = gemm_args_zero();
    a.M = 2 * S * B * p.nfr; a.N = p.ldc; a.K = frame_len;
    a.A = xz; a.lda = frame_shift;
    a.rows_out = p.nfr; a.rows_src = p.NF; a.rows_valid = p.nfr;
    a.W = dft; a.bias = nullptr; a.Y = C; a.ldc = p.ldc;
    SEPR_TRY(launch_gemm(PRO_PLAIN, EPI_STORE, a, SEPR_SITE_NONE, st));
  }
#define SEPR_MAGB2(SS)                                                                                                              \
  case SS:                                                                                                                          \
    hipLaunchKernelGGL((stft_pair_kernel<SS>), dim3(B, SS * SS), dim3(256), 0, st, C, p.ldc, B, p.nfr, nbin, scales, sums);        \
    hipLaunchKernelGGL((stft_pair_bwd_kernel<SS>), dim3(B, SS), dim3(256), 0, st, C, p.ldc, B, p.nfr, nbin, scales, sums, perm, gl, \
                       eps, dC, bp.ldd, dLdc);                                                                                              \
    break;
  switch (S) {
    SEPR_MAGB2(1)
    SEPR_MAGB2(2)
    SEPR_MAGB2(3)
    default: return SEPR_EINVAL;
  }
#undef SEPR_MAGB2
  SEPR_CHECK_LAUNCH("pit mag bwd: pair");
  {  // adjoint frames: fr[row][n] = sum_c dC[row][c] K[c][n]  ->  projection with the transposed kernel dft_t [frame_len][ldd] (zero padded)
    GemmArgs a = gemm_args_zero();
    a.M = S * B * p.nfr; a.N = frame_len; a.K = bp.ldd;
    a.A = dC; a.lda = bp.ldd; a.W = dft_t; a.bias = nullptr; a.Y = fr; a.ldc = frame_len;
    SEPR_TRY(launch_gemm(PRO_PLAIN, EPI_STORE, a, SEPR_SITE_NONE, st));
  }
  {
    const int gx = (p.Tpad + 255) / 256 < 64 ? (p.Tpad + 255) / 256 : 64;
#define SEPR_MAGB3(SS)                                                                                                              \
  case SS:                                                                                                                          \
    hipLaunchKernelGGL((stft_ola_kernel<SS>), dim3(gx, SS * B), dim3(256), 0, st, fr, xz, B, p.nfr, frame_len, frame_shift, p.Tpad, \
                       dLdc, part, p.nchunk, T, perm, eps, da);                                                                     \
    break;
    switch (S) {
      SEPR_MAGB3(1)
      SEPR_MAGB3(2)
      SEPR_MAGB3(3)
      default: return SEPR_EINVAL;
    }
#undef SEPR_MAGB3
  }
  hipLaunchKernelGGL(demean_kernel, dim3(S * B), dim3(256), 0, st, da, p.Tpad, T, dest);
  SEPR_CHECK_LAUNCH("pit mag bwd: adjoint");
  return SEPR_OK;
}
