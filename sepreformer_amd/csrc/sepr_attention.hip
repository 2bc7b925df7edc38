// EGA self-attention over the pooled sequence with the relative-position key bias.
//
// reference: MultiHeadAttention.forward, modules/network.py:103-122 with pos_k from
// RelativePositionalEncoding, modules/module.py:52-57,196-198:
//     scores[i,j] = (q_i . k_j + q_i . pe_k[clamp(i - j, -maxlen, maxlen-1) + maxlen]) / sqrt(dk)
// The reference materialises pos_k as [T',T',dk] (16 MB at T'=500, 85 MB at 1150) and the scores as
// [b,H,T',T'].  Here neither exists:
//   * a workgroup = 64 queries of one (sequence, head) = 4 waves x 16 queries; keys are streamed in
//     64-key LDS tiles (K rows, V transposed, and the 127 consecutive rows i-j of the [2*maxlen, dk]
//     table that the tile can touch - the bias is Toeplitz);
//   * q.k^T and p.v run on the f32 MFMA (v_mfma_f32_16x16x4_f32) in the transposed form
//     S^T[key][query]: a lane then owns ONE query column and 4 keys per 16-key sub-tile, so the softmax
//     statistics are per-lane scalars (two shuffles across the 4 lane groups), the probabilities are
//     already the B operand of the p.v MFMA (no cross-lane movement), and the running output
//     O^T[d][query] is rescaled by a per-lane factor;
//   * the relative-position term is a per-lane VALU dot product of the lane's own q row with 4
//     consecutive band rows (the skewed index i-j does not map onto an MFMA fragment);
//   * exact online softmax (fp32, rescale every sub-tile), masked keys contribute exp(-1e30 - m) = 0.
#include "sepr_pointwise.h"

namespace sepr {

template <int DK>
__global__ __launch_bounds__(256) void relattn_kernel(const float* __restrict__ QKV, float* __restrict__ O, int Tp, int F,
                                                     const float* __restrict__ pe, int maxlen, float inv_sqrt_dk) {
  constexpr int QB = 64, KT = 64;
  constexpr int KS = DK + 4;          // K tile row stride   [KT][KS]
  constexpr int VS = KT + 4;          // V^T row stride      [DK][VS]
  constexpr int PS = DK + 4;          // band row stride     [QB + KT - 1][PS]
  constexpr int NBAND = QB + KT - 1;
  constexpr int NC = DK / 16;         // 16-wide d slices
  __shared__ __attribute__((aligned(16))) float Ks[KT * KS];
  __shared__ __attribute__((aligned(16))) float Vts[DK * VS];
  __shared__ __attribute__((aligned(16))) float pes[NBAND * PS];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ii = lane & 15, g = lane >> 4;
  const int i0 = blockIdx.x * QB, h = blockIdx.y, seq = blockIdx.z;
  const int ld = 3 * F;
  const float* base = QKV + (long long)seq * Tp * ld + h * DK;
  const int i = i0 + 16 * w + ii;
  const bool active = i < Tp;

  // this lane's query row: full (for the VALU bias) and the MFMA B fragments d = 16c + 4g + r
  float qfull[DK];
  float4 qf[NC];
  {
    const float* qp = base + (long long)(active ? i : Tp - 1) * ld;
#pragma unroll
    for (int d = 0; d < DK; d += 4) {
      const float4 v = ld4(qp + d);
      qfull[d] = v.x * inv_sqrt_dk; qfull[d + 1] = v.y * inv_sqrt_dk;
      qfull[d + 2] = v.z * inv_sqrt_dk; qfull[d + 3] = v.w * inv_sqrt_dk;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float4 v = ld4(qp + 16 * c + 4 * g);
      qf[c] = make_float4(v.x * inv_sqrt_dk, v.y * inv_sqrt_dk, v.z * inv_sqrt_dk, v.w * inv_sqrt_dk);
    }
  }
  f32x4 o[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) o[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float mrun = -1e30f, lrun = 0.f;

  for (int j0 = 0; j0 < Tp; j0 += KT) {
    __syncthreads();   // previous tile fully consumed
    // ---- stage K rows, V transposed, and the band of the position table --------------------------------
    for (int idx = tid; idx < KT * (DK / 4); idx += 256) {
      const int jj = idx / (DK / 4), c4 = idx % (DK / 4);
      const int j = j0 + jj;
      float4 kv = zero4(), vv = zero4();
      if (j < Tp) {
        const float* kp = base + (long long)j * ld + F + 4 * c4;
        kv = ld4(kp);
        vv = ld4(kp + F);
      }
      st4(Ks + jj * KS + 4 * c4, kv);
      Vts[(4 * c4 + 0) * VS + jj] = vv.x;
      Vts[(4 * c4 + 1) * VS + jj] = vv.y;
      Vts[(4 * c4 + 2) * VS + jj] = vv.z;
      Vts[(4 * c4 + 3) * VS + jj] = vv.w;
    }
    for (int idx = tid; idx < NBAND * (DK / 4); idx += 256) {
      const int rr = idx / (DK / 4), c4 = idx % (DK / 4);
      int rel = i0 - j0 - (KT - 1) + rr;                      // i - j for band row rr
      rel = rel < -maxlen ? -maxlen : (rel > maxlen - 1 ? maxlen - 1 : rel);
      st4(pes + rr * PS + 4 * c4, ld4(pe + (long long)(rel + maxlen) * DK + 4 * c4));
    }
    __syncthreads();

    const int nsub = (Tp - j0 >= KT) ? KT / 16 : (Tp - j0 + 15) / 16;
    for (int jt = 0; jt < nsub; ++jt) {
      // S^T[key = 16 jt + 4g + r][query = ii]
      f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const float4 kf = ld4(Ks + (16 * jt + ii) * KS + 16 * c + 4 * g);
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[c].x, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[c].y, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[c].z, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[c].w, s, 0, 0, 0);
      }
      // relative-position bias: band row of (query 16w+ii, key 16jt+4g+r) is (16w+ii) - (16jt+4g+r) + KT-1
      const float* pb = pes + (16 * w + ii - 16 * jt - 4 * g + (KT - 1)) * PS;
      float bias[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* pr = pb - r * PS;
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < DK; d += 4) {
          const float4 p4 = ld4(pr + d);
          acc = fmaf(qfull[d], p4.x, acc);
          acc = fmaf(qfull[d + 1], p4.y, acc);
          acc = fmaf(qfull[d + 2], p4.z, acc);
          acc = fmaf(qfull[d + 3], p4.w, acc);
        }
        bias[r] = acc;
      }
      const int jbase = j0 + 16 * jt + 4 * g;
      float sv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) sv[r] = (jbase + r < Tp) ? s[r] + bias[r] : -1e30f;
      float mx = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mnew = fmaxf(mrun, mx);
      const float corr = __expf(mrun - mnew);
      float p[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) p[r] = __expf(sv[r] - mnew);
      lrun = lrun * corr + ((p[0] + p[1]) + (p[2] + p[3]));
      mrun = mnew;
      // O^T[d = 16c + 4g + r][query = ii] += V^T[d][key] . P[key][query]
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        o[c][0] *= corr; o[c][1] *= corr; o[c][2] *= corr; o[c][3] *= corr;
        const float4 vf = ld4(Vts + (16 * c + ii) * VS + 16 * jt + 4 * g);
        o[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.x, p[0], o[c], 0, 0, 0);
        o[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.y, p[1], o[c], 0, 0, 0);
        o[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.z, p[2], o[c], 0, 0, 0);
        o[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.w, p[3], o[c], 0, 0, 0);
      }
    }
  }
  float ltot = lrun + __shfl_xor(lrun, 16, 64);
  ltot += __shfl_xor(ltot, 32, 64);
  if (active) {
    const float inv = 1.0f / ltot;
    float* op = O + ((long long)seq * Tp + i) * F + h * DK + 4 * g;
#pragma unroll
    for (int c = 0; c < NC; ++c)
      st4(op + 16 * c, make_float4(o[c][0] * inv, o[c][1] * inv, o[c][2] * inv, o[c][3] * inv));
  }
}

int launch_relattn(const float* QKV, float* O, int n, int Tp, int F, int H, const float* pe_k, int maxlen, hipStream_t s) {
  if (n <= 0 || Tp <= 0) return SEPR_OK;
  if (H <= 0 || F % H != 0 || maxlen <= 0 || !pe_k || n > 65535) return SEPR_EINVAL;
  const int dk = F / H;
  const dim3 grid((Tp + 63) / 64, H, n);
  const float isd = 1.0f / sqrtf((float)dk);
  if (dk == 16) {
    hipLaunchKernelGGL((relattn_kernel<16>), grid, dim3(256), 0, s, QKV, O, Tp, F, pe_k, maxlen, isd);
  } else if (dk == 32) {
    hipLaunchKernelGGL((relattn_kernel<32>), grid, dim3(256), 0, s, QKV, O, Tp, F, pe_k, maxlen, isd);
  } else {
    return SEPR_EINVAL;
  }
  SEPR_CHECK_LAUNCH("relattn_kernel");
  return SEPR_OK;
}

}  // namespace sepr
