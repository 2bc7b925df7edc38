// EGA self-attention over the pooled sequence with the relative-position key bias.
//
// reference: MultiHeadAttention.forward, modules/network.py:103-122 with pos_k from
// RelativePositionalEncoding, modules/module.py:52-57,196-198:
//     scores[i,j] = (q_i . k_j + q_i . pe_k[clamp(i - j, -maxlen, maxlen-1) + maxlen]) / sqrt(dk)
// The reference materialises pos_k as [T',T',dk] (16 MB at T'=500, 85 MB at 1150).  Here the bias is
// Toeplitz: a workgroup of 128 queries x a tile of 128 keys only ever needs the 255 consecutive rows
// i-j of the [2*maxlen, dk] table, staged in LDS (padded to dk+4 floats per row: conflict-free 16-byte
// reads for consecutive lanes).  Scores are never written: one query per lane, keys streamed with an
// exact online softmax (rescale only when the running max moves).  k_j / v_j are wave-uniform, so they
// come through the scalar cache, not LDS.  This path is <5 % of the model's FLOPs;
// it stays on the VALU.
#include "sepr_pointwise.h"

namespace sepr {

template <int DK>
__global__ __launch_bounds__(128) void relattn_kernel(const float* __restrict__ QKV, float* __restrict__ O, int Tp, int F,
                                                     const float* __restrict__ pe, int maxlen, float inv_sqrt_dk) {
  constexpr int QB = 128, KT = 128, PS = DK + 4, NBAND = QB + KT - 1;
  __shared__ __attribute__((aligned(16))) float pes[NBAND * PS];
  const int tid = threadIdx.x;
  const int i0 = blockIdx.x * QB, h = blockIdx.y, seq = blockIdx.z;
  const int ld = 3 * F;
  const int i = i0 + tid;
  const bool active = i < Tp;
  const float* base = QKV + (long long)seq * Tp * ld + h * DK;

  float q[DK], o[DK];
  {
    const float* qp = base + (long long)(active ? i : Tp - 1) * ld;
#pragma unroll
    for (int d = 0; d < DK; d += 4) {
      const float4 v = ld4(qp + d);
      q[d] = v.x * inv_sqrt_dk; q[d + 1] = v.y * inv_sqrt_dk; q[d + 2] = v.z * inv_sqrt_dk; q[d + 3] = v.w * inv_sqrt_dk;
    }
  }
#pragma unroll
  for (int d = 0; d < DK; ++d) o[d] = 0.f;
  float mrun = -INFINITY, lrun = 0.f;

  for (int j0 = 0; j0 < Tp; j0 += KT) {
    __syncthreads();  // previous tile's band fully consumed
    for (int idx = tid; idx < NBAND * (DK / 4); idx += 128) {
      const int rr = idx / (DK / 4), c4 = idx % (DK / 4);
      int rel = i0 - j0 - (KT - 1) + rr;
      rel = rel < -maxlen ? -maxlen : (rel > maxlen - 1 ? maxlen - 1 : rel);
      st4(pes + rr * PS + 4 * c4, ld4(pe + (long long)(rel + maxlen) * DK + 4 * c4));
    }
    __syncthreads();
    const int jn = (Tp - j0 < KT) ? Tp - j0 : KT;
    for (int jj = 0; jj < jn; ++jj) {
      const float* kp = base + (long long)(j0 + jj) * ld + F;  // wave-uniform
      const float* vp = kp + F;
      const float* pp = pes + (tid - jj + KT - 1) * PS;
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < DK; d += 4) {
        const float4 kk = ld4(kp + d);
        const float4 p4 = ld4(pp + d);
        s = fmaf(q[d], kk.x + p4.x, s);
        s = fmaf(q[d + 1], kk.y + p4.y, s);
        s = fmaf(q[d + 2], kk.z + p4.z, s);
        s = fmaf(q[d + 3], kk.w + p4.w, s);
      }
      if (s > mrun) {  // rare after the first few keys
        const float corr = __expf(mrun - s);
        lrun *= corr;
#pragma unroll
        for (int d = 0; d < DK; ++d) o[d] *= corr;
        mrun = s;
      }
      const float p = __expf(s - mrun);
      lrun += p;
#pragma unroll
      for (int d = 0; d < DK; d += 4) {
        const float4 vv = ld4(vp + d);
        o[d] = fmaf(p, vv.x, o[d]);
        o[d + 1] = fmaf(p, vv.y, o[d + 1]);
        o[d + 2] = fmaf(p, vv.z, o[d + 2]);
        o[d + 3] = fmaf(p, vv.w, o[d + 3]);
      }
    }
  }
  if (active) {
    const float inv = 1.0f / lrun;
    float* op = O + ((long long)seq * Tp + i) * F + h * DK;
#pragma unroll
    for (int d = 0; d < DK; d += 4) st4(op + d, make_float4(o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv));
  }
}

int launch_relattn(const float* QKV, float* O, int n, int Tp, int F, int H, const float* pe_k, int maxlen, hipStream_t s) {
  if (n <= 0 || Tp <= 0) return SEPR_OK;
  if (H <= 0 || F % H != 0 || maxlen <= 0 || !pe_k) return SEPR_EINVAL;
  const int dk = F / H;
  const dim3 grid((Tp + 127) / 128, H, n);
  const float isd = 1.0f / sqrtf((float)dk);
  if (dk == 16) {
    hipLaunchKernelGGL((relattn_kernel<16>), grid, dim3(128), 0, s, QKV, O, Tp, F, pe_k, maxlen, isd);
  } else if (dk == 32) {
    hipLaunchKernelGGL((relattn_kernel<32>), grid, dim3(128), 0, s, QKV, O, Tp, F, pe_k, maxlen, isd);
  } else {
    return SEPR_EINVAL;
  }
  SEPR_CHECK_LAUNCH("relattn_kernel");
  return SEPR_OK;
}

}  // namespace sepr
