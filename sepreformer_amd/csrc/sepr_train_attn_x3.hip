// EGA attention backward on the bf16 MFMA (split fp32 operands, like the forward kernel relattn_x3_kernel of
// sepr_attention.hip whose TRAIN instantiation is this path's forward): flash-style, the [Tp, Tp] probabilities are never
// stored - the forward keeps one log-sum-exp per query row and both backward kernels recompute the scores of the tile
// they are working on (6 + 9 MFMAs per 16 x 32 block, against 2 x 4 bytes of HBM traffic per score for stored P and dS).
//
//   scores[i][j] = (q_i . k_j + q_i . pe[r(i,j)]) / sqrt(dk),  P = exp(scores - lse_i),  Pd = dropout(P)
//   D_i = dO_i . o_i;  dP = dropout'(dO_i . v_j);  dS = P (dP - D_i) / sqrt(dk)
//   dq_i = sum_j dS[i][j] (k_j + pe[r]);  dk_j = sum_i dS[i][j] q_i;  dv_j = sum_i Pd[i][j] dO_i;  dpe[r] += sum dS[i][j] q_i
//
// An MFMA contracts over the index that lives INSIDE a lane, never over the 16 lanes of a fragment row, so the two
// reductions need the two orientations of the score tile:
//   * relattn_bwd_q_x3_kernel: lane = query (the forward's orientation, S^T[key][query]).  dQ^T += K^T dS^T contracts over
//     keys = a lane's 8 values: the dS a lane holds ARE the B fragment, as P is for PV in the forward.  The relative-position
//     part of dq un-skews dS into band coordinates through a per-query LDS row (slot b = query - key + 31) and contracts it
//     with the transposed band of the table.  Also writes D_i and the dS tile rows for the table gradient (below).
//   * relattn_bwd_kv_x3_kernel: lane = key.  S[query][key] = Q k^T, the bias as R[query][b] = Q band^T skewed through LDS,
//     dK^T += Q^T dS and dV^T += dO^T Pd contract over queries.
//   * dpe: the diagonal sums over (sequence, head, query) have no GEMM form; relattn_band_kernel reads the dS rows once
//     (coalesced along the diagonal direction) on the VALU and relattn_band_reduce2_kernel folds the partial bands into the
//     table gradient, fixed order, no atomics.
// Dropout of the probabilities: element (row = (seq*H + h)*Tp + i, key j) keeps iff the 16-bit half j & 1 of
// sepr_drop_word(key(site 2), row, j >> 1) >= thr - the same draw in all three kernels.
#include "sepr_train.h"

namespace sepr {

typedef __bf16 ax_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 ax_bf16x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int AX_QB = 64, AX_KT = 64, AX_NBAND = AX_QB + AX_KT - 1;
constexpr int AX_PSK = 52;      // bias skew scratch row stride (floats; 48 used)
constexpr int AX_PSD = 68;      // dS un-skew scratch row stride (floats; 64 used)
constexpr int AX_BTS = 152;     // transposed band row stride (bf16): band rows 0..126 + zero pad up to column 143

__device__ __forceinline__ void ax_split4(const float4 v, ax_bf16x4& h, ax_bf16x4& l) {
  const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const __bf16 hh = (__bf16)x[e];
    h[e] = hh;
    l[e] = (__bf16)(x[e] - (float)hh);
  }
}
__device__ __forceinline__ void ax_split8(const float (&x)[8], ax_bf16x8& h, ax_bf16x8& l) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 hh = (__bf16)x[e];
    h[e] = hh;
    l[e] = (__bf16)(x[e] - (float)hh);
  }
}

// ONE (the plain-bf16 training precision, sepr_lin.planes == 1): operands rounded to bf16 once, ONE MFMA per product - no lo planes
// in LDS (the backward kernels then fit three workgroups per CU instead of two), no lo halves to split.
template <bool ONE>
__device__ __forceinline__ f32x4 ax_mma(const ax_bf16x8 ah, const ax_bf16x8 al, const ax_bf16x8 bh, const ax_bf16x8 bl, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
  if constexpr (!ONE) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
  }
  return c;
}

struct AxArgs {
  const float* QKV;    // [n, Tp, 3F]
  const float* O;      // [n, Tp, F]   forward output (D_i)
  const float* dO;     // [n, Tp, F]
  const float* lse;    // [n*H, Tp]
  float* dQKV;         // [n, Tp, 3F]
  float* Dbuf;         // [n*H, Tp]    D_i = dO_i . o_i (written by the query-major kernel, read by the key-major one)
  float* dS;           // [n*H, Tp, Tp] (query-major kernel -> table-gradient kernel)
  const float* pe;     // [2*maxlen, dk]
  int Tp, F, H, maxlen;
  float isd;
  unsigned thr;
  float dscale;
  unsigned long long seed;
  const unsigned long long* salt;
  int one;             // 1: the ONE instantiations (plain-bf16 precision)
  int ds16;            // 1 (plain-bf16 precision, round 4): the dS rows travel to the table-gradient kernel as bf16 - that kernel is a pure
                       // HBM stream over [n*H, Tp, Tp] (512 MB in fp32 at 64 sequences), the only consumer rounds to bf16-level accuracy anyway
};

// ---------------------------------------------------------------------------------------------------------------------
// query-major: D, dS rows, dQ
// ---------------------------------------------------------------------------------------------------------------------
template <int DK, bool ONE>
__global__ __launch_bounds__(256) void relattn_bwd_q_x3_kernel(const AxArgs a) {
  static_assert(DK == 16 || DK == 32, "head width");
  constexpr int QB = AX_QB, KT = AX_KT, NBAND = AX_NBAND;
  constexpr int KSB = DK + 8, VSB = KT + 8, OT = DK / 16;
  constexpr int NU = KT * (DK / 4) / 256, NBU = (127 * (DK / 4) + 255) / 256;
  constexpr int LO = ONE ? 0 : 1;                                                  // (the lo planes exist in the bf16x3 form only)
  __shared__ __attribute__((aligned(16))) __bf16 Kh[KT * KSB], Kl_[LO * KT * KSB + 8];       // K rows (scores)
  __shared__ __attribute__((aligned(16))) __bf16 Vh[KT * KSB], Vl_[LO * KT * KSB + 8];       // V rows (dP)
  __shared__ __attribute__((aligned(16))) __bf16 Kth[DK * VSB], Ktl_[LO * DK * VSB + 8];     // K^T (dQ)
  __shared__ __attribute__((aligned(16))) __bf16 Bh[NBAND * KSB], Bl_[LO * NBAND * KSB + 8]; // band rows (bias)
  __shared__ __attribute__((aligned(16))) __bf16 Bth[DK * AX_BTS], Btl_[LO * DK * AX_BTS + 8];   // band^T (dQ, position part)
  // ONE: every "lo" name aliases its hi plane - the reads below stay in bounds and are dead (ax_mma<true> ignores them)
  __bf16* const Kl = ONE ? Kh : Kl_;
  __bf16* const Vl = ONE ? Vh : Vl_;
  __bf16* const Ktl = ONE ? Kth : Ktl_;
  __bf16* const Bl = ONE ? Bh : Bl_;
  __bf16* const Btl = ONE ? Bth : Btl_;
  __shared__ __attribute__((aligned(16))) float Psk[4 * 16 * AX_PSK];
  __shared__ __attribute__((aligned(16))) float Psd[4 * 16 * AX_PSD];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ii = lane & 15, g = lane >> 4;
  const int Tp = a.Tp, F = a.F;
  const int i0 = blockIdx.x * QB, h = blockIdx.y, seq = blockIdx.z;
  const int ld = 3 * F;
  const long long nh = (long long)seq * a.H + h;
  const float* base = a.QKV + (long long)seq * Tp * ld + h * DK;
  const int i = i0 + 16 * w + ii;
  const bool active = i < Tp;
  const bool lowk = DK == 32 || g < 2;
  const int gk = DK == 32 ? g : (g & 1);
  const int go = DK == 32 ? 8 * g : 8 * (g < 2 ? g : 2);
  DropKey dkey = {0u, 0u};
  if (a.thr) dkey = sepr_drop_key(a.seed, a.salt, 2u);
  for (int r = tid; r < KT; r += 256) {
#pragma unroll
    for (int e = DK; e < KSB; ++e) {
      Kh[r * KSB + e] = Vh[r * KSB + e] = (__bf16)0.f;
      if constexpr (!ONE) Kl[r * KSB + e] = Vl[r * KSB + e] = (__bf16)0.f;
    }
  }
  for (int r = tid; r < NBAND; r += 256) {
#pragma unroll
    for (int e = DK; e < KSB; ++e) {
      Bh[r * KSB + e] = (__bf16)0.f;
      if constexpr (!ONE) Bl[r * KSB + e] = (__bf16)0.f;
    }
  }
  for (int e = tid; e < DK * AX_BTS; e += 256) {
    Bth[e] = (__bf16)0.f;
    if constexpr (!ONE) Btl[e] = (__bf16)0.f;
  }

  // B fragments of this lane's query: q (scaled) and dO; D_i; lse_i
  ax_bf16x8 qh, ql, gh, gl;
  float Di = 0.f;
  {
    const long long row = (long long)seq * Tp + (active ? i : Tp - 1);
    const float* qp = base + (long long)(active ? i : Tp - 1) * ld + 8 * gk;
    const float* gp = a.dO + row * F + h * DK + 8 * gk;
    const float* op = a.O + row * F + h * DK + 8 * gk;
    const float4 q0 = ld4(qp), q1 = ld4(qp + 4), g0 = ld4(gp), g1 = ld4(gp + 4), o0 = ld4(op), o1 = ld4(op + 4);
    float xq[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
    float xg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    const float xo[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (lowk) Di = fmaf(xg[e], xo[e], Di);
      xq[e] = lowk ? xq[e] * a.isd : 0.f;
      xg[e] = lowk ? xg[e] : 0.f;
    }
    ax_split8(xq, qh, ql);
    ax_split8(xg, gh, gl);
  }
  Di += __shfl_xor(Di, 16, 64);
  Di += __shfl_xor(Di, 32, 64);
  const float lse_i = a.lse[nh * Tp + (active ? i : Tp - 1)];
  if (active && g == 0) a.Dbuf[nh * Tp + i] = Di;

  f32x4 dq[OT];
#pragma unroll
  for (int t = 0; t < OT; ++t) dq[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float* const psk = Psk + (w * 16 + ii) * AX_PSK;
  float* const psd = Psd + (w * 16 + ii) * AX_PSD;
  float* const dsrow = a.dS + (nh * Tp + (active ? i : 0)) * Tp;

  float4 rk[NU], rv[NU], rb[NBU];
  auto fetch = [&](int j0) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int idx = tid + 256 * u;
      const int j = j0 + idx / (DK / 4), sc4 = idx % (DK / 4);
      rk[u] = zero4();
      rv[u] = zero4();
      if (j < Tp) {
        const float* kp = base + (long long)j * ld + F + 4 * sc4;
        rk[u] = ld4(kp);
        rv[u] = ld4(kp + F);
      }
    }
#pragma unroll
    for (int u = 0; u < NBU; ++u) {
      const int idx = tid + 256 * u;
      const int rr = idx / (DK / 4) < NBAND ? idx / (DK / 4) : NBAND - 1;
      int rel = i0 - j0 - (KT - 1) + rr;
      rel = rel < -a.maxlen ? -a.maxlen : (rel > a.maxlen - 1 ? a.maxlen - 1 : rel);
      rb[u] = ld4(a.pe + (long long)(rel + a.maxlen) * DK + 4 * (idx % (DK / 4)));
    }
  };
  fetch(0);
  for (int j0 = 0; j0 < Tp; j0 += KT) {
    __syncthreads();
    {
      ax_bf16x4 hh, ll;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int idx = tid + 256 * u;
        const int sjj = idx / (DK / 4), sc4 = idx % (DK / 4);
        ax_split4(rk[u], hh, ll);
        *reinterpret_cast<ax_bf16x4*>(Kh + sjj * KSB + 4 * sc4) = hh;
        if constexpr (!ONE) *reinterpret_cast<ax_bf16x4*>(Kl + sjj * KSB + 4 * sc4) = ll;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          Kth[(4 * sc4 + e) * VSB + sjj] = hh[e];
          if constexpr (!ONE) Ktl[(4 * sc4 + e) * VSB + sjj] = ll[e];
        }
        ax_split4(rv[u], hh, ll);
        *reinterpret_cast<ax_bf16x4*>(Vh + sjj * KSB + 4 * sc4) = hh;
        if constexpr (!ONE) *reinterpret_cast<ax_bf16x4*>(Vl + sjj * KSB + 4 * sc4) = ll;
      }
#pragma unroll
      for (int u = 0; u < NBU; ++u) {
        const int idx = tid + 256 * u;
        if (idx < NBAND * (DK / 4)) {
          const int rr = idx / (DK / 4), sc4 = idx % (DK / 4);
          ax_split4(rb[u], hh, ll);
          *reinterpret_cast<ax_bf16x4*>(Bh + rr * KSB + 4 * sc4) = hh;
          if constexpr (!ONE) *reinterpret_cast<ax_bf16x4*>(Bl + rr * KSB + 4 * sc4) = ll;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            Bth[(4 * sc4 + e) * AX_BTS + rr] = hh[e];
            if constexpr (!ONE) Btl[(4 * sc4 + e) * AX_BTS + rr] = ll[e];
          }
        }
      }
    }
    __syncthreads();
    if (j0 + KT < Tp) fetch(j0 + KT);

    const int npair = (Tp - j0 >= KT) ? KT / 32 : (Tp - j0 + 31) / 32;
    for (int p = 0; p < npair; ++p) {
      // ---- S^T[key = 16 s + 4g + r][query ii] and dP^T = V dO^T in the same layout ---------------------------------------
      f32x4 sc[2], dp[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int row = 32 * p + 16 * s + ii;
        const ax_bf16x8 kh = *reinterpret_cast<const ax_bf16x8*>(Kh + row * KSB + go);
        const ax_bf16x8 kl = *reinterpret_cast<const ax_bf16x8*>(Kl + row * KSB + go);
        const ax_bf16x8 vh = *reinterpret_cast<const ax_bf16x8*>(Vh + row * KSB + go);
        const ax_bf16x8 vl = *reinterpret_cast<const ax_bf16x8*>(Vl + row * KSB + go);
        sc[s] = ax_mma<ONE>(kh, kl, qh, ql, (f32x4){0.f, 0.f, 0.f, 0.f});
        dp[s] = ax_mma<ONE>(vh, vl, gh, gl, (f32x4){0.f, 0.f, 0.f, 0.f});
      }
      // ---- relative-position bias: P^T[b][query] = band[bb + b] . q, b = ql - kl + 31 --------------------------------------
      const int bb = 16 * w - 32 * p + 32;
#pragma unroll
      for (int tb = 0; tb < 3; ++tb) {
        const int row = bb + 16 * tb + ii;
        const int rc = row < NBAND ? row : NBAND - 1;
        const ax_bf16x8 bh = *reinterpret_cast<const ax_bf16x8*>(Bh + rc * KSB + go);
        const ax_bf16x8 bl = *reinterpret_cast<const ax_bf16x8*>(Bl + rc * KSB + go);
        const f32x4 r4 = ax_mma<ONE>(bh, bl, qh, ql, (f32x4){0.f, 0.f, 0.f, 0.f});
        st4(psk + 16 * tb + 4 * g, make_float4(r4[0], r4[1], r4[2], r4[3]));
      }
      float ds[8];
      const unsigned row_id = (unsigned)(nh * Tp + i);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int b0 = ii + 31 - 16 * s - 4 * g;
        const int jbase = j0 + 32 * p + 16 * s + 4 * g;
        float bias[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[r] = psk[b0 - r];
        float keep[4] = {a.dscale, a.dscale, a.dscale, a.dscale};
        if (a.thr) {
          const unsigned d0 = sepr_drop_word(dkey, row_id, (unsigned)jbase >> 1), d1 = sepr_drop_word(dkey, row_id, ((unsigned)jbase >> 1) + 1u);
          keep[0] = (d0 & 0xffffu) >= a.thr ? a.dscale : 0.f;
          keep[1] = (d0 >> 16) >= a.thr ? a.dscale : 0.f;
          keep[2] = (d1 & 0xffffu) >= a.thr ? a.dscale : 0.f;
          keep[3] = (d1 >> 16) >= a.thr ? a.dscale : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool kin = jbase + r < Tp;
          const float pv = kin ? __expf(sc[s][r] + bias[r] - lse_i) : 0.f;
          ds[4 * s + r] = pv * (dp[s][r] * keep[r] - Di) * a.isd;
        }
        // the dS rows go to HBM once, for the table gradient (16 bytes per lane, 64 contiguous bytes per query)
        if (a.ds16) {       // kernel-uniform
          __bf16* d16 = reinterpret_cast<__bf16*>(a.dS) + (nh * Tp + (active ? i : 0)) * Tp;
          if (active && jbase + 3 < Tp && (Tp & 3) == 0) {
            ax_bf16x4 h4 = {(__bf16)ds[4 * s], (__bf16)ds[4 * s + 1], (__bf16)ds[4 * s + 2], (__bf16)ds[4 * s + 3]};
            *reinterpret_cast<ax_bf16x4*>(d16 + jbase) = h4;
          } else if (active) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (jbase + r < Tp) d16[jbase + r] = (__bf16)ds[4 * s + r];
          }
        } else if (active && jbase + 3 < Tp && (Tp & 3) == 0) st4(dsrow + jbase, make_float4(ds[4 * s], ds[4 * s + 1], ds[4 * s + 2], ds[4 * s + 3]));
        else if (active) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (jbase + r < Tp) dsrow[jbase + r] = ds[4 * s + r];
        }
      }
      ax_bf16x8 dsh, dsl;
      ax_split8(ds, dsh, dsl);
      // ---- dQ^T[d][query] += K^T[d][key slots] . dS^T[key slots][query] ------------------------------------------------------
#pragma unroll
      for (int t = 0; t < OT; ++t) {
        const __bf16* kh0 = Kth + (16 * t + ii) * VSB + 32 * p + 4 * g;
        const __bf16* kl0 = Ktl + (16 * t + ii) * VSB + 32 * p + 4 * g;
        const ax_bf16x4 a0 = *reinterpret_cast<const ax_bf16x4*>(kh0), a1 = *reinterpret_cast<const ax_bf16x4*>(kh0 + 16);
        const ax_bf16x4 b0v = *reinterpret_cast<const ax_bf16x4*>(kl0), b1v = *reinterpret_cast<const ax_bf16x4*>(kl0 + 16);
        const ax_bf16x8 kth = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        const ax_bf16x8 ktl = {b0v[0], b0v[1], b0v[2], b0v[3], b1v[0], b1v[1], b1v[2], b1v[3]};
        dq[t] = ax_mma<ONE>(kth, ktl, dsh, dsl, dq[t]);
      }
      // ---- position part: un-skew dS into band slots b = ql - kl + 31 of this query's scratch row, contract with band^T -------
      st4(psd + 16 * g, zero4());
      st4(psd + 16 * g + 4, zero4());
      st4(psd + 16 * g + 8, zero4());
      st4(psd + 16 * g + 12, zero4());
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) psd[ii + 31 - 16 * s - 4 * g - r] = ds[4 * s + r];
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) {
        const float4 u0 = ld4(psd + 32 * tb + 8 * g), u1 = ld4(psd + 32 * tb + 8 * g + 4);
        const float xb[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
        ax_bf16x8 bsh, bsl;
        ax_split8(xb, bsh, bsl);
#pragma unroll
        for (int t = 0; t < OT; ++t) {
          const int col = bb + 32 * tb + 8 * g;
          const ax_bf16x8 bth = *reinterpret_cast<const ax_bf16x8*>(Bth + (16 * t + ii) * AX_BTS + col);
          const ax_bf16x8 btl = *reinterpret_cast<const ax_bf16x8*>(Btl + (16 * t + ii) * AX_BTS + col);
          dq[t] = ax_mma<ONE>(bth, btl, bsh, bsl, dq[t]);
        }
      }
    }
  }
  if (active) {
#pragma unroll
    for (int t = 0; t < OT; ++t)
      st4(a.dQKV + ((long long)seq * Tp + i) * ld + h * DK + 16 * t + 4 * g, make_float4(dq[t][0], dq[t][1], dq[t][2], dq[t][3]));
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// key-major: dK, dV
// ---------------------------------------------------------------------------------------------------------------------
template <int DK, bool ONE>
__global__ __launch_bounds__(256) void relattn_bwd_kv_x3_kernel(const AxArgs a) {
  static_assert(DK == 16 || DK == 32, "head width");
  constexpr int KB = AX_QB, QT = AX_KT, NBAND = AX_NBAND;       // 64 keys per workgroup, query tiles of 64
  constexpr int KSB = DK + 8, VSB = QT + 8, OT = DK / 16;
  constexpr int NU = QT * (DK / 4) / 256, NBU = (127 * (DK / 4) + 255) / 256;
  constexpr int PS2 = 52;
  constexpr int LO = ONE ? 0 : 1;
  __shared__ __attribute__((aligned(16))) __bf16 Qh[QT * KSB], Ql_[LO * QT * KSB + 8];       // Q rows (scores, bias)
  __shared__ __attribute__((aligned(16))) __bf16 Gh[QT * KSB], Gl_[LO * QT * KSB + 8];       // dO rows (dP)
  __shared__ __attribute__((aligned(16))) __bf16 Qth[DK * VSB], Qtl_[LO * DK * VSB + 8];     // Q^T (dK)
  __shared__ __attribute__((aligned(16))) __bf16 Gth[DK * VSB], Gtl_[LO * DK * VSB + 8];     // dO^T (dV)
  __shared__ __attribute__((aligned(16))) __bf16 Bh[NBAND * KSB], Bl_[LO * NBAND * KSB + 8]; // band rows
  __bf16* const Ql = ONE ? Qh : Ql_;          // (ONE: dead aliases, see the query-major kernel)
  __bf16* const Gl = ONE ? Gh : Gl_;
  __bf16* const Qtl = ONE ? Qth : Qtl_;
  __bf16* const Gtl = ONE ? Gth : Gtl_;
  __bf16* const Bl = ONE ? Bh : Bl_;
  __shared__ __attribute__((aligned(16))) float Ps2[4 * 32 * PS2];                 // [wave][query of the step][b]
  __shared__ __attribute__((aligned(16))) float lse_s[QT];
  __shared__ __attribute__((aligned(16))) float D_s[QT];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ii = lane & 15, g = lane >> 4;
  const int Tp = a.Tp, F = a.F;
  const int j0 = blockIdx.x * KB, h = blockIdx.y, seq = blockIdx.z;
  const int ld = 3 * F;
  const long long nh = (long long)seq * a.H + h;
  const float* base = a.QKV + (long long)seq * Tp * ld + h * DK;
  const float* gbase = a.dO + (long long)seq * Tp * F + h * DK;
  const int j = j0 + 16 * w + ii;
  const bool active = j < Tp;
  const bool lowk = DK == 32 || g < 2;
  const int gk = DK == 32 ? g : (g & 1);
  const int go = DK == 32 ? 8 * g : 8 * (g < 2 ? g : 2);
  DropKey dkey = {0u, 0u};
  if (a.thr) dkey = sepr_drop_key(a.seed, a.salt, 2u);
  for (int r = tid; r < QT; r += 256) {
#pragma unroll
    for (int e = DK; e < KSB; ++e) {
      Qh[r * KSB + e] = Gh[r * KSB + e] = (__bf16)0.f;
      if constexpr (!ONE) Ql[r * KSB + e] = Gl[r * KSB + e] = (__bf16)0.f;
    }
  }
  for (int r = tid; r < NBAND; r += 256) {
#pragma unroll
    for (int e = DK; e < KSB; ++e) {
      Bh[r * KSB + e] = (__bf16)0.f;
      if constexpr (!ONE) Bl[r * KSB + e] = (__bf16)0.f;
    }
  }
  // B fragments of this lane's key: k (scaled: (q isd) . k == q . (k isd)) and v
  ax_bf16x8 kh, kl, vh, vl;
  {
    const float* kp = base + (long long)(active ? j : Tp - 1) * ld + F + 8 * gk;
    const float4 k0 = ld4(kp), k1 = ld4(kp + 4), v0 = ld4(kp + F), v1 = ld4(kp + F + 4);
    float xk[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
    float xv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      xk[e] = lowk ? xk[e] * a.isd : 0.f;
      xv[e] = lowk ? xv[e] : 0.f;
    }
    ax_split8(xk, kh, kl);
    ax_split8(xv, vh, vl);
  }
  f32x4 dk_[OT], dv_[OT];
#pragma unroll
  for (int t = 0; t < OT; ++t) {
    dk_[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    dv_[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  float* const ps2 = Ps2 + w * 32 * PS2;

  float4 rq[NU], rg[NU], rb[NBU];
  float rl = 0.f, rd = 0.f;
  auto fetch = [&](int i0) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int idx = tid + 256 * u;
      const int i = i0 + idx / (DK / 4), sc4 = idx % (DK / 4);
      rq[u] = zero4();
      rg[u] = zero4();
      if (i < Tp) {
        rq[u] = ld4(base + (long long)i * ld + 4 * sc4);
        rg[u] = ld4(gbase + (long long)i * F + 4 * sc4);
      }
    }
#pragma unroll
    for (int u = 0; u < NBU; ++u) {
      const int idx = tid + 256 * u;
      const int rr = idx / (DK / 4) < NBAND ? idx / (DK / 4) : NBAND - 1;
      int rel = i0 - j0 - (KB - 1) + rr;                       // i - j of band row rr for this (query tile, key block)
      rel = rel < -a.maxlen ? -a.maxlen : (rel > a.maxlen - 1 ? a.maxlen - 1 : rel);
      rb[u] = ld4(a.pe + (long long)(rel + a.maxlen) * DK + 4 * (idx % (DK / 4)));
    }
    if (tid < QT) {
      const int i = i0 + tid;
      rl = i < Tp ? a.lse[nh * Tp + i] : 0.f;
      rd = i < Tp ? a.Dbuf[nh * Tp + i] : 0.f;
    }
  };
  fetch(0);
  for (int i0 = 0; i0 < Tp; i0 += QT) {
    __syncthreads();
    {
      ax_bf16x4 hh, ll;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int idx = tid + 256 * u;
        const int sii = idx / (DK / 4), sc4 = idx % (DK / 4);
        ax_split4(rq[u], hh, ll);
        *reinterpret_cast<ax_bf16x4*>(Qh + sii * KSB + 4 * sc4) = hh;
        if constexpr (!ONE) *reinterpret_cast<ax_bf16x4*>(Ql + sii * KSB + 4 * sc4) = ll;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          Qth[(4 * sc4 + e) * VSB + sii] = hh[e];
          if constexpr (!ONE) Qtl[(4 * sc4 + e) * VSB + sii] = ll[e];
        }
        ax_split4(rg[u], hh, ll);
        *reinterpret_cast<ax_bf16x4*>(Gh + sii * KSB + 4 * sc4) = hh;
        if constexpr (!ONE) *reinterpret_cast<ax_bf16x4*>(Gl + sii * KSB + 4 * sc4) = ll;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          Gth[(4 * sc4 + e) * VSB + sii] = hh[e];
          if constexpr (!ONE) Gtl[(4 * sc4 + e) * VSB + sii] = ll[e];
        }
      }
#pragma unroll
      for (int u = 0; u < NBU; ++u) {
        const int idx = tid + 256 * u;
        if (idx < NBAND * (DK / 4)) {
          ax_split4(rb[u], hh, ll);
          *reinterpret_cast<ax_bf16x4*>(Bh + (idx / (DK / 4)) * KSB + 4 * (idx % (DK / 4))) = hh;
          if constexpr (!ONE) *reinterpret_cast<ax_bf16x4*>(Bl + (idx / (DK / 4)) * KSB + 4 * (idx % (DK / 4))) = ll;
        }
      }
      if (tid < QT) {
        lse_s[tid] = rl;
        D_s[tid] = rd;
      }
    }
    __syncthreads();
    if (i0 + QT < Tp) fetch(i0 + QT);

    const int npair = (Tp - i0 >= QT) ? QT / 32 : (Tp - i0 + 31) / 32;
    for (int p = 0; p < npair; ++p) {
      // ---- S[query = 16 s + 4g + r][key ii], dP = dO v^T, and R[query][b] = Q band^T for the 47 slots of this step ----------
      f32x4 sc[2], dp[2];
      const int bbc = 32 * p - 16 * w + 48;                   // band row of (query ql of the step, key kl) = bbc + ql - kl + 15
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int row = 32 * p + 16 * s + ii;
        const ax_bf16x8 qh_ = *reinterpret_cast<const ax_bf16x8*>(Qh + row * KSB + go);
        const ax_bf16x8 ql_ = *reinterpret_cast<const ax_bf16x8*>(Ql + row * KSB + go);
        const ax_bf16x8 gh_ = *reinterpret_cast<const ax_bf16x8*>(Gh + row * KSB + go);
        const ax_bf16x8 gl_ = *reinterpret_cast<const ax_bf16x8*>(Gl + row * KSB + go);
        sc[s] = ax_mma<ONE>(qh_, ql_, kh, kl, (f32x4){0.f, 0.f, 0.f, 0.f});
        dp[s] = ax_mma<ONE>(gh_, gl_, vh, vl, (f32x4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int tb = 0; tb < 3; ++tb) {
          const int brow = bbc + 16 * tb + ii;
          const int rc = brow < NBAND ? (brow < 0 ? 0 : brow) : NBAND - 1;
          const ax_bf16x8 bh = *reinterpret_cast<const ax_bf16x8*>(Bh + rc * KSB + go);
          const ax_bf16x8 bl = *reinterpret_cast<const ax_bf16x8*>(Bl + rc * KSB + go);
          // the band fragment is the B operand here (n = slot b): its rows hold DK values (+ zero pad), the query rows carry isd
          const f32x4 r4 = ax_mma<ONE>(qh_, ql_, bh, bl, (f32x4){0.f, 0.f, 0.f, 0.f});
          // D[m = query 4g + r of half s][n = slot 16 tb + ii]
#pragma unroll
          for (int r = 0; r < 4; ++r) ps2[(16 * s + 4 * g + r) * PS2 + 16 * tb + ii] = r4[r];
        }
      }
      float ds[8], pd[8];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int ql0 = 16 * s + 4 * g;                        // query of the step held in register r: ql0 + r
        const float4 l4 = ld4(lse_s + 32 * p + ql0), d4 = ld4(D_s + 32 * p + ql0);
        const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq_[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int iq = i0 + 32 * p + ql0 + r;
          const float bias = ps2[(ql0 + r) * PS2 + (ql0 + r) - ii + 15];
          const bool qin = iq < Tp;
          const float pv = qin ? __expf(fmaf(bias, a.isd, sc[s][r]) - lq[r]) : 0.f;   // (R carries no 1/sqrt(dk): Q and the band are unscaled)
          float keep = a.dscale;
          if (a.thr) {
            const unsigned d = sepr_drop_word(dkey, (unsigned)(nh * Tp + iq), (unsigned)j >> 1);
            keep = ((j & 1) ? (d >> 16) : (d & 0xffffu)) >= a.thr ? a.dscale : 0.f;
          }
          ds[4 * s + r] = pv * (dp[s][r] * keep - dq_[r]) * a.isd;
          pd[4 * s + r] = pv * keep;
        }
      }
      ax_bf16x8 dsh, dsl, pdh, pdl;
      ax_split8(ds, dsh, dsl);
      ax_split8(pd, pdh, pdl);
      // ---- dK^T[d][key] += Q^T[d][query slots] . dS;  dV^T[d][key] += dO^T[d][query slots] . Pd ------------------------------
#pragma unroll
      for (int t = 0; t < OT; ++t) {
        const int off = (16 * t + ii) * VSB + 32 * p + 4 * g;
        const ax_bf16x4 a0 = *reinterpret_cast<const ax_bf16x4*>(Qth + off), a1 = *reinterpret_cast<const ax_bf16x4*>(Qth + off + 16);
        const ax_bf16x4 b0v = *reinterpret_cast<const ax_bf16x4*>(Qtl + off), b1v = *reinterpret_cast<const ax_bf16x4*>(Qtl + off + 16);
        const ax_bf16x8 qth = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        const ax_bf16x8 qtl = {b0v[0], b0v[1], b0v[2], b0v[3], b1v[0], b1v[1], b1v[2], b1v[3]};
        dk_[t] = ax_mma<ONE>(qth, qtl, dsh, dsl, dk_[t]);
        const ax_bf16x4 c0 = *reinterpret_cast<const ax_bf16x4*>(Gth + off), c1 = *reinterpret_cast<const ax_bf16x4*>(Gth + off + 16);
        const ax_bf16x4 e0 = *reinterpret_cast<const ax_bf16x4*>(Gtl + off), e1 = *reinterpret_cast<const ax_bf16x4*>(Gtl + off + 16);
        const ax_bf16x8 gth = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
        const ax_bf16x8 gtl = {e0[0], e0[1], e0[2], e0[3], e1[0], e1[1], e1[2], e1[3]};
        dv_[t] = ax_mma<ONE>(gth, gtl, pdh, pdl, dv_[t]);
      }
    }
  }
  if (active) {
    float* o = a.dQKV + ((long long)seq * Tp + j) * ld + h * DK + 4 * g;
#pragma unroll
    for (int t = 0; t < OT; ++t) {
      // dk carries one isd from dS; the k fragment's own isd belongs to the score, not to the gradient
      st4(o + F + 16 * t, make_float4(dk_[t][0], dk_[t][1], dk_[t][2], dk_[t][3]));
      st4(o + 2 * F + 16 * t, make_float4(dv_[t][0], dv_[t][1], dv_[t][2], dv_[t][3]));
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// table gradient: band[grp][rel + Tp - 1][d] = sum over the group's (sequence, head) pairs and queries of
// dS[nh][i][i - rel] * q[nh][i][d]   (rel = i - j in (-Tp, Tp)); consecutive threads = consecutive rel = consecutive j
// ---------------------------------------------------------------------------------------------------------------------
#ifndef SEPR_AXB_GROUP
#define SEPR_AXB_GROUP 4        // (sequence, head) pairs per workgroup of the table-gradient kernel (tools/variants.mk: 2, 1)
#endif
constexpr int AXB_GROUP = SEPR_AXB_GROUP;
#ifndef SEPR_AXB_ROWS
#define SEPR_AXB_ROWS 8         // (tools/variants.mk builds the 16-row form for the A/B)
#define SEPR_AXB_BPC 4
#endif
constexpr int AXB_ROWS = SEPR_AXB_ROWS;   // rows per load batch of the table-gradient kernel
constexpr int AXB_BPC = SEPR_AXB_BPC;     // batches per LDS chunk of q rows (4 * AXB_ROWS * AXB_BPC = 128 rows)
// One block = 64 consecutive relative offsets x 4 query partitions (one partition per wave).  Every lane of a wave walks the
// SAME queries i and reads dS[i][i - rel] when that key exists; consecutive lanes = consecutive rel = consecutive (descending)
// keys: coalesced reads of every dS row, each element once.  The q rows are wave-uniform: they are staged 128 rows at a time in
// LDS (coalesced) and read back as broadcasts.  Round 4 history of this loop, all on the same 32 MB of dS: a predicated load per
// row + scalar q loads = one memory latency per row (122 us; 211 us when the bf16 / fp32 choice was a run-time select); loads in
// batches of 8 unconditional rows (clamped address, value selected afterwards) with scalar q loads 136 us - the s_load chain was
// the latency; q through LDS + the next batch's dS loads issued before this batch's FMAs 116 us (1.1 TB/s, and 2 x the
// useful FMAs: half of the (row, offset) pairs have no key); rows limited to the ones that can meet the block's offsets 99 us; 16-row
// batches (more bytes in flight) 124 us - NOT latency-bound any more: ~35 instructions per (row, wave) at 4 cycles each are the
// bound, so: packed FMAs (two channels per issue), one clamp per address, a row-range validity test: see profiles.
template <int DK, bool DS16>
__global__ __launch_bounds__(256) void relattn_band_kernel(const float* __restrict__ QKV, const float* __restrict__ dS, float* __restrict__ band,
                                                          int NH, int Tp, int F, int H) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr int CH = 4 * AXB_ROWS * AXB_BPC;                 // q rows per LDS chunk (128)
  __shared__ __attribute__((aligned(16))) float qs[CH * DK];
  __shared__ float red[4][64][DK + 1];
  const int rl = threadIdx.x & 63;
  const int part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rel = blockIdx.x * 64 + rl - (Tp - 1);
  const int grp = blockIdx.y;
  // rows that can meet this block's 64 offsets: j = i - rel in [0, Tp)  =>  i in [r0, r0 + 63 + Tp) - half of the (row, offset)
  // grid of the launch is empty, and whole chunks of it are skipped here
  const int r0 = blockIdx.x * 64 - (Tp - 1);
  const int i_lo = max(0, r0), i_hi = min(Tp, r0 + 63 + Tp);
  const int c_lo = i_lo / CH * CH;
  // this lane's valid rows: i in [v_lo, v_lo + v_n)
  const int v_lo = max(0, rel);
  const unsigned v_n = (unsigned)max(0, min(Tp, rel + Tp) - v_lo);
  const int TT1 = Tp * Tp - 1;
  f32x2 acc[DK / 2];
#pragma unroll
  for (int d = 0; d < DK / 2; ++d) acc[d] = (f32x2){0.f, 0.f};
  for (int gi = 0; gi < AXB_GROUP; ++gi) {
    const int nh = grp * AXB_GROUP + gi;
    if (nh >= NH) break;
    const int n = nh / H, h = nh - n * H;
    const float* qb = QKV + (long long)n * Tp * 3 * F + h * DK;
    const float* sb = dS + (long long)nh * Tp * Tp;
    const unsigned short* sb16 = reinterpret_cast<const unsigned short*>(dS) + (long long)nh * Tp * Tp;
    // batch b of this wave = rows part + 4 (AXB_ROWS b + u), u < AXB_ROWS.  Element (i, j = i - rel) sits at i (Tp + 1) - rel of the
    // head's matrix; the index is clamped INTO the matrix (one v_med3: any element will do for a pair without a key - validity is a
    // row-range test where the value is consumed), every load is unconditional and RAW (the bf16 widening happens at the consumer
    // too): nothing touches the registers of the batch in flight, so its loads stay outstanding under the current batch's FMAs.
    auto load_batch = [&](int b, unsigned (&raw)[AXB_ROWS]) {
      int idx = (part + 4 * AXB_ROWS * b) * (Tp + 1) - rel;
#pragma unroll
      for (int u = 0; u < AXB_ROWS; ++u) {
        const int at = min(max(idx, 0), TT1);
        raw[u] = DS16 ? (unsigned)sb16[at] : __float_as_uint(sb[at]);
        idx += 4 * (Tp + 1);
      }
    };
    unsigned cur[AXB_ROWS], nx[AXB_ROWS];
    load_batch(c_lo / (4 * AXB_ROWS), cur);
    for (int c0 = c_lo, b = c_lo / (4 * AXB_ROWS); c0 < i_hi; c0 += CH) {
      __syncthreads();                                         // the previous chunk's rows are consumed
      for (int e = threadIdx.x; e < CH * (DK / 4); e += 256) {
        const int r = e / (DK / 4), c4 = e - r * (DK / 4);
        st4(qs + r * DK + 4 * c4, ld4(qb + (long long)min(c0 + r, Tp - 1) * 3 * F + 4 * c4));
      }
      __syncthreads();
#pragma unroll
      for (int bb = 0; bb < AXB_BPC; ++bb, ++b) {
        load_batch(b + 1, nx);                                 // (past the end: clamped addresses, discarded)
#pragma unroll
        for (int u = 0; u < AXB_ROWS; ++u) {
          const int i = part + 4 * (AXB_ROWS * b + u);
          const float v = DS16 ? __uint_as_float(cur[u] << 16) : __uint_as_float(cur[u]);
          const float sv = ((unsigned)(i - v_lo) < v_n) ? v : 0.f;
          const f32x2 s2 = (f32x2){sv, sv};
          const float* q = qs + (part + 4 * (AXB_ROWS * bb + u)) * DK;      // wave-uniform: LDS broadcast
#pragma unroll
          for (int d4 = 0; d4 < DK; d4 += 4) {
            const float4 qv = ld4(q + d4);
            acc[d4 / 2] = __builtin_elementwise_fma(s2, (f32x2){qv.x, qv.y}, acc[d4 / 2]);           // v_pk_fma_f32: two channels per issue
            acc[d4 / 2 + 1] = __builtin_elementwise_fma(s2, (f32x2){qv.z, qv.w}, acc[d4 / 2 + 1]);
          }
        }
#pragma unroll
        for (int u = 0; u < AXB_ROWS; ++u) cur[u] = nx[u];
      }
    }
  }
#pragma unroll
  for (int d = 0; d < DK / 2; ++d) {
    red[part][rl][2 * d] = acc[d][0];
    red[part][rl][2 * d + 1] = acc[d][1];
  }
  __syncthreads();
  // 64 offsets x DK channels summed over the 4 partitions in a fixed order
  for (int e = threadIdx.x; e < 64 * DK; e += 256) {
    const int r2 = e / DK, d = e - r2 * DK;
    const int rel2 = blockIdx.x * 64 + r2 - (Tp - 1);
    if (rel2 < Tp)
      band[((long long)grp * (2 * Tp - 1) + rel2 + Tp - 1) * DK + d] = (red[0][r2][d] + red[1][r2][d]) + (red[2][r2][d] + red[3][r2][d]);
  }
}
// dpe[r][d] += sum over groups and over the relative offsets that clamp to table row r
__global__ __launch_bounds__(256) void relattn_band_reduce2_kernel(const float* __restrict__ band, int ngroups, int Tp, int DK, int maxlen,
                                                                  float* __restrict__ dpe) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= 2 * maxlen * DK) return;
  const int r = e / DK, d = e - r * DK;
  int dlo = r - maxlen, dhi = r - maxlen;
  if (r == 0) dlo = -(Tp - 1);
  if (r == 2 * maxlen - 1) dhi = Tp - 1;
  if (dlo < -(Tp - 1)) dlo = -(Tp - 1);
  if (dhi > Tp - 1) dhi = Tp - 1;
  if (dlo > dhi) return;
  float s = 0.f;
  for (int rel = dlo; rel <= dhi; ++rel)
    for (int gq = 0; gq < ngroups; ++gq) s += band[((long long)gq * (2 * Tp - 1) + rel + Tp - 1) * DK + d];
  dpe[e] += s;
}
}  // namespace

size_t relattn_x3_bwd_ws(int n, int Tp, int F, int H) {
  if (n <= 0 || Tp <= 0 || H <= 0) return 0;
  const int DK = F / H;
  const long long NH = (long long)n * H;
  const long long ngroups = (NH + AXB_GROUP - 1) / AXB_GROUP;
  return align_up((size_t)NH * Tp * Tp * sizeof(float)) + align_up((size_t)NH * Tp * sizeof(float)) +
         align_up((size_t)ngroups * (2 * Tp - 1) * DK * sizeof(float));
}

int launch_relattn_x3_bwd(const float* QKV, const float* lse, const float* O, const float* dO, float* dQKV, float* dpe_g, int n, int Tp,
                          int F, int H, const float* pe_k, int maxlen, float p, unsigned long long seed, const unsigned long long* salt,
                          void* ws, size_t ws_bytes, hipStream_t s, int ds16, int one) {
  if (n <= 0) return SEPR_OK;
  if (!QKV || !lse || !O || !dO || !dQKV || !dpe_g || !pe_k || Tp <= 0 || H <= 0 || F % H || n > 65535 || !(p >= 0.f) || !(p < 1.f))
    return SEPR_EINVAL;
  const int DK = F / H;
  if (DK != 16 && DK != 32) return SEPR_EINVAL;
  if (!ws || ws_bytes < relattn_x3_bwd_ws(n, Tp, F, H)) return SEPR_EWORKSPACE;
  const long long NH = (long long)n * H;
  if (NH * Tp >= (1LL << 31)) return SEPR_EINVAL;              // dropout row ids are 32-bit
  const int ngroups = (int)((NH + AXB_GROUP - 1) / AXB_GROUP);
  char* wp = static_cast<char*>(ws);
  AxArgs a;
  a.QKV = QKV; a.O = O; a.dO = dO; a.lse = lse; a.dQKV = dQKV;
  a.dS = reinterpret_cast<float*>(wp);
  a.Dbuf = reinterpret_cast<float*>(wp + align_up((size_t)NH * Tp * Tp * sizeof(float)));
  float* band = reinterpret_cast<float*>(wp + align_up((size_t)NH * Tp * Tp * sizeof(float)) + align_up((size_t)NH * Tp * sizeof(float)));
  a.pe = pe_k; a.Tp = Tp; a.F = F; a.H = H; a.maxlen = maxlen;
  a.isd = 1.0f / sqrtf((float)DK);
  a.thr = p > 0.f ? sepr_drop_thr16(p) : 0u;
  a.dscale = p > 0.f ? sepr_drop_scale16(p) : 1.0f;
  a.seed = seed; a.salt = salt;
  a.ds16 = ds16 ? 1 : 0;
  a.one = one ? 1 : 0;
  const dim3 grid((Tp + 63) / 64, H, n);
  const dim3 bgrid((2 * Tp - 1 + 63) / 64, ngroups);
  if (DK == 16) {
    if (a.one) {
      hipLaunchKernelGGL((relattn_bwd_q_x3_kernel<16, true>), grid, dim3(256), 0, s, a);
      hipLaunchKernelGGL((relattn_bwd_kv_x3_kernel<16, true>), grid, dim3(256), 0, s, a);
    } else {
      hipLaunchKernelGGL((relattn_bwd_q_x3_kernel<16, false>), grid, dim3(256), 0, s, a);
      hipLaunchKernelGGL((relattn_bwd_kv_x3_kernel<16, false>), grid, dim3(256), 0, s, a);
    }
    if (a.ds16) hipLaunchKernelGGL((relattn_band_kernel<16, true>), bgrid, dim3(256), 0, s, QKV, a.dS, band, (int)NH, Tp, F, H);
    else hipLaunchKernelGGL((relattn_band_kernel<16, false>), bgrid, dim3(256), 0, s, QKV, a.dS, band, (int)NH, Tp, F, H);
  } else {
    if (a.one) {
      hipLaunchKernelGGL((relattn_bwd_q_x3_kernel<32, true>), grid, dim3(256), 0, s, a);
      hipLaunchKernelGGL((relattn_bwd_kv_x3_kernel<32, true>), grid, dim3(256), 0, s, a);
    } else {
      hipLaunchKernelGGL((relattn_bwd_q_x3_kernel<32, false>), grid, dim3(256), 0, s, a);
      hipLaunchKernelGGL((relattn_bwd_kv_x3_kernel<32, false>), grid, dim3(256), 0, s, a);
    }
    if (a.ds16) hipLaunchKernelGGL((relattn_band_kernel<32, true>), bgrid, dim3(256), 0, s, QKV, a.dS, band, (int)NH, Tp, F, H);
    else hipLaunchKernelGGL((relattn_band_kernel<32, false>), bgrid, dim3(256), 0, s, QKV, a.dS, band, (int)NH, Tp, F, H);
  }
  hipLaunchKernelGGL(relattn_band_reduce2_kernel, dim3((2 * maxlen * DK + 255) / 256), dim3(256), 0, s, band, ngroups, Tp, DK, maxlen, dpe_g);
  SEPR_CHECK_LAUNCH("relattn_x3 backward kernels");
  return SEPR_OK;
}

}  // namespace sepr
