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
#include "sepr_train.h"

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

// ---------------------------------------------------------------------------------------------------------------------
// bf16x3 form of the same attention for dk = 16 (default arithmetic).  relattn_kernel above is VALU-bound: per 16 keys
// a lane spends 64 FMAs + 16 LDS reads on the relative-position bias against 8 f32 MFMAs (PMC: 74 % VALU-active,
// 31 % MFMA-busy).  Here all three products run on the bf16 MFMA with split operands (hi.hi + hi.lo + lo.hi):
//   * keys are walked 32 at a time; S^T = K q^T is one K=32 MFMA step per 16 keys (dk = 16 fills half of K, the
//     other half of the fragments is zero);
//   * the relative-position term is a THIRD product, P^T[b][query] = band[b] . q for the 47 band rows a
//     (16 queries x 32 keys) block can touch (3 row tiles), skewed through a per-wave LDS scratch:
//     bias(query, key) = P[query][query - key + 31].  9 cheap MFMAs + 3 LDS writes + 8 LDS reads replace 128 FMAs;
//   * O^T += V^T P: the 8 probabilities a lane holds (keys 4g+r of both 16-key halves) are exactly one K=32 B fragment
//     when V^T is read with the matching key-slot order, so PV is 3 MFMAs per 32 keys with no data movement.
// K, V^T and the band are split into bf16 hi/lo planes once per 64-key tile while they are staged into LDS.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef SEPR_AT_ABL
#define SEPR_AT_ABL 0   // timing ablations of relattn_x3_kernel (WRONG results): 1 no relative-position product, 2 no exp,
                       // 4 K / V / band staged once (first key tile only), 8 no PV product, 16 no q.k product
#endif
#ifndef SEPR_AT_MASKPASS
#define SEPR_AT_MASKPASS 1   // key-bound mask as one wave-uniform pass (0: selects inside the score loop - rounds 1-4; 2: pass on every tile)
#endif
#ifndef SEPR_AT_NW
#define SEPR_AT_NW 4         // 16-query waves per workgroup of the Base inference kernel (8: 128 queries per staged tile, round-5 experiment)
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split4(const float4 v, bf16x4& h, bf16x4& l) {
  const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const __bf16 hh = (__bf16)x[e];
    h[e] = hh;
    l[e] = (__bf16)(x[e] - (float)hh);
  }
}

// DK = 16 (Base): dk fills half of the K = 32 MFMA step, lane groups 2,3 carry zeros.  DK = 32 (Large): the step is full,
// O^T has two 16-row tiles (two PV accumulators), the staging moves twice the K / V / band rows per thread.
// TRAIN (sepr_train_attn_x3.hip's forward): additionally writes lse[(seq*H + h)*Tp + i] = log sum_j exp(score_ij) - all the
// backward keeps of the probabilities - and applies inverted dropout to the probabilities that multiply V (network.py:121;
// the softmax denominator sums the undropped ones); mask of element (row = (seq*H + h)*Tp + i, key j) = 16-bit half j & 1 of
// sepr_drop_word(dkey, row, j >> 1) >= thr (sepr_train.h).
template <int DK, bool TRAIN = false, bool BP = false, bool ONE = false, int NWV = 4>
__global__ __launch_bounds__(64 * NWV, DK == 16 ? 4 : 2) void relattn_x3_kernel(const float* __restrict__ QKV, float* __restrict__ O, int Tp, int F,
                                                        const float* __restrict__ pe, int maxlen, float inv_sqrt_dk,
                                                        float* __restrict__ lse = nullptr, unsigned thr = 0u, float dscale = 1.0f,
                                                        unsigned long long seed = 0ull, const unsigned long long* __restrict__ salt = nullptr,
                                                        const unsigned short* __restrict__ pe_planes = nullptr) {
  static_assert(DK == 16 || DK == 32, "head width");
  // pe_planes (inference, round 4): the position table already split into bf16 hi / lo planes at pack time - the band rows of a key
  // tile are then copied global -> registers -> LDS as they are (half of this kernel's staged elements lose their VALU split)
  // BP is a template parameter: as a run-time flag every band fetch computed both tables' addresses and selected (16 VALU per key tile)
  static_assert(!(TRAIN && BP), "the training forward reads the fp32 table");
  // ONE (training forward of the plain-bf16 precision): operands rounded to bf16 once, one MFMA per product, no lo planes in LDS
  static_assert(TRAIN || !ONE, "inference keeps the split-fp32 products");
  constexpr bool bp = BP;
  DropKey dkey = {0u, 0u};
  if (TRAIN && thr) dkey = sepr_drop_key(seed, salt, 2u);
  static_assert(NWV == 4 || NWV == 8, "waves (16-query slices) per workgroup");
  constexpr int NT = 64 * NWV;        // NWV = 8: 128 queries share every staged K / V / band tile (SEPR_AT_NW, inference only)
  constexpr int QB = 16 * NWV, KT = 64;
  constexpr int KSB = DK + 8;         // K / band row stride in bf16 (DK used + 8 pad; DK = 16: the pad is the zero half of K = 32)
  constexpr int OT = DK / 16;         // 16-row tiles of O^T
  constexpr int NU = (KT * (DK / 4) + NT - 1) / NT;                       // K / V float4 per thread per key tile
  constexpr int NBU = ((QB + KT - 1) * (DK / 4) + NT - 1) / NT;             // band float4 per thread
  constexpr int VSB = KT + 8;         // V^T row stride in bf16 (144 B)
  constexpr int NBAND = QB + KT - 1;
  constexpr int PSK = 52;             // skew scratch row stride in floats (48 used)
  constexpr int LO = ONE ? 0 : 1;
  __shared__ __attribute__((aligned(16))) __bf16 Kh[KT * KSB], Kl_[LO * KT * KSB + 8];
  __shared__ __attribute__((aligned(16))) __bf16 Vh[DK * VSB], Vl_[LO * DK * VSB + 8];
  __shared__ __attribute__((aligned(16))) __bf16 Bh[NBAND * KSB], Bl_[LO * NBAND * KSB + 8];
  __bf16* const Kl = ONE ? Kh : Kl_;          // ONE: dead aliases (the reads through them stay in bounds and feed nothing)
  __bf16* const Vl = ONE ? Vh : Vl_;
  __bf16* const Bl = ONE ? Bh : Bl_;
  __shared__ __attribute__((aligned(16))) float Psk[NWV * 16 * PSK];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ii = lane & 15, g = lane >> 4;
  const int i0 = blockIdx.x * QB, h = blockIdx.y, seq = blockIdx.z;
  const int ld = 3 * F;
  const float* base = QKV + (long long)seq * Tp * ld + h * DK;
  const int i = i0 + 16 * w + ii;
  const bool active = i < Tp;
  const bool lowk = DK == 32 || g < 2;            // DK = 16: lane groups 2,3 carry the zero half of the K = 32 fragments:
  const int gk = DK == 32 ? g : (g & 1);          // they read the (zeroed) 8-element pad at the end of every K / band row
  const int go = DK == 32 ? 8 * g : 8 * (g < 2 ? g : 2);
  for (int r = tid; r < KT; r += NT) {
#pragma unroll
    for (int e = DK; e < KSB; ++e) {
      Kh[r * KSB + e] = (__bf16)0.f;
      if constexpr (!ONE) Kl[r * KSB + e] = (__bf16)0.f;
    }
  }
  for (int r = tid; r < NBAND; r += NT) {
#pragma unroll
    for (int e = DK; e < KSB; ++e) {
      Bh[r * KSB + e] = (__bf16)0.f;
      if constexpr (!ONE) Bl[r * KSB + e] = (__bf16)0.f;
    }
  }

  // B fragments of this lane's query (scaled), shared by the q.k and the q.band products
  bf16x8 qh, ql;
  {
    const float* qp = base + (long long)(active ? i : Tp - 1) * ld + 8 * gk;
    const float4 a = ld4(qp), b = ld4(qp + 4);
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = lowk ? x[e] * inv_sqrt_dk : 0.f;
      const __bf16 hh = (__bf16)v;
      qh[e] = hh;
      ql[e] = (__bf16)(v - (float)hh);
    }
  }
  f32x4 o[OT];                                 // O^T[d = 16 t + 4g + r][query ii]
#pragma unroll
  for (int t = 0; t < OT; ++t) o[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float mrun = -1e30f, lrun = 0.f;
  float* const psk = Psk + (w * 16 + ii) * PSK;

  // staging registers: NU K and V float4 per thread (64 keys x DK) and NBU band float4 (127 rows x DK)
  float4 rk[NU], rv[NU], rb[NBU];
  auto fetch = [&](int j0) {          // global -> registers for the key tile starting at j0
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int idx = tid + NT * u;
      const int j = j0 + idx / (DK / 4), sc4 = idx % (DK / 4);
      rk[u] = zero4();
      rv[u] = zero4();
      if (j < Tp && (NT * NU == KT * (DK / 4) || idx < KT * (DK / 4))) {
        const float* kp = base + (long long)j * ld + F + 4 * sc4;
        rk[u] = ld4(kp);
        rv[u] = ld4(kp + F);
      }
    }
#pragma unroll
    for (int u = 0; u < NBU; ++u) {
      const int idx = tid + NT * u;
      const int rr = idx / (DK / 4) < NBAND ? idx / (DK / 4) : NBAND - 1;
      int rel = i0 - j0 - (KT - 1) + rr;                      // i - j for band row rr
      rel = rel < -maxlen ? -maxlen : (rel > maxlen - 1 ? maxlen - 1 : rel);
      const long long off = (long long)(rel + maxlen) * DK + 4 * (idx % (DK / 4));
      if (bp) {       // 4 bf16 of the hi plane in .x/.y, of the lo plane in .z/.w (bit patterns carried in the float4 registers)
        const uint2 hh = *reinterpret_cast<const uint2*>(pe_planes + off);
        const uint2 ll = *reinterpret_cast<const uint2*>(pe_planes + 2LL * maxlen * DK + off);
        rb[u] = make_float4(__uint_as_float(hh.x), __uint_as_float(hh.y), __uint_as_float(ll.x), __uint_as_float(ll.y));
      } else {
        rb[u] = ld4(pe + off);
      }
    }
  };
  fetch(0);
  for (int j0 = 0; j0 < Tp; j0 += KT) {
    __syncthreads();   // previous tile fully consumed
    // ---- registers -> LDS: K rows, V transposed and the band of the position table as bf16 hi / lo planes ----------
    if (!(SEPR_AT_ABL & 4) || j0 == 0) {
      bf16x4 hh, ll;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int idx = tid + NT * u;
        const int sjj = idx / (DK / 4), sc4 = idx % (DK / 4);
        if (NT * NU > KT * (DK / 4) && idx >= KT * (DK / 4)) continue;
        split4(rk[u], hh, ll);
        *reinterpret_cast<bf16x4*>(Kh + sjj * KSB + 4 * sc4) = hh;
        if constexpr (!ONE) *reinterpret_cast<bf16x4*>(Kl + sjj * KSB + 4 * sc4) = ll;
        split4(rv[u], hh, ll);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          Vh[(4 * sc4 + e) * VSB + sjj] = hh[e];
          if constexpr (!ONE) Vl[(4 * sc4 + e) * VSB + sjj] = ll[e];
        }
      }
#pragma unroll
      for (int u = 0; u < NBU; ++u) {
        const int idx = tid + NT * u;
        if (idx < NBAND * (DK / 4)) {
          if (bp) {
            *reinterpret_cast<uint2*>(Bh + (idx / (DK / 4)) * KSB + 4 * (idx % (DK / 4))) = make_uint2(__float_as_uint(rb[u].x), __float_as_uint(rb[u].y));
            *reinterpret_cast<uint2*>(Bl + (idx / (DK / 4)) * KSB + 4 * (idx % (DK / 4))) = make_uint2(__float_as_uint(rb[u].z), __float_as_uint(rb[u].w));
          } else {
            split4(rb[u], hh, ll);
            *reinterpret_cast<bf16x4*>(Bh + (idx / (DK / 4)) * KSB + 4 * (idx % (DK / 4))) = hh;
            if constexpr (!ONE) *reinterpret_cast<bf16x4*>(Bl + (idx / (DK / 4)) * KSB + 4 * (idx % (DK / 4))) = ll;
          }
        }
      }
    }
    __syncthreads();
    if (j0 + KT < Tp && !(SEPR_AT_ABL & 4)) fetch(j0 + KT);   // the next tile's rows fly under this tile's arithmetic

    // ---- ONE online-softmax update per 64-key tile (round 4; one per 32 keys before): the scores of both 32-key pairs are formed
    //      first - two independent MFMA -> skew -> bias chains the scheduler can interleave - then one max / exchange / rescale
    //      round, then both P.V products.  Keys past Tp carry -1e30 (their V rows are staged as zeros), so a partial last tile
    //      simply runs both pairs.
    {
      float sv[2][2][4];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        // S^T[key = 16 s + 4g + r][query ii] for the two 16-key halves s
        f32x4 sc[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int row = 32 * p + 16 * s + ii;
          const bf16x8 kh = *reinterpret_cast<const bf16x8*>(Kh + row * KSB + go);
          const bf16x8 kl = *reinterpret_cast<const bf16x8*>(Kl + row * KSB + go);
          f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (!(SEPR_AT_ABL & 16)) {
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, qh, a, 0, 0, 0);
            if constexpr (!ONE) {
              a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, ql, a, 0, 0, 0);
              a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl, qh, a, 0, 0, 0);
            }
          } else {
            a[0] = (float)kh[0] + (float)kl[1];
          }
          sc[s] = a;
        }
        // relative-position term: P^T[b][query], band row of (query ql, key kl) is bb + b, b = ql - kl + 31
        const int bb = 16 * w - 32 * p + 32;
#pragma unroll
        for (int tb = 0; tb < ((SEPR_AT_ABL & 1) ? 0 : 3); ++tb) {
          const int row = bb + 16 * tb + (15 - ii);             // <= 126 except unused rows of the last tile; REVERSED inside the 16-row tile:
                                                                // the lane's four results are then band rows in DESCENDING order (see the store)
          const int rc = row < NBAND ? row : NBAND - 1;
          const bf16x8 bh = *reinterpret_cast<const bf16x8*>(Bh + rc * KSB + go);
          const bf16x8 bl = *reinterpret_cast<const bf16x8*>(Bl + rc * KSB + go);
          f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, qh, a, 0, 0, 0);
          if constexpr (!ONE) {
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, ql, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, qh, a, 0, 0, 0);
          }
          // a[r] = band row b = 16 tb + 15 - 4g - r of this query, stored MIRRORED (row b at float 47 - b = 32 - 16 tb + 4g + r): the four
          // bias values of a key group are then read in ASCENDING address order, i.e. as register pairs in the order of the score pairs they
          // are added to.  With the rows in natural order hipcc packed that add as v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0] (pair swap) -
          // the gfx950-faulty form of sepr_common.h norm4_pinned: THAT was round 4's "nondeterministic mask pass" (tools/isa_lint.py now
          // rejects the form; reversing the band rows in the A fragment costs nothing, reversing the results would cost v_pk_mov's)
          st4(psk + 32 - 16 * tb + 4 * g, make_float4(a[0], a[1], a[2], a[3]));
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int b0 = ii + 31 - 16 * s - 4 * g;              // b of key 16 s + 4g + 0; r steps down
          float bias[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) bias[r] = (SEPR_AT_ABL & 1) ? 0.f : psk[47 - b0 + r];  // unconditional: the reads issue back to back
#pragma unroll
          for (int r = 0; r < 4; ++r)
            sv[p][s][r] = (SEPR_AT_MASKPASS || j0 + 32 * p + 16 * s + 4 * g + r < Tp) ? sc[s][r] + bias[r] : -1e30f;
        }
      }
#if SEPR_AT_MASKPASS
      // only a tile that reaches past Tp pays the 16 key-bound selects: ONE wave-uniform pass (as selects inside the loop above they cost 48
      // VALU per tile and hipcc turned four of the bias reads into exec-masked blocks with their own LDS waits: 300 -> 212 VALU per full
      // tile).  Round 4 withdrew this form as "run-to-run nondeterministic, cause not established"; round 5 established it - not the pass
      // but the packed bias add the compiler formed around it (see the mirrored store above).  2 = the pass on every tile (bisecting aid)
      if (SEPR_AT_MASKPASS == 2 || j0 + KT > Tp) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (j0 + 32 * p + 16 * s + 4 * g + r >= Tp) sv[p][s][r] = -1e30f;
      }
#endif
      float mx = -1e30f;
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int s = 0; s < 2; ++s)
          mx = fmaxf(mx, fmaxf(fmaxf(sv[p][s][0], sv[p][s][1]), fmaxf(sv[p][s][2], sv[p][s][3])));
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mnew = fmaxf(mrun, mx);
      const float corr = __expf(mrun - mnew);
      bf16x8 ph[2], pl[2];
      float psum = 0.f;
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pv = (SEPR_AT_ABL & 2) ? sv[p][s][r] - mnew : __expf(sv[p][s][r] - mnew);
            psum += pv;
            const __bf16 hh = (__bf16)pv;
            ph[p][4 * s + r] = hh;
            pl[p][4 * s + r] = (__bf16)(pv - (float)hh);
          }
      if (TRAIN && thr) {   // dropped probabilities for the PV product only; keys 16 s + 4g + {0,1} / {2,3} are the element pairs
        const unsigned row = (unsigned)((seq * gridDim.y + h) * Tp + i);
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const unsigned jp = (unsigned)(j0 + 32 * p + 16 * s + 4 * g) >> 1;
            const unsigned d0 = sepr_drop_word(dkey, row, jp), d1 = sepr_drop_word(dkey, row, jp + 1u);
            const bool k0 = (d0 & 0xffffu) >= thr, k1 = (d0 >> 16) >= thr, k2 = (d1 & 0xffffu) >= thr, k3 = (d1 >> 16) >= thr;
            if (!k0) { ph[p][4 * s] = (__bf16)0.f; pl[p][4 * s] = (__bf16)0.f; }
            if (!k1) { ph[p][4 * s + 1] = (__bf16)0.f; pl[p][4 * s + 1] = (__bf16)0.f; }
            if (!k2) { ph[p][4 * s + 2] = (__bf16)0.f; pl[p][4 * s + 2] = (__bf16)0.f; }
            if (!k3) { ph[p][4 * s + 3] = (__bf16)0.f; pl[p][4 * s + 3] = (__bf16)0.f; }
          }
      }
      lrun = lrun * corr + psum;
      mrun = mnew;
      // ---- O^T[d][query] += V^T[d][key slots] . P[key slots][query]; slot e -> key 16 (e / 4) + 4g + e % 4 --------
#pragma unroll
      for (int t = 0; t < OT; ++t) {
        o[t][0] *= corr; o[t][1] *= corr; o[t][2] *= corr; o[t][3] *= corr;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const __bf16* vh0 = Vh + (16 * t + ii) * VSB + 32 * p + 4 * g;
          const __bf16* vl0 = Vl + (16 * t + ii) * VSB + 32 * p + 4 * g;
          const bf16x4 a0 = *reinterpret_cast<const bf16x4*>(vh0), a1 = *reinterpret_cast<const bf16x4*>(vh0 + 16);
          const bf16x4 b0v = *reinterpret_cast<const bf16x4*>(vl0), b1v = *reinterpret_cast<const bf16x4*>(vl0 + 16);
          const bf16x8 vh = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
          const bf16x8 vl = {b0v[0], b0v[1], b0v[2], b0v[3], b1v[0], b1v[1], b1v[2], b1v[3]};
          if (SEPR_AT_ABL & 8) {
            o[t][0] += (float)vh[0] * (float)ph[p][0] + (float)vl[1] * (float)pl[p][1];
            continue;
          }
          o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, ph[p], o[t], 0, 0, 0);
          if constexpr (!ONE) {
            o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, pl[p], o[t], 0, 0, 0);
            o[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl, ph[p], o[t], 0, 0, 0);
          }
        }
      }
    }
  }
  float ltot = lrun + __shfl_xor(lrun, 16, 64);
  ltot += __shfl_xor(ltot, 32, 64);
  if (active) {
    const float inv = (TRAIN ? dscale : 1.0f) / ltot;
#pragma unroll
    for (int t = 0; t < OT; ++t)
      st4(O + ((long long)seq * Tp + i) * F + h * DK + 16 * t + 4 * g,
          make_float4(o[t][0] * inv, o[t][1] * inv, o[t][2] * inv, o[t][3] * inv));
    if (TRAIN && lse && g == 0) lse[((long long)seq * gridDim.y + h) * Tp + i] = mrun + logf(ltot);
  }
}

// train forward on the bf16x3 kernel: O, lse [n*H*Tp]; p > 0: dropout of the probabilities (16-bit generator, site 2)
int launch_relattn_x3_train_fwd(const float* QKV, float* O, float* lse, int n, int Tp, int F, int H, const float* pe_k, int maxlen,
                                float p, unsigned long long seed, const unsigned long long* salt, hipStream_t s, int one) {
  if (n <= 0 || Tp <= 0) return SEPR_OK;
  if (H <= 0 || F % H != 0 || maxlen <= 0 || !pe_k || !lse || n > 65535 || !(p >= 0.f) || !(p < 1.f)) return SEPR_EINVAL;
  const int dk = F / H;
  const dim3 grid((Tp + 63) / 64, H, n);
  const float isd = 1.0f / sqrtf((float)dk);
  const unsigned thr = p > 0.f ? sepr_drop_thr16(p) : 0u;
  const float dscale = p > 0.f ? sepr_drop_scale16(p) : 1.0f;
  const unsigned short* none = nullptr;
  if (dk == 16 && one) hipLaunchKernelGGL((relattn_x3_kernel<16, true, false, true>), grid, dim3(256), 0, s, QKV, O, Tp, F, pe_k, maxlen, isd, lse, thr, dscale, seed, salt, none);
  else if (dk == 16) hipLaunchKernelGGL((relattn_x3_kernel<16, true, false, false>), grid, dim3(256), 0, s, QKV, O, Tp, F, pe_k, maxlen, isd, lse, thr, dscale, seed, salt, none);
  else if (dk == 32 && one) hipLaunchKernelGGL((relattn_x3_kernel<32, true, false, true>), grid, dim3(256), 0, s, QKV, O, Tp, F, pe_k, maxlen, isd, lse, thr, dscale, seed, salt, none);
  else if (dk == 32) hipLaunchKernelGGL((relattn_x3_kernel<32, true, false, false>), grid, dim3(256), 0, s, QKV, O, Tp, F, pe_k, maxlen, isd, lse, thr, dscale, seed, salt, none);
  else return SEPR_EINVAL;
  SEPR_CHECK_LAUNCH("relattn_x3_kernel<train>");
  return SEPR_OK;
}

int launch_relattn(const float* QKV, float* O, int n, int Tp, int F, int H, const float* pe_k, int maxlen, int x3,
                   hipStream_t s, const void* pe_planes) {
  if (n <= 0 || Tp <= 0) return SEPR_OK;
  if (H <= 0 || F % H != 0 || maxlen <= 0 || !pe_k || n > 65535) return SEPR_EINVAL;
  const int dk = F / H;
  const dim3 grid((Tp + 63) / 64, H, n);
  const float isd = 1.0f / sqrtf((float)dk);
  if (dk == 16 && x3) {
    if (pe_planes)
#if SEPR_AT_NW == 8
        hipLaunchKernelGGL((relattn_x3_kernel<16, false, true, false, 8>), dim3((Tp + 127) / 128, H, n), dim3(512), 0, s, QKV, O, Tp, F, pe_k, maxlen,
                           isd, (float*)nullptr, 0u, 1.0f, 0ull, (const unsigned long long*)nullptr, static_cast<const unsigned short*>(pe_planes));
#else
      hipLaunchKernelGGL((relattn_x3_kernel<16, false, true>), grid, dim3(256), 0, s, QKV, O, Tp, F, pe_k, maxlen, isd, (float*)nullptr, 0u, 1.0f, 0ull,
                         (const unsigned long long*)nullptr, static_cast<const unsigned short*>(pe_planes));
#endif
    else
      hipLaunchKernelGGL((relattn_x3_kernel<16, false, false>), grid, dim3(256), 0, s, QKV, O, Tp, F, pe_k, maxlen, isd, (float*)nullptr, 0u, 1.0f,
                         0ull, (const unsigned long long*)nullptr, (const unsigned short*)nullptr);
  } else if (dk == 32 && x3) {
    if (pe_planes)
      hipLaunchKernelGGL((relattn_x3_kernel<32, false, true>), grid, dim3(256), 0, s, QKV, O, Tp, F, pe_k, maxlen, isd, (float*)nullptr, 0u, 1.0f, 0ull,
                         (const unsigned long long*)nullptr, static_cast<const unsigned short*>(pe_planes));
    else
      hipLaunchKernelGGL((relattn_x3_kernel<32, false, false>), grid, dim3(256), 0, s, QKV, O, Tp, F, pe_k, maxlen, isd, (float*)nullptr, 0u, 1.0f,
                         0ull, (const unsigned long long*)nullptr, (const unsigned short*)nullptr);
  } else if (dk == 16) {
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
