// EGA attention for the training path (network.py:99-124 with the relative-position keys of module.py:52-57,196-198):
// forward that stores the softmax probabilities, and the backward through them.
//
//   scores[i][j] = (q_i . k_j + q_i . pe[r(i,j)]) / sqrt(dk),  r(i,j) = clamp(i - j, -maxlen, maxlen - 1) + maxlen
//   P = softmax_j(scores),  o_i = sum_j P[i][j] v_j
//   backward: D_i = dO_i . o_i;  dP = dO_i . v_j;  dS = P (dP - D_i) / sqrt(dk)
//             dq_i = sum_j dS[i][j] (k_j + pe[r]);  dk_j = sum_i dS[i][j] q_i;  dv_j = sum_i P[i][j] dO_i
//             dpe[r] += sum_{(n,h,i,j): r(i,j) = r} dS[i][j] q_i
// This is 5 % of the model's FLOPs, so the kernels are plain fp32 VALU code (exact arithmetic, no MFMA): a workgroup owns
// AT_QT query rows of one (sequence, head) with the full score rows in LDS.  The inference path keeps its own online-
// softmax MFMA kernel (sepr_attention.hip); this one exists because the backward needs P.
// All reductions are fixed-order (no atomics): dpe is accumulated per workgroup over a group of (sequence, head) pairs into
// a band of (Tp + AT_QT - 1) relative offsets and the bands are summed by a second kernel.
#include "sepr_train.h"

namespace sepr {
namespace {
constexpr int AT_QT = 8;        // query rows per workgroup
constexpr int AT_THREADS = 256;
constexpr int AT_GROUP = 8;     // (sequence, head) pairs a backward workgroup walks (band partial reuse)
constexpr int AT_TPMAX = 1792;  // 8 rows x 1792 x 4 B = 56 KB of LDS (+ 5 KB of staging, under the 64 KB dynamic limit)

__device__ __forceinline__ int relidx(int i, int j, int maxlen) {
  int r = i - j;
  r = r < -maxlen ? -maxlen : (r > maxlen - 1 ? maxlen - 1 : r);
  return r + maxlen;
}

// ---- forward ---------------------------------------------------------------------------------------------------------
template <int DK>
__global__ __launch_bounds__(AT_THREADS) void relattn_train_fwd_kernel(const float* __restrict__ QKV, float* __restrict__ O,
                                                                      float* __restrict__ P, int Tp, int F, int H,
                                                                      const float* __restrict__ pe, int maxlen, float isd, unsigned int thr,
                                                                      float dscale, unsigned long long seed, unsigned long long offset,
                                                                      const unsigned long long* __restrict__ salt) {
  seed = sepr_salted(seed, salt);
  extern __shared__ float sm[];
  float* sc = sm;                        // [AT_QT][Tp]
  float* qs = sm + AT_QT * Tp;           // [AT_QT][DK]
  float* red = qs + AT_QT * DK;          // [8 partitions][AT_QT][DK]
  const int nh = blockIdx.x, n = nh / H, h = nh - n * H;
  const int i0 = blockIdx.y * AT_QT;
  const int tid = threadIdx.x;
  const float* base = QKV + (long long)n * Tp * 3 * F + h * DK;
  for (int e = tid; e < AT_QT * DK; e += AT_THREADS) {
    const int i = i0 + e / DK;
    qs[e] = i < Tp ? base[(long long)i * 3 * F + (e % DK)] : 0.f;
  }
  __syncthreads();
  for (int j = tid; j < Tp; j += AT_THREADS) {
    float kj[DK];
#pragma unroll
    for (int d = 0; d < DK; d += 4) {
      const float4 v = ld4(base + (long long)j * 3 * F + F + d);
      kj[d] = v.x; kj[d + 1] = v.y; kj[d + 2] = v.z; kj[d + 3] = v.w;
    }
#pragma unroll
    for (int ii = 0; ii < AT_QT; ++ii) {
      const float* pr = pe + (long long)relidx(i0 + ii, j, maxlen) * DK;
      float a = 0.f;
#pragma unroll
      for (int d = 0; d < DK; d += 4) {
        const float4 pv = ld4(pr + d), qv = ld4(qs + ii * DK + d);
        a = fmaf(qv.x, kj[d] + pv.x, a);
        a = fmaf(qv.y, kj[d + 1] + pv.y, a);
        a = fmaf(qv.z, kj[d + 2] + pv.z, a);
        a = fmaf(qv.w, kj[d + 3] + pv.w, a);
      }
      sc[ii * Tp + j] = a * isd;
    }
  }
  __syncthreads();
  {  // softmax: 32 lanes per row
    const int row = tid >> 5, l = tid & 31;
    float mx = -3.0e38f;
    for (int j = l; j < Tp; j += 32) mx = fmaxf(mx, sc[row * Tp + j]);
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 32));
    float den = 0.f;
    for (int j = l; j < Tp; j += 32) {
      const float e = expf(sc[row * Tp + j] - mx);
      sc[row * Tp + j] = e;
      den += e;
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) den += __shfl_xor(den, o, 32);
    const float inv = 1.0f / den;
    const int i = i0 + row;
    float* prow = P + ((long long)nh * Tp + i) * Tp;
    for (int j = l; j < Tp; j += 32) {
      const float p = sc[row * Tp + j] * inv;
      sc[row * Tp + j] = p;
      if (i < Tp) prow[j] = p;
    }
  }
  __syncthreads();
  {  // o_i = sum_j P[i][j] v_j : thread = (partition 8, row 8, d4 group)
    constexpr int D4 = DK / 4;
    const int d4 = tid % D4, row = (tid / D4) % AT_QT, part = tid / (D4 * AT_QT);
    constexpr int NPART = AT_THREADS / (D4 * AT_QT);
    float4 acc = zero4();
    const unsigned long long rowoff = offset + ((unsigned long long)nh * Tp + (i0 + row)) * Tp;
    // (unrolled by 8: the V rows of 8 iterations are requested together - one global-load latency per 8 keys instead of per key;
    //  with one load in flight this loop was ~60 dependent L2 round trips per workgroup, the bulk of the kernel's time)
#pragma unroll 8
    for (int j = part; j < Tp; j += NPART) {
      float p = sc[row * Tp + j];
      if (thr) p = sepr_keep(seed, rowoff + j, thr) ? p * dscale : 0.f;      // network.py:121 (dropout on the probabilities)
      const float4 v = ld4(base + (long long)j * 3 * F + 2 * F + 4 * d4);
      acc.x = fmaf(p, v.x, acc.x); acc.y = fmaf(p, v.y, acc.y); acc.z = fmaf(p, v.z, acc.z); acc.w = fmaf(p, v.w, acc.w);
    }
    st4(red + ((part * AT_QT + row) * DK) + 4 * d4, acc);
    __syncthreads();
    if (part == 0) {
      float4 s = zero4();
      for (int pp = 0; pp < NPART; ++pp) {
        const float4 v = ld4(red + ((pp * AT_QT + row) * DK) + 4 * d4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      const int i = i0 + row;
      if (i < Tp) st4(O + ((long long)n * Tp + i) * F + h * DK + 4 * d4, s);
    }
  }
}

// ---- backward, row pass: dS rows, dQ, relative-position band partials -------------------------------------------------
template <int DK, int NOWN>
__global__ __launch_bounds__(AT_THREADS) void relattn_bwd_rows_kernel(const float* __restrict__ QKV, const float* __restrict__ P,
                                                                     const float* __restrict__ O, const float* __restrict__ dO,
                                                                     float* __restrict__ dQKV, float* __restrict__ dS,
                                                                     float* __restrict__ band, int NH, int Tp, int F, int H,
                                                                     const float* __restrict__ pe, int maxlen, float isd, unsigned int thr,
                                                                     float dscale, unsigned long long seed, unsigned long long offset,
                                                                     const unsigned long long* __restrict__ salt) {
  seed = sepr_salted(seed, salt);
  extern __shared__ float sm[];
  float* sc = sm;                        // [AT_QT][Tp]   P, then dS
  float* qs = sm + AT_QT * Tp;           // [AT_QT][DK]
  float* gs = qs + AT_QT * DK;           // [AT_QT][DK]   dO rows
  float* Ds = gs + AT_QT * DK;           // [AT_QT]
  float* red = Ds + AT_QT;               // [NPART][AT_QT][DK]
  constexpr int D4 = DK / 4;
  constexpr int NPART = AT_THREADS / (D4 * AT_QT);
  const int tid = threadIdx.x;
  const int i0 = blockIdx.y * AT_QT;
  const int nslot = Tp + AT_QT - 1;                     // relative offsets i - j seen by this query tile: [i0-Tp+1, i0+7]
  const int nown = (nslot * D4 + AT_THREADS - 1) / AT_THREADS;
  float4 bacc[NOWN];                                     // nown <= NOWN (the launcher picks the instantiation)
#pragma unroll
  for (int u = 0; u < NOWN; ++u) bacc[u] = zero4();

  for (int gi = 0; gi < AT_GROUP; ++gi) {
    const int nh = blockIdx.x * AT_GROUP + gi;
    if (nh >= NH) break;
    const int n = nh / H, h = nh - n * H;
    const float* base = QKV + (long long)n * Tp * 3 * F + h * DK;
    __syncthreads();
    for (int e = tid; e < AT_QT * DK; e += AT_THREADS) {
      const int i = i0 + e / DK;
      qs[e] = i < Tp ? base[(long long)i * 3 * F + (e % DK)] : 0.f;
      gs[e] = i < Tp ? dO[((long long)n * Tp + i) * F + h * DK + (e % DK)] : 0.f;
    }
    for (int e = tid; e < AT_QT * Tp; e += AT_THREADS) {
      const int i = i0 + e / Tp;
      sc[e] = i < Tp ? P[((long long)nh * Tp + i) * Tp + (e % Tp)] : 0.f;
    }
    __syncthreads();
    if (tid < AT_QT) {
      const int i = i0 + tid;
      float a = 0.f;
      if (i < Tp)
        for (int d = 0; d < DK; ++d) a = fmaf(gs[tid * DK + d], O[((long long)n * Tp + i) * F + h * DK + d], a);
      Ds[tid] = a;
    }
    __syncthreads();
    // dS[i][j] = P (dO_i . v_j - D_i) / sqrt(dk)
    for (int j = tid; j < Tp; j += AT_THREADS) {
      float vj[DK];
#pragma unroll
      for (int d = 0; d < DK; d += 4) {
        const float4 v = ld4(base + (long long)j * 3 * F + 2 * F + d);
        vj[d] = v.x; vj[d + 1] = v.y; vj[d + 2] = v.z; vj[d + 3] = v.w;
      }
#pragma unroll
      for (int ii = 0; ii < AT_QT; ++ii) {
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < DK; d += 4) {
          const float4 gv = ld4(gs + ii * DK + d);
          a = fmaf(gv.x, vj[d], a); a = fmaf(gv.y, vj[d + 1], a); a = fmaf(gv.z, vj[d + 2], a); a = fmaf(gv.w, vj[d + 3], a);
        }
        if (thr) a = sepr_keep(seed, offset + ((unsigned long long)nh * Tp + (i0 + ii)) * Tp + j, thr) ? a * dscale : 0.f;
        const float ds = sc[ii * Tp + j] * (a - Ds[ii]) * isd;
        sc[ii * Tp + j] = ds;
        if (i0 + ii < Tp) dS[((long long)nh * Tp + i0 + ii) * Tp + j] = ds;
      }
    }
    __syncthreads();
    {  // dq_i = sum_j dS[i][j] (k_j + pe[r(i,j)])
      const int d4 = tid % D4, row = (tid / D4) % AT_QT, part = tid / (D4 * AT_QT);
      float4 acc = zero4();
#pragma unroll 8
      for (int j = part; j < Tp; j += NPART) {
        const float s = sc[row * Tp + j];
        const float4 kv = ld4(base + (long long)j * 3 * F + F + 4 * d4);
        const float4 pv = ld4(pe + (long long)relidx(i0 + row, j, maxlen) * DK + 4 * d4);
        acc.x = fmaf(s, kv.x + pv.x, acc.x); acc.y = fmaf(s, kv.y + pv.y, acc.y);
        acc.z = fmaf(s, kv.z + pv.z, acc.z); acc.w = fmaf(s, kv.w + pv.w, acc.w);
      }
      st4(red + ((part * AT_QT + row) * DK) + 4 * d4, acc);
      __syncthreads();
      if (part == 0) {
        float4 s = zero4();
        for (int pp = 0; pp < NPART; ++pp) {
          const float4 v = ld4(red + ((pp * AT_QT + row) * DK) + 4 * d4);
          s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        const int i = i0 + row;
        if (i < Tp) st4(dQKV + ((long long)n * Tp + i) * 3 * F + h * DK + 4 * d4, s);
      }
    }
    // band[slot][d] += sum_i dS[i][i - delta] q_i,  delta = slot + i0 - Tp + 1
#pragma unroll
    for (int u = 0; u < NOWN; ++u) {
      if (u < nown) {
        const int e = tid + u * AT_THREADS;
        if (e < nslot * D4) {
          const int slot = e / D4, d4 = e - slot * D4;
          const int delta = slot + i0 - Tp + 1;
#pragma unroll
          for (int ii = 0; ii < AT_QT; ++ii) {
            const int j = i0 + ii - delta;
            if (j >= 0 && j < Tp && i0 + ii < Tp) {
              const float s = sc[ii * Tp + j];
              const float4 qv = ld4(qs + ii * DK + 4 * d4);
              bacc[u].x = fmaf(s, qv.x, bacc[u].x); bacc[u].y = fmaf(s, qv.y, bacc[u].y);
              bacc[u].z = fmaf(s, qv.z, bacc[u].z); bacc[u].w = fmaf(s, qv.w, bacc[u].w);
            }
          }
        }
      }
    }
  }
  float* bp = band + ((long long)blockIdx.x * gridDim.y + blockIdx.y) * nslot * DK;
#pragma unroll
  for (int u = 0; u < NOWN; ++u) {
    if (u < nown) {
      const int e = tid + u * AT_THREADS;
      if (e < nslot * D4) st4(bp + 4 * e, bacc[u]);
    }
  }
}

// ---- backward, column pass: dK, dV ------------------------------------------------------------------------------------
template <int DK>
__global__ __launch_bounds__(AT_THREADS) void relattn_bwd_cols_kernel(const float* __restrict__ QKV, const float* __restrict__ P,
                                                                     const float* __restrict__ dS, const float* __restrict__ dO,
                                                                     float* __restrict__ dQKV, int Tp, int F, int H, unsigned int thr,
                                                                     float dscale, unsigned long long seed, unsigned long long offset,
                                                                     const unsigned long long* __restrict__ salt) {
  seed = sepr_salted(seed, salt);
  constexpr int D4 = DK / 4;
  constexpr int NPART = AT_THREADS / (D4 * AT_QT);
  __shared__ float redk[NPART][AT_QT][DK], redv[NPART][AT_QT][DK];
  const int nh = blockIdx.x, n = nh / H, h = nh - n * H;
  const int j0 = blockIdx.y * AT_QT;
  const int tid = threadIdx.x;
  const int d4 = tid % D4, col = (tid / D4) % AT_QT, part = tid / (D4 * AT_QT);
  const float* base = QKV + (long long)n * Tp * 3 * F + h * DK;
  const int j = j0 + col;
  float4 ak = zero4(), av = zero4();
  if (j < Tp) {
#pragma unroll 8
    for (int i = part; i < Tp; i += NPART) {
      const long long e = ((long long)nh * Tp + i) * Tp + j;
      float p = P[e];
      const float s = dS[e];
      if (thr) p = sepr_keep(seed, offset + (unsigned long long)e, thr) ? p * dscale : 0.f;
      const float4 qv = ld4(base + (long long)i * 3 * F + 4 * d4);
      const float4 gv = ld4(dO + ((long long)n * Tp + i) * F + h * DK + 4 * d4);
      ak.x = fmaf(s, qv.x, ak.x); ak.y = fmaf(s, qv.y, ak.y); ak.z = fmaf(s, qv.z, ak.z); ak.w = fmaf(s, qv.w, ak.w);
      av.x = fmaf(p, gv.x, av.x); av.y = fmaf(p, gv.y, av.y); av.z = fmaf(p, gv.z, av.z); av.w = fmaf(p, gv.w, av.w);
    }
  }
  st4(&redk[part][col][4 * d4], ak);
  st4(&redv[part][col][4 * d4], av);
  __syncthreads();
  if (part == 0 && j < Tp) {
    float4 sk = zero4(), sv = zero4();
    for (int pp = 0; pp < NPART; ++pp) {
      const float4 a = ld4(&redk[pp][col][4 * d4]), b = ld4(&redv[pp][col][4 * d4]);
      sk.x += a.x; sk.y += a.y; sk.z += a.z; sk.w += a.w;
      sv.x += b.x; sv.y += b.y; sv.z += b.z; sv.w += b.w;
    }
    float* o = dQKV + ((long long)n * Tp + j) * 3 * F + h * DK + 4 * d4;
    st4(o + F, sk);
    st4(o + 2 * F, sv);
  }
}

// ---- band partials -> dpe: block = 64 (table row r, channel) pairs x 4 lanes over the query tiles --------------------------
__global__ __launch_bounds__(AT_THREADS) void relattn_band_reduce_kernel(const float* __restrict__ band, int ngroups, int ntiles, int Tp,
                                                                        int DK, int maxlen, float* __restrict__ dpe) {
  __shared__ float sh[4][64];
  const int el = threadIdx.x & 63, tl = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;
  const bool live = e < 2 * maxlen * DK;
  const int r = live ? e / DK : 0, d = live ? e - r * DK : 0;
  const int nslot = Tp + AT_QT - 1;
  // relative offsets delta = i - j in [-(Tp-1), Tp-1] that map to table row r
  int dlo = r - maxlen, dhi = r - maxlen;
  if (r == 0) dlo = -(Tp - 1);
  if (r == 2 * maxlen - 1) dhi = Tp - 1;
  if (dlo < -(Tp - 1)) dlo = -(Tp - 1);
  if (dhi > Tp - 1) dhi = Tp - 1;
  float s = 0.f;
  if (live) {
    for (int delta = dlo; delta <= dhi; ++delta) {
      for (int t = tl; t < ntiles; t += 4) {
        const int slot = delta - (t * AT_QT - Tp + 1);
        if (slot < 0 || slot >= nslot) continue;
#pragma unroll 4
        for (int g = 0; g < ngroups; ++g) s += band[(((long long)g * ntiles + t) * nslot + slot) * DK + d];
      }
    }
  }
  sh[tl][el] = s;
  __syncthreads();
  if (tl == 0 && live && dlo <= dhi) dpe[e] += (sh[0][el] + sh[1][el]) + (sh[2][el] + sh[3][el]);
}

size_t fwd_shm(int Tp, int DK) { return (size_t)(AT_QT * Tp + AT_QT * DK + (AT_THREADS / ((DK / 4) * AT_QT)) * AT_QT * DK) * sizeof(float); }
size_t bwd_shm(int Tp, int DK) {
  return (size_t)(AT_QT * Tp + 2 * AT_QT * DK + AT_QT + (AT_THREADS / ((DK / 4) * AT_QT)) * AT_QT * DK) * sizeof(float);
}
}  // namespace

size_t relattn_train_ws(int n, int Tp, int F, int H) {
  if (n <= 0 || Tp <= 0 || H <= 0) return 0;
  const int DK = F / H;
  const long long NH = (long long)n * H;
  const int ntiles = (Tp + AT_QT - 1) / AT_QT, ngroups = (int)((NH + AT_GROUP - 1) / AT_GROUP);
  return align_up((size_t)NH * Tp * Tp * sizeof(float)) + align_up((size_t)ngroups * ntiles * (Tp + AT_QT - 1) * DK * sizeof(float));
}

int launch_relattn_train_fwd(const float* QKV, float* O, float* P, int n, int Tp, int F, int H, const float* pe_k, int maxlen, float p,
                             unsigned long long seed, unsigned long long offset, hipStream_t s) {
  if (n <= 0) return SEPR_OK;
  if (!QKV || !O || !P || !pe_k || Tp <= 0 || Tp > AT_TPMAX || H <= 0 || F % H || !(p >= 0.f) || !(p < 1.f)) return SEPR_EINVAL;
  const int DK = F / H;
  const float isd = 1.0f / sqrtf((float)DK);
  const unsigned int thr = p > 0.f ? sepr_drop_threshold(p) : 0u;
  const float dscale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  const dim3 grid(n * H, (Tp + AT_QT - 1) / AT_QT);
  if (DK == 16) {
    hipLaunchKernelGGL((relattn_train_fwd_kernel<16>), grid, dim3(AT_THREADS), fwd_shm(Tp, 16), s, QKV, O, P, Tp, F, H, pe_k, maxlen, isd, thr,
                       dscale, seed, offset, drop_salt());
  } else if (DK == 32) {
    hipLaunchKernelGGL((relattn_train_fwd_kernel<32>), grid, dim3(AT_THREADS), fwd_shm(Tp, 32), s, QKV, O, P, Tp, F, H, pe_k, maxlen, isd, thr,
                       dscale, seed, offset, drop_salt());
  } else {
    return SEPR_EINVAL;
  }
  SEPR_CHECK_LAUNCH("relattn_train_fwd_kernel");
  return SEPR_OK;
}

int launch_relattn_bwd(const float* QKV, const float* P, const float* O, const float* dO, float* dQKV, float* dpe_g, int n, int Tp,
                       int F, int H, const float* pe_k, int maxlen, float p, unsigned long long seed, unsigned long long offset, void* ws,
                       size_t ws_bytes, hipStream_t s) {
  if (n <= 0) return SEPR_OK;
  if (!QKV || !P || !O || !dO || !dQKV || !dpe_g || !pe_k || Tp <= 0 || Tp > AT_TPMAX || H <= 0 || F % H || !(p >= 0.f) || !(p < 1.f))
    return SEPR_EINVAL;
  const unsigned int thr = p > 0.f ? sepr_drop_threshold(p) : 0u;
  const float dscale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  const int DK = F / H;
  if (DK != 16 && DK != 32) return SEPR_EINVAL;
  const int NH = n * H;
  const int ntiles = (Tp + AT_QT - 1) / AT_QT, ngroups = (NH + AT_GROUP - 1) / AT_GROUP;
  const int nslot = Tp + AT_QT - 1;
  const int nown = (nslot * (DK / 4) + AT_THREADS - 1) / AT_THREADS;
  if (nown > 32) return SEPR_EINVAL;
  if (!ws || ws_bytes < relattn_train_ws(n, Tp, F, H)) return SEPR_EWORKSPACE;
  float* dS = static_cast<float*>(ws);
  float* band = reinterpret_cast<float*>(static_cast<char*>(ws) + align_up((size_t)NH * Tp * Tp * sizeof(float)));
  const float isd = 1.0f / sqrtf((float)DK);
#define SEPR_ROWS(DD, NO)                                                                                                         \
  hipLaunchKernelGGL((relattn_bwd_rows_kernel<DD, NO>), dim3(ngroups, ntiles), dim3(AT_THREADS), bwd_shm(Tp, DD), s, QKV, P, O, dO, \
                     dQKV, dS, band, NH, Tp, F, H, pe_k, maxlen, isd, thr, dscale, seed, offset, drop_salt())
  if (DK == 16) {
    if (nown <= 8) SEPR_ROWS(16, 8);
    else if (nown <= 16) SEPR_ROWS(16, 16);
    else SEPR_ROWS(16, 32);
    hipLaunchKernelGGL((relattn_bwd_cols_kernel<16>), dim3(NH, ntiles), dim3(AT_THREADS), 0, s, QKV, P, dS, dO, dQKV, Tp, F, H, thr,
                       dscale, seed, offset, drop_salt());
  } else {
    if (nown <= 8) SEPR_ROWS(32, 8);
    else if (nown <= 16) SEPR_ROWS(32, 16);
    else SEPR_ROWS(32, 32);
    hipLaunchKernelGGL((relattn_bwd_cols_kernel<32>), dim3(NH, ntiles), dim3(AT_THREADS), 0, s, QKV, P, dS, dO, dQKV, Tp, F, H, thr,
                       dscale, seed, offset, drop_salt());
  }
#undef SEPR_ROWS
  hipLaunchKernelGGL(relattn_band_reduce_kernel, dim3((2 * maxlen * DK + 63) / 64), dim3(AT_THREADS), 0, s, band,
                     ngroups, ntiles, Tp, DK, maxlen, dpe_g);
  SEPR_CHECK_LAUNCH("relattn_bwd kernels");
  return SEPR_OK;
}

}  // namespace sepr
