// Weight-gradient contraction of the training path:  G[N][K] (+)= sum_m A[m][n] * B'[m][k]   (see sepr_train.h)
//
// Every parameter gradient of a projection is this "TN" product: the contraction runs over the M = batch x frames
// rows (up to 512 000) while N and K are small (16 .. 1024), the transpose of the forward projection's shape.
// Design (gfx950, wave64):
//   * one workgroup (4 waves) owns a 128 x 128 tile of G for a contiguous slice of the rows; the slices of one tile
//     are written as partial tiles to the workspace and summed in slice order by a second kernel (no atomics:
//     bit-reproducible gradients);
//   * a wave holds a 64 x 64 quadrant as 4 x 4 MFMA accumulators.  The MFMA contraction index is the ROW index m, so
//     both operands are needed "m-fastest": the staging threads read 8 consecutive rows x 4 columns (row-contiguous
//     512-byte segments per 32 lanes), transpose in registers and write one 16-byte [column][8 rows] vector per plane,
//     which is exactly the fragment a lane later reads (16 lanes x 16 B, conflict-free at an 80-byte row stride);
//   * bf16x3 arithmetic like the forward projections (sepr_gemm_x3.h): both operands are activations here, so both are
//     split into bf16 hi/lo while staging; products hi.hi + hi.lo + lo.hi on v_mfma_f32_16x16x32_bf16, fp32 accumulate.
//     The exact mode uses v_mfma_f32_16x16x4_f32 on fp32 LDS tiles (no transpose needed: one k per lane group);
//   * the B prologue replays what the forward projection did to its input (normalise with saved statistics, concat,
//     crop / nearest-upsample / overlapping-frame row maps), so no normalised or gathered copy is ever materialised;
//   * the column sums of A (bias gradients) ride along in the staging registers.
#include "sepr_train.h"
#include <stdio.h>
#include <stdlib.h>

#ifndef SEPR_TN_ABL
#define SEPR_TN_ABL 0   // timing ablations (wrong results): 1 no MFMAs, 2 no conversion / LDS staging, 4 no global loads
#endif
namespace sepr {

typedef __bf16 tn_bf16x8 __attribute__((ext_vector_type(8)));

namespace {
constexpr int TN_T = 128;          // tile edge (both n and k)
constexpr int TN_SLAB = 64;        // rows per staging step: 64 KB of operands in flight per workgroup (the launches are
                                   // latency-bound: M / 512 slices of a few hundred rows each)
constexpr int TN_NB = TN_SLAB / 32;
constexpr int TN_LDM = TN_SLAB + 8;   // bf16 elements per LDS row of the x3 planes (64 rows + pad: 144 B, conflict-free 16-B reads)
constexpr int TN_LDF = TN_T + 4;   // floats per LDS row of the f32 tiles
constexpr int TN_THREADS = 256;

struct TnPlan {
  int tn, tk, nsplit, rows_per_split;
};
inline TnPlan tn_plan(int M, int N, int K) {
  TnPlan p;
  p.tn = (N + TN_T - 1) / TN_T;
  p.tk = (K + TN_T - 1) / TN_T;
  const int tiles = p.tn * p.tk;
  // Row slices: enough workgroups to fill the chip (2 co-resident per CU = 512), but every slice pays a partial tile of
  // up to 64 KB written and re-read by the reduction, so a slice is at least 256 rows (its inputs: 256 x (N + K) x 4 B).
  // SEPR_TN_WGS / SEPR_TN_MINROWS: experiment knobs of tools/wgrad_bench.py (read once; the defaults are the product plan)
  static const int target_wgs = [] { const char* e = getenv("SEPR_TN_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 512; }();
  static const int min_rows = [] { const char* e = getenv("SEPR_TN_MINROWS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 256; }();
  int ns = target_wgs / tiles;
  if (ns < 1) ns = 1;
  const int max_by_rows = (M + min_rows - 1) / min_rows;
  if (ns > max_by_rows) ns = max_by_rows;
  if (ns < 1) ns = 1;
  int rps = (M + ns - 1) / ns;
  rps = (rps + TN_SLAB - 1) / TN_SLAB * TN_SLAB;
  p.nsplit = (M + rps - 1) / rps;
  p.rows_per_split = rps;
  return p;
}

// MD: 0 = exact f32 MFMA, 1 = bf16x3 split arithmetic, 2 = plain bf16 operands (hi planes only, one MFMA per product).
// GEN: the general B prologue (row maps, index tables, two-source concat, per-sequence statistics, masked A rows) - the fusion
//      conv, the output heads, the encoder / projector: a handful of launches per step.  !GEN: B is a plain [M][ldb] tensor,
//      optionally normalised with per-ROW statistics (STATS) - every block's projections, ~290 launches per step.
// The !GEN loader is straight-line: all 16 row loads of a thread are issued back to back from clamped (always valid)
// addresses, validity is a select afterwards, and the per-row (mean, rstd) pairs of a slab are fetched ONCE by 64 threads
// and handed out through LDS instead of 16 x 8-byte loads per staging thread.
#ifndef SEPR_TN_ONE_WPE
#define SEPR_TN_ONE_WPE 2   // waves per SIMD the plain-bf16 instantiations are compiled for (3: round-5 experiment, two LDS planes + 168 VGPRs)
#endif
template <int MD, bool GEN, bool STATS>
__global__ __launch_bounds__(TN_THREADS, MD == 2 ? SEPR_TN_ONE_WPE : 2) void gemm_tn_kernel(const TnArgs a, const TnPlan p, float* __restrict__ part,
                                                               float* __restrict__ cpart) {
  // x3: [A_hi, A_lo, B_hi, B_lo][128][72] bf16 = 73 728 B;  plain bf16: [A_hi, B_hi] = 36 864 B;  f32: [A, B][64][132] fp32 = 67 584 B
  // (the plain-bf16 form stays at two workgroups per CU all the same: 208 VGPRs - profiles/r05_v4_wgrad_3waves.txt)
  constexpr bool X3 = MD != 0, ONE = MD == 2;
  __shared__ __attribute__((aligned(16))) unsigned char smem[X3 ? (ONE ? 2 : 4) * TN_T * TN_LDM * 2 : 2 * TN_SLAB * TN_LDF * 4];   // ONE: hi planes only
  __shared__ float csum_s[4][TN_T];
  // DBUF: TWO slabs of operand rows in flight in registers (the loads of slab s+2 issued before the MFMAs of slab s, consumed two
  // barriers later).  Built in round 4 for the plain-bf16 instantiation and NOT enabled: next to the 64 accumulator and 32 fragment
  // registers a second 64-register row set spills 67 dwords per lane into the slab loop (scratch traffic shares vmcnt with the
  // operand stream), and the launch is HBM-stream-bound at two workgroups per CU anyway (64 KB per slab per workgroup against
  // ~500 cycles of MFMAs).  The code path is kept behind the constant for the day the accumulator tile shrinks.
  constexpr bool DBUF = false;
  __shared__ float2 st_s[DBUF ? 2 : 1][TN_SLAB];          // (mean, rstd) of the slab's rows (!GEN && STATS)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int fi = lane & 15, fg = lane >> 4;
  // XCD-aware walk: hardware block b runs on XCD b % 8, and each XCD has its own L2.  The tiles of one row slice share an operand
  // (B for the tn tiles of a k column, A for the tk tiles of an n row), so they get CONSECUTIVE virtual ids on ONE XCD - the shared
  // slab is fetched from HBM once per slice instead of once per tile (PMC, round 3: 1.67 x the algorithmic bytes before).
  const int gq = gridDim.x >> 3, gr = gridDim.x & 7, bx = blockIdx.x & 7;
  const int vb = bx * gq + (bx < gr ? bx : gr) + (blockIdx.x >> 3);
  const int tile = vb % (p.tn * p.tk), split = vb / (p.tn * p.tk);
  const int n0 = (tile / p.tk) * TN_T, k0 = (tile % p.tk) * TN_T;
  const int m_beg = split * p.rows_per_split;
  const int m_end = min(a.M, m_beg + p.rows_per_split);

  // ---- staging role: threads 0..127 stage A, 128..255 stage B; each 8 rows x 4 columns per slab ----
  const bool roleA = tid < 128;
  const int t7 = tid & 127;
#ifndef SEPR_TN_LANEMAP
#define SEPR_TN_LANEMAP 0
#endif
  // lane -> (column group cg, row group mg).  LANEMAP 1 (packed-bf16 planes only): the row group is the FAST lane index, so the 8 lanes
  // the LDS serves together store 4 row groups x 2 column groups = 8 distinct 16-byte bank windows (with cg fast, the 576-byte lane
  // stride of the transposing store reaches only 16 of the 32 banks: PMC 65 % conflict cycles, profiles/r03_v5_pmc_train_kernels.txt)
  const int cg = (SEPR_TN_LANEMAP && MD != 0) ? (t7 >> 2) : (t7 & 31), mg = (SEPR_TN_LANEMAP && MD != 0) ? (t7 & 3) : (t7 >> 5);
  const int col = (roleA ? n0 : k0) + 4 * cg;
  const bool col_ok = col < (roleA ? a.N : a.K);
  float4 rA[8 * TN_NB], rB[DBUF ? 8 * TN_NB : 1];
  float4 csum = zero4();
  float2 my_st = make_float2(0.f, 1.f);      // wave 2 only: statistics of row (slab base + lane) for the next slab

  const int row_safe = m_beg < a.M ? m_beg : 0;
  const int col_c = col_ok ? col : 0;
  auto load_slab = [&](int mb, float4 (&r)[8 * TN_NB]) {
    if (SEPR_TN_ABL & 4) return;
    if constexpr (!GEN) {
      const float* base = roleA ? a.A : a.B;                      // wave-uniform
      const long long ld = roleA ? a.lda : a.ldb;
      if (roleA ? a.a16 != 0 : a.b16 != 0) {                      // wave-uniform: a bf16 operand (8 bytes per 4 columns)
        const unsigned short* b16 = reinterpret_cast<const unsigned short*>(base);
#pragma unroll
        for (int e = 0; e < 8 * TN_NB; ++e) {
          const int m = mb + 32 * (e >> 3) + 8 * mg + (e & 7);
          const uint2 u = *reinterpret_cast<const uint2*>(b16 + (long long)(m < m_end ? m : row_safe) * ld + col_c);
          r[e] = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                             __uint_as_float(u.y & 0xffff0000u));
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8 * TN_NB; ++e) {
          const int m = mb + 32 * (e >> 3) + 8 * mg + (e & 7);
          r[e] = ld4(base + (long long)(m < m_end ? m : row_safe) * ld + col_c);
        }
      }
      if (STATS && wid == 2) {
        const int m = mb + lane;
        my_st = *reinterpret_cast<const float2*>(a.stats + 2LL * (m < m_end ? m : row_safe));
        if (m >= m_end) my_st = make_float2(0.f, 0.f);           // rows past the slice: (0 - 0) * 0
      }
      // (row / column validity is a select at CONSUMPTION, in store_slab: a select here would tie the loads' completion to this
      //  point of the instruction stream)
    } else {
      const bool useB2 = !roleA && a.B2 != nullptr && col_ok && col >= a.ksplit;
#pragma unroll
      for (int e = 0; e < 8 * TN_NB; ++e) {
        const int m0_ = mb + 32 * (e >> 3) + 8 * mg + (e & 7);
        const bool in = m0_ < m_end && col_ok;
        const int m = in ? m0_ : row_safe;
        if (roleA) {                                   // wave-uniform: waves 0,1 stage A, waves 2,3 stage B
          bool valid = in;
          if (a.mask_a && a.rows_out > 0) valid = valid && (m % a.rows_out) < a.rows_valid;
          const float4 v = ld4(a.A + (long long)m * a.lda + col_c);
          r[e] = valid ? v : zero4();
        } else {
          long long off = (long long)m * a.ldb;
          long long srow = m;
          bool valid = in;
          if (a.rows_out > 0) {                        // kernel-uniform
            const int seq = m / a.rows_out;
            const int rr = m - seq * a.rows_out;
            const bool rv = rr < a.rows_valid;
            valid = valid && rv;
            const int rc = rv ? rr : 0;
            const int rs = (a.idx ? a.idx[rc] : rc) >> a.b_shift;
            off = (long long)seq * a.seq_stride + (long long)rs * a.ldb;
            srow = seq;
          }
          const float* src = useB2 ? a.B2 + (long long)m * a.ldb2 + (col_c - a.ksplit) : a.B + off + col_c;
          float4 v = ld4(src);
          if (a.stats) {                               // kernel-uniform
            const long long si = a.stat_seq ? srow : m;
            const float2 st = *reinterpret_cast<const float2*>(a.stats + 2 * si);
            // (form pinned: the compiler's own packing of this expression splats st.y with op_sel:[0,1] - the gfx950 fault of
            //  sepr_common.h norm4_pinned; found as wrong weight gradients in round 3, root-caused in round 5)
            v = norm4_pinned(v, st.x, st.y);
          }
          r[e] = valid ? v : zero4();
        }
      }
    }
  };
  // the statistics a slab's B rows are normalised with travel wave 2 -> LDS -> every staging thread of B (published by the
  // barrier at the top of the loop; the previous slab's readers are past the barrier in the middle of the loop)
  auto publish_stats = [&](int buf) {
    if (!GEN && STATS && wid == 2) st_s[buf][lane] = my_st;
  };
  auto store_slab = [&](float4 (&r)[8 * TN_NB], int sbuf, int mb) {
#pragma clang fp contract(off)
    if constexpr (!GEN) {
#pragma unroll
      for (int e = 0; e < 8 * TN_NB; ++e) {
        const int m = mb + 32 * (e >> 3) + 8 * mg + (e & 7);
        if (!(m < m_end && col_ok)) r[e] = zero4();
      }
    }
    if (SEPR_TN_ABL & 2) {
#pragma unroll
      for (int e = 0; e < 8 * TN_NB; ++e) asm volatile("" ::"v"(r[e].x), "v"(r[e].y), "v"(r[e].z), "v"(r[e].w));
      return;
    }
    if (roleA) {
#pragma unroll
      for (int e = 0; e < 8 * TN_NB; ++e) { csum.x += r[e].x; csum.y += r[e].y; csum.z += r[e].z; csum.w += r[e].w; }
    } else if (!GEN && STATS) {
#pragma unroll
      for (int e = 0; e < 8 * TN_NB; ++e) {
        // (columns past K hold (0 - mean) * rstd garbage: they only ever reach accumulator columns that are never stored)
        const float2 st = st_s[sbuf][32 * (e >> 3) + 8 * mg + (e & 7)];
        // pinned like the general loader (round 6): the natural source - st an 8-byte pair, rstd its high dword - is what hipcc's SLP
        // vectoriser turns into v_pk_mul_f32 op_sel:[0,1], the gfx950-faulty form (sepr_common.h norm4_pinned; tools/isa_lint.py)
        r[e] = norm4_pinned(r[e], st.x, st.y);
      }
    }
    if (X3) {
      unsigned short* hi = reinterpret_cast<unsigned short*>(smem) + (roleA ? 0 : (ONE ? 1 : 2)) * TN_T * TN_LDM;
      unsigned short* lo = hi + TN_T * TN_LDM;
#pragma unroll
      for (int nb = 0; nb < TN_NB; ++nb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          tn_bf16x8 h, l;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float4 q = r[8 * nb + e];
            const float v = j == 0 ? q.x : (j == 1 ? q.y : (j == 2 ? q.z : q.w));
            const __bf16 vh = (__bf16)v;
            h[e] = vh;
            if (!ONE) l[e] = (__bf16)(v - (float)vh);
          }
          *reinterpret_cast<tn_bf16x8*>(hi + (4 * cg + j) * TN_LDM + 32 * nb + 8 * mg) = h;
          if (!ONE) *reinterpret_cast<tn_bf16x8*>(lo + (4 * cg + j) * TN_LDM + 32 * nb + 8 * mg) = l;
        }
    } else {
      float* dst = reinterpret_cast<float*>(smem) + (roleA ? 0 : 1) * TN_SLAB * TN_LDF;
#pragma unroll
      for (int e = 0; e < 8 * TN_NB; ++e) st4(dst + (32 * (e >> 3) + 8 * mg + (e & 7)) * TN_LDF + 4 * cg, r[e]);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // one slab's MFMAs out of the staged planes / tiles
  auto mma_slab = [&]() {
    if (X3) {
      const unsigned short* Ahi = reinterpret_cast<const unsigned short*>(smem);
      const unsigned short* Alo = Ahi + TN_T * TN_LDM;
      const unsigned short* Bhi = Ahi + (ONE ? 1 : 2) * TN_T * TN_LDM;      // ONE: [A_hi][B_hi] only, the lo pointers are never dereferenced
      const unsigned short* Blo = Bhi + TN_T * TN_LDM;
#pragma unroll
      for (int nb = 0; nb < TN_NB; ++nb) {
        tn_bf16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int ra = (wm * 64 + t * 16 + fi) * TN_LDM + 32 * nb + 8 * fg;
          const int rb = (wn * 64 + t * 16 + fi) * TN_LDM + 32 * nb + 8 * fg;
          ah[t] = *reinterpret_cast<const tn_bf16x8*>(Ahi + ra);
          bh[t] = *reinterpret_cast<const tn_bf16x8*>(Bhi + rb);
          if (!ONE) {
            al[t] = *reinterpret_cast<const tn_bf16x8*>(Alo + ra);
            bl[t] = *reinterpret_cast<const tn_bf16x8*>(Blo + rb);
          }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int kt = 0; kt < 4; ++kt) {
            if (SEPR_TN_ABL & 1) { acc[nt][kt][0] += (float)ah[nt][0] + (float)bh[kt][0]; continue; }
            acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[nt], bh[kt], acc[nt][kt], 0, 0, 0);
            if constexpr (!ONE) {
              acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[nt], bl[kt], acc[nt][kt], 0, 0, 0);
              acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[nt], bh[kt], acc[nt][kt], 0, 0, 0);
            }
          }
      }
    } else {
      const float* As = reinterpret_cast<const float*>(smem);
      const float* Bs = As + TN_SLAB * TN_LDF;
#pragma unroll
      for (int kk = 0; kk < TN_SLAB / 4; ++kk) {
        float av[4], bv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          av[t] = As[(4 * kk + fg) * TN_LDF + wm * 64 + t * 16 + fi];
          bv[t] = Bs[(4 * kk + fg) * TN_LDF + wn * 64 + t * 16 + fi];
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
            acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[nt], bv[kt], acc[nt][kt], 0, 0, 0);
      }
    }
  };

  if constexpr (DBUF) {
    // slab s lives in rA for even s, rB for odd s; its statistics in st_s[s & 1]
    if (m_beg < m_end) load_slab(m_beg, rA);
    publish_stats(0);
    if (m_beg + TN_SLAB < m_end) load_slab(m_beg + TN_SLAB, rB);
    publish_stats(1);
    for (int mb = m_beg; mb < m_end; mb += 2 * TN_SLAB) {
      __syncthreads();            // every wave is done reading the previous slab
      store_slab(rA, 0, mb);
      __syncthreads();
      if (mb + 2 * TN_SLAB < m_end) load_slab(mb + 2 * TN_SLAB, rA);
      mma_slab();
      publish_stats(0);           // statistics of slab mb + 2 slabs (readers of st_s[0] are past the barrier above)
      if (mb + TN_SLAB >= m_end) break;
      __syncthreads();
      store_slab(rB, 1, mb + TN_SLAB);
      __syncthreads();
      if (mb + 3 * TN_SLAB < m_end) load_slab(mb + 3 * TN_SLAB, rB);
      mma_slab();
      publish_stats(1);
    }
  } else {
    if (m_beg < m_end) load_slab(m_beg, rA);
    publish_stats(0);
    for (int mb = m_beg; mb < m_end; mb += TN_SLAB) {
      __syncthreads();            // every wave is done reading the previous slab
      store_slab(rA, 0, mb);
      __syncthreads();
      if (mb + TN_SLAB < m_end) load_slab(mb + TN_SLAB, rA);
      mma_slab();
      publish_stats(0);           // statistics of the slab just requested (no reader of st_s between the two barriers above and the next)
    }
  }

  // ---- partial tile -> workspace: part[split][n][k] ----
  float* pt = part + (long long)split * a.N * a.K;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int k = k0 + wn * 64 + kt * 16 + fi;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int n = n0 + wm * 64 + nt * 16 + 4 * fg + rr;
        if (n < a.N && k < a.K) pt[(long long)n * a.K + k] = acc[nt][kt][rr];
      }
    }
  // ---- column sums of A (bias gradient): only the k0 == 0 tile of each row block reports them ----
  if (cpart && k0 == 0) {
    __syncthreads();
    if (roleA) {
      csum_s[mg][4 * cg + 0] = csum.x; csum_s[mg][4 * cg + 1] = csum.y;
      csum_s[mg][4 * cg + 2] = csum.z; csum_s[mg][4 * cg + 3] = csum.w;
    }
    __syncthreads();
    if (tid < TN_T && n0 + tid < a.N)
      cpart[(long long)split * a.N + n0 + tid] = (csum_s[0][tid] + csum_s[1][tid]) + (csum_s[2][tid] + csum_s[3][tid]);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// gemm_tn16_kernel (round 6): the contraction for TWO bf16 operands [M][ld] (the plain-bf16 precision stores dh1, g, x^ and dropout(dy)
// of every GCFN block and the CLA intermediates as bf16) with N and K multiples of 128 - the largest contractions of a training step.
// gemm_tn_kernel stages its slabs through registers (8-byte loads, widen, transpose on the VALU, 16-byte LDS stores) with ONE slab in
// flight per workgroup, and its load phase and MFMA phase do not overlap (profiles/r03_v3_gemm_tn_ablation.txt): 3.0 TB/s on the
// 256 000-row launches.  Here no VALU instruction touches the operands:
//   * slabs of 32 rows x (128 + 128) columns go global -> LDS by LDS-DMA (16 B per lane, 128 contiguous bytes per row and instruction:
//     whole cache lines), into a ring of TN16_NS stages - three slabs (48 KB) in flight per workgroup while one multiplies, two
//     workgroups per CU; the copies are inline asm with counted vmcnt waits and one raw barrier per slab (sepr_common.h glds16_asm);
//   * the MFMA contraction index is the ROW index, so a lane needs 8 rows of one column: ds_read_b64_tr_b16, gfx950's transposing LDS
//     read, delivers 4 rows x 1 column per lane from a row-major image (two reads per fragment).  The slot of fragment element e of lane
//     group fg is row 16 (e >> 2) + 4 fg + (e & 3) of the slab for BOTH operands (any bijection works: the contraction sums over it);
//   * LDS image: one copy instruction = [8 rows][64 columns] = 64 slots of 16 B; the slot of (row r, 16-byte chunk c) is
//     8 r + (c ^ (((r >> 1) & 3) << 1)), which makes the 32 lanes the LDS serves together (rows 0-7 x 32 B of one 16-column tile) hit 16
//     distinct 16-byte bank windows - the copy's per-lane source address carries the permutation, the image itself is lane-linear;
//   * column sums of A (bias gradients): 4 extra MFMAs per slab against an all-ones B fragment in the k0 == 0 tiles.
// Partial tiles and their fixed-order reduction (tn_reduce_kernel) are those of gemm_tn_kernel: same plan, same workspace.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int TN16_NS = 4;                 // ring stages (one multiplying, TN16_NS - 1 in flight)
constexpr int TN16_ROWS = 32;              // rows per slab = one MFMA K step
constexpr int TN16_STAGE_B = 2 * TN16_ROWS * TN_T * 2;   // A + B images of one slab: 16 KB
typedef short tn_s16x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(TN_THREADS, 2) void gemm_tn16_kernel(const TnArgs a, const TnPlan p, float* __restrict__ part, float* __restrict__ cpart) {
  __shared__ __attribute__((aligned(1024))) unsigned char ring[TN16_NS * TN16_STAGE_B];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int fi = lane & 15, fg = lane >> 4;
  const int gq = gridDim.x >> 3, gr = gridDim.x & 7, bx = blockIdx.x & 7;     // XCD-aware walk: see gemm_tn_kernel
  const int vb = bx * gq + (bx < gr ? bx : gr) + (blockIdx.x >> 3);
  const int tile = vb % (p.tn * p.tk), split = vb / (p.tn * p.tk);
  const int n0 = (tile / p.tk) * TN_T, k0 = (tile % p.tk) * TN_T;
  const int m_beg = split * p.rows_per_split;
  const int m_end = min(a.M, m_beg + p.rows_per_split);
  const int nslab = m_end > m_beg ? (m_end - m_beg + TN16_ROWS - 1) / TN16_ROWS : 0;
  const bool csum_tile = cpart != nullptr && k0 == 0;

  // ---- copy role: wave w issues 4 copies per slab - column group w & 1, row groups 2 (w >> 1) and 2 (w >> 1) + 1, of A and of B ----
  // lane l of a copy writes slot l = (row r8 = l >> 3, position l & 7), i.e. reads chunk c = (l & 7) ^ (((r8 >> 1) & 3) << 1) of its row
  const int r8 = lane >> 3, cch = (lane & 7) ^ (((r8 >> 1) & 3) << 1);
  const int g64 = wid & 1, rg0 = 2 * (wid >> 1);
  const unsigned colA = (unsigned)(n0 + 64 * g64 + 8 * cch) * 2u, colB = (unsigned)(k0 + 64 * g64 + 8 * cch) * 2u;
  const unsigned ldaB = (unsigned)a.lda * 2u, ldbB = (unsigned)a.ldb * 2u;
  const unsigned ring_lds = lds_addr(ring);
  auto issue = [&](int s) {
    const int mb = m_beg + s * TN16_ROWS;
    const unsigned stage = ring_lds + (unsigned)((s % TN16_NS) * TN16_STAGE_B);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int m = mb + 8 * (rg0 + h) + r8;
      m = m < a.M ? m : a.M - 1;            // (rows past the tensor: a valid row, zeroed in LDS before the slab multiplies; slabs past the slice: never read)
      const unsigned blk = (unsigned)((g64 * 4 + rg0 + h) * 1024);
      glds16_asm(a.A, (unsigned)m * ldaB + colA, __builtin_amdgcn_readfirstlane(stage + blk));
      glds16_asm(a.B, (unsigned)m * ldbB + colB, __builtin_amdgcn_readfirstlane(stage + (unsigned)(TN16_ROWS * TN_T * 2) + blk));
    }
  };

  f32x4 acc[4][4], accs[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    accs[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // fragment read addresses inside a stage: lane (fg, i = fi) of the read h supplies row 16 h + 4 fg + (i >> 2), columns 16 t + 4 (i & 3) .. + 3
  // of its operand's column group: block (g, row group 2 h + (fg >> 1)), row r = 4 (fg & 1) + (i >> 2) of the block, chunk c = 2 t + ((i & 3) >> 1)
  const int rr = 4 * (fg & 1) + (fi >> 2), sw = ((rr >> 1) & 3) << 1;
  unsigned fa[4], fb[4];                      // byte offsets of tile t's first read (h = 0); h = 1: + 2 KB (two row groups further)
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const unsigned slot = (unsigned)(8 * rr + ((2 * t + ((fi & 3) >> 1)) ^ sw));
    const unsigned in_blk = slot * 16u + (unsigned)(fi & 1) * 8u + (unsigned)(fg >> 1) * 1024u;
    fa[t] = (unsigned)(wm * 4) * 1024u + in_blk;
    fb[t] = (unsigned)(TN16_ROWS * TN_T * 2) + (unsigned)(wn * 4) * 1024u + in_blk;
  }
  typedef __bf16 tn16_bf16x8 __attribute__((ext_vector_type(8)));
  const tn16_bf16x8 ones = {(__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f};

#pragma unroll
  for (int s = 0; s < TN16_NS - 1; ++s) issue(s);
  for (int s = 0; s < nslab; ++s) {
    // this wave's copies of slab s have landed (4 copies per wave and slab, issued in order: the 4 (NS - 2) behind them may still fly) ...
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (TN16_NS - 2)) : "memory");
    __builtin_amdgcn_s_barrier();             // ... and everybody's have; every wave is done reading slab s - 1
    asm volatile("" ::: "memory");
    issue(s + TN16_NS - 1);                   // into the stage slab s - 1 occupied (always issued: the counted wait above stays a constant)
    unsigned char* const stg = ring + (s % TN16_NS) * TN16_STAGE_B;
    if (m_beg + (s + 1) * TN16_ROWS > a.M) {  // the tensor ends inside this slab (last slice only): rows past it multiply as zeros
      for (int q = tid; q < 512; q += TN_THREADS) {               // A image: 8 blocks x 64 slots
        const int blk = q >> 6, r = 8 * (blk & 3) + ((q & 63) >> 3);
        if (m_beg + s * TN16_ROWS + r >= a.M) *reinterpret_cast<uint4*>(stg + q * 16) = make_uint4(0u, 0u, 0u, 0u);
      }
      __syncthreads();
    }
    tn16_bf16x8 af[4], bf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      tn_s16x4 lo, hi;
      lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_s16x4*)(stg + fa[t]));
      hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_s16x4*)(stg + fa[t] + 2048));
      af[t] = __builtin_bit_cast(tn16_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
      lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_s16x4*)(stg + fb[t]));
      hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_s16x4*)(stg + fb[t] + 2048));
      bf[t] = __builtin_bit_cast(tn16_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) acc[nt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[nt], bf[kt], acc[nt][kt], 0, 0, 0);
    if (csum_tile && wn == 0) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) accs[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[nt], ones, accs[nt], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the copies issued past the slice's end)

  float* pt = part + (long long)split * a.N * a.K;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int k = k0 + wn * 64 + kt * 16 + fi;
#pragma unroll
      for (int r = 0; r < 4; ++r) pt[(long long)(n0 + wm * 64 + nt * 16 + 4 * fg + r) * a.K + k] = acc[nt][kt][r];
    }
  if (csum_tile && wn == 0 && fi == 0) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) cpart[(long long)split * a.N + n0 + wm * 64 + nt * 16 + 4 * fg + r] = accs[nt][r];
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// tn_smallk_kernel (round 4): G[N][16] (+)= sum_m A[m][n] * B'[m][k] for the two filter gradients of the waveform ends - the
// ConvTranspose1d decoder weights (5 heads) and the Conv1d encoder weights: K = 16 taps, N = 256 basis channels, M = sequences x
// frames (256 000 rows at batch 16), B' = 16 consecutive samples of the frame's window (row map: seq * seq_stride + r * ldb).
// On the 128 x 128 MFMA tile core 7/8 of the B tile is padding, the general loader runs one workgroup per CU and the launch took
// 379 us (0.7 TB/s) for what is one pass over A.  Here: exact fp32 on the VALU (16 FMAs per A element: still far below the
// stream), a wave walks rows, lane = 4 consecutive columns (float4, 1 KB coalesced per row), the frame's 16 samples are
// wave-uniform (scalar loads), the 4 waves of a workgroup combine through LDS in a fixed order, row slices combine in
// tn_reduce_kernel as for every other contraction (same partial layout, same workspace plan, no atomics).
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tn_smallk_kernel(const TnArgs a, const TnPlan p, float* __restrict__ part) {
  constexpr int KK = 16;
  __shared__ float red[4][64][4 * KK + 4];        // [wave][lane][4 columns x 16 taps] (+ pad: 272-byte rows)
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int split = blockIdx.x;
  const int m_beg = split * p.rows_per_split;
  const int m_end = min(a.M, m_beg + p.rows_per_split);
  const bool col_ok = 4 * lane < a.N;
  float acc[4][KK];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int k = 0; k < KK; ++k) acc[c][k] = 0.f;
  const float* Ap = a.A + (col_ok ? 4 * lane : 0);
  constexpr int UN = 4;                           // rows in flight per wave
  for (int m0 = m_beg + w; m0 < m_end; m0 += 4 * UN) {
    float4 av[UN];
    float bv[UN][KK];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int m = m0 + 4 * u;                   // wave-uniform
      const int mc = m < m_end ? m : m_beg;
      av[u] = ld4(Ap + (long long)mc * a.lda);
      const int seq = mc / a.rows_out, r = mc - seq * a.rows_out;
      const float* bp = a.B + (long long)seq * a.seq_stride + (long long)r * a.ldb;       // wave-uniform address: scalar loads
#pragma unroll
      for (int k = 0; k < KK; ++k) bv[u][k] = bp[k];
      if (!(m < m_end && col_ok && r < a.rows_valid)) av[u] = zero4();
    }
#pragma unroll
    for (int u = 0; u < UN; ++u)
#pragma unroll
      for (int k = 0; k < KK; ++k) {
        acc[0][k] = fmaf(av[u].x, bv[u][k], acc[0][k]);
        acc[1][k] = fmaf(av[u].y, bv[u][k], acc[1][k]);
        acc[2][k] = fmaf(av[u].z, bv[u][k], acc[2][k]);
        acc[3][k] = fmaf(av[u].w, bv[u][k], acc[3][k]);
      }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int k = 0; k < KK; k += 4) st4(&red[w][lane][c * KK + k], make_float4(acc[c][k], acc[c][k + 1], acc[c][k + 2], acc[c][k + 3]));
  __syncthreads();
  // part[split][n][k], n = 4 * lane + c: 64 consecutive floats per lane; thread t sums element t + 256 j of the 4096
  float* pt = part + (long long)split * a.N * KK;
  for (int e = threadIdx.x; e < 64 * 4 * KK; e += 256) {
    const int l = e / (4 * KK), q = e - l * (4 * KK);
    if (4 * l + q / KK < a.N) pt[e] = (red[0][l][q] + red[1][l][q]) + (red[2][l][q] + red[3][l][q]);
  }
}

// block = 64 output elements x 4 lanes over the slices (fixed order inside a lane, lanes combined in a fixed order)
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ part, const float* __restrict__ cpart, int nsplit,
                                                       int N, int K, float* __restrict__ G, int ldg, int accumulate,
                                                       float* __restrict__ colsum, int colsum_acc) {
  __shared__ float sh[4][64];
  const long long total = (long long)N * K;
  const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const long long i = (long long)blockIdx.x * 64 + el;
  float s = 0.f;
  if (i < total) {
#pragma unroll 4
    for (int sp = sl; sp < nsplit; sp += 4) s += part[(long long)sp * total + i];
  } else if (i < total + N && colsum) {
#pragma unroll 4
    for (int sp = sl; sp < nsplit; sp += 4) s += cpart[(long long)sp * N + (i - total)];
  }
  sh[sl][el] = s;
  __syncthreads();
  if (sl != 0) return;
  s = (sh[0][el] + sh[1][el]) + (sh[2][el] + sh[3][el]);
  if (i < total) {
    const int n = (int)(i / K), k = (int)(i - (long long)n * K);
    float* g = G + (long long)n * ldg + k;
    *g = accumulate ? *g + s : s;
  } else if (i < total + N && colsum) {
    const int n = (int)(i - total);
    colsum[n] = colsum_acc ? colsum[n] + s : s;
  }
}
}  // namespace

size_t tn_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const TnPlan p = tn_plan(M, N, K);
  return align_up((size_t)p.nsplit * ((size_t)N * K + N) * sizeof(float));
}

int launch_gemm_tn(const TnArgs& a, int x3, void* ws, size_t ws_bytes, hipStream_t s_main) {
  if (a.M <= 0) return SEPR_OK;
  if (!a.A || !a.B || !a.G || a.N <= 0 || a.K <= 0 || (a.N % 4) || (a.K % 4) || (a.lda % 4) || (a.ldb % 4)) return SEPR_EINVAL;
  if (a.B2 && ((a.ksplit % 4) || (a.ldb2 % 4))) return SEPR_EINVAL;
  const TnPlan p = tn_plan(a.M, a.N, a.K);
  const size_t need = tn_workspace_bytes(a.M, a.N, a.K);
  if (!ws || ws_bytes < need) return SEPR_EWORKSPACE;
  // contraction + its split-M reduction run on the registered weight-gradient side stream when there is one (sepr_train.h wgrad_stream):
  // nothing in a backward walk consumes their output before the window's join
  hipStream_t s = wgrad_stream(s_main);
  float* part = static_cast<float*>(ws);
  float* cpart = part + (size_t)p.nsplit * a.N * a.K;
  const int grid = p.tn * p.tk * p.nsplit;
  // test switch (tools/probe/tn_fault.py, tests; read once, never set in the product): SEPR_TN_FORCE_GEN=1 routes every launch through the
  // general loader, so the public sepr_linear_wgrad_norm entry exercises it with per-row statistics at any size
  static const bool force_gen = [] { const char* e = getenv("SEPR_TN_FORCE_GEN"); return e && e[0] == '1'; }();
  const bool gen = force_gen || a.rows_out > 0 || a.B2 != nullptr || a.idx != nullptr || a.mask_a != 0 || a.stat_seq != 0;
  if (gen && (a.a16 || a.b16)) return SEPR_EINVAL;      // (checked before the profiling slot opens)
  long long slot = -1;
  const bool timed = prof_begin(SEPR_SITE_WGRAD, s, &slot);
  // the filter gradients of the waveform ends (K = 16 taps, exact arithmetic, windowed-frame row map): one pass over A on the VALU
  static const bool smallk_off = [] {
    const char* e = getenv("SEPR_TN_SMALLK");
    return e && e[0] == '0';
  }();
  if (x3 == 0 && a.K == 16 && a.N <= 256 && a.rows_out > 0 && !a.B2 && !a.idx && !a.stats && !a.mask_a && a.b_shift == 0 && !a.colsum &&
      !a.a16 && !a.b16 && p.tn * p.tk <= 2 && !smallk_off) {
    hipLaunchKernelGGL(tn_smallk_kernel, dim3(p.nsplit), dim3(256), 0, s, a, p, part);
    if (timed) {
      prof_end(slot, 2.0 * (double)a.M * (double)a.N * (double)a.K, s);
      prof_bytes((double)a.M * ((double)a.N * 4.0 + 64.0));
    }
    const long long total_e = (long long)a.N * a.K;
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((int)((total_e + 63) / 64)), dim3(256), 0, s, part, cpart, p.nsplit, a.N, a.K, a.G, a.ldg, a.accumulate,
                       (float*)nullptr, 0);
    SEPR_CHECK_LAUNCH("tn_smallk_kernel");
    return SEPR_OK;
  }
  float* cp = a.colsum ? cpart : nullptr;
  // (Rounds 3-4 ran the general loader at ONE workgroup per CU behind a 16 KB LDS pad: with two co-resident workgroups its bf16
  //  instantiations returned wrong, run-to-run different values in the even columns of the upper half of every B tile.  Round 5 found the
  //  cause - the packed-f32 op_sel fault described in sepr_common.h, triggered by the normalisation's v_pk_mul_f32 op_sel:[0,1] while the
  //  other workgroup's bf16 MFMAs run - pinned the instruction form and removed the pad: tools/probe/tn_fault.py, tests/test_train_gpu.py
  //  test_general_loader_two_workgroups_per_cu.)
#define SEPR_TN_LAUNCH(MD)                                                                                                   \
  do {                                                                                                                       \
    if (gen) hipLaunchKernelGGL((gemm_tn_kernel<MD, true, false>), dim3(grid), dim3(TN_THREADS), 0, s, a, p, part, cp);      \
    else if (a.stats) hipLaunchKernelGGL((gemm_tn_kernel<MD, false, true>), dim3(grid), dim3(TN_THREADS), 0, s, a, p, part, cp); \
    else hipLaunchKernelGGL((gemm_tn_kernel<MD, false, false>), dim3(grid), dim3(TN_THREADS), 0, s, a, p, part, cp);         \
  } while (0)
  // (A packed-row kernel for two bf16 operands - v_perm transpose, v_dot2 column sums, two slabs in flight, 5x fewer staging VALU
  //  instructions, bit-identical G - was built and measured in round 4: 182.55 vs 182.55 utt/s, no gain; git history 'gemm_tn16'.
  //  The contraction is not bound by its staging arithmetic.)
  // two bf16 operands, whole 128 x 128 tiles: the LDS-DMA + transposing-read kernel (gemm_tn16_kernel; SEPR_TN16=0 keeps the register-staged form)
  const bool tn16 = x3 == 2 && !gen && !a.stats && a.a16 && a.b16 && a.N % TN_T == 0 && a.K % TN_T == 0 && a.lda % 8 == 0 && a.ldb % 8 == 0 &&
                    (long long)a.M * a.lda * 2 < (1LL << 32) && (long long)a.M * a.ldb * 2 < (1LL << 32) &&
                    ((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.B & 15) == 0 && knob(SEPR_KNOB_TN16) != 0;
  if (tn16) hipLaunchKernelGGL(gemm_tn16_kernel, dim3(grid), dim3(TN_THREADS), 0, s, a, p, part, cp);
  else if (x3 == 2) SEPR_TN_LAUNCH(2);
  else if (x3) SEPR_TN_LAUNCH(1);
  else SEPR_TN_LAUNCH(0);
#undef SEPR_TN_LAUNCH
  if (timed) {
    prof_end(slot, 2.0 * (double)a.M * (double)a.N * (double)a.K, s);
    // both operands once (bf16 sources: 2 bytes per element) + the statistics: what a contraction over M rows has to read
    prof_bytes((double)a.M * ((double)a.N * (a.a16 ? 2.0 : 4.0) + (double)a.K * (a.b16 ? 2.0 : 4.0) + (a.stats ? 8.0 : 0.0)));
  }
  const long long total = (long long)a.N * a.K + a.N;
  const int rgrid = (int)((total + 63) / 64);
  hipLaunchKernelGGL(tn_reduce_kernel, dim3(rgrid), dim3(256), 0, s, part, cpart, p.nsplit, a.N, a.K, a.G, a.ldg, a.accumulate,
                     a.colsum, a.colsum_accumulate);
  SEPR_CHECK_LAUNCH("gemm_tn_kernel");
  return SEPR_OK;
}

}  // namespace sepr
