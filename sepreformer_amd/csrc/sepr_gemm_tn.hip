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

// The split-M reduction of ONE contraction, as a job: tn_reduce_kernel runs it on its own; a contraction kernel runs the job of the PREVIOUS
// contraction in blocks appended to its grid (blockIdx.x >= ngrid; launch_gemm_tn "riding reduction").  Block = 64 output elements x 4 lanes
// over the slices (fixed order inside a lane, lanes combined in a fixed order): the same arithmetic wherever it runs.
struct TnRed {
  const float* part;
  const float* cpart;
  int nsplit, N, K;
  float* G;
  int ldg, accumulate;
  float* colsum;
  int colsum_acc;
  int nblocks;          // 0: no job
};
__device__ __forceinline__ void tn_reduce_block(const TnRed& r, int blk, float* sh /* [4][64] */) {
  const long long total = (long long)r.N * r.K;
  const int el = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const long long i = (long long)blk * 64 + el;
  float s = 0.f;
  if (i < total) {
#pragma unroll 4
    for (int sp = sl; sp < r.nsplit; sp += 4) s += r.part[(long long)sp * total + i];
  } else if (i < total + r.N && r.colsum) {
#pragma unroll 4
    for (int sp = sl; sp < r.nsplit; sp += 4) s += r.cpart[(long long)sp * r.N + (i - total)];
  }
  sh[sl * 64 + el] = s;
  __syncthreads();
  if (sl != 0) return;
  s = (sh[el] + sh[64 + el]) + (sh[128 + el] + sh[192 + el]);
  if (i < total) {
    const int n = (int)(i / r.K), k = (int)(i - (long long)n * r.K);
    float* g = r.G + (long long)n * r.ldg + k;
    *g = r.accumulate ? *g + s : s;
  } else if (i < total + r.N && r.colsum) {
    const int n = (int)(i - total);
    r.colsum[n] = r.colsum_acc ? r.colsum[n] + s : s;
  }
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
                                                               float* __restrict__ cpart, const int ngrid, const TnRed red) {
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
  if ((int)blockIdx.x >= ngrid) {          // riding reduction: the previous contraction's partial tiles (launch_gemm_tn)
    tn_reduce_block(red, (int)blockIdx.x - ngrid, reinterpret_cast<float*>(smem));
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int fi = lane & 15, fg = lane >> 4;
  // XCD-aware walk: hardware block b runs on XCD b % 8, and each XCD has its own L2.  The tiles of one row slice share an operand
  // (B for the tn tiles of a k column, A for the tk tiles of an n row), so they get CONSECUTIVE virtual ids on ONE XCD - the shared
  // slab is fetched from HBM once per slice instead of once per tile (PMC, round 3: 1.67 x the algorithmic bytes before).
  const int gq = ngrid >> 3, gr = ngrid & 7, bx = blockIdx.x & 7;
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
// gemm_tnd_kernel (round 6): the plain-bf16 contraction with its operands staged by LDS-DMA - for plain [M][ld] operands (no row map, no
// normalisation prologue) with N and K multiples of 128: the largest contractions of a training step (the plain-bf16 precision stores dh1,
// g, x^ and dropout(dy) of every GCFN block and the CLA intermediates as bf16: A16 / B16; every other projection hands over fp32 rows).
// gemm_tn_kernel stages its slabs through registers (8-byte loads, widen, transpose on the VALU, 16-byte LDS stores) with ONE slab in
// flight per workgroup, and its load phase and MFMA phase do not overlap (profiles/r03_v3_gemm_tn_ablation.txt): 3.0 TB/s on the
// 256 000-row launches.  Here:
//   * slabs of 32 rows x (128 + 128) columns go global -> LDS by LDS-DMA (16 B per lane; 128 B of a bf16 row / 256 B of an fp32 row per
//     instruction: whole cache lines), into a ring of TND_NS stages - three slabs in flight per workgroup while one multiplies (two bf16
//     operands: 16 KB stages, two workgroups per CU; fp32 operands: 24 / 32 KB stages, one workgroup per CU); the copies are inline asm with
//     counted vmcnt waits and one raw barrier per slab (sepr_common.h glds16_asm);
//   * the MFMA contraction index is the ROW index, so a lane needs 8 rows of one column.  bf16 image: ds_read_b64_tr_b16, gfx950's
//     transposing LDS read, delivers 4 rows x 1 column per lane from a row-major image (two reads per fragment, no VALU at all); the slot
//     of fragment element e of lane group fg is row 16 (e >> 2) + 4 fg + (e & 3).  fp32 image: ds_read2st64_b32 (two rows, 256 B apart,
//     per instruction: four per fragment) and the same round-to-nearest conversion gemm_tn_kernel<2> applies while staging; element e
//     of lane group fg is row 8 fg + e.  (Any bijection works as long as both operands of a launch use the same one - the contraction sums
//     over it - so a mixed launch reads its bf16 image with the fp32 image's map, through plain 2-byte reads.)
//   * LDS images, lane-linear per copy instruction (the copy's per-lane SOURCE address carries the permutation):
//       bf16: [8 rows][64 columns] = 64 slots of 16 B; slot of (row r, chunk c) = 8 r + (c ^ (((r >> 1) & 3) << 1)): the 32 lanes the LDS
//             serves together (rows 0-7 x 32 B of one 16-column tile) hit 16 distinct 16-byte bank windows;
//       fp32: [4 rows][64 columns] = 64 slots; slot of (row r, chunk c) = 16 r + (c ^ 4 (row group bit 1)): lane groups fg = 0 / 1 (row
//             groups 2 fg + h) read bank halves 16 floats apart;
//   * column sums of A (bias gradients): 4 extra MFMAs per slab against an all-ones B fragment in the k0 == 0 tiles.
// Partial tiles and their fixed-order reduction (tn_reduce_kernel) are those of gemm_tn_kernel: same plan, same workspace.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int TND_NS = 4;                  // ring stages (one multiplying, TND_NS - 1 in flight)
constexpr int TND_ROWS = 32;               // rows per slab = one MFMA K step
typedef short tn_s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 tnd_bf16x8 __attribute__((ext_vector_type(8)));
template <bool A32, bool B32>
__global__ __launch_bounds__(TN_THREADS, (A32 || B32) ? 1 : 2) void gemm_tnd_kernel(const TnArgs a, const TnPlan p, float* __restrict__ part,
                                                                                float* __restrict__ cpart, const int ngrid, const TnRed red) {
  constexpr bool MIXED = A32 != B32;
  constexpr int IMG_A = TND_ROWS * TN_T * (A32 ? 4 : 2), IMG_B = TND_ROWS * TN_T * (B32 ? 4 : 2), STAGE_B = IMG_A + IMG_B;
  constexpr int NIA = A32 ? 4 : 2, NIB = B32 ? 4 : 2, NI = NIA + NIB;      // copies per wave and slab
  __shared__ __attribute__((aligned(1024))) unsigned char ring[TND_NS * STAGE_B];
  if ((int)blockIdx.x >= ngrid) {          // riding reduction: the previous contraction's partial tiles (launch_gemm_tn)
    tn_reduce_block(red, (int)blockIdx.x - ngrid, reinterpret_cast<float*>(ring));
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  const int fi = lane & 15, fg = lane >> 4;
  const int gq = ngrid >> 3, gr = ngrid & 7, bx = blockIdx.x & 7;             // XCD-aware walk: see gemm_tn_kernel
  const int vb = bx * gq + (bx < gr ? bx : gr) + (blockIdx.x >> 3);
  const int tile = vb % (p.tn * p.tk), split = vb / (p.tn * p.tk);
  const int n0 = (tile / p.tk) * TN_T, k0 = (tile % p.tk) * TN_T;
  const int m_beg = split * p.rows_per_split;
  const int m_end = min(a.M, m_beg + p.rows_per_split);
  const int nslab = m_end > m_beg ? (m_end - m_beg + TND_ROWS - 1) / TND_ROWS : 0;
  const bool csum_tile = cpart != nullptr && k0 == 0;

  // ---- copy role: wave w copies column group w & 1 (64 columns) of both operands: bf16 - row groups (8 rows) 2 (w >> 1) + {0, 1};
  //      fp32 - row groups (4 rows) 4 (w >> 1) + {0 .. 3}.  Lane l writes slot l of its copy's 1 KB block ----
  const int g64 = wid & 1, wq = wid >> 1;
  const int r8 = lane >> 3, c8 = (lane & 7) ^ (((r8 >> 1) & 3) << 1);       // bf16 block: row, source chunk (8 columns)
  const int r4 = lane >> 4, p4 = lane & 15;                                  // fp32 block: row, slot position (source chunk = p4 ^ 4 (rg bit 1))
  const unsigned ldaB = (unsigned)a.lda * (A32 ? 4u : 2u), ldbB = (unsigned)a.ldb * (B32 ? 4u : 2u);
  const unsigned ring_lds = lds_addr(ring);
  auto copy16 = [&](const float* src, unsigned ldB, int col0, unsigned img, int mb) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int rg = 2 * wq + h;
      int m = mb + 8 * rg + r8;
      m = m < a.M ? m : a.M - 1;            // (rows past the tensor: a valid row, zeroed in LDS before the slab multiplies; slabs past the slice: never read)
      glds16_asm(src, (unsigned)m * ldB + (unsigned)(col0 + 64 * g64 + 8 * c8) * 2u, __builtin_amdgcn_readfirstlane(img + (unsigned)((g64 * 4 + rg) * 1024)));
    }
  };
  auto copy32 = [&](const float* src, unsigned ldB, int col0, unsigned img, int mb) {
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int rg = 4 * wq + h;
      int m = mb + 4 * rg + r4;
      m = m < a.M ? m : a.M - 1;
      const int c4 = p4 ^ (((rg >> 1) & 1) << 2);
      glds16_asm(src, (unsigned)m * ldB + (unsigned)(col0 + 64 * g64 + 4 * c4) * 4u, __builtin_amdgcn_readfirstlane(img + (unsigned)((g64 * 8 + rg) * 1024)));
    }
  };
  auto issue = [&](int s) {
    const int mb = m_beg + s * TND_ROWS;
    const unsigned stage = ring_lds + (unsigned)((s % TND_NS) * STAGE_B);
    if constexpr (A32) copy32(a.A, ldaB, n0, stage, mb); else copy16(a.A, ldaB, n0, stage, mb);
    if constexpr (B32) copy32(a.B, ldbB, k0, stage + IMG_A, mb); else copy16(a.B, ldbB, k0, stage + IMG_A, mb);
  };

  f32x4 acc[4][4], accs[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    accs[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // ---- fragment reads ----
  // bf16 image, transposing read h of tile t: lane (fg, i = fi) supplies row 16 h + 4 fg + (i >> 2), columns 16 t + 4 (i & 3) .. + 3 of its
  // wave's column group g: block (g, row group 2 h + (fg >> 1)), row 4 (fg & 1) + (i >> 2) of the block, chunk 2 t + ((i & 3) >> 1)
  const int rr = 4 * (fg & 1) + (fi >> 2), sw = ((rr >> 1) & 3) << 1;
  auto frag16 = [&](const unsigned char* img, int g, int t) -> tnd_bf16x8 {
    const unsigned slot = (unsigned)(8 * rr + ((2 * t + ((fi & 3) >> 1)) ^ sw));
    const unsigned char* q = img + (unsigned)(g * 4 + (fg >> 1)) * 1024u + slot * 16u + (unsigned)(fi & 1) * 8u;
    const tn_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_s16x4*)(q));
    const tn_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_s16x4*)(q + 2048));
    return __builtin_bit_cast(tnd_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
  };
  // fp32 image: element e = row 8 fg + e, column 16 t + fi of column group g: block (g, row group 2 fg + (e >> 2)), row e & 3, chunk 4 t + (fi >> 2)
  auto frag32 = [&](const unsigned char* img, int g, int t) -> tnd_bf16x8 {
    const float* q = reinterpret_cast<const float*>(img + (unsigned)(g * 8 + 2 * fg) * 1024u + (unsigned)(((4 * t + (fi >> 2)) ^ ((fg & 1) << 2)) * 16 + (fi & 3) * 4));
    tnd_bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (__bf16)q[(e >> 2) * 256 + (e & 3) * 64];      // (row stride 256 B, row-group stride 1 KB)
    return v;
  };
  // bf16 image through the fp32 image's row map (mixed launches): 2-byte reads, rows 8 fg + e
  auto frag16m = [&](const unsigned char* img, int g, int t) -> tnd_bf16x8 {
    tnd_bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int r = 8 * fg + e, rb = r & 7;          // block (g, r >> 3), row rb, chunk c = 2 t + (fi >> 3), element fi & 7
      const unsigned slot = (unsigned)(8 * rb + ((2 * t + (fi >> 3)) ^ (((rb >> 1) & 3) << 1)));
      v[e] = *reinterpret_cast<const __bf16*>(img + (unsigned)(g * 4 + (r >> 3)) * 1024u + slot * 16u + (unsigned)(fi & 7) * 2u);
    }
    return v;
  };
  const tnd_bf16x8 ones = {(__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f, (__bf16)1.f};

#pragma unroll
  for (int s = 0; s < TND_NS - 1; ++s) issue(s);
  for (int s = 0; s < nslab; ++s) {
    // this wave's copies of slab s have landed (NI copies per wave and slab, issued in order: the NI (NS - 2) behind them may still fly) ...
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI * (TND_NS - 2)) : "memory");
    __builtin_amdgcn_s_barrier();             // ... and everybody's have; every wave is done reading slab s - 1
    asm volatile("" ::: "memory");
    issue(s + TND_NS - 1);                    // into the stage slab s - 1 occupied (always issued: the counted wait above stays a constant)
    unsigned char* const stg = ring + (s % TND_NS) * STAGE_B;
    if (m_beg + (s + 1) * TND_ROWS > a.M) {   // the tensor ends inside this slab (last slice only): rows of A past it multiply as zeros
      for (int q = tid; q < IMG_A / 16; q += TN_THREADS) {
        const int blk = q >> 6, r = A32 ? 4 * (blk & 7) + ((q & 63) >> 4) : 8 * (blk & 3) + ((q & 63) >> 3);
        if (m_beg + s * TND_ROWS + r >= a.M) *reinterpret_cast<uint4*>(stg + q * 16) = make_uint4(0u, 0u, 0u, 0u);
      }
      __syncthreads();
    }
    tnd_bf16x8 af[4], bf[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if constexpr (A32) af[t] = frag32(stg, wm, t);
      else if constexpr (MIXED) af[t] = frag16m(stg, wm, t);
      else af[t] = frag16(stg, wm, t);
      if constexpr (B32) bf[t] = frag32(stg + IMG_A, wn, t);
      else if constexpr (MIXED) bf[t] = frag16m(stg + IMG_A, wn, t);
      else bf[t] = frag16(stg + IMG_A, wn, t);
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

__global__ __launch_bounds__(256) void tn_reduce_kernel(const TnRed r) {
  __shared__ float sh[4 * 64];
  tn_reduce_block(r, (int)blockIdx.x, sh);
}
}  // namespace

size_t tn_workspace_bytes(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const TnPlan p = tn_plan(M, N, K);
  return align_up((size_t)p.nsplit * ((size_t)N * K + N) * sizeof(float));
}

// ---- riding reduction (round 6) -------------------------------------------------------------------------------------------------------
// A training step runs ~300 contractions, each followed by its split-M reduction: a 5-7 us launch of which ~5 us are launch latency and the
// drain of the contraction's tail.  While a deferred-finisher window is open (sepr_train.h) nothing reads a contraction's G / colsum before
// the window's flush when they live in the window's arena, so the reduction need not run right behind its contraction: its job is kept
// PENDING and runs in blocks appended to the grid of the NEXT contraction on the same stream (stream order: the partial tiles are complete;
// the extra blocks are over long before the contraction's own) - or on its own at the flush.  The partial tiles of such a job live in one half
// of a caller-provided double buffer (sepr_train_defer_parts) instead of the block's workspace, which the next block call carves differently:
// contraction i writes half i % 2 while the riding blocks read half (i - 1) % 2.  Same reduction arithmetic, same results.
namespace {
struct TnPending {
  bool valid = false;
  TnRed job;
  hipStream_t stream = nullptr;
};
struct TnParts {
  char* base = nullptr;
  size_t half = 0;
  int toggle = 0;
  bool used = false;
  hipStream_t stream = nullptr;     // the stream of the window's riding launches (another stream: that launch keeps the immediate form)
};
thread_local TnPending g_pend;
thread_local TnParts g_parts;
const TnRed kNoRed = {nullptr, nullptr, 0, 0, 0, nullptr, 0, 0, nullptr, 0, 0};
}  // namespace
int tn_flush_pending() {
  if (!g_pend.valid) return SEPR_OK;
  g_pend.valid = false;
  hipLaunchKernelGGL(tn_reduce_kernel, dim3(g_pend.job.nblocks), dim3(256), 0, g_pend.stream, g_pend.job);
  SEPR_CHECK_LAUNCH("tn_reduce_kernel");
  return SEPR_OK;
}
void tn_parts_set(void* parts, size_t bytes) {
  g_parts.base = static_cast<char*>(parts);
  g_parts.half = parts ? (bytes / 2) / 256 * 256 : 0;
  g_parts.toggle = 0;
  g_parts.used = false;
  g_parts.stream = nullptr;
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
  // this contraction's reduction can wait for the next contraction (see above): outputs in the window's arena, partial tiles in a buffer half
  bool defer_red = g_parts.base != nullptr && s == s_main && need <= g_parts.half && fin_defers(a.G) && (!a.colsum || fin_defers(a.colsum)) &&
                   (!g_parts.used || g_parts.stream == s);
  // a pending job rides on this launch when it was issued on this stream; otherwise it runs on its own now
  if (g_pend.valid && g_pend.stream != s) SEPR_TRY(tn_flush_pending());
  float* part = static_cast<float*>(ws);
  if (defer_red) {
    part = reinterpret_cast<float*>(g_parts.base + (size_t)g_parts.toggle * g_parts.half);
    g_parts.toggle ^= 1;
    g_parts.used = true;
    g_parts.stream = s;
  }
  float* cpart = part + (size_t)p.nsplit * a.N * a.K;
  const int grid = p.tn * p.tk * p.nsplit;
  // test switch (tools/probe/tn_fault.py, tests; read once, never set in the product): SEPR_TN_FORCE_GEN=1 routes every launch through the
  // general loader, so the public sepr_linear_wgrad_norm entry exercises it with per-row statistics at any size
  static const bool force_gen = [] { const char* e = getenv("SEPR_TN_FORCE_GEN"); return e && e[0] == '1'; }();
  const bool gen = force_gen || a.rows_out > 0 || a.B2 != nullptr || a.idx != nullptr || a.mask_a != 0 || a.stat_seq != 0;
  if (gen && (a.a16 || a.b16)) return SEPR_EINVAL;      // (checked before the profiling slot opens)
  // the filter gradients of the waveform ends (K = 16 taps, exact arithmetic, windowed-frame row map): one pass over A on the VALU
  static const bool smallk_off = [] {
    const char* e = getenv("SEPR_TN_SMALLK");
    return e && e[0] == '0';
  }();
  const bool smallk = x3 == 0 && a.K == 16 && a.N <= 256 && a.rows_out > 0 && !a.B2 && !a.idx && !a.stats && !a.mask_a && a.b_shift == 0 && !a.colsum &&
                      !a.a16 && !a.b16 && p.tn * p.tk <= 2 && !smallk_off;
  if (smallk) SEPR_TRY(tn_flush_pending());             // (its kernel carries no riding blocks)
  long long slot = -1;
  const bool timed = prof_begin(SEPR_SITE_WGRAD, s, &slot);
  TnRed mine;                                           // this contraction's reduction
  mine.part = part; mine.cpart = cpart; mine.nsplit = p.nsplit; mine.N = a.N; mine.K = a.K;
  mine.G = a.G; mine.ldg = a.ldg; mine.accumulate = a.accumulate;
  mine.colsum = a.colsum; mine.colsum_acc = a.colsum_accumulate;
  mine.nblocks = (int)(((long long)a.N * a.K + a.N + 63) / 64);
  if (smallk) {
    hipLaunchKernelGGL(tn_smallk_kernel, dim3(p.nsplit), dim3(256), 0, s, a, p, part);
    if (timed) {
      prof_end(slot, 2.0 * (double)a.M * (double)a.N * (double)a.K, s);
      prof_bytes((double)a.M * ((double)a.N * 4.0 + 64.0));
    }
    mine.colsum = nullptr;
    mine.nblocks = (int)(((long long)a.N * a.K + 63) / 64);
    if (defer_red) {
      g_pend.valid = true; g_pend.job = mine; g_pend.stream = s;
    } else {
      hipLaunchKernelGGL(tn_reduce_kernel, dim3(mine.nblocks), dim3(256), 0, s, mine);
    }
    SEPR_CHECK_LAUNCH("tn_smallk_kernel");
    return SEPR_OK;
  }
  float* cp = a.colsum ? cpart : nullptr;
  const TnRed red = g_pend.valid ? g_pend.job : kNoRed;
  const int gridr = grid + red.nblocks;                 // the pending job's blocks ride behind this contraction's
  g_pend.valid = false;
  // (Rounds 3-4 ran the general loader at ONE workgroup per CU behind a 16 KB LDS pad: with two co-resident workgroups its bf16
  //  instantiations returned wrong, run-to-run different values in the even columns of the upper half of every B tile.  Round 5 found the
  //  cause - the packed-f32 op_sel fault described in sepr_common.h, triggered by the normalisation's v_pk_mul_f32 op_sel:[0,1] while the
  //  other workgroup's bf16 MFMAs run - pinned the instruction form and removed the pad: tools/probe/tn_fault.py, tests/test_train_gpu.py
  //  test_general_loader_two_workgroups_per_cu.)
#define SEPR_TN_LAUNCH(MD)                                                                                                   \
  do {                                                                                                                       \
    if (gen) hipLaunchKernelGGL((gemm_tn_kernel<MD, true, false>), dim3(gridr), dim3(TN_THREADS), 0, s, a, p, part, cp, grid, red);      \
    else if (a.stats) hipLaunchKernelGGL((gemm_tn_kernel<MD, false, true>), dim3(gridr), dim3(TN_THREADS), 0, s, a, p, part, cp, grid, red); \
    else hipLaunchKernelGGL((gemm_tn_kernel<MD, false, false>), dim3(gridr), dim3(TN_THREADS), 0, s, a, p, part, cp, grid, red);         \
  } while (0)
  // (A packed-row kernel for two bf16 operands - v_perm transpose, v_dot2 column sums, two slabs in flight, 5x fewer staging VALU
  //  instructions, bit-identical G - was built and measured in round 4: 182.55 vs 182.55 utt/s, no gain; git history 'gemm_tn16'.
  //  The contraction is not bound by its staging arithmetic.)
  // plain operands, whole 128 x 128 tiles, plain-bf16 arithmetic: the LDS-DMA kernel (gemm_tnd_kernel; SEPR_TN16=0 keeps the register-staged form,
  // SEPR_TN16=1 (default) takes launches with two bf16 operands only, 2 fp32 / mixed operands too - measured not faster, profiles/r06_wgrad_dma.txt)
  const int tnd_knob = knob(SEPR_KNOB_TN16);
  const bool tnd = x3 == 2 && !gen && !a.stats && a.N % TN_T == 0 && a.K % TN_T == 0 && a.lda % (a.a16 ? 8 : 4) == 0 && a.ldb % (a.b16 ? 8 : 4) == 0 &&
                   (long long)a.M * a.lda * (a.a16 ? 2 : 4) < (1LL << 32) && (long long)a.M * a.ldb * (a.b16 ? 2 : 4) < (1LL << 32) &&
                   ((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.B & 15) == 0 && (tnd_knob >= 2 || (tnd_knob == 1 && a.a16 && a.b16));
  if (tnd) {
    if (a.a16 && a.b16) hipLaunchKernelGGL((gemm_tnd_kernel<false, false>), dim3(gridr), dim3(TN_THREADS), 0, s, a, p, part, cp, grid, red);
    else if (a.a16) hipLaunchKernelGGL((gemm_tnd_kernel<false, true>), dim3(gridr), dim3(TN_THREADS), 0, s, a, p, part, cp, grid, red);
    else if (a.b16) hipLaunchKernelGGL((gemm_tnd_kernel<true, false>), dim3(gridr), dim3(TN_THREADS), 0, s, a, p, part, cp, grid, red);
    else hipLaunchKernelGGL((gemm_tnd_kernel<true, true>), dim3(gridr), dim3(TN_THREADS), 0, s, a, p, part, cp, grid, red);
  }
  else if (x3 == 2) SEPR_TN_LAUNCH(2);
  else if (x3) SEPR_TN_LAUNCH(1);
  else SEPR_TN_LAUNCH(0);
#undef SEPR_TN_LAUNCH
  if (timed) {
    prof_end(slot, 2.0 * (double)a.M * (double)a.N * (double)a.K, s);
    // both operands once (bf16 sources: 2 bytes per element) + the statistics: what a contraction over M rows has to read
    prof_bytes((double)a.M * ((double)a.N * (a.a16 ? 2.0 : 4.0) + (double)a.K * (a.b16 ? 2.0 : 4.0) + (a.stats ? 8.0 : 0.0)));
  }
  if (defer_red) {
    g_pend.valid = true; g_pend.job = mine; g_pend.stream = s;
  } else {
    hipLaunchKernelGGL(tn_reduce_kernel, dim3(mine.nblocks), dim3(256), 0, s, mine);
  }
  SEPR_CHECK_LAUNCH("gemm_tn_kernel");
  return SEPR_OK;
}

}  // namespace sepr
