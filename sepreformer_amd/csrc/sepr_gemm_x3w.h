// Projection core, bf16x3 / plain-bf16 arithmetic, WIDE form: a workgroup owns a 128-row x 256-column output tile.
//
// Why (profiles/r03_v4_gemm_x3_large_ablation.txt): the 128 x 128 core of sepr_gemm_x3.h is bound by operand movement through
// L2 / Infinity Cache, not by the matrix pipe - per K = 32 step a workgroup pulls 16 KiB of packed weight fragments and 16 KiB
// of fp32 activations for 1.05 MFLOP (33 FLOP per byte), and removing every MFMA changes its time by < 1 %.  Here each of the 4
// waves holds a 128 x 64 accumulator tile (4 column tiles instead of 2), so one staged activation slab (and one bf16 split of
// it on the VALU, and one set of LDS fragment reads) feeds twice the MFMAs: 32 KiB of weights + 16 KiB of activations per
// 2.1 MFLOP = 44 FLOP per byte, half the LDS reads and half the split VALU work per MFMA.  The waves still own DISJOINT weight
// columns, so no weight fragment is fetched twice inside a workgroup and nothing has to be shared through LDS.
//
// What had to change to fit 128 accumulator registers per lane next to the operand pipeline (2 waves per SIMD = 256 VGPRs):
//   * K slabs of 32 (one MFMA K step per slab, 16 floats of activation prefetch per thread instead of 32); LDS row stride
//     96 B - conflict-free for the 16-lane groups of ds_read_b128 (slot = (6 fi + fg) mod 16 is a bijection on each group);
//   * the weight prefetch is split by plane: the hi planes of step s+1 and the lo planes of step s are in flight, never two
//     full fragment sets (the lo planes are only needed by the third MFMA group of a step: 64 MFMAs after their request);
//   * the epilogue runs as two passes of the shared 128 x 128 LDS-staged epilogue (sepr_gemm_epi.h), one per 128-column half.
// Used when the launch has an even number of 128-column tiles (value/gate tiles: 64 + 64) and enough tiles to fill the chip
// (sepr_gemm_x3.hip); everything else - prologues, row maps, epilogues, persistent XCD-aware tile walk - is the narrow core's.
#pragma once
#include "sepr_gemm_x3.h"

#ifndef SEPR_XW_REDERIVE
#define SEPR_XW_REDERIVE 1   // product (0: rounds 4-6 form, 4-20 spilled registers per instantiation; profiles/r06_gcfn_bwd_waits.txt, last block)
#endif
namespace sepr {

constexpr int XW_BKS = 32;                       // K extent of one LDS slab = one MFMA K step
constexpr int XW_LDK = XW_BKS + 16;              // bf16 elements per LDS row: 96 B
constexpr int XW_PLANE = GEMM_BM * XW_LDK;       // elements of one plane of one buffer

template <int PRO, int EPI, int TAG = 0>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_x3w_kernel(const GemmArgs a) {
  constexpr bool DWGLU = (EPI == EPI_DWGLU);
  constexpr bool GLU = (EPI == EPI_GLU) || (EPI == EPI_GLUSAVE) || DWGLU;
  constexpr bool ONE = (TAG & 16) != 0;          // plain bf16 operands (training precision "bf16")
  constexpr bool A16 = (TAG & 32) != 0;          // the A operand is a bf16 tensor
  static_assert(EPI != EPI_LNBWD, "the LayerNorm-backward epilogue needs whole rows in one 128-column tile");
  constexpr int ROWS_OUT = DWGLU ? GEMM_DW_ROWS : GEMM_BM;
  // [buffer][plane hi/lo][128 rows][48] bf16 = 49 152 B; the epilogue re-uses it as a [128][132] fp32 tile (67 584 B)
  constexpr int SMEM_B = (int)sizeof(float) * GEMM_BM * GEMM_HS;
  static_assert(SMEM_B >= (int)sizeof(unsigned short) * 2 * 2 * XW_PLANE, "slab buffers must fit under the epilogue tile");
  __shared__ __attribute__((aligned(16))) unsigned char smem_b[SMEM_B];
  unsigned short* const smem = reinterpret_cast<unsigned short*>(smem_b);

#if SEPR_XW_REDERIVE
  int tid = threadIdx.x;                   // re-derived from an opaque copy at the top of every tile (SEPR_XW_REDERIVE): hipcc then recomputes the per-thread LDS / global
  int lane = tid & 63;                     // offsets per tile instead of hoisting them out of the tile loop and spilling them
  int wn = tid >> 6;
  int fi = lane & 15, fg = lane >> 4;
  int srow = tid >> 1;
  int kh = (tid & 1) * 16;
#else
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wn = tid >> 6;                 // 4 waves side by side: 32 (value/gate: 16 + 16) columns of EACH 128-column half
  const int fi = lane & 15, fg = lane >> 4;
  const int srow = tid >> 1;               // staging: one row per thread pair,
  const int kh = (tid & 1) * 16;           //          half a slab (16 k = 4 float4) per thread
#endif

  const int NB = GLU ? (a.N / 2 + 63) / 64 : (a.N + GEMM_BN - 1) / GEMM_BN;     // 128-column tiles of the narrow core
  const int NB2 = (NB + 1) / 2;                                                 // wide tiles
  const int MB = (a.M + ROWS_OUT - 1) / ROWS_OUT;
  const int ntiles = ((MB + 7) / 8) * 8 * NB2;
  const int nslab = a.K / XW_BKS;
  const int kst = nslab;                   // K steps of the packed weight layout
  const uint4* const Wp = static_cast<const uint4*>(a.Wp);

  unsigned pa = 0u, pa2 = 0u;
  float mka = 0.f, mean = 0.f, rstd = 0.f;
  float4 ra[4];
  unsigned wbase[4] = {0u, 0u, 0u, 0u};    // uint4 index of this wave's four weight tiles at K step 0, plane 0: [half p][nt]

  auto setup = [&](int m0, int nb2) {
    const int m = m0 + srow;
    pa = 0u; pa2 = 0u; mka = 0.f; mean = 0.f; rstd = 0.f;
    if (m >= 0 && m < a.M) {
      long long src = m;
      int seq = 0;
      bool valid = true;
      if (a.rows_out > 0) {
        seq = m / a.rows_out;
        const int r = m - seq * a.rows_out;
        valid = r < a.rows_valid;
        const int rr = valid ? (a.idx ? a.idx[r] : r) : 0;
        src = (long long)seq * a.rows_src + (rr >> a.a_shift);
      }
      if (valid) {
        mka = 1.f;
        pa = (unsigned)(src * a.lda);
        if (PRO == PRO_CAT2) pa2 = (unsigned)((long long)m * a.lda2);
        if (PRO == PRO_NORM) {
          const long long si = a.stat_seq ? seq : m;
          mean = a.stats[2 * si];
          rstd = a.stats[2 * si + 1];
        }
      }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int nb = 2 * nb2 + p;
        int t16;
        bool ok;
        if (GLU) {
          const int c = nb * 64 + wn * 16;
          ok = c < a.N / 2;
          t16 = (nt == 0 ? 0 : (a.N / 2) / 16) + c / 16;
        } else {
          const int c = nb * GEMM_BN + wn * 32 + nt * 16;
          ok = c < a.N;
          t16 = c / 16;
        }
        wbase[2 * p + nt] = ok ? (unsigned)t16 * (unsigned)kst * 128u : 0u;   // columns past N read tile 0: never stored
      }
  };
  auto load_slab = [&](int s) {
    const int k = s * XW_BKS + kh;
    if constexpr (A16) {
      const unsigned short* src16 = reinterpret_cast<const unsigned short*>(a.A) + pa + k;
#pragma unroll
      for (int j = 0; j < 4; j += 2) {
        const uint4 u = *reinterpret_cast<const uint4*>(src16 + 4 * j);      // 8 bf16
        ra[j] = make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                            __uint_as_float(u.y & 0xffff0000u));
        ra[j + 1] = make_float4(__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u), __uint_as_float(u.w << 16),
                                __uint_as_float(u.w & 0xffff0000u));
      }
      return;
    }
    const float* src = a.A + pa + k;
    if (PRO == PRO_CAT2 && k >= a.ksplit) src = a.A2 + pa2 + (k - a.ksplit);
#pragma unroll
    for (int j = 0; j < 4; ++j) ra[j] = ld4(src + 4 * j);
  };
  auto store_slab = [&](int buf) {
#pragma clang fp contract(off)
    unsigned short* hi = smem + (buf * 2 + 0) * XW_PLANE + srow * XW_LDK + kh;
    unsigned short* lo = smem + (buf * 2 + 1) * XW_PLANE + srow * XW_LDK + kh;
    const float sc = (PRO == PRO_NORM) ? rstd * mka : mka;   // invalid rows: exactly zero (pad_signal)
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
      float v[8] = {ra[j].x, ra[j].y, ra[j].z, ra[j].w, ra[j + 1].x, ra[j + 1].y, ra[j + 1].z, ra[j + 1].w};
      bf16x8 h, l;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = (PRO == PRO_NORM) ? (v[e] - mean) * sc : v[e] * sc;     // (the narrow core's arithmetic, bit for bit)
        const __bf16 xh = (__bf16)x;
        h[e] = xh;
        if (!ONE) l[e] = (__bf16)(x - (float)xh);
      }
      *reinterpret_cast<bf16x8*>(hi + 4 * j) = h;
      if (!ONE) *reinterpret_cast<bf16x8*>(lo + 4 * j) = l;
    }
  };
  // weight fragments of K step ks, ONE plane (global, fragment order: one coalesced 1 KiB load per tile)
  auto load_wp = [&](int ks, int plane, uint4 (&w)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) w[t] = Wp[wbase[t] + (unsigned)ks * 128u + (unsigned)plane * 64u + lane];
  };

  f32x4 acc[4][8];
  // MFMAs of one 64-row half of the slab against all four column tiles; PART 0: hi.hi and hi(w).lo(x), PART 1: lo(w).hi(x)
  auto mma_half = [&](const unsigned short* ph, const unsigned short* pl, int half, const uint4 (&wh)[4], const uint4 (&wl)[4]) {
    bf16x8 xh[4], xl[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int off = ((half * 4 + t) * 16 + fi) * XW_LDK + 8 * fg;
      xh[t] = *reinterpret_cast<const bf16x8*>(ph + off);
      if (!ONE) xl[t] = *reinterpret_cast<const bf16x8*>(pl + off);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const bf16x8 wf = *reinterpret_cast<const bf16x8*>(&wh[nt]);
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc[nt][half * 4 + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xh[t], acc[nt][half * 4 + t], 0, 0, 0);
    }
    if constexpr (!ONE) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(&wh[nt]);
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[nt][half * 4 + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xl[t], acc[nt][half * 4 + t], 0, 0, 0);
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(&wl[nt]);
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[nt][half * 4 + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xh[t], acc[nt][half * 4 + t], 0, 0, 0);
      }
    }
  };
  auto decode = [&](int tile, int& m0, int& nb2) -> bool {
    const int q = tile >> 3;
    const int mb = (q / NB2) * 8 + (tile & 7);
    nb2 = q % NB2;
    m0 = mb * ROWS_OUT - (DWGLU ? 1 : 0);
    return mb < MB;
  };
  auto epilogue = [&](const int m0, const int nb2) {
    float* const Hs = reinterpret_cast<float*>(smem_b);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if (p == 1) {
        if (2 * nb2 + 1 >= NB) break;        // odd tile count: the last wide tile has one half only (uniform over the workgroup)
        __syncthreads();                     // pass 0 has been read out of the staging tile
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const int cl = GLU ? (nt * 64 + wn * 16 + 4 * fg) : (wn * 32 + nt * 16 + 4 * fg);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
          const f32x4 c = acc[2 * p + nt][mt];
          st4(Hs + (mt * 16 + fi) * GEMM_HS + cl, make_float4(c[0], c[1], c[2], c[3]));
        }
      }
      __syncthreads();
      epilogue_from_lds<EPI>(a, Hs, m0, 2 * nb2 + p, tid);
    }
  };

  // ---- walk the tiles ---------------------------------------------------------------------------------
  int tile = blockIdx.x;
  int m0 = 0, nb2 = 0;
  while (tile < ntiles && !decode(tile, m0, nb2)) tile += gridDim.x;
  if (tile >= ntiles) return;
  setup(m0, nb2);
  uint4 wh[4], wl[4], whn[4];
  load_wp(0, 0, wh);
  load_slab(0);
  while (true) {
#if SEPR_XW_REDERIVE
    asm volatile("" : "+v"(tid));
    lane = tid & 63; wn = tid >> 6; fi = lane & 15; fg = lane >> 4; srow = tid >> 1; kh = (tid & 1) * 16;
#endif
    store_slab(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
      const int cur = s & 1;
      const unsigned short* ph = smem + (cur * 2 + 0) * XW_PLANE;
      const unsigned short* pl = smem + (cur * 2 + 1) * XW_PLANE;
      // vmcnt retires in order: this step's lo planes (L2) first, then the next step's hi planes (L2), then the next slab (HBM)
      if constexpr (!ONE) load_wp(s, 1, wl);
      if (s + 1 < nslab) {
        load_wp(s + 1, 0, whn);
        load_slab(s + 1);
      }
      mma_half(ph, pl, 0, wh, wl);
      mma_half(ph, pl, 1, wh, wl);
      if (s + 1 < nslab) {
        store_slab(cur ^ 1);
#pragma unroll
        for (int t = 0; t < 4; ++t) wh[t] = whn[t];
      }
      __syncthreads();
    }
    // next tile of this workgroup: its first slab and weight fragments go in flight under the epilogue
    const int m0c = m0, nbc = nb2;
    int nxt = tile + gridDim.x;
    while (nxt < ntiles && !decode(nxt, m0, nb2)) nxt += gridDim.x;
    const bool more = nxt < ntiles;
    if (more) {
      setup(m0, nb2);
      load_wp(0, 0, wh);
      load_slab(0);
    }
    epilogue(m0c, nbc);
    if constexpr (EPI == EPI_RES) {   // (the gate epilogue's instantiation has no registers to spare: its launcher adds a rowstats launch)
      if (a.stats_out && NB == 2) {   // (workgroup-uniform) the tile holds whole output rows: their LayerNorm statistics for the next block
        __syncthreads();              // every wave's rows of the tile are written (workgroup-scope visibility of the global stores)
        const int sub = tid & 15, nf4 = a.N >> 2;
        const float invF = 1.0f / (float)a.N;
#pragma unroll 1
        for (int r = tid >> 4; r < GEMM_BM; r += GEMM_THREADS / 16) {
          const int m = m0c + r;
          if (m < a.M) {
            float mean, rstd;
            rowstats_one<4>(a.Y + (long long)m * a.ldc, nf4, invF, a.stats_eps, sub, mean, rstd);   // (NB == 2: at most 256 columns)
            if (sub == 0) *reinterpret_cast<float2*>(a.stats_out + 2LL * m) = make_float2(mean, rstd);
          }
        }
      }
    }
    if (!more) break;
    tile = nxt;
    __syncthreads();   // the epilogue staged the tile through the slab buffers
  }
}

inline int gemm_tiles_wide(const GemmArgs& a, int epi) {
  const bool glu = (epi == EPI_GLU) || (epi == EPI_GLUSAVE) || (epi == EPI_DWGLU);
  const int rows = (epi == EPI_DWGLU) ? GEMM_DW_ROWS : GEMM_BM;
  const int NB = glu ? (a.N / 2 + 63) / 64 : (a.N + GEMM_BN - 1) / GEMM_BN;
  const int MB = (a.M + rows - 1) / rows;
  return ((MB + 7) / 8) * 8 * ((NB + 1) / 2);
}

}  // namespace sepr
