// Projection core, second precision: "bf16x3" split-fp32 on the bf16 MFMA.
//
// Why: the exact f32 MFMA runs at 1/16 of the bf16 rate (157 TFLOP/s chip peak), which caps the whole
// separator at ~500 utt/s however good the kernel is.  Splitting every fp32 operand into two bf16 parts,
//     x = x_hi + x_lo,  x_hi = bf16(x),  x_lo = bf16(x - x_hi)        (|x - x_hi - x_lo| <= 2^-17 |x|)
// and computing  x.w ~= x_hi.w_hi + x_hi.w_lo + x_lo.w_hi  (fp32 accumulate; the dropped lo.lo term is
// ~2^-16 relative) costs 3 bf16 MFMAs instead of 8 f32 MFMAs per 16x16x32 block: 3/16 of the matrix time.
// Measured agreement with the fp32 oracle is ~100 dB end to end (SURVEY.md section 7 probe: 102-104 dB),
// well inside the 80 dB / 1e-3 dB SI-SNR gates; plain bf16 operands (47 dB) are not.
// With the matrix pipe 5x cheaper the projections become HBM/LDS-bound, so the structure differs from
// sepr_gemm.h where it matters:
//   * weights never touch LDS: they are pre-split on the host into bf16 hi/lo planes stored in MFMA
//     fragment order (pack.py::pack_x3), so a wave fetches each 16x32 weight fragment as ONE coalesced
//     1 KiB global load straight into VGPRs (L2-resident: <= 0.8 MB per matrix), one K step ahead;
//   * activations are staged once per 64-wide K slab: global fp32 -> registers -> normalise -> split ->
//     two bf16 LDS planes (row stride 160 B: conflict-free 16-byte fragment reads), double-buffered;
//   * wave tile is 128 rows x 32 columns (4 waves side by side), so the 4 waves read disjoint weight
//     fragments and share the activation planes;
//   * LayerNorm's gamma / beta are folded into the packed weights / bias on the host, the prologue only
//     applies (x - mean) * rstd;
//   * same persistent tile walk, same LDS-staged row-contiguous epilogues (sepr_gemm_epi.h).
#pragma once
#include "sepr_gemm_epi.h"

namespace sepr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef SEPR_ABL_NOLOAD
#define SEPR_ABL_NOLOAD 0
#endif
#ifndef SEPR_X3_DEEP16
#define SEPR_X3_DEEP16 1   // fp32 source rows on the single-plane arithmetic (training precision "bf16"): two slabs in flight (0: one; tools/variants.mk x3deep0)
#endif
#ifndef SEPR_X3_RAW16
#define SEPR_X3_RAW16 1   // bf16 source rows on the single-plane arithmetic: raw staging, two slabs in flight (0: the widened one-slab form; tools/variants.mk x3raw0)
#endif
#ifndef SEPR_ABL_NOSTORE
#define SEPR_ABL_NOSTORE 0
#endif
#ifndef SEPR_ABL_NOW
#define SEPR_ABL_NOW 0
#endif
#ifndef SEPR_ABL_NOMMA
#define SEPR_ABL_NOMMA 0
#endif
constexpr int X3_BKS = 64;                  // K extent of one LDS slab
#ifndef SEPR_X3_LDK_PAD
#define SEPR_X3_LDK_PAD 16   // 16: 160-byte rows (rounds 1-3); 8: 144-byte rows
#endif
constexpr int X3_LDK = X3_BKS + SEPR_X3_LDK_PAD;   // bf16 elements per LDS row
constexpr int X3_PLANE = GEMM_BM * X3_LDK;  // elements of one plane of one buffer

template <int PRO, int EPI, int TAG = 0>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_x3_kernel(const GemmArgs a) {
  constexpr bool DWGLU = (EPI == EPI_DWGLU);
  constexpr bool GLU = (EPI == EPI_GLU) || (EPI == EPI_GLUSAVE) || DWGLU;
  // TAG bit 4: plain bf16 operands ("bf16" training precision): activations rounded to bf16 (hi plane only), weights'
  // hi plane only, ONE MFMA per product instead of three; everything else (staging, tile walk, epilogues) is shared
  constexpr bool ONE = (TAG & 16) != 0;
  // TAG bit 5: the A operand is a bf16 tensor (lda in elements; PRO_PLAIN, no row map): the plain-bf16 training precision keeps
  // the [rows, 6F] input-gradient operand of the GCFN in bf16 (sepr_gcfn_bwd_fused.hip)
  constexpr bool A16 = (TAG & 32) != 0;
  constexpr int ROWS_OUT = DWGLU ? GEMM_DW_ROWS : GEMM_BM;
  // [buffer][plane hi/lo][128 rows][80] bf16 = 81 920 B; the epilogue re-uses it as a [128][132] fp32 tile
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * 2 * X3_PLANE];
  static_assert(sizeof(unsigned short) * 2 * 2 * X3_PLANE >= sizeof(float) * GEMM_BM * GEMM_HS, "epilogue tile must fit");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wn = tid >> 6;                 // 4 waves side by side: 32 output columns each
  const int fi = lane & 15, fg = lane >> 4;
  const int srow = tid >> 1;               // staging: one row per thread pair,
  const int kh = (tid & 1) * 32;           //          half a slab (32 k = 8 float4) per thread

  const int NB = GLU ? (a.N / 2 + 63) / 64 : (a.N + GEMM_BN - 1) / GEMM_BN;
  const int MB = (a.M + ROWS_OUT - 1) / ROWS_OUT;
  const int ntiles = ((MB + 7) / 8) * 8 * NB;
  const int nslab = a.K / X3_BKS;
  const int kst = a.K / 32;                // K steps of the packed weight layout
  const uint4* const Wp = static_cast<const uint4*>(a.Wp);

  // ---- staging state of the tile being loaded (one row per thread) -----------------------------------
  unsigned pa = 0u, pa2 = 0u;
  float mka = 0.f, mean = 0.f, rstd = 0.f;
  float4 ra[8];
  unsigned wbase[2] = {0u, 0u};            // uint4 index of this wave's two weight tiles at K step 0, plane 0

  auto setup = [&](int m0, int nb) {
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
    // 16-column weight tiles of this wave (columns past N read tile 0: never stored)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
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
      wbase[nt] = ok ? (unsigned)t16 * (unsigned)kst * 128u : 0u;   // 2 planes x 64 lanes per K step
    }
  };
  auto load_slab_to = [&](int s, float4 (&ra)[8]) {
#if SEPR_ABL_NOLOAD
    if (s > 0 || blockIdx.x != (unsigned)a.M) return;   // timing ablation
#endif
    const int k = s * X3_BKS + kh;
    if constexpr (A16) {
      const unsigned short* src16 = reinterpret_cast<const unsigned short*>(a.A) + pa + k;
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
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
    for (int j = 0; j < 8; ++j) ra[j] = ld4(src + 4 * j);
  };
  auto load_slab = [&](int s) { load_slab_to(s, ra); };
  // RAW (bf16 source rows on the single-plane arithmetic: the [rows, 6F] input-gradient operand of the plain-bf16 GCFN backward, K = 6F = 12
  // slabs): the slab's bf16 values ARE the LDS plane - no widening, no conversion - so a slab is 4 x 16 bytes per thread and TWO slabs are
  // kept in flight in the registers ONE widened slab took (round 6; SEPR_X3_RAW16=0: the widened one-slab form)
  constexpr bool RAW = A16 && ONE && PRO == PRO_PLAIN && (SEPR_X3_RAW16 != 0);
  // DEEP (every other single-plane instantiation, i.e. the plain-bf16 TRAINING precision only): the same two-slabs-in-flight schedule with two
  // fp32 register sets (+32 registers), conversion at the LDS store as before (SEPR_X3_DEEP16=0: one slab; tools/variants.mk x3deep0)
  constexpr bool DEEP = ONE && !RAW && (SEPR_X3_DEEP16 != 0);
  [[maybe_unused]] float4 rb[8];
  [[maybe_unused]] uint4 rq0[4], rq1[4];      // (two named sets: a runtime-indexed array would live in scratch)
  [[maybe_unused]] auto load_raw = [&](int s, uint4 (&r)[4]) {
    const unsigned short* src16 = reinterpret_cast<const unsigned short*>(a.A) + pa + (s * X3_BKS + kh);
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = *reinterpret_cast<const uint4*>(src16 + 8 * j);
  };
  [[maybe_unused]] auto store_raw = [&](int buf, const uint4 (&r)[4]) {
    unsigned short* hi = smem + (buf * 2 + 0) * X3_PLANE + srow * X3_LDK + kh;
    const bool ok = mka != 0.f;                            // rows past M / outside the row map: exactly zero
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<uint4*>(hi + 8 * j) = ok ? r[j] : make_uint4(0u, 0u, 0u, 0u);
  };
  auto store_slab_from = [&](int buf, const float4 (&ra)[8]) {
#pragma clang fp contract(off)
    unsigned short* hi = smem + (buf * 2 + 0) * X3_PLANE + srow * X3_LDK + kh;
    unsigned short* lo = smem + (buf * 2 + 1) * X3_PLANE + srow * X3_LDK + kh;
    const float sc = (PRO == PRO_NORM) ? rstd * mka : mka;   // invalid rows: exactly zero (pad_signal)
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      float v[8] = {ra[j].x, ra[j].y, ra[j].z, ra[j].w, ra[j + 1].x, ra[j + 1].y, ra[j + 1].z, ra[j + 1].w};
      bf16x8 h, l;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float x = (PRO == PRO_NORM) ? (v[e] - mean) * sc : v[e] * sc;
        const __bf16 xh = (__bf16)x;
        h[e] = xh;
        if (!ONE) l[e] = (__bf16)(x - (float)xh);
      }
      *reinterpret_cast<bf16x8*>(hi + 4 * j) = h;
      if (!ONE) *reinterpret_cast<bf16x8*>(lo + 4 * j) = l;
    }
  };
  auto store_slab = [&](int buf) { store_slab_from(buf, ra); };
  // weight fragments of K step ks (global, fragment order: one coalesced 1 KiB load per tile and plane)
  auto load_w = [&](int ks, uint4 (&wh)[2], uint4 (&wl)[2]) {
#if SEPR_ABL_NOW
    if (ks > 0 || blockIdx.x != (unsigned)a.M) return;  // timing ablation
#endif
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const uint4* p = Wp + wbase[nt] + (unsigned)ks * 128u + lane;
      wh[nt] = p[0];
      if (!ONE) wl[nt] = p[64];
    }
  };

  f32x4 acc[2][8];
  auto mma_half = [&](const unsigned short* ph, const unsigned short* pl, int kk, int half, const uint4 (&wh)[2],
                      const uint4 (&wl)[2]) {
#if SEPR_ABL_NOMMA
    if (blockIdx.x != (unsigned)a.M) return;            // timing ablation
#endif
    bf16x8 xh[4], xl[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int off = ((half * 4 + t) * 16 + fi) * X3_LDK + kk * 32 + 8 * fg;
      xh[t] = *reinterpret_cast<const bf16x8*>(ph + off);
      if (!ONE) xl[t] = *reinterpret_cast<const bf16x8*>(pl + off);
    }
    bf16x8 wfh[2], wfl[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      wfh[nt] = *reinterpret_cast<const bf16x8*>(&wh[nt]);
      if (!ONE) wfl[nt] = *reinterpret_cast<const bf16x8*>(&wl[nt]);
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        acc[nt][half * 4 + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfh[nt], xh[t], acc[nt][half * 4 + t], 0, 0, 0);
    if constexpr (!ONE) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[nt][half * 4 + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfh[nt], xl[t], acc[nt][half * 4 + t], 0, 0, 0);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[nt][half * 4 + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfl[nt], xh[t], acc[nt][half * 4 + t], 0, 0, 0);
    }
  };
  auto decode = [&](int tile, int& m0, int& nb) -> bool {
    const int q = tile >> 3;
    const int mb = (q / NB) * 8 + (tile & 7);
    nb = q % NB;
    m0 = mb * ROWS_OUT - (DWGLU ? 1 : 0);
    return mb < MB;
  };
  auto epilogue = [&](const int m0, const int nb) {
#if SEPR_ABL_NOSTORE
    {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(acc[i][j]));
      return;
    }
#endif
    float* const Hs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      const int cl = GLU ? (nt * 64 + wn * 16 + 4 * fg) : (wn * 32 + nt * 16 + 4 * fg);
#pragma unroll
      for (int mt = 0; mt < 8; ++mt) {
        const f32x4 c = acc[nt][mt];
        st4(Hs + (mt * 16 + fi) * GEMM_HS + cl, make_float4(c[0], c[1], c[2], c[3]));
      }
    }
    __syncthreads();
    epilogue_from_lds<EPI>(a, Hs, m0, nb, tid);
  };

  // ---- walk the tiles ---------------------------------------------------------------------------------
  int tile = blockIdx.x;
  int m0 = 0, nb = 0;
  while (tile < ntiles && !decode(tile, m0, nb)) tile += gridDim.x;
  if (tile >= ntiles) return;
  setup(m0, nb);
  uint4 wh[2], wl[2], wh2[2], wl2[2];
  if constexpr (RAW || DEEP) {
    // two slabs in flight: slab s + 2 is requested before slab s multiplies and written to LDS two barriers later.  The weight fragments of BOTH K
    // steps of the next slab are requested in front of it (vmcnt retires in order: waiting for them must not wait for the slab behind them)
    uint4 wh3[2], wh4[2];
    load_w(0, wh, wl);
    load_w(1, wh2, wl2);
    auto ld2 = [&](int s, auto& set) {
      if constexpr (RAW) load_raw(s, set); else load_slab_to(s, set);
    };
    auto st2 = [&](int buf, const auto& set) {
      if constexpr (RAW) store_raw(buf, set); else store_slab_from(buf, set);
    };
    auto& set0 = [&]() -> auto& { if constexpr (RAW) return rq0; else return ra; }();
    auto& set1 = [&]() -> auto& { if constexpr (RAW) return rq1; else return rb; }();
    ld2(0, set0);
    if (nslab > 1) ld2(1, set1);
    // one slab: `mine` held slab s (in LDS since the barrier that opened this step) and takes slab s + 2, `other` holds slab s + 1
    auto step = [&](int s, auto& mine, auto& other) {
      const int cur = s & 1;
      const unsigned short* ph = smem + (cur * 2 + 0) * X3_PLANE;
      if (s + 1 < nslab) {                                      // (wl / wl2 are never read in the single-plane arithmetic)
        load_w(2 * s + 2, wh3, wl);
        load_w(2 * s + 3, wh4, wl2);
      }
      if (s + 2 < nslab) ld2(s + 2, mine);
      mma_half(ph, ph, 0, 0, wh, wl);
      mma_half(ph, ph, 0, 1, wh, wl);
      mma_half(ph, ph, 1, 0, wh2, wl2);
      mma_half(ph, ph, 1, 1, wh2, wl2);
      if (s + 1 < nslab) {
        st2(cur ^ 1, other);
        wh[0] = wh3[0]; wh[1] = wh3[1];                         // the next slab's fragments (K steps 2 s + 2, 2 s + 3)
        wh2[0] = wh4[0]; wh2[1] = wh4[1];
      }
      __syncthreads();
    };
    while (true) {
      st2(0, set0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      __syncthreads();
      for (int s = 0; s < nslab; s += 2) {
        step(s, set0, set1);
        if (s + 1 < nslab) step(s + 1, set1, set0);
      }
      const int m0c = m0, nbc = nb;
      int nxt = tile + gridDim.x;
      while (nxt < ntiles && !decode(nxt, m0, nb)) nxt += gridDim.x;
      const bool more = nxt < ntiles;
      if (more) {
        setup(m0, nb);
        load_w(0, wh, wl);
        load_w(1, wh2, wl2);
        ld2(0, set0);
        if (nslab > 1) ld2(1, set1);
      }
      epilogue(m0c, nbc);
      if (!more) break;
      tile = nxt;
      __syncthreads();   // the epilogue staged the tile through the slab buffers
    }
    return;
  }
  load_w(0, wh, wl);
  load_slab(0);
  while (true) {
    store_slab(0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    for (int s = 0; s < nslab; ++s) {
      const int cur = s & 1;
      const unsigned short* ph = smem + (cur * 2 + 0) * X3_PLANE;
      const unsigned short* pl = smem + (cur * 2 + 1) * X3_PLANE;
      // two K steps per slab; the next step's weight fragments are in flight under the current MFMAs.
      // vmcnt retires in order: the (L2-resident) weight fragments are requested BEFORE the next slab's HBM
      // loads, so waiting for them does not also wait for the slab.
      load_w(2 * s + 1, wh2, wl2);
      if (s + 1 < nslab) load_slab(s + 1);
      mma_half(ph, pl, 0, 0, wh, wl);
      mma_half(ph, pl, 0, 1, wh, wl);
      if (s + 1 < nslab) load_w(2 * s + 2, wh, wl);
      mma_half(ph, pl, 1, 0, wh2, wl2);
      mma_half(ph, pl, 1, 1, wh2, wl2);
      if (s + 1 < nslab) store_slab(cur ^ 1);
      __syncthreads();
    }
    // next tile of this workgroup: its first slab and weight fragments go in flight under the epilogue
    const int m0c = m0, nbc = nb;
    int nxt = tile + gridDim.x;
    while (nxt < ntiles && !decode(nxt, m0, nb)) nxt += gridDim.x;
    const bool more = nxt < ntiles;
    if (more) {
      setup(m0, nb);
      load_w(0, wh, wl);
      load_slab(0);
    }
    epilogue(m0c, nbc);
    if (!more) break;
    tile = nxt;
    __syncthreads();   // the epilogue staged the tile through the slab buffers
  }
}

}  // namespace sepr
