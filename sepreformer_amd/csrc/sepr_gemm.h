// Row-projection core: Y = epilogue( prologue(X)[M,K] . W[N,K]^T + bias ) on the f32 MFMA pipe.
//
// Every dense contraction of the separator (94 % of its FLOPs) is a "tall" product:
// M = batch x frames rows (up to 512 000), K and N in {F .. 8F}.  One workgroup (4 waves, 256 threads)
// owns a 128-row x 128-column output tile; each wave a 64 x 64 sub-tile held as 4 x 4 accumulators of
// v_mfma_f32_16x16x4_f32 (exact fp32 fmaf-chain numerics, 157 TFLOP/s chip peak).
//
// Operand roles are swapped on purpose: the MFMA "A" operand is the WEIGHT fragment (row index = output
// column n) and "B" is the ACTIVATION fragment (column index = row m), so a lane ends up with four
// consecutive output columns of one row (D[row = 4*(lane>>4) + r][col = lane&15]) and the epilogue
// stores float4s.  Both fragments are "row, 4 consecutive k" 16-byte LDS reads; one K=16 step is four
// MFMAs whose k-slots are {4g + r}: lane group g = lane>>4 supplies k = 16*kk + 4*g + r to MFMA r.
//
// The K loop is register-prefetched and LDS double-buffered (BK = 32, one barrier per K tile); two
// workgroups fit a CU (72 KiB LDS each) so one's global latency hides under the other's MFMAs.
// Prologues (LayerNorm / GroupNorm apply, row gather, two-source concat) run while staging A into LDS,
// epilogues (bias, GLU, GELU, LayerScale + residual, sigmoid gate, speaker split, ReLU mask) on the
// accumulators, so no normalised / activated intermediate ever goes to HBM.
#pragma once
#include "sepr_gemm_epi.h"

namespace sepr {

#ifndef SEPR_GEMM_PERSIST
#define SEPR_GEMM_PERSIST 1   // 1: <= 2 workgroups per CU walking tiles, 0: one tile per workgroup (A/B builds)
#endif
#ifndef SEPR_GEMM_STAGGER
#define SEPR_GEMM_STAGGER 1   // 1: de-phase the two co-resident workgroups of a CU at kernel start
#endif
#ifndef SEPR_ABL_NOLOAD
#define SEPR_ABL_NOLOAD 0
#endif
#ifndef SEPR_ABL_NOSTORE
#define SEPR_ABL_NOSTORE 0
#endif
#ifndef SEPR_GEMM_LDS_PAD
#define SEPR_GEMM_LDS_PAD 8   // floats of padding per 32-float LDS row: 8 -> stride 40, conflict-free b128 reads
#endif
constexpr int GEMM_BK = 32, GEMM_LDS_STRIDE = GEMM_BK + SEPR_GEMM_LDS_PAD;

// TAG does not change the code: it gives the two GCFN projections (60 % of the model's FLOPs) their own
// kernel symbols, so a rocprofv3 kernel trace separates them from the other users of the same
// prologue/epilogue pair (TAG 1 = GCFN F->6F, TAG 2 = GCFN 3F->F, 0 = everything else).
//
// The kernel is PERSISTENT: the grid is at most 2 workgroups per CU and each workgroup walks a strided
// list of output tiles.  K is only 128-512 here (4-16 K tiles), so a one-tile-per-workgroup kernel spends
// a third of every wave's life parked on the first HBM round trip and on the store tail (measured: 28 %
// SQ_WAIT_ANY, 60 % MFMA utilisation at K = 128 against 80 % at K = 4096).  Walking tiles lets the first
// K tile of the NEXT output tile be fetched before the epilogue of the current one, so the only exposed
// memory latency is the very first tile's.
template <int PRO, int EPI, int TAG = 0>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_kernel(const GemmArgs a) {
  constexpr bool DWGLU = (EPI == EPI_DWGLU);
  constexpr bool GLU = (EPI == EPI_GLU) || (EPI == EPI_GLUSAVE) || DWGLU;   // value / gate column pairing of the weight tile
  constexpr int ROWS_OUT = DWGLU ? GEMM_DW_ROWS : GEMM_BM;
  constexpr int LS = GEMM_LDS_STRIDE;
  __shared__ __attribute__((aligned(16))) float smem[2 * (GEMM_BM + GEMM_BN) * LS];
  float* const As0 = smem;
  float* const Bs0 = smem + 2 * GEMM_BM * LS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int fi = lane & 15, fg = lane >> 4;
  const int c4 = tid & 7;    // staging: float4 column of the 32-wide K tile
  const int r0 = tid >> 3;   // staging: rows r0 + 32 i

  // XCD-aware tile order: workgroup b runs on XCD b % 8 and the grid is a multiple of 8, so a workgroup
  // stays on "its" XCD for every tile it walks; the NB column tiles of one row tile are consecutive on
  // the same XCD, so the A rows they share are served by one L2.
  const int NB = GLU ? (a.N / 2 + 63) / 64 : (a.N + GEMM_BN - 1) / GEMM_BN;
  const int MB = (a.M + ROWS_OUT - 1) / ROWS_OUT;
  const int ntiles = ((MB + 7) / 8) * 8 * NB;
  const int nk = a.K / GEMM_BK;

  // ---- staging state of the tile being loaded ------------------------------------------------------
  // 32-bit element offsets from the kernel-argument bases (SGPR base + VGPR offset addressing; the
  // kernel has to stay within 256 registers for two waves per SIMD).  Row validity is a 0/1 multiplier:
  // loads are unconditional (invalid rows read row 0) and the staging code is branch-free.
  unsigned pa[4], pa2[4], pw[4];
  float mka[4], mean[4], rstd[4];
  float4 ra[4], rb[4];
  float4 g4 = make_float4(1.f, 1.f, 1.f, 1.f), b4 = zero4();

  auto setup = [&](int m0, int nb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + r0 + 32 * i;
      pa[i] = 0u;
      pa2[i] = 0u;
      mka[i] = 0.f;
      mean[i] = 0.f;
      rstd[i] = 0.f;
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
          mka[i] = 1.f;
          pa[i] = (unsigned)(src * a.lda);
          if (PRO == PRO_CAT2) pa2[i] = (unsigned)((long long)m * a.lda2);
          if (PRO == PRO_NORM) {
            const long long si = a.stat_seq ? seq : m;
            mean[i] = a.stats[2 * si];
            rstd[i] = a.stats[2 * si + 1];
          }
        }
      }
      // weight rows of the tile: plain = 128 consecutive rows; GLU = 64 value rows then their 64 gate rows
      const int rr = r0 + 32 * i;
      int wrow;
      bool wvalid;
      if (GLU) {
        const int c = nb * 64 + (rr & 63);
        wvalid = c < a.N / 2;
        wrow = (rr < 64) ? c : a.N / 2 + c;
      } else {
        wrow = nb * GEMM_BN + rr;
        wvalid = wrow < a.N;
      }
      pw[i] = wvalid ? (unsigned)wrow * (unsigned)a.K : 0u;
    }
  };
  auto load_tile = [&](int kt) {
#if SEPR_ABL_NOLOAD
    if (kt > 0 || blockIdx.x != (unsigned)a.M) return;   // timing ablation: no global loads after setup
#endif
    const int k = kt * GEMM_BK + 4 * c4;
    if (PRO == PRO_NORM && a.gamma) {   // gamma == NULL: affine already folded into W / bias (training path)
      g4 = ld4(a.gamma + k);
      b4 = ld4(a.beta + k);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (PRO == PRO_CAT2) {
        ra[i] = (k < a.ksplit) ? ld4(a.A + (pa[i] + k)) : ld4(a.A2 + (pa2[i] + (k - a.ksplit)));
      } else {
        ra[i] = ld4(a.A + (pa[i] + k));
      }
      rb[i] = ld4(a.W + (pw[i] + k));
    }
  };
  auto store_tile = [&](int buf) {
#pragma clang fp contract(off)
    float* As = As0 + buf * GEMM_BM * LS;
    float* Bs = Bs0 + buf * GEMM_BN * LS;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 v = ra[i];
      if (PRO == PRO_NORM) {
        // invalid rows carry mean = rstd = 0 and mask 0: exactly zero (pad_signal semantics)
        const float sc = rstd[i] * mka[i];
        v.x = fmaf((v.x - mean[i]) * sc, g4.x, b4.x * mka[i]);
        v.y = fmaf((v.y - mean[i]) * sc, g4.y, b4.y * mka[i]);
        v.z = fmaf((v.z - mean[i]) * sc, g4.z, b4.z * mka[i]);
        v.w = fmaf((v.w - mean[i]) * sc, g4.w, b4.w * mka[i]);
      } else {
        v.x *= mka[i]; v.y *= mka[i]; v.z *= mka[i]; v.w *= mka[i];
      }
      st4(As + (r0 + 32 * i) * LS + 4 * c4, v);
      // weight rows past N read row 0: they only feed output columns that are never stored
      st4(Bs + (r0 + 32 * i) * LS + 4 * c4, rb[i]);
    }
  };

  // LDS row of the weight fragment nt of this wave
  int brow[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
    brow[nt] = GLU ? ((nt >> 1) * 64 + wn * 32 + (nt & 1) * 16) : (wn * 64 + nt * 16);

  f32x4 acc[4][4];
  auto read_frags = [&](const float* As, const float* Bs, int kk, float4 (&wf)[4], float4 (&xf)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      wf[t] = ld4(Bs + (brow[t] + fi) * LS + kk * 16 + 4 * fg);
      xf[t] = ld4(As + (wm * 64 + t * 16 + fi) * LS + kk * 16 + 4 * fg);
    }
  };
  auto mma16 = [&](const float4 (&wf)[4], const float4 (&xf)[4]) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt].x, xf[mt].x, acc[nt][mt], 0, 0, 0);
        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt].y, xf[mt].y, acc[nt][mt], 0, 0, 0);
        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt].z, xf[mt].z, acc[nt][mt], 0, 0, 0);
        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt].w, xf[mt].w, acc[nt][mt], 0, 0, 0);
      }
  };
  auto decode = [&](int tile, int& m0, int& nb) -> bool {
    const int q = tile >> 3;
    const int mb = (q / NB) * 8 + (tile & 7);
    nb = q % NB;
    m0 = mb * ROWS_OUT - (DWGLU ? 1 : 0);   // DWGLU: tile row 0 is the frame before the first output
    return mb < MB;
  };

  auto epilogue = [&](const int m0, const int nb) {
#if SEPR_ABL_NOSTORE
    {   // timing ablation: keep the accumulators live, store nothing
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
      return;
    }
#endif
    // accumulators -> LDS (the staging buffers are free after the K loop's closing barrier), then the
    // shared row-contiguous epilogue
    float* const Hs = smem;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int cl = GLU ? ((nt >> 1) * 64 + wn * 32 + (nt & 1) * 16 + 4 * fg) : (wn * 64 + nt * 16 + 4 * fg);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const f32x4 c = acc[nt][mt];
        st4(Hs + (wm * 64 + mt * 16 + fi) * GEMM_HS + cl, make_float4(c[0], c[1], c[2], c[3]));
      }
    }
    __syncthreads();
    epilogue_from_lds<EPI>(a, Hs, m0, nb, tid);
  };

#if SEPR_GEMM_STAGGER
  // De-phase the two workgroups that share a CU.  They are dispatched together and run identical tile
  // sequences, so without this they stay in lockstep: both parked on memory at the same time, then both
  // contending for the matrix pipe (measured: 28 % SQ_WAIT_ANY yet only 60 % MFMA utilisation with two
  // resident waves per SIMD).  Half of the workgroups - one of every co-resident pair under either
  // plausible placement (round-robin over the XCD's 32 CUs, or CU-by-CU) - start half a tile period late.
  {
    const int ql = blockIdx.x >> 3;
    if (((ql & 1) ^ ((ql >> 5) & 1)) != 0) {
      const int naps = (nk * 2048 + 6144) / (64 * 100);
      for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(100);
    }
  }
#endif
  // ---- walk the tiles ---------------------------------------------------------------------------------
  int tile = blockIdx.x;
  int m0 = 0, nb = 0;
  while (tile < ntiles && !decode(tile, m0, nb)) tile += gridDim.x;
  if (tile >= ntiles) return;
  setup(m0, nb);
  load_tile(0);
  while (true) {
    // (entering: registers hold K tile 0 of (m0, nb); every wave is past its reads of the staging LDS)
    store_tile(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < nk) load_tile(kt + 1);
      const float* As = As0 + cur * GEMM_BM * LS;
      const float* Bs = Bs0 + cur * GEMM_BN * LS;
#pragma unroll
      for (int kk = 0; kk < GEMM_BK / 16; ++kk) {
        float4 wf[4], xf[4];
        read_frags(As, Bs, kk, wf, xf);
        mma16(wf, xf);
      }
      if (kt + 1 < nk) store_tile(cur ^ 1);
      __syncthreads();
    }
    // next tile of this workgroup: start its first K tile now, under the epilogue below
    const int m0c = m0, nbc = nb;
    int nxt = tile + gridDim.x;
    while (nxt < ntiles && !decode(nxt, m0, nb)) nxt += gridDim.x;
    const bool more = nxt < ntiles;
    if (more) {
      setup(m0, nb);
      load_tile(0);
    }
    epilogue(m0c, nbc);
    if (!more) break;
    tile = nxt;
    __syncthreads();              // the epilogue staged the tile through the staging LDS
  }
}

}  // namespace sepr
