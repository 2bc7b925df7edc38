// Shared pieces of the two projection cores (f32 MFMA: sepr_gemm.h, bf16x3 split MFMA: sepr_gemm_x3.h):
// the argument block and the row-contiguous epilogue that runs from an LDS-staged accumulator tile.
#pragma once
#include "sepr_common.h"

namespace sepr {

enum { PRO_PLAIN = 0, PRO_NORM = 1, PRO_CAT2 = 2 };
enum { EPI_STORE = 0, EPI_GLU = 1, EPI_GELU = 2, EPI_RES = 3, EPI_GATE = 4, EPI_SPLIT = 5, EPI_MASK = 6, EPI_DWGLU = 7, EPI_LNBWD = 8,
       // training only (their own kinds, so that the inference instantiations of EPI_GLU / EPI_RES compile to the same code as before):
       EPI_GLUSAVE = 9,    // EPI_GLU that also stores the pre-activation rows [M][N] to Ysave (the GLU backward needs both halves)
       EPI_RESDROP = 10 }; // EPI_RES with inverted dropout on (acc + bias) before LayerScale + residual (16-bit generator, sepr_drop_word)

struct GemmArgs {
  int M, N, K;
  // ---- A side ----
  const float* A;   // source rows, leading dimension lda
  int lda;
  const float* A2;  // PRO_CAT2: second source for k >= ksplit, leading dimension lda2
  int lda2;
  int ksplit;
  // Row map.  rows_out == 0: source row = m.  Otherwise m -> (seq = m / rows_out, r = m % rows_out);
  // the row is all-zero unless r < rows_valid; source row = seq*rows_src + (idx ? idx[r] : r) >> a_shift.
  // (a_shift applies to A only, never to A2: the fusion conv reads lo at t>>1 and skip at t.)
  int rows_out, rows_src, rows_valid, a_shift;
  const int* idx;
  // PRO_NORM: v = (a - mean) * rstd * gamma[k] + beta[k]; (mean, rstd) = stats[2*i], i = m (stat_seq == 0)
  // or m / rows_out (stat_seq == 1)
  const float* stats;
  int stat_seq;
  const float* gamma;
  const float* beta;
  // ---- W side ----
  const float* W;     // [N][K] row-major (torch Linear / 1x1 Conv weight)
  const float* bias;  // [N] or null
  // ---- output ----
  float* Y;
  int ldc;
  const float* R;    // EPI_RES: residual [M][ldc] or null.  EPI_GATE: x [M][ldc]
  const float* ls;   // EPI_RES: per-column scale [N] or null
  const float* aux;  // EPI_GATE: att [M/fac][N]; EPI_MASK: enc [(M/rows_out/S)*rows_out ...][N]
  int T, Tp, fac;    // EPI_GATE: frames per sequence, pooled frames, T/Tp.  EPI_SPLIT: T
  int S, Fs;         // EPI_SPLIT / EPI_MASK: speakers; EPI_SPLIT: F
  // EPI_DWGLU (GCFN): depthwise k=3 conv along frames + GLU applied to the projected tile before it
  // leaves the CU.  dw_w [3][N] tap-major, dw_b [N]; frames per sequence in T.  Output Y is [M][N/2].
  const float* dw_w;
  const float* dw_b;
  // bf16x3 core only: weights pre-split into bf16 (hi, lo) planes in MFMA-fragment order
  // ([N/16][K/32][plane][lane][8], see pack.py::pack_x3); LayerNorm gamma/beta are folded into Wp / bias.
  const void* Wp;
  int bf1;   // bf16x3 core: 1 = plain bf16 operands (one MFMA per product; hi planes only), 0 = the split arithmetic
  int a16;   // bf1 only, PRO_PLAIN / EPI_STORE / EPI_LNBWD: A is a bf16 tensor (lda in elements)
  // EPI_LNBWD (training: the input-gradient projection behind a LayerNorm, N <= 128 = one column tile): the tile holds whole rows
  // of dxh = d(loss)/d(normalised input), so the LayerNorm backward runs in the epilogue:
  //   Y[m] = (R ? R[m] : 0) + rstd_m (dxh - mean_f dxh - xh mean_f (dxh xh)) + (aux2 ? aux2[(m / T) Tp + (m % T) / fac] / fac : 0)
  // with xh = (aux[m] - mean_m) rstd_m, (mean, rstd) = stats[2m..]; aux = the LayerNorm's input x, aux2 = a pooled gradient (EGA)
  const float* aux2;
  // EPI_GLUSAVE: pre-activation rows (value | gate, bias added) [M][N], leading dimension N
  float* Ysave;
  // EPI_RESDROP: element (m, col) is kept iff its 16-bit draw of sepr_drop_word(key(seed, salt, site), m, col >> 1) >= drop_thr
  unsigned drop_thr, drop_site;
  float drop_scale;
  unsigned long long drop_seed;
  const unsigned long long* drop_salt;
  // bf16x3 core (round 6): when non-null, (mean, rstd) of every OUTPUT row (LayerNorm statistics over the N = ldc columns, eps stats_eps)
  // land in stats_out [M][2] - the next block's LayerNorm needs them.  A launch whose workgroup tile holds whole rows (the wide core at
  // N = 256) computes them at the end of the tile, from the rows it has just written (L2-hot), with the arithmetic of rowstats_kernel;
  // any other launch is followed by a rowstats launch.  Either way the values are identical.
  float* stats_out;
  float stats_eps;
};


constexpr int GEMM_BM = 128, GEMM_BN = 128;
constexpr int GEMM_THREADS = 256;
// EPI_DWGLU tiles overlap by one frame on each side (the conv halo is recomputed): 126 new rows per tile
constexpr int GEMM_DW_ROWS = GEMM_BM - 2;
constexpr int GEMM_HS = GEMM_BN + 4;   // row stride (floats) of the LDS-staged accumulator tile

inline int gemm_tiles(const GemmArgs& a, int epi) {
  const bool glu = (epi == EPI_GLU) || (epi == EPI_GLUSAVE) || (epi == EPI_DWGLU);
  const int rows = (epi == EPI_DWGLU) ? GEMM_DW_ROWS : GEMM_BM;
  const int NB = glu ? (a.N / 2 + 63) / 64 : (a.N + GEMM_BN - 1) / GEMM_BN;
  const int MB = (a.M + rows - 1) / rows;
  return ((MB + 7) / 8) * 8 * NB;   // always a multiple of 8 (XCD-aware decode)
}

inline GemmArgs gemm_args_zero() {
  GemmArgs a;
  __builtin_memset(&a, 0, sizeof(a));
  return a;
}

// host launchers: validate shapes, attribute the launch to a profiling site
int launch_gemm(int pro, int epi, const GemmArgs& a, int site, hipStream_t stream);      // f32 MFMA core
int launch_gemm_x3(int pro, int epi, const GemmArgs& a, int site, hipStream_t stream);   // bf16x3 core
// profiling hooks shared by both launchers (sepr_gemm.hip)
bool prof_begin(int site, hipStream_t stream, long long* slot);
void prof_end(long long slot, double flops, hipStream_t stream);
void prof_bytes(double bytes);   // algorithmic HBM bytes of the launch just timed (optional)
int persistent_grid();

// ---------------------------------------------------------------------------------------------------------
// Epilogue from LDS.  Hs holds the raw accumulators of one 128-row tile, [128][GEMM_HS] fp32: plain tiles
// keep their 128 columns in order, value/gate tiles (GLU, DWGLU) keep the 64 value columns in 0..63 and
// their gates in 64..127.  In the MFMA C layout a direct store would touch 16 rows x 64 B per instruction
// (partial cache lines; measured: the direct store tail cost 20 % of the kernel); from LDS a wave writes
// whole 256/512-byte row segments, and the residual / gate / mask operands are read the same way.
// The caller has issued __syncthreads() after writing Hs.
// ---------------------------------------------------------------------------------------------------------
template <int EPI>
__device__ __forceinline__ void epilogue_from_lds(const GemmArgs& a, const float* Hs, const int m0, const int nb,
                                                  const int tid) {
  // Every fused multiply-add below is written as fmaf(); implicit contraction is off so that the compiler
  // cannot fuse some unrolled row iterations and not others (that made results depend on a row's position
  // inside its tile at the 1-ulp level, i.e. a batch of 32 was not bit-identical to 32 single runs).
#pragma clang fp contract(off)
  constexpr int HS = GEMM_HS;
  if (EPI == EPI_DWGLU) {
    // GCFN: h = acc + b1, depthwise k=3 conv along frames (zero padding at sequence ends), GLU
    // (reference modules/network.py:61-65).  Tile row 0 / 127 are halo frames.
    const int half = a.N / 2;
    const int q4 = tid & 15, rg = tid >> 4;                     // 16 float4 columns x 16 strips of 8 rows
    const int gc = nb * 64 + 4 * q4;
    if (gc >= half) return;
    const float4 wv0 = ld4(a.dw_w + gc), wv1 = ld4(a.dw_w + a.N + gc), wv2 = ld4(a.dw_w + 2 * a.N + gc);
    const float4 wg0 = ld4(a.dw_w + half + gc), wg1 = ld4(a.dw_w + a.N + half + gc), wg2 = ld4(a.dw_w + 2 * a.N + half + gc);
    const float4 bv = ld4(a.dw_b + gc), bg = ld4(a.dw_b + half + gc);
    const float4 hv = ld4(a.bias + gc), hg = ld4(a.bias + half + gc);     // Linear bias b1 (value / gate)
    auto ldv = [&](int r) { float4 v = ld4(Hs + r * HS + 4 * q4); v.x += hv.x; v.y += hv.y; v.z += hv.z; v.w += hv.w; return v; };
    auto ldg = [&](int r) { float4 v = ld4(Hs + r * HS + 64 + 4 * q4); v.x += hg.x; v.y += hg.y; v.z += hg.z; v.w += hg.w; return v; };
    const int rs = 1 + 8 * rg;
    float4 pv = ldv(rs - 1), pg = ldg(rs - 1);
    float4 cv = ldv(rs), cg = ldg(rs);
    int t = (m0 + rs) % a.T;                                    // m0 + rs >= 0 always
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int r = rs + q;
      const int m = m0 + r;
      if (r > GEMM_DW_ROWS || m >= a.M) break;
      const float4 nv = ldv(r + 1), ng = ldg(r + 1);
      const float f0 = (t == 0) ? 0.f : 1.f;                    // zero padding at sequence starts / ends
      const float f2 = (t == a.T - 1) ? 0.f : 1.f;
      float4 v, g;
      v.x = fmaf(wv2.x * f2, nv.x, fmaf(wv1.x, cv.x, fmaf(wv0.x * f0, pv.x, bv.x)));
      v.y = fmaf(wv2.y * f2, nv.y, fmaf(wv1.y, cv.y, fmaf(wv0.y * f0, pv.y, bv.y)));
      v.z = fmaf(wv2.z * f2, nv.z, fmaf(wv1.z, cv.z, fmaf(wv0.z * f0, pv.z, bv.z)));
      v.w = fmaf(wv2.w * f2, nv.w, fmaf(wv1.w, cv.w, fmaf(wv0.w * f0, pv.w, bv.w)));
      g.x = fmaf(wg2.x * f2, ng.x, fmaf(wg1.x, cg.x, fmaf(wg0.x * f0, pg.x, bg.x)));
      g.y = fmaf(wg2.y * f2, ng.y, fmaf(wg1.y, cg.y, fmaf(wg0.y * f0, pg.y, bg.y)));
      g.z = fmaf(wg2.z * f2, ng.z, fmaf(wg1.z, cg.z, fmaf(wg0.z * f0, pg.z, bg.z)));
      g.w = fmaf(wg2.w * f2, ng.w, fmaf(wg1.w, cg.w, fmaf(wg0.w * f0, pg.w, bg.w)));
      st4(a.Y + (long long)m * a.ldc + gc,
          make_float4(v.x * sigmoid_f(g.x), v.y * sigmoid_f(g.y), v.z * sigmoid_f(g.z), v.w * sigmoid_f(g.w)));
      pv = cv; pg = cg; cv = nv; cg = ng;
      t = (t + 1 == a.T) ? 0 : t + 1;
    }
  } else if (EPI == EPI_LNBWD) {
    const int q4 = tid & 31, rg = tid >> 5;       // 32 float4 columns x 8 strips of 16 rows; a row = the 32 lanes of a half wave
    const int ncol = 4 * q4;
    const bool cok = ncol < a.N;
    const int cc = cok ? ncol : 0;
    const float invN = 1.0f / (float)a.N, invfac = a.fac > 0 ? 1.0f / (float)a.fac : 0.f;
#pragma unroll
    for (int ib = 0; ib < 16; ib += 8) {
      if (m0 + rg * 16 + ib >= a.M) break;        // uniform over the half wave that shares the shuffles below
      float4 xs[8], rs[8], ps[8];
      float2 st[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int mr = m0 + rg * 16 + ib + i;
        const long long m = mr < a.M ? mr : a.M - 1;
        xs[i] = ld4(a.aux + m * a.ldc + cc);
        st[i] = *reinterpret_cast<const float2*>(a.stats + 2 * m);
        rs[i] = a.R ? ld4(a.R + m * a.ldc + cc) : zero4();
        ps[i] = zero4();
        if (a.aux2) {
          const long long seq = m / a.T;
          ps[i] = ld4(a.aux2 + (seq * a.Tp + (m - seq * a.T) / a.fac) * a.ldc + cc);
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = rg * 16 + ib + i;
        const int m = m0 + r;
        float4 v = cok ? ld4(Hs + r * HS + ncol) : zero4();
        const float mean = st[i].x, rstd = st[i].y;
        float4 xh = make_float4((xs[i].x - mean) * rstd, (xs[i].y - mean) * rstd, (xs[i].z - mean) * rstd, (xs[i].w - mean) * rstd);
        if (!cok) xh = zero4();
        float s1 = (v.x + v.y) + (v.z + v.w);
        float s2 = (v.x * xh.x + v.y * xh.y) + (v.z * xh.z + v.w * xh.w);
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
          s1 += __shfl_xor(s1, o, 32);
          s2 += __shfl_xor(s2, o, 32);
        }
        const float m1 = s1 * invN, m2 = s2 * invN;
        if (m < a.M && cok) {
          float4 o4 = make_float4(rstd * (v.x - m1 - xh.x * m2), rstd * (v.y - m1 - xh.y * m2), rstd * (v.z - m1 - xh.z * m2),
                                  rstd * (v.w - m1 - xh.w * m2));
          o4.x += rs[i].x; o4.y += rs[i].y; o4.z += rs[i].z; o4.w += rs[i].w;
          if (a.aux2) { o4.x = fmaf(ps[i].x, invfac, o4.x); o4.y = fmaf(ps[i].y, invfac, o4.y); o4.z = fmaf(ps[i].z, invfac, o4.z); o4.w = fmaf(ps[i].w, invfac, o4.w); }
          st4(a.Y + (long long)m * a.ldc + ncol, o4);
        }
      }
    }
  } else if (EPI == EPI_GLU || EPI == EPI_GLUSAVE) {
    const int q4 = tid & 15, rg = tid >> 4;       // 16 float4 columns (64 outputs) x 16 strips of 8 rows
    const int ncol = nb * 64 + 4 * q4;
    if (ncol >= a.N / 2) return;
    const float4 bv = ld4(a.bias + ncol), bg = ld4(a.bias + a.N / 2 + ncol);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int r = rg * 8 + i;
      const int m = m0 + r;
      if (m >= a.M) break;
      const float4 v = ld4(Hs + r * HS + 4 * q4), g = ld4(Hs + r * HS + 64 + 4 * q4);
      if (EPI == EPI_GLUSAVE) {   // the same (acc + bias) values the GLU below consumes
        st4(a.Ysave + (long long)m * a.N + ncol, make_float4(v.x + bv.x, v.y + bv.y, v.z + bv.z, v.w + bv.w));
        st4(a.Ysave + (long long)m * a.N + a.N / 2 + ncol, make_float4(g.x + bg.x, g.y + bg.y, g.z + bg.z, g.w + bg.w));
      }
      st4(a.Y + (long long)m * a.ldc + ncol,
          make_float4((v.x + bv.x) * sigmoid_f(g.x + bg.x), (v.y + bv.y) * sigmoid_f(g.y + bg.y),
                      (v.z + bv.z) * sigmoid_f(g.z + bg.z), (v.w + bv.w) * sigmoid_f(g.w + bg.w)));
    }
  } else {
    const int q4 = tid & 31, rg = tid >> 5;       // 32 float4 columns (128 outputs) x 8 strips of 16 rows
    const int ncol = nb * GEMM_BN + 4 * q4;
    if (ncol >= a.N) return;
    const float4 bias = a.bias ? ld4(a.bias + ncol) : zero4();
    float4 lsv = make_float4(1.f, 1.f, 1.f, 1.f);
    if ((EPI == EPI_RES || EPI == EPI_RESDROP) && a.ls) lsv = ld4(a.ls + ncol);
    DropKey dkey = {0u, 0u};
    if (EPI == EPI_RESDROP) {
      dkey = sepr_drop_key(a.drop_seed, a.drop_salt, a.drop_site);
      lsv.x *= a.drop_scale; lsv.y *= a.drop_scale; lsv.z *= a.drop_scale; lsv.w *= a.drop_scale;   // keep scale rides in the LayerScale
    }
    const int split_s = (EPI == EPI_SPLIT) ? ncol / a.Fs : 0;
    const int split_f = (EPI == EPI_SPLIT) ? ncol - split_s * a.Fs : 0;
    // Rows go in batches of 8: the side operands of a batch (residual, gate inputs, encoder frames) are
    // requested together from clamped, branch-free addresses BEFORE any of them is consumed, so a tile pays two
    // global-load latencies instead of sixteen (a load guarded by "row < M" cannot be hoisted above the guard of
    // the previous row; that serialisation was ~40 % of these kernels).  Only the stores are predicated.
    const bool has_r = (EPI != EPI_RES && EPI != EPI_RESDROP) || a.R != nullptr;
#pragma unroll
    for (int ib = 0; ib < 16; ib += 8) {
      if (m0 + rg * 16 + ib >= a.M) break;          // whole batch past the last row
      float4 s1[8], s2[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int mr = m0 + rg * 16 + ib + i;
        const int m = mr < a.M ? mr : a.M - 1;
        s1[i] = zero4();
        s2[i] = zero4();
        if (EPI == EPI_RES || EPI == EPI_RESDROP) {
          if (has_r) s1[i] = ld4(a.R + (long long)m * a.ldc + ncol);
        } else if (EPI == EPI_GATE) {
          const int seq = m / a.T;
          const int t = m - seq * a.T;
          s1[i] = ld4(a.R + (long long)m * a.ldc + ncol);
          s2[i] = ld4(a.aux + ((long long)seq * a.Tp + t / a.fac) * a.N + ncol);
        } else if (EPI == EPI_MASK) {
          const int seq = m / a.rows_out;             // b*S + s
          const int l = m - seq * a.rows_out;
          s1[i] = ld4(a.aux + ((long long)(seq / a.S) * a.rows_out + l) * a.N + ncol);
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = rg * 16 + ib + i;
        const int m = m0 + r;
        if (m < a.M) {
          float4 v = ld4(Hs + r * HS + 4 * q4);
          v.x += bias.x; v.y += bias.y; v.z += bias.z; v.w += bias.w;
          float* const out = a.Y + (long long)m * a.ldc + ncol;
          if (EPI == EPI_STORE) {
            st4(out, v);
          } else if (EPI == EPI_GELU) {
            st4(out, make_float4(gelu_exact(v.x), gelu_exact(v.y), gelu_exact(v.z), gelu_exact(v.w)));
          } else if (EPI == EPI_RES || EPI == EPI_RESDROP) {
            if (EPI == EPI_RESDROP) {   // reference network.py:171,187 (CLA) under train(): Dropout on the projection's output
              const unsigned d0 = sepr_drop_word(dkey, (unsigned)m, (unsigned)(ncol >> 1)), d1 = sepr_drop_word(dkey, (unsigned)m, (unsigned)(ncol >> 1) + 1u);
              v.x = (d0 & 0xffffu) >= a.drop_thr ? v.x : 0.f;
              v.y = (d0 >> 16) >= a.drop_thr ? v.y : 0.f;
              v.z = (d1 & 0xffffu) >= a.drop_thr ? v.z : 0.f;
              v.w = (d1 >> 16) >= a.drop_thr ? v.w : 0.f;
            }
            if (has_r) {
              const float4 rr = s1[i];
              v.x = fmaf(v.x, lsv.x, rr.x); v.y = fmaf(v.y, lsv.y, rr.y);
              v.z = fmaf(v.z, lsv.z, rr.z); v.w = fmaf(v.w, lsv.w, rr.w);
            } else {
              v.x *= lsv.x; v.y *= lsv.y; v.z *= lsv.z; v.w *= lsv.w;
            }
            st4(out, v);
          } else if (EPI == EPI_GATE) {
            const float4 x = s1[i], u = s2[i];
            st4(out, make_float4(fmaf(sigmoid_f(v.x), u.x, x.x), fmaf(sigmoid_f(v.y), u.y, x.y),
                                 fmaf(sigmoid_f(v.z), u.z, x.z), fmaf(sigmoid_f(v.w), u.w, x.w)));
          } else if (EPI == EPI_SPLIT) {
            // column n = s*F + f  ->  Y[((b*S + s)*T + t)*F + f]   (reference module.py:123)
            const int b = m / a.T;
            const int t = m - b * a.T;
            st4(a.Y + (((long long)b * a.S + split_s) * a.T + t) * a.Fs + split_f, v);
          } else if (EPI == EPI_MASK) {
            const float4 e = s1[i];
            st4(out, make_float4(fmaxf(v.x, 0.f) * e.x, fmaxf(v.y, 0.f) * e.y, fmaxf(v.z, 0.f) * e.z, fmaxf(v.w, 0.f) * e.w));
          }
        }
      }
    }
  }
}

}  // namespace sepr
