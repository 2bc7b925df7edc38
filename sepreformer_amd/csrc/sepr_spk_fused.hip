// Fully fused speaker-attention block on the bf16x3 core (reference modules/network.py:229-247 around the
// MultiHeadAttention of :69-124, two speakers):
//     y = x + layer_scale * Linear_out( softmax_over_speakers( q k^T / sqrt(dk) ) v ),   q,k,v = Linear(LayerNorm(x))
// in ONE kernel.  The [rows, 3F] q/k/v tensor, the mixed [rows, F] tensor and the LayerNorm statistics never reach
// HBM: traffic is "read x once, write y once" (1 KB per row at F = 128) instead of 6 KB for
// rowstats + q/k/v projection + speaker mix + output projection.
//
// Same row-stationary skeleton as sepr_gcfn_fused.hip (gcfn_fused3_kernel); what differs:
//   * a wave owns 16 frames of BOTH speakers of one mixture: frame tile 0 = speaker 0, frame tile 1 = speaker 1
//     (speaker s of frame (b,t) is row (b*2+s)*T + t), so the attention across speakers is lane-local: a lane's
//     two accumulator sets are the two speakers of the same frame.  No halo frames;
//   * the hidden walk is over head pairs (32 output channels of q, k and v each): per head 3 tiles x F/32 K steps
//     of MFMAs give q, k, v (bias = accumulator init) with a lane holding 4 of the head's 16 channels for both
//     speakers; the 2x2 scores are 4-channel partial dot products summed over the 4 lane groups (two xor
//     shuffles); softmax over two speakers is a logistic of the score difference; the mixed values are split to
//     bf16 hi/lo and ARE the B fragment of the output projection's K step for that head pair (weights packed in the
//     matching k-slot order, exactly the down-projection trick of the GCFN kernel);
//   * per head pair 48 KB of q/k/v fragments + 16 KB of output-projection fragments + 4 KB constants go
//     global -> LDS by LDS-DMA under the arithmetic of the previous pair; two 4-wave workgroups share a CU.
#include "sepr_gemm_epi.h"
#include <stdlib.h>

namespace sepr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct SpkFusedArgs {
  const float* x;     // [nS*T, F], sequences ordered b*2 + s
  float* y;           // [nS*T, F]
  int NF, T;          // frames over all mixtures (= B*T), frames per sequence
  const void* w1p;    // per head pair: [6 tiles: q0 q1 k0 k1 v0 v1][F/32][plane][64][8] bf16 (LayerNorm gamma folded),
                      // then 4 KB of constants: [6 tiles][16 channels] fp32 biases (beta folded), zero padded
  const void* w2p;    // [F/32][F/16][plane][64][8] bf16, k-slot order (g,e) -> e<4 ? 4g+e : 16+4g+e-4
  const float* bo;    // [F]
  const float* ls;    // [F]
  float eps, inv_sqrt_dk;
};

// DK32 (round 6, the Large variants: F = 256, 8 heads of 32 channels): a 32-channel chunk is ONE head - its two 16-channel tile triples
// (q0 k0 v0 | q1 k1 v1, the same packed layout) - so the 2x2 scores sum both tiles' partial dot products before the softmax and both tiles
// are mixed with the same probabilities.  128 registers of frame planes + 128 of accumulators: the one-wave-per-SIMD regime (one 136 KB
// workgroup per CU), like gcfn_fused3_kernel<256, ..>.
template <int F, int NW, bool DK32 = false>
__global__ __launch_bounds__(64 * NW, F > 128 ? 1 : (2 * NW) / 4) void spk_fused_kernel(const SpkFusedArgs a) {
  constexpr int MT = 2;                  // = speakers
  constexpr int NT = 64 * NW;
  constexpr int TILE = 16 * NW;          // frames per workgroup tile
  constexpr int EH = (16 * MT * NW) / 64;
  constexpr int KS = F / 32;
  constexpr int NCH = F / 32;            // head pairs
  constexpr int FT = F / 16;
  constexpr int W1F_U4 = 6 * KS * 2 * 64;
  constexpr int CS_U4 = 256;
  constexpr int W1_U4 = W1F_U4 + CS_U4;
  constexpr int W2_U4 = FT * 2 * 64;
  constexpr int OS = F + 4;
  __shared__ __attribute__((aligned(16))) uint4 wl[W1F_U4 + W2_U4 + 2 * CS_U4];
  static_assert(sizeof(uint4) * (W1F_U4 + W2_U4) >= sizeof(float) * 64 * OS, "epilogue staging must fit");
  static_assert(W1F_U4 % NT == 0 && CS_U4 % NT == 0 && W2_U4 % NT == 0 && W1F_U4 / NT <= 24 && (16 * MT * NW) % 64 == 0, "copy / epilogue partition");
  const uint4* const w1s = wl;
  const uint4* const w2s = wl + W1F_U4;
  uint4* const csl = wl + W1F_U4 + W2_U4;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fi = lane & 15, fg = lane >> 4;
  const int ntiles = (a.NF + TILE - 1) / TILE;
  const uint4* const W1g = static_cast<const uint4*>(a.w1p);
  const uint4* const W2g = static_cast<const uint4*>(a.w2p);

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // ---- this wave's 16 frames x 2 speakers: load, LayerNorm, split ------------------------------------------
    bf16x8 xh[MT][KS], xl[MT][KS];
    {
      const int f = tile * TILE + w * 16 + fi;
      const bool valid = f < a.NF;
      const int b = valid ? f / a.T : 0, t = valid ? f - b * a.T : 0;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const float* xp = a.x + ((long long)(b * MT + mt) * a.T + t) * F + 8 * fg;
        float v[KS][8];
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const float4 p = ld4(xp + 32 * ks), q = ld4(xp + 32 * ks + 4);
          v[ks][0] = p.x; v[ks][1] = p.y; v[ks][2] = p.z; v[ks][3] = p.w;
          v[ks][4] = q.x; v[ks][5] = q.y; v[ks][6] = q.z; v[ks][7] = q.w;
#pragma unroll
          for (int e = 0; e < 8; ++e) s += v[ks][e];
        }
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        const float mean = s * (1.0f / F);
        float d = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float c = v[ks][e] - mean;
            d = fmaf(c, c, d);
          }
        d += __shfl_xor(d, 16, 64);
        d += __shfl_xor(d, 32, 64);
        const float rstd = 1.0f / sqrtf(d * (1.0f / F) + a.eps);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          bf16x8 h, l;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float xn = (v[ks][e] - mean) * rstd;
            const __bf16 hh = (__bf16)xn;
            h[e] = hh;
            l[e] = (__bf16)(xn - (float)hh);
          }
          xh[mt][ks] = h;
          xl[mt][ks] = l;
        }
      }
    }
    f32x4 acc[FT][MT];
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[ft][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- weight chunks: global -> LDS by LDS-DMA (protocol of gcfn_fused3_kernel) ------------------------------
#ifndef SEPR_SPK_ASMDMA
#define SEPR_SPK_ASMDMA 1   // inline-asm LDS-DMA (sepr_common.h glds16_asm): the copies are waited for at the chunk barriers only
#endif
    [[maybe_unused]] const int ws = __builtin_amdgcn_readfirstlane(w);
    auto dma = [&](const uint4* gbase, uint4* lbase, int nblk) {
      unsigned loff = (unsigned)lane * 16u;
      asm volatile("" : "+v"(loff));
#pragma unroll
      for (int i = 0; i < 24; ++i) {       // (24: the q/k/v fragments of a head at F = 256 are 96 KB)
        if (i >= nblk) break;
#if SEPR_SPK_ASMDMA
        glds16_asm(gbase + (i * NW + ws) * 64, loff, __builtin_amdgcn_readfirstlane(lds_addr(lbase + (i * NW + ws) * 64)));
#else
        const int blk = i * NW + w;
        const char* src = reinterpret_cast<const char*>(gbase + blk * 64) + loff;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lbase + blk * 64), 16, 0, 0);
#endif
      }
    };
    auto dma_w1 = [&](int c) {
      dma(W1g + (long long)c * W1_U4, wl, W1F_U4 / NT);
      dma(W1g + (long long)c * W1_U4 + W1F_U4, csl + (c & 1) * CS_U4, CS_U4 / NT);
    };
    auto dma_w2 = [&](int c) { dma(W2g + (long long)c * W2_U4, wl + W1F_U4, W2_U4 / NT); };
#ifndef SEPR_SPK_ABL
#define SEPR_SPK_ABL 0      // timing ablations (wrong results): 1 = weight chunks copied once per tile, 2 = no chunk
#endif                      // barriers, 4 = no chunk loop (prologue + epilogue only), 8 = no score shuffles
    auto dma_barrier = [&]() {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if (SEPR_SPK_ABL & 2) == 0
      __syncthreads();
#endif
    };
    // fragment pair (bf16 hi plane, lo plane): head hh of the pair, group g = 3*ks + (0 q | 1 k | 2 v)
    auto ld_up = [&](int hh, int g, uint4 (&d)[2]) {
      const uint4* p = w1s + ((((g % 3) * 2 + hh) * KS + (g / 3)) * 2) * 64 + lane;
      d[0] = p[0];
      d[1] = p[64];
    };
    auto ld_dn = [&](int ft, uint4 (&d)[2]) {
      const uint4* p = w2s + (ft * 2) * 64 + lane;
      d[0] = p[0];
      d[1] = p[64];
    };

    __syncthreads();   // the previous tile's epilogue staging is fully consumed
    dma_w1(0);
    dma_w2(0);
    dma_barrier();     // head pair 0 landed
    for (int c = 0; c < ((SEPR_SPK_ABL & 4) ? 0 : NCH); ++c) {
      bf16x8 gh[MT], gw[MT];          // mixed values (bf16 hi / lo) in output-projection k-slot order, per speaker
      [[maybe_unused]] float sc0[MT][MT];          // DK32: the first tile's partial scores and v values
      [[maybe_unused]] f32x4 v0[MT];
      uint4 fb[4][2];                 // fragment ring: the next PAIR of MFMA groups in flight under the current pair
      ld_up(0, 0, fb[0]);
      ld_up(0, 1, fb[1]);
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const float* cs = reinterpret_cast<const float*>(csl + (c & 1) * CS_U4) + 4 * fg;
        // ---- q, k, v of this head for both speakers; accumulators start at the bias -----------------------------
        f32x4 pq[3][MT];
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
          const float4 bv = ld4(cs + (tt * 2 + hh) * 16);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) pq[tt][mt] = (f32x4){bv.x, bv.y, bv.z, bv.w};
        }
        // MFMA groups go in PAIRS (two weight tiles x two speakers = four independent accumulators, issued
        // term-major): back-to-back MFMAs on the same accumulator are 4 instructions apart instead of 2, which
        // is what the 16x16x32 pipeline depth needs; the next pair's fragments are requested first.
#pragma unroll
        for (int gp = 0; gp < 3 * KS; gp += 2) {
          if (gp + 2 < 3 * KS) {
            ld_up(hh, gp + 2, fb[(gp + 2) & 3]);
            ld_up(hh, gp + 3, fb[(gp + 3) & 3]);
          }
          __builtin_amdgcn_sched_barrier(0);
          const bf16x8 wh0 = *reinterpret_cast<const bf16x8*>(&fb[gp & 3][0]);
          const bf16x8 wl0 = *reinterpret_cast<const bf16x8*>(&fb[gp & 3][1]);
          const bf16x8 wh1 = *reinterpret_cast<const bf16x8*>(&fb[(gp + 1) & 3][0]);
          const bf16x8 wl1 = *reinterpret_cast<const bf16x8*>(&fb[(gp + 1) & 3][1]);
          const int ks0 = gp / 3, tt0 = gp % 3, ks1 = (gp + 1) / 3, tt1 = (gp + 1) % 3;
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) pq[tt0][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh0, xh[mt][ks0], pq[tt0][mt], 0, 0, 0);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) pq[tt1][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh1, xh[mt][ks1], pq[tt1][mt], 0, 0, 0);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) pq[tt0][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh0, xl[mt][ks0], pq[tt0][mt], 0, 0, 0);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) pq[tt1][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh1, xl[mt][ks1], pq[tt1][mt], 0, 0, 0);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) pq[tt0][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl0, xh[mt][ks0], pq[tt0][mt], 0, 0, 0);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) pq[tt1][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl1, xh[mt][ks1], pq[tt1][mt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (hh == 0) {                           // the second head's first fragments arrive under the mix below
          ld_up(1, 0, fb[0]);
          ld_up(1, 1, fb[1]);
        } else {
          dma_barrier();                         // every wave has read its q/k/v fragments of pair c; the pair's
                                                 // output-projection fragments have landed
#if (SEPR_SPK_ABL & 1) == 0
          if (c + 1 < NCH) dma_w1(c + 1);        // lands under the mix + output projection below
#endif
          ld_dn(0, fb[0]);
          ld_dn(1, fb[1]);
        }
        // ---- 2x2 attention across the speakers of each frame (network.py:106-122 with T = S) --------------------
        float sc[MT][MT];
#pragma unroll
        for (int qa = 0; qa < MT; ++qa)
#pragma unroll
          for (int kc = 0; kc < MT; ++kc) {
            float p = pq[0][qa][0] * pq[1][kc][0];
            p = fmaf(pq[0][qa][1], pq[1][kc][1], p);
            p = fmaf(pq[0][qa][2], pq[1][kc][2], p);
            p = fmaf(pq[0][qa][3], pq[1][kc][3], p);
#if (SEPR_SPK_ABL & 8) == 0
            p += __shfl_xor(p, 16, 64);          // the tile's 16 channels live in the 4 lane groups
            p += __shfl_xor(p, 32, 64);
#endif
            sc[qa][kc] = p;
          }
        if constexpr (DK32) {
          // one 32-channel head: tile 0 parks its partial scores and its v values, tile 1 completes the scores and mixes both tiles
          if (hh == 0) {
#pragma unroll
            for (int qa = 0; qa < MT; ++qa)
#pragma unroll
              for (int kc = 0; kc < MT; ++kc) sc0[qa][kc] = sc[qa][kc];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) v0[mt] = pq[2][mt];
          } else {
#pragma unroll
            for (int qa = 0; qa < MT; ++qa) {
              const float p0 = sigmoid_f(((sc0[qa][0] + sc[qa][0]) - (sc0[qa][1] + sc[qa][1])) * a.inv_sqrt_dk);
              const float p1 = 1.0f - p0;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float o0 = fmaf(p0, v0[0][r], p1 * v0[1][r]), o1 = fmaf(p0, pq[2][0][r], p1 * pq[2][1][r]);
                const __bf16 h0 = (__bf16)o0, h1 = (__bf16)o1;
                gh[qa][r] = h0;
                gw[qa][r] = (__bf16)(o0 - (float)h0);
                gh[qa][4 + r] = h1;
                gw[qa][4 + r] = (__bf16)(o1 - (float)h1);
              }
            }
          }
        } else {
#pragma unroll
        for (int qa = 0; qa < MT; ++qa) {
          // softmax over two keys: p0 = 1 / (1 + exp(s1 - s0)), p1 = 1 - p0
          const float p0 = sigmoid_f((sc[qa][0] - sc[qa][1]) * a.inv_sqrt_dk);
          const float p1 = 1.0f - p0;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float o = fmaf(p0, pq[2][0][r], p1 * pq[2][1][r]);
            const __bf16 hb = (__bf16)o;
            gh[qa][4 * hh + r] = hb;
            gw[qa][4 * hh + r] = (__bf16)(o - (float)hb);
          }
        }
        }
      }
      // ---- output-projection K step of this head pair -------------------------------------------------------------
#pragma unroll
      for (int ft = 0; ft < FT; ft += 2) {
        if (ft + 2 < FT) {
          ld_dn(ft + 2, fb[(ft + 2) & 3]);
          ld_dn(ft + 3, fb[(ft + 3) & 3]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 wh0 = *reinterpret_cast<const bf16x8*>(&fb[ft & 3][0]);
        const bf16x8 wl0 = *reinterpret_cast<const bf16x8*>(&fb[ft & 3][1]);
        const bf16x8 wh1 = *reinterpret_cast<const bf16x8*>(&fb[(ft + 1) & 3][0]);
        const bf16x8 wl1 = *reinterpret_cast<const bf16x8*>(&fb[(ft + 1) & 3][1]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[ft][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh0, gh[mt], acc[ft][mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[ft + 1][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh1, gh[mt], acc[ft + 1][mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[ft][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh0, gw[mt], acc[ft][mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[ft + 1][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh1, gw[mt], acc[ft + 1][mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[ft][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl0, gh[mt], acc[ft][mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[ft + 1][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl1, gh[mt], acc[ft + 1][mt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      dma_barrier();                             // output-projection fragments consumed; pair c+1's q/k/v
#if (SEPR_SPK_ABL & 1) == 0
      if (c + 1 < NCH) dma_w2(c + 1);            // fragments have landed
#endif
    }

    // ---- epilogue: y = x + ls * (acc + bo), two waves at a time through LDS -------------------------------------
    float* const Os = reinterpret_cast<float*>(wl);
    constexpr int WPP = 64 / (16 * MT);      // waves per 64-row epilogue pass
    constexpr int Q = F / 4;                 // float4 per row
    constexpr int RPP = NT / Q;              // rows per pass
    constexpr int NP = 64 / RPP;
    static_assert(64 % RPP == 0, "epilogue pass partition");
    const int q4 = tid % Q, rr = tid / Q;
    const float4 bo = ld4(a.bo + 4 * q4), lsv = ld4(a.ls + 4 * q4);
#pragma unroll 1
    for (int half = 0; half < EH; ++half) {
      if (half > 0) __syncthreads();   // previous pass fully stored (the chunk loop ended on a barrier)
      float4 xr[NP];
      long long mrow[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int row = rr + p * RPP;          // 0..63: WPP waves x (speaker, frame)
        const int ww = WPP * half + row / (16 * MT), lr = row % (16 * MT);
        const int f = tile * TILE + ww * 16 + (lr & 15);
        const bool ok = f < a.NF;
        const int b = ok ? f / a.T : 0, t = ok ? f - b * a.T : 0;
        const long long m = (long long)(b * MT + (lr >> 4)) * a.T + t;
        mrow[p] = ok ? m : -1;
        xr[p] = ld4(a.x + m * F + 4 * q4);
      }
      if (w / WPP == half) {
        float* base = Os + (w % WPP) * (16 * MT) * OS;
#pragma unroll
        for (int ft = 0; ft < FT; ++ft)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            const f32x4 v = acc[ft][mt];
            st4(base + (16 * mt + fi) * OS + 16 * ft + 4 * fg, make_float4(v[0], v[1], v[2], v[3]));
          }
      }
      __syncthreads();
      {
#pragma clang fp contract(off)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          if (mrow[p] >= 0) {
            const float4 o = ld4(Os + (rr + p * RPP) * OS + 4 * q4);
            st4(a.y + mrow[p] * F + 4 * q4,
                make_float4(fmaf(o.x + bo.x, lsv.x, xr[p].x), fmaf(o.y + bo.y, lsv.y, xr[p].y),
                            fmaf(o.z + bo.z, lsv.z, xr[p].z), fmaf(o.w + bo.w, lsv.w, xr[p].w)));
          }
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// spk_hs_kernel (round 6): the HIDDEN-SPLIT form for launches of at most one tile per CU (batch 1) - the design of gcfn_hs_kernel
// (sepr_gcfn_fused.hip).  The four waves hold the SAME 16*FMT frames of both speakers; wave w owns head pair w (q, k, v of its two heads,
// the 2x2 speaker attention, the mix), then output tiles 2w, 2w+1 over the four K steps.  Weight fragments have one reader each and go
// global -> registers; the mixed planes cross the waves once through LDS in the B-fragment lane layout they already have; ONE barrier.
// Same packed weights, same products in the same order per accumulator as spk_fused_kernel<128, 4>: bit-identical.  F = 128, 16-channel heads.
// ---------------------------------------------------------------------------------------------------------------------
template <int FMT>
__global__ __launch_bounds__(256, 1) void spk_hs_kernel(const SpkFusedArgs a) {
  constexpr int F = 128, KS = F / 32, NW = 4, NCH = F / 32, FT = F / 16, FTW = FT / NW, S = 2, MT = S * FMT;
  constexpr int W1F_U4 = 6 * KS * 2 * 64, CS_U4 = 256, W1_U4 = W1F_U4 + CS_U4, W2_U4 = FT * 2 * 64;
  constexpr int TILE = 16 * FMT;
  static_assert(NCH == NW && FTW == 2, "one head pair and two output tiles per wave");
  __shared__ __attribute__((aligned(16))) uint4 hs[NCH * 2 * MT * 64];   // mixed tensor: [K step = head pair][plane][speaker x frame tile][lane]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fi = lane & 15, fg = lane >> 4;
  const int ws = __builtin_amdgcn_readfirstlane(w);
  const unsigned loff = (unsigned)lane * 16u;
  const uint4* const W1g = static_cast<const uint4*>(a.w1p) + (long long)ws * W1_U4;
  const uint4* const W2g = static_cast<const uint4*>(a.w2p);
  auto ldu = [&](const uint4* base, int blk) -> uint4 {   // 16 bytes of this lane from the 1 KiB block blk behind the wave-uniform base
    return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base + blk * 64) + loff);
  };
  const int f0 = blockIdx.x * TILE;
  // rows of this lane: index m2 = 2 * frame tile + speaker
  long long mrow[MT];
  bool okf[FMT];
#pragma unroll
  for (int q = 0; q < FMT; ++q) {
    const int f = f0 + 16 * q + fi;
    okf[q] = f < a.NF;
    const int b = okf[q] ? f / a.T : 0, t = okf[q] ? f - b * a.T : 0;
#pragma unroll
    for (int sp = 0; sp < S; ++sp) mrow[2 * q + sp] = (long long)(b * S + sp) * a.T + t;
  }
  // request order = arrival order: biases, frames, the head pair's fragments
  float4 bias[3][2];
  {
    const float* csg = reinterpret_cast<const float*>(W1g + W1F_U4) + 4 * fg;
#pragma unroll
    for (int tt = 0; tt < 3; ++tt)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) bias[tt][hh] = ld4(csg + (tt * 2 + hh) * 16);
  }
  float xv[MT][KS][8];
#pragma unroll
  for (int m2 = 0; m2 < MT; ++m2) {
    const float* xp = a.x + mrow[m2] * F + 8 * fg;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float4 p = ld4(xp + 32 * ks), q = ld4(xp + 32 * ks + 4);
      xv[m2][ks][0] = p.x; xv[m2][ks][1] = p.y; xv[m2][ks][2] = p.z; xv[m2][ks][3] = p.w;
      xv[m2][ks][4] = q.x; xv[m2][ks][5] = q.y; xv[m2][ks][6] = q.z; xv[m2][ks][7] = q.w;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  uint4 wr[2][3 * KS * 2];     // head hh of the pair: [K step][q | k | v][plane]; wr[0] later takes this wave's W_out fragments [K step][tile][plane]
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int tt = 0; tt < 3; ++tt)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) wr[hh][(ks * 3 + tt) * 2 + pl] = ldu(W1g, ((tt * 2 + hh) * KS + ks) * 2 + pl);
  __builtin_amdgcn_sched_barrier(0);
  bf16x8 xh[MT][KS], xl[MT][KS];
#pragma unroll
  for (int m2 = 0; m2 < MT; ++m2) {
    float (&v)[KS][8] = xv[m2];
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[ks][e];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.0f / F);
    float d = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float c = v[ks][e] - mean;
        d = fmaf(c, c, d);
      }
    d += __shfl_xor(d, 16, 64);
    d += __shfl_xor(d, 32, 64);
    const float rstd = 1.0f / sqrtf(d * (1.0f / F) + a.eps);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 h, l;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xn = (v[ks][e] - mean) * rstd;
        const __bf16 hh = (__bf16)xn;
        h[e] = hh;
        l[e] = (__bf16)(xn - (float)hh);
      }
      xh[m2][ks] = h;
      xl[m2][ks] = l;
    }
  }

  // ---- phase 1: this wave's head pair ---------------------------------------------------------------------------------------------
  bf16x8 gh[MT], gw[MT];          // mixed values (bf16 hi / lo) in output-projection k-slot order
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    f32x4 pq[3][MT];
#pragma unroll
    for (int tt = 0; tt < 3; ++tt)
#pragma unroll
      for (int m2 = 0; m2 < MT; ++m2) pq[tt][m2] = (f32x4){bias[tt][hh].x, bias[tt][hh].y, bias[tt][hh].z, bias[tt][hh].w};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 wh[3], wlo[3];
#pragma unroll
      for (int tt = 0; tt < 3; ++tt) {
        wh[tt] = *reinterpret_cast<const bf16x8*>(&wr[hh][(ks * 3 + tt) * 2]);
        wlo[tt] = *reinterpret_cast<const bf16x8*>(&wr[hh][(ks * 3 + tt) * 2 + 1]);
      }
#pragma unroll
      for (int tt = 0; tt < 3; ++tt)
#pragma unroll
        for (int m2 = 0; m2 < MT; ++m2) pq[tt][m2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[tt], xh[m2][ks], pq[tt][m2], 0, 0, 0);
#pragma unroll
      for (int tt = 0; tt < 3; ++tt)
#pragma unroll
        for (int m2 = 0; m2 < MT; ++m2) pq[tt][m2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[tt], xl[m2][ks], pq[tt][m2], 0, 0, 0);
#pragma unroll
      for (int tt = 0; tt < 3; ++tt)
#pragma unroll
        for (int m2 = 0; m2 < MT; ++m2) pq[tt][m2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo[tt], xh[m2][ks], pq[tt][m2], 0, 0, 0);
    }
    if (hh == 0) {   // head 0's fragment registers are free: this wave's W_out fragments (4 K steps x 2 tiles x 2 planes) fly under head 1
      // (scheduling fences: hipcc otherwise sinks the loads down to their first use - and waits for each of them there)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int t = 0; t < FTW; ++t)
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) wr[0][(c * FTW + t) * 2 + pl] = ldu(W2g + (long long)c * W2_U4, (FTW * ws + t) * 2 + pl);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- 2x2 attention across the speakers of each frame (network.py:106-122 with T = S) ----------------------------------------
#pragma unroll
    for (int q = 0; q < FMT; ++q) {
      float sc[S][S];
#pragma unroll
      for (int qa = 0; qa < S; ++qa)
#pragma unroll
        for (int kc = 0; kc < S; ++kc) {
          float p = pq[0][2 * q + qa][0] * pq[1][2 * q + kc][0];
          p = fmaf(pq[0][2 * q + qa][1], pq[1][2 * q + kc][1], p);
          p = fmaf(pq[0][2 * q + qa][2], pq[1][2 * q + kc][2], p);
          p = fmaf(pq[0][2 * q + qa][3], pq[1][2 * q + kc][3], p);
          p += __shfl_xor(p, 16, 64);          // the tile's 16 channels live in the 4 lane groups
          p += __shfl_xor(p, 32, 64);
          sc[qa][kc] = p;
        }
#pragma unroll
      for (int qa = 0; qa < S; ++qa) {
        // softmax over two keys: p0 = 1 / (1 + exp(s1 - s0)), p1 = 1 - p0
        const float p0 = sigmoid_f((sc[qa][0] - sc[qa][1]) * a.inv_sqrt_dk);
        const float p1 = 1.0f - p0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float o = fmaf(p0, pq[2][2 * q + 0][r], p1 * pq[2][2 * q + 1][r]);
          const __bf16 hb = (__bf16)o;
          gh[2 * q + qa][4 * hh + r] = hb;
          gw[2 * q + qa][4 * hh + r] = (__bf16)(o - (float)hb);
        }
      }
    }
  }
#pragma unroll
  for (int m2 = 0; m2 < MT; ++m2) {
    *reinterpret_cast<bf16x8*>(&hs[((w * 2 + 0) * MT + m2) * 64 + lane]) = gh[m2];
    *reinterpret_cast<bf16x8*>(&hs[((w * 2 + 1) * MT + m2) * 64 + lane]) = gw[m2];
  }
  // the residual rows (this lane's 2 x 4 channels per row) fly under the barrier and phase 2
  float4 xr[MT][FTW];
#pragma unroll
  for (int m2 = 0; m2 < MT; ++m2)
#pragma unroll
    for (int t = 0; t < FTW; ++t) xr[m2][t] = ld4(a.x + mrow[m2] * F + 32 * w + 16 * t + 4 * fg);
  __syncthreads();

  // ---- phase 2: output tiles 2w, 2w+1 over the four K steps, in order -----------------------------------------------------------------
  f32x4 acc[FTW][MT];
#pragma unroll
  for (int t = 0; t < FTW; ++t)
#pragma unroll
    for (int m2 = 0; m2 < MT; ++m2) acc[t][m2] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    bf16x8 ah[MT], aw[MT];
#pragma unroll
    for (int m2 = 0; m2 < MT; ++m2) {
      ah[m2] = *reinterpret_cast<const bf16x8*>(&hs[((c * 2 + 0) * MT + m2) * 64 + lane]);
      aw[m2] = *reinterpret_cast<const bf16x8*>(&hs[((c * 2 + 1) * MT + m2) * 64 + lane]);
    }
#pragma unroll
    for (int t = 0; t < FTW; ++t) {
      const bf16x8 wh = *reinterpret_cast<const bf16x8*>(&wr[0][(c * FTW + t) * 2]);
      const bf16x8 wlo = *reinterpret_cast<const bf16x8*>(&wr[0][(c * FTW + t) * 2 + 1]);
#pragma unroll
      for (int m2 = 0; m2 < MT; ++m2) acc[t][m2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah[m2], acc[t][m2], 0, 0, 0);
#pragma unroll
      for (int m2 = 0; m2 < MT; ++m2) acc[t][m2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, aw[m2], acc[t][m2], 0, 0, 0);
#pragma unroll
      for (int m2 = 0; m2 < MT; ++m2) acc[t][m2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, ah[m2], acc[t][m2], 0, 0, 0);
    }
  }
  // ---- epilogue: y = x + ls * (acc + bo); fragment row 4 fg + r of tile 2w + t is channel 32 w + 16 t + 4 fg + r --------------------
  {
#pragma clang fp contract(off)
#pragma unroll
    for (int t = 0; t < FTW; ++t) {
      const int ch = 32 * w + 16 * t + 4 * fg;
      const float4 bo = ld4(a.bo + ch), lsv = ld4(a.ls + ch);
#pragma unroll
      for (int m2 = 0; m2 < MT; ++m2) {
        if (okf[m2 >> 1]) {
          const f32x4 o = acc[t][m2];
          st4(a.y + mrow[m2] * F + ch, make_float4(fmaf(o[0] + bo.x, lsv.x, xr[m2][t].x), fmaf(o[1] + bo.y, lsv.y, xr[m2][t].y),
                                                   fmaf(o[2] + bo.z, lsv.z, xr[m2][t].z), fmaf(o[3] + bo.w, lsv.w, xr[m2][t].w)));
        }
      }
    }
  }
}

int launch_spk_fused(const SpkFusedArgs& a, int F, int site, hipStream_t stream) {
  if (a.NF <= 0) return SEPR_OK;
  if (!a.x || !a.y || !a.w1p || !a.w2p || !a.bo || !a.ls || a.T <= 0 || a.NF % a.T != 0) return SEPR_EINVAL;
  if (a.x == a.y) return SEPR_EINVAL;
  if (F != 128 && F != 256) return SEPR_EINVAL;
  long long slot = -1;
  const bool timed = prof_begin(site, stream, &slot);
  const int ntiles = (a.NF + 63) / 64;
  const int cap = persistent_grid();
  const int grid = ntiles < cap ? ntiles : cap;
  // at most one tile per CU: the hidden-split form, 16- or 32-frame tiles (SEPR_SPK_HS=0 switches it off)
  static const bool hs_on = [] {
    const char* e = getenv("SEPR_SPK_HS");
    return !(e && e[0] == '0');
  }();
  const int cus = cap / 2;
  if (F == 128 && hs_on && (a.NF + 15) / 16 <= cus) hipLaunchKernelGGL((spk_hs_kernel<1>), dim3((a.NF + 15) / 16), dim3(256), 0, stream, a);
  else if (F == 128 && hs_on && (a.NF + 31) / 32 <= cus) hipLaunchKernelGGL((spk_hs_kernel<2>), dim3((a.NF + 31) / 32), dim3(256), 0, stream, a);
  else
  if (F == 256) hipLaunchKernelGGL((spk_fused_kernel<256, 4, true>), dim3(ntiles < cap / 2 ? ntiles : cap / 2), dim3(256), 0, stream, a);   // 32-channel heads, one workgroup per CU
  else hipLaunchKernelGGL((spk_fused_kernel<128, 4>), dim3(grid), dim3(256), 0, stream, a);
  // algorithmic FLOPs: q/k/v and output projections of both speakers' rows
  if (timed) prof_end(slot, 2.0 * a.NF * (2.0 * F * 3 * F + 2.0 * F * F), stream);
  SEPR_CHECK_LAUNCH("spk_fused_kernel");
  return SEPR_OK;
}

}  // namespace sepr
