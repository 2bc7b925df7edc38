// The two projection stages of the CLA block on the bf16x3 core, row-stationary like sepr_gcfn_fused.hip
// (reference modules/network.py:159-187):
//
//   cla_head_kernel   u = GLU( Linear_F->2F( LayerNorm(x) ) )                               (:175-177)
//   [dwconv_same_pk_kernel: c = depthwise k=65 conv of u along frames, sepr_pointwise.hip]   (:178-180)
//   cla_tail_kernel   y = x + layer_scale * Linear_2F->F( GELU( BN( Linear_F->2F(c) ) ) )    (:181-187)
//
// Neither the LayerNorm statistics nor the [rows, 2F] hidden tensor between linear2 and linear3 reach HBM; a CLA
// block moves 7 F floats per frame (x, u, u, c, c, x, y) instead of 11.5 F and takes 3 launches instead of 5.
//
// Both kernels: a wave owns 32 consecutive frames (frame = 2*fi + mt, no halo), loads them once straight into the
// MFMA B-fragment layout, keeps them as bf16 hi/lo in registers; weights arrive per hidden chunk by LDS-DMA in
// fragment order and are read two MFMA groups ahead; bias = accumulator init; outputs leave through an LDS
// staging tile as row-contiguous 512 B stores.
//   head: chunk = 32 output channels (value tiles v0 v1 + gate tiles g0 g1, the GCFN up-projection layout); the
//         gated values are collected in a [F x 32 frames] register tile.
//   tail: chunk = 64 hidden channels (4 tiles up, 2 K steps down); GELU (gelu_fast, 1.5e-7 erf) in registers; the activated
//         values are the B fragments of the down-projection (k-slot-ordered weights, as in the GCFN kernel).
#include "sepr_gemm_epi.h"
#include <stdlib.h>

namespace sepr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct ClaFusedArgs {
  const float* x;     // head: block input [M,F];  tail: conv output c [M,F]
  const float* res;   // tail: block input (residual) [M,F]
  float* y;           // head: u [M,F];  tail: block output [M,F]
  int M;
  const void* w1p;    // head: per 32-channel chunk [v0 v1 g0 g1][F/32][plane][64][8] bf16 (gamma folded) + 4 KB constants
                      //       [4 tiles][16] biases;  tail: per 64-channel chunk [4 tiles][F/32][plane][64][8] + 4 KB [4][16]
  const void* w2p;    // tail: [2F/32][F/16][plane][64][8] bf16, k-slot order
  const float* b3;    // tail [F]
  const float* ls;    // tail [F]
  float eps;
  const float* att;   // gate: pooled attention output [n, Tp, F]
  int T, Tp, fac;     // gate: frames per sequence, pooled frames, T / Tp
  int ldy, pool;      // q / k / v form (launch_ega_qkv): output row stride (3F) and the pooling factor of the input rows (row m = mean of rows
                      // m*pool .. m*pool + pool-1 of x, adaptive_avg_pool1d with T % Tp == 0); 0 elsewhere
  // gate with the attention's output projection folded in (o != null): att rows of a tile = ls_o * (Linear_out(o rows) + b_o) are computed by
  // the tile's workgroup into att_w (= att) before its epilogue reads them - the separate projection launch is gone
  const float* o;     // attention output before linear_out [Mp, F]
  float* att_w;       // = att, writable
  const void* wop;    // linear_out weights as fragments [F/16][F/32][plane][64][8] bf16 (pack.py::pack_outproj_fused)
  const float* bo;    // [F]
  const float* lso;   // [F] the attention's LayerScale
  int Mp;             // pooled rows
};

namespace {

constexpr int CF_NW = 4, CF_NT = 256, CF_MT = 2;

// ---- shared pieces -----------------------------------------------------------------------------------------------
// 32 frames of a wave -> bf16 hi/lo B fragments; NORM: LayerNorm without affine (gamma/beta live in the weights)
template <int F, bool NORM, int MT = CF_MT>
__device__ __forceinline__ void load_frames(const float* __restrict__ X, int m0, int M, float eps, int fi, int fg,
                                            bf16x8 (&xh)[MT][F / 32], bf16x8 (&xl)[MT][F / 32]) {
  constexpr int KS = F / 32;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + MT * fi + mt;
    const bool valid = m < M;
    const float* xp = X + (long long)(valid ? m : 0) * F + 8 * fg;
    float v[KS][8];
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float4 p = ld4(xp + 32 * ks), q = ld4(xp + 32 * ks + 4);
      v[ks][0] = p.x; v[ks][1] = p.y; v[ks][2] = p.z; v[ks][3] = p.w;
      v[ks][4] = q.x; v[ks][5] = q.y; v[ks][6] = q.z; v[ks][7] = q.w;
      if (NORM) {
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[ks][e];
      }
    }
    float mean = 0.f, rstd = 1.f;
    if (NORM) {
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      mean = s * (1.0f / F);
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
      rstd = 1.0f / sqrtf(d * (1.0f / F) + eps);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 h, l;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xn = NORM ? (v[ks][e] - mean) * rstd : v[ks][e];
        const __bf16 hh = (__bf16)xn;
        h[e] = hh;
        l[e] = (__bf16)(xn - (float)hh);
      }
      xh[mt][ks] = h;
      xl[mt][ks] = l;
    }
  }
}

// the q / k / v form's frames: row m = mean of `pool` consecutive rows of X (summed in row order, then * 1/pool, like pool_stats_kernel), LayerNorm
template <int F, int MT>
__device__ __forceinline__ void load_frames_pooled(const float* __restrict__ X, int m0, int M, int pool, float eps, int fi, int fg,
                                                   bf16x8 (&xh)[MT][F / 32], bf16x8 (&xl)[MT][F / 32]) {
  constexpr int KS = F / 32;
  const float inv = 1.0f / (float)pool;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + MT * fi + mt;
    const bool valid = m < M;
    const float* xp = X + (long long)(valid ? m : 0) * pool * F + 8 * fg;
    float v[KS][8];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float4 p = ld4(xp + 32 * ks), q = ld4(xp + 32 * ks + 4);
      v[ks][0] = p.x; v[ks][1] = p.y; v[ks][2] = p.z; v[ks][3] = p.w;
      v[ks][4] = q.x; v[ks][5] = q.y; v[ks][6] = q.z; v[ks][7] = q.w;
    }
    // RB rows requested together (clamped addresses past the last one), added in row order: the sum does not depend on RB
    constexpr int RB = MT == 1 ? 4 : 2;
    for (int j0 = 1; j0 < pool; j0 += RB) {
      float4 pq[RB][KS][2];
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const float* xq = xp + (long long)(j0 + i < pool ? j0 + i : 0) * F;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          pq[i][ks][0] = ld4(xq + 32 * ks);
          pq[i][ks][1] = ld4(xq + 32 * ks + 4);
        }
      }
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        if (j0 + i < pool) {
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const float4 p = pq[i][ks][0], q = pq[i][ks][1];
            v[ks][0] += p.x; v[ks][1] += p.y; v[ks][2] += p.z; v[ks][3] += p.w;
            v[ks][4] += q.x; v[ks][5] += q.y; v[ks][6] += q.z; v[ks][7] += q.w;
          }
        }
      }
    }
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[ks][e] *= inv;
        s += v[ks][e];
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
    const float rstd = 1.0f / sqrtf(d * (1.0f / F) + eps);
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

#ifndef SEPR_CF_ASMDMA
#define SEPR_CF_ASMDMA 1   // inline-asm LDS-DMA (sepr_common.h glds16_asm): the copies are waited for at the chunk barriers only
#endif
__device__ __forceinline__ void dma_blocks(const uint4* gbase, uint4* lbase, int nblk, int lane, int w) {
  // 1 KiB per wave instruction; per-lane byte offset laundered so the addresses are not hoisted and spilled
  unsigned loff = (unsigned)lane * 16u;
  asm volatile("" : "+v"(loff));
  [[maybe_unused]] const int ws = __builtin_amdgcn_readfirstlane(w);
#pragma unroll
  for (int i = 0; i < 17; ++i) {       // (17: the head's chunk at F = 256 is 64 KB of fragments + the 4 KB constants block)
    if (i >= nblk) break;
#if SEPR_CF_ASMDMA
    glds16_asm(gbase + (i * CF_NW + ws) * 64, loff, __builtin_amdgcn_readfirstlane(lds_addr(lbase + (i * CF_NW + ws) * 64)));
#else
    const int blk = i * CF_NW + w;
    const char* src = reinterpret_cast<const char*>(gbase + blk * 64) + loff;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lbase + blk * 64), 16, 0, 0);
#endif
  }
}
__device__ __forceinline__ void dma_barrier() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// [F x 32 frames] register tile -> LDS -> row-contiguous stores
//   ST_PLAIN: y = tile;  ST_RES: y = res + ls * (tile + b);  ST_GATE: y = res + sigmoid(tile) * att[seq, t / fac]
constexpr int ST_PLAIN = 0, ST_RES = 1, ST_GATE = 2;
template <int F, int MODE, int MT = CF_MT>
__device__ __forceinline__ void store_tile(const f32x4 (&acc)[F / 16][MT], float* Os, const ClaFusedArgs& a, int tile0,
                                           int tid, int w, int fi, int fg, int ldy = F, int col = 0) {
  constexpr int FT = F / 16, OS = F + 4;
  constexpr int EH = (16 * MT * CF_NW) / 64, WPP = 64 / (16 * MT);
  constexpr int Q = F / 4, RPP = CF_NT / Q, NP = 64 / RPP;
  static_assert(64 % RPP == 0, "epilogue pass partition");
  const int q4 = tid % Q, rr = tid / Q;
  constexpr bool RES = (MODE != ST_PLAIN);
  float4 bb = zero4(), lsv = zero4();
  if (MODE == ST_RES) {
    bb = ld4(a.b3 + 4 * q4);
    lsv = ld4(a.ls + 4 * q4);
  }
#pragma unroll 1
  for (int half = 0; half < EH; ++half) {
    if (half > 0) __syncthreads();   // previous pass fully stored
    float4 xr[NP], ar[NP];
    int mrow[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int m = tile0 + 64 * half + rr + p * RPP;     // staging row = frame inside the tile
      mrow[p] = m < a.M ? m : -1;
      const int mc = m < a.M ? m : 0;
      if (RES) xr[p] = ld4(a.res + (long long)mc * F + 4 * q4);
      if (MODE == ST_GATE) {
        const int seq = mc / a.T, t = mc - seq * a.T;
        ar[p] = ld4(a.att + ((long long)seq * a.Tp + t / a.fac) * F + 4 * q4);
      }
    }
    if (w / WPP == half) {
      float* base = Os + (w % WPP) * (16 * MT) * OS;
#pragma unroll
      for (int ft = 0; ft < FT; ++ft)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const f32x4 v = acc[ft][mt];
          st4(base + (MT * fi + mt) * OS + 16 * ft + 4 * fg, make_float4(v[0], v[1], v[2], v[3]));
        }
    }
    __syncthreads();
    {
#pragma clang fp contract(off)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        if (mrow[p] >= 0) {
          const float4 o = ld4(Os + (rr + p * RPP) * OS + 4 * q4);
          float* dst = a.y + (long long)mrow[p] * ldy + col + 4 * q4;
          if (MODE == ST_RES) {
            st4(dst, make_float4(fmaf(o.x + bb.x, lsv.x, xr[p].x), fmaf(o.y + bb.y, lsv.y, xr[p].y),
                                 fmaf(o.z + bb.z, lsv.z, xr[p].z), fmaf(o.w + bb.w, lsv.w, xr[p].w)));
          } else if (MODE == ST_GATE) {
            st4(dst, make_float4(fmaf(sigmoid_f(o.x), ar[p].x, xr[p].x), fmaf(sigmoid_f(o.y), ar[p].y, xr[p].y),
                                 fmaf(sigmoid_f(o.z), ar[p].z, xr[p].z), fmaf(sigmoid_f(o.w), ar[p].w, xr[p].w)));
          } else {
            st4(dst, o);
          }
        }
      }
    }
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// head: u = GLU(Linear(LayerNorm(x)))
// ---------------------------------------------------------------------------------------------------------------------
// GATE = false: the CLA head.  GATE = true: the EGA gate  y = x + sigmoid(Linear_F->F(LayerNorm(x))) * upsample(att)
// (network.py:132-135,151-153): the same chunk walk with the four tiles of a chunk as four plain output tiles
// (64 output channels per chunk) and the gate applied in the store pass.
// F = 256 (Large, round 6): 128 registers of frame planes + 128 of output tile - the ONE-wave-per-SIMD regime (one 139 KB workgroup per CU)
// MT = 1 (round 6): 16-frame waves, 64-frame tiles - for launches with fewer 128-frame tiles than half the CUs (batch 1: Engine._inference_sample): twice
// the workgroups, half the dependent chain per workgroup; a frame's arithmetic does not depend on the tiling (bit-identical)
// QKV (round 6; with GATE's plain-tile chunks): the EGA attention's input side in ONE launch - adaptive average pooling of the frames,
// LayerNorm, q / k / v projection (network.py:146, :99-102): three output groups of F channels, each the gate's chunk walk, stored to the
// [rows, 3F] tensor the attention kernel reads.  Replaces pool_stats_kernel + the generic projection (and the pooled copy of x).
template <int F, bool GATE, int MT = CF_MT, bool QKV = false>
__global__ __launch_bounds__(CF_NT, F > 128 ? 1 : 2) void cla_head_kernel(const ClaFusedArgs a) {
  static_assert(!QKV || GATE, "the q / k / v form uses the plain-tile chunks");
  constexpr int NG = QKV ? 3 : 1;
  constexpr int NW = CF_NW, NT = CF_NT;
  constexpr int TILE = 16 * MT * NW;
  constexpr int KS = F / 32;
  constexpr int NCH = GATE ? F / 64 : F / 32;   // output channels per chunk: 64 plain or 32 gated
  constexpr int FT = F / 16;
  constexpr int W1F_U4 = 4 * KS * 2 * 64;
  constexpr int CS_U4 = 256;
  constexpr int W1_U4 = W1F_U4 + CS_U4;
  constexpr int OS = F + 4;
  // two (fragments + constants) buffers: chunk c+1 is copied while chunk c is multiplied
  __shared__ __attribute__((aligned(16))) uint4 wl[2 * W1_U4];
  static_assert(sizeof(uint4) * 2 * W1_U4 >= sizeof(float) * 64 * OS, "epilogue staging must fit");
  static_assert(W1_U4 % NT == 0 && W1_U4 / NT <= 17, "copy partition");

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fi = lane & 15, fg = lane >> 4;
  const int ntiles = (a.M + TILE - 1) / TILE;
  const uint4* const W1g = static_cast<const uint4*>(a.w1p);

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    bf16x8 xh[MT][KS], xl[MT][KS];
    if constexpr (QKV) load_frames_pooled<F, MT>(a.x, tile * TILE + w * 16 * MT, a.M, a.pool, a.eps, fi, fg, xh, xl);
    else load_frames<F, true, MT>(a.x, tile * TILE + w * 16 * MT, a.M, a.eps, fi, fg, xh, xl);
    if constexpr (GATE && !QKV) {
      if (a.o) {
        // ---- folded output projection of the attention: this tile's pooled rows p0 .. p0 + R-1 (fac | TILE: they belong to no other tile).
        //      Wave w owns output tiles 2w, 2w+1; rows as B fragments straight from o, weights global -> registers (one reader per fragment).
        constexpr int FTW = FT / NW;
        static_assert(FTW * NW == FT, "output tiles over the waves");
        const int R = TILE / a.fac, p0 = (tile * TILE) / a.fac;
        const int ws = __builtin_amdgcn_readfirstlane(w);
        const unsigned loff = (unsigned)lane * 16u;
        const uint4* const Wob = static_cast<const uint4*>(a.wop) + (long long)(FTW * ws) * KS * 2 * 64;
        uint4 wo[FTW][KS][2];
#pragma unroll
        for (int t = 0; t < FTW; ++t)
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
              wo[t][ks][pl] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(Wob + ((t * KS + ks) * 2 + pl) * 64) + loff);
        float4 bov[FTW], lov[FTW];
#pragma unroll
        for (int t = 0; t < FTW; ++t) {
          bov[t] = ld4(a.bo + 16 * (FTW * w + t) + 4 * fg);
          lov[t] = ld4(a.lso + 16 * (FTW * w + t) + 4 * fg);
        }
        for (int nt = 0; nt * 16 < R; ++nt) {
          const int pr = p0 + 16 * nt + fi;
          const bool ok = 16 * nt + fi < R && pr < a.Mp;
          const float* op = a.o + (long long)(ok ? pr : 0) * F + 8 * fg;
          bf16x8 oh[KS], ol[KS];
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const float4 p = ld4(op + 32 * ks), q = ld4(op + 32 * ks + 4);
            const float v[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const __bf16 hh = (__bf16)v[e];
              oh[ks][e] = hh;
              ol[ks][e] = (__bf16)(v[e] - (float)hh);
            }
          }
          f32x4 pa[FTW];
#pragma unroll
          for (int t = 0; t < FTW; ++t) pa[t] = (f32x4){bov[t].x, bov[t].y, bov[t].z, bov[t].w};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int t = 0; t < FTW; ++t) {
              const bf16x8 wh = *reinterpret_cast<const bf16x8*>(&wo[t][ks][0]), wlo = *reinterpret_cast<const bf16x8*>(&wo[t][ks][1]);
              pa[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, oh[ks], pa[t], 0, 0, 0);
              pa[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ol[ks], pa[t], 0, 0, 0);
              pa[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, oh[ks], pa[t], 0, 0, 0);
            }
          if (ok) {
#pragma unroll
            for (int t = 0; t < FTW; ++t)
              st4(a.att_w + (long long)pr * F + 16 * (FTW * w + t) + 4 * fg,
                  make_float4(pa[t][0] * lov[t].x, pa[t][1] * lov[t].y, pa[t][2] * lov[t].z, pa[t][3] * lov[t].w));
          }
        }
        // (the stores reach the other waves of the workgroup through the __syncthreads() of the chunk walk below: the epilogue reads them after it)
      }
    }
    // (small launches: the three output groups are three workgroups - grid.y = 3 - each with its own copy of the frames)
    const int g0 = gridDim.y > 1 ? (int)blockIdx.y : 0, g1 = gridDim.y > 1 ? g0 + 1 : NG;
#pragma unroll 1
    for (int grp = g0; grp < g1; ++grp) {
    const uint4* const Wg = W1g + (long long)grp * NCH * W1_U4;
    f32x4 acc[FT][MT];                   // gated values: channel tile ft = 2*c + j

    __syncthreads();   // the previous tile's (group's) epilogue staging is fully consumed
    dma_blocks(Wg, wl, W1_U4 / NT, lane, w);
    dma_barrier();
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const uint4* const w1s = wl + (c & 1) * W1_U4;
      if (c + 1 < NCH) dma_blocks(Wg + (long long)(c + 1) * W1_U4, wl + ((c + 1) & 1) * W1_U4, W1_U4 / NT, lane, w);
      auto ld_up = [&](int j, int g, uint4 (&d)[2]) {   // g = 2*ks + (0 value tile | 1 gate tile)
        const uint4* p = w1s + ((((g & 1) * 2 + j) * KS + (g >> 1)) * 2) * 64 + lane;
        d[0] = p[0];
        d[1] = p[64];
      };
      const float* cs = reinterpret_cast<const float*>(w1s + W1F_U4) + 4 * fg;   // [v0 v1 g0 g1][16]
      uint4 fb[3][2];
      ld_up(0, 0, fb[0]);
      ld_up(0, 1, fb[1]);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4 hv[MT], hg[MT];
        {
          const float4 bv = ld4(cs + j * 16), bg = ld4(cs + (2 + j) * 16);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            hv[mt] = (f32x4){bv.x, bv.y, bv.z, bv.w};
            hg[mt] = (f32x4){bg.x, bg.y, bg.z, bg.w};
          }
        }
#pragma unroll
        for (int g = 0; g < 2 * KS; ++g) {
          if (g + 2 < 2 * KS) ld_up(j, g + 2, fb[(g + 2) % 3]);
          __builtin_amdgcn_sched_barrier(0);
          const bf16x8 wh = *reinterpret_cast<const bf16x8*>(&fb[g % 3][0]);
          const bf16x8 wlo = *reinterpret_cast<const bf16x8*>(&fb[g % 3][1]);
          const int ks = g >> 1;
          if ((g & 1) == 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) hv[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[mt][ks], hv[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) hv[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl[mt][ks], hv[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) hv[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, xh[mt][ks], hv[mt], 0, 0, 0);
          } else {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) hg[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[mt][ks], hg[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) hg[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl[mt][ks], hg[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) hg[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, xh[mt][ks], hg[mt], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (j == 0) {
          ld_up(1, 0, fb[0]);
          ld_up(1, 1, fb[1]);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if (GATE) {
            acc[4 * c + j][mt] = hv[mt];
            acc[4 * c + 2 + j][mt] = hg[mt];
          } else {
            f32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = hv[mt][r] * sigmoid_f(hg[mt][r]);
            acc[2 * c + j][mt] = o;
          }
        }
      }
      dma_barrier();   // chunk c fully read by every wave, chunk c+1 landed
    }
    store_tile<F, (GATE && !QKV) ? ST_GATE : ST_PLAIN, MT>(acc, reinterpret_cast<float*>(wl), a, tile * TILE, tid, w, fi, fg, QKV ? a.ldy : F, grp * F);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// tail: y = x + ls * Linear3(GELU(Linear2'(c)))      (eval BatchNorm folded into Linear2')
// ---------------------------------------------------------------------------------------------------------------------
template <int F, int MT = CF_MT>
__global__ __launch_bounds__(CF_NT, F > 128 ? 1 : 2) void cla_tail_kernel(const ClaFusedArgs a) {
  constexpr int NW = CF_NW, NT = CF_NT;
  constexpr int TILE = 16 * MT * NW;
  constexpr int KS = F / 32;
  constexpr int NCH = 2 * F / 64;        // 64 hidden channels per chunk
  constexpr int FT = F / 16;
  constexpr int W1F_U4 = 4 * KS * 2 * 64;
  constexpr int CS_U4 = 256;
  constexpr int W1_U4 = W1F_U4 + CS_U4;
  constexpr int W2_U4 = 2 * FT * 2 * 64;  // two K steps of the down-projection
  constexpr int OS = F + 4;
  __shared__ __attribute__((aligned(16))) uint4 wl[W1F_U4 + W2_U4 + 2 * CS_U4];
  static_assert(sizeof(uint4) * (W1F_U4 + W2_U4) >= sizeof(float) * 64 * OS, "epilogue staging must fit");
  static_assert(W1F_U4 % NT == 0 && CS_U4 % NT == 0 && W2_U4 % NT == 0 && W1F_U4 / NT <= 16 && W2_U4 / NT <= 16, "copy partition");
  const uint4* const w1s = wl;
  const uint4* const w2s = wl + W1F_U4;
  uint4* const csl = wl + W1F_U4 + W2_U4;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fi = lane & 15, fg = lane >> 4;
  const int ntiles = (a.M + TILE - 1) / TILE;
  const uint4* const W1g = static_cast<const uint4*>(a.w1p);
  const uint4* const W2g = static_cast<const uint4*>(a.w2p);

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    bf16x8 xh[MT][KS], xl[MT][KS];
    load_frames<F, false, MT>(a.x, tile * TILE + w * 16 * MT, a.M, 0.f, fi, fg, xh, xl);
    f32x4 acc[FT][MT];
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[ft][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto dma_w1 = [&](int c) {
      dma_blocks(W1g + (long long)c * W1_U4, wl, W1F_U4 / NT, lane, w);
      dma_blocks(W1g + (long long)c * W1_U4 + W1F_U4, csl + (c & 1) * CS_U4, CS_U4 / NT, lane, w);
    };
    auto dma_w2 = [&](int c) { dma_blocks(W2g + (long long)c * W2_U4, wl + W1F_U4, W2_U4 / NT, lane, w); };
    auto ld_up = [&](int g, uint4 (&d)[2]) {          // g = tile*KS + ks
      const uint4* p = w1s + (g * 2) * 64 + lane;
      d[0] = p[0];
      d[1] = p[64];
    };
    auto ld_dn = [&](int g, uint4 (&d)[2]) {          // g = kstep*FT + ft
      const uint4* p = w2s + (g * 2) * 64 + lane;
      d[0] = p[0];
      d[1] = p[64];
    };

    __syncthreads();   // the previous tile's epilogue staging is fully consumed
    dma_w1(0);
    dma_w2(0);
    dma_barrier();
    for (int c = 0; c < NCH; ++c) {
      bf16x8 gh[2][MT], gw[2][MT];     // activated values (bf16 hi / lo) per down-projection K step, k-slot order
      const float* cs = reinterpret_cast<const float*>(csl + (c & 1) * CS_U4) + 4 * fg;
      uint4 fb[3][2];
      ld_up(0, fb[0]);
      ld_up(1, fb[1]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f32x4 h[MT];
        {
          const float4 bv = ld4(cs + j * 16);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) h[mt] = (f32x4){bv.x, bv.y, bv.z, bv.w};
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int g = j * KS + ks;
          if (g + 2 < 4 * KS) ld_up(g + 2, fb[(g + 2) % 3]);
          __builtin_amdgcn_sched_barrier(0);
          const bf16x8 wh = *reinterpret_cast<const bf16x8*>(&fb[g % 3][0]);
          const bf16x8 wlo = *reinterpret_cast<const bf16x8*>(&fb[g % 3][1]);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) h[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[mt][ks], h[mt], 0, 0, 0);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) h[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl[mt][ks], h[mt], 0, 0, 0);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) h[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, xh[mt][ks], h[mt], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (j == 3) {
          dma_barrier();                         // every wave has read its up-projection fragments of chunk c;
                                                 // this chunk's down-projection fragments have landed
          if (c + 1 < NCH) dma_w1(c + 1);        // lands under the GELU + down-projection below
          ld_dn(0, fb[0]);
          ld_dn(1, fb[1]);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float o = gelu_fast(h[mt][r]);
            const __bf16 hb = (__bf16)o;
            gh[j >> 1][mt][4 * (j & 1) + r] = hb;
            gw[j >> 1][mt][4 * (j & 1) + r] = (__bf16)(o - (float)hb);
          }
      }
#pragma unroll
      for (int g = 0; g < 2 * FT; ++g) {
        if (g + 2 < 2 * FT) ld_dn(g + 2, fb[(g + 2) % 3]);
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 wh = *reinterpret_cast<const bf16x8*>(&fb[g % 3][0]);
        const bf16x8 wlo = *reinterpret_cast<const bf16x8*>(&fb[g % 3][1]);
        const int s = g / FT, ft = g % FT;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[ft][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, gh[s][mt], acc[ft][mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[ft][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, gw[s][mt], acc[ft][mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[ft][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, gh[s][mt], acc[ft][mt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      dma_barrier();                             // down-projection fragments consumed; chunk c+1's up-projection
      if (c + 1 < NCH) dma_w2(c + 1);            // fragments have landed
    }
    store_tile<F, ST_RES, MT>(acc, reinterpret_cast<float*>(wl), a, tile * TILE, tid, w, fi, fg);
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// cla_tail_hs_kernel (round 6): the HIDDEN-SPLIT form of the tail for launches of at most one tile per CU (batch 1) - the design of
// gcfn_hs_kernel (sepr_gcfn_fused.hip).  The four waves hold the SAME 16*MT frames; wave w owns hidden chunk w (64 channels: four tiles up)
// and then output tiles 2w, 2w+1 over all 8 K steps.  Weight fragments have one reader each and go global -> registers (the 32 KB of a wave's
// chunk are requested at the start; the registers of a tile's fragments take the wave's W3 fragments as soon as the tile is multiplied);
// the activated planes cross the waves once through LDS, in the B-fragment lane layout they already have; ONE barrier.  Same packed
// weights, same products in the same order as cla_tail_kernel: bit-identical (tests/test_gpu_parity.py).  F = 128.
// ---------------------------------------------------------------------------------------------------------------------
template <int MT>
__global__ __launch_bounds__(256, 1) void cla_tail_hs_kernel(const ClaFusedArgs a) {
  constexpr int F = 128, KS = F / 32, NW = 4, NCH = 2 * F / 64, FT = F / 16, FTW = FT / NW, NQ = 2 * NCH;
  constexpr int W1F_U4 = 4 * KS * 2 * 64, CS_U4 = 256, W1_U4 = W1F_U4 + CS_U4, W2_U4 = 2 * FT * 2 * 64;
  constexpr int TILE = 16 * MT;
  static_assert(NCH == NW && FTW == 2, "one hidden chunk and two output tiles per wave");
  __shared__ __attribute__((aligned(16))) uint4 hs[NQ * 2 * MT * 64];   // activated tensor: [K step][plane][frame tile][lane]
  __shared__ __attribute__((aligned(16))) float cst[NW][64];            // this wave's chunk biases [4 tiles][16] (wave-private: no barrier)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fi = lane & 15, fg = lane >> 4;
  const int ws = __builtin_amdgcn_readfirstlane(w);
  const unsigned loff = (unsigned)lane * 16u;
  const uint4* const W1g = static_cast<const uint4*>(a.w1p) + (long long)ws * W1_U4;
  const uint4* const W2g = static_cast<const uint4*>(a.w2p);
  auto ldu = [&](const uint4* base, int blk) -> uint4 {   // 16 bytes of this lane from the 1 KiB block blk behind the wave-uniform base
    return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base + blk * 64) + loff);
  };
  const int tile0 = blockIdx.x * TILE;
  // request order = arrival order: biases, frames, then the chunk's fragments
  const uint4 cb = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(W1g + W1F_U4) + (loff & 255u));
  bf16x8 xh[MT][KS], xl[MT][KS];
  float xv[MT][KS][8];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = tile0 + MT * fi + mt;
    const float* xp = a.x + (long long)(m < a.M ? m : 0) * F + 8 * fg;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float4 p = ld4(xp + 32 * ks), q = ld4(xp + 32 * ks + 4);
      xv[mt][ks][0] = p.x; xv[mt][ks][1] = p.y; xv[mt][ks][2] = p.z; xv[mt][ks][3] = p.w;
      xv[mt][ks][4] = q.x; xv[mt][ks][5] = q.y; xv[mt][ks][6] = q.z; xv[mt][ks][7] = q.w;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  uint4 wr[4][2 * KS];      // tile j of the chunk: [K step][plane]; later this wave's W3 fragments of K steps 2j, 2j+1: [step][tile][plane]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k2 = 0; k2 < 2 * KS; ++k2) wr[j][k2] = ldu(W1g, j * 2 * KS + k2);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 h, l;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xn = xv[mt][ks][e];
        const __bf16 hh = (__bf16)xn;
        h[e] = hh;
        l[e] = (__bf16)(xn - (float)hh);
      }
      xh[mt][ks] = h;
      xl[mt][ks] = l;
    }
  if (lane < 16) reinterpret_cast<uint4*>(&cst[w][0])[lane] = cb;

  // ---- phase 1: this wave's hidden chunk ---------------------------------------------------------------------------------------
  bf16x8 gh[2][MT], gw[2][MT];     // activated values (bf16 hi / lo) of the chunk's two K steps, k-slot order
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f32x4 h[MT];
    {
      const float4 bv = ld4(&cst[w][0] + j * 16 + 4 * fg);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) h[mt] = (f32x4){bv.x, bv.y, bv.z, bv.w};
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 wh = *reinterpret_cast<const bf16x8*>(&wr[j][2 * ks]);
      const bf16x8 wlo = *reinterpret_cast<const bf16x8*>(&wr[j][2 * ks + 1]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) h[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[mt][ks], h[mt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) h[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl[mt][ks], h[mt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) h[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, xh[mt][ks], h[mt], 0, 0, 0);
    }
    // (scheduling fences: hipcc otherwise sinks the loads down to their first use - and waits for each of them there)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int t = 0; t < FTW; ++t)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) wr[j][(s * FTW + t) * 2 + pl] = ldu(W2g + (long long)j * W2_U4, ((s * FT + FTW * ws + t) * 2 + pl));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float o = gelu_fast(h[mt][r]);
        const __bf16 hb = (__bf16)o;
        gh[j >> 1][mt][4 * (j & 1) + r] = hb;
        gw[j >> 1][mt][4 * (j & 1) + r] = (__bf16)(o - (float)hb);
      }
  }
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      *reinterpret_cast<bf16x8*>(&hs[(((2 * w + s) * 2 + 0) * MT + mt) * 64 + lane]) = gh[s][mt];
      *reinterpret_cast<bf16x8*>(&hs[(((2 * w + s) * 2 + 1) * MT + mt) * 64 + lane]) = gw[s][mt];
    }
  // the residual rows (this lane's 2 x 4 channels per frame) fly under the barrier and phase 2
  float4 xr[MT][FTW];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = tile0 + MT * fi + mt;
    const float* rp = a.res + (long long)(m < a.M ? m : 0) * F + 32 * w + 4 * fg;
#pragma unroll
    for (int t = 0; t < FTW; ++t) xr[mt][t] = ld4(rp + 16 * t);
  }
  __syncthreads();

  // ---- phase 2: output tiles 2w, 2w+1 over the 8 K steps, in order --------------------------------------------------------------
  f32x4 acc[FTW][MT];
#pragma unroll
  for (int t = 0; t < FTW; ++t)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[t][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  uint4 hb[2][2][MT];             // [ring][plane][frame tile]
#pragma unroll
  for (int pl = 0; pl < 2; ++pl)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) hb[0][pl][mt] = hs[((0 * 2 + pl) * MT + mt) * 64 + lane];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    if (q + 1 < NQ) {
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) hb[(q + 1) & 1][pl][mt] = hs[(((q + 1) * 2 + pl) * MT + mt) * 64 + lane];
    }
    bf16x8 ah[MT], aw[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      ah[mt] = *reinterpret_cast<const bf16x8*>(&hb[q & 1][0][mt]);
      aw[mt] = *reinterpret_cast<const bf16x8*>(&hb[q & 1][1][mt]);
    }
#pragma unroll
    for (int t = 0; t < FTW; ++t) {
      const bf16x8 wh = *reinterpret_cast<const bf16x8*>(&wr[q >> 1][((q & 1) * FTW + t) * 2]);
      const bf16x8 wlo = *reinterpret_cast<const bf16x8*>(&wr[q >> 1][((q & 1) * FTW + t) * 2 + 1]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ah[mt], acc[t][mt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, aw[mt], acc[t][mt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, ah[mt], acc[t][mt], 0, 0, 0);
    }
  }
  // ---- epilogue: y = res + ls * (acc + b3); fragment row 4 fg + r of tile 2w + t is channel 32 w + 16 t + 4 fg + r ----------------
  {
#pragma clang fp contract(off)
#pragma unroll
    for (int t = 0; t < FTW; ++t) {
      const int ch = 32 * w + 16 * t + 4 * fg;
      const float4 bb = ld4(a.b3 + ch), lsv = ld4(a.ls + ch);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = tile0 + MT * fi + mt;
        if (m < a.M) {
          const f32x4 o = acc[t][mt];
          st4(a.y + (long long)m * F + ch, make_float4(fmaf(o[0] + bb.x, lsv.x, xr[mt][t].x), fmaf(o[1] + bb.y, lsv.y, xr[mt][t].y),
                                                        fmaf(o[2] + bb.z, lsv.z, xr[mt][t].z), fmaf(o[3] + bb.w, lsv.w, xr[mt][t].w)));
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// cla_head_hs_kernel (round 6): the output-split form of the three head-type launches for at most one tile per CU (batch 1).  The four waves
// hold the SAME 16*MT frames (LayerNorm computed by each) and split the OUTPUT tiles; a weight fragment has one reader and goes global ->
// registers; no LDS, no barrier, direct stores.  Per accumulator the operand sequence of cla_head_kernel: bit-identical.
//   HS_GLU   CLA head:  wave w = chunk w (v0 v1 g0 g1 -> 32 gated channels)
//   HS_GATE  EGA gate with the folded output projection (a.o set): wave w = output tiles 2w, 2w+1 of BOTH projections - the attention's
//            LayerScale(linear_out(o)) of a frame's pooled row is computed in the lane that needs it (one N tile per frame tile, rows chosen
//            per lane) and never leaves the registers
//   HS_QKV   pooling + LayerNorm + q / k / v: wave w = output tiles 6w .. 6w+5 of the [rows, 3F] tensor
// ---------------------------------------------------------------------------------------------------------------------
constexpr int HS_GLU = 0, HS_GATE = 1, HS_QKV = 2;
template <int MODE, int MT>
__global__ __launch_bounds__(256, 1) void cla_head_hs_kernel(const ClaFusedArgs a) {
  constexpr int F = 128, KS = F / 32;
  constexpr int W1F_U4 = 4 * KS * 2 * 64, CS_U4 = 256, W1_U4 = W1F_U4 + CS_U4;
  constexpr int NTW = MODE == HS_GLU ? 4 : (MODE == HS_GATE ? 2 : 6);     // plain 16-row tiles of the projection per wave
  constexpr int TILE = 16 * MT;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fi = lane & 15, fg = lane >> 4;
  const int ws = __builtin_amdgcn_readfirstlane(w);
  const unsigned loff = (unsigned)lane * 16u;
  const uint4* const W1g = static_cast<const uint4*>(a.w1p);
  auto ldu = [&](const uint4* base, int blk) -> uint4 {   // 16 bytes of this lane from the 1 KiB block blk behind the wave-uniform base
    return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base + blk * 64) + loff);
  };
  const int tile0 = blockIdx.x * TILE;
  // this wave's tile i: global plain-tile index gt = NTW * w + i -> chunk gt / 4, tile gt % 4 of the packed weights
  float4 bias[NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    const int gt = NTW * ws + i;
    bias[i] = ld4(reinterpret_cast<const float*>(W1g + (long long)(gt >> 2) * W1_U4 + W1F_U4) + (gt & 3) * 16 + 4 * fg);
  }
  // all of this wave's fragments are requested before the frames (they land under the frame loads and the LayerNorm; hipcc would otherwise
  // sink each load to its first use and wait for it there: 23 us instead of 13 for the q / k / v form)
  uint4 wf[NTW][KS][2];
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    const int gt = NTW * ws + i;
    const uint4* const base = W1g + (long long)(gt >> 2) * W1_U4 + (gt & 3) * KS * 2 * 64;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      wf[i][ks][0] = ldu(base, ks * 2);
      wf[i][ks][1] = ldu(base, ks * 2 + 1);
    }
  }
  // gate: the folded output projection's fragments and the pooled attention rows of this lane's frames, requested up front as well
  constexpr int NTO = MODE == HS_GATE ? NTW : 1, MTO = MODE == HS_GATE ? MT : 1;
  uint4 wo[NTO][KS][2];
  float4 orow[MTO][KS][2];
  if constexpr (MODE == HS_GATE) {
    const uint4* const Wob = static_cast<const uint4*>(a.wop) + (long long)(NTW * ws) * KS * 2 * 64;
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) wo[t][ks][pl] = ldu(Wob, (t * KS + ks) * 2 + pl);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = tile0 + MT * fi + mt;
      const int mc = m < a.M ? m : 0;
      const int seq = mc / a.T, t_ = mc - seq * a.T;
      const float* op = a.o + ((long long)seq * a.Tp + t_ / a.fac) * F + 8 * fg;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        orow[mt][ks][0] = ld4(op + 32 * ks);
        orow[mt][ks][1] = ld4(op + 32 * ks + 4);
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  // ---- frames (every wave holds all of them; lane fi holds frames MT*fi .. MT*fi + MT-1) ----
  bf16x8 xh[MT][KS], xl[MT][KS];
  if constexpr (MODE == HS_QKV) load_frames_pooled<F, MT>(a.x, tile0, a.M, a.pool, a.eps, fi, fg, xh, xl);
  else load_frames<F, true, MT>(a.x, tile0, a.M, a.eps, fi, fg, xh, xl);
  f32x4 acc[NTW][MT];
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[i][mt] = (f32x4){bias[i].x, bias[i].y, bias[i].z, bias[i].w};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 wh = *reinterpret_cast<const bf16x8*>(&wf[i][ks][0]), wlo = *reinterpret_cast<const bf16x8*>(&wf[i][ks][1]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[mt][ks], acc[i][mt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl[mt][ks], acc[i][mt], 0, 0, 0);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, xh[mt][ks], acc[i][mt], 0, 0, 0);
    }
  }
  if constexpr (MODE == HS_GLU) {
    // tiles 0 1 = value, 2 3 = gate of chunk w: gated channels 32 w + 16 j + 4 fg + r
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = tile0 + MT * fi + mt;
      if (m < a.M) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f32x4 hv = acc[j][mt], hg = acc[2 + j][mt];
          st4(a.y + (long long)m * F + 32 * w + 16 * j + 4 * fg,
              make_float4(hv[0] * sigmoid_f(hg[0]), hv[1] * sigmoid_f(hg[1]), hv[2] * sigmoid_f(hg[2]), hv[3] * sigmoid_f(hg[3])));
        }
      }
    }
  } else if constexpr (MODE == HS_QKV) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = tile0 + MT * fi + mt;
      if (m < a.M) {
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
          const f32x4 v = acc[i][mt];
          st4(a.y + (long long)m * a.ldy + 16 * (NTW * w + i) + 4 * fg, make_float4(v[0], v[1], v[2], v[3]));
        }
      }
    }
  } else {
    // ---- folded output projection for the pooled rows of this lane's frames, output tiles 2w, 2w+1 ----
    float4 bov[NTW], lov[NTW];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      bov[t] = ld4(a.bo + 16 * (NTW * w + t) + 4 * fg);
      lov[t] = ld4(a.lso + 16 * (NTW * w + t) + 4 * fg);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = tile0 + MT * fi + mt;
      bf16x8 oh[KS], ol[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float4 p = orow[mt][ks][0], q = orow[mt][ks][1];
        const float v[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const __bf16 hh = (__bf16)v[e];
          oh[ks][e] = hh;
          ol[ks][e] = (__bf16)(v[e] - (float)hh);
        }
      }
      f32x4 pa[NTW];
#pragma unroll
      for (int t = 0; t < NTW; ++t) pa[t] = (f32x4){bov[t].x, bov[t].y, bov[t].z, bov[t].w};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          const bf16x8 wh = *reinterpret_cast<const bf16x8*>(&wo[t][ks][0]), wlo = *reinterpret_cast<const bf16x8*>(&wo[t][ks][1]);
          pa[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, oh[ks], pa[t], 0, 0, 0);
          pa[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, ol[ks], pa[t], 0, 0, 0);
          pa[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, oh[ks], pa[t], 0, 0, 0);
        }
      if (m < a.M) {
#pragma clang fp contract(off)
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
          const int ch = 16 * (NTW * w + t) + 4 * fg;
          const float4 xr = ld4(a.res + (long long)m * F + ch);
          const float4 ar = make_float4(pa[t][0] * lov[t].x, pa[t][1] * lov[t].y, pa[t][2] * lov[t].z, pa[t][3] * lov[t].w);
          const f32x4 g = acc[t][mt];
          st4(a.y + (long long)m * F + ch, make_float4(fmaf(sigmoid_f(g[0]), ar.x, xr.x), fmaf(sigmoid_f(g[1]), ar.y, xr.y),
                                                       fmaf(sigmoid_f(g[2]), ar.z, xr.z), fmaf(sigmoid_f(g[3]), ar.w, xr.w)));
        }
      }
    }
  }
}

// frame tiles per wave (2 / 4: 32- / 64-frame workgroup tiles) of the output-split head forms for a launch of M rows, 0 = not taken
// (SEPR_CF_HEAD_HS=0 switches them off)
static int head_hs_tiles(int M, int cus) {
  static const bool on = [] {
    const char* e = getenv("SEPR_CF_HEAD_HS");
    return !(e && e[0] == '0');
  }();
  if (!on) return 0;
  if ((M + 31) / 32 <= cus) return 2;
  if ((M + 63) / 64 <= cus) return 4;
  return 0;
}

// small launches (round 6): when even the 64-frame tiles of the MT = 1 instantiations leave workgroup slots free (cap = two per CU), a launch's
// duration is one workgroup's dependent chain - halve it.  SEPR_CF_SMALL=<max 64-frame tiles> overrides the bound (0: never; A/B).
static bool small_launch(int M, int cap) {
  static const int bound = [] {
    const char* e = getenv("SEPR_CF_SMALL");
    return e ? atoi(e) : -1;
  }();
  return (M + 63) / 64 <= (bound < 0 ? cap : bound);
}

int launch_cla_head(const ClaFusedArgs& a, int F, int site, hipStream_t stream) {
  if (a.M <= 0) return SEPR_OK;
  if (!a.x || !a.y || !a.w1p || a.x == a.y || (F != 128 && F != 256)) return SEPR_EINVAL;
  long long slot = -1;
  const bool timed = prof_begin(site, stream, &slot);
  const int ntiles = (a.M + 127) / 128;
  const int cap = persistent_grid();
  const int hs = F == 128 ? head_hs_tiles(a.M, cap / 2) : 0;
  if (hs == 2) hipLaunchKernelGGL((cla_head_hs_kernel<HS_GLU, 2>), dim3((a.M + 31) / 32), dim3(256), 0, stream, a);
  else if (hs == 4) hipLaunchKernelGGL((cla_head_hs_kernel<HS_GLU, 4>), dim3((a.M + 63) / 64), dim3(256), 0, stream, a);
  else
  if (F == 256) hipLaunchKernelGGL((cla_head_kernel<256, false>), dim3(ntiles < cap / 2 ? ntiles : cap / 2), dim3(CF_NT), 0, stream, a);   // one workgroup per CU
  else if (small_launch(a.M, cap)) hipLaunchKernelGGL((cla_head_kernel<128, false, 1>), dim3((a.M + 63) / 64), dim3(CF_NT), 0, stream, a);
  else hipLaunchKernelGGL((cla_head_kernel<128, false>), dim3(ntiles < cap ? ntiles : cap), dim3(CF_NT), 0, stream, a);
  if (timed) prof_end(slot, (double)a.M * 2.0 * F * 2 * F, stream);
  SEPR_CHECK_LAUNCH("cla_head_kernel");
  return SEPR_OK;
}

int launch_ega_gate(const ClaFusedArgs& a, int F, int site, hipStream_t stream) {
  if (a.M <= 0) return SEPR_OK;
  if (!a.x || !a.res || !a.att || !a.y || !a.w1p || a.x == a.y || (F != 128 && F != 256) || a.T <= 0 || a.Tp <= 0 || a.fac <= 0)
    return SEPR_EINVAL;
  if (a.o && (!a.att_w || a.att_w != a.att || !a.wop || !a.bo || !a.lso || a.Mp <= 0 || F != 128 || 64 % a.fac != 0)) return SEPR_EINVAL;
  long long slot = -1;
  const bool timed = prof_begin(site, stream, &slot);
  const int ntiles = (a.M + 127) / 128;
  const int cap = persistent_grid();
  const int hs = (F == 128 && a.o) ? head_hs_tiles(a.M, cap / 2) : 0;
  if (hs == 2) hipLaunchKernelGGL((cla_head_hs_kernel<HS_GATE, 2>), dim3((a.M + 31) / 32), dim3(256), 0, stream, a);
  else if (hs == 4) hipLaunchKernelGGL((cla_head_hs_kernel<HS_GATE, 4>), dim3((a.M + 63) / 64), dim3(256), 0, stream, a);
  else
  if (F == 256) hipLaunchKernelGGL((cla_head_kernel<256, true>), dim3(ntiles < cap / 2 ? ntiles : cap / 2), dim3(CF_NT), 0, stream, a);
  else if (small_launch(a.M, cap)) hipLaunchKernelGGL((cla_head_kernel<128, true, 1>), dim3((a.M + 63) / 64), dim3(CF_NT), 0, stream, a);
  else hipLaunchKernelGGL((cla_head_kernel<128, true>), dim3(ntiles < cap ? ntiles : cap), dim3(CF_NT), 0, stream, a);
  if (timed) prof_end(slot, (double)a.M * 2.0 * F * F, stream);
  SEPR_CHECK_LAUNCH("ega_gate_kernel");
  return SEPR_OK;
}

// pooling + LayerNorm + q / k / v projection of the EGA attention: a.x = block input [M * pool, F], a.y = qkv [M, 3F] (a.ldy = 3F), a.M = pooled rows
int launch_ega_qkv(const ClaFusedArgs& a, int F, int site, hipStream_t stream) {
  if (a.M <= 0) return SEPR_OK;
  if (!a.x || !a.y || !a.w1p || a.x == a.y || F != 128 || a.pool <= 0 || a.ldy < 3 * F) return SEPR_EINVAL;
  long long slot = -1;
  const bool timed = prof_begin(site, stream, &slot);
  const int ntiles = (a.M + 127) / 128;
  const int cap = persistent_grid();
  // (the output-split form of this launch - cla_head_hs_kernel<HS_QKV, .>, SEPR_CF_QKV_HS=1 - measured 19.1 us against 13.4 us for the three
  //  workgroups per tile below: 32 workgroups at 1000 pooled rows, each wave repeating the pooled loads; profiles/r06_gcfn_hidden_split.txt (8))
  static const bool qkv_hs = [] {
    const char* e = getenv("SEPR_CF_QKV_HS");
    return e && e[0] == '1';
  }();
  const int hs = qkv_hs ? head_hs_tiles(a.M, cap / 2) : 0;
  if (hs == 2) hipLaunchKernelGGL((cla_head_hs_kernel<HS_QKV, 2>), dim3((a.M + 31) / 32), dim3(256), 0, stream, a);
  else if (hs == 4) hipLaunchKernelGGL((cla_head_hs_kernel<HS_QKV, 4>), dim3((a.M + 63) / 64), dim3(256), 0, stream, a);
  else
  if (small_launch(3 * a.M, cap)) hipLaunchKernelGGL((cla_head_kernel<128, true, 1, true>), dim3((a.M + 63) / 64, 3), dim3(CF_NT), 0, stream, a);
  else if (small_launch(a.M, cap)) hipLaunchKernelGGL((cla_head_kernel<128, true, 1, true>), dim3((a.M + 63) / 64), dim3(CF_NT), 0, stream, a);
  else hipLaunchKernelGGL((cla_head_kernel<128, true, CF_MT, true>), dim3(ntiles < cap ? ntiles : cap), dim3(CF_NT), 0, stream, a);
  if (timed) prof_end(slot, (double)a.M * 2.0 * F * 3 * F, stream);
  SEPR_CHECK_LAUNCH("ega_qkv_kernel");
  return SEPR_OK;
}

int launch_cla_tail(const ClaFusedArgs& a, int F, int site, hipStream_t stream) {
  if (a.M <= 0) return SEPR_OK;
  if (!a.x || !a.res || !a.y || !a.w1p || !a.w2p || !a.b3 || !a.ls || a.x == a.y || (F != 128 && F != 256)) return SEPR_EINVAL;
  long long slot = -1;
  const bool timed = prof_begin(site, stream, &slot);
  const int ntiles = (a.M + 127) / 128;
  const int cap = persistent_grid();
  // at most one tile per CU: the hidden-split form, 32- or 64-frame tiles (SEPR_CF_HS=0 switches it off, 2 / 4 force a tile size)
  static const int hs_force = [] {
    const char* e = getenv("SEPR_CF_HS");
    return e && e[0] ? atoi(e) : -1;
  }();
  const int cus = cap / 2;
  const int hs_mt = hs_force == 0 ? 0 : (hs_force == 4 ? ((a.M + 63) / 64 <= cus ? 4 : 0) : ((a.M + 31) / 32 <= cus ? 2 : (hs_force < 0 && (a.M + 63) / 64 <= cus ? 4 : 0)));
  if (F == 256) hipLaunchKernelGGL((cla_tail_kernel<256>), dim3(ntiles < cap / 2 ? ntiles : cap / 2), dim3(CF_NT), 0, stream, a);
  else if (hs_mt == 2) hipLaunchKernelGGL((cla_tail_hs_kernel<2>), dim3((a.M + 31) / 32), dim3(256), 0, stream, a);
  else if (hs_mt == 4) hipLaunchKernelGGL((cla_tail_hs_kernel<4>), dim3((a.M + 63) / 64), dim3(256), 0, stream, a);
  else if (small_launch(a.M, cap)) hipLaunchKernelGGL((cla_tail_kernel<128, 1>), dim3((a.M + 63) / 64), dim3(CF_NT), 0, stream, a);
  else hipLaunchKernelGGL((cla_tail_kernel<128>), dim3(ntiles < cap ? ntiles : cap), dim3(CF_NT), 0, stream, a);
  if (timed) prof_end(slot, (double)a.M * (2.0 * F * 2 * F + 2.0 * 2 * F * F), stream);
  SEPR_CHECK_LAUNCH("cla_tail_kernel");
  return SEPR_OK;
}

}  // namespace sepr
