// Fully fused GCFN block on the bf16x3 core (reference modules/network.py:46-66):
//     y = x + layer_scale * Linear_3F->F( GLU( dwconv_k3( Linear_F->6F( LayerNorm(x) ) ) ) )
// in ONE kernel.  Neither the [rows, 6F] hidden tensor nor the [rows, 3F] gated tensor nor the LayerNorm
// statistics ever reach HBM: algorithmic traffic is "read x once, write y once" (1 KB per frame at F=128
// against ~4 KB for the two-projection form).
//
// Row-stationary design.  A wave owns 32 consecutive frames (30 outputs + one halo frame on each side, the
// halo is recomputed so waves never exchange activations):
//   * its frames are loaded once, straight from global memory into the MFMA B-fragment layout (lane = frame,
//     4 lane groups x 8 consecutive channels per K step); LayerNorm statistics are two shuffles across the 4
//     lane groups; the normalised frames are split into bf16 hi/lo and stay in registers for the whole tile;
//   * the hidden dimension is walked in chunks of 32 value + 32 gate channels.  Per chunk the packed weight
//     fragments (48 KB for F=128: up-projection tiles + the matching K slice of the down-projection) are
//     copied global -> LDS once per workgroup in fragment order (conflict-free 16-byte reads);
//   * up-projection: A = weight fragment (rows = hidden channel), B = frame fragment, so a lane holds 4
//     consecutive hidden channels of ONE frame.  The depthwise k=3 convolution runs along frames = along the
//     16 lanes of a DPP row (row_ror:1 / row_ror:15 + a select at the two tile seams), GLU in registers;
//   * the 8 gated values a lane then holds (4 from each 16-channel tile of the chunk) ARE the B fragment of
//     the down-projection's K step: the down-projection weights are packed with the matching k-slot order,
//     so the gated tensor is never transposed, staged or stored;
//   * down-projection accumulates [F channels] x [32 frames] in 64 accumulator registers across the chunks;
//   * epilogue: bias, LayerScale, residual; staged through LDS two waves at a time for row-contiguous stores.
// Two 4-wave workgroups share a CU (48 KB LDS each): one copies its next weight chunk while the other
// multiplies.
//
// The same kernel has a plain mode (template parameter MODE = 1): Linear -> GLU -> Linear without LayerNorm, conv,
// LayerScale and residual, with a runtime hidden width and input / output row maps - SpkSplitStage's two 1x1 convolutions
// (modules/module.py:114-116,123) and OutputLayer's two projections (:250-256), one launch per 128 output columns.
#include "sepr_gcfn_fused.h"
#include "sepr_train.h"
#include <stdlib.h>

namespace sepr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// frames written per wave = 16*MT - 2 (one recomputed halo frame on each side), per workgroup = NW times that


__device__ __forceinline__ float dpp_ror1(float v) {   // lane i <- lane (i-1) mod 16 of its 16-lane row
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x121, 0xf, 0xf, true));   // (every lane has a source: no `old`)
}
// shifts instead of rotations: the lane with no source in its 16-lane row (0 for shr, 15 for shl) keeps `old`
__device__ __forceinline__ float dpp_shr1(float old, float v) {   // lane i <- lane i-1; lane 0 <- old
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_shl1(float old, float v) {   // lane i <- lane i+1; lane 15 <- old
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_rol1(float v) {   // lane i <- lane (i+1) mod 16
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x12f, 0xf, 0xf, true));
}

// ---------------------------------------------------------------------------------------------------------
// gcfn_fused3_kernel (the default).  Its predecessor (git history: gcfn_fused_kernel, frame = 16*mt + fi) spent 4.7
// VALU instructions per MFMA and waited on every LDS fragment read; what changed:
//  * frames are INTERLEAVED over the two frame tiles (frame = 2*fi + mt instead of 16*mt + fi): the previous /
//    next frame of a lane's tile-0 / tile-1 value is the lane's own other register, the remaining neighbour is one
//    DPP row rotation, and the two rotation wrap-arounds land exactly on the two halo frames whose outputs are
//    discarded - no seam selects (12 DPP + 4 selects per channel before, 4 DPP now);
//  * the up-projection bias is the MFMA accumulator's initial value instead of an add per element;
//  * the zero-padding flags of the depthwise conv (4 multiplies per element) only exist in the instantiation run
//    by waves whose 32 frames touch a sequence boundary (1 wave in ~260 at T=8000);
//  * weight fragments are read from LDS two MFMA groups ahead (explicit ring + scheduling barriers), the first
//    groups of the next phase are requested before the conv's VALU work.
// ---------------------------------------------------------------------------------------------------------
#ifndef SEPR_GF_ABL
#define SEPR_GF_ABL 0   // timing ablations (wrong results): 1 no chunk loop, 2 weight chunks copied once per tile,
                        // 4 no chunk barriers, 8 one LDS fragment read per chunk, 16 no exp/rcp in the GLU,
                        // 32 / 64: PROXY of "fold the depthwise conv into the up-projection" (round-3 review item 6: three
                        //   K = F MFMA passes against tap-scaled weights, the conv becomes two neighbour adds): the up-projection's
                        //   MFMAs, its LDS fragment reads and the W1 chunk copy run THREE times, the conv keeps its neighbour
                        //   exchange but uses adds instead of tap FMAs (32), or drops the exchange as well (64: the most
                        //   optimistic bound - no DPP at all)
#endif
template <bool V>
struct bool_c { static constexpr bool value = V; };

#ifndef SEPR_GF3_RING
#define SEPR_GF3_RING 2
#endif
#ifndef SEPR_GF3_REDERIVE
#define SEPR_GF3_REDERIVE 1   // bit 1: the thread index made opaque at the top of every tile, bit 2: again in front of the epilogue - what is derived from it is then
                              // recomputed there instead of being computed once, hoisted and kept alive (or spilled) across the chunk loop
#endif
#ifndef SEPR_GF3_UPFIRST
#define SEPR_GF3_UPFIRST 1   // 1: both up-projections before both convolutions (the next chunk copy gets one more conv to land; ~1 %)
#endif
#ifndef SEPR_GF3_PRIO
#define SEPR_GF3_PRIO 0
#endif
#ifndef SEPR_GF3_XCH
#define SEPR_GF3_XCH 1
#endif
#ifndef SEPR_GF3_WGPRIO
#define SEPR_GF3_WGPRIO 0
#endif
#ifndef SEPR_GF3_ASMDMA
#define SEPR_GF3_ASMDMA 1   // 1: the weight-chunk copies are inline-asm LDS-DMA (invisible to hipcc's waitcnt pass, which otherwise puts an
                            // s_waitcnt vmcnt(0) in front of the first LDS read behind a copy it knows about - tools/isa_trace.py shows one in
                            // the middle of the up-projection of the 4-wave kernel); the copies are then waited for at the two chunk barriers only
#endif
#ifndef SEPR_GF3_FENCE256
#define SEPR_GF3_FENCE256 1   // 0 (A/B): no scheduling fences around the MFMA groups of the F = 256 (one-wave-per-SIMD) instantiations
#endif
#define SEPR_GF3_SCHED_FENCE() do { if constexpr (F <= 128 || SEPR_GF3_FENCE256) __builtin_amdgcn_sched_barrier(0); } while (0)
#ifndef SEPR_GF3_RESX
#define SEPR_GF3_RESX 0   // EXPERIMENT (round-3 review item 6): 1 = the residual x is rebuilt from the bf16 hi + lo planes the wave
                          // already holds ((hi + lo) / rstd + mean, 2^-17 relative) instead of being re-read from HBM in the
                          // epilogue: removes 512 of the 1.7 KB per frame the kernel moves; costs end-to-end agreement (measured:
                          // profiles/r03_v2_gcfn_resx_experiment.txt).  Off in the product build.
#endif
// ONE (training precision "bf16" only): plain bf16 operands - the normalised frames, the weights and the gated tensor are used
// as their bf16 hi plane alone, ONE MFMA per product instead of three, no lo-plane split on the VALU, and only the hi-plane
// blocks of every packed weight chunk are copied to LDS (half the L2 -> LDS stream).  Same packed weights, same LDS layout.
template <int F, int MT, int NW, int MODE = 0, bool TRAIN = false, bool ONE = false, int LAT = 0>
__global__ __launch_bounds__(64 * NW, (LAT > 0 || F > 128 || MT > 2) ? (NW > 4 ? 2 : 1) : (2 * NW) / 4) void gcfn_fused3_kernel(const GcfnFusedArgs a) {
  constexpr bool PLAIN = MODE >= 1;   // frames are independent: no halo, no seam exchange, no conv
  constexpr bool FOLD = MODE == 2;    // OutputLayer with the AudioDecoder folded into its second projection (see launch_glumlp_fold)
  static_assert(!(TRAIN && PLAIN), "the train instantiation is the GCFN block");
  static_assert(!ONE || TRAIN, "plain bf16 operands do not pass the inference parity gate: a training arithmetic only");
  const bool drop = TRAIN && a.drop_thr > 0u;
  DropKey dk0 = {0u, 0u}, dk1 = {0u, 0u};
  if (drop) {
    dk0 = sepr_drop_key(a.seed, a.salt, 0u);
    dk1 = sepr_drop_key(a.seed, a.salt, 1u);
  }
  static_assert(MT == 1 || MT == 2 || MT == 4, "frame tiles per wave");
  constexpr int RD = SEPR_GF3_RING;      // LDS fragment read-ahead, in MFMA groups
  // (Round 6, measured no: with MT = 1 the three MFMAs of a group run back to back on ONE accumulator; issuing the value and the gate tile of a
  //  K step - and two output tiles of the down-projection - interleaved, from a 6-slot fragment ring, changed nothing: 24.8 vs 23.5 us per
  //  batch-1 launch, profiles/r06_b1_latency.txt.  Back-to-back accumulation into one tile is forwarded at the issue rate.)
  constexpr int RDX = RD, FBN = RD + 1;
  constexpr bool UF = SEPR_GF3_UPFIRST != 0;
  constexpr int NT = 64 * NW;
  // XCH: the waves of a workgroup cover 16*MT*NW CONTIGUOUS frames and hand each other the conv's neighbour frame at
  // the wave seams through LDS (published by the barrier the chunk already has after the up-projections), so only the
  // two frames at the workgroup's ends are recomputed halo: 126 outputs per 128 frames instead of 120, and - what
  // matters more - 64000 x 2^k rows are then just under 512 x 2^k tiles, i.e. full launch rounds instead of
  // "one round + a 4 % tail" (534 tiles on 512 slots).  Without XCH every wave carries its own two halo frames.
  constexpr bool XCH = (SEPR_GF3_XCH != 0) && MT >= 2 && UF && !PLAIN && LAT == 0;   // (ring form: ONE barrier per chunk - with a second, LDS-only
                                                                                    //  barrier for the seam exchange it measured 2 % slower: r06_gcfn_ring.txt)
  constexpr int HALO = PLAIN ? 0 : 1;
  constexpr int WSTR = (XCH || PLAIN) ? 16 * MT : 16 * MT - 2;            // frames a wave advances
  constexpr int FOLD_HB = 3;                                       // (K - 1) / stride of the folded ConvTranspose1d(k = 16, stride 4)
  constexpr int GF_TILE = FOLD ? NW * 16 * MT - FOLD_HB : (XCH ? NW * 16 * MT - 2 : NW * WSTR);   // output frames per workgroup tile
  constexpr int EH = (16 * MT * NW + 63) / 64;   // epilogue passes of up to 64 frames
  constexpr int KS = F / 32;
  const int NCH = PLAIN ? a.nch : 3 * F / 32;
  constexpr int FT = FOLD ? 1 : F / 16;   // 16-row output tiles of the down-projection (FOLD: the 16 decoder taps)
  constexpr int W1F_U4 = 4 * KS * 2 * 64;
  constexpr int CS_U4 = 256;
  constexpr int W1_U4 = W1F_U4 + CS_U4;
  constexpr int W2_U4 = FT * 2 * 64;
  constexpr int OS = F + 4;
  // LAT > 0 (latency form, launches with fewer tiles than CUs): a ring of LAT whole-chunk images [W1 fragments | constants | W2 fragments]
  // (52 KB each at F = 128; one workgroup per CU, so the LDS the second workgroup would use is free), chunk c + LAT - 1 requested
  // while chunk c multiplies, ONE raw barrier per chunk and counted vmcnt waits.  The copies are inline asm: hipcc (ROCm 7.2) drains
  // vmcnt to 0 in front of the first LDS read behind any LDS-DMA it knows about, which is what serialised copy and multiply before.
  constexpr int NST = LAT > 0 ? LAT : 1;
  constexpr int STG_U4 = W1_U4 + W2_U4;
  static_assert(LAT == 0 || (!TRAIN && !FOLD && F == 128), "ring form: inference instantiations");
  __shared__ __attribute__((aligned(16))) uint4 wl[LAT > 0 ? NST * STG_U4 : W1F_U4 + W2_U4 + 2 * CS_U4];
  __shared__ __attribute__((aligned(16))) float xch[XCH ? NW * 2 * 2 * 2 * 16 : 4];   // [wave][first|last frame][j][v|g][16 ch]
  static_assert(FOLD || sizeof(uint4) * (W1F_U4 + W2_U4) >= sizeof(float) * 64 * OS, "epilogue staging must fit");
  static_assert(W1F_U4 / 64 <= 16 * NW, "copy partition");
  const uint4* w1s = wl;                      // (LAT: re-pointed at the chunk's ring stage)
  const uint4* w2s = wl + (LAT > 0 ? W1_U4 : W1F_U4);
  uint4* csl = wl + (LAT > 0 ? W1F_U4 : W1F_U4 + W2_U4);

#if SEPR_GF3_REDERIVE
  int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;       // re-derived from an opaque copy at the top of every tile and in front of the epilogue (SEPR_GF3_REDERIVE)
  int fi = lane & 15, fg = lane >> 4;
#else
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fi = lane & 15, fg = lane >> 4;
#endif
  const int ntiles = FOLD ? a.fold_nseq * a.fold_tps : (a.M + GF_TILE - 1) / GF_TILE;
  const uint4* const W1g = static_cast<const uint4*>(a.w1p);
  const uint4* const W2g = static_cast<const uint4*>(a.w2p);

  // ---- weight chunks: global -> LDS by LDS-DMA, 1 KiB per wave instruction ---------------------------
#if SEPR_GF3_REDERIVE
  [[maybe_unused]] int ws = __builtin_amdgcn_readfirstlane(w);         // the wave index as a scalar
#else
  [[maybe_unused]] const int ws = __builtin_amdgcn_readfirstlane(w);   // the wave index as a scalar
#endif
  auto dma = [&](const uint4* gbase, uint4* lbase, int nblk) {   // nblk 1 KiB blocks, dealt round-robin to the waves
    unsigned loff = (unsigned)lane * 16u;
    asm volatile("" : "+v"(loff));
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (i * NW >= nblk) break;
      const int blk = i * NW + w;
      if (blk < nblk) {
        const char* src = reinterpret_cast<const char*>(gbase + blk * 64) + loff;
#if SEPR_GF3_ASMDMA
        (void)src;
        glds16_asm(gbase + (i * NW + ws) * 64, loff, __builtin_amdgcn_readfirstlane(lds_addr(lbase + (i * NW + ws) * 64)));
#else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lbase + blk * 64), 16, 0, 0);
#endif
      }
    }
  };
  auto dma_hi = [&](const uint4* gbase, uint4* lbase, int nblk) {   // ONE: the even (bf16 hi plane) 1 KiB blocks only
    unsigned loff = (unsigned)lane * 16u;
    asm volatile("" : "+v"(loff));
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (2 * i * NW >= nblk) break;
      const int blk = 2 * (i * NW + w);
      if (blk < nblk) {
        const char* src = reinterpret_cast<const char*>(gbase + blk * 64) + loff;
#if SEPR_GF3_ASMDMA
        (void)src;
        glds16_asm(gbase + 2 * (i * NW + ws) * 64, loff, __builtin_amdgcn_readfirstlane(lds_addr(lbase + 2 * (i * NW + ws) * 64)));
#else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lbase + blk * 64), 16, 0, 0);
#endif
      }
    }
  };
  auto dma_w1 = [&](int c) {
    if constexpr (ONE) dma_hi(W1g + (long long)c * W1_U4, wl, W1F_U4 / 64);
    else dma(W1g + (long long)c * W1_U4, wl, W1F_U4 / 64);
    if (SEPR_GF_ABL & 96) {   // fold proxy: three tap-scaled weight sets = three times the up-projection copy volume
      dma(W1g + (long long)c * W1_U4, wl, W1F_U4 / 64);
      dma(W1g + (long long)c * W1_U4, wl, W1F_U4 / 64);
    }
    dma(W1g + (long long)c * W1_U4 + W1F_U4, csl + (c & 1) * CS_U4, CS_U4 / 64);
  };
  auto dma_w2 = [&](int c) {
    if constexpr (ONE) dma_hi(W2g + (long long)c * W2_U4, wl + W1F_U4, W2_U4 / 64);
    else dma(W2g + (long long)c * W2_U4, wl + W1F_U4, W2_U4 / 64);
  };
  auto dma_barrier = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!(SEPR_GF_ABL & 4)) __syncthreads();
  };
  // LAT: one chunk image into ring stage c % NST.  Every wave issues exactly LAT_NI copies (block index wraps, so a few blocks are
  // copied twice with the same bytes): the counted waits below are compile-time constants.
  constexpr int LAT_N1 = (W1_U4 / 64 + NW - 1) / NW, LAT_N2 = (W2_U4 / 64 + NW - 1) / NW, LAT_NI = LAT_N1 + LAT_N2;
  [[maybe_unused]] const unsigned wl_lds = lds_addr(wl);
  [[maybe_unused]] auto lat_dma = [&](int c) {
    const unsigned stage = wl_lds + (unsigned)((c % NST) * STG_U4 * 16);
    const uint4* g1 = W1g + (long long)c * W1_U4;
    const uint4* g2 = W2g + (long long)c * W2_U4;
    unsigned loff = (unsigned)lane * 16u;
    asm volatile("" : "+v"(loff));
#pragma unroll
    for (int i = 0; i < LAT_N1; ++i) {
      const int blk = (i * NW + ws) % (W1_U4 / 64);
      glds16_asm(g1 + blk * 64, loff, __builtin_amdgcn_readfirstlane(stage + (unsigned)blk * 1024u));
    }
#pragma unroll
    for (int i = 0; i < LAT_N2; ++i) {
      const int blk = (i * NW + ws) % (W2_U4 / 64);
      glds16_asm(g2 + blk * 64, loff, __builtin_amdgcn_readfirstlane(stage + (unsigned)(W1_U4 * 16) + (unsigned)blk * 1024u));
    }
  };
  // LAT, top of chunk c: this wave's share of chunk c has landed (later chunks may stay in flight), then the barrier that (a) publishes
  // every wave's share and (b) says every wave is done reading chunk c - 1, whose stage the next request overwrites
  [[maybe_unused]] auto lat_enter = [&](int c) {
    const int ahead = NCH - 1 - c < NST - 2 ? NCH - 1 - c : NST - 2;
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LAT_NI) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LAT_NI) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (c + NST - 1 < NCH && !(SEPR_GF_ABL & 2)) lat_dma(c + NST - 1);
    w1s = wl + (c % NST) * STG_U4;
    csl = wl + (c % NST) * STG_U4 + W1F_U4;
    w2s = wl + (c % NST) * STG_U4 + W1_U4;
  };
  static_assert(LAT <= 4 && 2 * LAT_NI < 64, "vmcnt immediates");
  // fragment pair (bf16 hi plane, lo plane) of one 16-channel tile at one K step
  auto ld_up = [&](int j, int g, uint4 (&d)[2]) {   // g = 2*ks + (0 value tile | 1 gate tile)
    const uint4* p = w1s + ((((g & 1) * 2 + j) * KS + (g >> 1)) * 2) * 64 + lane;
    if ((SEPR_GF_ABL & 8) && (j | g)) return;   // ablation: one fragment read per chunk
    d[0] = p[0];
    if constexpr (!ONE) d[1] = p[64];
  };
  auto ld_dn = [&](int ft, uint4 (&d)[2]) {
    const uint4* p = w2s + (ft * 2) * 64 + lane;
    if ((SEPR_GF_ABL & 8) && ft) return;
    d[0] = p[0];
    if constexpr (!ONE) d[1] = p[64];
  };

  // De-phase the two workgroups that share a CU: dispatched together they run their MFMA bursts (up- / down-projection)
  // and their VALU phases (depthwise conv + GLU + bf16 split) in lockstep, so the matrix pipe idles while both are in VALU
  // code and is contended while both multiply.  Every other co-resident workgroup (same parity rule as the projection
  // core, sepr_gemm.h) starts a fraction of a chunk period late.
#if SEPR_GF3_WGPRIO
  // EXPERIMENT (round 5): static wave priority for ONE of the two workgroups that share a CU (same parity rule as the stagger below), no per-phase
  // flips - MI355X_MICROARCH.md "two waves per SIMD", item 4: the second-dispatched wave of a SIMD is the arbitration loser on every segment
  {
    const int qp = blockIdx.x >> 3;
    if (((qp & 1) ^ ((qp >> 5) & 1)) != 0) __builtin_amdgcn_s_setprio(SEPR_GF3_WGPRIO);
  }
#endif
  if (a.stagger > 0) {
    const int ql = blockIdx.x >> 3;
    if (((ql & 1) ^ ((ql >> 5) & 1)) != 0)
      for (int i = 0; i < a.stagger; i += 100) __builtin_amdgcn_s_sleep(100);
  }
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
#if SEPR_GF3_REDERIVE
    if constexpr ((SEPR_GF3_REDERIVE & 1) != 0) {
      asm volatile("" : "+v"(tid));
      lane = tid & 63; w = tid >> 6; fi = lane & 15; fg = lane >> 4; ws = __builtin_amdgcn_readfirstlane(w);
    }
#endif
    // chunk 0 of the weights is requested first: it lands under the frame loads and the LayerNorm below
    __syncthreads();   // the previous tile's epilogue staging is fully consumed
    if constexpr (LAT > 0) {
#pragma unroll
      for (int c0 = 0; c0 < NST - 1; ++c0)
        if (c0 < NCH) lat_dma(c0);
    } else {
      dma_w1(0);
      dma_w2(0);
    }
    // ---- this wave's 32 frames (lane fi holds frames 2*fi and 2*fi+1): load, LayerNorm, split --------------
    const int mw0 = tile * GF_TILE + w * WSTR - HALO;           // wave frame 0 (GCFN: tile frame 0 is halo)
    bf16x8 xh[MT][KS], xl[MT][KS];
#if SEPR_GF3_RESX
    float mu_[MT], sg_[MT];
#endif
    float f0[MT], f2[MT];                                        // conv zero-padding flags (sequence start / end)
    bool edge_lane = false;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      int m = mw0 + MT * fi + mt;
      bool valid = (m >= 0 && m < a.M);
      if constexpr (FOLD) {   // tiles walk ONE sequence: frame l of sequence seq, the first FOLD_HB frames of a tile are recomputed halo
        const int seq = tile / a.fold_tps, l = (tile - seq * a.fold_tps) * GF_TILE - FOLD_HB + w * 16 * MT + MT * fi + mt;
        valid = l >= 0 && l < a.T;
        m = seq * a.in_src + l;       // the crop of module.py:250: rows l >= T of the source sequence are never read
      }
      const int trow = (valid && !PLAIN) ? m % a.T : -2;
      f0[mt] = (trow == 0) ? 0.f : 1.f;
      f2[mt] = (trow == a.T - 1) ? 0.f : 1.f;
      edge_lane = edge_lane || trow == 0 || trow == a.T - 1;
      long long mi = valid ? m : 0;
      if (PLAIN && !FOLD && a.in_rows > 0) mi = (long long)(mi / a.in_rows) * a.in_src + mi % a.in_rows;
      const float* xp = a.x + mi * F + 8 * fg;
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
      const float mean = PLAIN ? 0.f : s * (1.0f / F);
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
      const float rstd = valid ? (PLAIN ? 1.0f : 1.0f / sqrtf(d * (1.0f / F) + a.eps)) : 0.f;   // invalid frames: exactly zero
#if SEPR_GF3_RESX
      mu_[mt] = mean;
      sg_[mt] = PLAIN ? 1.0f : sqrtf(d * (1.0f / F) + a.eps);
#endif
      bool own_row = false;
      if (TRAIN) {   // the frames this workgroup OUTPUTS (not its halo) report their statistics: all the backward needs
        const int lr = MT * fi + mt, bf = w * WSTR + lr;
        const bool own = XCH ? (bf >= 1 && bf <= GF_TILE) : (lr >= 1 && lr <= 16 * MT - 2);
        own_row = own && valid;
        if (a.stats && fg == 0 && valid && own) *reinterpret_cast<float2*>(a.stats + 2LL * m) = make_float2(mean, rstd);
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8 h, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xn = (v[ks][e] - mean) * rstd;
          const __bf16 hh = (__bf16)xn;
          h[e] = hh;
          if constexpr (!ONE) l[e] = (__bf16)(xn - (float)hh);
        }
        xh[mt][ks] = h;
        if constexpr (!ONE) xl[mt][ks] = l;
        if constexpr (TRAIN && ONE) {   // side output for the backward: the bf16 rows exactly as the MFMAs read them (16 B per lane and K step)
          if (a.xhat16 && own_row) *reinterpret_cast<bf16x8*>(static_cast<__bf16*>(a.xhat16) + (long long)m * F + 32 * ks + 8 * fg) = h;
        }
      }
    }
    const bool edge = __builtin_amdgcn_ballot_w64(edge_lane) != 0ull;   // wave-uniform
    f32x4 acc[FT][MT];
#pragma unroll
    for (int ft = 0; ft < FT; ++ft)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[ft][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto chunks = [&](auto edge_c) {
      constexpr bool EDGE = decltype(edge_c)::value;
      for (int c = 0; c < ((SEPR_GF_ABL & 1) ? 0 : NCH); ++c) {
        if constexpr (LAT > 0) lat_enter(c);
        bf16x8 gh[MT], gw[MT];          // gated values (bf16 hi / lo) in down-projection k-slot order
        uint4 fb[FBN][2];               // fragment ring: RDX MFMA groups in flight ahead of the one being multiplied
#pragma unroll
        for (int g = 0; g < RDX; ++g) ld_up(0, g, fb[g]);
        f32x4 hvA[UF ? 2 : 1][MT], hgA[UF ? 2 : 1][MT];
#pragma unroll
        for (int jj = 0; jj < (UF ? 4 : 2); ++jj) {
          // UF (up-first): both up-projections, then both convolutions - the copy of the next chunk's up-projection
          // fragments is then issued one conv earlier and has conv + conv + down-projection to land
          const int j = UF ? (jj & 1) : jj;
          const bool do_up = !UF || jj < 2, do_conv = !UF || jj >= 2;
          const float* cs = reinterpret_cast<const float*>(csl + (LAT > 0 ? 0 : (c & 1) * CS_U4)) + j * 160 + 4 * fg;
          f32x4 (&hv)[MT] = hvA[UF ? j : 0];
          f32x4 (&hg)[MT] = hgA[UF ? j : 0];
          if (do_up) {
          // ---- up-projection, accumulators start at the bias ------------------------------------------------
          {
            const float4 bv = ld4(cs), bg = ld4(cs + 16);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              hv[mt] = (f32x4){bv.x, bv.y, bv.z, bv.w};
              hg[mt] = (f32x4){bg.x, bg.y, bg.z, bg.w};
            }
          }
          if (SEPR_GF3_PRIO) __builtin_amdgcn_s_setprio(1);   // MFMA phases win the issue arbitration over the
#pragma unroll                                                 // other workgroup's VALU phases
          for (int rep3 = 0; rep3 < ((SEPR_GF_ABL & 96) ? 3 : 1); ++rep3)
#pragma unroll
          for (int g = 0; g < 2 * KS; ++g) {
            if ((SEPR_GF_ABL & 96) && rep3 > 0 && g < RD) ld_up(j, g, fb[g % (RD + 1)]);   // (proxy: each pass re-reads its fragments)
            if (g + RD < 2 * KS) ld_up(j, g + RD, fb[(g + RD) % (RD + 1)]);
            SEPR_GF3_SCHED_FENCE();
            const bf16x8 wh = *reinterpret_cast<const bf16x8*>(&fb[g % (RD + 1)][0]);
            [[maybe_unused]] const bf16x8 wlo = *reinterpret_cast<const bf16x8*>(&fb[g % (RD + 1)][ONE ? 0 : 1]);
            const int ks = g >> 1;
            if ((g & 1) == 0) {
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) hv[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[mt][ks], hv[mt], 0, 0, 0);
              if constexpr (!ONE) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) hv[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl[mt][ks], hv[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) hv[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, xh[mt][ks], hv[mt], 0, 0, 0);
              }
            } else {
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) hg[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[mt][ks], hg[mt], 0, 0, 0);
              if constexpr (!ONE) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) hg[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl[mt][ks], hg[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) hg[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, xh[mt][ks], hg[mt], 0, 0, 0);
              }
            }
            SEPR_GF3_SCHED_FENCE();
          }
          if (SEPR_GF3_PRIO) __builtin_amdgcn_s_setprio(0);
          if (j == 0) {                            // the second tile pair's first fragments arrive under the conv
#pragma unroll
            for (int g = 0; g < RDX; ++g) ld_up(1, g, fb[g]);
          } else {
            if (XCH) {   // this wave's first / last frame of both tile pairs for the neighbouring waves' conv
#pragma unroll
              for (int jx = 0; jx < 2; ++jx) {
                float* e0 = xch + ((w * 2 + 0) * 2 + jx) * 32 + 4 * fg;
                float* e1 = xch + ((w * 2 + 1) * 2 + jx) * 32 + 4 * fg;
                if (fi == 0) {
                  st4(e0, make_float4(hvA[jx][0][0], hvA[jx][0][1], hvA[jx][0][2], hvA[jx][0][3]));
                  st4(e0 + 16, make_float4(hgA[jx][0][0], hgA[jx][0][1], hgA[jx][0][2], hgA[jx][0][3]));
                }
                if (fi == 15) {
                  st4(e1, make_float4(hvA[jx][MT - 1][0], hvA[jx][MT - 1][1], hvA[jx][MT - 1][2], hvA[jx][MT - 1][3]));
                  st4(e1 + 16, make_float4(hgA[jx][MT - 1][0], hgA[jx][MT - 1][1], hgA[jx][MT - 1][2], hgA[jx][MT - 1][3]));
                }
              }
            }
            if constexpr (LAT == 0) {
            dma_barrier();                         // every wave has read its up-projection fragments of chunk c;
                                                   // this chunk's down-projection fragments have landed
            if (c + 1 < NCH && !(SEPR_GF_ABL & 2)) dma_w1(c + 1);        // lands under the conv + down-projection below
            }
#pragma unroll
            for (int g = 0; g < RDX; ++g)
              if (g < FT) ld_dn(g, fb[g]);
          }
          }
          if (!do_conv) continue;
          if (PLAIN) {   // no conv: GLU straight on the projection (the gate rows of W1 / b1 are pre-scaled by -log2 e)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const float g1 = glu_prescaled(hv[mt][r], hg[mt][r]);
                const __bf16 hh = (__bf16)g1;
                gh[mt][4 * j + r] = hh;
                gw[mt][4 * j + r] = (__bf16)(g1 - (float)hh);
              }
            continue;
          }
          // ---- depthwise k=3 conv along frames, GLU ---------------------------------------------------------
          // frame 2*fi+mt: tile 0's previous frame is the left lane's tile-1 value, its next frame the lane's own
          // tile-1 value (and mirrored for tile 1); the rotations wrap onto the two halo frames only.
          float xpv[4] = {0.f, 0.f, 0.f, 0.f}, xpg[4] = {0.f, 0.f, 0.f, 0.f}, xnv[4] = {0.f, 0.f, 0.f, 0.f}, xng[4] = {0.f, 0.f, 0.f, 0.f};
          if (XCH) {   // last frame of the wave before, first frame of the wave after (the tile's two end frames are halo)
            const int wp = w > 0 ? w - 1 : 0, wn = w + 1 < NW ? w + 1 : NW - 1;
            const float4 a0 = ld4(xch + ((wp * 2 + 1) * 2 + j) * 32 + 4 * fg), a1 = ld4(xch + ((wp * 2 + 1) * 2 + j) * 32 + 16 + 4 * fg);
            const float4 b0 = ld4(xch + ((wn * 2 + 0) * 2 + j) * 32 + 4 * fg), b1 = ld4(xch + ((wn * 2 + 0) * 2 + j) * 32 + 16 + 4 * fg);
            xpv[0] = a0.x; xpv[1] = a0.y; xpv[2] = a0.z; xpv[3] = a0.w;
            xpg[0] = a1.x; xpg[1] = a1.y; xpg[2] = a1.z; xpg[3] = a1.w;
            xnv[0] = b0.x; xnv[1] = b0.y; xnv[2] = b0.z; xnv[3] = b0.w;
            xng[0] = b1.x; xng[1] = b1.y; xng[2] = b1.z; xng[3] = b1.w;
          }
          float gl[MT][4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float wv0 = cs[2 * 16 + r], wv1 = cs[3 * 16 + r], wv2 = cs[4 * 16 + r];
            const float wg0 = cs[5 * 16 + r], wg1 = cs[6 * 16 + r], wg2 = cs[7 * 16 + r];
            const float cbv = cs[8 * 16 + r], cbg = cs[9 * 16 + r];
            float cv[MT], cg[MT], pv[MT], pg[MT], nv[MT], ng[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              cv[mt] = hv[mt][r];
              cg[mt] = hg[mt][r];
            }
            if (XCH) {
              // wave seams: lane SHIFTS leave lane 0 / 15 of the 16-lane row without a source - they keep the shift's `old`
              // operand, the neighbouring wave's frame (one DPP move per neighbour, no select, no rotation fix-up)
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) {
                pv[mt] = mt > 0 ? cv[mt - 1] : dpp_shr1(xpv[r], cv[MT - 1]);
                pg[mt] = mt > 0 ? cg[mt - 1] : dpp_shr1(xpg[r], cg[MT - 1]);
                nv[mt] = mt + 1 < MT ? cv[mt + 1] : dpp_shl1(xnv[r], cv[0]);
                ng[mt] = mt + 1 < MT ? cg[mt + 1] : dpp_shl1(xng[r], cg[0]);
              }
            } else {
#pragma unroll
              for (int mt = 0; mt < MT; ++mt) {
                pv[mt] = mt > 0 ? cv[mt - 1] : dpp_ror1(cv[MT - 1]);
                pg[mt] = mt > 0 ? cg[mt - 1] : dpp_ror1(cg[MT - 1]);
                nv[mt] = mt + 1 < MT ? cv[mt + 1] : dpp_rol1(cv[0]);
                ng[mt] = mt + 1 < MT ? cg[mt + 1] : dpp_rol1(cg[0]);
              }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              const float a0v = EDGE ? wv0 * f0[mt] : wv0, a2v = EDGE ? wv2 * f2[mt] : wv2;
              const float a0g = EDGE ? wg0 * f0[mt] : wg0, a2g = EDGE ? wg2 * f2[mt] : wg2;
              float val = fmaf(a2v, nv[mt], fmaf(wv1, cv[mt], fmaf(a0v, pv[mt], cbv)));
              float gat = fmaf(a2g, ng[mt], fmaf(wg1, cg[mt], fmaf(a0g, pg[mt], cbg)));
              if (SEPR_GF_ABL & 32) { val = (cv[mt] + pv[mt]) + nv[mt]; gat = (cg[mt] + pg[mt]) + ng[mt]; }
              if (SEPR_GF_ABL & 64) { val = cv[mt] + cv[mt]; gat = cg[mt] + cg[mt]; }
              gl[mt][r] = (SEPR_GF_ABL & 16) ? val * gat : glu_prescaled(val, gat);   // 16: no transcendentals (gate taps are pre-scaled)
            }
          }
          if (TRAIN && drop) {   // network.py:55 dropout on the gated tensor: hidden channel 32c + 16j + 4fg + r of frame mw0 + MT*fi + mt
            const unsigned pr = (unsigned)(16 * c + 8 * j + 2 * fg);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
              unsigned row = (unsigned)(mw0 + MT * fi + mt);
              asm volatile("" : "+v"(row));   // keeps the row term of the hash out of loop-invariant registers (no in-loop spills)
              const unsigned d0 = sepr_drop_word(dk0, row, pr), d1 = sepr_drop_word(dk0, row, pr + 1u);
              gl[mt][0] = (d0 & 0xffffu) >= a.drop_thr ? gl[mt][0] : 0.f;
              gl[mt][1] = (d0 >> 16) >= a.drop_thr ? gl[mt][1] : 0.f;
              gl[mt][2] = (d1 & 0xffffu) >= a.drop_thr ? gl[mt][2] : 0.f;
              gl[mt][3] = (d1 >> 16) >= a.drop_thr ? gl[mt][3] : 0.f;
            }
          }
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const __bf16 hh = (__bf16)gl[mt][r];
              gh[mt][4 * j + r] = hh;
              if constexpr (!ONE) gw[mt][4 * j + r] = (__bf16)(gl[mt][r] - (float)hh);
            }
        }
        // ---- down-projection K step of this chunk -----------------------------------------------------------
        if (SEPR_GF3_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ft = 0; ft < FT; ++ft) {
          if (ft + RD < FT) ld_dn(ft + RD, fb[(ft + RD) % (RD + 1)]);
          SEPR_GF3_SCHED_FENCE();
          const bf16x8 wh = *reinterpret_cast<const bf16x8*>(&fb[ft % (RD + 1)][0]);
          [[maybe_unused]] const bf16x8 wlo = *reinterpret_cast<const bf16x8*>(&fb[ft % (RD + 1)][ONE ? 0 : 1]);
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) acc[ft][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, gh[mt], acc[ft][mt], 0, 0, 0);
          if constexpr (!ONE) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[ft][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, gw[mt], acc[ft][mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[ft][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, gh[mt], acc[ft][mt], 0, 0, 0);
          }
          SEPR_GF3_SCHED_FENCE();
        }
        if (SEPR_GF3_PRIO) __builtin_amdgcn_s_setprio(0);
        if constexpr (LAT == 0) {
        dma_barrier();                             // down-projection fragments consumed; chunk c+1's up-projection
        if (c + 1 < NCH && !(SEPR_GF_ABL & 2)) dma_w2(c + 1);            // fragments have landed
        }
      }
    };
    if constexpr (LAT == 0) dma_barrier();     // chunk 0 landed
    if (edge) chunks(bool_c<true>{}); else chunks(bool_c<false>{});
    if constexpr (LAT > 0) __syncthreads();    // every wave is done with the last chunk's stage (nothing is in flight: the last wait was
                                               // vmcnt(0)); the epilogue staging below overwrites the ring

    if constexpr (FOLD) {
      // ---- folded decoder epilogue: acc = the 16 ConvTranspose1d taps of every frame; overlap-add (stride 4) inside the tile,
      //      summed newest frame first like decoder_kernel, so a sample's value does not depend on the tiling ----------------
      constexpr int DS = 20;   // staging row stride in floats (16-byte aligned, spreads the gather over the banks)
      float* const Ds = reinterpret_cast<float*>(wl);
      const int seq = tile / a.fold_tps, jt = tile - seq * a.fold_tps;
      const float4 bf = ld4(a.b2 + 4 * fg);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int lr = w * 16 * MT + MT * fi + mt, l = jt * GF_TILE - FOLD_HB + lr;
        const f32x4 v = acc[0][mt];
        const bool ok = l >= 0 && l < a.T;
        st4(Ds + lr * DS + 4 * fg, ok ? make_float4(v[0] + bf.x, v[1] + bf.y, v[2] + bf.z, v[3] + bf.w) : zero4());
      }
      __syncthreads();
      const int sp = seq % a.out_S, bb = seq / a.out_S;
      float* const dst = a.y + ((long long)sp * (a.fold_nseq / a.out_S) + bb) * a.fold_Tout;
      for (int tl = tid; tl < 4 * GF_TILE; tl += NT) {
        const int tau = 4 * jt * GF_TILE + tl;
        if (tau >= a.fold_Tout) break;
        const int lr = (tl >> 2) + FOLD_HB, ph = tl & 3;
        float yv = 0.f;
#pragma unroll
        for (int jj = 0; jj <= FOLD_HB; ++jj) yv += Ds[(lr - jj) * DS + ph + 4 * jj];
        dst[tau] = yv;
      }
      continue;   // (the next tile starts on a barrier: the staging is consumed before the weight copies overwrite it)
    }
    // ---- epilogue: y = x + ls * (acc + b2), two waves at a time through LDS ---------------------------------
#if SEPR_GF3_REDERIVE
    if constexpr ((SEPR_GF3_REDERIVE & 2) != 0) {              // the epilogue's per-thread addresses are derived HERE, not in front of the chunk loop
      asm volatile("" : "+v"(tid));
      lane = tid & 63; w = tid >> 6; fi = lane & 15; fg = lane >> 4;
    }
#endif
    float* const Os = reinterpret_cast<float*>(wl);
    constexpr int WPP = 64 / (16 * MT);   // waves per 64-frame epilogue pass
    constexpr int Q = F / 4;                 // float4 per row
    constexpr int RPP = NT / Q;              // rows per pass
    constexpr int NP = (64 + RPP - 1) / RPP;
    const int q4 = tid % Q, rr = tid / Q;
    const float4 b2 = ld4(a.b2 + 4 * q4);
    float4 lsv = PLAIN ? zero4() : ld4(a.ls + 4 * q4);
    const float dsc = (TRAIN && drop) ? a.drop_scale : 1.0f;   // keep scale of the gated-tensor dropout: linear, applied to the sum
    if (TRAIN && drop) { lsv.x *= dsc; lsv.y *= dsc; lsv.z *= dsc; lsv.w *= dsc; }   // ... and of the output dropout
#pragma unroll 1
    for (int half = 0; half < EH; ++half) {
      if (half > 0) __syncthreads();   // previous pass fully stored (the chunk loop ended on a barrier)
      // the residual rows of this pass are requested first, from clamped branch-free addresses: they fly under
      // the staging writes and the barrier, and the pass pays one global-load latency instead of one per row
      float4 xr[NP];
      int mrow[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int row = rr + p * RPP;          // 0..63: WPP waves x 16*MT frames
        const int ww = WPP * half + row / (16 * MT), lr = row % (16 * MT);
        const int bf = ww * WSTR + lr;                           // frame inside the workgroup tile (GCFN: 0 = halo)
        const int m = tile * GF_TILE - HALO + bf;
        const bool ok = row < 64 && ww < NW && m < a.M &&
                        (PLAIN ? true : (XCH ? (bf >= 1 && bf <= GF_TILE) : (lr >= 1 && lr <= 16 * MT - 2)));
        mrow[p] = ok ? m : -1;
        // (timing ablations, wrong results: SEPR_GF_ABL & 128 = no residual re-read at all - the upper bound of what removing the second read of x
        //  could buy; & 256 = the re-read comes from a COLD address range (the output tensor of the launch) instead of the rows the tile loaded
        //  ~20 us earlier - if that costs nothing either, the re-read is off the critical path wherever it is served from)
        xr[p] = (PLAIN || (SEPR_GF3_RESX && !TRAIN) || (SEPR_GF_ABL & 128)) ? zero4()
                : ld4(((SEPR_GF_ABL & 256) ? a.y : a.x) + (long long)(ok ? m : 0) * F + 4 * q4);
      }
      if (w / WPP == half) {
        float* base = Os + (w % WPP) * (16 * MT) * OS;
#pragma unroll
        for (int ft = 0; ft < FT; ++ft)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            f32x4 v = acc[ft][mt];
#if SEPR_GF3_RESX
            if (!PLAIN && !TRAIN) {   // the FINAL y = x_rec + ls (acc + b2) is staged; the store pass below only copies
              const int c0 = 32 * (ft >> 1) + 8 * fg + 4 * (ft & 1);
              const float4 l4 = ld4(a.ls + c0), b4 = ld4(a.b2 + c0);
              const float ll[4] = {l4.x, l4.y, l4.z, l4.w}, bb4[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int e = 4 * (ft & 1) + r;
                const float xn = (float)xh[mt][ft >> 1][e] + (float)xl[mt][ft >> 1][e];
                v[r] = fmaf(v[r] + bb4[r], ll[r], fmaf(xn, sg_[mt], mu_[mt]));
              }
            }
#endif
            st4(base + (MT * fi + mt) * OS + 32 * (ft >> 1) + 8 * fg + 4 * (ft & 1), make_float4(v[0], v[1], v[2], v[3]));   // w2p row order
          }
      }
      __syncthreads();
      {
#pragma clang fp contract(off)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          if (mrow[p] >= 0) {
            const float4 o = ld4(Os + (rr + p * RPP) * OS + 4 * q4);
            if (PLAIN) {
              long long mo = mrow[p];
              if (a.out_S > 0) mo = ((long long)(mo / a.out_T) * a.out_S + a.out_s) * a.out_T + mo % a.out_T;
              st4(a.y + mo * a.ldy + a.col_off + 4 * q4, make_float4(o.x + b2.x, o.y + b2.y, o.z + b2.z, o.w + b2.w));
            } else if (SEPR_GF3_RESX && !TRAIN) {
              st4(a.y + (long long)mrow[p] * F + 4 * q4, o);
            } else {
              float4 v = TRAIN ? make_float4(fmaf(o.x, dsc, b2.x), fmaf(o.y, dsc, b2.y), fmaf(o.z, dsc, b2.z), fmaf(o.w, dsc, b2.w))
                               : make_float4(o.x + b2.x, o.y + b2.y, o.z + b2.z, o.w + b2.w);
              if (TRAIN && drop) {   // network.py:57 dropout on the block's output (its keep scale rides in lsv)
                const unsigned d0 = sepr_drop_word(dk1, (unsigned)mrow[p], 2u * q4), d1 = sepr_drop_word(dk1, (unsigned)mrow[p], 2u * q4 + 1u);
                v.x = (d0 & 0xffffu) >= a.drop_thr ? v.x : 0.f;
                v.y = (d0 >> 16) >= a.drop_thr ? v.y : 0.f;
                v.z = (d1 & 0xffffu) >= a.drop_thr ? v.z : 0.f;
                v.w = (d1 >> 16) >= a.drop_thr ? v.w : 0.f;
              }
              st4(a.y + (long long)mrow[p] * F + 4 * q4,
                  make_float4(fmaf(v.x, lsv.x, xr[p].x), fmaf(v.y, lsv.y, xr[p].y), fmaf(v.z, lsv.z, xr[p].z), fmaf(v.w, lsv.w, xr[p].w)));
            }
          }
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------
// gcfn_hs_kernel (round 6): the HIDDEN-SPLIT form for launches with fewer tiles than CUs (batch 1, Engine._inference_sample).
// In the row-stationary kernel above every wave walks the WHOLE hidden dimension for its own 16*MT frames: with one 16-frame wave per SIMD
// each weight fragment read from LDS feeds three MFMAs, every wave reads all 590 KB of packed weights out of LDS, and the tile pays 12
// chunk barriers - 23.5 us per launch where the MFMAs alone are 5.8 us (profiles/r06_b1_launches.txt).  Here the four waves of a workgroup
// hold the SAME 16*MT frames and split the hidden dimension instead:
//   phase 1  wave w owns chunks w, w+4, w+8: up-projection, conv, GLU for all frames of the tile.  A weight fragment has exactly one
//            reader, so it goes global -> registers (one coalesced 1 KiB load per fragment, requested a half-chunk ahead), not through LDS;
//            each fragment now feeds 3*MT MFMAs.  The gated tensor (bf16 hi / lo, already in the down-projection's B-fragment lane
//            layout) is written to LDS: [chunk = K step][plane][frame tile][lane] x 16 B;
//   barrier  the only one of the tile;
//   phase 2  wave w owns output tiles 2w, 2w+1 (32 of the 128 channels) for all frames: the K steps are walked in chunk order with the
//            three products in the order of the kernel above, B fragments from LDS, its 48 KB slice of W2 global -> registers (requested
//            under the last chunk's conv).  A lane ends with 8 consecutive channels of a frame: bias, LayerScale, residual, direct stores.
// Every accumulator sees exactly the operand sequence of gcfn_fused3_kernel (same packed weights, same products, same order): the result
// is bit-identical, which the parity tests check against the batched launch (tests/test_gpu_parity.py).  F = 128 only.
// ---------------------------------------------------------------------------------------------------------
#ifndef SEPR_HS_WGS
#define SEPR_HS_WGS 1   // workgroups per CU the 30-frame instantiation is compiled for; 2 (256 registers per wave) spills 124 dwords: the tile's
                        // live set is 64 (frame planes) + 128 (fragment prefetch) + 48 (accumulators, gated planes) + the conv's temporaries
#endif
// PLAIN (MODE 1 of the kernel above: SpkSplit's and OutputLayer's GLU-MLP): no LayerNorm, conv, halo, LayerScale, residual; CPW = hidden
// chunks per wave (a.nch = 4 CPW chunks: 2 for OutputLayer, 4 for SpkSplit, 3 for the GCFN block), input / output row maps as there.
template <int MT, int CPW = 3, bool PLAIN = false>
__global__ __launch_bounds__(256, MT == 2 ? SEPR_HS_WGS : 1) void gcfn_hs_kernel(const GcfnFusedArgs a) {
  constexpr int F = 128, KS = F / 32, NW = 4, NCH = NW * CPW, FT = F / 16, FTW = FT / NW;
  constexpr int NG = NCH / 4;                        // groups of 4 K steps of the down-projection (one 16-fragment register block each)
  static_assert(PLAIN || CPW == 3, "the GCFN block has 3F / 32 = 12 hidden chunks");
  constexpr int W1F_U4 = 4 * KS * 2 * 64, CS_U4 = 256, W1_U4 = W1F_U4 + CS_U4, W2_U4 = FT * 2 * 64;
  constexpr int HALO = PLAIN ? 0 : 1;
  constexpr int TILE = 16 * MT - 2 * HALO;           // output frames per workgroup (GCFN: first and last frame of the tile are recomputed halo)
  constexpr int CSF = 320;                           // live floats of a chunk's constant block ([2 tile pairs][10][16])
  static_assert(CPW * NW == NCH && FTW * NW == FT && FTW == 2, "hidden / output split over the four waves");
  __shared__ __attribute__((aligned(16))) uint4 hs[NCH * 2 * MT * 64];     // gated tensor: [K step][plane][frame tile][lane]
  __shared__ __attribute__((aligned(16))) float cst[NW][CPW][CSF];          // this wave's chunk constants (wave-private: no barrier)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int fi = lane & 15, fg = lane >> 4;
  const uint4* const W1g = static_cast<const uint4*>(a.w1p);
  const uint4* const W2g = static_cast<const uint4*>(a.w2p);
  const int tile = blockIdx.x;
  const int mw0 = tile * TILE - HALO;

  // wave-uniform bases + ONE 32-bit per-lane byte offset (SGPR-base addressing: no 64-bit per-lane pointers to keep alive across the tile)
  const int ws = __builtin_amdgcn_readfirstlane(w);
  const unsigned loff = (unsigned)lane * 16u;
  auto ldu = [&](const uint4* base, int blk) -> uint4 {   // 16 bytes of this lane from the 1 KiB block blk behind the wave-uniform base
    return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base + blk * 64) + loff);
  };
  // fragments of one tile pair j of chunk c: [value | gate][K step][plane]
  auto ld_w1 = [&](int c, int j, uint4 (&d)[16]) {
    const uint4* base = W1g + (long long)c * W1_U4;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int k2 = 0; k2 < 2 * KS; ++k2) d[t * 2 * KS + k2] = ldu(base, (t * 2 + j) * KS * 2 + k2);
  };
  // this wave's W2 fragments of K steps 4q .. 4q+3: [step][tile 2w | 2w+1][plane]
  auto ld_w2 = [&](int q, uint4 (&d)[16]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const uint4* base = W2g + (long long)(4 * q + s) * W2_U4 + (FTW * ws) * 2 * 64;
#pragma unroll
      for (int e = 0; e < 4; ++e) d[s * 4 + e] = ldu(base, e);
    }
  };
  // request order = arrival order: the chunk constants and the frames first (the LayerNorm starts as soon as they are in), the first
  // chunk's 32 KB of fragments land under it
  uint4 ct[CPW][2];
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    const uint4* base = W1g + (long long)(ws + NW * i) * W1_U4 + W1F_U4;
    ct[i][0] = ldu(base, 0);
    ct[i][1] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base + 64) + (loff & 255u));
  }
  // ---- the tile's frames (every wave holds all of them; lane fi holds frames MT*fi .. MT*fi + MT-1): load, LayerNorm, split ----
  bf16x8 xh[MT][KS], xl[MT][KS];
  float f0[MT], f2[MT];
  bool edge_lane = false;
  float xv[MT][KS][8];
  bool vld[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mw0 + MT * fi + mt;
    const bool valid = (m >= 0 && m < a.M);
    vld[mt] = valid;
    const int trow = (valid && !PLAIN) ? m % a.T : -2;
    f0[mt] = (trow == 0) ? 0.f : 1.f;
    f2[mt] = (trow == a.T - 1) ? 0.f : 1.f;
    edge_lane = edge_lane || trow == 0 || trow == a.T - 1;
    long long mi = valid ? m : 0;
    if (PLAIN && a.in_rows > 0) mi = (long long)(mi / a.in_rows) * a.in_src + mi % a.in_rows;
    const float* xp = a.x + mi * F + 8 * fg;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float4 p = ld4(xp + 32 * ks), q = ld4(xp + 32 * ks + 4);
      xv[mt][ks][0] = p.x; xv[mt][ks][1] = p.y; xv[mt][ks][2] = p.z; xv[mt][ks][3] = p.w;
      xv[mt][ks][4] = q.x; xv[mt][ks][5] = q.y; xv[mt][ks][6] = q.z; xv[mt][ks][7] = q.w;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  uint4 wa[16], wb[16], wc[16];
  ld_w1(ws, 0, wa);
  ld_w1(ws, 1, wb);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    float (&v)[KS][8] = xv[mt];
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[ks][e];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mean = PLAIN ? 0.f : s * (1.0f / F);
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
    const float rstd = vld[mt] ? (PLAIN ? 1.0f : 1.0f / sqrtf(d * (1.0f / F) + a.eps)) : 0.f;   // invalid frames: exactly zero
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
  const bool edge = __builtin_amdgcn_ballot_w64(edge_lane) != 0ull;   // wave-uniform
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    reinterpret_cast<uint4*>(&cst[w][i][0])[lane] = ct[i][0];
    if (lane < 16) reinterpret_cast<uint4*>(&cst[w][i][0])[64 + lane] = ct[i][1];
  }

  // ---- phase 1: this wave's chunks ----------------------------------------------------------------------------------------------
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    const int c = ws + NW * i;
    const bool last = i + 1 == CPW;
    f32x4 hvA[2][MT], hgA[2][MT];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float* cs = &cst[w][i][0] + j * 160 + 4 * fg;
      uint4 (&wf)[16] = j == 0 ? wa : wb;
      f32x4 (&hv)[MT] = hvA[j];
      f32x4 (&hg)[MT] = hgA[j];
      {
        const float4 bv = ld4(cs), bg = ld4(cs + 16);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          hv[mt] = (f32x4){bv.x, bv.y, bv.z, bv.w};
          hg[mt] = (f32x4){bg.x, bg.y, bg.z, bg.w};
        }
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 vh = *reinterpret_cast<const bf16x8*>(&wf[2 * ks]), vl = *reinterpret_cast<const bf16x8*>(&wf[2 * ks + 1]);
        const bf16x8 gh_ = *reinterpret_cast<const bf16x8*>(&wf[2 * KS + 2 * ks]), gl_ = *reinterpret_cast<const bf16x8*>(&wf[2 * KS + 2 * ks + 1]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) hv[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, xh[mt][ks], hv[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) hg[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh_, xh[mt][ks], hg[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) hv[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vh, xl[mt][ks], hv[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) hg[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gh_, xl[mt][ks], hg[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) hv[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vl, xh[mt][ks], hv[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) hg[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(gl_, xh[mt][ks], hg[mt], 0, 0, 0);
      }
      // the registers of this tile pair's fragments are free: request what runs in them next (the next chunk's pair, or W2)
      // (scheduling fences: hipcc otherwise sinks the loads down to their first use - and waits for each of them there)
      __builtin_amdgcn_sched_barrier(0);
      if (!(SEPR_GF_ABL & 512)) {   // (timing ablation 512, wrong results: the weight fragments are requested once)
        if (!last) ld_w1(c + NW, j, wf);
        else if (j < NG) ld_w2(j, wf);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (last && NG > 2 && !(SEPR_GF_ABL & 512)) ld_w2(2, wc);     // (the frame planes are dead from here on)
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 gh[MT], gw[MT];      // gated values (bf16 hi / lo) in down-projection k-slot order
    auto conv = [&](int j, auto edge_c) {
      constexpr bool EDGE = decltype(edge_c)::value;
      const float* cs = &cst[w][i][0] + j * 160 + 4 * fg;
      f32x4 (&hv)[MT] = hvA[j];
      f32x4 (&hg)[MT] = hgA[j];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float wv0 = cs[2 * 16 + r], wv1 = cs[3 * 16 + r], wv2 = cs[4 * 16 + r];
        const float wg0 = cs[5 * 16 + r], wg1 = cs[6 * 16 + r], wg2 = cs[7 * 16 + r];
        const float cbv = cs[8 * 16 + r], cbg = cs[9 * 16 + r];
        float cv[MT], cg[MT], pv[MT], pg[MT], nv[MT], ng[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          cv[mt] = hv[mt][r];
          cg[mt] = hg[mt][r];
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {   // the rotations wrap onto the two halo frames only
          pv[mt] = mt > 0 ? cv[mt - 1] : dpp_ror1(cv[MT - 1]);
          pg[mt] = mt > 0 ? cg[mt - 1] : dpp_ror1(cg[MT - 1]);
          nv[mt] = mt + 1 < MT ? cv[mt + 1] : dpp_rol1(cv[0]);
          ng[mt] = mt + 1 < MT ? cg[mt + 1] : dpp_rol1(cg[0]);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const float a0v = EDGE ? wv0 * f0[mt] : wv0, a2v = EDGE ? wv2 * f2[mt] : wv2;
          const float a0g = EDGE ? wg0 * f0[mt] : wg0, a2g = EDGE ? wg2 * f2[mt] : wg2;
          const float val = fmaf(a2v, nv[mt], fmaf(wv1, cv[mt], fmaf(a0v, pv[mt], cbv)));
          const float gat = fmaf(a2g, ng[mt], fmaf(wg1, cg[mt], fmaf(a0g, pg[mt], cbg)));
          const float g1 = glu_prescaled(val, gat);
          const __bf16 hh = (__bf16)g1;
          gh[mt][4 * j + r] = hh;
          gw[mt][4 * j + r] = (__bf16)(g1 - (float)hh);
        }
      }
    };
    if constexpr (PLAIN) {   // no conv: GLU straight on the projection (the gate rows of W1 / b1 are pre-scaled by -log2 e)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float g1 = glu_prescaled(hvA[j][mt][r], hgA[j][mt][r]);
            const __bf16 hh = (__bf16)g1;
            gh[mt][4 * j + r] = hh;
            gw[mt][4 * j + r] = (__bf16)(g1 - (float)hh);
          }
    } else
    if (SEPR_GF_ABL & 1024) {   // (timing ablation 1024, wrong results: no conv / GLU / split)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        gh[mt] = *reinterpret_cast<const bf16x8*>(&hvA[0][mt]);
        gw[mt] = *reinterpret_cast<const bf16x8*>(&hgA[1][mt]);
      }
    } else
    // (the sequence-boundary form of the conv is taken by the whole wave or not at all: one small branch per chunk, the MFMA code is common)
    if (edge) {
      conv(0, bool_c<true>{});
      conv(1, bool_c<true>{});
    } else {
      conv(0, bool_c<false>{});
      conv(1, bool_c<false>{});
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      *reinterpret_cast<bf16x8*>(&hs[((c * 2 + 0) * MT + mt) * 64 + lane]) = gh[mt];
      *reinterpret_cast<bf16x8*>(&hs[((c * 2 + 1) * MT + mt) * 64 + lane]) = gw[mt];
    }
  }

  // the residual rows (8 consecutive channels of this wave's 32 per lane and frame) fly under the barrier and phase 2
  float4 xr[MT][2];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mw0 + MT * fi + mt;
    const float* xp = a.x + (long long)((m >= 0 && m < a.M) ? m : 0) * F + 32 * w + 8 * fg;
    xr[mt][0] = PLAIN ? zero4() : ld4(xp);
    xr[mt][1] = PLAIN ? zero4() : ld4(xp + 4);
  }
  __syncthreads();

  // ---- phase 2: output tiles 2w, 2w+1 over all K steps, chunk order ---------------------------------------------------------------
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
  for (int c = 0; c < NCH; ++c) {
    if (c + 1 < NCH) {
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) hb[(c + 1) & 1][pl][mt] = hs[(((c + 1) * 2 + pl) * MT + mt) * 64 + lane];
    }
    uint4 (&wq)[16] = (c >> 2) % 3 == 0 ? wa : ((c >> 2) % 3 == 1 ? wb : wc);
    const int s = c & 3;
    bf16x8 gh[MT], gw[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      gh[mt] = *reinterpret_cast<const bf16x8*>(&hb[c & 1][0][mt]);
      gw[mt] = *reinterpret_cast<const bf16x8*>(&hb[c & 1][1][mt]);
    }
    bf16x8 wh[FTW], wlo[FTW];
#pragma unroll
    for (int t = 0; t < FTW; ++t) {
      wh[t] = *reinterpret_cast<const bf16x8*>(&wq[s * 4 + t * 2]);
      wlo[t] = *reinterpret_cast<const bf16x8*>(&wq[s * 4 + t * 2 + 1]);
    }
#pragma unroll
    for (int t = 0; t < FTW; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[t], gh[mt], acc[t][mt], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < FTW; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[t], gw[mt], acc[t][mt], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < FTW; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo[t], gh[mt], acc[t][mt], 0, 0, 0);
    if (s == 3 && (c >> 2) + 3 < NG) {   // (SpkSplit: 16 K steps) this group's register block takes the fragments of the group three ahead
      __builtin_amdgcn_sched_barrier(0);
      ld_w2((c >> 2) + 3, wq);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: y = x + ls * (acc + b2); fragment row 4 fg + r of tile 2w + t is channel 32 w + 8 fg + 4 t + r ---------------------
  {
#pragma clang fp contract(off)
    const int ch = 32 * w + 8 * fg;
    const float4 b2a = ld4(a.b2 + ch), b2b = ld4(a.b2 + ch + 4);
    const float4 lsa = PLAIN ? zero4() : ld4(a.ls + ch), lsb = PLAIN ? zero4() : ld4(a.ls + ch + 4);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int lr = MT * fi + mt, m = mw0 + lr;
      if (lr >= HALO && lr <= 16 * MT - 1 - HALO && m < a.M) {
        const f32x4 o0 = acc[0][mt], o1 = acc[1][mt];
        const float4 v0 = make_float4(o0[0] + b2a.x, o0[1] + b2a.y, o0[2] + b2a.z, o0[3] + b2a.w);
        const float4 v1 = make_float4(o1[0] + b2b.x, o1[1] + b2b.y, o1[2] + b2b.z, o1[3] + b2b.w);
        if constexpr (PLAIN) {
          long long mo = m;
          if (a.out_S > 0) mo = ((long long)(mo / a.out_T) * a.out_S + a.out_s) * a.out_T + mo % a.out_T;
          float* yq = a.y + mo * a.ldy + a.col_off + ch;
          st4(yq, v0);
          st4(yq + 4, v1);
          continue;
        }
        float* yp = a.y + (long long)m * F + ch;
        st4(yp, make_float4(fmaf(v0.x, lsa.x, xr[mt][0].x), fmaf(v0.y, lsa.y, xr[mt][0].y), fmaf(v0.z, lsa.z, xr[mt][0].z), fmaf(v0.w, lsa.w, xr[mt][0].w)));
        st4(yp + 4, make_float4(fmaf(v1.x, lsb.x, xr[mt][1].x), fmaf(v1.y, lsb.y, xr[mt][1].y), fmaf(v1.z, lsb.z, xr[mt][1].z), fmaf(v1.w, lsb.w, xr[mt][1].w)));
      }
    }
  }
}

// (The X/Y two-group experiment "v5" - one 8-wave workgroup per CU, MFMA stream and VALU stream in sibling waves - was
//  measured not faster in round 2 and removed in round 4; it lives in git history at a01c6d4, sepr_gcfn_fused5.inc.)
// (A one-workgroup-per-CU, software-pipelined variant of this kernel - "v4", 8 waves, doubled weight buffers, one
//  barrier per chunk - was measured 10 % slower and removed in round 2; it lives in git history at 7cf5a73.)
#ifndef SEPR_GF3_MT
#define SEPR_GF3_MT 2   // v3 frame tiles per wave: 2 -> 4 waves x 30 frames (2 waves per SIMD), 1 -> 6 waves x 14 frames (3 per SIMD)
#endif
[[maybe_unused]] constexpr int GF3_MT = SEPR_GF3_MT, GF3_NW = (SEPR_GF3_MT == 1) ? 6 : 4;

// Latency form of the small-launch instantiations (template parameter LAT): taken when a launch has at most one tile per CU, i.e. when its
// duration IS one workgroup's chunk walk (batch 1, Engine._inference_sample): SEPR_GF_LAT=0 switches it off.
static int lat_ring() {
  static const int v = [] {
    const char* e = getenv("SEPR_GF_LAT");
    return (e && e[0] ? atoi(e) : 3) != 0 ? 3 : 0;
  }();
  return v;
}
static int lat_nw() {   // A/B: SEPR_GF_LAT_NW=6 keeps the 6-wave tiles for every latency-form launch
  static const int v = [] {
    const char* e = getenv("SEPR_GF_LAT_NW");
    return e && e[0] ? atoi(e) : 4;
  }();
  return v;
}
static int lat_max_tiles() {
  static const int v = [] {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return cus;
  }();
  return v;
}

// Frame tiles per wave (2 / 3 / 4: 30- / 46- / 62-frame workgroup tiles) of the hidden-split form for a launch of M rows, 0 = not taken: the smallest
// tile with which the launch fits one workgroup per CU (per-launch times by size: profiles/r06_gcfn_hidden_split.txt).  SEPR_GF_HS=0 switches the
// form off, 2 / 3 / 4 force one tile size (A/B).
static int hs_tiles(int M) {
  static const int force = [] {
    const char* e = getenv("SEPR_GF_HS");
    return e && e[0] ? atoi(e) : -1;
  }();
  const int cus = lat_max_tiles();
  if (force == 0) return 0;
  if (force >= 2 && force <= 4) return (M + 16 * force - 3) / (16 * force - 2) <= cus ? force : 0;
  for (int mt = 2; mt <= 4; ++mt)
    if ((M + 16 * mt - 3) / (16 * mt - 2) <= cus) return mt;
  return 0;
}

static int hs_tiles_plain(int M) {   // same for the PLAIN instantiations (no halo: 32- / 48-frame tiles); SEPR_GF_HS_PLAIN=0 switches them off
  static const int on = [] {
    const char* e = getenv("SEPR_GF_HS_PLAIN");
    const char* g = getenv("SEPR_GF_HS");
    return !((e && e[0] == '0') || (g && g[0] == '0'));
  }();
  if (!on) return 0;
  const int cus = lat_max_tiles();
  for (int mt = 2; mt <= 3; ++mt)
    if ((M + 16 * mt - 1) / (16 * mt) <= cus) return mt;
  return 0;
}

// Plain GLU-MLP (MODE 1): one launch computes F = 128 output columns; a wider output takes one launch per 128 columns
// (the up-projection is recomputed: 1.5x the MFMAs of a single pass, still ~2.5x faster than two generic projections with the
// [rows, hidden] tensor through HBM).
int launch_glumlp_fused(const GcfnFusedArgs& a, int F, int site, hipStream_t stream) {
  if (a.M <= 0) return SEPR_OK;
  if (!a.x || !a.y || !a.w1p || !a.w2p || !a.b2 || a.nch <= 0 || a.ldy < F || a.x == a.y || (F != 128 && F != 256)) return SEPR_EINVAL;
  long long slot = -1;
  const bool timed = prof_begin(site, stream, &slot);
  const int cap = persistent_grid();
  if (F == 256) {   // Large (round 6): F output columns per launch in the one-wave-per-SIMD regime, as launch_gcfn_fused does at F = 256
    const int ntiles = (a.M + 127) / 128, cus = lat_max_tiles();
    hipLaunchKernelGGL((gcfn_fused3_kernel<256, 2, 4, 1>), dim3(ntiles < cus ? ntiles : cus), dim3(256), 0, stream, a);
    if (timed) prof_end(slot, (double)a.M * (2.0 * F * 64.0 * a.nch + 2.0 * 32.0 * a.nch * F), stream);
    SEPR_CHECK_LAUNCH("glumlp_fused_kernel<256>");
    return SEPR_OK;
  }
  if (const int mt = (a.nch == 8 || a.nch == 16) ? hs_tiles_plain(a.M) : 0) {
    // one tile per CU at most: the hidden-split form (gcfn_hs_kernel, PLAIN), 32- or 48-frame tiles
    const int nt = (a.M + 16 * mt - 1) / (16 * mt);
    if (a.nch == 8 && mt == 2) hipLaunchKernelGGL((gcfn_hs_kernel<2, 2, true>), dim3(nt), dim3(256), 0, stream, a);
    else if (a.nch == 8) hipLaunchKernelGGL((gcfn_hs_kernel<3, 2, true>), dim3(nt), dim3(256), 0, stream, a);
    else if (mt == 2) hipLaunchKernelGGL((gcfn_hs_kernel<2, 4, true>), dim3(nt), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((gcfn_hs_kernel<3, 4, true>), dim3(nt), dim3(256), 0, stream, a);
  } else
  if (a.M < 12000) {
    const int ntiles = (a.M + 95) / 96;
    // one tile per CU at most: the latency form (3-stage weight ring, one workgroup per CU), 4 waves (one per SIMD) when that still fits
    const int lat = ntiles <= lat_max_tiles() ? lat_ring() : 0;
    const int nt4 = (a.M + 63) / 64;
    if (lat && nt4 <= lat_max_tiles() && lat_nw() != 6)
      hipLaunchKernelGGL((gcfn_fused3_kernel<128, 1, 4, 1, false, false, 3>), dim3(nt4), dim3(256), 0, stream, a);
    else if (lat) hipLaunchKernelGGL((gcfn_fused3_kernel<128, 1, 6, 1, false, false, 3>), dim3(ntiles), dim3(384), 0, stream, a);
    else hipLaunchKernelGGL((gcfn_fused3_kernel<128, 1, 6, 1>), dim3(ntiles < cap ? ntiles : cap), dim3(384), 0, stream, a);
  } else {
    const int ntiles = (a.M + 127) / 128;
    hipLaunchKernelGGL((gcfn_fused3_kernel<128, 2, 4, 1>), dim3(ntiles < cap ? ntiles : cap), dim3(256), 0, stream, a);
  }
  if (timed) prof_end(slot, (double)a.M * (2.0 * F * 64.0 * a.nch + 2.0 * 32.0 * a.nch * F), stream);
  SEPR_CHECK_LAUNCH("glumlp_fused_kernel");
  return SEPR_OK;
}

// OutputLayer (no mask) + AudioDecoder in ONE launch (modules/module.py:249-256 + :278-283 with masking = False, model.py:28,42-44):
// there is no nonlinearity between end_conv1x1.2 (2F -> N, bias) and ConvTranspose1d(N -> 1, k = 16, stride 4), so the packers fold them
// (fp64): W_fold[16, 2F] = wdec^T W2, b_fold[16] = wdec^T b2.  The down-projection is then ONE 16-row tile (the 16 taps of a frame) and
// the epilogue is the transposed convolution's overlap-add.  Tiles walk one sequence each (125 output frame slots + 3 recomputed halo
// frames), the [rows, N] basis tensor and the separate decoder launch are gone.  a: x [nseq, in_src, F], y = wav [out_S, nseq / out_S,
// fold_Tout], T = frames per sequence L, fold_nseq sequences, b2 = b_fold, w2p = [nch][1][2][64][8] bf16 k-slot fragments of W_fold.
int launch_glumlp_fold(const GcfnFusedArgs& a_in, int F, int site, hipStream_t stream) {
  if (a_in.fold_nseq <= 0 || a_in.T <= 0) return SEPR_OK;
  GcfnFusedArgs a = a_in;
  if (!a.x || !a.y || !a.w1p || !a.w2p || !a.b2 || a.nch <= 0 || (F != 128 && F != 256) || a.out_S <= 0 || a.fold_nseq % a.out_S != 0 || a.in_src < a.T)
    return SEPR_EINVAL;
  constexpr int tile_slots = 4 * 16 * 2 - 3;
  a.fold_tps = (a.T + 3 + tile_slots - 1) / tile_slots;
  a.fold_Tout = 4 * (a.T - 1) + 16;
  a.M = a.fold_nseq * a.in_src;
  long long slot = -1;
  const bool timed = prof_begin(site, stream, &slot);
  const int cap = persistent_grid();
  const long long ntiles = (long long)a.fold_nseq * a.fold_tps;
  if (F == 256) {
    const int cus = lat_max_tiles();
    hipLaunchKernelGGL((gcfn_fused3_kernel<256, 2, 4, 2>), dim3((int)(ntiles < cus ? ntiles : cus)), dim3(256), 0, stream, a);
  } else {
    hipLaunchKernelGGL((gcfn_fused3_kernel<128, 2, 4, 2>), dim3((int)(ntiles < cap ? ntiles : cap)), dim3(256), 0, stream, a);
  }
  // algorithmic FLOPs of what the launch replaces: both projections + the transposed convolution
  if (timed) prof_end(slot, (double)a.fold_nseq * a.T * (2.0 * F * 64.0 * a.nch + 2.0 * 32.0 * a.nch * a.fold_N + 2.0 * a.fold_N * 16.0), stream);
  SEPR_CHECK_LAUNCH("glumlp_fold_kernel");
  return SEPR_OK;
}

int launch_gcfn_fused(const GcfnFusedArgs& a_in, int F, int site, hipStream_t stream) {
  if (a_in.M <= 0) return SEPR_OK;
  static const int stagger = [] {
    const char* e = getenv("SEPR_GF_STAGGER");
    return e && e[0] ? atoi(e) : 0;
  }();
  GcfnFusedArgs a = a_in;
  a.stagger = stagger;
  if (!a.x || !a.y || !a.w1p || !a.w2p || !a.b2 || !a.ls || a.T <= 0 || (F != 64 && F != 128 && F != 256)) return SEPR_EINVAL;
  if (a.x == a.y) return SEPR_EINVAL;   // halo frames of a tile are outputs of its neighbours
  if (F == 256) {
    // Large (round 6): the row-stationary form at F = 256 needs 128 registers of frame planes + 128 of down-projection accumulators per
    // wave - it exists in the ONE-wave-per-SIMD regime (512 registers per wave: four 30-frame waves, one 106 KB workgroup per CU), which the
    // inline-asm weight copies make viable (they fly under the multiplies instead of being waited for).  Inference only.
    if (a.train) return SEPR_EINVAL;
    long long slot256 = -1;
    const bool timed256 = prof_begin(site, stream, &slot256);
    const int cus = lat_max_tiles();
    const int nt = (a.M + 4 * 32 - 2 - 1) / (4 * 32 - 2);
    hipLaunchKernelGGL((gcfn_fused3_kernel<256, 2, 4>), dim3(nt < cus ? nt : cus), dim3(256), 0, stream, a);
    if (timed256) prof_end(slot256, (double)a.M * (2.0 * F * 6 * F + 2.0 * 3 * 6 * F + 2.0 * 3 * F * F), stream);
    SEPR_CHECK_LAUNCH("gcfn_fused_kernel<256>");
    return SEPR_OK;
  }
  if (a.planes == 1 && !a.train) return SEPR_EINVAL;   // plain bf16 operands: a training arithmetic only
  long long slot = -1;
  const bool timed = prof_begin(site, stream, &slot);
  // Small launches (fewer 120-frame tiles than workgroup slots, e.g. batch 1 or the bottleneck stage) take the
  // 14-frame-wave instantiation: 1.4x more, shorter tiles.  A frame's arithmetic does not depend on the tiling, so the
  // result is bit-identical; for large launches the 30-frame form is 1.7x faster per frame (weight re-use).
  static const int small_rows = [] {
    const char* e = getenv("SEPR_GF_SMALL_ROWS");
    return e && e[0] ? atoi(e) : 17000;
  }();
  if (GF3_MT == 2 && a.M < small_rows) {
    constexpr int tile_rows = 6 * 14;
    const int ntiles = (a.M + tile_rows - 1) / tile_rows;
    const int cap = persistent_grid();
    const int grid = ntiles < cap ? ntiles : cap;
    if (a.train && a.planes == 1) {
      if (F == 128) hipLaunchKernelGGL((gcfn_fused3_kernel<128, 1, 6, 0, true, true>), dim3(grid), dim3(384), 0, stream, a);
      else hipLaunchKernelGGL((gcfn_fused3_kernel<64, 1, 6, 0, true, true>), dim3(grid), dim3(384), 0, stream, a);
    } else if (a.train) {
      if (F == 128) hipLaunchKernelGGL((gcfn_fused3_kernel<128, 1, 6, 0, true>), dim3(grid), dim3(384), 0, stream, a);
      else hipLaunchKernelGGL((gcfn_fused3_kernel<64, 1, 6, 0, true>), dim3(grid), dim3(384), 0, stream, a);
    } else if (F == 128 && hs_tiles(a.M) > 0) {
      // one tile per CU at most: the hidden-split form (gcfn_hs_kernel) with the smallest tile that fits the launch on the chip
      const int mt = hs_tiles(a.M), nt = (a.M + 16 * mt - 3) / (16 * mt - 2);
      if (mt == 2) hipLaunchKernelGGL((gcfn_hs_kernel<2>), dim3(nt), dim3(256), 0, stream, a);
      else if (mt == 3) hipLaunchKernelGGL((gcfn_hs_kernel<3>), dim3(nt), dim3(256), 0, stream, a);
      else hipLaunchKernelGGL((gcfn_hs_kernel<4>), dim3(nt), dim3(256), 0, stream, a);
    } else if (F == 128) {
      const int lat = ntiles <= lat_max_tiles() ? lat_ring() : 0;
      const int nt4 = (a.M + 4 * 14 - 1) / (4 * 14);
      if (lat && nt4 <= lat_max_tiles() && lat_nw() != 6)
        hipLaunchKernelGGL((gcfn_fused3_kernel<128, 1, 4, 0, false, false, 3>), dim3(nt4), dim3(256), 0, stream, a);
      else if (lat) hipLaunchKernelGGL((gcfn_fused3_kernel<128, 1, 6, 0, false, false, 3>), dim3(grid), dim3(384), 0, stream, a);
      else hipLaunchKernelGGL((gcfn_fused3_kernel<128, 1, 6>), dim3(grid), dim3(384), 0, stream, a);
    } else if (F == 64) {
      hipLaunchKernelGGL((gcfn_fused3_kernel<64, 1, 6>), dim3(grid), dim3(384), 0, stream, a);
    } else {
      return SEPR_EINVAL;
    }
  } else {
    {
      constexpr int tile_rows = (SEPR_GF3_XCH && GF3_MT == 2 && SEPR_GF3_UPFIRST) ? GF3_NW * 16 * GF3_MT - 2 : GF3_NW * (16 * GF3_MT - 2);
      const int ntiles = (a.M + tile_rows - 1) / tile_rows;
      const int cap = persistent_grid();
      const int grid = ntiles < cap ? ntiles : cap;
      static const int big_ring = [] {   // EXPERIMENT (round 6): the ring form for LARGE launches - one 8-wave workgroup per CU, 30-frame waves
        const char* e = getenv("SEPR_GF_BIG_RING");
        return e && e[0] ? atoi(e) : 0;
      }();
      if (big_ring >= 2 && !a.train && F == 128) {
        // EXPERIMENT (round 6): the one-wave-per-SIMD regime at F = 128 - four 64-frame waves (MT = 4, 512 registers), one workgroup per CU;
        // 2 = two-barrier form with the seam exchange (254-frame tiles), 3 = ring form without it (248-frame tiles)
        const int cus = lat_max_tiles();
        if (big_ring == 2) {
          const int nt = (a.M + 253) / 254;
          hipLaunchKernelGGL((gcfn_fused3_kernel<128, 4, 4>), dim3(nt < cus ? nt : cus), dim3(256), 0, stream, a);
        } else {
          const int nt = (a.M + 247) / 248;
          hipLaunchKernelGGL((gcfn_fused3_kernel<128, 4, 4, 0, false, false, 3>), dim3(nt < cus ? nt : cus), dim3(256), 0, stream, a);
        }
      } else
      if (big_ring && !a.train && F == 128) {
        const int nt8 = (a.M + 8 * 30 - 1) / (8 * 30);
        const int cus = lat_max_tiles();
        hipLaunchKernelGGL((gcfn_fused3_kernel<128, 2, 8, 0, false, false, 3>), dim3(nt8 < cus ? nt8 : cus), dim3(512), 0, stream, a);
      } else
      if (a.train && a.planes == 1) {
        if (F == 128) hipLaunchKernelGGL((gcfn_fused3_kernel<128, GF3_MT, GF3_NW, 0, true, true>), dim3(grid), dim3(64 * GF3_NW), 0, stream, a);
        else hipLaunchKernelGGL((gcfn_fused3_kernel<64, GF3_MT, GF3_NW, 0, true, true>), dim3(grid), dim3(64 * GF3_NW), 0, stream, a);
      } else if (a.train) {
        if (F == 128) hipLaunchKernelGGL((gcfn_fused3_kernel<128, GF3_MT, GF3_NW, 0, true>), dim3(grid), dim3(64 * GF3_NW), 0, stream, a);
        else hipLaunchKernelGGL((gcfn_fused3_kernel<64, GF3_MT, GF3_NW, 0, true>), dim3(grid), dim3(64 * GF3_NW), 0, stream, a);
      } else if (F == 128) {
        hipLaunchKernelGGL((gcfn_fused3_kernel<128, GF3_MT, GF3_NW>), dim3(grid), dim3(64 * GF3_NW), 0, stream, a);
      } else if (F == 64) {
        hipLaunchKernelGGL((gcfn_fused3_kernel<64, GF3_MT, GF3_NW>), dim3(grid), dim3(64 * GF3_NW), 0, stream, a);
      } else {
        return SEPR_EINVAL;
      }
    }
  }
  // algorithmic FLOPs of the block: both projections + the depthwise conv
  if (timed) prof_end(slot, (double)a.M * (2.0 * F * 6 * F + 2.0 * 3 * 6 * F + 2.0 * 3 * F * F), stream);
  SEPR_CHECK_LAUNCH("gcfn_fused_kernel");
  return SEPR_OK;
}

}  // namespace sepr
