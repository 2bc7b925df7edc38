// Fused middle of the GCFN backward (reference modules/network.py:46-66 under torch.autograd).
//
// The fused train forward (sepr_gcfn_fused.hip, TRAIN instantiation) keeps nothing of a GCFN block but the LayerNorm
// statistics of its input rows: neither the [rows, 6F] hidden tensor h1 nor the [rows, 3F] gated tensor ever reach HBM.
// This kernel is the backward's counterpart.  Per tile of 60 frames x 64 hidden (value, gate) channel pairs it
//   1. recomputes  h1 = LN(x) . (W1 gamma)^T + b1'                      (MFMA, K = F; the forward's up-projection)
//   2. computes    dgd = dropout1(dy) . (ls W2)                          (MFMA, K = F; input gradient of net2.2 + LayerScale)
//   3. from an LDS tile: depthwise k=3 conv, GLU, the gated-tensor dropout, the GLU backward and the conv's transpose:
//        c = b + w0 h[t-1] + w1 h[t] + w2 h[t+1],  gated = c_v sigmoid(c_g),  g = drop0(gated)      -> g   [rows, 3F]
//        dc_v = drop0'(dgd) sig,  dc_g = drop0'(dgd) c_v sig (1 - sig)
//        dh1[t] = w0 dc[t+1] + w1 dc[t] + w2 dc[t-1]                                                -> dh1 [rows, 6F]
//        dw_k += sum_t dc[t] h[t+k-1],  db += sum_t dc[t]      (per-tile partials, fixed-order reduction afterwards)
// and - when the output dropout is live - writes dropout1(dy) once for the weight-gradient contraction of net2.2.
// It replaces the dropout pass over dy, the 3F-wide input-gradient projection and the GLU / conv middle kernel of the
// unfused backward (and their [rows, 3F] / [rows, 6F] round trips: 9 KB per row at F = 128 instead of 17 KB), and makes
// the 9F + 2 floats per row the unfused forward had to save unnecessary (1.1 GB per 4 s utterance for Base).
//
// Structure: the generic projection core's (sepr_gemm_x3.h) - activations fp32 -> registers -> normalise / mask -> bf16
// planes in LDS (double buffered), packed weight fragments L2 -> VGPR, wave tile = 64 rows x (16 value + 16 gate + 16
// gradient columns), four waves side by side - with two accumulation phases and the row-window epilogue above.  Rows
// m0-2 .. m0+61 are computed per tile (two halo frames on each side: dh1[t] needs dc[t+-1], which needs h1[t+-2]).
// PLANES = 3: bf16x3 split arithmetic; PLANES = 1: plain bf16 operands (the "bf16" training precision).
#include "sepr_train.h"
#include <stdlib.h>

namespace sepr {

typedef __bf16 gb_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gb_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gb_store4(void* base, long long idx, const float4 v, const bool as16) {
  if (as16) {
    gb_bf16x4 h = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    *reinterpret_cast<gb_bf16x4*>(static_cast<__bf16*>(base) + idx) = h;
  } else {
    st4(static_cast<float*>(base) + idx, v);
  }
}

typedef unsigned gb_u32x4 __attribute__((ext_vector_type(4)));
// PL form: the weight fragments are loaded by inline asm, i.e. hidden from hipcc's vmcnt bookkeeping - beside LDS-DMA in flight hipcc waits
// vmcnt(0) for ANY ordinary VGPR-destination load, which would drain the whole slab pipeline at the first MFMA (guide: "three .s-level
// traps" (b)).  The waits below are the counted ones and name the destination registers ("+v") so that no consumer can move above them.
__device__ __forceinline__ void gb_asm_load(gb_u32x4& d, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void gb_wait_vm4(gb_u32x4& a, gb_u32x4& b, gb_u32x4& c, gb_u32x4& d) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void gb_wait_vm2(gb_u32x4& a, gb_u32x4& b) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}

#ifndef SEPR_GB_PL_WGS
#define SEPR_GB_PL_WGS 3   // workgroups per CU the plane-staged (PL) middle kernel is compiled and launched for
#endif
#ifndef SEPR_GB_REGEPI
#define SEPR_GB_REGEPI 0   // 0 (product): the row-window epilogue of gcfn_bwd_mid_kernel through LDS tiles (rounds 2-5); 1: in registers (round 6, second session:
                           // built, bit-identical outputs, measured equal at two workgroups per CU and slower at three - hipcc spills 50-60 registers into
                           // the 168 of the three-workgroup regime; tools/variants.mk gbepi2w / gbepi3w, profiles/r06_gcfn_bwd_regepi.txt)
#endif
#ifndef SEPR_GB_ONEBAR
#define SEPR_GB_ONEBAR 0   // PL form: 0 = a counted wait + barrier in front of every slab's MFMAs (slab q multiplies while slabs q+1.. land); 1 = ONE wait for all
                           // the tile's slabs and one barrier in front of step 0, steps >= 1 wait for their own weight fragments only (round 6, last session:
                           // tools/variants.mk gbonebar)
#endif
#ifndef SEPR_GB_TOPWAIT
#define SEPR_GB_TOPWAIT 0
#endif
#ifndef SEPR_GB_CONSTLDS
#define SEPR_GB_CONSTLDS 1  // PL form, persistent launches in which a workgroup keeps ONE column block (acc_part): the block's depthwise taps / biases (8 float4 per
                           // column quad) and up-projection biases (2 x 64) are parked in 2.5 KB of LDS once per workgroup instead of being fetched from L2 by every
                           // tile right in front of their use (two exposed L2 latencies per tile); 53 760 B per workgroup, still three per CU
#endif
#ifndef SEPR_GB_REDERIVE
#define SEPR_GB_REDERIVE 3   // product: bit 1 the plane-staged form, bit 2 the register-staged forms too (0 = rounds 4-6: 168 registers + 10 spilled / 256 + 9-13; profiles/r06_gcfn_bwd_waits.txt)
#endif
#ifndef SEPR_GB_SLIDE
#define SEPR_GB_SLIDE 2     // (bit mask: 1 = pass A, 2 = pass B) LDS epilogue, interior tiles: a thread's 4 consecutive rows share their conv windows - pass A reads each h1 row of its 6-row window once
                           // (12 ds_read_b128 instead of 24), pass B takes the neighbour rows of dc from its own registers (4 reads instead of 16); bit-identical
#endif
constexpr int GB_CONST_B = SEPR_GB_CONSTLDS ? (8 * 16 + 32) * 16 : 0;
// slab row s = mt * 16 + fi of the middle kernel holds frame 4 fi + mt of the tile (REGEPI) - so that frame neighbours are a lane's own
// accumulator tiles or one DPP row shift away - or frame s (LDS epilogue)
__device__ __forceinline__ int gb_frame(int s) { return SEPR_GB_REGEPI ? 4 * (s & 15) + (s >> 4) : s; }
// DPP row shifts (16-lane rows): lane i <- lane i - 1 / i + 1; the lane without a source gets 0
__device__ __forceinline__ float gb_shr1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));
}
__device__ __forceinline__ float gb_shl1(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, false));
}
__device__ __forceinline__ float2 ld2g(const float* p) { return *reinterpret_cast<const float2*>(p); }
// two fp32 -> one bf16x2 word (round to nearest even, the conversion gb_store4 applies): low half = a
__device__ __forceinline__ unsigned gb_pack2(float a, float b) {
  typedef __bf16 gb_bf16x2 __attribute__((ext_vector_type(2)));
  const gb_bf16x2 h = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ float4 gb_shr4(float4 v) { return make_float4(gb_shr1(v.x), gb_shr1(v.y), gb_shr1(v.z), gb_shr1(v.w)); }
__device__ __forceinline__ float4 gb_shl4(float4 v) { return make_float4(gb_shl1(v.x), gb_shl1(v.y), gb_shl1(v.z), gb_shl1(v.w)); }
// inclusive scan over the 16 lanes of a DPP row (row_shr 1, 2, 4, 8; lanes shifted in from outside the row add 0): lane 15 holds the row's sum
__device__ __forceinline__ float gb_row_total(float v) {
#pragma clang fp contract(off)
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, false));
  return v;
}

namespace {
template <bool V>
struct gb_bool { static constexpr bool value = V; };
// Compile-time replay of the PL form's VMEM issue order (gcfn_bwd_mid_kernel below): per wave  D0 W0 D1 W1 D2 .. D(NQ-1)  in the prologue,
// then W(s+2) right after the MFMAs of step s.  D = `dma` LDS-DMA copies of a slab, W = 4 fragment loads for an up-projection step
// (q < nsl), 2 for a dgd step.  Returns how many VMEM operations have been issued AFTER the last one step q depends on (its slab D_q and
// its fragments W_q) when step q waits - vmcnt retires in order, so that is exactly the s_waitcnt vmcnt(N) of the step.  The hand-counted
// constants C0..C2 in the kernel are static_assert-ed against this replay: a change of DMA_PER_WAVE, NSL or the issue order that is not
// carried into the waits stops the build instead of becoming a data race.
constexpr int gb_pl_outstanding(int q, int nsl, int dma) {
  const int nq = 2 * nsl;
  int pos = 0, last_d[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < nq; ++i) {
    pos += dma;
    last_d[i] = pos;
    if (i < 2) {
      pos += i < nsl ? 4 : 2;
      last_w[i] = pos;
    }
  }
  for (int s = 0; s < q; ++s)
    if (s + 2 < nq) {
      pos += (s + 2) < nsl ? 4 : 2;
      last_w[s + 2] = pos;
    }
  const int need = last_d[q] > last_w[q] ? last_d[q] : last_w[q];
  return pos - need;
}
constexpr int GB_BM = 64;            // rows per tile (incl. halo)
constexpr int GB_OUT = GB_BM - 4;    // rows a tile outputs
constexpr int GB_BKS = 64;           // K extent of one LDS slab
#ifndef SEPR_GB_LDK_PAD
#define SEPR_GB_LDK_PAD 16
#endif
constexpr int GB_LDK = GB_BKS + SEPR_GB_LDK_PAD;  // bf16 per LDS row (160 B; 144 B with pad 8)
constexpr int GB_HS = 128 + 4;       // fp32 row stride of the h1 / dc tile (64 value + 64 gate columns)
constexpr int GB_DS = 64 + 4;        // fp32 row stride of the dgd tile
constexpr int GB_RS = 36;            // fp32 row stride of the depthwise-partial scratch [4 waves][16 column quads][32 sums] that reuses the dgd tile
static_assert(4 * 16 * GB_RS <= GB_BM * GB_DS, "the reduction scratch must fit the dgd tile");
constexpr int GB_THREADS = 256;

struct GcfnBwdArgs {
  const float* x;       // [M][F] block input
  const float* stats;   // [M][2] (mean, rstd) of the input rows, from the forward
  const float* dy;      // [M][F] gradient w.r.t. the block output
  int M, T, F;
  const void* w1p;      // pack_x3 fragments of W1 * gamma [6F][F]        (sepr_gcfn_tw.up.wp)
  const float* b1;      // [6F] net1.1.bias + W1 . beta                   (sepr_gcfn_tw.up.b)
  const void* w2tp;     // pack_x3 fragments of (ls * W2)^T [3F][F]       (sepr_gcfn_tw.down_t.wp)
  const float* dw_w;    // [3][6F] depthwise taps, tap-major
  const float* dw_b;    // [6F]
  void* g;              // out [M][3F] gated tensor after its dropout (the input net2.2 multiplied); fp32, or bf16 when out16
  void* dh1;            // out [M][6F] gradient w.r.t. the up-projection's output; fp32 or bf16
  int out16;            // 1: g / dh1 are stored as bf16 - the plain-bf16 precision rounds them to bf16 as MFMA operands of the two
                        // weight-gradient contractions and of the input-gradient projection anyway, so the stored form loses nothing
                        // and halves 4.6 of the block's ~9 KB of HBM traffic per row
  float* dyq;           // out [M][F] dropout1(dy), or null (no output dropout: the contraction reads dy itself)
  float* part;          // out [MB][3F][8] depthwise weight / bias gradient partials (w0 w1 w2 b of the value, then of the gate)
  unsigned drop_thr;    // 16-bit keep threshold of both sites (0 = no dropout)
  float drop_scale;     // 1 / (1 - p_eff)
  unsigned long long seed;
  const unsigned long long* salt;
  // PL instantiation (plain-bf16 precision, round 4): both MFMA operands arrive as bf16 rows [M][F] - xh16 written by the fused forward
  // (sepr_gcfn_fused.hip xhat16), dy16 = bf16(dropout1(dy)) written by gcfn_dyplane_kernel below - and are staged by LDS-DMA
  const unsigned short* xh16;
  const unsigned short* dy16;
  // 1: the launch is a persistent grid in which every workgroup keeps ONE column block (grid a multiple of 8 NB, more tiles than workgroups): the depthwise
  // partials of a workgroup's tiles are summed in two registers per thread and leave ONCE, as row (first tile's mb) of part - grid / NB partial rows instead of
  // one per row tile (4 267 at 256 000 rows), no per-tile store, a 30x smaller reduction behind the kernel (round 6)
  int acc_part;
};

// NSL = F / 64 slabs per operand.  All 2 * NSL activation slabs of a tile (x, then dy) are requested TOGETHER, one tile ahead:
// the loads of tile t+1 are issued when tile t's accumulators have been staged and fly under its whole row-window epilogue.
// (First version: one slab in flight at a time, requested during the previous slab's ~800-cycle MFMA phase - every slab paid
//  an exposed L2 / HBM latency and a 64-row tile took 16 us.  Also measured, round 3: giving up the tile-ahead prefetch for a third
//  workgroup per CU - 168 VGPRs, 46 dwords spilled once per tile - is SLOWER, 337 us against 283 us per launch at batch 16: the tile
//  is not latency-bound but the sum of ~1 650 VALU instructions, ~250 LDS instructions and 144 MFMAs per wave between 9 barriers.)
// PL (round 4; ONE only): the 2 * NSL slabs of a tile are bf16 already and go global -> LDS by LDS-DMA (global_load_lds, 16 B per
// lane, per-lane global address, linear LDS image = [64 rows][160 B]: 10 lane slots per row, the last two masked off as the row pad).
// No staging VALU at all (39 % of the VALU stream of the register-staged form, profiles/r03_v5_pmc_train_kernels.txt) and no 64-register
// tile-ahead prefetch: the kernel fits three workgroups per CU (51 KB of LDS each), which cover the DMA latency a tile now pays at its
// start (the slab buffers alias the epilogue tiles, so the next tile's slabs cannot fly under the epilogue).
template <int PLANES, int NSL, bool PL = false>
__global__ __launch_bounds__(GB_THREADS, PL ? SEPR_GB_PL_WGS : 2) void gcfn_bwd_mid_kernel(const GcfnBwdArgs a) {
  constexpr bool ONE = PLANES == 1;
  static_assert(!PL || ONE, "plane staging exists for the plain-bf16 arithmetic");
  constexpr int NP = ONE ? 1 : 2;                              // bf16 planes per LDS buffer
  constexpr int PLANE_E = GB_BM * GB_LDK;                      // elements of one plane
  constexpr int DMA_PER_WAVE = GB_BM * 128 / 1024 / 4;                  // PL: LDS-DMA instructions per wave and slab (2)
  constexpr size_t SLAB_B = PL ? (size_t)2 * NSL * GB_BM * 128 : sizeof(unsigned short) * 2 * NP * PLANE_E;
  // (the LDS epilogue's tiles, or the register epilogue's 6 parked 16-byte chunks per thread, alias the slab buffers)
  constexpr size_t TILE_B = SEPR_GB_REGEPI ? (size_t)6 * GB_THREADS * 16 : sizeof(float) * GB_BM * (GB_HS + GB_DS);
  constexpr size_t MAIN_B = SLAB_B > TILE_B ? SLAB_B : TILE_B;
  __shared__ __attribute__((aligned(16))) unsigned char smem[MAIN_B + (PL ? GB_CONST_B : 0)];
  [[maybe_unused]] float* const Cs = reinterpret_cast<float*>(smem + MAIN_B);   // CONSTLDS: [8 taps / biases][16 column quads][4], then [value | gate][16 quads][4] of b1
  unsigned short* const slab = reinterpret_cast<unsigned short*>(smem);
  [[maybe_unused]] float* const Hs = reinterpret_cast<float*>(smem);            // [64][GB_HS]: h1 (+ b1), later dc   (aliases the slab buffers)
  [[maybe_unused]] float* const Ds = Hs + GB_BM * GB_HS;                        // [64][GB_DS]: dgd, later the reduction scratch

#if SEPR_GB_REDERIVE
  int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;      // re-derived at the top of every tile (see SEPR_GB_REDERIVE)
  int fi = lane & 15, fg = lane >> 4;
#else
  const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  const int fi = lane & 15, fg = lane >> 4;
#endif
  const int F = a.F, C3 = 3 * F;
  const int NB = C3 / 64;
  const int MB = (a.M + GB_OUT - 1) / GB_OUT;
  const int ntiles = ((MB + 7) / 8) * 8 * NB;
  constexpr int nsl = NSL;
  const int kst = F / 32;
  const uint4* const W1 = static_cast<const uint4*>(a.w1p);
  const uint4* const W2 = static_cast<const uint4*>(a.w2tp);
  const bool drop = a.drop_thr > 0u;
  DropKey dk0 = {0u, 0u}, dk1 = {0u, 0u};
  if (drop) {
    dk0 = sepr_drop_key(a.seed, a.salt, 0u);
    dk1 = sepr_drop_key(a.seed, a.salt, 1u);
  }
  // staging role: one row per 4 threads, 16 consecutive k per thread and slab
  const int srow = tid >> 2, kq = (tid & 3) * 16;

  // tile walk: same row tile -> same XCD for all its column blocks (block b runs on XCD b % 8): x / dy rows are fetched once per L2
  auto decode = [&](int tile, int& mb, int& nb) -> bool {
    const int u = tile >> 3;
    mb = (u / NB) * 8 + (tile & 7);
    nb = u % NB;
    return mb < MB;
  };
  float4 ra[PL ? 1 : 2 * NSL][4];                              // this tile's slabs (x: 0..NSL-1, dy: NSL..2NSL-1), 16 k per thread each
  float2 rst = make_float2(0.f, 0.f);                          // (mean, rstd) of the staged row
  // PL: slab q of row tile mb_ -> slab buffer q & 1.  Rows outside [0, M) read a clamped (finite) row: everything they feed is either
  // masked by the sequence-end flags of the conv or belongs to rows the epilogue skips.
  auto dma_slab = [&](int q, int mb_) {
    if constexpr (PL) {
      // (wave-uniform base + 32-bit per-lane byte offset: SGPR-base addressing, no 64-bit per-lane pointers to keep alive)
      const char* plane = reinterpret_cast<const char*>(q < NSL ? a.xh16 : a.dy16) + (q < NSL ? q : q - NSL) * (GB_BKS * 2);
      // Every slab of the tile has its own 8 KB buffer (all aliasing the epilogue tiles): [64 rows][128 B], the 16-byte chunk c of row r
      // stored at chunk position c ^ (r & 7).  The XOR swizzle makes the MFMA fragment reads (16-lane groups of ds_read_b128: rows
      // fi, chunks {2j, 2j+1}) hit 16 distinct 16-byte slots of the 256-byte bank row WITHOUT row padding - so the image is exactly 8
      // LDS-DMA instructions, two per wave, no masked lanes and no divergent issue code (hipcc counts vmcnt across straight-line code
      // only; anything conditional in here turns every later wait into vmcnt(0)).
      unsigned char* dst = smem + (size_t)q * (GB_BM * 128) + wn * (DMA_PER_WAVE * 1024);
#pragma unroll
      for (int i = 0; i < DMA_PER_WAVE; ++i) {
        const int pos = (wn * DMA_PER_WAVE + i) * 64 + lane, row = pos >> 3, c = (pos & 7) ^ (row & 7);
        int msn = mb_ * GB_OUT - 2 + gb_frame(row);
        msn = msn < 0 ? 0 : (msn > a.M - 1 ? a.M - 1 : msn);
        const unsigned off = (unsigned)msn * (unsigned)(2 * F) + (unsigned)(c * 16);     // M * F * 2 < 2^32 (checked by the launcher)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(plane + off),
                                         (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
      }
    }
  };
  auto load_tile = [&](int mb_) {
    if constexpr (PL) return;
    const int msn = mb_ * GB_OUT - 2 + gb_frame(srow);
    const long long row = (msn >= 0 && msn < a.M) ? msn : 0;
    const float* px = a.x + row * F + kq;
    const float* pd = a.dy + row * F + kq;
#pragma unroll
    for (int q = 0; q < NSL; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ra[q][j] = ld4(px + q * GB_BKS + 4 * j);
        ra[NSL + q][j] = ld4(pd + q * GB_BKS + 4 * j);
      }
    rst = *reinterpret_cast<const float2*>(a.stats + 2 * row);
  };
  int tile = blockIdx.x, mb = 0, nb = 0;
  while (tile < ntiles && !decode(tile, mb, nb)) tile += gridDim.x;
  if (tile < ntiles) load_tile(mb);
  [[maybe_unused]] const bool cl_on = SEPR_GB_CONSTLDS && PL && a.acc_part != 0;    // (acc_part: nb is the same for every tile of this workgroup)
  if constexpr (SEPR_GB_CONSTLDS && PL) {
    if (cl_on && tile < ntiles) {
      const int C3_ = 3 * a.F, C6_ = 6 * a.F;
      if (tid < 128) {
        const int k = tid >> 4, q = tid & 15;                    // k: wv0 wv1 wv2 wg0 wg1 wg2 cbv cbg
        const int hc_ = 64 * nb + 4 * q;
        const float* src = k < 6 ? a.dw_w + (k % 3) * C6_ + (k >= 3 ? C3_ : 0) + hc_ : a.dw_b + (k == 7 ? C3_ : 0) + hc_;
        st4(Cs + (k * 16 + q) * 4, ld4(src));
      } else if (tid < 160) {
        const int j = tid - 128;                                 // 0..15 value quads, 16..31 gate quads
        st4(Cs + (128 + j) * 4, ld4(a.b1 + (j >= 16 ? C3_ : 0) + 64 * nb + 4 * (j & 15)));
      }
      // (visible to every wave behind the first tile's barriers)
    }
  }
#if SEPR_GB_TOPWAIT == 2
  if constexpr (PL) __builtin_amdgcn_s_waitcnt(0x0F70);       // (experiment: the wait once, in front of the tile loop)
#endif
  [[maybe_unused]] float wacc[2] = {0.f, 0.f};                 // acc_part: this thread's two of the workgroup's 512 partial sums
  [[maybe_unused]] const int mb_first = mb, nb_first = nb;
  [[maybe_unused]] const bool any_tile = tile < ntiles;
  while (tile < ntiles) {
#if SEPR_GB_REDERIVE
    if constexpr (PL || (SEPR_GB_REDERIVE & 2) != 0) {         // an opaque copy of the thread index: everything derived from it (LDS addresses, lane roles) is recomputed
      asm volatile("" : "+v"(tid));                            // per tile instead of being hoisted out of the loop and kept (or spilled) across all of its phases
      lane = tid & 63; wn = tid >> 6; fi = lane & 15; fg = lane >> 4;
    }
#endif
    const int m0 = mb * GB_OUT;
    const int sfr = gb_frame(srow);                            // frame (of the tile) in slab row srow
    const int ms = m0 - 2 + sfr;                               // the row this thread stages
    const bool svalid = ms >= 0 && ms < a.M;
    const float mean = rst.x, rstd = rst.y;
    auto store_slab = [&](int q) {
#pragma clang fp contract(off)
      if constexpr (PL) return;
      float v[16];
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[4 * j] = ra[q][j].x; v[4 * j + 1] = ra[q][j].y; v[4 * j + 2] = ra[q][j].z; v[4 * j + 3] = ra[q][j].w; }
      if (q < nsl) {
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = svalid ? (v[e] - mean) * rstd : 0.f;
      } else {
        const int f0c = (q - nsl) * GB_BKS + kq;                 // first output channel of this thread's 16
        if (drop) {                                            // network.py:57: the output dropout's mask, keep scale folded in
#pragma unroll
          for (int e = 0; e < 16; e += 2) {
            const unsigned d = sepr_drop_word(dk1, (unsigned)ms, (unsigned)((f0c + e) >> 1));
            v[e] = (d & 0xffffu) >= a.drop_thr ? v[e] * a.drop_scale : 0.f;
            v[e + 1] = (d >> 16) >= a.drop_thr ? v[e + 1] * a.drop_scale : 0.f;
          }
          // one column block writes dropout1(dy) for the weight-gradient contraction of net2.2 (output rows only)
          if (a.dyq && nb == 0 && sfr >= 2 && sfr < 2 + GB_OUT && svalid) {
            float* o = a.dyq + (long long)ms * F + f0c;
#pragma unroll
            for (int j = 0; j < 4; ++j) st4(o + 4 * j, make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]));
          }
        }
        if (!svalid) {
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] = 0.f;
        }
      }
      unsigned short* hi = slab + ((q & 1) * NP + 0) * PLANE_E + srow * GB_LDK + kq;
      unsigned short* lo = slab + ((q & 1) * NP + (NP - 1)) * PLANE_E + srow * GB_LDK + kq;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        gb_bf16x8 h, l;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const __bf16 xh = (__bf16)v[8 * j + e];
          h[e] = xh;
          if (!ONE) l[e] = (__bf16)(v[8 * j + e] - (float)xh);
        }
        *reinterpret_cast<gb_bf16x8*>(hi + 8 * j) = h;
        if (!ONE) *reinterpret_cast<gb_bf16x8*>(lo + 8 * j) = l;
      }
    };
    // weight fragments of one 16-row tile at K step ks: (hi, lo) planes, one coalesced 1 KiB read each
    auto load_w = [&](const uint4* W, int t16, int ks, gb_u32x4 (&w)[2]) {
      const uint4* p = W + ((long long)(t16 * kst + ks) * 2) * 64 + lane;
      if constexpr (PL) {
        gb_asm_load(w[0], p);
      } else {
        w[0] = *reinterpret_cast<const gb_u32x4*>(p);
        if (!ONE) w[1] = *reinterpret_cast<const gb_u32x4*>(p + 64);
      }
    };

    f32x4 hv[4], hg[4], dd[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      hv[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      hg[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dd[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int tv = 4 * nb + wn, tg = C3 / 16 + 4 * nb + wn;     // this wave's value / gate tile of W1, tv also its W2^T tile

#if SEPR_GB_TOPWAIT == 1
    if constexpr (PL) __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0), a wait hipcc's own bookkeeping sees (see SEPR_GB_TOPWAIT above)
#endif
    __syncthreads();   // the previous tile's epilogue is done with the LDS tiles that alias the slab buffers
    gb_u32x4 wset[2][2][2][2];                                 // [q & 1][a | b][K step of the slab][plane]; a: value (or W2^T), b: gate
    auto load_wq = [&](int q, gb_u32x4 (&ws_)[2][2][2]) {
      const bool up_ = q < nsl;
      const int s_ = up_ ? q : q - nsl;
      load_w(up_ ? W1 : W2, tv, 2 * s_, ws_[0][0]);
      load_w(up_ ? W1 : W2, tv, 2 * s_ + 1, ws_[0][1]);
      if (up_) {
        load_w(W1, tg, 2 * s_, ws_[1][0]);
        load_w(W1, tg, 2 * s_ + 1, ws_[1][1]);
      }
    };
    // PL: ALL slabs of the tile go in flight at once (one exposed DMA latency per tile instead of one per slab).  Issue order per wave:
    //   D0 W0 D1 W1 D2 .. D(2 NSL - 1), then W(q + 2) right after the MFMAs of step q - so "slab q and the weights of step q have landed"
    // is a compile-time vmcnt (DMA_PER_WAVE copies per slab, 4 fragment loads per up-projection step, 2 per dgd step; vmcnt retires in order)
    if constexpr (PL) {
#pragma unroll
      for (int q = 0; q < 2 * nsl; ++q) {
        dma_slab(q, mb);
        if (q < 2) load_wq(q, wset[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < 2 * nsl; ++q) {
      const bool up = q < nsl;
      gb_u32x4 (&wa)[2][2] = wset[q & 1][0];
      gb_u32x4 (&wb)[2][2] = wset[q & 1][1];
      if constexpr (PL) {
        constexpr int NQ = 2 * NSL;
        constexpr int C0 = DMA_PER_WAVE * (NQ - 1) + (NQ > 1 ? (1 < NSL ? 4 : 2) : 0);      // behind W0: D1 W1 D2 ..
        constexpr int C1 = DMA_PER_WAVE * (NQ - 2) + (NQ > 2 ? (2 < NSL ? 4 : 2) : 0);      // behind W1: D2 .. and W2 (issued after step 0)
        constexpr int C2 = (NQ > 3) ? (3 < NSL ? 4 : 2) : 0;                                // behind W2: W3 (issued after step 1)
        static_assert(NQ <= 8 && C0 == gb_pl_outstanding(0, NSL, DMA_PER_WAVE), "vmcnt of step 0 does not match the issue order");
        static_assert(NQ < 2 || C1 == gb_pl_outstanding(1, NSL, DMA_PER_WAVE), "vmcnt of step 1 does not match the issue order");
        static_assert(NQ < 3 || C2 == gb_pl_outstanding(2, NSL, DMA_PER_WAVE), "vmcnt of step 2 does not match the issue order");
        static_assert(NQ < 4 || gb_pl_outstanding(3, NSL, DMA_PER_WAVE) == 0, "steps >= 3 wait vmcnt(0): nothing may be issued behind W3");
        if constexpr (SEPR_GB_ONEBAR) {
          // every copy and both prologue fragment sets were issued before step 0: one vmcnt(0) covers D0 .. D(NQ-1), W0, W1.  Steps q >= 2 wait for
          // W(q) with W(q+1) - issued behind step q-1's MFMAs - still in flight.
          if (q == 0) {
            gb_wait_vm4<0>(wset[0][0][0][0], wset[0][0][1][0], wset[0][1][0][0], wset[0][1][1][0]);
            if constexpr (NQ > 1) {
              if constexpr (1 < NSL) gb_wait_vm4<0>(wset[1][0][0][0], wset[1][0][1][0], wset[1][1][0][0], wset[1][1][1][0]);
              else gb_wait_vm2<0>(wset[1][0][0][0], wset[1][0][1][0]);
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
          } else if (q >= 2) {
            constexpr int NXT4 = 4, NXT2 = 2;
            const bool more = q + 1 < NQ, more_up = q + 1 < NSL;
            if (up) {
              if (more && more_up) gb_wait_vm4<NXT4>(wa[0][0], wa[1][0], wb[0][0], wb[1][0]);
              else if (more) gb_wait_vm4<NXT2>(wa[0][0], wa[1][0], wb[0][0], wb[1][0]);
              else gb_wait_vm4<0>(wa[0][0], wa[1][0], wb[0][0], wb[1][0]);
            } else {
              if (more && more_up) gb_wait_vm2<NXT4>(wa[0][0], wa[1][0]);
              else if (more) gb_wait_vm2<NXT2>(wa[0][0], wa[1][0]);
              else gb_wait_vm2<0>(wa[0][0], wa[1][0]);
            }
          }
        } else {
        if (up) {
          if (q == 0) gb_wait_vm4<C0>(wa[0][0], wa[1][0], wb[0][0], wb[1][0]);
          else if (q == 1) gb_wait_vm4<C1>(wa[0][0], wa[1][0], wb[0][0], wb[1][0]);
          else if (q == 2) gb_wait_vm4<C2>(wa[0][0], wa[1][0], wb[0][0], wb[1][0]);
          else gb_wait_vm4<0>(wa[0][0], wa[1][0], wb[0][0], wb[1][0]);
        } else {
          if (q == 0) gb_wait_vm2<C0>(wa[0][0], wa[1][0]);
          else if (q == 1) gb_wait_vm2<C1>(wa[0][0], wa[1][0]);
          else if (q == 2) gb_wait_vm2<C2>(wa[0][0], wa[1][0]);
          else gb_wait_vm2<0>(wa[0][0], wa[1][0]);
        }
        // a bare s_barrier, not __syncthreads(): the fence of __syncthreads() makes hipcc drain vmcnt to 0 (the LDS-DMA of the later
        // slabs with it).  What the barrier must order is exactly what the counted wait above covers: this wave's share of slab q.
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        }
      } else {
        load_wq(q, wset[q & 1]);
        store_slab(q);
        __syncthreads();
      }
      const unsigned short* ph = PL ? slab + q * (GB_BM * 64) : slab + ((q & 1) * NP + 0) * PLANE_E;      // PL: own buffer per slab, swizzled rows
      const unsigned short* pl = slab + ((q & 1) * NP + (NP - 1)) * PLANE_E;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        gb_bf16x8 xh[4], xl[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const int off = PL ? (mt * 16 + fi) * 64 + (((kk * 4 + fg) ^ (fi & 7)) * 8) : (mt * 16 + fi) * GB_LDK + kk * 32 + 8 * fg;
          xh[mt] = *reinterpret_cast<const gb_bf16x8*>(ph + off);
          if (!ONE) xl[mt] = *reinterpret_cast<const gb_bf16x8*>(pl + off);
        }
        const gb_bf16x8 ah = *reinterpret_cast<const gb_bf16x8*>(&wa[kk][0]);
        const gb_bf16x8 al = *reinterpret_cast<const gb_bf16x8*>(&wa[kk][ONE ? 0 : 1]);
        if (up) {
          const gb_bf16x8 bh = *reinterpret_cast<const gb_bf16x8*>(&wb[kk][0]);
          const gb_bf16x8 bl = *reinterpret_cast<const gb_bf16x8*>(&wb[kk][ONE ? 0 : 1]);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            hv[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xh[mt], hv[mt], 0, 0, 0);
            hg[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, xh[mt], hg[mt], 0, 0, 0);
          }
          if constexpr (!ONE) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
              hv[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xl[mt], hv[mt], 0, 0, 0);
              hg[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh, xl[mt], hg[mt], 0, 0, 0);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
              hv[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, xh[mt], hv[mt], 0, 0, 0);
              hg[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl, xh[mt], hg[mt], 0, 0, 0);
            }
          }
        } else {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) dd[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xh[mt], dd[mt], 0, 0, 0);
          if constexpr (!ONE) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) dd[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, xl[mt], dd[mt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) dd[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, xh[mt], dd[mt], 0, 0, 0);
          }
        }
      }
      if constexpr (PL) {
        if (q + 2 < 2 * nsl) load_wq(q + 2, wset[q & 1]);      // this step's fragment registers are free again
      }
    }
#if SEPR_GB_REGEPI
    // ---- epilogue in registers (round 6) -------------------------------------------------------------------------------------------
    // The accumulators already hold what the row window needs: lane (fi, fg) of wave wn has, per slab-row tile mt, the 4 hidden channels
    // 64 nb + 16 wn + 4 fg + r of ONE frame - value, gate and dgd.  With slab row mt * 16 + fi holding frame 4 fi + mt (the copy's row map,
    // gb_frame), the previous / next frame of a lane's tile mt is its own tile mt -+ 1; across the 4-frame seams it is the neighbouring lane's
    // tile 3 / tile 0 - ONE DPP row shift, and the two lanes without a neighbour (fi = 0 / 15) are the two halo frames at the tile's ends whose
    // dc is never used.  So the h1 / dgd / dc tiles, their 5 barriers, ~260 KB of LDS traffic per tile and the 64 ds_bpermute of the partial
    // sums are gone (rounds 2-5 form: -DSEPR_GB_REGEPI=0, tools/variants.mk gbepi0); the depthwise partial sums reduce over the 16 frame lanes of
    // a DPP row (each wave owns its own 16 channel pairs: no cross-wave step).  Element arithmetic unchanged (same fmaf order as the LDS form).
    // Register diet (the plane-staged kernel has 168 per lane at three workgroups per CU): the 4 channels of a lane are walked as two PAIRS -
    // taps, dc values and partial sums of one pair live at a time (a pair is also one dropout word and one packed bf16x2 output register).
    // the next tile of this workgroup (register-staged forms: its activation slabs go in flight now, under the epilogue below)
    // The second channel pair's inputs wait in LDS - the slab buffers are dead once every wave has left the MFMA phase; each lane parks and
    // later re-reads ITS OWN 24 values (6 conflict-free 16-byte chunks, chunk j of thread t at (256 j + t) x 16 B): no exchange, no further barrier
    __syncthreads();
    {
      float4* const pk = reinterpret_cast<float4*>(smem) + tid;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) pk[256 * mt] = make_float4(hv[mt][2], hv[mt][3], hg[mt][2], hg[mt][3]);
      pk[256 * 4] = make_float4(dd[0][2], dd[0][3], dd[1][2], dd[1][3]);
      pk[256 * 5] = make_float4(dd[2][2], dd[2][3], dd[3][2], dd[3][3]);
    }
    int nxt = tile + gridDim.x, mb_n = 0, nb_n = 0;
    while (nxt < ntiles && !decode(nxt, mb_n, nb_n)) nxt += gridDim.x;
    if (nxt < ntiles) load_tile(mb_n);
    const int cl = wn * 16 + 4 * fg, hc = 64 * nb + cl;          // hidden value channel of accumulator element 0 (gate: C3 + hc)
    const int C6 = 2 * C3;
    const int lo_t = (m0 >= 1) ? (m0 - 1) % a.T : 0;
    const bool edge_tile = !(m0 >= 2 && lo_t >= 1 && lo_t + 62 <= a.T - 2 && m0 + GB_BM - 2 <= a.M);
    constexpr bool as16 = ONE;            // (the launcher stores g / dh1 as bf16 exactly for the plain-bf16 arithmetic and checks it: out16 == ONE)
    auto passes = [&](auto edge_c) {
#pragma clang fp contract(off)
      constexpr bool EDGE = decltype(edge_c)::value;
      // outputs of the lane's 4 frames x 4 channels, held until both pairs are done (8- / 16-byte stores): bf16x2 words (as16) or floats
      [[maybe_unused]] unsigned gP[4][2], ovP[4][2], ogP[4][2];
      [[maybe_unused]] float gF[4][4], ovF[4][4], ogF[4][4];
      float f0s[4], f2s[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        f0s[mt] = 1.f;
        f2s[mt] = 1.f;
        if constexpr (EDGE) {                                  // zero padding at the sequence ends (tiles that touch one: ~1 in 130)
          const int m = m0 - 2 + 4 * fi + mt, t = (m >= 0 ? m : 0) % a.T;
          f0s[mt] = t > 0 ? 1.f : 0.f;
          f2s[mt] = t < a.T - 1 ? 1.f : 0.f;
        }
      }
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {                          // channels hc + 2 pr, hc + 2 pr + 1
        asm volatile("" ::: "memory");                         // (keeps the second pair's tap loads behind the first pair's work: registers)
        __builtin_amdgcn_sched_barrier(0);
        const int hp = hc + 2 * pr;
        const float2 bv = ld2g(a.b1 + hp), bg = ld2g(a.b1 + C3 + hp);
        const float2 wv0 = ld2g(a.dw_w + hp), wv1 = ld2g(a.dw_w + C6 + hp), wv2 = ld2g(a.dw_w + 2 * C6 + hp);
        const float2 wg0 = ld2g(a.dw_w + C3 + hp), wg1 = ld2g(a.dw_w + C6 + C3 + hp), wg2 = ld2g(a.dw_w + 2 * C6 + C3 + hp);
        const float2 cbv = ld2g(a.dw_b + hp), cbg = ld2g(a.dw_b + C3 + hp);
        float Hv[4][2], Hg[4][2], Dd[4][2];
        if (pr == 0) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            Hv[mt][0] = hv[mt][0] + bv.x; Hv[mt][1] = hv[mt][1] + bv.y;
            Hg[mt][0] = hg[mt][0] + bg.x; Hg[mt][1] = hg[mt][1] + bg.y;
            Dd[mt][0] = dd[mt][0]; Dd[mt][1] = dd[mt][1];
          }
        } else {
          const float4* const pk = reinterpret_cast<const float4*>(smem) + tid;
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const float4 q = pk[256 * mt];
            Hv[mt][0] = q.x + bv.x; Hv[mt][1] = q.y + bv.y;
            Hg[mt][0] = q.z + bg.x; Hg[mt][1] = q.w + bg.y;
          }
          const float4 q4 = pk[256 * 4], q5 = pk[256 * 5];
          Dd[0][0] = q4.x; Dd[0][1] = q4.y; Dd[1][0] = q4.z; Dd[1][1] = q4.w;
          Dd[2][0] = q5.x; Dd[2][1] = q5.y; Dd[3][0] = q5.z; Dd[3][1] = q5.w;
        }
        float dcv[4][2], dcg[4][2], acc[2][8];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int k = 0; k < 8; ++k) acc[e][k] = 0.f;
        const float pv0[2] = {gb_shr1(Hv[3][0]), gb_shr1(Hv[3][1])}, pg0[2] = {gb_shr1(Hg[3][0]), gb_shr1(Hg[3][1])};   // frame 4 fi - 1
        const float nv3[2] = {gb_shl1(Hv[0][0]), gb_shl1(Hv[0][1])}, ng3[2] = {gb_shl1(Hg[0][0]), gb_shl1(Hg[0][1])};   // frame 4 fi + 4
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const int fo = 4 * fi + mt, m = m0 - 2 + fo;
          const bool has_dc = fo >= 1 && fo <= GB_BM - 2 && m >= 0 && m < a.M;
          const bool own = fo >= 2 && fo < 2 + GB_OUT && m < a.M;      // rows this tile outputs
          float keep[2] = {1.f, 1.f};
          if (drop) {                                          // network.py:55: mask of the gated tensor, element (m, hp + e)
            const unsigned d0 = sepr_drop_word(dk0, (unsigned)m, (unsigned)(hp >> 1));
            keep[0] = (d0 & 0xffffu) >= a.drop_thr ? a.drop_scale : 0.f;
            keep[1] = (d0 >> 16) >= a.drop_thr ? a.drop_scale : 0.f;
          }
          float gd[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float w0v = e ? wv0.y : wv0.x, w1v = e ? wv1.y : wv1.x, w2v = e ? wv2.y : wv2.x, cbvv = e ? cbv.y : cbv.x;
            const float w0g = e ? wg0.y : wg0.x, w1g = e ? wg1.y : wg1.x, w2g = e ? wg2.y : wg2.x, cbgg = e ? cbg.y : cbg.x;
            float hvm = mt > 0 ? Hv[mt > 0 ? mt - 1 : 0][e] : pv0[e], hgm = mt > 0 ? Hg[mt > 0 ? mt - 1 : 0][e] : pg0[e];
            float hvp = mt < 3 ? Hv[mt < 3 ? mt + 1 : 3][e] : nv3[e], hgp = mt < 3 ? Hg[mt < 3 ? mt + 1 : 3][e] : ng3[e];
            const float hvc = Hv[mt][e], hgc = Hg[mt][e];
            if constexpr (EDGE) {
              hvm *= f0s[mt]; hgm *= f0s[mt];
              hvp *= f2s[mt]; hgp *= f2s[mt];
            }
            const float cv = fmaf(w2v, hvp, fmaf(w1v, hvc, fmaf(w0v, hvm, cbvv)));
            const float cg = fmaf(w2g, hgp, fmaf(w1g, hgc, fmaf(w0g, hgm, cbgg)));
            const float sg = sigmoid_f(cg);
            gd[e] = cv * sg * keep[e];
            const float d = Dd[mt][e] * keep[e];
            const float dv_ = d * sg, dg_ = d * cv * sg * (1.f - sg);
            dcv[mt][e] = has_dc ? dv_ : 0.f;
            dcg[mt][e] = has_dc ? dg_ : 0.f;
            if (own) {
              acc[e][0] = fmaf(dv_, hvm, acc[e][0]); acc[e][1] = fmaf(dv_, hvc, acc[e][1]);
              acc[e][2] = fmaf(dv_, hvp, acc[e][2]); acc[e][3] += dv_;
              acc[e][4] = fmaf(dg_, hgm, acc[e][4]); acc[e][5] = fmaf(dg_, hgc, acc[e][5]);
              acc[e][6] = fmaf(dg_, hgp, acc[e][6]); acc[e][7] += dg_;
            }
          }
          if constexpr (as16) gP[mt][pr] = gb_pack2(gd[0], gd[1]);
          else { gF[mt][2 * pr] = gd[0]; gF[mt][2 * pr + 1] = gd[1]; }
          __builtin_amdgcn_sched_barrier(0);                   // (one frame at a time: hipcc otherwise interleaves all eight and spills)
        }
        // ---- transpose of the conv: dh[t] = w0 dc[t+1] + w1 dc[t] + w2 dc[t-1]   (frames of OTHER sequences do not contribute: f0 / f2) ----
        const float qv0[2] = {gb_shr1(dcv[3][0]), gb_shr1(dcv[3][1])}, qg0[2] = {gb_shr1(dcg[3][0]), gb_shr1(dcg[3][1])};
        const float rv3[2] = {gb_shl1(dcv[0][0]), gb_shl1(dcv[0][1])}, rg3[2] = {gb_shl1(dcg[0][0]), gb_shl1(dcg[0][1])};
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const float f0 = f0s[mt], f2 = f2s[mt];
          float ov[2], og[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const float w0v = e ? wv0.y : wv0.x, w1v = e ? wv1.y : wv1.x, w2v = e ? wv2.y : wv2.x;
            const float w0g = e ? wg0.y : wg0.x, w1g = e ? wg1.y : wg1.x, w2g = e ? wg2.y : wg2.x;
            const float pv = mt > 0 ? dcv[mt > 0 ? mt - 1 : 0][e] : qv0[e], pg = mt > 0 ? dcg[mt > 0 ? mt - 1 : 0][e] : qg0[e];
            const float nv = mt < 3 ? dcv[mt < 3 ? mt + 1 : 3][e] : rv3[e], ng = mt < 3 ? dcg[mt < 3 ? mt + 1 : 3][e] : rg3[e];
            ov[e] = fmaf(w0v * f2, nv, fmaf(w1v, dcv[mt][e], (w2v * f0) * pv));
            og[e] = fmaf(w0g * f2, ng, fmaf(w1g, dcg[mt][e], (w2g * f0) * pg));
          }
          if constexpr (as16) { ovP[mt][pr] = gb_pack2(ov[0], ov[1]); ogP[mt][pr] = gb_pack2(og[0], og[1]); }
          else { ovF[mt][2 * pr] = ov[0]; ovF[mt][2 * pr + 1] = ov[1]; ogF[mt][2 * pr] = og[0]; ogF[mt][2 * pr + 1] = og[1]; }
        }
        // depthwise gradient partials of this tile and pair: sum over the 16 lanes (frames) of each DPP row - an inclusive scan, lane 15 ends up
        // with the row's total - and one 64-byte store per row: part[mb][pair hp + e][8]
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int k = 0; k < 8; ++k) acc[e][k] = gb_row_total(acc[e][k]);
        if (fi == 15) {
          float* po = reinterpret_cast<float*>(reinterpret_cast<char*>(a.part) + ((unsigned)mb * (unsigned)C3 + (unsigned)hp) * 32u);
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            st4(po + 8 * e, make_float4(acc[e][0], acc[e][1], acc[e][2], acc[e][3]));
            st4(po + 8 * e + 4, make_float4(acc[e][4], acc[e][5], acc[e][6], acc[e][7]));
          }
        }
      }
      // ---- the lane's output rows: g [M][3F], dh1 [M][6F] (value half, gate half); 32-bit byte offsets from the (scalar) tensor bases - the
      //      launcher checks M * 6F * 4 < 2^32 - so that no 64-bit per-lane address is formed early and kept alive (or spilled) ----
      char* const gB = static_cast<char*>(a.g);
      char* const hB = static_cast<char*>(a.dh1);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const int fo = 4 * fi + mt, m = m0 - 2 + fo;
        if (!(fo >= 2 && fo < 2 + GB_OUT && m < a.M)) continue;
        if constexpr (as16) {
          const unsigned og_ = ((unsigned)m * (unsigned)C3 + (unsigned)hc) * 2u, oh_ = ((unsigned)m * (unsigned)C6 + (unsigned)hc) * 2u;
          *reinterpret_cast<uint2*>(gB + og_) = make_uint2(gP[mt][0], gP[mt][1]);
          *reinterpret_cast<uint2*>(hB + oh_) = make_uint2(ovP[mt][0], ovP[mt][1]);
          *reinterpret_cast<uint2*>(hB + oh_ + (unsigned)C3 * 2u) = make_uint2(ogP[mt][0], ogP[mt][1]);
        } else {
          const unsigned og_ = ((unsigned)m * (unsigned)C3 + (unsigned)hc) * 4u, oh_ = ((unsigned)m * (unsigned)C6 + (unsigned)hc) * 4u;
          *reinterpret_cast<float4*>(gB + og_) = make_float4(gF[mt][0], gF[mt][1], gF[mt][2], gF[mt][3]);
          *reinterpret_cast<float4*>(hB + oh_) = make_float4(ovF[mt][0], ovF[mt][1], ovF[mt][2], ovF[mt][3]);
          *reinterpret_cast<float4*>(hB + oh_ + (unsigned)C3 * 4u) = make_float4(ogF[mt][0], ogF[mt][1], ogF[mt][2], ogF[mt][3]);
        }
      }
    };
    if (edge_tile) passes(gb_bool<true>{}); else passes(gb_bool<false>{});
#else
    __syncthreads();   // every wave is done with the slab buffers: they become the h1 / dgd tiles

    // ---- stage h1 (+ bias) and dgd: row = frame, a lane holds 4 consecutive channels of one frame per accumulator ----
    {
      const int cl = wn * 16 + 4 * fg;
      const float4 bv = cl_on ? ld4(Cs + (128 + (cl >> 2)) * 4) : ld4(a.b1 + 64 * nb + cl);
      const float4 bg = cl_on ? ld4(Cs + (144 + (cl >> 2)) * 4) : ld4(a.b1 + C3 + 64 * nb + cl);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        float* hr = Hs + (mt * 16 + fi) * GB_HS;
        st4(hr + cl, make_float4(hv[mt][0] + bv.x, hv[mt][1] + bv.y, hv[mt][2] + bv.z, hv[mt][3] + bv.w));
        st4(hr + 64 + cl, make_float4(hg[mt][0] + bg.x, hg[mt][1] + bg.y, hg[mt][2] + bg.z, hg[mt][3] + bg.w));
        st4(Ds + (mt * 16 + fi) * GB_DS + cl, make_float4(dd[mt][0], dd[mt][1], dd[mt][2], dd[mt][3]));
      }
    }
    // the next tile of this workgroup: all its activation slabs go in flight now, under the epilogue below
    int nxt = tile + gridDim.x, mb_n = 0, nb_n = 0;
    while (nxt < ntiles && !decode(nxt, mb_n, nb_n)) nxt += gridDim.x;
    if (nxt < ntiles) load_tile(mb_n);
    __syncthreads();

    // ---- pass A: conv, GLU, dropout, GLU backward for rows 1..62; thread = 4 hidden channel pairs x 4 consecutive rows ----
    const int q4 = tid & 15, strip = tid >> 4;
    const int c4 = 4 * q4, hc = 64 * nb + c4;                    // hidden value channel of column 0 (gate: C3 + hc)
    const int C6 = 2 * C3;
    float4 wv0, wv1, wv2, wg0, wg1, wg2, cbv, cbg;
    if (cl_on) {
      const float* cq = Cs + 4 * q4;
      wv0 = ld4(cq); wv1 = ld4(cq + 64); wv2 = ld4(cq + 128); wg0 = ld4(cq + 192); wg1 = ld4(cq + 256); wg2 = ld4(cq + 320);
      cbv = ld4(cq + 384); cbg = ld4(cq + 448);
    } else {
      wv0 = ld4(a.dw_w + hc); wv1 = ld4(a.dw_w + C6 + hc); wv2 = ld4(a.dw_w + 2 * C6 + hc);
      wg0 = ld4(a.dw_w + C3 + hc); wg1 = ld4(a.dw_w + C6 + C3 + hc); wg2 = ld4(a.dw_w + 2 * C6 + C3 + hc);
      cbv = ld4(a.dw_b + hc); cbg = ld4(a.dw_b + C3 + hc);
    }
    float acc8[4][8];                                          // [column][w0 w1 w2 b of the value, then of the gate]
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int k = 0; k < 8; ++k) acc8[e][k] = 0.f;
    // Interior tiles (every row this tile touches - m0 - 2 .. m0 + 61 - lies strictly inside ONE sequence: all but ~1 tile in 130 at T = 8000)
    // run the instantiation WITHOUT the conv's zero-padding flags: 24 multiplies per row and thread out of both passes (round 5; the fused
    // forward has done the same since round 1).  Wave-uniform choice, bit-identical results (a flag of 1.0 multiplies exactly).
    const int lo_t = (m0 >= 1) ? (m0 - 1) % a.T : 0;
    const bool edge_tile = !(m0 >= 2 && lo_t >= 1 && lo_t + 62 <= a.T - 2 && m0 + GB_BM - 2 <= a.M);
    auto passes = [&](auto edge_c) {
    constexpr bool EDGE = decltype(edge_c)::value;
    float4 dcv[4], dcg[4];
    constexpr bool SLIDE = (SEPR_GB_SLIDE & 1) && !EDGE;        // pass A
    constexpr bool SLIDE_B = (SEPR_GB_SLIDE & 2) && !EDGE;      // pass B
    {
#pragma clang fp contract(off)
      // SLIDE: rows r - 1, r of the window, carried from row to row (rows outside the tile are clamped: the rows that would use them are skipped)
      [[maybe_unused]] float4 wvm, wgm, wvc, wgc;
      if constexpr (SLIDE) {
        const float* h0 = Hs + (4 * strip) * GB_HS + c4;
        const float* hm = strip > 0 ? h0 - GB_HS : h0;
        wvm = ld4(hm); wgm = ld4(hm + 64);
        wvc = ld4(h0); wgc = ld4(h0 + 64);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * strip + i;
        const int m = m0 - 2 + r;
        dcv[i] = zero4();
        dcg[i] = zero4();
        float4 hvc, hgc, hvm, hgm, hvp, hgp;
        if constexpr (SLIDE) {
          const float* hp = Hs + (r + 1 < GB_BM ? r + 1 : GB_BM - 1) * GB_HS + c4;
          hvp = ld4(hp); hgp = ld4(hp + 64);
          hvm = wvm; hgm = wgm; hvc = wvc; hgc = wgc;
          wvm = wvc; wgm = wgc; wvc = hvp; wgc = hgp;
        }
        if (r < 1 || r > GB_BM - 2 || m < 0 || m >= a.M) continue;
        const float* hr = Hs + r * GB_HS + c4;
        if constexpr (!SLIDE) {
          hvc = ld4(hr); hgc = ld4(hr + 64);
          hvm = ld4(hr - GB_HS); hgm = ld4(hr - GB_HS + 64); hvp = ld4(hr + GB_HS); hgp = ld4(hr + GB_HS + 64);
        }
        if constexpr (EDGE) {                                  // zero padding at the sequence ends (tiles that touch one: ~1 in 130)
          const int t = m % a.T;
          const float f0 = t > 0 ? 1.f : 0.f, f2 = t < a.T - 1 ? 1.f : 0.f;
          hvm.x *= f0; hvm.y *= f0; hvm.z *= f0; hvm.w *= f0;
          hgm.x *= f0; hgm.y *= f0; hgm.z *= f0; hgm.w *= f0;
          hvp.x *= f2; hvp.y *= f2; hvp.z *= f2; hvp.w *= f2;
          hgp.x *= f2; hgp.y *= f2; hgp.z *= f2; hgp.w *= f2;
        }
        float4 keep = make_float4(1.f, 1.f, 1.f, 1.f);
        if (drop) {                                            // network.py:55: mask of the gated tensor, element (m, hc + e)
          const unsigned d0 = sepr_drop_word(dk0, (unsigned)m, (unsigned)(hc >> 1)), d1 = sepr_drop_word(dk0, (unsigned)m, (unsigned)(hc >> 1) + 1u);
          keep.x = (d0 & 0xffffu) >= a.drop_thr ? a.drop_scale : 0.f;
          keep.y = (d0 >> 16) >= a.drop_thr ? a.drop_scale : 0.f;
          keep.z = (d1 & 0xffffu) >= a.drop_thr ? a.drop_scale : 0.f;
          keep.w = (d1 >> 16) >= a.drop_thr ? a.drop_scale : 0.f;
        }
        const float4 dg = ld4(Ds + r * GB_DS + c4);
        const bool own = r >= 2 && r < 2 + GB_OUT;              // rows this tile outputs (m < M already checked)
        float4 gd;
#define SEPR_GB_ELEM(X, E)                                                                            \
  {                                                                                                   \
    const float cv = fmaf(wv2.X, hvp.X, fmaf(wv1.X, hvc.X, fmaf(wv0.X, hvm.X, cbv.X)));             \
    const float cg = fmaf(wg2.X, hgp.X, fmaf(wg1.X, hgc.X, fmaf(wg0.X, hgm.X, cbg.X)));             \
    const float sg = sigmoid_f(cg);                                                                   \
    gd.X = cv * sg * keep.X;                                                                          \
    const float d = dg.X * keep.X;                                                                    \
    const float dv_ = d * sg, dg_ = d * cv * sg * (1.f - sg);                                        \
    dcv[i].X = dv_;                                                                                   \
    dcg[i].X = dg_;                                                                                   \
    if (own) {                                                                                        \
      acc8[E][0] = fmaf(dv_, hvm.X, acc8[E][0]); acc8[E][1] = fmaf(dv_, hvc.X, acc8[E][1]);          \
      acc8[E][2] = fmaf(dv_, hvp.X, acc8[E][2]); acc8[E][3] += dv_;                                  \
      acc8[E][4] = fmaf(dg_, hgm.X, acc8[E][4]); acc8[E][5] = fmaf(dg_, hgc.X, acc8[E][5]);          \
      acc8[E][6] = fmaf(dg_, hgp.X, acc8[E][6]); acc8[E][7] += dg_;                                  \
    }                                                                                                 \
  }
        SEPR_GB_ELEM(x, 0)
        SEPR_GB_ELEM(y, 1)
        SEPR_GB_ELEM(z, 2)
        SEPR_GB_ELEM(w, 3)
#undef SEPR_GB_ELEM
        if (own) gb_store4(a.g, (long long)m * C3 + hc, gd, a.out16 != 0);
      }
      // depthwise gradient partials: sum over the 4 strips of this wave (lanes 16 apart), then over the 4 waves through LDS
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float v = acc8[e][k];
          v += __shfl_xor(v, 16, 64);
          v += __shfl_xor(v, 32, 64);
          acc8[e][k] = v;
        }
    }
    __syncthreads();   // all reads of h1 / dgd done: dc overwrites h1 in place, the reduction scratch overwrites dgd
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float* hr = Hs + (4 * strip + i) * GB_HS + c4;
      st4(hr, dcv[i]);
      st4(hr + 64, dcg[i]);
    }
    if (lane < 16) {
      // (row stride 36 floats, not 32: the 16 lanes' 16-byte stores then hit 16 distinct bank groups instead of two - PMC, round 5: LDS bank
      //  conflicts were 9.8 % of this kernel's CU cycles)
      float* rs = Ds + (wn * 16 + q4) * GB_RS;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        st4(rs + 8 * e, make_float4(acc8[e][0], acc8[e][1], acc8[e][2], acc8[e][3]));
        st4(rs + 8 * e + 4, make_float4(acc8[e][4], acc8[e][5], acc8[e][6], acc8[e][7]));
      }
    }
    __syncthreads();
    // ---- pass B: transpose of the conv, rows 2..61 ----
    {
#pragma clang fp contract(off)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = 4 * strip + i;
        const int m = m0 - 2 + r;
        if (r < 2 || r >= 2 + GB_OUT || m >= a.M) continue;
        float f0 = 1.f, f2 = 1.f;
        if constexpr (EDGE) {
          const int t = m % a.T;
          f0 = t > 0 ? 1.f : 0.f;
          f2 = t < a.T - 1 ? 1.f : 0.f;
        }
        const float* hr = Hs + r * GB_HS + c4;
        float4 pv, pg, nv, ng;
        if constexpr (SLIDE_B) {                               // the neighbour rows inside the thread's own strip are its own registers (what it stored above)
          if (i > 0) { pv = dcv[i > 0 ? i - 1 : 0]; pg = dcg[i > 0 ? i - 1 : 0]; }
          else { pv = ld4(hr - GB_HS); pg = ld4(hr - GB_HS + 64); }
          if (i < 3) { nv = dcv[i < 3 ? i + 1 : 3]; ng = dcg[i < 3 ? i + 1 : 3]; }
          else { nv = ld4(hr + GB_HS); ng = ld4(hr + GB_HS + 64); }
        } else {
          pv = ld4(hr - GB_HS); pg = ld4(hr - GB_HS + 64); nv = ld4(hr + GB_HS); ng = ld4(hr + GB_HS + 64);
        }
        // dh[t] = w0 dc[t+1] + w1 dc[t] + w2 dc[t-1]   (frames of OTHER sequences do not contribute: f0 / f2)
        float4 ov, og;
        ov.x = fmaf(wv0.x * f2, nv.x, fmaf(wv1.x, dcv[i].x, (wv2.x * f0) * pv.x));
        ov.y = fmaf(wv0.y * f2, nv.y, fmaf(wv1.y, dcv[i].y, (wv2.y * f0) * pv.y));
        ov.z = fmaf(wv0.z * f2, nv.z, fmaf(wv1.z, dcv[i].z, (wv2.z * f0) * pv.z));
        ov.w = fmaf(wv0.w * f2, nv.w, fmaf(wv1.w, dcv[i].w, (wv2.w * f0) * pv.w));
        og.x = fmaf(wg0.x * f2, ng.x, fmaf(wg1.x, dcg[i].x, (wg2.x * f0) * pg.x));
        og.y = fmaf(wg0.y * f2, ng.y, fmaf(wg1.y, dcg[i].y, (wg2.y * f0) * pg.y));
        og.z = fmaf(wg0.z * f2, ng.z, fmaf(wg1.z, dcg[i].z, (wg2.z * f0) * pg.z));
        og.w = fmaf(wg0.w * f2, ng.w, fmaf(wg1.w, dcg[i].w, (wg2.w * f0) * pg.w));
        gb_store4(a.dh1, (long long)m * C6 + hc, ov, a.out16 != 0);
        gb_store4(a.dh1, (long long)m * C6 + C3 + hc, og, a.out16 != 0);
      }
    }
    };
    if (edge_tile) passes(gb_bool<true>{}); else passes(gb_bool<false>{});
    // per-tile depthwise partials: part[mb][pair][8], the 64 pairs of this column block are 512 consecutive floats
    {
      float* po = a.part + ((long long)mb * C3 + 64 * nb) * 8;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int o = tid + GB_THREADS * h;                    // (q, slot) = (o / 32, o % 32)
        const float* d = Ds + (o >> 5) * GB_RS + (o & 31);
        const float t = (d[0] + d[16 * GB_RS]) + (d[32 * GB_RS] + d[48 * GB_RS]);
        if (a.acc_part) wacc[h] += t;
        else po[o] = t;
      }
    }
#endif
    tile = nxt;
    mb = mb_n;
    nb = nb_n;
  }
#if !SEPR_GB_REGEPI
  if (a.acc_part && any_tile) {
    float* po = a.part + ((long long)mb_first * C3 + 64 * nb_first) * 8;
    po[tid] = wacc[0];
    po[tid + GB_THREADS] = wacc[1];
  }
#endif
}
}  // namespace

size_t gcfn_bwd_fused_ws(long long M, int F) {   // partials + the pre-reduction scratch of launch_gcfn_mid_reduce
  const long long MB = (M + GB_OUT - 1) / GB_OUT;
  return align_up((size_t)MB * 3 * F * 8 * sizeof(float)) + gcfn_mid_reduce_ws(3 * F);
}

namespace {
// dy [M][F] fp32 -> out [M][F] bf16 = bf16(dropout1(dy)) (network.py:57's mask, 16-bit generator site 1, keep scale folded in; p = 0: a
// plain conversion): the dy operand of the PL middle kernel and the A operand of net2.2's weight-gradient contraction
__global__ __launch_bounds__(256) void gcfn_dyplane_kernel(const float* __restrict__ dy, unsigned short* __restrict__ out, long long M, int F,
                                                           unsigned thr, float scale, unsigned long long seed,
                                                           const unsigned long long* __restrict__ salt) {
  const DropKey dk1 = sepr_drop_key(seed, salt, 1u);
  const int f8 = F >> 3;
  const long long total = M * f8;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long m = i / f8;
    const int c = (int)(i - m * f8) * 8;
    const float4 p0 = ld4(dy + m * F + c), p1 = ld4(dy + m * F + c + 4);
    float v[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
    if (thr) {
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const unsigned d = sepr_drop_word(dk1, (unsigned)m, (unsigned)((c + e) >> 1));
        v[e] = (d & 0xffffu) >= thr ? v[e] * scale : 0.f;
        v[e + 1] = (d >> 16) >= thr ? v[e + 1] * scale : 0.f;
      }
    }
    gb_bf16x8 h;
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = (__bf16)v[e];
    *reinterpret_cast<gb_bf16x8*>(out + m * F + c) = h;
  }
}
}  // namespace
int launch_gcfn_dyplane(const float* dy, void* out16, long long M, int F, float p, unsigned long long seed, const unsigned long long* salt,
                        hipStream_t st) {
  if (M <= 0) return SEPR_OK;
  if (!dy || !out16 || F % 8 || !(p >= 0.f) || !(p < 1.f) || M > 0x7fffffffLL) return SEPR_EINVAL;
  const long long blocks = (M * (F >> 3) + 255) / 256;
  hipLaunchKernelGGL(gcfn_dyplane_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, st, dy,
                     static_cast<unsigned short*>(out16), M, F, p > 0.f ? sepr_drop_thr16(p) : 0u, p > 0.f ? sepr_drop_scale16(p) : 1.0f, seed, salt);
  SEPR_CHECK_LAUNCH("gcfn_dyplane_kernel");
  return SEPR_OK;
}

int launch_gcfn_bwd_fused(const float* x, const float* stats, const float* dy, int n, int T, int F, const sepr_gcfn_tw* w, void* g,
                          void* dh1, int out16, float* dyq, float* dw_g, float* db_g, float p, unsigned long long seed,
                          const unsigned long long* salt, void* ws, size_t ws_bytes, hipStream_t st, const void* xh16, const void* dy16) {
  const long long M = (long long)n * T;
  if (M <= 0) return SEPR_OK;
  if (!x || !stats || !dy || !w || !w->up.wp || !w->up.b || !w->down_t.wp || !w->dw_w || !w->dw_b || !g || !dh1 || !dw_g || !db_g ||
      (F != 64 && F != 128) || !(p >= 0.f) || !(p < 1.f))
    return SEPR_EINVAL;
  if (M > 0x7fffffffLL / 8) return SEPR_EINVAL;
  if (xh16 && dy16 && M * F * 2 >= (1LL << 32)) return SEPR_EINVAL;   // 32-bit byte offsets of the plane-staged form
  if (!ws || ws_bytes < gcfn_bwd_fused_ws(M, F)) return SEPR_EWORKSPACE;
  GcfnBwdArgs a;
  a.x = x; a.stats = stats; a.dy = dy; a.M = (int)M; a.T = T; a.F = F;
  a.w1p = w->up.wp; a.b1 = w->up.b; a.w2tp = w->down_t.wp; a.dw_w = w->dw_w; a.dw_b = w->dw_b;
  a.g = g; a.dh1 = dh1; a.out16 = out16; a.dyq = p > 0.f ? dyq : nullptr;
  a.part = static_cast<float*>(ws);
  a.drop_thr = p > 0.f ? sepr_drop_thr16(p) : 0u;
  a.drop_scale = p > 0.f ? sepr_drop_scale16(p) : 1.0f;
  a.seed = seed; a.salt = salt;
  if (p > 0.f && !dyq && !(xh16 && dy16)) return SEPR_EINVAL;      // (the plane-staged form gets dropout1(dy) as dy16)
  const int MB = (int)((M + GB_OUT - 1) / GB_OUT), NB = 3 * F / 64;
  const int ntiles = ((MB + 7) / 8) * 8 * NB;
  const int cap = persistent_grid();
  const int grid = ntiles < cap ? ntiles : cap;
  long long slot = -1;
  const bool timed = prof_begin(SEPR_SITE_GCFN_BWD, st, &slot);
  const bool one = w->up.planes == 1;
  if (SEPR_GB_REGEPI && ((out16 != 0) != one || M * 6LL * F * 4 >= (1LL << 32))) return SEPR_EINVAL;   // (the register epilogue: bf16 outputs <=> plain-bf16 arithmetic; 32-bit store offsets)
  a.xh16 = static_cast<const unsigned short*>(xh16);
  a.dy16 = static_cast<const unsigned short*>(dy16);
  const bool pl = one && xh16 && dy16;
  a.acc_part = 0;
  int nslots = MB;                                                     // rows of `part` the reduction walks
  if (pl) {
    const int g3 = (ntiles < cap / 2 * SEPR_GB_PL_WGS) ? ntiles : cap / 2 * SEPR_GB_PL_WGS;   // SEPR_GB_PL_WGS workgroups per CU
    static const bool acc_off = [] { const char* e = getenv("SEPR_GB_ACC"); return e && e[0] == '0'; }();     // (A/B switch, read once)
    if (!SEPR_GB_REGEPI && !acc_off && ntiles > g3 && g3 % (8 * NB) == 0) {   // every workgroup keeps its column block: tile b + k g3 -> nb = (b >> 3) % NB
      a.acc_part = 1;
      nslots = g3 / NB;                                                // first tiles' mb = 0 .. g3 / NB - 1, all below MB (ntiles > g3)
    }
    if (F == 128) hipLaunchKernelGGL((gcfn_bwd_mid_kernel<1, 2, true>), dim3(g3), dim3(GB_THREADS), 0, st, a);
    else hipLaunchKernelGGL((gcfn_bwd_mid_kernel<1, 1, true>), dim3(g3), dim3(GB_THREADS), 0, st, a);
  } else if (F == 128) {
    if (one) hipLaunchKernelGGL((gcfn_bwd_mid_kernel<1, 2>), dim3(grid), dim3(GB_THREADS), 0, st, a);
    else hipLaunchKernelGGL((gcfn_bwd_mid_kernel<3, 2>), dim3(grid), dim3(GB_THREADS), 0, st, a);
  } else if (F == 64) {
    if (one) hipLaunchKernelGGL((gcfn_bwd_mid_kernel<1, 1>), dim3(grid), dim3(GB_THREADS), 0, st, a);
    else hipLaunchKernelGGL((gcfn_bwd_mid_kernel<3, 1>), dim3(grid), dim3(GB_THREADS), 0, st, a);
  } else {
    return SEPR_EINVAL;   // (the fused pair exists for F = 64 / 128: sepr_gcfn_fused.hip)
  }
  // algorithmic FLOPs per row: recomputed up-projection 2 F 6F + input gradient of net2.2 2 F 3F + conv / GLU forward and backward
  if (timed) {
    prof_end(slot, (double)M * (18.0 * F * F + 60.0 * 3 * F), st);
    prof_bytes((double)M * (8.0 * F + 8.0 + (out16 ? 2.0 : 4.0) * 9.0 * F + (p > 0.f ? 4.0 * F : 0.0)));   // x, dy, stats in; g, dh1 (, dyq) out
  }
  SEPR_CHECK_LAUNCH("gcfn_bwd_mid_kernel");
  float* scratch = reinterpret_cast<float*>(static_cast<char*>(ws) + align_up((size_t)MB * 3 * F * 8 * sizeof(float)));
  return launch_gcfn_mid_reduce(a.part, nslots, 3 * F, dw_g, db_g, scratch, st);
}

}  // namespace sepr
