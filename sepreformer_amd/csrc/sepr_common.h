// Shared device helpers and host-side launch plumbing for libsepr_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/sepr.h"

namespace sepr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- math -----------------------------------------------------------------------------------------
// exact-erf GELU (torch.nn.GELU() default, reference modules/module.py:17,70, modules/network.py:169)
// (explicit fmaf / no implicit contraction: the result of an element must not depend on how the compiler
//  happened to vectorise the loop iteration it sits in)
__device__ __forceinline__ float gelu_exact(float x) {
#pragma clang fp contract(off)
  const float h = 0.5f * x;
  return fmaf(h, erff(x * 0.70710678118654752440f), h);
}
// GELU for the fused bf16x3 kernels: erf by Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7, i.e. ~140 dB on the
// activation, 40 dB below the split-fp32 products feeding it), written so that 1 + erf(z) for z < 0 is the
// tail term itself (no cancellation).  Branch-free, 2 transcendentals: ~16 instructions against ~60 for the
// two-range erff() every lane of a wave ends up executing.  The exact-f32 path keeps gelu_exact.
__device__ __forceinline__ float gelu_fast(float x) {
#pragma clang fp contract(off)
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = (p * t) * __builtin_amdgcn_exp2f((x * x) * -0.72134752044448170368f);   // exp(-z^2)
  const float r = x >= 0.f ? 2.0f - e : e;                                                  // 1 + erf(x / sqrt 2)
  return (0.5f * x) * r;
}
// sigmoid with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of an IEEE division (~10 instructions):
// the GLU / gate epilogues evaluate it for every hidden element
__device__ __forceinline__ float sigmoid_f(float x) {
#pragma clang fp contract(off)
  return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}

// IEEE-division sigmoid of the training path's row-wise kernels (their backward formulas use sig (1 - sig))
__device__ __forceinline__ float sigmoid_exact(float x) { return 1.0f / (1.0f + expf(-x)); }

// GLU on a gate that arrives PRE-SCALED by -log2(e) (the packers fold the factor into the gate's conv taps and bias):
// value * sigmoid(gate) = value * rcp(1 + exp2(gs)), one multiply less per hidden element than sigmoid_f
__device__ __forceinline__ float glu_prescaled(float val, float gs) {
#pragma clang fp contract(off)
  return val * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(gs));
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
// 4 fp32 -> 4 bf16 (round to nearest even), one 8-byte store at element index idx of a bf16 tensor addressed through a float*
__device__ __forceinline__ void st4_bf16(float* base, long long idx, float4 v) {
  typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
  const bf16x4_t h = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
  *reinterpret_cast<bf16x4_t*>(reinterpret_cast<__bf16*>(base) + idx) = h;
}

// gfx950 fault (tools/probe/pk_opsel.hip): a packed-f32 VALU instruction whose LOW result half selects the HIGH dword
// of src1 (VOP3P op_sel:[x,1]) returns wrong low halves in lanes 16-31 / 48-63 while another wave of the SIMD executes bf16 MFMAs.  hipcc's
// SLP vectoriser emits that form whenever it splats an ODD register into a packed operand - e.g. rstd, the high dword of a loaded
// (mean, rstd) pair - or builds a horizontal add.  tools/isa_lint.py stops the build on any such instruction; these helpers pin the
// safe forms where the natural source would produce it.
//   (v - mean) * rstd on four values: broadcast pairs built with moves, packed instructions WITHOUT operand selects
__device__ __forceinline__ float4 norm4_pinned(float4 v, float mean, float rstd) {
  typedef float sepr_f2 __attribute__((ext_vector_type(2)));
  sepr_f2 xy = {v.x, v.y}, zw = {v.z, v.w};
  const sepr_f2 mm = {mean, mean}, rr = {rstd, rstd};
  asm volatile("v_pk_add_f32 %0, %0, %2 neg_lo:[0,1] neg_hi:[0,1]\n\tv_pk_add_f32 %1, %1, %2 neg_lo:[0,1] neg_hi:[0,1]\n\t"
               "v_pk_mul_f32 %0, %0, %3\n\tv_pk_mul_f32 %1, %1, %3" : "+v"(xy), "+v"(zw) : "v"(mm), "v"(rr));
  return make_float4(xy[0], xy[1], zw[0], zw[1]);
}

// ---- LayerNorm row statistics ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float reduce16(float v) {
  v += __shfl_xor(v, 8, 16);
  v += __shfl_xor(v, 4, 16);
  v += __shfl_xor(v, 2, 16);
  v += __shfl_xor(v, 1, 16);
  return v;
}
// (the empty asm keeps hipcc from turning the final add into a packed horizontal add - v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0], a form
//  that is faulty on gfx950 beside bf16 MFMAs: norm4_pinned above, tools/isa_lint.py)
__device__ __forceinline__ float sum4(float4 v) {
  float a = v.x + v.y;
  const float b = v.z + v.w;
  asm volatile("" : "+v"(a));
  return a + b;
}
// (mean, rstd) of ONE row p[0 .. 4 nf4) by the 16 lanes sub = 0..15 of a 16-lane group: two passes over registers (exact mean, then centred
// variance; reference torch.nn.LayerNorm, modules/network.py:50,81,133,162).  THE arithmetic of the LayerNorm statistics: rowstats_kernel
// (sepr_pointwise.hip) and the statistics tail of the wide projection core (sepr_gemm_x3w.h, round 6) both call it, so a row's statistics
// do not depend on which of the two produced them.
// Every operation is spelled out (no implicit contraction) so that the two call sites compile to the same arithmetic, and the squares are
// summed through the pinned forms: the natural "(a a + b b) + (c c + e e)" is what hipcc turns into the gfx950-faulty packed horizontal add.
// NI = float4 chunks per lane (16 NI >= nf4): rows of up to 512 columns take 8, up to 256 columns 4 - the chunks past nf4 add exact zeros, so
// the result does not depend on NI.
template <int NI = 8>
__device__ __forceinline__ void rowstats_one(const float* __restrict__ p, int nf4, float invF, float eps, int sub, float& mean, float& rstd) {
#pragma clang fp contract(off)
  float4 v[NI];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = sub + 16 * i;
    v[i] = (c < nf4) ? ld4(p + 4 * c) : zero4();
    s += sum4(v[i]);
  }
  mean = reduce16(s) * invF;
  float d = 0.f;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int c = sub + 16 * i;
    if (c < nf4) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, e = v[i].w - mean;
      float t0 = fmaf(b, b, a * a);
      const float t1 = fmaf(e, e, cc * cc);
      asm volatile("" : "+v"(t0));
      d += t0 + t1;
    }
  }
  rstd = 1.0f / sqrtf(fmaf(reduce16(d), invF, eps));
}

// ---- LDS-DMA as inline asm ------------------------------------------------------------------------------------------------------------
// hipcc (ROCm 7.2) puts an s_waitcnt vmcnt(0) in front of the first LDS read behind any LDS-DMA it knows about (it cannot tell the copy's
// destination from the other LDS reads of the array), which serialises "request the next weight chunk, multiply the current one" - the whole
// point of the copy (tools/isa_trace.py shows the waits).  Issued from inline asm the copy is invisible to the waitcnt pass and the caller
// owns the ordering: s_waitcnt vmcnt(N) + a barrier before the first read of the destination.
// one wave-wide copy: 64 lanes x 16 bytes -> 1 KiB at the wave-uniform LDS address
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"   // m0 is "reserved": hipcc re-materialises it before each of its own uses
// (scalar-base form: wave-uniform 64-bit source base in an SGPR pair + one 32-bit per-lane byte offset - no 64-bit per-lane addresses to
//  keep alive or spill; a spilled address would be re-loaded by a scratch_load, and the wait for THAT drains every copy in flight)
__device__ __forceinline__ void glds16_asm(const void* src_wave, unsigned lane_off, unsigned lds_wave) {
  // s_nop 4: the instructions inside an asm statement are invisible to hipcc's hazard recognizer.  The source base may have been written by
  // a VALU instruction just before the statement (v_readfirstlane / v_readlane - e.g. an SGPR restored from its spill lane), and "VALU writes
  // SGPR -> VMEM reads that SGPR" needs 5 wait states on gfx9-family parts; the same gap covers "s_mov m0 -> LDS-DMA".
#ifndef SEPR_GLDS_NOP
#define SEPR_GLDS_NOP 3     // s_mov (1) + s_nop 3 (4) = the 5 wait states; 0 = round 6's first form, A/B only (tools/variants.mk gldsnop0): +0.7 % Base, +1.7 % Large, unsafe
#endif
#define SEPR_STR2(x) #x
#define SEPR_STR(x) SEPR_STR2(x)
  asm volatile("s_mov_b32 m0, %2\n\ts_nop " SEPR_STR(SEPR_GLDS_NOP) "\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(src_wave), "s"(lds_wave) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p; }
#pragma clang diagnostic pop

// butterfly sums over the lanes of a wave64
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- dropout generator pieces shared by the training kernels and the projection epilogues (documented in sepr_train.h) ----
__device__ __forceinline__ unsigned long long sepr_mix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ unsigned sepr_hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
struct DropKey { unsigned ka, kb; };
__device__ __forceinline__ DropKey sepr_drop_key(unsigned long long seed, const unsigned long long* salt, unsigned site) {
  const unsigned long long z = sepr_mix64(seed ^ (salt ? *salt : 0ull) ^ ((unsigned long long)(site + 1) << 56));
  DropKey k;
  k.ka = (unsigned)z;
  k.kb = (unsigned)(z >> 32);
  return k;
}
__device__ __forceinline__ unsigned sepr_drop_word(DropKey k, unsigned row, unsigned pair) {
  return sepr_hash32((row * 0x9E3779B1u + k.ka) ^ (pair * 0x85EBCA77u + k.kb));
}

// ---- host side ------------------------------------------------------------------------------------
void set_hip_error(hipError_t e, const char* where);

#define SEPR_CHECK_LAUNCH(where)                       \
  do {                                                 \
    hipError_t e__ = hipGetLastError();                \
    if (e__ != hipSuccess) {                           \
      ::sepr::set_hip_error(e__, where);               \
      return SEPR_EHIP;                                \
    }                                                  \
  } while (0)

#define SEPR_TRY(expr)                \
  do {                                \
    int rc__ = (expr);                \
    if (rc__ != SEPR_OK) return rc__; \
  } while (0)

// latched A/B switches (include/sepr.h SEPR_KNOB_*): one getenv per switch per process, not per launch (sepr_api.hip)
int knob(int id);

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// sequential carve-out of the caller's workspace
struct Arena {
  char* base;
  size_t size, off;
  Arena(void* p, size_t n) : base(static_cast<char*>(p)), size(n), off(0) {}
  float* f32(size_t count) { return static_cast<float*>(take(count * sizeof(float))); }
  double* f64(size_t count) { return static_cast<double*>(take(count * sizeof(double))); }
  void* take(size_t bytes) {
    size_t o = align_up(off);
    off = o + bytes;
    if (base == nullptr || off > size) return nullptr;
    return base + o;
  }
  bool ok() const { return base != nullptr && off <= size; }
};

inline int cdiv(long long a, long long b) { return static_cast<int>((a + b - 1) / b); }

}  // namespace sepr
