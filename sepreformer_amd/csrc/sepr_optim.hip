// Optimizer step of the reference training loop (engine.py:76-77: clip_grad_norm_(max_norm) + AdamW.step) as THREE launches over
// the whole model, whatever its number of parameter tensors (Base: 1 312):
//   opt_sqsum_kernel   sum of squares of the flat gradient buffer, 1 024 block partials in double (fixed order, no atomics)
//   opt_prep_kernel    one block: total norm, clip coefficient, step counter += 1, the bias corrections of this step
//   opt_adamw_kernel   one block per 1 024 elements of one tensor (block table), decoupled weight decay + Adam update
// torch's route is one multi_tensor_apply launch per ~36 tensors (37 launches for the update, 12 for the norms, 37 for the clip
// scaling: 1.8 ms per step at 14.7 M parameters = 260 GB/s); these three stream the 14.7 M x 28 bytes once (~0.15 ms).
// Arithmetic = torch's fused AdamW (fused_adam_utils.cuh adam_math, decoupled weight decay): fp32 per element, the bias corrections
// from pow() in double; the clip coefficient is min(1, max_norm / (norm + 1e-6)) applied to the gradient inside the update.
#include "sepr_common.h"
#include <math.h>

namespace sepr {
namespace {
constexpr int OPT_TPB = 256;
constexpr int OPT_BLOCK = 1024;       // elements per block of the update kernel (one float4 per thread)
constexpr int OPT_PARTS = 1024;       // block partials of the norm

__global__ __launch_bounds__(OPT_TPB) void opt_sqsum_kernel(const float* __restrict__ g, long long n, double* __restrict__ part) {
  __shared__ double sh[OPT_TPB / 64];
  // block b owns the contiguous range [b * per, (b + 1) * per): the partial of a block does not depend on the grid
  const long long per = ((n + OPT_PARTS - 1) / OPT_PARTS + 3) / 4 * 4;
  const long long beg = (long long)blockIdx.x * per;
  const long long end = beg + per < n ? beg + per : n;
  double acc = 0.0;
  for (long long i = beg + 4LL * threadIdx.x; i < end; i += 4LL * OPT_TPB) {
    if (i + 3 < end) {
      const float4 v = ld4(g + i);
      acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    } else {
      for (long long j = i; j < end; ++j) acc += (double)g[j] * g[j];
    }
  }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// scal[0] = total norm, [1] = clip coefficient, [2] = lr / bias_correction1, [3] = sqrt(bias_correction2), [4] = lr * weight_decay
__global__ __launch_bounds__(OPT_TPB) void opt_prep_kernel(const double* __restrict__ part, int with_norm, float max_norm, const float* __restrict__ lr,
                                                          double beta1, double beta2, float weight_decay, double* __restrict__ step,
                                                          float* __restrict__ scal) {
  __shared__ double sh[OPT_TPB];
  double acc = 0.0;
  if (with_norm)
    for (int i = threadIdx.x; i < OPT_PARTS; i += OPT_TPB) acc += part[i];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int o = OPT_TPB / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  const float norm = (float)sqrt(sh[0]);
  float clip = 1.0f;
  if (with_norm && max_norm > 0.f) {
    clip = max_norm / (norm + 1e-6f);
    // torch.clamp(max = 1) semantics of clip_grad_norm_: a non-finite norm gives a NaN coefficient that reaches EVERY gradient element
    // (the step then poisons the weights visibly, as with torch, instead of applying the finite elements unclipped)
    clip = !(clip >= 1.0f) ? clip : 1.0f;
  }
  const double t = *step + 1.0;
  *step = t;
  const double bc1 = 1.0 - pow(beta1, t), bc2 = 1.0 - pow(beta2, t);
  const float l = *lr;
  scal[0] = norm;
  scal[1] = clip;
  scal[2] = (float)((double)l / bc1);
  scal[3] = (float)sqrt(bc2);
  scal[4] = l * weight_decay;
}

struct OptTables {
  float* const* params;           // [ntensors]
  const long long* grad_off;      // [ntensors] element offset of the tensor's gradient in the flat gradient buffer
  const long long* state_off;     // [ntensors] element offset of its moments in exp_avg / exp_avg_sq
  const int* numel;               // [ntensors]
  const int2* blocks;             // [nblocks] (tensor, first element)
};

__global__ __launch_bounds__(OPT_TPB) void opt_adamw_kernel(const OptTables t, const float* __restrict__ grads, float* __restrict__ exp_avg,
                                                           float* __restrict__ exp_avg_sq, const float* __restrict__ scal, float omb1,
                                                           float beta2, float omb2, float eps) {
#pragma clang fp contract(off)
  const int2 b = t.blocks[blockIdx.x];
  const int ti = b.x;
  const int n = t.numel[ti];
  const int e0 = b.y + 4 * threadIdx.x;
  if (e0 >= n) return;
  float* __restrict__ p = t.params[ti] + e0;
  const float* __restrict__ g = grads + t.grad_off[ti] + e0;
  const long long so = t.state_off[ti] + e0;
  const float clip = scal[1], step_size = scal[2], bc2_sqrt = scal[3], lr_wd = scal[4];
  const int cnt = n - e0 < 4 ? n - e0 : 4;
  float pv[4], gv[4], mv[4], vv[4];
  if (cnt == 4) {       // every slice starts 16-byte aligned (parameter allocations, 64-element gradient and state slots)
    const float4 p4 = ld4(p), g4 = ld4(g), m4 = ld4(exp_avg + so), v4 = ld4(exp_avg_sq + so);
    pv[0] = p4.x; pv[1] = p4.y; pv[2] = p4.z; pv[3] = p4.w;
    gv[0] = g4.x; gv[1] = g4.y; gv[2] = g4.z; gv[3] = g4.w;
    mv[0] = m4.x; mv[1] = m4.y; mv[2] = m4.z; mv[3] = m4.w;
    vv[0] = v4.x; vv[1] = v4.y; vv[2] = v4.z; vv[3] = v4.w;
  } else {
    for (int e = 0; e < cnt; ++e) { pv[e] = p[e]; gv[e] = g[e]; mv[e] = exp_avg[so + e]; vv[e] = exp_avg_sq[so + e]; }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (e >= cnt) break;
    const float gr = gv[e] * clip;
    float pr = pv[e];
    pr -= lr_wd * pr;                                         // decoupled weight decay
    const float m = mv[e] + omb1 * (gr - mv[e]);              // lerp(exp_avg, grad, 1 - beta1)
    const float v = beta2 * vv[e] + omb2 * gr * gr;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    pr -= step_size * m / denom;
    pv[e] = pr; mv[e] = m; vv[e] = v;
  }
  if (cnt == 4) {
    st4(p, make_float4(pv[0], pv[1], pv[2], pv[3]));
    st4(exp_avg + so, make_float4(mv[0], mv[1], mv[2], mv[3]));
    st4(exp_avg_sq + so, make_float4(vv[0], vv[1], vv[2], vv[3]));
  } else {
    for (int e = 0; e < cnt; ++e) { p[e] = pv[e]; exp_avg[so + e] = mv[e]; exp_avg_sq[so + e] = vv[e]; }
  }
}
}  // namespace
}  // namespace sepr

using namespace sepr;

extern "C" int sepr_adamw_block_elems(void) { return OPT_BLOCK; }
extern "C" size_t sepr_adamw_workspace(void) { return OPT_PARTS * sizeof(double); }

extern "C" int sepr_adamw_step(const sepr_adamw_tables* t, const float* grads, long long grads_numel, float* exp_avg, float* exp_avg_sq,
                               double* step, const float* lr, double beta1, double beta2, double eps, double weight_decay, double max_norm,
                               float* scal, void* ws, size_t ws_bytes, sepr_stream_t stream) {
  if (!t || !t->params || !t->grad_off || !t->state_off || !t->numel || !t->blocks || t->nblocks <= 0 || !grads || grads_numel <= 0 ||
      !exp_avg || !exp_avg_sq || !step || !lr || !scal)
    return SEPR_EINVAL;
  if (!(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0) || !(weight_decay >= 0.0)) return SEPR_EINVAL;
  if ((reinterpret_cast<uintptr_t>(grads) & 15) || (reinterpret_cast<uintptr_t>(exp_avg) & 15) || (reinterpret_cast<uintptr_t>(exp_avg_sq) & 15))
    return SEPR_EINVAL;
  const int with_norm = max_norm > 0.0 ? 1 : 0;
  if (with_norm && (!ws || ws_bytes < sepr_adamw_workspace())) return SEPR_EWORKSPACE;
  hipStream_t s = static_cast<hipStream_t>(stream);
  double* part = static_cast<double*>(ws);
  if (with_norm) hipLaunchKernelGGL(opt_sqsum_kernel, dim3(OPT_PARTS), dim3(OPT_TPB), 0, s, grads, grads_numel, part);
  hipLaunchKernelGGL(opt_prep_kernel, dim3(1), dim3(OPT_TPB), 0, s, part, with_norm, (float)max_norm, lr, beta1, beta2, (float)weight_decay, step, scal);
  OptTables k;
  k.params = t->params; k.grad_off = t->grad_off; k.state_off = t->state_off; k.numel = t->numel;
  k.blocks = reinterpret_cast<const int2*>(t->blocks);
  hipLaunchKernelGGL(opt_adamw_kernel, dim3(t->nblocks), dim3(OPT_TPB), 0, s, k, grads, exp_avg, exp_avg_sq, scal, (float)(1.0 - beta1), (float)beta2,
                     (float)(1.0 - beta2), (float)eps);
  SEPR_CHECK_LAUNCH("adamw step kernels");
  return SEPR_OK;
}
