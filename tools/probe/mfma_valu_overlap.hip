// How do a dense MFMA stream and a VALU stream share one SIMD?  8 waves per workgroup (wave w and w+4 share a SIMD), one
// workgroup per CU.  Waves 0-3 run NM back-to-back v_mfma_f32_16x16x32_bf16 (two independent accumulator chains, like the
// GCFN kernel), waves 4-7 run NV VALU instructions of a chosen kind.  Reported: cycles (s_memtime) of each role alone and
// together.  mode bit 0: MFMA waves active, bit 1: VALU waves active; swap = roles exchanged (VALU on the older waves).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, bool ACC_AGPR>
__global__ __launch_bounds__(512, 2) void k(unsigned long long* out, float* sink, int mode, int swap, int iters) {
  __shared__ float big[30000];
  big[threadIdx.x] = 0.f;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool mfma_role = swap ? (w >= 4) : (w < 4);
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  auto valu_body = [&]() {
      float x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3, x4 = 1.5f, x5 = 0.5f, x6 = 0.25f, x7 = 2.f;
      float y0 = 1, y1 = 2, y2 = 3, y3 = 4, z0 = 1, z1 = 2, z2 = 3, z3 = 4, u0 = 5, u1 = 6, u2 = 7, u3 = 8;
      for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {   // 8 independent v_fma_f32
          asm volatile(
              "v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5\n\t"
              "v_fma_f32 %0, %0, %6, %7\n\tv_fma_f32 %1, %1, %6, %7\n\tv_fma_f32 %2, %2, %6, %7\n\tv_fma_f32 %3, %3, %6, %7\n\t"
              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(x4), "v"(x5), "v"(x6), "v"(x7));
        } else if (KIND == 1) {   // 8 v_pk_fma_f32 (pairs x0:x1, x2:x3)
          asm volatile(
              "v_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %1, %1, %2, %3\n\tv_pk_fma_f32 %0, %0, %3, %2\n\tv_pk_fma_f32 %1, %1, %3, %2\n\t"
              "v_pk_fma_f32 %0, %0, %2, %3\n\tv_pk_fma_f32 %1, %1, %2, %3\n\tv_pk_fma_f32 %0, %0, %3, %2\n\tv_pk_fma_f32 %1, %1, %3, %2\n\t"
              : "+v"(*(double*)&x0), "+v"(*(double*)&x2) : "v"(*(double*)&x4), "v"(*(double*)&x6));
        } else if (KIND == 2) {   // 8 transcendentals
          asm volatile(
              "v_exp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_rcp_f32 %3, %3\n\t"
              "v_exp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_rcp_f32 %3, %3\n\t"
              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        } else if (KIND == 4) {   // 16 independent v_fma_f32 (dependency distance 16)
          asm volatile(
              "v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5\n\t"
              "v_fma_f32 %8, %8, %6, %7\n\tv_fma_f32 %9, %9, %6, %7\n\tv_fma_f32 %10, %10, %6, %7\n\tv_fma_f32 %11, %11, %6, %7\n\t"
              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(x4), "v"(x5), "v"(x6), "v"(x7), "v"(y0), "v"(y1), "v"(y2), "v"(y3));
          asm volatile(
              "v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5\n\t"
              : "+v"(z0), "+v"(z1), "+v"(z2), "+v"(z3) : "v"(x4), "v"(x5));
          asm volatile(
              "v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5\n\t"
              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(x6), "v"(x7));
        } else if (KIND == 3) {   // 8 DPP moves / cndmask mix
          asm volatile(
              "v_mov_b32_dpp %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n\tv_cndmask_b32 %2, %2, %3, vcc\n\t"
              "v_mov_b32_dpp %1, %0 row_ror:15 row_mask:0xf bank_mask:0xf\n\tv_cndmask_b32 %3, %3, %2, vcc\n\t"
              "v_mov_b32_dpp %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n\tv_cndmask_b32 %2, %2, %3, vcc\n\t"
              "v_mov_b32_dpp %1, %0 row_ror:15 row_mask:0xf bank_mask:0xf\n\tv_cndmask_b32 %3, %3, %2, vcc\n\t"
              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) :: "vcc");
        }
      }
      sink[threadIdx.x] = x0 + x1 + x2 + x3 + y0 + y1 + y2 + y3 + z0 + z1 + z2 + z3 + u0 + u1 + u2 + u3;
  };
  if (mfma_role) {
    if (mode & 4) valu_body();
    if (mode & 1) {
      bf16x8 a, b;
      for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
      f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
      for (int it = 0; it < iters; ++it) {
        if (ACC_AGPR) {
          asm volatile(
              "v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n\t"
              "v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n\t"
              "v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n\t"
              "v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n\t"
              : "+a"(c0), "+a"(c1) : "v"(a), "v"(b));
        } else {
          asm volatile(
              "v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n\t"
              "v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n\t"
              "v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n\t"
              "v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %2, %3, %1\n\t"
              : "+v"(c0), "+v"(c1) : "v"(a), "v"(b));
        }
      }
      asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
      sink[threadIdx.x] = c0[0] + c1[1];
    }
  } else {
    if (mode & 2) valu_body();
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 8 + w] = t1 - t0;
  if (big[threadIdx.x] != 0.f) sink[0] = 1.f;
}

template <int KIND, bool AG>
void run(const char* name, int iters) {
  const double ipi = KIND == 4 ? 16.0 : 8.0;
  unsigned long long* d; float* sink;
  const int nb = 256;
  (void)hipMalloc(&d, nb * 8 * 8); (void)hipMalloc(&sink, 512 * 4);
  static unsigned long long h[256 * 8];
  for (int swap = 0; swap < 2; ++swap) {
    double res[8][2] = {};
    for (int mode : {1, 2, 3, 6}) {
      for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<KIND, AG>), dim3(nb), dim3(512), 0, 0, d, sink, mode, swap, iters);
      (void)hipDeviceSynchronize();
      (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      double m = 0, v = 0;
      for (int b = 0; b < nb; ++b)
        for (int w = 0; w < 8; ++w) {
          const bool mr = swap ? (w >= 4) : (w < 4);
          (mr ? m : v) += (double)h[b * 8 + w] / (nb * 4);
        }
      res[mode][0] = m; res[mode][1] = v;
    }
    printf("%-26s acc=%s %s | MFMA alone %7.0f cyc (%.1f/MFMA) | VALU alone %7.0f cyc (%.1f/inst) | together: MFMA %7.0f VALU %7.0f | VALU on BOTH waves of a SIMD: %7.0f / %7.0f cyc (%.1f/inst/wave)\n",
           name, AG ? "AGPR" : "VGPR", swap ? "VALU waves older" : "MFMA waves older", res[1][0], res[1][0] / (8.0 * iters), res[2][1],
           res[2][1] / (ipi * iters), res[3][0], res[3][1], res[6][0], res[6][1], res[6][1] / (ipi * iters));
  }
  (void)hipFree(d); (void)hipFree(sink);
}

int main() {
  const int iters = 200;   // 1600 MFMAs vs 1600 VALU instructions
  run<0, false>("v_fma_f32", iters);
  run<0, true>("v_fma_f32", iters);
  run<1, false>("v_pk_fma_f32", iters);
  run<1, true>("v_pk_fma_f32", iters);
  run<2, false>("v_exp/v_rcp", iters);
  run<3, false>("dpp mov + cndmask", iters);
  run<4, false>("v_fma_f32 x16 independent", iters);
  return 0;
}
