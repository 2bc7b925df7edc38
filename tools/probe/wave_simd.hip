// Which SIMD does wave w of a 512-thread (8-wave) workgroup land on?  (HW_REG_HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8])
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(unsigned* out) {
  __shared__ float big[26000];   // ~104 KB: one workgroup per CU, as the GCFN kernel
  big[threadIdx.x] = 0.f;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
  if (big[threadIdx.x] != 0.f) out[0] = 0;
}
int main() {
  unsigned* d;
  const int nb = 512;
  hipMalloc(&d, nb * 8 * 4);
  hipLaunchKernelGGL(k, dim3(nb), dim3(512), 0, 0, d);
  hipDeviceSynchronize();
  static unsigned h[nb * 8];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int hist[8][4] = {};
  for (int b = 0; b < nb; ++b)
    for (int w = 0; w < 8; ++w) hist[w][(h[b * 8 + w] >> 4) & 3]++;
  for (int w = 0; w < 8; ++w) printf("wave %d: simd0 %d simd1 %d simd2 %d simd3 %d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  for (int b = 0; b < 6; ++b) {
    printf("block %d:", b);
    for (int w = 0; w < 8; ++w) printf(" w%d->simd%u(cu%u,slot%u)", w, (h[b * 8 + w] >> 4) & 3, (h[b * 8 + w] >> 8) & 15, h[b * 8 + w] & 15);
    printf("\n");
  }
  return 0;
}
