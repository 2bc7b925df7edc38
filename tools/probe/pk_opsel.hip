// Stand-alone reproducer (round 5) of the fault behind two "cause not established" incidents of rounds 3-4:
//
//   On gfx950 a packed-f32 VALU instruction whose LOW result half selects the HIGH dword of a source pair - VOP3P op_sel bit set, e.g.
//   `v_pk_mul_f32 v[0:1], v[0:1], v[2:3] op_sel:[0,1]` - returns a WRONG low half in lanes 16-31 and 48-63 whenever ANOTHER wave on the same
//   SIMD (a co-resident workgroup) is executing bf16 MFMAs at that moment.  High halves, lanes 0-15 / 32-47, the op_sel_hi modifiers, the
//   un-modified packed forms and the scalar forms are never wrong; wait states around the instruction do not help; one workgroup per CU
//   (no MFMA of another wave can overlap the packed instruction) never fails.
//
// Every thread reads operands from global memory, evaluates one packed form by inline asm and the same arithmetic with scalar VALU
// instructions on the same registers, and counts bit mismatches per result half and lane group.  Phases alternate like a staged GEMM
// (packed arithmetic -> barrier -> burst of MFMAs -> barrier), so with two or more workgroups per CU one workgroup's packed phase overlaps
// another's MFMA phase.
//     hipcc --offload-arch=gfx950 -O2 pk_opsel.hip -o pk_opsel ;  ./pk_opsel <dynamic LDS bytes per workgroup> <workgroups> <MFMAs per phase> <mfma kind>
//     LDS 76000 -> two workgroups per CU, 100000 -> one, 40000 -> four;  mfma kind 0 = v_mfma_f32_16x16x32_bf16, 1 = v_mfma_f32_16x16x4_f32,
//     2 = v_mfma_f32_32x32x16_bf16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
constexpr int NFORM = 11;
static const char* FORM_NAME[NFORM] = {"v_pk_mul_f32 op_sel:[0,1]", "v_pk_mul_f32 op_sel:[1,0]", "v_pk_fma_f32 op_sel:[1,0,0]", "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]",
                                       "v_pk_mul_f32 op_sel_hi:[0,1]", "v_pk_fma_f32 op_sel_hi:[0,1,1]", "v_pk_mul_f32 (no modifier)",
                                       "v_pk_mov_b32 op_sel:[1,0]", "v_pk_mov_b32 op_sel:[0,1]", "v_pk_mov_b32 op_sel:[1,1]", "v_pk_mov_b32 op_sel:[0,0]"};
template <int FORM>
__device__ __forceinline__ void eval(f2 a, f2 b, f2 c, f2& pk, float& lo, float& hi) {
  pk = c;
  if (FORM == 0) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(pk) : "v"(a), "v"(b)); asm volatile("v_mul_f32 %0, %2, %5\n\tv_mul_f32 %1, %3, %5" : "=&v"(lo), "=&v"(hi) : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1])); }
  if (FORM == 1) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(pk) : "v"(a), "v"(b)); asm volatile("v_mul_f32 %0, %3, %4\n\tv_mul_f32 %1, %3, %5" : "=&v"(lo), "=&v"(hi) : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1])); }
  if (FORM == 2) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(pk) : "v"(a), "v"(b)); lo = c[0]; hi = c[1];
                   asm volatile("v_fma_f32 %0, %3, %4, %0\n\tv_fma_f32 %1, %3, %5, %1" : "+v"(lo), "+v"(hi) : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1])); }
  if (FORM == 3) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(pk) : "v"(a), "v"(b)); asm volatile("v_add_f32 %0, %2, %5\n\tv_add_f32 %1, %3, %4" : "=&v"(lo), "=&v"(hi) : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1])); }
  if (FORM == 4) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(pk) : "v"(a), "v"(b)); asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %2, %5" : "=&v"(lo), "=&v"(hi) : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1])); }
  if (FORM == 5) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(pk) : "v"(a), "v"(b)); lo = c[0]; hi = c[1];
                   asm volatile("v_fma_f32 %0, %2, %4, %0\n\tv_fma_f32 %1, %2, %5, %1" : "+v"(lo), "+v"(hi) : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1])); }
  if (FORM >= 7) {   // v_pk_mov_b32: D.lo = op_sel[0] ? S0.hi : S0.lo,  D.hi = op_sel[1] ? S1.hi : S1.lo
    if (FORM == 7) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(pk) : "v"(a), "v"(b));
    if (FORM == 8) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,1]" : "=v"(pk) : "v"(a), "v"(b));
    if (FORM == 9) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,1]" : "=v"(pk) : "v"(a), "v"(b));
    if (FORM == 10) asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[0,0]" : "=v"(pk) : "v"(a), "v"(b));
    float t0 = (FORM == 7 || FORM == 9) ? a[1] : a[0], t1 = (FORM == 8 || FORM == 9) ? b[1] : b[0];
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(lo), "=&v"(hi) : "v"(t0), "v"(t1));
  }
  if (FORM == 6) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(pk) : "v"(a), "v"(b)); asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %5" : "=&v"(lo), "=&v"(hi) : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1])); }
}
template <int FORM>
__global__ __launch_bounds__(256) void probe(const float4* __restrict__ data, const float2* __restrict__ stats, int rows, int ld4,
                                             unsigned long long* __restrict__ bad, float* __restrict__ sink, int mfmas, int kind) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, cg = tid & 31, mg = tid >> 5;
  unsigned long long mism = 0;      // [15:0] low half, [31:16] high half
  float acc = 0.f;
  for (int r0 = blockIdx.x * 64; r0 + 64 <= rows; r0 += gridDim.x * 64) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int m = r0 + 8 * mg + e;
      const float4 v = data[(long long)m * ld4 + cg];
      const float2 st = stats[m];
      f2 pk;
      float lo, hi;
      eval<FORM>((f2){v.x, v.y}, (f2){st.x, st.y}, (f2){v.z, v.w}, pk, lo, hi);
      mism += (__float_as_uint(pk[0]) != __float_as_uint(lo)) + ((unsigned long long)(__float_as_uint(pk[1]) != __float_as_uint(hi)) << 16);
      lds[(8 * mg + e) * 132 + 4 * cg] = pk[0];
      acc += lds[((8 * mg + e) * 132 + 4 * cg + 64) % (64 * 132)];
    }
    __syncthreads();
    if (mfmas > 0) {
      const bf8 a = *reinterpret_cast<const bf8*>(lds + 4 * (tid & 63)), b = *reinterpret_cast<const bf8*>(lds + 1024 + 4 * (tid & 63));
      f4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0;
      f16v w0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int i = 0; i < mfmas; i += 2) {
        if (kind == 0) { c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c1, 0, 0, 0); }
        if (kind == 1) { c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(acc, lds[tid], c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(lds[tid], acc, c1, 0, 0, 0); }
        if (kind == 2) { w0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, w0, 0, 0, 0); w0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, w0, 0, 0, 0); }
      }
      acc += c0[0] + c1[1] + w0[3];
      __syncthreads();
    }
  }
  atomicAdd(&bad[threadIdx.x & 63], mism);
  if (acc == 12345.f) sink[0] = acc;
}
typedef void (*kern_t)(const float4*, const float2*, int, int, unsigned long long*, float*, int, int);
int main(int argc, char** argv) {
  const int lds = argc > 1 ? atoi(argv[1]) : 76000, grid = argc > 2 ? atoi(argv[2]) : 512, mfmas = argc > 3 ? atoi(argv[3]) : 96, kind = argc > 4 ? atoi(argv[4]) : 0;
  const int rows = 512000, ld4 = 32;
  float4* d; float2* s; unsigned long long* bad; float* sink;
  if (hipMalloc(&d, (size_t)rows * ld4 * 16) != hipSuccess || hipMalloc(&s, (size_t)rows * 8) != hipSuccess || hipMalloc(&bad, 64 * 8) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
  float* h = (float*)malloc((size_t)rows * ld4 * 16); float* hs = (float*)malloc((size_t)rows * 8);
  srand(1);
  for (size_t i = 0; i < (size_t)rows * ld4 * 4; ++i) h[i] = (float)rand() / RAND_MAX * 4.f - 1.5f;
  for (int i = 0; i < rows; ++i) { hs[2 * i] = (float)rand() / RAND_MAX; hs[2 * i + 1] = 0.3f + (float)rand() / RAND_MAX; }
  (void)hipMemcpy(d, h, (size_t)rows * ld4 * 16, hipMemcpyHostToDevice); (void)hipMemcpy(s, hs, (size_t)rows * 8, hipMemcpyHostToDevice);
  const kern_t K[NFORM] = {probe<0>, probe<1>, probe<2>, probe<3>, probe<4>, probe<5>, probe<6>, probe<7>, probe<8>, probe<9>, probe<10>};
  printf("LDS %d B/workgroup, %d workgroups, %d MFMAs per phase of kind %d\n", lds, grid, mfmas, kind);
  for (int f = 0; f < NFORM; ++f) {
    (void)hipFuncSetAttribute((const void*)K[f], hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    int occ = 0; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, K[f], 256, lds);
    (void)hipMemset(bad, 0, 64 * 8);
    hipLaunchKernelGGL(K[f], dim3(grid), dim3(256), lds, 0, d, s, rows, ld4, bad, sink, mfmas, kind);
    unsigned long long hb[64]; (void)hipMemcpy(hb, bad, sizeof(hb), hipMemcpyDeviceToHost);
    unsigned long long lo = 0, hi = 0, la = 0, lb = 0;
    for (int l = 0; l < 64; ++l) { const unsigned long long a = hb[l] & 0xffff, b = (hb[l] >> 16) & 0xffff; lo += a; hi += b; ((l & 16) ? lb : la) += a + b; }
    printf("  %-44s workgroups/CU %d: wrong low halves %6llu  high halves %6llu | lanes 0-15,32-47: %6llu  lanes 16-31,48-63: %6llu  (%s)\n", FORM_NAME[f], occ, lo, hi, la, lb,
           hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
