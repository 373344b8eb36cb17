// Hardware probe (GPU box): does split work hide behind a wave's own MFMAs?  One wave per SIMD; 256 x v_mfma_f32_16x16x32_f16 on four
// independent accumulators, alone and with 1 / 2 / 3 half-rate v_fma_mixlo_f16 (or 2 / 4 full-rate v_fma_f32) behind each.
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_valu.hip -o tools/probe/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))
#define MF(ACC) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b));
#define MIX(R, A) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(R) : "v"(A), "s"(s));
#define FMA(A) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(A) : "s"(s));
__global__ void k(float* out, unsigned long long* cyc, float s) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.01f + i); b[i] = (_Float16)(i - 3.5f); }
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  unsigned r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  unsigned long long t[8];
  t[0] = __builtin_readcyclecounter();
  REP64(MF(c0) MF(c1) MF(c2) MF(c3))
  t[1] = __builtin_readcyclecounter();
  REP64(MF(c0) MIX(r0, x0) MF(c1) MIX(r1, x1) MF(c2) MIX(r2, x2) MF(c3) MIX(r3, x3))
  t[2] = __builtin_readcyclecounter();
  REP64(MF(c0) MIX(r0, x0) MIX(r1, x1) MF(c1) MIX(r2, x2) MIX(r3, x3) MF(c2) MIX(r0, x1) MIX(r1, x2) MF(c3) MIX(r2, x3) MIX(r3, x0))
  t[3] = __builtin_readcyclecounter();
  REP64(MF(c0) MIX(r0, x0) MIX(r1, x1) MIX(r2, x2) MF(c1) MIX(r3, x3) MIX(r0, x1) MIX(r1, x2) MF(c2) MIX(r2, x3) MIX(r3, x0) MIX(r0, x2) MF(c3) MIX(r1, x3) MIX(r2, x0) MIX(r3, x1))
  t[4] = __builtin_readcyclecounter();
  REP64(MF(c0) FMA(x0) FMA(x1) MF(c1) FMA(x2) FMA(x3) MF(c2) FMA(x0) FMA(x1) MF(c3) FMA(x2) FMA(x3))
  t[5] = __builtin_readcyclecounter();
  REP64(MF(c0) FMA(x0) FMA(x1) FMA(x2) FMA(x3) MF(c1) FMA(x0) FMA(x1) FMA(x2) FMA(x3) MF(c2) FMA(x0) FMA(x1) FMA(x2) FMA(x3) MF(c3) FMA(x0) FMA(x1) FMA(x2) FMA(x3))
  t[6] = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + x0 + x1 + x2 + x3 + __builtin_bit_cast(float, r0 ^ r1 ^ r2 ^ r3);
  if (blockIdx.x == 0 && threadIdx.x == 0) for (int i = 0; i < 6; ++i) cyc[i] = t[i + 1] - t[i];
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 64);
  const char* nm[6] = {"MFMA alone", "MFMA + 1 v_fma_mixlo_f16", "MFMA + 2 v_fma_mixlo_f16", "MFMA + 3 v_fma_mixlo_f16", "MFMA + 2 v_fma_f32", "MFMA + 4 v_fma_f32"};
  for (int wgs : {256, 512}) {
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, out, cyc, 1024.f);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, out, cyc, 1024.f);
    unsigned long long h[6];
    hipMemcpy(h, cyc, 48, hipMemcpyDeviceToHost);
    printf("%d waves per SIMD:\n", wgs / 256);
    for (int i = 0; i < 6; ++i) printf("  %-28s %.1f cycles per MFMA\n", nm[i], h[i] / 256.0);
  }
  return 0;
}
