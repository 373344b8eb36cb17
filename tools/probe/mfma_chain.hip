// Hardware probe (GPU box): does a chain of v_mfma_f32_16x16x32_f16 on ONE accumulator run at the rate of independent accumulators?
// 256 MFMAs per variant: the same accumulator every time; groups of 3 on one accumulator (the split product's a_lo b_hi, a_hi b_lo, a_hi b_hi),
// round-robin over 4 accumulators; 2 / 4 / 8 accumulators round-robin.  1, 2 and 3 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_chain.hip -o tools/probe/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define REP4(X) X X X X
#define REP8(X) X X X X X X X X
#define REP16(X) REP4(REP4(X))
#define REP32(X) REP8(REP4(X))
#define REP64(X) REP8(REP8(X))
#define MF(ACC) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b));
__global__ void k(float* out, unsigned long long* cyc) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.01f + i); b[i] = (_Float16)(i - 3.5f); }
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
  unsigned long long t[8];
  t[0] = __builtin_readcyclecounter();
  REP64(REP4(MF(c0)))
  t[1] = __builtin_readcyclecounter();
  REP16(MF(c0) MF(c0) MF(c0) MF(c1) MF(c1) MF(c1) MF(c2) MF(c2) MF(c2) MF(c3) MF(c3) MF(c3) MF(c0) MF(c0) MF(c0) MF(c1))      // 256 MFMAs, triples
  t[2] = __builtin_readcyclecounter();
  REP64(MF(c0) MF(c1) MF(c0) MF(c1))
  t[3] = __builtin_readcyclecounter();
  REP64(MF(c0) MF(c1) MF(c2) MF(c3))
  t[4] = __builtin_readcyclecounter();
  REP32(MF(c0) MF(c1) MF(c2) MF(c3) MF(c4) MF(c5) MF(c6) MF(c7))
  t[5] = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3];
  if (blockIdx.x == 0 && threadIdx.x == 0) for (int i = 0; i < 5; ++i) cyc[i] = t[i + 1] - t[i];
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 1024 * 4); hipMalloc(&cyc, 64);
  const char* nm[5] = {"one accumulator", "triples on one accumulator, 4 accumulators in turn", "2 accumulators alternating", "4 accumulators round-robin", "8 accumulators round-robin"};
  for (int waves : {1, 2, 3}) {
    hipLaunchKernelGGL(k, dim3(256), dim3(256 * waves), 0, 0, out, cyc);
    hipLaunchKernelGGL(k, dim3(256), dim3(256 * waves), 0, 0, out, cyc);
    unsigned long long h[5];
    hipMemcpy(h, cyc, 40, hipMemcpyDeviceToHost);
    printf("%d wave(s) per SIMD (shader-clock cycles per MFMA of wave 0, while every wave of the CU runs the same stream):\n", waves);
    for (int i = 0; i < 5; ++i) printf("  %-52s %.1f\n", nm[i], h[i] / 256.0);
  }
  return 0;
}
