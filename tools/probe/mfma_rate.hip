// Hardware probe (GPU box): v_mfma_f32_16x16x32_f16 throughput of the chip against the waves per SIMD that issue them (wall clock, hipEvents).
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_rate.hip -o tools/probe/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))
#define MF(ACC) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b));
__global__ void k(float* out, int reps) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.01f + i); b[i] = (_Float16)(i - 3.5f); }
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int r = 0; r < reps; ++r) { REP64(MF(c0) MF(c1) MF(c2) MF(c3)) }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
int main() {
  float* out; hipMalloc(&out, 4096 * 1024 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 200;
  for (int waves : {1, 2, 3, 4, 8}) {
    const int threads = 256 * (waves > 4 ? 4 : waves), wgs = 256 * (waves > 4 ? waves / 4 : 1);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(threads), 0, 0, out, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(threads), 0, 0, out, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)wgs * (threads / 64) * reps * 256.0;
    printf("%d wave(s) per SIMD: %.3f ms, %.0f TFLOP/s (16x16x32 f16 = 16384 FLOP per wave instruction)\n", waves, ms, mfmas * 16384 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
