// Hardware probe (GPU box): what a launch costs before it computes anything -- back-to-back launches (one stream) of a kernel that only
// touches its LDS, for the grid shapes / LDS footprints / register counts of the conv kernels.  us per launch = ramp + drain floor.
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/launch_floor.hip -o tools/probe/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
template <int REGS>
__global__ __launch_bounds__(256) void k(float* out, int n) {
  extern __shared__ float sm[];
  float r[REGS];
#pragma unroll
  for (int i = 0; i < REGS; ++i) r[i] = threadIdx.x * 0.5f + i;
  sm[threadIdx.x] = r[0];
  __syncthreads();
  float s = sm[(threadIdx.x + 1) & 255];
#pragma unroll
  for (int i = 0; i < REGS; ++i) s += r[i] * s;          // keeps the registers allocated
  if (n < 0) out[blockIdx.x * 256 + threadIdx.x] = s;    // never taken: no global traffic
}
template <int REGS>
static void run(int wgs, int lds, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<REGS>), dim3(wgs), dim3(256), lds, 0, out, 1);
  hipEventRecord(e0);
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((k<REGS>), dim3(wgs), dim3(256), lds, 0, out, 1);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%5d workgroups x 256 threads, %2d KB LDS, ~%3d VGPRs: %.2f us per launch\n", wgs, lds / 1024, REGS + 8, ms * 1000 / 200);
}
int main() {
  float* out; hipMalloc(&out, 1 << 20);
  run<8>(256, 1024, out); run<8>(768, 1024, out); run<8>(768, 42 * 1024, out); run<112>(768, 42 * 1024, out);
  run<112>(512, 40 * 1024, out); run<112>(1536, 42 * 1024, out); run<112>(3072, 42 * 1024, out); run<200>(512, 48 * 1024, out);
  return 0;
}
