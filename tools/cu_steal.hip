// Experiment helper (not part of libssv_hip): keeps `wgs` workgroups of 256 threads resident for `cycles` shader cycles, to see
// what a training step loses when a communication library's channels hold that many CU slots (tools/ab_cu_steal.py).
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(256) void spin_kernel(long long cycles, int* sink) {
  const long long t0 = clock64();
  int x = 0;
  while (clock64() - t0 < cycles) { __builtin_amdgcn_s_sleep(32); ++x; }
  if (x == -1) *sink = x;
}
extern "C" int cu_steal(int wgs, long long cycles, void* stream) {
  hipLaunchKernelGGL(spin_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, cycles, (int*)nullptr);
  return (int)hipGetLastError();
}
