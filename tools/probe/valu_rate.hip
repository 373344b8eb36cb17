// Hardware probe (GPU box): issue cost of the operand-split instructions.  One wave per SIMD (256 threads, 1 workgroup per CU), N
// independent instructions of one kind between two s_memtime reads; prints cycles per instruction.
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/valu_rate.hip -o tools/probe/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))
__global__ void k(float* out, unsigned long long* cyc, float s) {
  float a0 = threadIdx.x * 1.25f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
  unsigned r0 = 0, r1 = 0, r2 = 0, r3 = 0;
  typedef float f2v __attribute__((ext_vector_type(2)));
  f2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a3}, p3 = {a0, a2};
  unsigned long long t[9];
  t[0] = __builtin_readcyclecounter();
  REP64(asm volatile("v_fma_mixlo_f16 %0, %4, %8, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %1, %5, %8, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %2, %6, %8, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]\n v_fma_mixlo_f16 %3, %7, %8, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "s"(s));)
  t[1] = __builtin_readcyclecounter();
  REP64(asm volatile("v_fma_mixhi_f16 %0, %4, %8, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]\n v_fma_mixhi_f16 %1, %5, %8, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]\n v_fma_mixhi_f16 %2, %6, %8, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]\n v_fma_mixhi_f16 %3, %7, %8, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "s"(s));)
  t[2] = __builtin_readcyclecounter();
  REP64(asm volatile("v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "s"(s));)
  t[3] = __builtin_readcyclecounter();
  REP64(asm volatile("v_cvt_pkrtz_f16_f32 %0, %4, %5\n v_cvt_pkrtz_f16_f32 %1, %5, %6\n v_cvt_pkrtz_f16_f32 %2, %6, %7\n v_cvt_pkrtz_f16_f32 %3, %7, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
  t[4] = __builtin_readcyclecounter();
  REP64(asm volatile("v_fma_mixlo_f16 %0, %4, %8, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n v_fma_mixlo_f16 %1, %5, %8, -%1 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n v_fma_mixlo_f16 %2, %6, %8, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n v_fma_mixlo_f16 %3, %7, %8, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "s"(s));)
  t[5] = __builtin_readcyclecounter();
  REP64(asm volatile("v_pk_mul_f32 %0, %0, %4 op_sel_hi:[1,0]\n v_pk_mul_f32 %1, %1, %4 op_sel_hi:[1,0]\n v_pk_mul_f32 %2, %2, %4 op_sel_hi:[1,0]\n v_pk_mul_f32 %3, %3, %4 op_sel_hi:[1,0]" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(p0));)
  t[6] = __builtin_readcyclecounter();
  REP64(asm volatile("v_cvt_pk_f16_f32 %0, %4, %5\n v_cvt_pk_f16_f32 %1, %5, %6\n v_cvt_pk_f16_f32 %2, %6, %7\n v_cvt_pk_f16_f32 %3, %7, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
  t[7] = __builtin_readcyclecounter();
  REP64(asm volatile("v_cndmask_b32 %0, 0, %4, vcc\n v_cndmask_b32 %1, 0, %5, vcc\n v_cndmask_b32 %2, 0, %6, vcc\n v_cndmask_b32 %3, 0, %7, vcc" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");)
  t[8] = __builtin_readcyclecounter();
  a0 += p0.x + p1.y + p2.x + p3.y;
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + __builtin_bit_cast(float, r0 ^ r1 ^ r2 ^ r3);
  if (blockIdx.x == 0 && threadIdx.x == 0) for (int i = 0; i < 8; ++i) cyc[i] = t[i + 1] - t[i];
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4 * 8); hipMalloc(&cyc, 64);
  for (int wgs : {256, 512, 1024}) {
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, out, cyc, 1024.f);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, out, cyc, 1024.f);
    unsigned long long h[8];
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    const char* nm[8] = {"v_fma_mixlo_f16", "v_fma_mixhi_f16", "v_fma_f32", "v_cvt_pkrtz_f16_f32", "v_fma_mixlo_f16 (f16 addend)", "v_pk_mul_f32", "v_cvt_pk_f16_f32", "v_cndmask_b32"};
    printf("%d workgroups of 256 (%d waves per SIMD):\n", wgs, wgs / 256);
    for (int i = 0; i < 8; ++i) printf("  %-30s %.2f cycles per instruction (256 issued back to back by one wave)\n", nm[i], h[i] / 256.0);
  }
  return 0;
}
