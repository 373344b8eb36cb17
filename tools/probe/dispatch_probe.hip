// Hardware probe (GPU box): onto which CU does the dispatcher put workgroup j of an XCD?  A one-round launch (768 workgroups of 256 threads at
// 3 per CU: the footprint of gemm_nn_bf3_kernel<3,1,7>) records XCC_ID and HW_ID per workgroup.  Printed for XCD 0: the CU (se.sh.cu) of the
// XCD-local workgroups j = blockIdx.x >> 3 in launch order, and how many distinct CUs the triples (j, j + 32, j + 64) and (3j, 3j+1, 3j+2) hit.
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/dispatch_probe.hip -o tools/probe/dispatch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
struct Rec { unsigned xcc, hwid; unsigned long long t; };
template <int REGS>
__global__ __launch_bounds__(256) void probe(Rec* rec, float* sink, int spin) {
  extern __shared__ float sm[];
  float r[REGS];
#pragma unroll
  for (int i = 0; i < REGS; ++i) r[i] = threadIdx.x * 0.5f + i;
  sm[threadIdx.x] = r[0];
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  float s = sm[(threadIdx.x + 1) & 255];
  for (int k = 0; k < spin; ++k)
#pragma unroll
    for (int i = 0; i < REGS; ++i) s = s * 1.0001f + r[i];
  if (spin < 0) sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    Rec q;
    q.xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf;     // HW_REG_XCC_ID[3:0]
    q.hwid = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);          // HW_REG_HW_ID
    q.t = t0;
    rec[blockIdx.x] = q;
  }
  if (s == 12345.678f) sink[0] = s;
}
template <int REGS>
static void run(int wgs, int lds, int spin) {
  Rec* rec; float* sink;
  hipMalloc(&rec, sizeof(Rec) * wgs); hipMalloc(&sink, 4 << 20);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((probe<REGS>), dim3(wgs), dim3(256), lds, 0, rec, sink, spin);
  hipDeviceSynchronize();
  std::vector<Rec> h(wgs);
  hipMemcpy(h.data(), rec, sizeof(Rec) * wgs, hipMemcpyDeviceToHost);
  auto cu = [&](int lin) { const unsigned w = h[lin].hwid; return (int)(((w >> 13) & 7) * 100 + ((w >> 12) & 1) * 50 + ((w >> 8) & 15)); };
  printf("== %d workgroups, %d KB LDS, ~%d VGPRs, spin %d\n", wgs, lds / 1024, REGS + 8, spin);
  std::set<int> xccs; for (int i = 0; i < wgs; ++i) xccs.insert(h[i].xcc);
  printf("XCC ids seen: "); for (int x : xccs) printf("%d ", x); printf("\n");
  int mism = 0; for (int i = 0; i < wgs; ++i) if ((int)h[i].xcc != (i & 7) && 0) ++mism;
  // is blockIdx & 7 a fixed XCC?
  std::map<int, std::set<int>> xofm; for (int i = 0; i < wgs; ++i) xofm[i & 7].insert(h[i].xcc);
  printf("blockIdx & 7 -> XCC: "); for (auto& kv : xofm) { printf("%d:{", kv.first); for (int x : kv.second) printf("%d", x); printf("} "); } printf("\n");
  const int q = wgs / 8;
  printf("XCD of blockIdx 0: CU (se*100 + sh*50 + cu) of local workgroup j, in j order:\n");
  std::set<int> cus;
  for (int j = 0; j < q; ++j) { printf("%4d", cu(8 * j)); cus.insert(cu(8 * j)); if (j % 32 == 31) printf("\n"); }
  printf("\ndistinct CUs on that XCD: %zu\n", cus.size());
  if (q >= 96) {
    int same_stride = 0, same_consec = 0;
    for (int j = 0; j < 32; ++j) { std::set<int> a{cu(8 * j), cu(8 * (j + 32)), cu(8 * (j + 64))}; if (a.size() == 1) ++same_stride; }
    for (int j = 0; j < 32; ++j) { std::set<int> a{cu(8 * 3 * j), cu(8 * (3 * j + 1)), cu(8 * (3 * j + 2))}; if (a.size() == 1) ++same_consec; }
    printf("triples (j, j+32, j+64) on ONE CU: %d of 32;  triples (3j, 3j+1, 3j+2) on ONE CU: %d of 32\n", same_stride, same_consec);
  }
  unsigned long long tmin = ~0ull; for (auto& r : h) tmin = r.t < tmin ? r.t : tmin;
  printf("start time of local workgroup j on that XCD (us after the first of the launch):\n");
  for (int j = 0; j < q; ++j) { printf("%5.1f", (h[8 * j].t - tmin) * 0.01); if (j % 32 == 31) printf("\n"); }
  printf("\n"); fflush(stdout);
  hipFree(rec); hipFree(sink);
}
int main() {
  run<72>(768, 37 * 1024, 2000);     // one round at 3 per CU
  run<72>(512, 37 * 1024, 2000);     // 2 per CU
  run<72>(1536, 37 * 1024, 500);     // two rounds
  run<150>(512, 60 * 1024, 2000);    // the 128-row tile's footprint: 2 per CU
  return 0;
}
