// Hardware probe (GPU box): does an LDS-DMA operand ring deliver the k = 1 kernels' weight stream faster than VGPR staging?
// Models gemm_pwln_kernel<4,4,1,1> (513 x 1300 links): one 8-wave workgroup per CU, every wave streams ITS 4 row blocks of both fp16 planes
// (hi, lo) of a 512 x 544 weight -- 8 pieces of 1 KB per 32-channel chunk and wave, 64 KB per chunk and CU, 17 chunks per tile, every workgroup the
// same 1.1 MB (L2-resident) -- and, optionally, issues the tile's MFMAs on what it loaded (NT = 4 column blocks: 12 MFMAs per row block and chunk).
//   mode V<S>: VGPR staging, S fragment sets of 8 KB per wave (S = 1: the shipped kernel's rolling re-load; S = 2: a whole chunk ahead)
//   mode D<R>: per-wave LDS ring of R 1-KB slots filled by global_load_lds_dwordx4 (no VGPR destination), counted vmcnt, ds_read_b128 back
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/ldsdma_ring.hip -o tools/probe/ldsdma_ring
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

#define NCH 17
#define RB 4                         // row blocks per wave
#define PIECES (2 * RB)              // 1-KB pieces per chunk and wave (hi, lo of every row block)

__device__ __forceinline__ f32x4 mma(const uint4& a, const uint4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// piece p of chunk ch for (wave): plane (p & 1), row block wave * RB + (p >> 1)
__device__ __forceinline__ long piece_off(int wave, int ch, int p, long plane_bytes) {
  return (long)(p & 1) * plane_bytes + ((long)(wave * RB + (p >> 1)) * NCH + ch) * 1024;
}

// ---- VGPR staging -----------------------------------------------------------------------------------------------------
template <int SETS, int MFMA>
__global__ __launch_bounds__(512, 2) void vgpr_kernel(const char* __restrict__ w, long plane_bytes, int tiles, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint4 f[SETS][PIECES];
  f32x4 acc[RB][4];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const uint4 bfrag = make_uint4(lane, lane + 1, lane + 2, lane + 3);
  unsigned x = 0;
  auto load = [&](int set, int p, int ch) { f[set][p] = *reinterpret_cast<const uint4*>(w + piece_off(wave, ch, p, plane_bytes) + lane * 16); };
  const int total = tiles * NCH;
#pragma unroll
  for (int s = 0; s < SETS; ++s)
#pragma unroll
    for (int p = 0; p < PIECES; ++p) load(s, p, s % NCH);
  for (int c = 0; c < total; c += SETS) {
#pragma unroll
    for (int s = 0; s < SETS; ++s) {
      const int nxt = (c + s + SETS) % NCH;
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        if constexpr (MFMA) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            acc[i][t] = mma(f[s][2 * i + 1], bfrag, acc[i][t]);
            acc[i][t] = mma(f[s][2 * i], bfrag, acc[i][t]);
            acc[i][t] = mma(f[s][2 * i], bfrag, acc[i][t]);
          }
        } else {
          x ^= f[s][2 * i].x ^ f[s][2 * i].w ^ f[s][2 * i + 1].x ^ f[s][2 * i + 1].w;
        }
        __builtin_amdgcn_sched_barrier(0);
        load(s, 2 * i, nxt); load(s, 2 * i + 1, nxt);           // the rolling re-load: row block i's fragments of a later chunk
      }
    }
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) r += acc[i][t][0] + acc[i][t][3];
  if (r == 12345.678f || x == 0x9e3779b9u) out[blockIdx.x] = r;
}

// ---- per-wave LDS-DMA ring ----------------------------------------------------------------------------------------------
__device__ __forceinline__ void glds16(const char* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int R, int MFMA>
__global__ __launch_bounds__(512, 2) void dma_kernel(const char* __restrict__ w, long plane_bytes, int tiles, float* __restrict__ out) {
  extern __shared__ uint4 ring[];                       // [wave][R][64 lanes]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint4* mine = ring + (size_t)wave * R * 64;
  const unsigned lds_base = (unsigned)(size_t)mine;     // LDS byte address of the wave's ring (low 32 bits of the shared pointer)
  f32x4 acc[RB][4];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const uint4 bfrag = make_uint4(lane, lane + 1, lane + 2, lane + 3);
  unsigned x = 0;
  static_assert(R % 2 == 0 && R >= 2 && R <= 32, "ring of whole (hi, lo) pairs");
  const int total_pairs = tiles * NCH * RB;             // one pair = the hi and lo piece of one row block of one chunk
  auto issue_pair = [&](int q) {                        // pair q -> slots (2 q) % R, (2 q + 1) % R
    const int ch = (q / RB) % NCH, i = q % RB;
    const int s0 = (2 * q) % R;
    glds16(w + piece_off(wave, ch, 2 * i, plane_bytes) + lane * 16, __builtin_amdgcn_readfirstlane(lds_base + s0 * 1024));
    glds16(w + piece_off(wave, ch, 2 * i + 1, plane_bytes) + lane * 16, __builtin_amdgcn_readfirstlane(lds_base + (s0 + 1) * 1024));
  };
#pragma unroll
  for (int q = 0; q < R / 2; ++q) issue_pair(q);
  for (int q0 = 0; q0 < total_pairs; q0 += RB) {
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int q = q0 + i;
      wait_vm<R - 2>();                                 // all but the R / 2 - 1 youngest pairs have landed: pair q is in LDS
      const int s0 = (2 * q) % R;
      const uint4 hi = mine[s0 * 64 + lane], lo = mine[(s0 + 1) * 64 + lane];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      issue_pair(q + R / 2);                            // refill the slots just read (the stream runs past the end: same addresses, harmless)
      if constexpr (MFMA) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc[i][t] = mma(lo, bfrag, acc[i][t]);
          acc[i][t] = mma(hi, bfrag, acc[i][t]);
          acc[i][t] = mma(hi, bfrag, acc[i][t]);
        }
      } else {
        x ^= hi.x ^ hi.w ^ lo.x ^ lo.w;
      }
    }
  }
  wait_vm<0>();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) r += acc[i][t][0] + acc[i][t][3];
  if (r == 12345.678f || x == 0x9e3779b9u) out[blockIdx.x] = r;
}

template <typename K>
static double timed(K launch, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); launch();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); exit(1); }
  return ms / reps * 1e3;
}

int main() {
  const long plane_bytes = 32L * NCH * 1024;           // 32 row blocks x 17 chunks x 1 KB = 557 KB per plane
  const int tiles = 6, nwg = 256;
  char* w; float* out;
  hipMalloc(&w, 2 * plane_bytes + 4096);
  hipMalloc(&out, 4096);
  hipMemset(w, 0, 2 * plane_bytes + 4096);
  const double bytes_per_cu = (double)tiles * NCH * 8 * PIECES * 1024;
  const double mfma_us = (double)tiles * NCH * 2 * RB * 12 * 16 / 2.4e3;        // per SIMD: 2 waves x 48 MFMAs x 16 cycles per chunk at 2.4 GHz
  printf("weight stream of gemm_pwln<4,4,1,1>: %d workgroups x 8 waves, %d tiles x %d chunks x 64 KB per CU; MFMA floor %.1f us\n", nwg, tiles, NCH, mfma_us);
#define RUN_V(S, M) { double us = timed([&] { hipLaunchKernelGGL((vgpr_kernel<S, M>), dim3(nwg), dim3(512), 0, 0, w, plane_bytes, tiles, out); }, 5); \
    printf("VGPR staging, %d set(s) (%2d KB in flight per wave), MFMAs %s: %7.1f us  %5.1f GB/s per CU  %.2f us per chunk\n", S, 8 * S, M ? "on " : "off", us, bytes_per_cu / us / 1e3, us / (tiles * NCH)); fflush(stdout); }
#define RUN_D(R, M) { double us = timed([&] { hipLaunchKernelGGL((dma_kernel<R, M>), dim3(nwg), dim3(512), (size_t)8 * R * 1024, 0, w, plane_bytes, tiles, out); }, 5); \
    printf("LDS-DMA ring, %2d slots per wave (%3d KB of LDS), MFMAs %s: %7.1f us  %5.1f GB/s per CU  %.2f us per chunk\n", R, 8 * R, M ? "on " : "off", us, bytes_per_cu / us / 1e3, us / (tiles * NCH)); fflush(stdout); }
  hipFuncSetAttribute((const void*)dma_kernel<16, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)dma_kernel<16, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)dma_kernel<18, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)dma_kernel<18, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)dma_kernel<12, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)dma_kernel<12, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)dma_kernel<8, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)dma_kernel<8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  RUN_V(1, 0) RUN_V(2, 0) RUN_V(1, 1) RUN_V(2, 1)
  RUN_D(8, 0) RUN_D(12, 0) RUN_D(16, 0) RUN_D(18, 0)
  RUN_D(8, 1) RUN_D(12, 1) RUN_D(16, 1) RUN_D(18, 1)
  return 0;
}
