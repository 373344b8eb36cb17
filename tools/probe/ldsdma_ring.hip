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


// ---- the whole chunk loop of gemm_pwln_kernel<4,4,1,1>, added stage by stage ---------------------------------------------------------------------
// weights: VGPR rolling re-load (above).  XFRAG: the 8 input fragments of a 32-channel sub-chunk come from an LDS image (ds_read_b128).  BARRIER: one
// __syncthreads per chunk.  STAGE: the input tile really is staged -- raw fp32 loads one chunk ahead, split into fp16 hi / lo, two ds_write_b128 per
// slot -- into the other of two images.  BK: channels per chunk (32 = the shipped kernel: 256 of the 512 threads stage; 64: two sub-chunks of MFMAs
// per barrier, all 512 threads stage).
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2v;
__device__ __forceinline__ void split8(const float (&v)[8], float sc, uint4& hi, uint4& lo) {
  unsigned h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = v[2 * i] * sc, b = v[2 * i + 1] * sc;
    const _Float16 ha = (_Float16)a, hb = (_Float16)b;
    const _Float16 la = (_Float16)(a - (float)ha), lb = (_Float16)(b - (float)hb);
    h[i] = (unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
    l[i] = (unsigned)__builtin_bit_cast(unsigned short, la) | ((unsigned)__builtin_bit_cast(unsigned short, lb) << 16);
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]); lo = make_uint4(l[0], l[1], l[2], l[3]);
}
template <int XFRAG, int BARRIER, int STAGE, int BK>
__global__ __launch_bounds__(512, 2) void loop_kernel(const char* __restrict__ w, long plane_bytes, int tiles, const float* __restrict__ X, int Lrow, float* __restrict__ out) {
  constexpr int SUB = BK / 32, SLOTS = 4 * SUB * 64;          // 16-byte slots of one chunk image: [k-group][column]
  __shared__ uint4 img[2][2 * SLOTS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kq = lane >> 4, nq = lane & 15;
  f32x4 acc[RB][4];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  uint4 f[PIECES];
  auto load = [&](int p, int ch) { f[p] = *reinterpret_cast<const uint4*>(w + piece_off(wave, ch, p, plane_bytes) + lane * 16); };
  for (int i = tid; i < 2 * 2 * SLOTS; i += 512) (&img[0][0])[i] = make_uint4(i, i + 1, i + 2, i + 3);
#pragma unroll
  for (int p = 0; p < PIECES; ++p) load(p, 0);
  const bool stager = tid < SLOTS;
  const int skg = tid / 64, scol = tid % 64;
  const float* xcol = X + (long)(blockIdx.x % 640) * 64 + scol;          // this workgroup's 64-column tile of a (K rows x Lrow) matrix
  float rx[8];
  auto prefetchX = [&](int c) {                                          // chunk c: channels c * BK + 8 * skg + i
#pragma unroll
    for (int i = 0; i < 8; ++i) rx[i] = xcol[(long)((c % (NCH / SUB)) * BK + 8 * skg + i) * Lrow];
  };
  auto commitX = [&](int c) {
    if (!stager) return;
    uint4 h, l;
    split8(rx, 16384.f, h, l);
    img[c & 1][tid] = h; img[c & 1][SLOTS + tid] = l;
  };
  // STAGE 2 (BK = 32): all 512 threads stage, 4 channels each (half a slot: one ds_write_b64 per plane)
  const int hslot = tid >> 1, hhalf = tid & 1;                            // slot = (k-group, column), half = channels 0-3 / 4-7 of the group
  const float* xcolh = X + (long)(blockIdx.x % 640) * 64 + (hslot % 64);
  float rh[4];
  auto prefetchH = [&](int c) {
#pragma unroll
    for (int i = 0; i < 4; ++i) rh[i] = xcolh[(long)((c % NCH) * 32 + 8 * (hslot / 64) + 4 * hhalf + i) * Lrow];
  };
  auto commitH = [&](int c) {
    unsigned h[2], l[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float a = rh[2 * i] * 16384.f, b = rh[2 * i + 1] * 16384.f;
      const _Float16 ha = (_Float16)a, hb = (_Float16)b;
      const _Float16 la = (_Float16)(a - (float)ha), lb = (_Float16)(b - (float)hb);
      h[i] = (unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
      l[i] = (unsigned)__builtin_bit_cast(unsigned short, la) | ((unsigned)__builtin_bit_cast(unsigned short, lb) << 16);
    }
    uint2* ih = reinterpret_cast<uint2*>(&img[c & 1][hslot]) + hhalf;
    uint2* il = reinterpret_cast<uint2*>(&img[c & 1][SLOTS + hslot]) + hhalf;
    *ih = make_uint2(h[0], h[1]); *il = make_uint2(l[0], l[1]);
  };
  if constexpr (STAGE == 1 || STAGE == 3) { if (stager) prefetchX(0); commitX(0); if (stager) prefetchX(1); }
  if constexpr (STAGE == 2) { prefetchH(0); commitH(0); prefetchH(1); }
  unsigned ph[4], pl[4];                                                   // STAGE 3: the commit in pieces
  __syncthreads();
  const int total = tiles * NCH / SUB;                                   // chunks of BK channels
  const uint4 bconst = make_uint4(lane, lane + 1, lane + 2, lane + 3);
  for (int c = 0; c < total; ++c) {
#pragma unroll
    for (int sb = 0; sb < SUB; ++sb) {
      const int ch32 = (c * SUB + sb + 1) % NCH;                         // the 32-channel weight chunk the rolling re-load fetches
      uint4 bh[4], bl[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if constexpr (XFRAG) { const int s_ = (sb * 4 + kq) * 64 + t * 16 + nq; bh[t] = img[c & 1][s_]; bl[t] = img[c & 1][SLOTS + s_]; }
        else { bh[t] = bconst; bl[t] = bconst; }
      }
#pragma unroll
      for (int i = 0; i < RB; ++i) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          acc[i][t] = mma(f[2 * i + 1], bh[t], acc[i][t]);
          acc[i][t] = mma(f[2 * i], bl[t], acc[i][t]);
          acc[i][t] = mma(f[2 * i], bh[t], acc[i][t]);
        }
        __builtin_amdgcn_sched_barrier(0);
        load(2 * i, ch32); load(2 * i + 1, ch32);
        if constexpr (STAGE == 3) {                                        // one piece of the staging behind every row block's MFMAs
          const int slot8 = sb * RB + i;                                   // 0 .. 7 (BK = 64)
          if (slot8 < 4) {
            const float a = rx[2 * slot8] * 16384.f, b = rx[2 * slot8 + 1] * 16384.f;
            const _Float16 ha = (_Float16)a, hb = (_Float16)b;
            const _Float16 la = (_Float16)(a - (float)ha), lb = (_Float16)(b - (float)hb);
            ph[slot8] = (unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
            pl[slot8] = (unsigned)__builtin_bit_cast(unsigned short, la) | ((unsigned)__builtin_bit_cast(unsigned short, lb) << 16);
          } else if (slot8 == 4) img[(c + 1) & 1][tid] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
          else if (slot8 == 5) img[(c + 1) & 1][SLOTS + tid] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
          else {
#pragma unroll
            for (int j = 0; j < 4; ++j) rx[4 * (slot8 - 6) + j] = xcol[(long)(((c + 2) % (NCH / SUB)) * BK + 8 * skg + 4 * (slot8 - 6) + j) * Lrow];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if constexpr (STAGE == 1) { commitX(c + 1); if (stager) prefetchX(c + 2); }
    if constexpr (STAGE == 2) { commitH(c + 1); prefetchH(c + 2); }
    if constexpr (BARRIER) __syncthreads();
  }
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) r += acc[i][t][0] + acc[i][t][3];
  if (r == 12345.678f) out[blockIdx.x] = r;
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
  const int tiles = 8, nwg = 256;          // (tiles x 17 chunks: even, so the 64-channel form divides)
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
  // ---- the chunk loop, stage by stage
  float* X;
  const int Lrow = 640 * 64;
  hipMalloc(&X, (size_t)576 * Lrow * 4);
  hipMemset(X, 0, (size_t)576 * Lrow * 4);
#define RUN_L(XF, BA, ST, BK_) { double us = timed([&] { hipLaunchKernelGGL((loop_kernel<XF, BA, ST, BK_>), dim3(nwg), dim3(512), 0, 0, w, plane_bytes, tiles, X, Lrow, out); }, 5); \
    printf("chunk loop: weights + MFMAs%s%s%s, %d channels per barrier: %7.1f us  %.2f us per 32-channel chunk\n", XF ? " + input fragments from LDS" : "", BA ? " + barrier" : "", \
           ST == 1 ? " + input staged (load, split, ds_write)" : ST == 2 ? " + input staged by all 512 threads (4 channels each)" : ST == 3 ? " + input staged in pieces behind the row blocks" : "", BK_, us, us / (tiles * NCH)); fflush(stdout); }
  RUN_L(0, 0, 0, 32) RUN_L(1, 0, 0, 32) RUN_L(1, 1, 0, 32) RUN_L(1, 1, 1, 32)
  RUN_L(1, 1, 2, 32)
  RUN_L(1, 1, 0, 64) RUN_L(1, 1, 1, 64) RUN_L(1, 1, 3, 64)
  return 0;
}
