// Persistent LayerNorm / gate kernels (round 5).  A translation unit of its own because it is compiled with -fno-slp-vectorize: hipcc's SLP
// vectoriser pairs the channel steps of these kernels into v_pk_* instructions, whose operands must sit in adjacent registers -- the copies and
// the stretched live ranges took ln_gate_bwd_pers_kernel<8, 32> from 213 VGPRs to 256 + 154 spilled.
#include <stdlib.h>
#include <type_traits>
#include "ssv_common.h"

#define LN_EPS 1e-5f
// v_rcp_f32 (1 ulp) instead of the IEEE division sequence (two div_scale, rcp, four fma, div_fmas, div_fixup: ten instructions per element)
__device__ __forceinline__ float sigmoidf_(float v) { return __builtin_amdgcn_rcpf(1.f + __expf(-v)); }
__device__ __forceinline__ float col_sum(float v) { return ssv_row16_sum(v); }

// ---- persistent backward (round 5) ----------------------------------------------------------------------------------------
// The launches above are ONE lock-step round: every workgroup reads its tile, then sums, then writes, so a small launch costs
// launch + read phase + sums + write phase + drain whatever the code inside does (DESIGN: 26 us for 74.5 MB at C = 256 / L = 325, with
// 16-column AND with 64-column tiles).  Here a workgroup is resident for the whole launch and owns a contiguous RANGE of columns of one
// batch item -- L / wpi of them, `wpi` workgroups per item chosen so that B * wpi workgroups are one per CU: equal
// bytes per CU, no second round -- and walks it in 16-column sub-tiles with TWO register sets: the loads of sub-tile k + 1 (and its
// column statistics) are issued before sub-tile k's first pass, so its sums, its second pass and its stores run under them.  Same
// arithmetic per element and per column as ln_gate_bwd_kernel up to the sigmoid's reciprocal; the per-channel parameter-gradient partials
// of ALL sub-tiles of a workgroup are accumulated per thread and
// written as ONE partial row per workgroup: B * wpi rows (256) instead of one per tile (672 at L = 325), and one scale-list entry.
// Sum N per-thread values over the G channel groups of the thread's column, G = 4 * NW: the four groups of a wave first (two lane
// exchanges per value), then the NW waves through LDS (`red`: [NW][N][16]) in a fixed order.
template <int NW, int N>
__device__ __forceinline__ void group_sums_waves(float (&v)[N], float* red, int col, int wave) {
#pragma unroll
  for (int k = 0; k < N; ++k) { v[k] += __shfl_xor(v[k], 16); v[k] += __shfl_xor(v[k], 32); }
  __syncthreads();
  if ((threadIdx.x & 63) < 16) {
#pragma unroll
    for (int k = 0; k < N; ++k) red[(wave * N + k) * 16 + col] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; ++k) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[(w * N + k) * 16 + col];
    v[k] = s;
  }
}
template <int CPT, int G, int PLDS>
__global__ __launch_bounds__(16 * G) void ln_gate_bwd_pers_kernel(
    const float* __restrict__ dY, long dy_bs, const float* __restrict__ H, const float* __restrict__ X, long x_bs,
    const float* __restrict__ stats,
    const float* __restrict__ g1, const float* __restrict__ b1, const float* __restrict__ g2, const float* __restrict__ b2,
    float* __restrict__ dH, float* __restrict__ dXres, long dx_bs, float* __restrict__ part, float* __restrict__ amax, int C, int L, int wpi) {
  static_assert(CPT <= 16 && G % 4 == 0, "one accumulator lane per channel step; whole waves");      // C == CPT * G exactly (the launcher checks)
  constexpr int NW = G / 4;
  __shared__ float red[NW * 4 * 16];
  __shared__ float amx[NW];
  __shared__ float4 prm[PLDS ? CPT * G : 1];          // PLDS: (gamma1, beta1, gamma2, beta2) per channel, read per step (frees 4 * CPT registers)
  const int col = threadIdx.x & 15, g = threadIdx.x >> 4, wave = threadIdx.x >> 6;
  const unsigned wg = ssv_xcd_order(blockIdx.x, gridDim.x);
  const int b = (int)(wg / (unsigned)wpi), j = (int)(wg % (unsigned)wpi);
  // sub-tile k of workgroup j is tile k * wpi + j of the item: at any time the wpi workgroups of an item (neighbours in launch order: one XCD)
  // work on ADJACENT 16-column tiles, so the two halves of every 128-byte line are asked for at the same time through the same L2.  (With
  // a contiguous column range per workgroup the other half of each line came due a sub-tile later, ~12 us, by when the L2 had dropped it:
  // 2.1 TB/s at every length against 2.7-3.5 of the tile kernels.)
  const int tiles_item = (L + 15) >> 4;
  const int ntile = (tiles_item - j + wpi - 1) / wpi;
  const int s1 = L;
  // every stream through a buffer resource: uniform base + ONE per-thread byte offset per sub-tile + a scalar row offset per channel step
  // (with plain pointers hipcc kept a 64-bit VGPR pointer per stream and step: 256 VGPRs + 238 AGPRs for a 16-step form of this kernel)
  const __amdgpu_buffer_rsrc_t rdy = ssv_buf(dY + (long)b * dy_bs), rx = ssv_buf(X + (long)b * x_bs);
  const __amdgpu_buffer_rsrc_t rh1 = ssv_buf(H + (long)b * 2 * C * L), rh2 = ssv_buf(H + (long)b * 2 * C * L + (long)C * L);
  const __amdgpu_buffer_rsrc_t wh1 = ssv_buf(dH + (long)b * 2 * C * L), wh2 = ssv_buf(dH + (long)b * 2 * C * L + (long)C * L);
  const __amdgpu_buffer_rsrc_t wdx = ssv_buf(dXres + (long)b * dx_bs);
  const float* __restrict__ sb = stats + (long)b * 4 * L;
  // LayerNorm parameters of this thread's channels: loaded once for the whole range
  float pg1[PLDS ? 1 : CPT], pb1[PLDS ? 1 : CPT], pg2[PLDS ? 1 : CPT], pb2[PLDS ? 1 : CPT];
  if constexpr (PLDS) {
    for (int c = threadIdx.x; c < C; c += 16 * G) prm[c] = make_float4(g1[c], b1[c], g2[c], b2[c]);
    __syncthreads();
  } else {
#pragma unroll
    for (int i = 0; i < CPT; ++i) { const int c = g + G * i; pg1[i] = g1[c]; pb1[i] = b1[c]; pg2[i] = g2[c]; pb2[i] = b2[c]; }
  }
  float raw[2][CPT][4];                 // per set: dy, x, h1, h2 -> (first pass, in place) a1, a2, xhat1, xhat2
  float cst[2][4];                      // mu1, r1, mu2, r2 of the thread's column
  // per-channel parameter-gradient partials: summed over the workgroup's columns PER THREAD (its own column of every sub-tile) and reduced over
  // the 16 column lanes ONCE at the end -- a reduction per channel step (six butterflies of four DPP stages) was a third of the kernel's VALU work
  float pacc[6][CPT];
#pragma unroll
  for (int q = 0; q < 6; ++q)
#pragma unroll
    for (int i = 0; i < CPT; ++i) pacc[q][i] = 0.f;
  float am = 0.f;
  const unsigned rstep4 = (unsigned)G * (unsigned)L * 4u;
  auto st32 = [](float v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, (int)voff, (int)soff, 0);
  };
  auto issue = [&](auto bufc, int k) __attribute__((always_inline)) {
    constexpr int bf = decltype(bufc)::value;
    const int t = min(16 * (k * wpi + j) + col, s1 - 1);    // clamped into the item: always legal, masked at use
    const unsigned vo = ((unsigned)g * (unsigned)L + (unsigned)t) * 4u;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const unsigned so = (unsigned)i * rstep4;
      raw[bf][i][0] = ssv_buf_f32(rdy, vo, so); raw[bf][i][1] = ssv_buf_f32(rx, vo, so);
      raw[bf][i][2] = ssv_buf_f32(rh1, vo, so); raw[bf][i][3] = ssv_buf_f32(rh2, vo, so);
    }
    cst[bf][0] = sb[t]; cst[bf][1] = sb[L + t]; cst[bf][2] = sb[2L * L + t]; cst[bf][3] = sb[3L * L + t];
  };
  auto process = [&](auto bufc, int k) __attribute__((always_inline)) {
    constexpr int bf = decltype(bufc)::value;
    const int t = 16 * (k * wpi + j) + col;
    const bool tv = t < s1;
    const unsigned vo = ((unsigned)g * (unsigned)L + (unsigned)min(t, s1 - 1)) * 4u;
    const float mu1 = cst[bf][0], r1 = cst[bf][1], mu2 = cst[bf][2], r2 = cst[bf][3];
    float gs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      float q1, qb1, q2, qb2;
      if constexpr (PLDS) { const float4 q = prm[g + G * i]; q1 = q.x; qb1 = q.y; q2 = q.z; qb2 = q.w; }
      else { q1 = pg1[i]; qb1 = pb1[i]; q2 = pg2[i]; qb2 = pb2[i]; }
      const float dy = tv ? raw[bf][i][0] : 0.f, x = raw[bf][i][1];
      const float gg1 = tv ? q1 : 0.f, gg2 = tv ? q2 : 0.f;
      const float xh1 = tv ? (raw[bf][i][2] - mu1) * r1 : 0.f;
      const float xh2 = tv ? (raw[bf][i][3] - mu2) * r2 : 0.f;
      const float n1 = xh1 * gg1 + qb1;
      const float n2 = xh2 * gg2 + qb2;
      const float s = sigmoidf_(n1);
      const float dn2 = dy * s;
      const float dn1 = dy * (n2 - x) * s * (1.f - s);
      if (tv) st32(dy * (1.f - s), wdx, vo, (unsigned)i * rstep4);
      pacc[0][i] += dn1 * xh1; pacc[1][i] += dn1; pacc[2][i] += dn2 * xh2; pacc[3][i] += dn2;
      const float a1 = dn1 * gg1, a2 = dn2 * gg2;
      raw[bf][i][0] = a1; raw[bf][i][1] = a2; raw[bf][i][2] = xh1; raw[bf][i][3] = xh2;
      gs[0] += a1; gs[1] += a1 * xh1; gs[2] += a2; gs[3] += a2 * xh2;
    }
    const float inv = 1.f / (float)C;
    group_sums_waves<NW, 4>(gs, red, col, wave);
    const float m1 = gs[0] * inv, mh1 = gs[1] * inv, m2 = gs[2] * inv, mh2 = gs[3] * inv;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const float d1 = tv ? r1 * (raw[bf][i][0] - m1 - raw[bf][i][2] * mh1) : 0.f;
      const float d2 = tv ? r2 * (raw[bf][i][1] - m2 - raw[bf][i][3] * mh2) : 0.f;
      if (tv) { st32(d1, wh1, vo, (unsigned)i * rstep4); st32(d2, wh2, vo, (unsigned)i * rstep4); }
      am = fmaxf(am, fmaxf(fabsf(d1), fabsf(d2)));
      pacc[4][i] += d1; pacc[5][i] += d2;
    }
  };
  // The sub-tile whose loads are issued is always one ahead of the one being processed.  The last sub-tiles are peeled: a path that issues
  // and a path that does not must not share the code that waits (hipcc then waits for the stricter of the two: vmcnt(0), no overlap at all).
  const std::integral_constant<int, 0> B0{};
  const std::integral_constant<int, 1> B1{};
  issue(B0, 0);
  int k = 0;
  for (; k + 2 < ntile; k += 2) {
    issue(B1, k + 1);
    __builtin_amdgcn_sched_barrier(0);                       // the next sub-tile's loads stay above this sub-tile's arithmetic
    process(B0, k);
    issue(B0, k + 2);
    __builtin_amdgcn_sched_barrier(0);
    process(B1, k + 1);
  }
  if (k + 1 < ntile) {
    issue(B1, k + 1);
    __builtin_amdgcn_sched_barrier(0);
    process(B0, k);
    process(B1, k + 1);
  } else {
    process(B0, k);
  }
  // this workgroup's partial row
  {
    float* pblk = part + (long)wg * 6 * C;
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int i = 0; i < CPT; ++i) {
        const float t = col_sum(pacc[q][i]);
        if (col == 0) pblk[q * C + g + G * i] = t;
      }
  }
  if (amax) {                                    // max |dH| over the workgroup's range: entry j of the item's list, the last workgroup zeroes the rest
    am = ssv_wg_max<NW>(am, amx);
    if (threadIdx.x == 0) {
      const int na = ssv_amax_rows_(L);
      amax[(long)b * na + j] = am;
      if (j == wpi - 1) for (int e = wpi; e < na; ++e) amax[(long)b * na + e] = 0.f;
    }
  }
}


int ssv_launch_ln_gate_bwd_pers(const float* dY, long dy_bs, const float* H, const float* X, long x_bs, const float* stats, const float* g1, const float* b1,
                                const float* g2, const float* b2, float* dH, float* dXres, long dx_bs, float* part, float* amax, int B, int C, int L, int wpi,
                                hipStream_t st) {
  if (C != 256 || wpi < 1 || wpi > ssv_cdiv(L, 16)) return ssv_fail(SSV_UNSUPPORTED, "persistent gate backward: C = %d, %d workgroups per item", C, wpi);
  hipLaunchKernelGGL((ln_gate_bwd_pers_kernel<8, 32, 1>), dim3(B * wpi), dim3(512), 0, st, dY, dy_bs, H, X, x_bs, stats, g1, b1, g2, b2, dH, dXres, dx_bs, part, amax, C, L, wpi);
  return ssv_check_launch("ln_gate_bwd_pers");
}
