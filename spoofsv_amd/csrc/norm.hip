// Channel-axis LayerNorm kernels on (B, C, T) tensors (T fastest), gfx950.  HBM-bound.
//
// The reference permutes to (B, T, C) to run nn.LayerNorm(C); here the tensor stays (B, C, T):
// a 256-thread workgroup owns 16 consecutive time columns of one batch item, laid out as
// 16 columns (tid & 15) x 16 channel groups (tid >> 4).  Thread (col, g) keeps channels g, g+16, ...
// of its column in registers (one global read per element), so mean and variance are a true
// two-pass computation (mean first, then sum of squared deviations) without a second trip to memory.
// Cross-group sums go through a 16x16 LDS array; per-channel parameter-gradient partials are reduced
// over the 16 columns with lane shuffles and written per workgroup, then summed in a fixed order by
// reduce_partials (bitwise reproducible; no float atomics).
//
//   ln_gate_*: highwayConv epilogue  y = sigmoid(LN1(H1))*LN2(H2) + (1-sigmoid(LN1(H1)))*x
//   ln_act_* : y = act(LN(x)), act in {none, relu, sigmoid}
#include <stdlib.h>
#include "ssv_common.h"

#define LN_EPS 1e-5f
// channel steps of loads in flight per thread in the backward kernels (the step being computed included), see ln_gate_bwd_kernel;
// depths 2, 4, 6, 8 were measured in-step: equal within 1 % -- 4 keeps the hot instantiations at 128 VGPRs (4 waves per SIMD)
#define LN_PIPE(CPT) ((CPT) < 4 ? (CPT) : 4)

#ifndef SSV_SIGMOID_DIV
#define SSV_SIGMOID_DIV 0      // (tuning builds: 1 = the IEEE division sequence of rounds 1-4)
#endif
// v_rcp_f32 (1 ulp) instead of the IEEE division sequence (two div_scale, rcp, four fma, div_fmas, div_fixup: ten instructions per element)
__device__ __forceinline__ float sigmoidf_(float v) {
#if SSV_SIGMOID_DIV
  return 1.f / (1.f + __expf(-v));
#else
  return __builtin_amdgcn_rcpf(1.f + __expf(-v));
#endif
}

// Sum `v` over the G channel groups of this thread's column. `red` is a [G][16] LDS array.
template <int G>
__device__ __forceinline__ float group_sum(float v, float* red, int col, int g) {
  __syncthreads();
  red[g * 16 + col] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < G; ++i) s += red[i * 16 + col];
  return s;
}

// N sums in ONE LDS round trip (two barriers for all of them instead of two per sum): `red` is [N][G][16].  Per value the G
// partials are added in the same order as group_sum does, so results are bit-identical to N separate calls.
template <int G, int N>
__device__ __forceinline__ void group_sums(float (&v)[N], float* red, int col, int g) {
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; ++k) red[(k * G + g) * 16 + col] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; ++k) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < G; ++i) s += red[(k * G + i) * 16 + col];
    v[k] = s;
  }
}

// Sum over the 16 columns (16 consecutive lanes share one channel group = one DPP row): VALU only, no LDS traffic
// (the backward kernels make 3 of these per element).
__device__ __forceinline__ float col_sum(float v) { return ssv_row16_sum(v); }


// XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with a private L2.  A
// column tile is only 16 floats = 64 B of every channel row, so with the natural order the two tiles that share each
// 128-byte line (and the tiles that share a DRAM page) always land on different XCDs.  Re-number so that every XCD
// walks one contiguous range of tiles: co-running workgroups of an XCD then touch adjacent 64-byte pieces of the same rows.
__device__ __forceinline__ void xcd_tile(int& bx, int& by) {
  const unsigned gx = gridDim.x, total = gx * gridDim.y, lin = blockIdx.y * gx + blockIdx.x;
  const unsigned xcd = lin & 7, idx = lin >> 3, q = total >> 3, r = total & 7;
  const unsigned nl = xcd * q + (xcd < r ? xcd : r) + idx;
  bx = (int)(nl % gx);
  by = (int)(nl / gx);
}

// ------------------------------------------------------------------------------------------------
template <int CPT, int G>
__global__ __launch_bounds__(16 * G) void ln_gate_fwd_kernel(
    const float* __restrict__ H, long h_bs, const float* __restrict__ X, long x_bs,
    const float* __restrict__ g1, const float* __restrict__ b1, const float* __restrict__ g2, const float* __restrict__ b2,
    float* __restrict__ Y, long y_bs, float* __restrict__ stats, float* __restrict__ amax, int C, int L) {
  __shared__ float red[2 * 16 * G];
  __shared__ float amx[G / 4];
  const int col = threadIdx.x & 15, g = threadIdx.x >> 4;
  int bx, by;
  xcd_tile(bx, by);
  const int t = bx * 16 + col, b = by;
  const bool tv = t < L;
  // addressing: wave-uniform 64-bit bases + one 32-bit per-thread offset (c = g, column t), stepped by 16 rows
  const unsigned o0 = (unsigned)g * (unsigned)L + (unsigned)t, ostep = (unsigned)G * (unsigned)L;
  const float* __restrict__ Hb1 = H + (long)b * h_bs;
  const float* __restrict__ Hb2 = Hb1 + (long)C * L;
  float h1[CPT], h2[CPT];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = g + G * i;
    const bool v = tv && c < C;
    h1[i] = v ? Hb1[o0 + i * ostep] : 0.f;
    h2[i] = v ? Hb2[o0 + i * ostep] : 0.f;
    s1 += h1[i]; s2 += h2[i];
  }
  const float inv = 1.f / (float)C;
  float ms[2] = {s1, s2};
  group_sums<G, 2>(ms, red, col, g);
  const float mu1 = ms[0] * inv, mu2 = ms[1] * inv;
  float q1 = 0.f, q2 = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const bool v = (g + G * i) < C;
    const float d1 = v ? h1[i] - mu1 : 0.f, d2 = v ? h2[i] - mu2 : 0.f;
    q1 += d1 * d1; q2 += d2 * d2;
  }
  float qs[2] = {q1, q2};
  group_sums<G, 2>(qs, red, col, g);
  const float r1 = rsqrtf(qs[0] * inv + LN_EPS), r2 = rsqrtf(qs[1] * inv + LN_EPS);
  float am = 0.f;
  if (tv) {
    if (g == 0 && stats) {
      float* sb = stats + (long)b * 4 * L + t;
      sb[0] = mu1; sb[L] = r1; sb[2L * L] = mu2; sb[3L * L] = r2;
    }
    const float* __restrict__ Xb = X + (long)b * x_bs;
    float* __restrict__ Yb = Y + (long)b * y_bs;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int c = g + G * i;
      if (c < C) {
        const float n1 = (h1[i] - mu1) * r1 * g1[c] + b1[c];
        const float n2 = (h2[i] - mu2) * r2 * g2[c] + b2[c];
        const float s = sigmoidf_(n1);
        const float yv = s * n2 + (1.f - s) * Xb[o0 + i * ostep];
        Yb[o0 + i * ostep] = yv;
        am = fmaxf(am, fabsf(yv));
      }
    }
  }
  // split-fp16 operand scale of the NEXT convolution: this tile's max |y| (one entry per workgroup, ssv_amax_rows(L) per item)
  if (amax) {
    am = ssv_wg_max<G / 4>(am, amx);
    if (threadIdx.x == 0) {
      const int na = ssv_amax_rows_(L);            // list length: >= the tiles of this kernel; the last tile zeroes the rest
      amax[(long)b * na + bx] = am;
      if (bx == (int)gridDim.x - 1) for (int e = gridDim.x; e < na; ++e) amax[(long)b * na + e] = 0.f;
    }
  }
}

// The same forward with the column statistics ALREADY known: the conv kernel that produced H left, per 64-row group and
// column, the mean and the sum of squared deviations (GemmNNB::colstats).  Without a reduction the channels can be cut freely:
// a workgroup owns 64 columns x a QUARTER of the channels of one batch item (4x the workgroups of a whole-column tile: at
// L = 186 a 64-column tile alone gives 96 workgroups for 256 CUs, and the kernel ran no faster than the reducing one).  128 threads first merge the C / 64 groups of each half
// (Chan, equal counts) into mean / rstd, then all 256 threads stream -- thread (cq, g) takes columns 4 cq .. 4 cq + 3 of
// channels c0 + g, c0 + g + 16, ... as 16-byte accesses (256 contiguous bytes of a row per 16 lanes instead of 64 with the
// 16-column tiles above), nothing is kept in registers across channels.
typedef float f4ln __attribute__((ext_vector_type(4), aligned(4)));
__global__ __launch_bounds__(256) void ln_gate_fwd_stream_kernel(
    const float* __restrict__ H, const float* __restrict__ X, long x_bs, const float* __restrict__ cst,
    const float* __restrict__ g1, const float* __restrict__ b1, const float* __restrict__ g2, const float* __restrict__ b2,
    float* __restrict__ Y, long y_bs, float* __restrict__ stats, float* __restrict__ amax, int namax, int C, int L) {
  __shared__ float st[2][64][2];
  __shared__ float amx[4];
  int bxq, by;
  xcd_tile(bxq, by);
  const int bx = bxq >> 2, quarter = bxq & 3;               // the four channel quarters of a column tile are neighbours (same XCD)
  const int tid = threadIdx.x, b = by, t0 = bx * 64;
  if (tid < 128) {
    const int half = tid >> 6, col = tid & 63, t = t0 + col;
    if (t < L) {
      const int P = C >> 6, MG = 2 * P;
      const float* __restrict__ q = cst + (((long)b * MG + half * P) * L + t) * 2;
      // all (mean, M2) pairs of the column loaded before the first is used: clamped index, no branch around a load (with the loads
      // under `i < P` hipcc issued them one by one, each behind a wait for the one before: P dependent round trips at the head of a 12-us kernel)
      typedef float f2ln __attribute__((ext_vector_type(2), aligned(8)));
      f2ln pr[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) pr[i] = *reinterpret_cast<const f2ln*>(q + (long)min(i, P - 1) * L * 2);
      float mu[8], mean = 0.f, m2 = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { mu[i] = i < P ? pr[i].x : 0.f; mean += mu[i]; m2 += i < P ? pr[i].y : 0.f; }
      mean /= (float)P;
#pragma unroll
      for (int i = 0; i < 8; ++i) if (i < P) { const float d = mu[i] - mean; m2 += 64.f * d * d; }
      const float r = rsqrtf(m2 / (float)C + LN_EPS);
      st[half][col][0] = mean; st[half][col][1] = r;
      if (stats && quarter == 0) { float* sb = stats + (long)b * 4 * L + (long)(2 * half) * L + t; sb[0] = mean; sb[L] = r; }
    }
  }
  __syncthreads();
  const int cq = tid & 15, g = tid >> 4, t = t0 + 4 * cq;
  float am = 0.f;
  if (t < L) {
    float mu1[4], r1[4], mu2[4], r2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { mu1[j] = st[0][4 * cq + j][0]; r1[j] = st[0][4 * cq + j][1]; mu2[j] = st[1][4 * cq + j][0]; r2[j] = st[1][4 * cq + j][1]; }
    const float* __restrict__ H1 = H + (long)b * 2 * C * L + t;
    const float* __restrict__ H2 = H1 + (long)C * L;
    const float* __restrict__ Xb = X + (long)b * x_bs + t;
    float* __restrict__ Yb = Y + (long)b * y_bs + t;
    const bool full = t + 3 < L;
    const int cq4 = C >> 2, c0 = quarter * cq4;
#pragma unroll 4
    for (int c = c0 + g; c < c0 + cq4; c += 16) {
      const unsigned o = (unsigned)c * (unsigned)L;
      const float ga1 = g1[c], be1 = b1[c], ga2 = g2[c], be2 = b2[c];
      float h1[4], h2[4], xv[4], y[4];
      if (full) {
        const f4ln a = *reinterpret_cast<const f4ln*>(H1 + o), d = *reinterpret_cast<const f4ln*>(H2 + o), e = *reinterpret_cast<const f4ln*>(Xb + o);
#pragma unroll
        for (int j = 0; j < 4; ++j) { h1[j] = a[j]; h2[j] = d[j]; xv[j] = e[j]; }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const bool v = t + j < L; h1[j] = v ? H1[o + j] : 0.f; h2[j] = v ? H2[o + j] : 0.f; xv[j] = v ? Xb[o + j] : 0.f; }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float n1 = (h1[j] - mu1[j]) * r1[j] * ga1 + be1;
        const float n2 = (h2[j] - mu2[j]) * r2[j] * ga2 + be2;
        const float s = sigmoidf_(n1);
        y[j] = s * n2 + (1.f - s) * xv[j];
      }
      if (full) {
        *reinterpret_cast<f4ln*>(Yb + o) = (f4ln){y[0], y[1], y[2], y[3]};
        am = fmaxf(fmaxf(am, fmaxf(fabsf(y[0]), fabsf(y[1]))), fmaxf(fabsf(y[2]), fabsf(y[3])));
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (t + j < L) { Yb[o + j] = y[j]; am = fmaxf(am, fabsf(y[j])); }
      }
    }
  }
  if (amax) {                                    // this workgroup's max |y| for the next conv's operand scale: entry 4 * tile + quarter
    am = ssv_wg_max<4>(am, amx);                 // of the item's list (ssv_amax_rows(L) = 4 per 64 columns)
    if (tid == 0) amax[(long)b * namax + bxq] = am;
  }
}
int ssv_launch_ln_gate_fwd_stream(const float* H, const float* X, long x_bs, const float* colstats, const float* g1, const float* b1, const float* g2,
                                  const float* b2, float* Y, long y_bs, float* stats, float* amax, int B, int C, int L, hipStream_t st) {
  if (C % 64 != 0 || C > 512) return ssv_fail(SSV_UNSUPPORTED, "streaming highway gate: %d channels (multiples of 64 up to 512)", C);
  if ((long)2 * C * L >= (1L << 31)) return ssv_fail(SSV_UNSUPPORTED, "LayerNorm: one batch item exceeds 2^31 elements");
  if (ssv_shape_log_on()) {
    char note[64];
    snprintf(note, sizeof note, "B=%d C=%d L=%d", B, C, L);
    ssv_shape_log("ln_gate_fwd_stream_kernel", dim3(4 * ssv_cdiv(L, 64), B), dim3(256), 0.0, 16.0 * B * C * L, note);
  }
  hipLaunchKernelGGL(ln_gate_fwd_stream_kernel, dim3(4 * ssv_cdiv(L, 64), B), dim3(256), 0, st, H, X, x_bs, colstats, g1, b1, g2, b2, Y, y_bs, stats, amax,
                     ssv_amax_rows_(L), C, L);
  return ssv_check_launch("ln_gate_fwd_stream");
}

// Tuning builds only (-DSSV_LN_STAMP): thread 0 of each of the first 4096 workgroups of ln_gate_bwd_kernel records s_memrealtime (100 MHz, one
// clock for the device) at entry and exit and the shader clock after its load / reduce / store phases; tools/ln_stamps.py reads them.
#ifdef SSV_LN_STAMP
__device__ unsigned long long ssv_ln_stamps[4096 * 8];
extern "C" int ssv_debug_ln_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ssv_ln_stamps), sizeof(ssv_ln_stamps)); }
#define LN_STAMP(k, v) do { const unsigned w_ = blockIdx.y * gridDim.x + blockIdx.x; if (threadIdx.x == 0 && w_ < 4096u) ssv_ln_stamps[w_ * 8 + (k)] = (v); } while (0)
#else
#define LN_STAMP(k, v) do {} while (0)
#endif
// part layout: [block][6][C] = dgamma1, dbeta1, dgamma2, dbeta2, dbiasH1, dbiasH2
template <int CPT, int G>
__global__ __launch_bounds__(16 * G) void ln_gate_bwd_kernel(
    const float* __restrict__ dY, long dy_bs, const float* __restrict__ H, const float* __restrict__ X, long x_bs,
    const float* __restrict__ stats,
    const float* __restrict__ g1, const float* __restrict__ b1, const float* __restrict__ g2, const float* __restrict__ b2,
    float* __restrict__ dH, float* __restrict__ dXres, long dx_bs, float* __restrict__ part, float* __restrict__ amax, int C, int L) {
  __shared__ float red[4 * 16 * G];
  __shared__ float amx[G / 4];
  float am = 0.f;
  LN_STAMP(0, __builtin_amdgcn_s_memrealtime());
  LN_STAMP(1, __builtin_readcyclecounter());
  const int col = threadIdx.x & 15, g = threadIdx.x >> 4;
  int bx, by;
  xcd_tile(bx, by);
  const int t = bx * 16 + col, b = by;
  const bool tv = t < L;
  const unsigned o0 = (unsigned)g * (unsigned)L + (unsigned)t, ostep = (unsigned)G * (unsigned)L;
  const float* __restrict__ Hb1 = H + (long)b * 2 * C * L;
  const float* __restrict__ Hb2 = Hb1 + (long)C * L;
  float* __restrict__ dHb1 = dH + (long)b * 2 * C * L;
  float* __restrict__ dHb2 = dHb1 + (long)C * L;
  const float* __restrict__ dYb = dY + (long)b * dy_bs;
  const float* __restrict__ Xb = X + (long)b * x_bs;
  float* __restrict__ dXb = dXres + (long)b * dx_bs;
  float mu1 = 0.f, r1 = 0.f, mu2 = 0.f, r2 = 0.f;
  if (tv) {
    const float* sb = stats + (long)b * 4 * L + t;
    mu1 = sb[0]; r1 = sb[L]; mu2 = sb[2L * L]; r2 = sb[3L * L];
  }
  float xh1[CPT], xh2[CPT], a1[CPT], a2[CPT];
  float* pblk = part + ((long)by * gridDim.x + bx) * 6 * C;
  float sa1 = 0.f, sah1 = 0.f, sa2 = 0.f, sah2 = 0.f;
  // Load stream, software-pipelined (round 4).  With the loads of channel step i under `if (valid)` hipcc kept each step in a
  // block of its own: 8 loads, s_waitcnt vmcnt(0), arithmetic, stores, next step -- one exposed memory round trip (~1.1 us under
  // load) per step, 16 in a row: 33-40 k of a workgroup's ~42 k cycles (tools/ln_stamps.py), and only ~10 KB in flight per CU,
  // i.e. 2.5-3 TB/s by Little's law whatever the HBM could deliver.  Now the loads are UNCONDITIONAL (channel and column clamped
  // into the tensor: always legal; the validity mask is applied where the values are used) and steps i+1 .. i+PD-1 are in
  // flight while step i is computed.  Same arithmetic in the same order: results are bit-identical.
  constexpr int PD = LN_PIPE(CPT);
  const unsigned tc = (unsigned)min(t, L - 1);
  float qdy[PD], qx[PD], qh1[PD], qh2[PD], qg1[PD], qb1[PD], qg2[PD], qb2[PD];
  auto issue = [&](int i) __attribute__((always_inline)) {
    const int sl = i % PD, c = min(g + G * i, C - 1);
    const unsigned o = (unsigned)c * (unsigned)L + tc;
    qdy[sl] = dYb[o]; qx[sl] = Xb[o]; qh1[sl] = Hb1[o]; qh2[sl] = Hb2[o];
    qg1[sl] = g1[c]; qb1[sl] = b1[c]; qg2[sl] = g2[c]; qb2[sl] = b2[c];
  };
#pragma unroll
  for (int i = 0; i < PD - 1 && i < CPT; ++i) issue(i);
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = g + G * i, sl = i % PD;
    const bool v = tv && c < C;
    if (i + PD - 1 < CPT) issue(i + PD - 1);
    __builtin_amdgcn_sched_barrier(0);             // the loads above stay above: or the scheduler sinks them back to their use
    const float dy = v ? qdy[sl] : 0.f, x = qx[sl];
    const float gg1 = v ? qg1[sl] : 0.f, gg2 = v ? qg2[sl] : 0.f;
    xh1[i] = v ? (qh1[sl] - mu1) * r1 : 0.f;
    xh2[i] = v ? (qh2[sl] - mu2) * r2 : 0.f;
    const float n1 = xh1[i] * gg1 + qb1[sl];
    const float n2 = xh2[i] * gg2 + qb2[sl];
    const float s = sigmoidf_(n1);
    const float dn2 = dy * s;
    const float dn1 = dy * (n2 - x) * s * (1.f - s);
    if (v) dXb[o0 + i * ostep] = dy * (1.f - s);
    // per-channel parameter-gradient partials over this block's 16 columns
    const float p0 = col_sum(dn1 * xh1[i]), p1 = col_sum(dn1), p2 = col_sum(dn2 * xh2[i]), p3 = col_sum(dn2);
    if (col == 0 && c < C) { pblk[c] = p0; pblk[C + c] = p1; pblk[2 * C + c] = p2; pblk[3 * C + c] = p3; }
    a1[i] = dn1 * gg1; a2[i] = dn2 * gg2;
    sa1 += a1[i]; sah1 += a1[i] * xh1[i]; sa2 += a2[i]; sah2 += a2[i] * xh2[i];
  }
  const float inv = 1.f / (float)C;
  float gs[4] = {sa1, sah1, sa2, sah2};
#ifdef SSV_LN_STAMP
  asm volatile("" : "+v"(gs[0]), "+v"(gs[1]), "+v"(gs[2]), "+v"(gs[3]));     // every load of the first phase has landed
#endif
  LN_STAMP(2, __builtin_readcyclecounter());
  group_sums<G, 4>(gs, red, col, g);
  LN_STAMP(3, __builtin_readcyclecounter());
  const float m1 = gs[0] * inv, mh1 = gs[1] * inv, m2 = gs[2] * inv, mh2 = gs[3] * inv;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = g + G * i;
    const bool v = tv && c < C;
    const float d1 = v ? r1 * (a1[i] - m1 - xh1[i] * mh1) : 0.f;
    const float d2 = v ? r2 * (a2[i] - m2 - xh2[i] * mh2) : 0.f;
    if (v) { dHb1[o0 + i * ostep] = d1; dHb2[o0 + i * ostep] = d2; }
    am = fmaxf(am, fmaxf(fabsf(d1), fabsf(d2)));
    const float q0 = col_sum(d1), q1 = col_sum(d2);
    if (col == 0 && c < C) { pblk[4 * C + c] = q0; pblk[5 * C + c] = q1; }
  }
  LN_STAMP(4, __builtin_readcyclecounter());
#ifdef SSV_LN_STAMP
  __builtin_amdgcn_s_waitcnt(0);                    // (stamp builds: the stores of this wave have been acknowledged)
  LN_STAMP(5, __builtin_readcyclecounter());
  LN_STAMP(6, __builtin_amdgcn_s_memrealtime());
#endif
  if (amax) {                                    // max |dH| of this tile: operand scale of the two conv gradients (split-fp16)
    am = ssv_wg_max<G / 4>(am, amx);
    if (threadIdx.x == 0) {
      const int na = ssv_amax_rows_(L);            // list length: >= the tiles of this kernel; the last tile zeroes the rest
      amax[(long)b * na + bx] = am;
      if (bx == (int)gridDim.x - 1) for (int e = gridDim.x; e < na; ++e) amax[(long)b * na + e] = 0.f;
    }
  }
}

// ---- wide-tile backward (round 4) ---------------------------------------------------------------------------------------
// Same mathematics as ln_gate_bwd_kernel on a tile of 16 * VEC columns x ALL channels with 1024 threads: thread (cq, g) owns VEC
// consecutive columns (one 4 * VEC-byte access per channel row: 64 * VEC contiguous bytes of a row per 16 lanes instead of 64) of channels
// g, g + 64, ... (CPT steps).  Why: stamps (tools/ln_stamps.py) showed the 16-column kernel's workgroups spending 85 % of their ~20 us waiting
// for 16 dependent batches of 4-byte loads; with the loads pipelined (above) the kernel moved by 5 % only -- what remains is the rate the
// memory system gives 64-byte row pieces.  Here every load of the tile is issued before the first is used (CPT * 4 vector loads in flight per
// thread, landing IN the arrays that later hold xhat / a), rows are read by raw buffer loads whose range is the batch item (no clamps, no
// branches: a piece past the item's end reads 0, columns past L are masked at use).
// Cross-group sums: 64 groups -> LDS [64][4 sums][16 VEC columns], summed in a fixed order by the first 64 * VEC threads.
// Partial rows are COMPACT: one row per tile, gridDim.x rows per item (ssv_ln_gate_bwd_rows / ssv_ln_act_bwd_rows tell the callers how many
// rows a launch writes; until the round-4 second session a tile also zeroed VEC - 1 rows of the 16-column numbering, which every reduction
// then read: 2,624 rows for 672 at L = 1300).  Scale-list entries keep the 16-column numbering: tile bx owns entries VEC * bx .. VEC * bx +
// VEC - 1 of the item's list (the first carries the tile's maximum, the others are zeroed).
template <int CPT, int VEC>
__global__ __launch_bounds__(1024) void ln_gate_bwd_wide_kernel(
    const float* __restrict__ dY, long dy_bs, const float* __restrict__ H, const float* __restrict__ X, long x_bs,
    const float* __restrict__ stats,
    const float* __restrict__ g1, const float* __restrict__ b1, const float* __restrict__ g2, const float* __restrict__ b2,
    float* __restrict__ dH, float* __restrict__ dXres, long dx_bs, float* __restrict__ part, float* __restrict__ amax, int C, int L) {
  constexpr int G = 64, TC = 16 * VEC;
  typedef float vecf __attribute__((ext_vector_type(VEC), aligned(4)));
  __shared__ float red[G * 4 * TC];                 // [g][sum][column of the tile]
  __shared__ float tot[4 * TC];
  __shared__ float amx[16];
  const int cq = threadIdx.x & 15, g = threadIdx.x >> 4;
  int bx, by;
  xcd_tile(bx, by);
  const int t = bx * TC + VEC * cq, b = by;
  bool cv[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) cv[j] = t + j < L;
  const float* __restrict__ Hb1 = H + (long)b * 2 * C * L;
  float* __restrict__ dHb1 = dH + (long)b * 2 * C * L;
  float* __restrict__ dHb2 = dHb1 + (long)C * L;
  float* __restrict__ dXb = dXres + (long)b * dx_bs;
  // buffer resources over this batch item's rows: a load may run past a row (into the next one: masked) or past the item (reads 0)
  const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dY + (long)b * dy_bs), 0, C * L * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X + (long)b * x_bs), 0, C * L * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Hb1), 0, 2 * C * L * 4, 0x00020000);
  auto ldv = [&](__amdgpu_buffer_rsrc_t r, unsigned off, float (&v)[VEC]) __attribute__((always_inline)) {
    if constexpr (VEC == 4) { const f32x4 q = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0)); v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3]; }
    else if constexpr (VEC == 2) { typedef float f2 __attribute__((ext_vector_type(2))); const f2 q = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0)); v[0] = q[0]; v[1] = q[1]; }
    else v[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
  };
  // every load of the tile first: they land in the arrays that later hold xhat1, xhat2, a1, a2
  float xh1[CPT][VEC], xh2[CPT][VEC], a1[CPT][VEC], a2[CPT][VEC];     // raw: h1, h2, dy, x
  float pg1[CPT], pb1[CPT], pg2[CPT], pb2[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = min(g + G * i, C - 1);
    const unsigned o = ((unsigned)c * (unsigned)L + (unsigned)t) * 4u;
    ldv(rh, o, xh1[i]); ldv(rh, o + (unsigned)C * (unsigned)L * 4u, xh2[i]); ldv(rdy, o, a1[i]); ldv(rx, o, a2[i]);
    pg1[i] = g1[c]; pb1[i] = b1[c]; pg2[i] = g2[c]; pb2[i] = b2[c];
  }
  float mu1[VEC], r1[VEC], mu2[VEC], r2[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const float* sb = stats + (long)b * 4 * L + min(t + j, L - 1);
    mu1[j] = sb[0]; r1[j] = sb[L]; mu2[j] = sb[2L * L]; r2[j] = sb[3L * L];
  }
  float* pblk = part + ((long)by * gridDim.x + bx) * 6 * C;                    // this tile's partial row: one per tile, compact (ssv_ln_gate_bwd_rows)
  float sa1[VEC], sah1[VEC], sa2[VEC], sah2[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) sa1[j] = sah1[j] = sa2[j] = sah2[j] = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = g + G * i;
    const bool cok = c < C;
    float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f, dxr[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const bool v = cok && cv[j];
      const float dy = v ? a1[i][j] : 0.f, x = a2[i][j];
      const float gg1 = v ? pg1[i] : 0.f, gg2 = v ? pg2[i] : 0.f;
      const float h1 = v ? (xh1[i][j] - mu1[j]) * r1[j] : 0.f;
      const float h2 = v ? (xh2[i][j] - mu2[j]) * r2[j] : 0.f;
      const float n1 = h1 * gg1 + pb1[i], n2 = h2 * gg2 + pb2[i];
      const float s = sigmoidf_(n1);
      const float dn2 = dy * s, dn1 = dy * (n2 - x) * s * (1.f - s);
      dxr[j] = dy * (1.f - s);
      q0 += dn1 * h1; q1 += dn1; q2 += dn2 * h2; q3 += dn2;
      xh1[i][j] = h1; xh2[i][j] = h2; a1[i][j] = dn1 * gg1; a2[i][j] = dn2 * gg2;
      sa1[j] += a1[i][j]; sah1[j] += a1[i][j] * h1; sa2[j] += a2[i][j]; sah2[j] += a2[i][j] * h2;
    }
    if (cok) {
      float* dst = dXb + (long)c * L + t;
      if (cv[VEC - 1]) { vecf o; for (int j = 0; j < VEC; ++j) o[j] = dxr[j]; *reinterpret_cast<vecf*>(dst) = o; }
      else for (int j = 0; j < VEC; ++j) if (cv[j]) dst[j] = dxr[j];
    }
    q0 = col_sum(q0); q1 = col_sum(q1); q2 = col_sum(q2); q3 = col_sum(q3);
    if (cq == 0 && cok) { pblk[c] = q0; pblk[C + c] = q1; pblk[2 * C + c] = q2; pblk[3 * C + c] = q3; }
  }
  // cross-group sums
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    red[(g * 4 + 0) * TC + VEC * cq + j] = sa1[j]; red[(g * 4 + 1) * TC + VEC * cq + j] = sah1[j];
    red[(g * 4 + 2) * TC + VEC * cq + j] = sa2[j]; red[(g * 4 + 3) * TC + VEC * cq + j] = sah2[j];
  }
  __syncthreads();
  if (threadIdx.x < 4 * TC) {
    float sum = 0.f;
#pragma unroll 8
    for (int k = 0; k < G; ++k) sum += red[k * 4 * TC + threadIdx.x];
    tot[threadIdx.x] = sum;
  }
  __syncthreads();
  const float inv = 1.f / (float)C;
  float m1[VEC], mh1[VEC], m2[VEC], mh2[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    m1[j] = tot[0 * TC + VEC * cq + j] * inv; mh1[j] = tot[1 * TC + VEC * cq + j] * inv;
    m2[j] = tot[2 * TC + VEC * cq + j] * inv; mh2[j] = tot[3 * TC + VEC * cq + j] * inv;
  }
  float am = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = g + G * i;
    const bool cok = c < C;
    float d1[VEC], d2[VEC], q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const bool v = cok && cv[j];
      d1[j] = v ? r1[j] * (a1[i][j] - m1[j] - xh1[i][j] * mh1[j]) : 0.f;
      d2[j] = v ? r2[j] * (a2[i][j] - m2[j] - xh2[i][j] * mh2[j]) : 0.f;
      am = fmaxf(am, fmaxf(fabsf(d1[j]), fabsf(d2[j])));
      q0 += d1[j]; q1 += d2[j];
    }
    if (cok) {
      float* o1 = dHb1 + (long)c * L + t;
      float* o2 = dHb2 + (long)c * L + t;
      if (cv[VEC - 1]) {
        vecf u, w;
        for (int j = 0; j < VEC; ++j) { u[j] = d1[j]; w[j] = d2[j]; }
        *reinterpret_cast<vecf*>(o1) = u; *reinterpret_cast<vecf*>(o2) = w;
      } else for (int j = 0; j < VEC; ++j) if (cv[j]) { o1[j] = d1[j]; o2[j] = d2[j]; }
    }
    q0 = col_sum(q0); q1 = col_sum(q1);
    if (cq == 0 && cok) { pblk[4 * C + c] = q0; pblk[5 * C + c] = q1; }
  }
  if (amax) {
    am = ssv_wg_max<16>(am, amx);
    if (threadIdx.x == 0) {
      const int na = ssv_amax_rows_(L);
      float* al = amax + (long)b * na;
      al[VEC * bx] = am;
      for (int r = 1; r < VEC; ++r) if (VEC * bx + r < na) al[VEC * bx + r] = 0.f;
      if (bx == (int)gridDim.x - 1) for (int e = VEC * (int)gridDim.x; e < na; ++e) al[e] = 0.f;
    }
  }
}


// ------------------------------------------------------------------------------------------------
template <int CPT, int G>
__global__ __launch_bounds__(16 * G) void ln_act_fwd_kernel(
    const float* __restrict__ X, long x_bs, const float* __restrict__ gam, const float* __restrict__ bet,
    float* __restrict__ Y, long y_bs, float* __restrict__ stats, float* __restrict__ amax, int C, int L, int act) {
  __shared__ float red[16 * G];
  __shared__ float amx[G / 4];
  const int col = threadIdx.x & 15, g = threadIdx.x >> 4;
  int bx, by;
  xcd_tile(bx, by);
  const int t = bx * 16 + col, b = by;
  const bool tv = t < L;
  const unsigned o0 = (unsigned)g * (unsigned)L + (unsigned)t, ostep = (unsigned)G * (unsigned)L;
  const float* __restrict__ Xb = X + (long)b * x_bs;
  float x[CPT];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = g + G * i;
    x[i] = (tv && c < C) ? Xb[o0 + i * ostep] : 0.f;
    s += x[i];
  }
  const float inv = 1.f / (float)C;
  const float mu = group_sum<G>(s, red, col, g) * inv;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const float d = ((g + G * i) < C) ? x[i] - mu : 0.f;
    q += d * d;
  }
  const float r = rsqrtf(group_sum<G>(q, red, col, g) * inv + LN_EPS);
  float am = 0.f;
  if (tv) {
    if (g == 0 && stats) { stats[(long)b * 2 * L + t] = mu; stats[(long)b * 2 * L + L + t] = r; }
    float* __restrict__ Yb = Y + (long)b * y_bs;
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      const int c = g + G * i;
      if (c < C) {
        float n = (x[i] - mu) * r * gam[c] + bet[c];
        if (act == 1) n = fmaxf(n, 0.f);
        else if (act == 2) n = sigmoidf_(n);
        Yb[o0 + i * ostep] = n;
        am = fmaxf(am, fabsf(n));
      }
    }
  }
  if (amax) {                                    // see ln_gate_fwd_kernel
    am = ssv_wg_max<G / 4>(am, amx);
    if (threadIdx.x == 0) {
      const int na = ssv_amax_rows_(L);            // list length: >= the tiles of this kernel; the last tile zeroes the rest
      amax[(long)b * na + bx] = am;
      if (bx == (int)gridDim.x - 1) for (int e = gridDim.x; e < na; ++e) amax[(long)b * na + e] = 0.f;
    }
  }
}

// part layout: [block][3][C] = dgamma, dbeta, dbias (= column sums of dX)
template <int CPT, int G>
__global__ __launch_bounds__(16 * G) void ln_act_bwd_kernel(
    const float* __restrict__ dY, long dy_bs, const float* __restrict__ X, long x_bs, const float* __restrict__ stats,
    const float* __restrict__ gam, const float* __restrict__ bet,
    float* __restrict__ dX, long dx_bs, float* __restrict__ part, float* __restrict__ amax, int C, int L, int act) {
  __shared__ float red[2 * 16 * G];
  __shared__ float amx[G / 4];
  float am = 0.f;
  const int col = threadIdx.x & 15, g = threadIdx.x >> 4;
  int bx, by;
  xcd_tile(bx, by);
  const int t = bx * 16 + col, b = by;
  const bool tv = t < L;
  float mu = 0.f, r = 0.f;
  if (tv) { mu = stats[(long)b * 2 * L + t]; r = stats[(long)b * 2 * L + L + t]; }
  const unsigned o0 = (unsigned)g * (unsigned)L + (unsigned)t, ostep = (unsigned)G * (unsigned)L;
  const float* __restrict__ dYb = dY + (long)b * dy_bs;
  const float* __restrict__ Xb = X + (long)b * x_bs;
  float* __restrict__ dXb = dX + (long)b * dx_bs;
  float xh[CPT], a[CPT];
  float* pblk = part + ((long)by * gridDim.x + bx) * 3 * C;
  float sa = 0.f, sah = 0.f;
  // software-pipelined unconditional loads, masks applied at use: see ln_gate_bwd_kernel
  constexpr int PD = LN_PIPE(CPT);
  const unsigned tc = (unsigned)min(t, L - 1);
  float qdy[PD], qx[PD], qg[PD], qb[PD];
  auto issue = [&](int i) __attribute__((always_inline)) {
    const int sl = i % PD, c = min(g + G * i, C - 1);
    const unsigned o = (unsigned)c * (unsigned)L + tc;
    qdy[sl] = dYb[o]; qx[sl] = Xb[o]; qg[sl] = gam[c]; qb[sl] = bet[c];
  };
#pragma unroll
  for (int i = 0; i < PD - 1 && i < CPT; ++i) issue(i);
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = g + G * i, sl = i % PD;
    const bool v = tv && c < C;
    if (i + PD - 1 < CPT) issue(i + PD - 1);
    __builtin_amdgcn_sched_barrier(0);
    const float dy = v ? qdy[sl] : 0.f, gg = v ? qg[sl] : 0.f;
    xh[i] = v ? (qx[sl] - mu) * r : 0.f;
    const float n = xh[i] * gg + qb[sl];
    float dn;
    if (act == 1) dn = n > 0.f ? dy : 0.f;
    else if (act == 2) { const float s = sigmoidf_(n); dn = dy * s * (1.f - s); }
    else dn = dy;
    const float p0 = col_sum(dn * xh[i]), p1 = col_sum(dn);
    if (col == 0 && c < C) { pblk[c] = p0; pblk[C + c] = p1; }
    a[i] = dn * gg;
    sa += a[i]; sah += a[i] * xh[i];
  }
  const float inv = 1.f / (float)C;
  float gs[2] = {sa, sah};
  group_sums<G, 2>(gs, red, col, g);
  const float m = gs[0] * inv, mh = gs[1] * inv;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = g + G * i;
    const bool v = tv && c < C;
    const float d = v ? r * (a[i] - m - xh[i] * mh) : 0.f;
    if (v) dXb[o0 + i * ostep] = d;
    am = fmaxf(am, fabsf(d));
    const float q0 = col_sum(d);
    if (col == 0 && c < C) pblk[2 * C + c] = q0;
  }
  if (amax) {                                    // see ln_gate_bwd_kernel
    am = ssv_wg_max<G / 4>(am, amx);
    if (threadIdx.x == 0) {
      const int na = ssv_amax_rows_(L);            // list length: >= the tiles of this kernel; the last tile zeroes the rest
      amax[(long)b * na + bx] = am;
      if (bx == (int)gridDim.x - 1) for (int e = gridDim.x; e < na; ++e) amax[(long)b * na + e] = 0.f;
    }
  }
}

// Wide-tile form of ln_act_bwd_kernel (see ln_gate_bwd_wide_kernel): 16 * VEC columns x all channels, 1024 threads, every load first.
template <int CPT, int VEC>
__global__ __launch_bounds__(1024) void ln_act_bwd_wide_kernel(
    const float* __restrict__ dY, long dy_bs, const float* __restrict__ X, long x_bs, const float* __restrict__ stats,
    const float* __restrict__ gam, const float* __restrict__ bet,
    float* __restrict__ dX, long dx_bs, float* __restrict__ part, float* __restrict__ amax, int C, int L, int act) {
  constexpr int G = 64, TC = 16 * VEC;
  typedef float vecf __attribute__((ext_vector_type(VEC), aligned(4)));
  __shared__ float red[G * 2 * TC];
  __shared__ float tot[2 * TC];
  __shared__ float amx[16];
  const int cq = threadIdx.x & 15, g = threadIdx.x >> 4;
  int bx, by;
  xcd_tile(bx, by);
  const int t = bx * TC + VEC * cq, b = by;
  bool cv[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) cv[j] = t + j < L;
  float* __restrict__ dXb = dX + (long)b * dx_bs;
  const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dY + (long)b * dy_bs), 0, C * L * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(X + (long)b * x_bs), 0, C * L * 4, 0x00020000);
  auto ldv = [&](__amdgpu_buffer_rsrc_t r, unsigned off, float (&v)[VEC]) __attribute__((always_inline)) {
    if constexpr (VEC == 4) { const f32x4 q = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0)); v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3]; }
    else if constexpr (VEC == 2) { typedef float f2 __attribute__((ext_vector_type(2))); const f2 q = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0)); v[0] = q[0]; v[1] = q[1]; }
    else v[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
  };
  float xh[CPT][VEC], a[CPT][VEC];                  // raw: x, dy
  float pg[CPT], pb[CPT];
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = min(g + G * i, C - 1);
    const unsigned o = ((unsigned)c * (unsigned)L + (unsigned)t) * 4u;
    ldv(rx, o, xh[i]); ldv(rdy, o, a[i]);
    pg[i] = gam[c]; pb[i] = bet[c];
  }
  float mu[VEC], r[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { const float* sb = stats + (long)b * 2 * L + min(t + j, L - 1); mu[j] = sb[0]; r[j] = sb[L]; }
  float* pblk = part + ((long)by * gridDim.x + bx) * 3 * C;                    // one partial row per tile, compact (ssv_ln_act_bwd_rows)
  float sa[VEC], sah[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) sa[j] = sah[j] = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = g + G * i;
    const bool cok = c < C;
    float q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const bool v = cok && cv[j];
      const float dy = v ? a[i][j] : 0.f, gg = v ? pg[i] : 0.f;
      const float h = v ? (xh[i][j] - mu[j]) * r[j] : 0.f;
      const float n = h * gg + pb[i];
      float dn;
      if (act == 1) dn = n > 0.f ? dy : 0.f;
      else if (act == 2) { const float s = sigmoidf_(n); dn = dy * s * (1.f - s); }
      else dn = dy;
      q0 += dn * h; q1 += dn;
      xh[i][j] = h; a[i][j] = dn * gg;
      sa[j] += a[i][j]; sah[j] += a[i][j] * h;
    }
    q0 = col_sum(q0); q1 = col_sum(q1);
    if (cq == 0 && cok) { pblk[c] = q0; pblk[C + c] = q1; }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) { red[(g * 2 + 0) * TC + VEC * cq + j] = sa[j]; red[(g * 2 + 1) * TC + VEC * cq + j] = sah[j]; }
  __syncthreads();
  if (threadIdx.x < 2 * TC) {
    float sum = 0.f;
#pragma unroll 8
    for (int k = 0; k < G; ++k) sum += red[k * 2 * TC + threadIdx.x];
    tot[threadIdx.x] = sum;
  }
  __syncthreads();
  const float inv = 1.f / (float)C;
  float m[VEC], mh[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { m[j] = tot[VEC * cq + j] * inv; mh[j] = tot[TC + VEC * cq + j] * inv; }
  float am = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = g + G * i;
    const bool cok = c < C;
    float d[VEC], q0 = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      d[j] = (cok && cv[j]) ? r[j] * (a[i][j] - m[j] - xh[i][j] * mh[j]) : 0.f;
      am = fmaxf(am, fabsf(d[j]));
      q0 += d[j];
    }
    if (cok) {
      float* o = dXb + (long)c * L + t;
      if (cv[VEC - 1]) { vecf u; for (int j = 0; j < VEC; ++j) u[j] = d[j]; *reinterpret_cast<vecf*>(o) = u; }
      else for (int j = 0; j < VEC; ++j) if (cv[j]) o[j] = d[j];
    }
    q0 = col_sum(q0);
    if (cq == 0 && cok) pblk[2 * C + c] = q0;
  }
  if (amax) {
    am = ssv_wg_max<16>(am, amx);
    if (threadIdx.x == 0) {
      const int na = ssv_amax_rows_(L);
      float* al = amax + (long)b * na;
      al[VEC * bx] = am;
      for (int rr = 1; rr < VEC; ++rr) if (VEC * bx + rr < na) al[VEC * bx + rr] = 0.f;
      if (bx == (int)gridDim.x - 1) for (int e = VEC * (int)gridDim.x; e < na; ++e) al[e] = 0.f;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Second order: the WGAN-GP gradient penalty (train/adversarial_wasserstein_gp.py:300-308) differentiates the critic's
// input gradient, i.e. it needs the gradient of the BACKWARD kernels above.  For a scalar s = <v, dX> with
// dX = ln_act_bwd(act = none)(gn; x):   a = gamma*gn,  u = r (v - mean v - xh mean(v xh)),
//   ds/dgn = gamma u,   ds/dgamma = sum_cols gn u,
//   ds/dx  = -r xh s_col / C - r^2 (w - mean w - xh mean(w xh)),   w = v mean(a xh) + a mean(v xh),
// with s_col = r (<v,a> - C mean v mean a - C mean(v xh) mean(a xh)) this column's share of s.  part: [block][C] = dgamma.
template <int CPT, int G>
__global__ __launch_bounds__(16 * G) void ln_bwd2_kernel(
    const float* __restrict__ V, long v_bs, const float* __restrict__ GN, long gn_bs, const float* __restrict__ X, long x_bs,
    const float* __restrict__ stats, const float* __restrict__ gam,
    float* __restrict__ dGN, long dgn_bs, float* __restrict__ dX, long dx_bs, float* __restrict__ part, int C, int L) {
  __shared__ float red[5 * 16 * G];
  const int col = threadIdx.x & 15, g = threadIdx.x >> 4;
  int bx, by;
  xcd_tile(bx, by);
  const int t = bx * 16 + col, b = by;
  const bool tv = t < L;
  float mu = 0.f, r = 0.f;
  if (tv) { mu = stats[(long)b * 2 * L + t]; r = stats[(long)b * 2 * L + L + t]; }
  const unsigned o0 = (unsigned)g * (unsigned)L + (unsigned)t, ostep = (unsigned)G * (unsigned)L;
  const float* __restrict__ Vb = V + (long)b * v_bs;
  const float* __restrict__ GNb = GN + (long)b * gn_bs;
  const float* __restrict__ Xb = X + (long)b * x_bs;
  float xh[CPT], v[CPT], gn[CPT];
  float sv = 0.f, svx = 0.f, sa = 0.f, sax = 0.f, sva = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = g + G * i;
    xh[i] = 0.f; v[i] = 0.f; gn[i] = 0.f;
    if (tv && c < C) {
      xh[i] = (Xb[o0 + i * ostep] - mu) * r;
      v[i] = Vb[o0 + i * ostep];
      gn[i] = GNb[o0 + i * ostep];
      const float a = gn[i] * gam[c];
      sv += v[i]; svx += v[i] * xh[i]; sa += a; sax += a * xh[i]; sva += v[i] * a;
    }
  }
  const float inv = 1.f / (float)C;
  float gs[5] = {sv, svx, sa, sax, sva};
  group_sums<G, 5>(gs, red, col, g);
  const float m_v = gs[0] * inv, m_vx = gs[1] * inv, m_a = gs[2] * inv, m_ax = gs[3] * inv, m_va = gs[4] * inv;
  const float s_c = r * (m_va - m_v * m_a - m_vx * m_ax);            // s_col / C
  const float mw = m_v * m_ax + m_a * m_vx, mwx = 2.f * m_vx * m_ax;
  float* pblk = part + ((long)by * gridDim.x + bx) * C;
  float* __restrict__ dGNb = dGN + (long)b * dgn_bs;
  float* __restrict__ dXb = dX + (long)b * dx_bs;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = g + G * i;
    const bool ok = tv && c < C;
    const float gg = c < C ? gam[c] : 0.f;
    const float u = ok ? r * (v[i] - m_v - xh[i] * m_vx) : 0.f;
    if (ok) {
      const float w = v[i] * m_ax + gn[i] * gg * m_vx;
      dGNb[o0 + i * ostep] = gg * u;
      dXb[o0 + i * ostep] = -r * xh[i] * s_c - r * r * (w - mw - xh[i] * mwx);
    }
    const float p0 = col_sum(gn[i] * u);
    if (col == 0 && c < C) pblk[c] = p0;
  }
}

// s = <v1, dH1> + <v2, dH2> + <vx, dXres> for (dH, dXres) = ln_gate_bwd(gy; h, x).  With, per LayerNorm i, a_i = gamma_i gn_i
// (gn2 = gy g, gn1 = gy (n2 - x) g', g = sigmoid(n1), g' = g(1-g), g'' = g'(1-2g)), u_i as above and t_i = gamma_i u_i:
//   ds/dgy = t2 g + t1 (n2 - x) g' + vx (1 - g)
//   e2 = ds/dn2 = t1 gy g',   ds/dx = -e2,   e1 = ds/dn1 = (t2 - vx) gy g' + t1 gy (n2 - x) g''
//   ds/dh_i = LNbwd_i(e_i) + (second-order LayerNorm term of ln_bwd2 with v_i, a_i)
//   ds/dgamma_i = sum_cols gn_i u_i + e_i xh_i,   ds/dbeta_i = sum_cols e_i.        part: [block][4][C] = dg1, db1, dg2, db2
template <int CPT, int G>
__global__ __launch_bounds__(16 * G) void ln_gate_bwd2_kernel(
    const float* __restrict__ VH, const float* __restrict__ VX, long vx_bs, const float* __restrict__ GY, long gy_bs,
    const float* __restrict__ H, const float* __restrict__ X, long x_bs, const float* __restrict__ stats,
    const float* __restrict__ g1, const float* __restrict__ b1, const float* __restrict__ g2, const float* __restrict__ b2,
    float* __restrict__ dGY, long dgy_bs, float* __restrict__ dH, float* __restrict__ dX, long dx_bs,
    float* __restrict__ part, int C, int L) {
  __shared__ float red[10 * 16 * G];
  const int col = threadIdx.x & 15, g = threadIdx.x >> 4;
  int bx, by;
  xcd_tile(bx, by);
  const int t = bx * 16 + col, b = by;
  const bool tv = t < L;
  const unsigned o0 = (unsigned)g * (unsigned)L + (unsigned)t, ostep = (unsigned)G * (unsigned)L;
  const long hb = (long)b * 2 * C * L, cl = (long)C * L;
  const float* __restrict__ H1 = H + hb;
  const float* __restrict__ H2 = H1 + cl;
  const float* __restrict__ V1 = VH + hb;
  const float* __restrict__ V2 = V1 + cl;
  const float* __restrict__ VXb = VX + (long)b * vx_bs;
  const float* __restrict__ GYb = GY + (long)b * gy_bs;
  const float* __restrict__ Xb = X + (long)b * x_bs;
  float mu1 = 0.f, r1 = 0.f, mu2 = 0.f, r2 = 0.f;
  if (tv) {
    const float* sb = stats + (long)b * 4 * L + t;
    mu1 = sb[0]; r1 = sb[L]; mu2 = sb[2L * L]; r2 = sb[3L * L];
  }
  // kept across the phases: normalised inputs, upstream v, first-backward gn; and the element-wise factors of e1, e2
  float xh1[CPT], xh2[CPT], v1[CPT], v2[CPT], gn1[CPT], gn2[CPT];
  float gy[CPT], dq[CPT], gp[CPT], gpp[CPT], vx[CPT], sg[CPT];      // gy, n2 - x, g', g'', vx, g
  float s[10] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = g + G * i;
    xh1[i] = xh2[i] = v1[i] = v2[i] = gn1[i] = gn2[i] = gy[i] = dq[i] = gp[i] = gpp[i] = vx[i] = sg[i] = 0.f;
    if (tv && c < C) {
      const float gg1 = g1[c], gg2 = g2[c];
      xh1[i] = (H1[o0 + i * ostep] - mu1) * r1;
      xh2[i] = (H2[o0 + i * ostep] - mu2) * r2;
      const float n1 = xh1[i] * gg1 + b1[c], n2 = xh2[i] * gg2 + b2[c];
      sg[i] = sigmoidf_(n1);
      gp[i] = sg[i] * (1.f - sg[i]);
      gpp[i] = gp[i] * (1.f - 2.f * sg[i]);
      gy[i] = GYb[o0 + i * ostep];
      dq[i] = n2 - Xb[o0 + i * ostep];
      vx[i] = VXb[o0 + i * ostep];
      v1[i] = V1[o0 + i * ostep];
      v2[i] = V2[o0 + i * ostep];
      gn2[i] = gy[i] * sg[i];
      gn1[i] = gy[i] * dq[i] * gp[i];
      const float a1 = gn1[i] * gg1, a2 = gn2[i] * gg2;
      s[0] += v1[i]; s[1] += v1[i] * xh1[i]; s[2] += a1; s[3] += a1 * xh1[i]; s[4] += v1[i] * a1;
      s[5] += v2[i]; s[6] += v2[i] * xh2[i]; s[7] += a2; s[8] += a2 * xh2[i]; s[9] += v2[i] * a2;
    }
  }
  const float inv = 1.f / (float)C;
  group_sums<G, 10>(s, red, col, g);
#pragma unroll
  for (int k = 0; k < 10; ++k) s[k] *= inv;
  const float sc1 = r1 * (s[4] - s[0] * s[2] - s[1] * s[3]), sc2 = r2 * (s[9] - s[5] * s[7] - s[6] * s[8]);
  const float mw1 = s[0] * s[3] + s[2] * s[1], mwx1 = 2.f * s[1] * s[3];
  const float mw2 = s[5] * s[8] + s[7] * s[6], mwx2 = 2.f * s[6] * s[8];
  float* pblk = part + ((long)by * gridDim.x + bx) * 4 * C;
  float* __restrict__ dGYb = dGY + (long)b * dgy_bs;
  float* __restrict__ dXb = dX + (long)b * dx_bs;
  float e1[CPT], e2[CPT];
  float sb1 = 0.f, sbx1 = 0.f, sb2 = 0.f, sbx2 = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = g + G * i;
    const bool ok = tv && c < C;
    const float gg1 = c < C ? g1[c] : 0.f, gg2 = c < C ? g2[c] : 0.f;
    const float u1 = ok ? r1 * (v1[i] - s[0] - xh1[i] * s[1]) : 0.f;
    const float u2 = ok ? r2 * (v2[i] - s[5] - xh2[i] * s[6]) : 0.f;
    const float t1 = gg1 * u1, t2 = gg2 * u2;
    e2[i] = t1 * gy[i] * gp[i];
    e1[i] = (t2 - vx[i]) * gy[i] * gp[i] + t1 * gy[i] * dq[i] * gpp[i];
    if (ok) {
      dGYb[o0 + i * ostep] = t2 * sg[i] + t1 * dq[i] * gp[i] + vx[i] * (1.f - sg[i]);
      dXb[o0 + i * ostep] = -e2[i];
    }
    const float bb1 = gg1 * e1[i], bb2 = gg2 * e2[i];
    sb1 += bb1; sbx1 += bb1 * xh1[i]; sb2 += bb2; sbx2 += bb2 * xh2[i];
    const float p0 = col_sum(gn1[i] * u1 + e1[i] * xh1[i]), p1 = col_sum(e1[i]);
    const float p2 = col_sum(gn2[i] * u2 + e2[i] * xh2[i]), p3 = col_sum(e2[i]);
    if (col == 0 && c < C) { pblk[c] = p0; pblk[C + c] = p1; pblk[2 * C + c] = p2; pblk[3 * C + c] = p3; }
    // second-order LayerNorm terms, stored in v_i (no longer needed): -r xh s_col/C - r^2 (w - mean w - xh mean(w xh))
    const float w1 = v1[i] * s[3] + gn1[i] * gg1 * s[1], w2 = v2[i] * s[8] + gn2[i] * gg2 * s[6];
    v1[i] = -r1 * xh1[i] * sc1 - r1 * r1 * (w1 - mw1 - xh1[i] * mwx1);
    v2[i] = -r2 * xh2[i] * sc2 - r2 * r2 * (w2 - mw2 - xh2[i] * mwx2);
  }
  float bs[4] = {sb1, sbx1, sb2, sbx2};
  group_sums<G, 4>(bs, red, col, g);
  const float mb1 = bs[0] * inv, mbx1 = bs[1] * inv, mb2 = bs[2] * inv, mbx2 = bs[3] * inv;
  float* __restrict__ dH1 = dH + hb;
  float* __restrict__ dH2 = dH1 + cl;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = g + G * i;
    if (tv && c < C) {
      dH1[o0 + i * ostep] = r1 * (g1[c] * e1[i] - mb1 - xh1[i] * mbx1) + v1[i];
      dH2[o0 + i * ostep] = r2 * (g2[c] * e2[i] - mb2 - xh2[i] * mbx2) + v2[i];
    }
  }
}

// Sum the per-workgroup partial rows part[blk][n] in two levels, in a fixed order (bitwise reproducible).
#ifndef SSV_RED_ROWS
#define SSV_RED_ROWS 32
#endif
// Level 1: workgroup (x, y) folds rows y, y+RED_ROWS, ... into row y IN PLACE (every element of row y is read
// and written by the same thread, so no other workgroup touches it).  Level 2: out[i] = sum of rows 0..RED_ROWS-1.
#define RED_ROWS SSV_RED_ROWS
// (eight rows in flight per thread and trip at both levels: with two / four these launches were chains of 5 + 16 memory round trips --
//  11 + 6.4 us per use, eleven uses per training step)
__global__ __launch_bounds__(256) void reduce_partials_l1_kernel(float* __restrict__ part, int n, int nblk) {
  const int i = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (i >= n || y >= nblk) return;
  float s[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) s[q] = 0.f;
  int k = y;
  for (; k + 7 * RED_ROWS < nblk; k += 8 * RED_ROWS) {
#pragma unroll
    for (int q = 0; q < 8; ++q) s[q] += part[(long)(k + q * RED_ROWS) * n + i];
  }
  for (; k < nblk; k += RED_ROWS) s[0] += part[(long)k * n + i];
  part[(long)y * n + i] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}
__global__ __launch_bounds__(256) void reduce_partials_l2_kernel(const float* __restrict__ part, float* __restrict__ out, int n, int rows) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) s[q] = 0.f;
  int k = 0;
  for (; k + 7 < rows; k += 8) {
#pragma unroll
    for (int q = 0; q < 8; ++q) s[q] += part[(long)(k + q) * n + i];
  }
  for (; k < rows; ++k) s[0] += part[(long)k * n + i];
  out[i] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}
// One launch for moderate block counts: a workgroup owns 32 outputs; its 8 row groups each fold rows rg, rg+8, ... and
// the eight sums are combined through LDS in a fixed order.
__global__ __launch_bounds__(256) void reduce_partials_1_kernel(const float* __restrict__ part, float* __restrict__ out, int n, int nblk) {
  __shared__ float red[8][32];
  const int li = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + li;
  float a[16];                                        // sixteen rows in flight per trip (see ssv_fold_rows in misc.hip: the same fold)
#pragma unroll
  for (int u = 0; u < 16; ++u) a[u] = 0.f;
  if (i < n) {
    int k = rg;
    for (; k + 120 < nblk; k += 128) {
#pragma unroll
      for (int u = 0; u < 16; ++u) a[u] += part[(long)(k + 8 * u) * n + i];
    }
    for (; k < nblk; k += 8) a[0] += part[(long)k * n + i];
  }
  red[rg][li] = (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) + (((a[8] + a[9]) + (a[10] + a[11])) + ((a[12] + a[13]) + (a[14] + a[15])));
  __syncthreads();
  if (rg == 0 && i < n) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) t += red[r][li];
    out[i] = t;
  }
}
static int reduce_partials(float* part, float* out, int n, int nblk, hipStream_t st) {
  if (nblk <= 768) {
    hipLaunchKernelGGL(reduce_partials_1_kernel, dim3(ssv_cdiv(n, 32)), dim3(256), 0, st, (const float*)part, out, n, nblk);
    return ssv_check_launch("reduce_partials_1");
  }
  const int rows = nblk < RED_ROWS ? nblk : RED_ROWS;
  if (nblk > RED_ROWS) {
    hipLaunchKernelGGL(reduce_partials_l1_kernel, dim3(ssv_cdiv(n, 256), RED_ROWS), dim3(256), 0, st, part, n, nblk);
    SSV_TRY(ssv_check_launch("reduce_partials_l1"));
  }
  hipLaunchKernelGGL(reduce_partials_l2_kernel, dim3(ssv_cdiv(n, 256)), dim3(256), 0, st, (const float*)part, out, n, rows);
  return ssv_check_launch("reduce_partials_l2");
}

int ssv_reduce_partial_rows(const float* part, float* out, int n, int nblk, hipStream_t st) { return reduce_partials((float*)part, out, n, nblk, st); }

// ------------------------------------------------------------------------------------------------
// G = channel groups per 16-column tile (workgroup = 16 * G threads), CPT = channels per thread = ceil(C / G).
// Measured (tools/bench_ln.py, B = 32): up to 256 channels 16 groups are best at every length; 512 channels run 15-25 %
// faster on 32 groups (half the registers per thread: 224 -> ~130 VGPRs in the gate backward, so twice the waves per CU);
// the 513-channel backward is fastest on 64 groups (113 -> 92 us at L = 1300), its forward on 16.
// SSV_LN_GROUPS forces G (tuning aid).
static void ln_log(const char* name, int cpt, int g, dim3 grid, double bytes_per_elem, int B, int C, int L, int threads = 0) {
  if (!ssv_shape_log_on()) return;
  char nm[96], note[64];
  snprintf(nm, sizeof nm, "%s<%d, %d>", name, cpt, g);
  snprintf(note, sizeof note, "B=%d C=%d L=%d", B, C, L);
  ssv_shape_log(nm, grid, dim3(threads ? threads : 16 * g), 0.0, bytes_per_elem * B * C * L, note);
}
static int ln_groups(int C, bool bwd) {
  if (const char* e = ssv_tuning(SSV_T_LN_GROUPS)) { const int g = atoi(e); if (g == 16 || g == 32 || g == 64) return g; }
  if (C > 512) return bwd ? 64 : 16;
  return C > 256 ? 32 : 16;
}
#define LN_CASE(G_, N_) if (_g == G_ && _cpt <= N_) { ln_log(LN_NAME, N_, G_, grid, LN_BYTES, B, C, L); CALL(N_, G_); } else
#define LN_DISPATCH(BWD, C, L, CALL)                                    \
  do {                                                                  \
    const int _g = ln_groups(C, BWD);                                   \
    const int _cpt = ((C) + _g - 1) / _g;                               \
    if ((long)(2 * (C) + 2 * _g) * (long)L >= (1L << 31)) return ssv_fail(SSV_UNSUPPORTED, "LayerNorm: one batch item exceeds 2^31 elements"); \
    LN_CASE(16, 2) LN_CASE(16, 4) LN_CASE(16, 8) LN_CASE(16, 16) LN_CASE(16, 32) LN_CASE(16, 33) LN_CASE(16, 64) \
    LN_CASE(32, 4) LN_CASE(32, 8) LN_CASE(32, 16) LN_CASE(32, 17) LN_CASE(32, 32)                               \
    LN_CASE(64, 4) LN_CASE(64, 8) LN_CASE(64, 9) LN_CASE(64, 16)                                               \
    return ssv_fail(SSV_UNSUPPORTED, "LayerNorm over %d channels not supported (max 1024)", (C)); \
  } while (0)

int ssv_launch_ln_gate_fwd(const float* H, long h_bs, const float* X, long x_bs, const float* g1, const float* b1,
                           const float* g2, const float* b2, float* Y, long y_bs, float* stats, int B, int C, int L,
                           hipStream_t st, float* amax) {
  dim3 grid(ssv_cdiv(L, 16), B);
#define LN_NAME "ln_gate_fwd_kernel"
#define LN_BYTES 16.0
#define CALL(N, G) hipLaunchKernelGGL((ln_gate_fwd_kernel<N, G>), grid, dim3(16 * G), 0, st, H, h_bs, X, x_bs, g1, b1, g2, b2, Y, y_bs, stats, amax, C, L)
  if (C > 512) return ssv_fail(SSV_UNSUPPORTED, "highway gate over %d channels not supported (max 512)", C);
  LN_DISPATCH(false, C, L, CALL);
#undef CALL
#undef LN_NAME
#undef LN_BYTES
  return ssv_check_launch("ln_gate_fwd");
}

int ssv_ln_gate_bwd_nblk(int B, int L) { return B * ssv_cdiv(L, 16); }      // upper bound of the partial rows a backward launch writes (buffer sizes)
// Columns per tile / 16 of the backward launch for this shape (1: the 16-column kernels; 2 or 4: the wide-tile kernels), and the partial rows it
// writes -- one per tile, compact.  The ONE place the choice is made: the launchers and every caller that sums the rows ask here.
// wide tiles (see ln_gate_bwd_wide_kernel), measured in-step against the 16-column kernel (round 4, B = 32): C = 512: L = 186 -21 %, L = 1300
// -11.5 %; C = 256: L = 1300 -13.6 %, L = 325 and 650 equal (192 / 352 workgroups of 1024 threads leave CUs idle: the 16-column kernel stays)
int ssv_ln_gate_bwd_vec(int C, int L, bool has_amax) {
  if (C <= 512 && C % 64 == 0 && C >= 128 && (C > 256 ? L >= 64 : L >= 1024) && (long)2 * C * L * 4 < (1L << 31) &&
      (!has_amax || ssv_amax_rows_(L) >= ssv_cdiv(L, 16)))
    return C <= 256 ? 4 : 2;
  return 1;
}
int ssv_ln_act_bwd_vec(int C, int L, bool has_amax) {
  if (C > 256 && C <= 576 && L >= 64 && (long)C * L * 4 < (1L << 31) && (!has_amax || ssv_amax_rows_(L) >= ssv_cdiv(L, 16))) return 4;
  return 1;
}
// Workgroups per batch item of the PERSISTENT gate backward (ln_gate_bwd_pers_kernel), or 0 when the shape runs the tile kernels: 256
// channels only (8 channel steps per thread on 32 groups; C = 512 keeps ln_gate_bwd_wide), B * wpi workgroups = the CUs (SSV_LN_PERSIST=<slots> overrides,
// 0 switches the kernel off), every workgroup at least one 16-column sub-tile, one scale-list entry per workgroup.
int ssv_launch_ln_gate_bwd_pers(const float* dY, long dy_bs, const float* H, const float* X, long x_bs, const float* stats, const float* g1, const float* b1,
                                const float* g2, const float* b2, float* dH, float* dXres, long dx_bs, float* part, float* amax, int B, int C, int L, int wpi,
                                hipStream_t st);       // norm_pers.hip
int ssv_ln_gate_bwd_wpi(int B, int C, int L, bool has_amax) {
  int slots = 256;
  if (const char* e = ssv_tuning(SSV_T_LN_PERSIST)) slots = atoi(e);
  if (slots <= 0 || C != 256 || (long)2 * C * L >= (1L << 31) || B > slots) return 0;
  int wpi = slots / B;
  if (wpi > ssv_cdiv(L, 16)) wpi = ssv_cdiv(L, 16);
  if (has_amax && wpi > ssv_amax_rows_(L)) wpi = ssv_amax_rows_(L);
  return wpi < 1 ? 0 : wpi;
}
int ssv_ln_gate_bwd_rows(int B, int C, int L, bool has_amax) {
  if (const int wpi = ssv_ln_gate_bwd_wpi(B, C, L, has_amax)) return B * wpi;
  return B * ssv_cdiv(L, 16 * ssv_ln_gate_bwd_vec(C, L, has_amax));
}
int ssv_ln_act_bwd_rows(int B, int C, int L, bool has_amax) { return B * ssv_cdiv(L, 16 * ssv_ln_act_bwd_vec(C, L, has_amax)); }

int ssv_launch_ln_gate_bwd(const float* dY, long dy_bs, const float* H, const float* X, long x_bs, const float* stats,
                           const float* g1, const float* b1, const float* g2, const float* b2, float* dH, float* dXres,
                           long dx_bs, float* part, float* pgrads /* [6][C] */, int B, int C, int L, hipStream_t st, float* amax) {
  dim3 grid(ssv_cdiv(L, 16), B);
#define LN_NAME "ln_gate_bwd_kernel"
#define LN_BYTES 28.0
#define CALL(N, G) hipLaunchKernelGGL((ln_gate_bwd_kernel<N, G>), grid, dim3(16 * G), 0, st, dY, dy_bs, H, X, x_bs, stats, g1, b1, g2, b2, dH, dXres, dx_bs, part, amax, C, L)
  if (C > 512) return ssv_fail(SSV_UNSUPPORTED, "highway gate over %d channels not supported (max 512)", C);
  const int vec = ssv_ln_gate_bwd_vec(C, L, amax != nullptr);
  const int wpi = ssv_ln_gate_bwd_wpi(B, C, L, amax != nullptr);
  if (wpi > 0) {
    dim3 gp(B * wpi);
    ln_log("ln_gate_bwd_pers_kernel", C / 32, 32, gp, LN_BYTES, B, C, L);
    SSV_TRY(ssv_launch_ln_gate_bwd_pers(dY, dy_bs, H, X, x_bs, stats, g1, b1, g2, b2, dH, dXres, dx_bs, part, amax, B, C, L, wpi, st));
  } else if (vec > 1) {
#define WIDE(N, V) do { dim3 gw(ssv_cdiv(L, 16 * V), B); ln_log("ln_gate_bwd_wide_kernel", N, V, gw, LN_BYTES, B, C, L, 1024); \
      hipLaunchKernelGGL((ln_gate_bwd_wide_kernel<N, V>), gw, dim3(1024), 0, st, dY, dy_bs, H, X, x_bs, stats, g1, b1, g2, b2, dH, dXres, dx_bs, part, amax, C, L); } while (0)
    if (C <= 128) WIDE(2, 4); else if (C <= 256) WIDE(4, 4); else WIDE(8, 2);
#undef WIDE
  } else
  LN_DISPATCH(true, C, L, CALL);
#undef CALL
#undef LN_NAME
#undef LN_BYTES
  SSV_TRY(ssv_check_launch("ln_gate_bwd"));
  if (!pgrads) return 0;        // the caller sums the partial rows itself (fused with the weight-gradient slab sum)
  return reduce_partials(part, pgrads, 6 * C, ssv_ln_gate_bwd_rows(B, C, L, amax != nullptr), st);
}

int ssv_launch_ln_act_fwd(const float* X, long x_bs, const float* gam, const float* bet, float* Y, long y_bs, float* stats,
                          int B, int C, int L, int act, hipStream_t st, float* amax) {
  dim3 grid(ssv_cdiv(L, 16), B);
#define LN_NAME "ln_act_fwd_kernel"
#define LN_BYTES 8.0
#define CALL(N, G) hipLaunchKernelGGL((ln_act_fwd_kernel<N, G>), grid, dim3(16 * G), 0, st, X, x_bs, gam, bet, Y, y_bs, stats, amax, C, L, act)
  LN_DISPATCH(false, C, L, CALL);
#undef CALL
#undef LN_NAME
#undef LN_BYTES
  return ssv_check_launch("ln_act_fwd");
}

int ssv_launch_ln_act_bwd(const float* dY, long dy_bs, const float* X, long x_bs, const float* stats, const float* gam,
                          const float* bet, float* dX, long dx_bs, float* part, float* pgrads /* [3][C] */, int B, int C,
                          int L, int act, hipStream_t st, float* amax) {
  dim3 grid(ssv_cdiv(L, 16), B);
#define LN_NAME "ln_act_bwd_kernel"
#define LN_BYTES 12.0
#define CALL(N, G) hipLaunchKernelGGL((ln_act_bwd_kernel<N, G>), grid, dim3(16 * G), 0, st, dY, dy_bs, X, x_bs, stats, gam, bet, dX, dx_bs, part, amax, C, L, act)
  const int vec = ssv_ln_act_bwd_vec(C, L, amax != nullptr);
  if (vec > 1) {
#define WIDE(N, V) do { dim3 gw(ssv_cdiv(L, 16 * V), B); ln_log("ln_act_bwd_wide_kernel", N, V, gw, LN_BYTES, B, C, L, 1024); \
      hipLaunchKernelGGL((ln_act_bwd_wide_kernel<N, V>), gw, dim3(1024), 0, st, dY, dy_bs, X, x_bs, stats, gam, bet, dX, dx_bs, part, amax, C, L, act); } while (0)
    if (C <= 512) WIDE(8, 4); else WIDE(9, 4);
#undef WIDE
  } else
  LN_DISPATCH(true, C, L, CALL);
#undef CALL
#undef LN_NAME
#undef LN_BYTES
  SSV_TRY(ssv_check_launch("ln_act_bwd"));
  if (!pgrads) return 0;        // the caller sums the partial rows itself (fused with the weight-gradient slab sum)
  return reduce_partials(part, pgrads, 3 * C, B * ssv_cdiv(L, 16 * vec), st);
}

// ---- second-order launchers (critics: at most 256 channels, 16 groups) ------------------------------------------------
#define LN2_DISPATCH(C, CALL)                                                                   \
  do {                                                                                          \
    const int _cpt = ((C) + 15) / 16;                                                           \
    if (_cpt <= 2) { CALL(2, 16); } else if (_cpt <= 4) { CALL(4, 16); } else if (_cpt <= 8) { CALL(8, 16); } \
    else if (_cpt <= 16) { CALL(16, 16); }                                                      \
    else return ssv_fail(SSV_UNSUPPORTED, "second-order LayerNorm over %d channels not supported (max 256)", (C)); \
  } while (0)

int ssv_launch_ln_bwd2(const float* V, long v_bs, const float* GN, long gn_bs, const float* X, long x_bs, const float* stats,
                       const float* gam, float* dGN, long dgn_bs, float* dX, long dx_bs, float* part, float* dgamma,
                       int B, int C, int L, hipStream_t st) {
  dim3 grid(ssv_cdiv(L, 16), B);
  if ((long)(C + 32) * (long)L >= (1L << 31)) return ssv_fail(SSV_UNSUPPORTED, "LayerNorm: one batch item exceeds 2^31 elements");
#define CALL(N, G) hipLaunchKernelGGL((ln_bwd2_kernel<N, G>), grid, dim3(16 * G), 0, st, V, v_bs, GN, gn_bs, X, x_bs, stats, gam, dGN, dgn_bs, dX, dx_bs, part, C, L)
  LN2_DISPATCH(C, CALL);
#undef CALL
  SSV_TRY(ssv_check_launch("ln_bwd2"));
  return reduce_partials(part, dgamma, C, (int)(grid.x * grid.y), st);
}

int ssv_launch_ln_gate_bwd2(const float* VH, const float* VX, long vx_bs, const float* GY, long gy_bs, const float* H,
                            const float* X, long x_bs, const float* stats, const float* g1, const float* b1, const float* g2,
                            const float* b2, float* dGY, long dgy_bs, float* dH, float* dX, long dx_bs, float* part,
                            float* pgrads /* [4][C] */, int B, int C, int L, hipStream_t st) {
  dim3 grid(ssv_cdiv(L, 16), B);
  if ((long)(2 * C + 32) * (long)L >= (1L << 31)) return ssv_fail(SSV_UNSUPPORTED, "LayerNorm: one batch item exceeds 2^31 elements");
#define CALL(N, G) hipLaunchKernelGGL((ln_gate_bwd2_kernel<N, G>), grid, dim3(16 * G), 0, st, VH, VX, vx_bs, GY, gy_bs, H, X, x_bs, stats, g1, b1, g2, b2, dGY, dgy_bs, dH, dX, dx_bs, part, C, L)
  LN2_DISPATCH(C, CALL);
#undef CALL
  SSV_TRY(ssv_check_launch("ln_gate_bwd2"));
  return reduce_partials(part, pgrads, 4 * C, (int)(grid.x * grid.y), st);
}
