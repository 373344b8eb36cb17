// Partial maxima of |x| (split-fp16 operand scales) and the weight pre-split into fp16 / bf16 hi and lo planes in MFMA fragment order.
#include "bf3_common.h"

// ---- partial maxima of |x| (split-fp16 operand scales) ----------------------------------------------------------------
// grid (npb, B): workgroup (i, b) scans the i-th of npb equal pieces of item b (n dense floats at x + b * x_bs) and writes
// out[b * npb + i].  A maximum is order-independent, so the result does not depend on the partition.
// Four 16-byte loads per thread are in flight before the first maximum is taken: with one workgroup or a few per CU the scan is a chain of
// memory round trips otherwise (8 pieces per item, 10 round trips each at C = 256 / L = 325: 7.3 us per launch, 10.9 us at 513 x 1300).
__device__ __forceinline__ float ssv_absmax4(const f32x4 q) { return fmaxf(fmaxf(fabsf(q[0]), fabsf(q[1])), fmaxf(fabsf(q[2]), fabsf(q[3]))); }
__device__ __forceinline__ float ssv_absmax_scan(const float* __restrict__ xb, const long lo, const long hi) {
  float v = 0.f;
  if ((((size_t)xb) & 15) == 0) {
    long i = lo + 4L * threadIdx.x;
    for (; i + 3 * 1024 + 3 < hi; i += 4096) {
      const f32x4 q0 = *reinterpret_cast<const f32x4*>(xb + i), q1 = *reinterpret_cast<const f32x4*>(xb + i + 1024);
      const f32x4 q2 = *reinterpret_cast<const f32x4*>(xb + i + 2048), q3 = *reinterpret_cast<const f32x4*>(xb + i + 3072);
      v = fmaxf(v, fmaxf(fmaxf(ssv_absmax4(q0), ssv_absmax4(q1)), fmaxf(ssv_absmax4(q2), ssv_absmax4(q3))));
    }
    for (; i + 3 < hi; i += 1024) v = fmaxf(v, ssv_absmax4(*reinterpret_cast<const f32x4*>(xb + i)));
    for (; i < hi; ++i) v = fmaxf(v, fabsf(xb[i]));              // at most 3 elements, one thread
  } else {
    for (long i = lo + threadIdx.x; i < hi; i += 256) v = fmaxf(v, fabsf(xb[i]));
  }
  return v;
}
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, long x_bs, long n, float* __restrict__ out, int npb) {
  __shared__ float sm[4];
  const float* __restrict__ xb = x + (long)blockIdx.y * x_bs;
  const long piece = (((n + npb - 1) / npb) + 3) & ~3L;
  const long lo = (long)blockIdx.x * piece, hi = min(lo + piece, n);
  float v = ssv_absmax_scan(xb, lo, hi);
  v = ssv_wg_max<4>(v, sm);
  if (threadIdx.x == 0) out[(long)blockIdx.y * npb + blockIdx.x] = v;
}
int ssv_launch_absmax(const float* x, long x_bs, int B, long n, float* out, int npb, hipStream_t st) {
  hipLaunchKernelGGL(absmax_kernel, dim3(npb, B), dim3(256), 0, st, x, x_bs, n, out, npb);
  return ssv_check_launch("absmax");
}

// ---- weight pre-split ---------------------------------------------------------------------------------------------
// Source w[m*sm + k*sk + j*sj] (M x K per tap j).  Output: bf16 hi / lo planes in MFMA FRAGMENT ORDER,
//   [tap j][row block mb = m/16][chunk ch = k/32][k-group kg = (k/8)%4][row m%16][8 x bf16],
// rows padded to a multiple of 16 and k to Kpad (zeros).  One (mb, ch) block is 1 KB and is exactly what a wave's
// 64 lanes load as one A fragment (lane = kg*16 + m%16, 16 bytes each): the load covers 8 whole cache lines.  With the
// plain [row][k] order the same fragment touched 16 lines and used half of each, doubling L2->L1 traffic for weights.
__global__ __launch_bounds__(256) void pack_split_kernel(const float* __restrict__ w, __bf16* __restrict__ hi, __bf16* __restrict__ lo,
                                                         int M, int K, int Kpad, int KT, long sm, long sk, long sj, int perm_h,
                                                         int nch_total, int ch_off) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int MB = (M + 15) >> 4, NCH = Kpad >> 5;
  const long n = (long)KT * MB * NCH * 512;
  if (i >= n) return;
  const int e = (int)(i & 7), r16 = (int)((i >> 3) & 15), kg = (int)((i >> 7) & 3);
  const long blk = i >> 9;
  const int ch = (int)(blk % NCH), mb = (int)((blk / NCH) % MB), j = (int)(blk / ((long)NCH * MB));
  const int m = mb * 16 + r16, k = ch * 32 + kg * 8 + e;
  const int ms = perm_h ? (m & 3) * perm_h + (m >> 2) : m;       // LSTM: gate-interleaved output rows
  const float v = (m < M && k < K) ? w[(long)ms * sm + (long)k * sk + (long)j * sj] : 0.f;
  const __bf16 h = (__bf16)v;
  const long d = ((((long)j * MB + mb) * nch_total + ch + ch_off) << 9) + (i & 511);
  hi[d] = h;
  lo[d] = (__bf16)(v - (float)h);
}
// split-fp16 planes of one dense weight: same element map (perm_h / nch_total / ch_off as in pack_split_kernel); the scale comes from
// the `nlist` partial maxima at `list` (written by absmax_kernel just before -- of this weight alone, or of every weight that shares
// the scale), the inverse scale goes to *inv_out for the GEMM's epilogue.
__global__ __launch_bounds__(256) void pack_split_f16_kernel(const float* __restrict__ w, _Float16* __restrict__ hi, _Float16* __restrict__ lo,
                                                             int M, int K, int Kpad, int KT, long sm, long sk, long sj, int perm_h,
                                                             int nch_total, int ch_off, const float* __restrict__ list, int nlist, float* __restrict__ inv_out) {
  __shared__ float smx[4];
  float sc, inv;
  ssv_pow2_scale(ssv_list_max<4>(list, nlist, smx), sc, inv);
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i == 0) *inv_out = inv;
  const int MB = (M + 15) >> 4, NCH = Kpad >> 5;
  const long n = (long)KT * MB * NCH * 512;
  if (i >= n) return;
  const int e = (int)(i & 7), r16 = (int)((i >> 3) & 15), kg = (int)((i >> 7) & 3);
  const long blk = i >> 9;
  const int ch = (int)(blk % NCH), mb = (int)((blk / NCH) % MB), j = (int)(blk / ((long)NCH * MB));
  const int m = mb * 16 + r16, k = ch * 32 + kg * 8 + e;
  const int ms = perm_h ? (m & 3) * perm_h + (m >> 2) : m;       // LSTM: gate-interleaved output rows
  const float v = (m < M && k < K) ? w[(long)ms * sm + (long)k * sk + (long)j * sj] : 0.f;
  const _Float16 h = (_Float16)(v * sc);
  const long d = ((((long)j * MB + mb) * nch_total + ch + ch_off) << 9) + (i & 511);
  hi[d] = h;
  lo[d] = (_Float16)__builtin_fmaf(v, sc, -(float)h);
}
int ssv_launch_pack_split_f16_list(const float* w, void* hi, void* lo, int M, int K, int Kpad, int KT, long sm, long sk, long sj, int perm_h,
                                   const float* list, int nlist, float* inv_out, hipStream_t st, int nch_total, int ch_off) {
  const long n = (long)KT * ((M + 15) / 16 * 16) * Kpad;
  if (nch_total <= 0) { nch_total = Kpad / 32; ch_off = 0; }
  hipLaunchKernelGGL(pack_split_f16_kernel, dim3(ssv_cdiv(n, 256)), dim3(256), 0, st, w, (_Float16*)hi, (_Float16*)lo, M, K, Kpad, KT, sm, sk, sj, perm_h,
                     nch_total, ch_off, list, nlist, inv_out);
  return ssv_check_launch("pack_split_f16");
}
int ssv_launch_pack_split_f16(const float* w, long w_elems, void* hi, void* lo, int M, int K, int Kpad, int KT, long sm, long sk, long sj, float* aux,
                              hipStream_t st) {
  SSV_TRY(ssv_launch_absmax(w, 0, 1, w_elems, aux, 64, st));
  return ssv_launch_pack_split_f16_list(w, hi, lo, M, K, Kpad, KT, sm, sk, sj, 0, aux, 64, aux + 64, st, 0, 0);
}
int ssv_launch_pack_split(const float* w, void* hi, void* lo, int M, int K, int Kpad, int KT, long sm, long sk, long sj, int perm_h, hipStream_t st,
                          int nch_total, int ch_off) {
  const long n = (long)KT * ((M + 15) / 16 * 16) * Kpad;
  if (nch_total <= 0) { nch_total = Kpad / 32; ch_off = 0; }
  hipLaunchKernelGGL(pack_split_kernel, dim3(ssv_cdiv(n, 256)), dim3(256), 0, st, w, (__bf16*)hi, (__bf16*)lo, M, K, Kpad, KT, sm, sk, sj, perm_h,
                     nch_total, ch_off);
  return ssv_check_launch("pack_split");
}

// Many weights in one launch (resident pre-split weights, ssv_conv_pack_multi): workgroup -> job by binary search over
// the jobs' first_block, then the same element map as pack_split_kernel, 1024 elements per workgroup.
#define PACK_PER_BLOCK 256
// split-fp16: partial maxima of every weight first (grid (SSV_PACK_AMAX_PER_WEIGHT, njobs / 2); the forward job 2i and the
// transposed job 2i + 1 read the same dense tensor of M K KT floats), then the pack kernel scales by the resulting power of two
// and leaves 2^-e at the job's inv_out for the GEMM epilogues.
__global__ __launch_bounds__(256) void pack_amax_multi_kernel(const ssv_pack_job* __restrict__ jobs, float* __restrict__ amax) {
  __shared__ float sm[4];
  const ssv_pack_job j = jobs[2 * blockIdx.y];
  const long n = (long)j.M * j.K * j.KT;
  const long piece = (((n + gridDim.x - 1) / gridDim.x) + 3) & ~3L;
  const long lo = (long)blockIdx.x * piece, hi = min(lo + piece, n);
  float v = ssv_absmax_scan(j.w, lo, hi);          // (round 4: one 4-byte load per thread and trip before -- 43 us per launch, 1.1 TB/s)
  v = ssv_wg_max<4>(v, sm);
  if (threadIdx.x == 0) amax[(long)blockIdx.y * gridDim.x + blockIdx.x] = v;
}
// One thread = one 16-byte slot position (16-row block mb, 32-channel chunk ch, k-group kg, row r16) in ALL taps: it reads 8 x KT
// weights -- contiguous for the forward planes (sk == KT: 8 x KT floats in a row), 8 short runs one channel-stride apart for the
// transposed planes (lanes of a quarter wave are neighbouring rows there, i.e. neighbouring runs) -- and writes KT (hi, lo) slot pairs.
// (The first version took one ELEMENT per thread: 2-byte stores, 12-byte-stride loads; 90 us per launch, 2.1 TB/s.)
template <int F16, int KT>
__device__ __forceinline__ void pack_slot(const ssv_pack_job& j, const long slot, const float sc) {
  typedef __attribute__((address_space(1))) const float gfloat;               // (table pointers: see ssv_global)
  typedef unsigned vu4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(1))) vu4 guint4;
  const int MB = (j.M + 15) >> 4, NCH = j.Kpad >> 5;
  const int r16 = (int)(slot & 15), kg = (int)((slot >> 4) & 3);
  const long blk = slot >> 6;
  const int ch = (int)(blk % NCH), mb = (int)(blk / NCH);
  const int m = mb * 16 + r16, k0 = ch * 32 + kg * 8;
  const long n = (long)KT * MB * NCH * 512;
  guint4* hip = (guint4*)j.planes;
  guint4* lop = (guint4*)((char*)j.planes + (((size_t)n * 2 + 255) & ~(size_t)255));
  gfloat* w = (gfloat*)j.w + (long)m * j.sm + (long)k0 * j.sk;
  float v[8][KT];
  if (m < j.M && k0 + 8 <= j.K) {
    if (j.sk == KT) {
      float flat[8 * KT];
#pragma unroll
      for (int q = 0; q < 8 * KT; ++q) flat[q] = w[q];
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int t = 0; t < KT; ++t) v[e][t] = flat[e * KT + t];
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int t = 0; t < KT; ++t) v[e][t] = w[(long)e * j.sk + t];
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
      for (int t = 0; t < KT; ++t) v[e][t] = (m < j.M && k0 + e < j.K) ? w[(long)e * j.sk + t] : 0.f;
  }
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    unsigned hw[4], lw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned short hb[2], lb[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float x = v[2 * q + u][t];
        if constexpr (F16) {
          const _Float16 h = (_Float16)(x * sc);
          const _Float16 l = (_Float16)__builtin_fmaf(x, sc, -(float)h);
          hb[u] = __builtin_bit_cast(unsigned short, h); lb[u] = __builtin_bit_cast(unsigned short, l);
        } else {
          const __bf16 h = (__bf16)x;
          const __bf16 l = (__bf16)(x - (float)h);
          hb[u] = __builtin_bit_cast(unsigned short, h); lb[u] = __builtin_bit_cast(unsigned short, l);
        }
      }
      hw[q] = hb[0] | ((unsigned)hb[1] << 16); lw[q] = lb[0] | ((unsigned)lb[1] << 16);
    }
    const long o = (((long)t * MB + mb) * NCH + ch) * 64 + kg * 16 + r16;     // 16-byte slot index
    hip[o] = vu4{hw[0], hw[1], hw[2], hw[3]};
    lop[o] = vu4{lw[0], lw[1], lw[2], lw[3]};
  }
}
template <int F16>
__global__ __launch_bounds__(256) void pack_multi_kernel(const ssv_pack_job* __restrict__ jobs, int njobs, const float* __restrict__ amax) {
  __shared__ float smx[4];
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {                                   // last job with first_block <= blockIdx.x
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const ssv_pack_job j = jobs[lo];
  float sc = 1.f, inv = 1.f;
  if constexpr (F16) {
    ssv_pow2_scale(ssv_list_max<4>(amax + (long)(lo >> 1) * SSV_PACK_AMAX_PER_WEIGHT, SSV_PACK_AMAX_PER_WEIGHT, smx), sc, inv);
    if ((int)blockIdx.x == j.first_block && threadIdx.x == 0 && j.inv_out) *j.inv_out = inv;
  }
  const long slots = (long)((j.M + 15) >> 4) * (j.Kpad >> 5) * 64;
  const long slot = (long)((int)blockIdx.x - j.first_block) * PACK_PER_BLOCK + threadIdx.x;
  if (slot >= slots) return;
  if (j.KT == 3) pack_slot<F16, 3>(j, slot, sc);
  else if (j.KT == 1) pack_slot<F16, 1>(j, slot, sc);
  else if (j.KT == 2) pack_slot<F16, 2>(j, slot, sc);
}
int ssv_pack_job_blocks(const ssv_pack_job& j) {
  const long slots = (long)((j.M + 15) / 16) * (j.Kpad / 32) * 64;      // 16-byte slot positions per tap; a thread takes one in all taps
  return (int)((slots + PACK_PER_BLOCK - 1) / PACK_PER_BLOCK);
}
int ssv_launch_pack_multi(const ssv_pack_job* jobs_dev, int njobs, int nblocks, float* amax_ws, hipStream_t st) {
  if (amax_ws) {
    hipLaunchKernelGGL(pack_amax_multi_kernel, dim3(SSV_PACK_AMAX_PER_WEIGHT, njobs / 2), dim3(256), 0, st, jobs_dev, amax_ws);
    SSV_TRY(ssv_check_launch("pack_amax_multi"));
    hipLaunchKernelGGL(pack_multi_kernel<1>, dim3(nblocks), dim3(256), 0, st, jobs_dev, njobs, (const float*)amax_ws);
  } else {
    hipLaunchKernelGGL(pack_multi_kernel<0>, dim3(nblocks), dim3(256), 0, st, jobs_dev, njobs, (const float*)nullptr);
  }
  return ssv_check_launch("pack_multi");
}

