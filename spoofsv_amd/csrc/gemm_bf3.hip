// Split-bf16 ("bf16x3") implicit-GEMM kernels for gfx950: fp32 operands are split as x = hi + lo with
// hi = bf16(x), lo = bf16(x - hi); a product is accumulated in fp32 as  a_lo*b_hi + a_hi*b_lo + a_hi*b_hi  on
// v_mfma_f32_16x16x32_bf16 (the dropped lo*lo term is 2^-18 relative).  Three bf16 MFMAs cover 16x16x32 MACs in
// 48 cycles where the fp32 MFMA needs 8 x 32 = 256: 5.3x the matrix-core rate at ~1e-5 relative error, inside
// the 1e-3 budget of the hot path.  ssv_set_precision(0) switches every caller back to the exact fp32 MFMA kernels.
//
// Operand fragments of the 16x16x32 MFMA hold 8 consecutive k per lane (16 bytes).  Both LDS images are laid out
// [k-group of 8][row][8 x bf16]: a fragment read is one ds_read_b128 at (kg*rows + row)*16 -- consecutive rows are
// consecutive 16-byte slots, and because every k-group plane is a multiple of 256 bytes the four 16-lane groups of
// ds_read_b128 hit disjoint banks (conflict-free for any row offset, hence for any dilation shift).
//
//   pack_split / pack_multi: weights -> bf16 hi / lo planes in MFMA fragment order (per call, or resident: one launch for
//     all weights of a model after each optimizer step).
//   gemm_nn_bf3 (4 waves) / gemm_nn_bf3w (8-16 waves, k=1 over long rows): Conv1d forward / data gradient, deconvolution
//     halves, LSTM products (EPI = 1: cell update in the epilogue, wavefront over layers in grid.y).  Weight fragments go
//     L2 -> registers; the input tile is split while it is staged (one fp32 global read per element; 8 channels of one
//     column form a slot, two LDS images = one barrier per chunk), taps address the same slots at column offsets.
//   gemm_nt_bf3: Conv1d weight gradient.  The reduction runs over time, so a dilation shift would be a misaligned
//     shift along k; each tap gets its own staged copy of the input rows at its exact shift, one tap per step.
//   All kernels re-number their workgroups so that every XCD walks a contiguous tile range (ssv_xcd_order).
#include <stdio.h>
#include <type_traits>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "ssv_common.h"
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));   // 4-byte aligned 16-byte load

__device__ __forceinline__ void split8(const float (&v)[8], uint4& hi, uint4& lo) {
  bf16x8 h, l;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 t = (__bf16)v[i];
    h[i] = t;
    l[i] = (__bf16)(v[i] - (float)t);
  }
  hi = __builtin_bit_cast(uint4, h);
  lo = __builtin_bit_cast(uint4, l);
}
// split-fp16 (ssv_common.h, "split-fp16"): hi = fp16(v s), lo = fp16(v s - hi); s is a power of two, so v s is exact and
// v s - hi is an exact fp32 number: the only roundings are the two conversions.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
// Written with v_fma_mix{lo,hi}_f16 (fp32 x fp32 + {0, -fp16} -> fp16 half of a register): 16 VALU instructions per 8
// elements, the scale included.  From the plain C++ form (kept below for the host pass) hipcc builds 4 v_pk_mul_f32 + 4
// v_cvt_pk_f16_f32 + 8 v_cvt_f32_f16 + 4 v_pk_fma_f32 + 4 v_cvt_pk_f16_f32 = 24, against 20 for the bf16 split -- measured
// as +5..7 % on every GEMM kernel of the step.  Bit-identical results (checked on the device, signed zeros aside).
__device__ __forceinline__ void split2h(float a, float b, float s, unsigned& h, unsigned& l) {
#if defined(__HIP_DEVICE_COMPILE__)
  // s: the (wave-uniform) scale, in a scalar register -- one constant-bus operand per instruction, no VGPR for it
  // (Round 4, measured in-step and on the GE2E embedder, nothing moved by more than the run-to-run 1 %: (i) scale by v_pk_mul_f32, hi by
  //  v_cvt_pk_f16_f32, lo by two v_fma_mix -- 2 full-rate + 2 half-rate instructions per pair instead of these 4 half-rate ones
  //  (tools/probe/valu_rate.hip: 5 vs 9-11 cycles), bit-identical; (ii) the slot's validity folded into a per-thread scale, sparing the
  //  select per element.  The split is not what these kernels wait for; MI355X_MICROARCH.md prices packed f32 VALU beside MFMAs as an anti-lever.)
  asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(h) : "v"(a), "s"(s));
  asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(h) : "v"(b), "s"(s));
  asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a), "s"(s), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(b), "s"(s), "v"(h));
#else
  (void)a; (void)b; (void)s; h = l = 0;
#endif
}
__device__ __forceinline__ void split8h(const float (&v)[8], float s, uint4& hi, uint4& lo) {
  unsigned h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) split2h(v[2 * i], v[2 * i + 1], s, h[i], l[i]);
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  lo = make_uint4(l[0], l[1], l[2], l[3]);
}
template <int F16>
__device__ __forceinline__ void split8s(const float (&v)[8], float s, uint4& hi, uint4& lo) {
  if constexpr (F16) split8h(v, s, hi, lo); else split8(v, hi, lo);
}
// one PAIR of elements -> one packed dword of the hi plane and one of the lo plane
template <int F16>
__device__ __forceinline__ void split_pair(float a, float b, float s, unsigned& h, unsigned& l) {
  if constexpr (F16) split2h(a, b, s, h, l);
  else {
    const __bf16 ha = (__bf16)a, hb = (__bf16)b;
    const __bf16 la = (__bf16)(a - (float)ha), lb = (__bf16)(b - (float)hb);
    h = (unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
    l = (unsigned)__builtin_bit_cast(unsigned short, la) | ((unsigned)__builtin_bit_cast(unsigned short, lb) << 16);
  }
}
// The weight-gradient kernel's order of the 8 time steps of a window inside a fragment: dword q holds steps (q, q + 4).  Any order is
// right as long as both operands use it; this one lets a window loaded as two 4-dword tuples be split IN PLACE, pair by pair
// (dword q of the first tuple and dword q of the second go in, dword q of hi and dword q of lo come out).
template <int F16>
__device__ __forceinline__ void split8p(const float (&v)[8], float s, uint4& hi, uint4& lo) {
  unsigned h[4], l[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) split_pair<F16>(v[q], v[q + 4], s, h[q], l[q]);
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  lo = make_uint4(l[0], l[1], l[2], l[3]);
}
// one 16x16x32 MFMA on 16-byte operand fragments: bf16 or fp16 inputs, fp32 accumulate
template <int F16>
__device__ __forceinline__ f32x4 mma16(const uint4& a, const uint4& b, const f32x4& c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

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

// ---- NN ---------------------------------------------------------------------------------------------------------------
// Waves split the M axis, so a weight row is only ever used by ONE wave: weight fragments go straight from global memory
// (L2-resident, pre-split, fragment-shaped 16-byte loads) into MFMA operand registers, one K chunk ahead (two register
// sets, the chunk loop is unrolled by two).  Only the input tile, which all four waves share, is staged in LDS -- this
// removes 2/3 of the LDS writes and 1/4 of the LDS reads of a version that staged both operands.
// (third waves per SIMD for the small k = 1 tile: 168 VGPRs in the split-bf16 form, 174 in the split-fp16 one without the bound -- +22 % time)
// Tuning builds only (-DSSV_NN_STAMP): thread 0 of workgroup (0, 0) records s_memtime at five points of every K chunk; ssv_debug_nn_stamps().
#ifdef SSV_NN_STAMP
#ifndef SSV_NN_STAMP_WG
#define SSV_NN_STAMP_WG 0     // (stamp builds) 1: the launch's last workgroup instead of its first
#endif
__device__ unsigned long long ssv_nn_stamps[128];
__device__ unsigned long long ssv_nn_rt[4096];        // s_memrealtime (100 MHz, one clock for the device) at entry / exit of the first 2048 workgroups
extern "C" int ssv_debug_nn_realtime(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ssv_nn_rt), sizeof(ssv_nn_rt)); }
#define NN_RT(which) do { const unsigned w_ = blockIdx.y * gridDim.x + blockIdx.x; if (threadIdx.x == 0 && w_ < 2048u) ssv_nn_rt[2 * w_ + (which)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define NN_STAMP(k) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && (unsigned)ch < 15u) ssv_nn_stamps[ch * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#define NN_STAMP_AT(slot) do { if ((SSV_NN_STAMP_WG ? (blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1) : (blockIdx.x == 0 && blockIdx.y == 0)) && threadIdx.x == 0) \
    ssv_nn_stamps[120 + (slot)] = __builtin_readcyclecounter(); } while (0)       /* SSV_NN_STAMP_WG=1: the launch's last workgroup instead of its first */
extern "C" int ssv_debug_nn_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ssv_nn_stamps), sizeof(ssv_nn_stamps)); }
#else
#define NN_STAMP(k) do {} while (0)
#define NN_STAMP_AT(slot) do {} while (0)
#define NN_RT(which) do {} while (0)
#endif
#ifndef SSV_NN_XONE
#define SSV_NN_XONE 1        // one-path input prefetch where it measured faster in-step (k = 1 tiles -2 %, 128 x 112 k = 3 tiles -2.5 %; the 64-row and
                             // 96-column k = 3 tiles and every 54-column-halo tile were equal or up to 15 % SLOWER with it and keep the two-form prefetch)
#endif
#define SSV_NN_XBUF(KT, WM, NT) (!((KT) == 3 && (NT) == 6))   // input rows by buffer loads (ssv_buf) or through pointers: in-step, per tile -- the k = 1
                                                             // tiles are 5-12 % faster with buffer loads, the 96-column k = 3 tiles 4-6 % with pointers, the rest equal

#ifndef SSV_NN_HALO_SMALL
#define SSV_NN_HALO_SMALL 16  // k = 3 layers whose taps span at most this many columns run the narrow-halo instantiation (tuning builds: -1 = never)
#endif
// waves per SIMD the register allocation must leave room for (the second __launch_bounds__ argument).  Round 5: the 128 x 112 k = 3 tile with the
// 16-column halo at THREE (168 VGPRs, 6 spilled, three workgroups per CU instead of two): 177.6 -> 170.4 us in-step over its ten launches.
// (The same for the 128 x 96 tile: 32 spilled, 63.2 -> 70.3 us; the 64 x 96 tile at four, 128 VGPRs: 47.6 -> 49.5 us.  Not kept.)
#define SSV_NNB_WAVES(KT, WM, NT, EPI, HW) \
  (((KT) == 1 && (WM) == 2 && (NT) == 4 && (EPI) == 0) || ((KT) == 3 && (WM) == 2 && (NT) == 7 && (EPI) == 0 && (HW) == 16) ? 3 : 2)
// HW: the halo (columns beyond the tile that the taps reach) the instantiation stages for -- 54 (dilation 27, any form) or 16: most layers of the
// models have dilation 1 or 3, and with the 54-column halo a third of the loads, splits and LDS stores of their chunks went into columns no tap reads
// (112 + 54 -> 176 staged columns = 3 slots per thread; 112 + 16 -> 128 = 2).
template <int KT, int WM, int NT, int EPI, int F16, int HW = 54>
__global__ __launch_bounds__(256, SSV_NNB_WAVES(KT, WM, NT, EPI, HW)) void gemm_nn_bf3_kernel(const GemmNNB p, const int mtiles, const int smin, const int span) {
  constexpr int BM = 64 * WM, BN = 16 * NT;
  constexpr int HALO = (KT == 1) ? 0 : HW;
  constexpr int WX = ((BN + HALO + 15) / 16) * 16;         // staged columns, plane = WX*16 B = multiple of 256 B
  constexpr int X_SLOTS = 4 * WX;
  constexpr int NX = (X_SLOTS + 255) / 256;
  // two images of the staged input tile: the MFMAs of chunk c read image c & 1 while chunk c+1 is split into the other
  // one -- one barrier per chunk, and the split (VALU) runs under the MFMAs instead of between two barriers
  // (the epilogue re-uses the memory to turn the accumulator tiles into row-contiguous stores: 4 waves x 16 rows x (BN + 4))
  // (+ 4 * WM * BN (mean, M2) pairs behind the parked tiles when the caller wants the output's column statistics)
  constexpr int IMG = 2 * X_SLOTS, EPI_U4 = (4 * 16 * (BN + 4) + 4 * WM * BN * 2) / 4;
  constexpr int LDS_U4 = 2 * IMG > EPI_U4 ? 2 * IMG : EPI_U4;
  __shared__ uint4 lds_all[LDS_U4];
  uint4 (*lds)[IMG] = reinterpret_cast<uint4 (*)[IMG]>(lds_all);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned wg = ssv_xcd_order(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);   // see ssv_xcd_order
  const int bxx = (int)(wg % gridDim.x), b = (int)(wg / gridDim.x);
  const int mt = bxx % mtiles, ntile = bxx / mtiles;
  const int m0 = mt * BM, n0 = ntile * BN;
  const float* __restrict__ Xb = p.X + (long)b * p.sxb;
  // LSTM wavefront (see GemmNNB): layer / frame of this grid.y entry, the second K segment, the chunks to run
  const float* __restrict__ X2b = nullptr;
  int lstm_layer = 0, lstm_t = 0;
  bool lstm_l0 = false;                       // this entry is layer 0 riding in a wavefront launch (GemmNNB::A0hi)
  if constexpr (EPI == 1) {
    if (p.lstm_D > 0) {
      lstm_layer = p.lstm_lo + b;
      lstm_t = p.lstm_s - lstm_layer;
      const long HN = (long)p.perm_h * p.N;
      lstm_l0 = p.A0hi != nullptr && lstm_layer == 0;
      Xb = p.lstm_out + ((long)max(lstm_layer - 1, 0) * p.lstm_D + lstm_t % p.lstm_D) * HN;
      X2b = p.lstm_out + ((long)lstm_layer * p.lstm_D + (lstm_t + p.lstm_D - 1) % p.lstm_D) * HN - (long)p.xsplit * 32 * (long)p.sxc;
      if (lstm_l0) { Xb = X2b + (long)p.xsplit * 32 * (long)p.sxc; X2b = nullptr; }      // one segment: the layer's own h_{t-1}
    }
  }
  // chunks to run: all of K, except that an LSTM entry at its first frame has no h_{t-1} segment (layer 0 riding along: nothing but that segment)
  const int nchunks_all = lstm_l0 ? p.xsplit : p.Kpad / 32;
  const int nchunks = (EPI == 1 && p.lstm_D > 0 && lstm_t == 0) ? (lstm_l0 ? 0 : p.xsplit) : nchunks_all;
  const int W = BN + span;
  const int kq = lane >> 4, nq = lane & 15;

  f32x4 acc[WM][NT];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Weight fragments.  KT == 3: one register set; tap j of the NEXT chunk is re-loaded into its registers right after
  // tap j's MFMAs of the current chunk have been issued (2/3 of a chunk of lead time).  KT == 1: two sets, alternating.
  constexpr int NSET = (KT == 1) ? 2 : 1;
  uint4 Ah_[NSET][KT][WM], Al_[NSET][KT][WM];
  float rx[NX][8];

  // this lane's weight fragments: row block (m0 + wave*WM*16 + i*16) / 16, 16 bytes at lane*16 of each 1 KB (mb, chunk) block
  const int MB = (p.M + 15) >> 4;
  long arow[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int mb = (m0 + wave * WM * 16 + i * 16) >> 4;
    arow[i] = (long)min(mb, MB - 1) * nchunks_all * 512 + lane * 8;     // blocks past M re-read the last one: never stored
  }
  const long aplane = (long)MB * nchunks_all * 512;

  // uniform byte base (batch item / layer, tap, chunk) + a per-lane 32-bit byte offset: the saddr form of global_load
  unsigned arowb[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i) arowb[i] = (unsigned)(arow[i] * 2);
  // (LSTM wavefront with layer 0 riding along: entry 0 reads the planes A0hi / A0lo, entry b >= 1 the planes of layer b at (b - 1) * sab)
  const long aent = (EPI == 1 && p.A0hi) ? (long)(b > 0 ? b - 1 : 0) * p.sab : (long)b * p.sab;
  const __amdgpu_buffer_rsrc_t rsAh = ssv_buf(lstm_l0 ? p.A0hi : p.Ahi + aent), rsAl = ssv_buf(lstm_l0 ? p.A0lo : p.Alo + aent);   // (see ssv_buf)
  auto loadA = [&](int set, int j, int ch) {
    const unsigned ub = (unsigned)((j * aplane + (long)ch * 512) * 2);                                  // wave-uniform byte offset
    // (buffer loads everywhere: equal or 1-3 % faster than loads through pointers, measured in-step per tile)
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      Ah_[set][j][i] = ssv_buf_u4(rsAh, arowb[i], ub);
      Al_[set][j][i] = ssv_buf_u4(rsAl, arowb[i], ub);
    }
  };
  // Input staging, two halves.  prefetchX only ISSUES loads (raw values, addresses clamped into the batch item so every
  // load is legal); validity masks are applied in commitX one chunk later, right before the split.  A mask applied at
  // load time would make hipcc wait for each load (or branch around it), serialising 24 L2 round trips per chunk.
  // Address arithmetic is hoisted: element (chunk ch, slot channel 8*kg+i, column) lives at
  //   [Xb + (ch*32 + i)*L]  (wave-uniform, scalar ALU)  +  [8*kg*L + column]  (per thread, computed once),
  // and the column mask is computed once; only a ragged last chunk (Kc % 32 != 0) needs per-channel clamps and masks.
  const int Lrow = (int)p.sxc;
  const __amdgpu_buffer_rsrc_t rsX = ssv_buf(Xb), rsX2 = ssv_buf(X2b ? X2b : Xb);        // (see ssv_buf)
  unsigned voff[NX], voffb[NX];
  bool cvs[NX];
#pragma unroll
  for (int r = 0; r < NX; ++r) {
    const int e = tid + 256 * r;
    const int kg = e / WX, col = e % WX;
    const int gcol = n0 + smin + col;
    cvs[r] = e < X_SLOTS && col < W && gcol >= 0 && gcol < p.Lx;
    voff[r] = (unsigned)((e < X_SLOTS ? 8 * kg : 0) * Lrow + min(max(gcol, 0), p.Lx - 1) * p.sxn);
    voffb[r] = voff[r] * 4u;           // BYTE offset of the buffer load (a row's offset is added as its scalar operand)
  }
  const bool ragged = (p.Kc & 31) != 0;
  // split-fp16: xs = 2^ex scales the input while it is split, us = 2^-(ea + ex) the accumulators in the epilogue.  The weights'
  // inverse scale is requested here (a scalar load) and first USED in the epilogue: nothing in the prologue waits for it.  The input's
  // scale is needed before the first split; x_namax == 0 (the LSTM products: |h| < 1 by construction) means the constant 2^14, no list.
  float xs = 1.f, xinv = 1.f, ainv = 1.f;
  if constexpr (F16) ainv = *p.a_inv;
  auto scales = [&]() {
    if constexpr (F16) {
      if (p.x_namax == 0) { xs = 16384.f; xinv = 1.f / 16384.f; }
      else {
        float sc, inv;
        ssv_pow2_scale(ssv_wave_list_max(p.x_amax + (long)b * p.x_amax_bs, p.x_namax), sc, inv);
        xs = ssv_uniform(sc);
        xinv = ssv_uniform(inv);
      }
    }
  };
  // Convolutions (EPI == 0) prefetch on ONE path: a buffer whose range is the batch item's Kc rows, so the channels of a ragged last chunk past
  // Kc read 0 (tools/probe/buf_oob.hip: voffset + soffset is checked against the range, per dword) and no "last, partial chunk" form is needed.
  // Not for the branch: hipcc lays an if / else out as two tests in a row, its s_waitcnt bookkeeping then sees a path on which NEITHER form ran, and
  // in front of every chunk's first MFMA it waited for all but the weight fragments' own loads -- i.e. for the input loads issued a few hundred
  // cycles earlier, one exposed round trip per chunk (round 5; the steady / tail split below had removed only the "is there a chunk c + 2" tests).
  const __amdgpu_buffer_rsrc_t rsXr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Xb), 0,
      (int)(((long)(p.Kc - 1) * Lrow + (long)(p.Lx - 1) * p.sxn + 1) * 4), 0x00020000);
  auto prefetchX = [&](int ch) {
    if constexpr (EPI == 0 && SSV_NN_XONE && (KT == 1 || (WM == 2 && NT == 7 && HW == 16))) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const unsigned so = (unsigned)((ch * 32 + i) * Lrow) * 4u;                // uniform row offset: scalar arithmetic
#pragma unroll
        for (int r = 0; r < NX; ++r) rx[r][i] = ssv_buf_f32(rsXr, voffb[r], so);
      }
    } else if (!ragged || ch + 1 < nchunks) {
      const bool seg2 = EPI == 1 && X2b && ch >= p.xsplit;                       // (LSTM: the h_{t-1} segment of K)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if constexpr (SSV_NN_XBUF(KT, WM, NT)) {
          const unsigned so = (unsigned)((ch * 32 + i) * Lrow) * 4u;              // uniform row offset: scalar arithmetic
#pragma unroll
          for (int r = 0; r < NX; ++r) rx[r][i] = (EPI == 1 && seg2) ? ssv_buf_f32(rsX2, voffb[r], so) : ssv_buf_f32(rsX, voffb[r], so);
        } else {
          const char* __restrict__ rowp = (const char*)((seg2 ? X2b : Xb) + (long)(ch * 32 + i) * Lrow);     // uniform
#pragma unroll
          for (int r = 0; r < NX; ++r) rx[r][i] = *reinterpret_cast<const float*>(rowp + voffb[r]);
        }
      }
    } else {                                                                   // last, partial chunk: clamp channels
#pragma unroll
      for (int r = 0; r < NX; ++r) {
        const int e = tid + 256 * r;
        const int kg = (e < X_SLOTS) ? e / WX : 0;
        const unsigned colo = voff[r] - (unsigned)(8 * kg * Lrow);
#pragma unroll
        for (int i = 0; i < 8; ++i) rx[r][i] = ssv_buf_f32(rsX, ((unsigned)min(ch * 32 + 8 * kg + i, p.Kc - 1) * (unsigned)Lrow + colo) * 4u, 0u);
      }
    }
  };
  auto commitX = [&](int ch) {
    uint4* Xh = lds[ch & 1];
    uint4* Xl = lds[ch & 1] + X_SLOTS;
    const bool last_ragged = ragged && ch + 1 == nchunks;
#pragma unroll
    for (int r = 0; r < NX; ++r) {
      const int e = tid + 256 * r;
      if (e < X_SLOTS) {
        float v[8];
        if (!last_ragged) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = cvs[r] ? rx[r][i] : 0.f;
        } else {
          const int kg = e / WX;
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = (cvs[r] && ch * 32 + 8 * kg + i < p.Kc) ? rx[r][i] : 0.f;
        }
        uint4 h, l;
        split8s<F16>(v, xs, h, l);
        Xh[e] = h; Xl[e] = l;            // slot index = kg*WX + col = e
      }
    }
  };

  int offj[KT];
#pragma unroll
  for (int j = 0; j < KT; ++j) offj[j] = p.shift[j] - smin;

  // The input fragments of column block t + FD are read from LDS before the MFMAs of block t are issued (2 reads, 3 WM MFMAs per
  // block): hipcc on its own issues a block's reads right in front of its MFMAs and parks the wave for the LDS latency NT times per tap.
  constexpr int FD = NT > 1 ? 1 : 0;
  auto tap = [&](int set, int j, int ch) {
    const uint4* Xh = lds[ch & 1];
    const uint4* Xl = lds[ch & 1] + X_SLOTS;
    uint4 fb[FD + 1][2];
    auto frag = [&](int t, uint4 (&f)[2]) __attribute__((always_inline)) {
      const int xs_ = kq * WX + t * 16 + nq + offj[j];
      f[0] = Xh[xs_]; f[1] = Xl[xs_];
    };
#pragma unroll
    for (int t = 0; t < FD; ++t) frag(t, fb[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t + FD < NT) frag(t + FD, fb[(t + FD) % (FD + 1)]);
      if (FD > 0) __builtin_amdgcn_sched_barrier(0);                          // or the scheduler sinks the reads back to their use
      const uint4 bh = fb[t % (FD + 1)][0];
      const uint4 bl = fb[t % (FD + 1)][1];
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        acc[i][t] = mma16<F16>(Al_[set][j][i], bh, acc[i][t]);
        acc[i][t] = mma16<F16>(Ah_[set][j][i], bl, acc[i][t]);
        acc[i][t] = mma16<F16>(Ah_[set][j][i], bh, acc[i][t]);
      }
    }
  };

  // Prologue: chunk 0 staged, chunk 1 in flight.  Chunk c: MFMAs on image c & 1 (weight fragments of chunk c+1 re-loaded
  // tap by tap behind them), then chunk c+1 is split into the other image and the loads of chunk c+2 are issued.
  // The loop comes in two forms: STEADY, for the chunks whose every load / commit is known to be due, has no "is there a chunk
  // c + 2" tests -- not for the branches: hipcc's s_waitcnt bookkeeping merges over all paths, and with the prefetch under a test it
  // waited for vmcnt(0) in front of each chunk's first MFMA, i.e. for the input loads issued a few hundred cycles earlier (one
  // exposed L2 round trip per chunk; the weight-gradient kernel has the full story at its STEADY).  The last chunks run the tested form.
  using ST_ = std::integral_constant<bool, true>;
  using TL_ = std::integral_constant<bool, false>;
  NN_STAMP_AT(0);
  NN_RT(0);
#ifdef SSV_NN_STAMP
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) ssv_nn_stamps[127] = __builtin_readcyclecounter();     // the first workgroup's entry, for the ramp
#endif
  if constexpr (KT == 1) {
    NN_STAMP_AT(1);
    if (nchunks > 0) {
      loadA(0, 0, 0);
      prefetchX(0);
      scales();
      commitX(0);
      if (nchunks > 1) { prefetchX(1); loadA(1, 0, 1); }
    }
    __syncthreads();
    auto pair = [&](auto steady, int ch) __attribute__((always_inline)) -> bool {
      constexpr bool ST = decltype(steady)::value;
      NN_STAMP(0);
      tap(0, 0, ch);
      NN_STAMP(1);
      if (!ST && ch + 1 >= nchunks) return false;
      commitX(ch + 1);
      NN_STAMP(2);
      if (ST || ch + 2 < nchunks) { prefetchX(ch + 2); loadA(0, 0, ch + 2); }
      NN_STAMP(3);
      __syncthreads();
      NN_STAMP(4);
      tap(1, 0, ch + 1);
      if (ST || ch + 2 < nchunks) {
        commitX(ch + 2);
        if (ST || ch + 3 < nchunks) { prefetchX(ch + 3); loadA(1, 0, ch + 3); }
      }
      __syncthreads();
      NN_STAMP(5);
      return true;
    };
    int ch = 0;
    for (; ch + 3 < nchunks; ch += 2) pair(ST_{}, ch);
    for (; ch < nchunks; ch += 2)
      if (!pair(TL_{}, ch)) break;
  } else {
#pragma unroll
    for (int j = 0; j < KT; ++j) loadA(0, j, 0);
    prefetchX(0);
    scales();
    commitX(0);
    if (nchunks > 1) prefetchX(1);
    __syncthreads();
    NN_STAMP_AT(1);
    auto chunk = [&](auto steady, int ch) __attribute__((always_inline)) {
      constexpr bool ST = decltype(steady)::value;
      const bool more = ST || ch + 1 < nchunks;
      NN_STAMP(0);
#pragma unroll
      for (int j = 0; j < KT; ++j) {
        tap(0, j, ch);
        __builtin_amdgcn_sched_barrier(0);      // keep the re-load behind this tap's MFMAs, and later taps' LDS reads behind it
        if (more) loadA(0, j, ch + 1);
      }
      NN_STAMP(1);
      if (more) {
        commitX(ch + 1);
        NN_STAMP(2);
        if (ST || ch + 2 < nchunks) prefetchX(ch + 2);
      }
      NN_STAMP(3);
      __syncthreads();
      NN_STAMP(4);
    };
    int ch = 0;
    // (the steady form holds more values live: 140 -> 190 VGPRs for the 64 x 112 tile, whose hot launches are 768 workgroups and need
    // three per CU to run in one round -- +20 % on it; the 64 x 96 tile's launches are 512 workgroups and gain 7 % from it)
    for (; ch + 2 < nchunks; ++ch) chunk(ST_{}, ch);
    for (; ch < nchunks; ++ch) chunk(TL_{}, ch);
  }

  NN_STAMP_AT(2);
  const float us = F16 ? ssv_uniform(xinv * ainv) : 1.f;
  float* __restrict__ Cb = p.C + (long)b * p.scb;
  const float* __restrict__ Rb = p.R ? p.R + (long)b * p.srb : nullptr;
  if constexpr (EPI == 1) {
    if (p.A0hi && !lstm_l0) Rb = nullptr;         // the input projection in R belongs to layer 0 alone
    // Fused LSTM cell (torch gate order i, f, g, o).  Rows were packed gate-interleaved, so the four accumulator rows a
    // lane holds for a 16-row tile (rows kq*4 .. kq*4+3) are the four gates of ONE hidden unit at column nq.
    const int H = p.perm_h;
    float* cst = p.cstate;
    const float* __restrict__ bia = p.bias ? p.bias + (long)b * p.sbb : nullptr;
    const float* __restrict__ bib = p.bias_b ? p.bias_b + (long)b * p.sbb : nullptr;
    bool first = p.first != 0;
    float* cnew = p.cstate;                       // where c_t goes (same place as c_{t-1} unless every frame is kept)
    float* __restrict__ gsave = nullptr;
    if (p.lstm_D > 0) {
      const long HN = (long)H * p.N;
      cst += (long)lstm_layer * HN;
      cnew = cst;
      Cb = p.lstm_out + ((long)lstm_layer * p.lstm_D + lstm_t % p.lstm_D) * HN;
      first = lstm_t == 0;
      if (p.gates_out) {
        cnew = p.cstate + ((long)lstm_layer * p.lstm_D + lstm_t) * HN;
        cst = cnew - HN;                            // c_{t-1}: the previous frame of the same layer (not read at t = 0)
        gsave = p.gates_out + ((long)lstm_layer * p.lstm_D + lstm_t) * 4 * HN;
      }
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const int row0 = m0 + wave * WM * 16 + i * 16 + kq * 4;       // = 4 * unit
      const int u = row0 >> 2;
      if (u >= H) continue;
      float add[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) add[r] = (bia ? bia[r * H + u] : 0.f) + (bib ? bib[r * H + u] : 0.f);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int gn = n0 + t * 16 + nq;
        if (gn >= p.N) continue;
        float gte[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) gte[r] = (F16 ? acc[i][t][r] * us : acc[i][t][r]) + add[r] + (Rb ? Rb[(long)(row0 + r) * p.srm + gn] : 0.f);
        const float gi = 1.f / (1.f + expf(-gte[0])), gf = 1.f / (1.f + expf(-gte[1]));
        const float gg = tanhf(gte[2]), go = 1.f / (1.f + expf(-gte[3]));
        const long ci = (long)u * p.N + gn;
        const float cn = (first ? 0.f : gf * cst[ci]) + gi * gg;
        cnew[ci] = cn;
        if (gsave) {
          const long HN = (long)H * p.N;
          gsave[ci] = gi; gsave[HN + ci] = gf; gsave[2 * HN + ci] = gg; gsave[3 * HN + ci] = go;
        }
        Cb[(long)u * p.scm + gn] = go * tanhf(cn);
      }
    }
    return;
  }
  if (p.scn == 1) {
    // Row-contiguous stores.  An MFMA accumulator holds 4 rows x 1 column per lane, so storing it directly writes 64-byte
    // pieces of 4 different rows per instruction (measured: the epilogue was 7.3 of 36.6 us at C = 256, L = 325).  The tile
    // goes through LDS instead (free after the K loop): every wave parks its 16 x BN block row-major and reads it back as
    // 16-byte vectors along the row -- a store instruction then covers up to 448 contiguous bytes of one or two rows.
    constexpr int LDW = BN + 4;                                 // row pitch in floats: 16-byte aligned, bank-conflict free
    float* stage = reinterpret_cast<float*>(lds_all) + wave * 16 * LDW;
    __syncthreads();                                            // every wave is done reading the last chunk's image
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const int rbase = m0 + wave * WM * 16 + i * 16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gm = rbase + kq * 4 + r;
        const int gmc = min(gm, p.M - 1);
        const int gb = p.perm_h ? (gmc & 3) * p.perm_h + (gmc >> 2) : gmc;
        float add = 0.f;
        if (p.bias) add += p.bias[gb];
        if (p.bias_b) add += p.bias_b[(long)b * p.sbb + gb];
#pragma unroll
        for (int t = 0; t < NT; ++t) stage[(kq * 4 + r) * LDW + t * 16 + nq] = (F16 ? acc[i][t][r] * us : acc[i][t][r]) + add;
      }
      // the block is private to the wave: no workgroup barrier, the LDS operations of one wave complete in order
#pragma unroll
      for (int it = 0; it < NT; ++it) {
        const int e = lane + 64 * it;                           // 16-byte vector index in the 16 x BN block
        const int row = e / (BN / 4), c4 = e % (BN / 4);
        const int gm = rbase + row, gn = n0 + c4 * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * LDW + c4 * 4);
        if (gm < p.M && gn < p.N) {
          float* dst = Cb + (long)gm * p.scm + gn;
          if (gn + 3 < p.N) {
            f4u o = {v[0], v[1], v[2], v[3]};
            if (Rb) { const f4u rr = *reinterpret_cast<const f4u*>(Rb + (long)gm * p.srm + gn); o += rr; }
            *reinterpret_cast<f4u*>(dst) = o;
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (gn + q < p.N) dst[q] = v[q] + (Rb ? Rb[(long)gm * p.srm + gn + q] : 0.f);
          }
        }
      }
      if (p.colstats) {
        // (after the row stores: they are asynchronous and the kernel cannot end before they drain, so they go first)
        // column statistics of this 16-row block: a lane reads ITS column of the parked block (consecutive lanes, consecutive
        // words: conflict-free), two passes over 16 values in registers
        float* cst = reinterpret_cast<float*>(lds_all) + 4 * 16 * LDW + (wave * WM + i) * BN * 2;
#pragma unroll
        for (int c0 = 0; c0 < BN; c0 += 64) {
          const int c = c0 + lane;
          if (c < BN) {
            float v[16], sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { v[r] = stage[r * LDW + c]; sum += v[r]; }
            const float mean = sum * (1.f / 16.f);
            float m2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d = v[r] - mean; m2 += d * d; }
            cst[2 * c] = mean; cst[2 * c + 1] = m2;
          }
        }
      }
    }
    if (p.colstats) {
      // four 16-row blocks -> one 64-row group (Chan's merge of equal counts), WM groups per tile
      __syncthreads();
      const float* cst = reinterpret_cast<const float*>(lds_all) + 4 * 16 * LDW;
      for (int e = tid; e < WM * BN; e += 256) {
        const int grp = e / BN, c = e % BN, gn = n0 + c;
        if (gn >= p.N) continue;
        float mu[4], mean = 0.f, m2 = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) { mu[q] = cst[((grp * 4 + q) * BN + c) * 2]; mean += mu[q]; m2 += cst[((grp * 4 + q) * BN + c) * 2 + 1]; }
        mean *= 0.25f;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float d = mu[q] - mean; m2 += 16.f * d * d; }
        float* dst = p.colstats + (((long)b * (p.M >> 6) + (m0 >> 6) + grp) * p.N + gn) * 2;
        dst[0] = mean; dst[1] = m2;
      }
    }
    NN_STAMP_AT(3);
    NN_RT(1);
    return;
  }
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gm = m0 + wave * WM * 16 + i * 16 + kq * 4 + r;
      if (gm >= p.M) continue;
      const int gb = p.perm_h ? (gm & 3) * p.perm_h + (gm >> 2) : gm;   // bias index in the caller's (torch) row order
      float add = 0.f;
      if (p.bias) add += p.bias[gb];
      if (p.bias_b) add += p.bias_b[(long)b * p.sbb + gb];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int gn = n0 + t * 16 + nq;
        if (gn >= p.N) continue;
        float v = (F16 ? acc[i][t][r] * us : acc[i][t][r]) + add;
        if (Rb) v += Rb[(long)gm * p.srm + gn];
        Cb[(long)gm * p.scm + (long)gn * p.scn] = v;
      }
    }
}

// ---- NN, wide workgroup ------------------------------------------------------------------------------------------------
// The 4-wave kernel above moves (A tile + X tile) bytes from L2 per chunk for only 128 x 112 outputs: at the bf16 MFMA rate
// that is ~34 B/clk/CU, and an ablation (MFMAs removed: 356 of 503 us remain; loads removed: 167 us) shows it is bound by
// L2->CU traffic, not by the matrix cores.  Here a workgroup has 4 x NWN waves (M x N): the NWN wave columns share one
// weight tile, so both operands are staged in LDS once per workgroup and the L2 bytes per MAC drop by ~2x (128 x 336 tile:
// 16 B/clk/CU at the full MFMA rate).  Same LDS image layout [k-group][row][8 x bf16] (conflict-free b128 reads), same
// "issue raw loads, mask at commit" staging and hoisted addressing as above.
// (two 8-wave workgroups per CU = 4 waves per SIMD need <= 128 VGPRs: the split-bf16 form of the 128 x 192 tile has 122, the split-fp16 one 134
// without the bound -- +42 % time; the second __launch_bounds__ argument is waves per SIMD in HIP)
#ifndef SSV_NNBW_XROW
#define SSV_NNBW_XROW 1      // (tuning builds: 0 = M = 128 j + 1 on the 16-wave kernel with a fifth row tile, as before)
#endif
#ifndef SSV_NNBW_PARK
#define SSV_NNBW_PARK 1      // (tuning builds: 0 = the output stored straight from the accumulator layout, 64-byte pieces of 16 rows per instruction)
#endif
template <int KT, int WM, int NT, int NWN, int F16, int XR = 0>
__global__ __launch_bounds__(256 * NWN, (NWN == 2 && KT == 1) ? 4 : 1) void gemm_nn_bf3w_kernel(const GemmNNB p, const int mtiles, const int smin, const int span) {
  constexpr int T = 256 * NWN;
  constexpr int BM = 64 * WM, BN = 16 * NT * NWN;
  constexpr int HALO = (KT == 1) ? 0 : 54;
  constexpr int WX = ((BN + HALO + 15) / 16) * 16;
  constexpr int A_SLOTS = KT * 4 * BM, X_SLOTS = 4 * WX;
  constexpr int NA = (A_SLOTS + T - 1) / T, NX = (X_SLOTS + T - 1) / T;
  // (the epilogue re-uses the staging memory to park every wave's 16 x (16 NT) blocks for row-contiguous stores, as gemm_nn_bf3_kernel does)
  constexpr int LDWP = 16 * NT + 4, PARK_U4 = SSV_NNBW_PARK ? (4 * NWN * 16 * LDWP) / 4 : 0;
  constexpr int STAGE_U4 = 2 * A_SLOTS + 2 * X_SLOTS;
  __shared__ uint4 lds[STAGE_U4 > PARK_U4 ? STAGE_U4 : PARK_U4];
  uint4* Ah = lds;
  uint4* Al = lds + A_SLOTS;
  uint4* Xh = lds + 2 * A_SLOTS;
  uint4* Xl = Xh + X_SLOTS;
  // XR: the launch covers rows 0 .. M - 2 with its tiles (p.M is the FULL row count); row M - 1 is added by the workgroups of row tile 0 as fp32
  // dot products of its weights (xw, staged once) with the raw input values every staging thread holds before it splits them
  constexpr int XW_MAX = XR ? 1056 : 1;
  __shared__ float xw[XW_MAX];
  __shared__ float xsum[XR ? 4 * WX : 1];
  static_assert(!XR || (KT == 1 && X_SLOTS <= 2 * T), "extra row: k = 1, at most two slots per thread");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 3, wn = wave >> 2;
  const unsigned wg = ssv_xcd_order(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);   // see ssv_xcd_order
  const int bxx = (int)(wg % gridDim.x), b = (int)(wg / gridDim.x);
  const int mt = bxx % mtiles, ntile = bxx / mtiles;
  const int m0 = mt * BM, n0 = ntile * BN;
  const float* __restrict__ Xb = p.X + (long)b * p.sxb;
  const int W = BN + span;
  const int nchunks = p.Kpad / 32;
  const int kq = lane >> 4, nq = lane & 15;
  const int Mt = XR ? p.M - 1 : p.M;                      // rows the tiles cover
  const bool xr_on = XR && mt == 0;
  float xacc[NX];
#pragma unroll
  for (int r = 0; r < NX; ++r) xacc[r] = 0.f;
  if constexpr (XR) {
    if (xr_on) for (int k = tid; k < nchunks * 32; k += T) xw[k] = k < p.Kc ? p.xrow_w[(long)k * p.xrow_sk] : 0.f;    // (visible after the loop's first barrier)
  }

  f32x4 acc[WM][NT];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  uint4 rah[NA], ral[NA];
  float rx[NX][8];

  // weight staging: slot f -> (tap j, row block, k-group kg, row%16) in the packed fragment order, so a wave reads 1 KB
  // of contiguous global memory per load
  const int MB = (p.M + 15) >> 4;
  auto a_off = [&](int r) -> long {
    const int f = min(tid + T * r, A_SLOTS - 1);
    const int r16 = f & 15, kg = (f >> 4) & 3, mbl = (f >> 6) % (BM / 16), j = f / (4 * BM);
    return (((long)j * MB + min((m0 >> 4) + mbl, MB - 1)) * nchunks) * 512 + (kg * 16 + r16) * 8;
  };
  auto a_slot = [&](int r) -> int {
    const int f = min(tid + T * r, A_SLOTS - 1);
    const int r16 = f & 15, kg = (f >> 4) & 3, mbl = (f >> 6) % (BM / 16), j = f / (4 * BM);
    return (j * 4 + kg) * BM + mbl * 16 + r16;
  };
  const int Lrow = (int)p.sxc;
  unsigned voff[NX];
  unsigned cvmask = 0;
#pragma unroll
  for (int r = 0; r < NX; ++r) {
    const int e = tid + T * r;
    const int kg = e / WX, col = e % WX;
    const int gcol = n0 + smin + col;
    if (e < X_SLOTS && col < W && gcol >= 0 && gcol < p.Lx) cvmask |= 1u << r;
    voff[r] = (unsigned)((e < X_SLOTS ? 8 * kg : 0) * Lrow + min(max(gcol, 0), p.Lx - 1));
  }
  const bool ragged = (p.Kc & 31) != 0;
  float xs = 1.f, us = 1.f;           // split-fp16 scales, see gemm_nn_bf3_kernel

  // buffer loads (see ssv_buf): per-thread byte offsets fixed for the kernel, the chunk / row offset a scalar operand
  const __amdgpu_buffer_rsrc_t rsAh = ssv_buf(p.Ahi), rsAl = ssv_buf(p.Alo), rsX = ssv_buf(Xb);
  unsigned aoffb[NA];
#pragma unroll
  for (int r = 0; r < NA; ++r) aoffb[r] = (unsigned)(a_off(r) * 2);
  auto prefetch = [&](int ch) {
#pragma unroll
    for (int r = 0; r < NA; ++r) {
      rah[r] = ssv_buf_u4(rsAh, aoffb[r], (unsigned)ch * 1024u);
      ral[r] = ssv_buf_u4(rsAl, aoffb[r], (unsigned)ch * 1024u);
    }
    if (!ragged || ch + 1 < nchunks) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const unsigned so = (unsigned)((ch * 32 + i) * Lrow) * 4u;                                     // wave-uniform row offset
#pragma unroll
        for (int r = 0; r < NX; ++r) rx[r][i] = ssv_buf_f32(rsX, voff[r] * 4u, so);
      }
    } else {
#pragma unroll
      for (int r = 0; r < NX; ++r) {
        const int e = tid + T * r;
        const int kg = (e < X_SLOTS) ? e / WX : 0;
        const unsigned colo = voff[r] - (unsigned)(8 * kg * Lrow);
#pragma unroll
        for (int i = 0; i < 8; ++i) rx[r][i] = ssv_buf_f32(rsX, ((unsigned)min(ch * 32 + 8 * kg + i, p.Kc - 1) * (unsigned)Lrow + colo) * 4u, 0u);
      }
    }
  };
  auto commit = [&](int ch) {
#pragma unroll
    for (int r = 0; r < NA; ++r)
      if (tid + T * r < A_SLOTS) { const int sl = a_slot(r); Ah[sl] = rah[r]; Al[sl] = ral[r]; }
    const bool last_ragged = ragged && ch + 1 == nchunks;
#pragma unroll
    for (int r = 0; r < NX; ++r) {
      const int e = tid + T * r;
      if (e < X_SLOTS) {
        float v[8];
        if (!last_ragged) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = ((cvmask >> r) & 1) ? rx[r][i] : 0.f;
        } else {
          const int kg = e / WX;
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = (((cvmask >> r) & 1) && ch * 32 + 8 * kg + i < p.Kc) ? rx[r][i] : 0.f;
        }
        if constexpr (XR) {
          if (xr_on) {
            const float* wq = xw + ch * 32 + 8 * (e / WX);
#pragma unroll
            for (int i = 0; i < 8; ++i) xacc[r] = __builtin_fmaf(wq[i], v[i], xacc[r]);
          }
        }
        uint4 h, l;
        split8s<F16>(v, xs, h, l);
        Xh[e] = h; Xl[e] = l;
      }
    }
  };

  int offj[KT];
#pragma unroll
  for (int j = 0; j < KT; ++j) offj[j] = p.shift[j] - smin;
  const int arow = kq * BM + wm * WM * 16 + nq;
  const int xcol = kq * WX + wn * NT * 16 + nq;

  prefetch(0);
  if constexpr (F16) {
    float sc, inv;
    ssv_pow2_scale(ssv_wave_list_max(p.x_amax + (long)b * p.x_amax_bs, p.x_namax), sc, inv);
    xs = ssv_uniform(sc);
    us = ssv_uniform(inv * *p.a_inv);
  }
  for (int ch = 0; ch < nchunks; ++ch) {
    __syncthreads();
    commit(ch);
    __syncthreads();
    if (ch + 1 < nchunks) prefetch(ch + 1);
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      uint4 ah[WM], al[WM];
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        ah[i] = Ah[j * 4 * BM + arow + i * 16];
        al[i] = Al[j * 4 * BM + arow + i * 16];
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int xs_ = xcol + t * 16 + offj[j];
        const uint4 bh = Xh[xs_];
        const uint4 bl = Xl[xs_];
#pragma unroll
        for (int i = 0; i < WM; ++i) {
          acc[i][t] = mma16<F16>(al[i], bh, acc[i][t]);
          acc[i][t] = mma16<F16>(ah[i], bl, acc[i][t]);
          acc[i][t] = mma16<F16>(ah[i], bh, acc[i][t]);
        }
      }
    }
  }

  float* __restrict__ Cb = p.C + (long)b * p.scb;
  const float* __restrict__ Rb = p.R ? p.R + (long)b * p.srb : nullptr;
  if constexpr (SSV_NNBW_PARK) {
    if (p.scn == 1) {
      __syncthreads();                                            // every wave is done reading the last chunk's image
      float* pk = reinterpret_cast<float*>(lds) + wave * 16 * LDWP;
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const int rbase = m0 + wm * WM * 16 + i * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gmc = min(rbase + kq * 4 + r, Mt - 1);
          float add = 0.f;
          if (p.bias) add += p.bias[gmc];
          if (p.bias_b) add += p.bias_b[(long)b * p.sbb + gmc];
#pragma unroll
          for (int t = 0; t < NT; ++t) pk[(kq * 4 + r) * LDWP + t * 16 + nq] = (F16 ? acc[i][t][r] * us : acc[i][t][r]) + add;
        }
        // (the block is private to the wave: its LDS operations complete in order)
#pragma unroll
        for (int it = 0; it < NT; ++it) {
          const int e = lane + 64 * it;
          const int row = e / (4 * NT), c4 = e % (4 * NT);
          const int gm = rbase + row, gn = n0 + wn * NT * 16 + c4 * 4;
          const f32x4 v = *reinterpret_cast<const f32x4*>(pk + row * LDWP + c4 * 4);
          if (gm < Mt && gn < p.N) {
            float* dst = Cb + (long)gm * p.scm + gn;
            if (gn + 3 < p.N) {
              f4u o = {v[0], v[1], v[2], v[3]};
              if (Rb) { const f4u rr = *reinterpret_cast<const f4u*>(Rb + (long)gm * p.srm + gn); o += rr; }
              *reinterpret_cast<f4u*>(dst) = o;
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j) if (gn + j < p.N) dst[j] = v[j] + (Rb ? Rb[(long)gm * p.srm + gn + j] : 0.f);
            }
          }
        }
      }
    }
  }
  if (!SSV_NNBW_PARK || p.scn != 1) {
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gm = m0 + wm * WM * 16 + i * 16 + kq * 4 + r;
      if (gm >= Mt) continue;
      float add = 0.f;
      if (p.bias) add += p.bias[gm];
      if (p.bias_b) add += p.bias_b[(long)b * p.sbb + gm];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int gn = n0 + wn * NT * 16 + t * 16 + nq;
        if (gn >= p.N) continue;
        float v = (F16 ? acc[i][t][r] * us : acc[i][t][r]) + add;
        if (Rb) v += Rb[(long)gm * p.srm + gn];
        Cb[(long)gm * p.scm + gn] = v;
      }
    }
  }
  if constexpr (XR) {
    if (xr_on) {                                            // (workgroup-uniform) row M - 1: the four k-groups' partial sums of a column, then bias and residual
#pragma unroll
      for (int r = 0; r < NX; ++r) { const int e = tid + T * r; if (e < X_SLOTS) xsum[e] = xacc[r]; }
      __syncthreads();
      for (int c = tid; c < WX; c += T) {
        const int gn = n0 + c, gm = p.M - 1;
        if (c < BN && gn < p.N) {
          float v = (xsum[c] + xsum[WX + c]) + (xsum[2 * WX + c] + xsum[3 * WX + c]);
          if (p.bias) v += p.bias[gm];
          if (p.bias_b) v += p.bias_b[(long)b * p.sbb + gm];
          if (Rb) v += Rb[(long)gm * p.srm + gn];
          Cb[(long)gm * p.scm + gn] = v;
        }
      }
    }
  }
}

// rows 0 .. M - 2 in tiles, row M - 1 beside the staging (GemmNNB::xrow_w)
static int launch_nnbw_xrow(const GemmNNB& g, hipStream_t st, int smin, int span) {
  const int mtiles = ssv_cdiv(g.M - 1, 128), ntiles = ssv_cdiv(g.N, 192);
  if (ssv_shape_log_on()) {
    char nm[96], note[96];
    snprintf(nm, sizeof nm, "gemm_nn_bf3w_kernel<1, 2, 6, 2, %d, 1>", g.f16);
    snprintf(note, sizeof note, "B=%d M=%d N=%d K=%d k=1 (last row beside the tiles)", g.B, g.M, g.N, g.Kc);
    ssv_shape_log(nm, dim3(mtiles * ntiles, g.B), dim3(512), 2.0 * g.B * g.M * g.N * g.Kc,
                  4.0 * ((double)g.B * g.Kc * g.N + (double)g.B * g.M * g.N + (double)g.M * g.Kc), note);
  }
  if (g.f16) hipLaunchKernelGGL((gemm_nn_bf3w_kernel<1, 2, 6, 2, 1, 1>), dim3(mtiles * ntiles, g.B), dim3(512), 0, st, g, mtiles, smin, span);
  else hipLaunchKernelGGL((gemm_nn_bf3w_kernel<1, 2, 6, 2, 0, 1>), dim3(mtiles * ntiles, g.B), dim3(512), 0, st, g, mtiles, smin, span);
  return ssv_check_launch("gemm_nn_bf3w (extra row)");
}
template <int KT, int WM, int NT, int NWN>
static int launch_nnbw(const GemmNNB& g, hipStream_t st, int smin, int span) {
  const int mtiles = ssv_cdiv(g.M, 64 * WM), ntiles = ssv_cdiv(g.N, 16 * NT * NWN);
  if (ssv_shape_log_on()) {
    char nm[96], note[96];
    snprintf(nm, sizeof nm, "gemm_nn_bf3w_kernel<%d, %d, %d, %d, %d>", KT, WM, NT, NWN, g.f16);
    snprintf(note, sizeof note, "B=%d M=%d N=%d K=%d k=%d", g.B, g.M, g.N, g.Kc, KT);
    ssv_shape_log(nm, dim3(mtiles * ntiles, g.B), dim3(256 * NWN), 2.0 * g.B * g.M * g.N * g.Kc * KT,
                  4.0 * ((double)g.B * g.Kc * g.N + (double)g.B * g.M * g.N + (double)g.M * g.Kc * KT), note);
  }
  if (g.f16) hipLaunchKernelGGL((gemm_nn_bf3w_kernel<KT, WM, NT, NWN, 1>), dim3(mtiles * ntiles, g.B), dim3(256 * NWN), 0, st, g, mtiles, smin, span);
  else hipLaunchKernelGGL((gemm_nn_bf3w_kernel<KT, WM, NT, NWN, 0>), dim3(mtiles * ntiles, g.B), dim3(256 * NWN), 0, st, g, mtiles, smin, span);
  return ssv_check_launch("gemm_nn_bf3w");
}

template <int KT, int WM, int NT>
static int launch_nnb(const GemmNNB& g, hipStream_t st, int smin, int span) {
  const int mtiles = ssv_cdiv(g.M, 64 * WM), ntiles = ssv_cdiv(g.N, 16 * NT);
  if constexpr (KT == 1) {
    if (g.epi == 1) {
      if (g.f16) hipLaunchKernelGGL((gemm_nn_bf3_kernel<KT, WM, NT, 1, 1>), dim3(mtiles * ntiles, g.B), dim3(256), 0, st, g, mtiles, smin, span);
      else hipLaunchKernelGGL((gemm_nn_bf3_kernel<KT, WM, NT, 1, 0>), dim3(mtiles * ntiles, g.B), dim3(256), 0, st, g, mtiles, smin, span);
      return ssv_check_launch("gemm_nn_bf3_lstm");
    }
  }
  if (ssv_shape_log_on()) {
    char nm[96], note[96];
    snprintf(nm, sizeof nm, "gemm_nn_bf3_kernel<%d, %d, %d, 0, %d, %d>", KT, WM, NT, g.f16, (KT == 3 && span <= SSV_NN_HALO_SMALL) ? SSV_NN_HALO_SMALL : 54);
    snprintf(note, sizeof note, "B=%d M=%d N=%d K=%d k=%d%s", g.B, g.M, g.N, g.Kc, KT, g.colstats ? " +colstats" : "");
    ssv_shape_log(nm, dim3(mtiles * ntiles, g.B), dim3(256), 2.0 * g.B * g.M * g.N * g.Kc * KT,
                  4.0 * ((double)g.B * g.Kc * g.N + (double)g.B * g.M * g.N * (g.R ? 2 : 1) + (double)g.M * g.Kc * KT), note);
  }
  if constexpr (KT == 3) {
    if (span <= SSV_NN_HALO_SMALL) {
      if (g.f16) hipLaunchKernelGGL((gemm_nn_bf3_kernel<KT, WM, NT, 0, 1, SSV_NN_HALO_SMALL>), dim3(mtiles * ntiles, g.B), dim3(256), 0, st, g, mtiles, smin, span);
      else hipLaunchKernelGGL((gemm_nn_bf3_kernel<KT, WM, NT, 0, 0, SSV_NN_HALO_SMALL>), dim3(mtiles * ntiles, g.B), dim3(256), 0, st, g, mtiles, smin, span);
      return ssv_check_launch("gemm_nn_bf3");
    }
  }
  if (g.f16) hipLaunchKernelGGL((gemm_nn_bf3_kernel<KT, WM, NT, 0, 1>), dim3(mtiles * ntiles, g.B), dim3(256), 0, st, g, mtiles, smin, span);
  else hipLaunchKernelGGL((gemm_nn_bf3_kernel<KT, WM, NT, 0, 0>), dim3(mtiles * ntiles, g.B), dim3(256), 0, st, g, mtiles, smin, span);
  return ssv_check_launch("gemm_nn_bf3");
}

template <int KT>
static int pick_nnb(const GemmNNB& g, hipStream_t st, int smin, int span) {
  {
    // measured (round-1 tile sweep; in-step re-check: tools/sweep_force.sh): the wide workgroup wins for kernel-size-1 convolutions over long sequences
    // (SSRN's 513-channel layers: 150 -> 205 TFLOP/s); the k=3 layers are as fast or faster on the 4-wave kernel.
    // 128 x 192 tiles (8 waves) are the faster wide shape (513 -> 512 channels: 104 -> 87 us, 256 -> 512: 60 -> 49 us) except
    // when M leaves a nearly empty last row tile (M = 513), where the 16-wave 128 x 448 tile loses less (round-1 sweep)
    // ... and only when the 128 x 192 tiling still gives every CU a workgroup: a single long utterance (the vocoder's DFT
    // at B = 1: 1026 x 1300 x 1024) is 27-56 wide tiles, a fifth of the chip; the cost model below then picks small tiles.
    if (KT == 1 && !g.epi && !g.perm_h && !g.colstats && g.sxn == 1 && g.scn == 1 && g.N >= 1024 && g.M >= 256 && g.Kc >= 256 &&
        (long)ssv_cdiv(g.M, 128) * ssv_cdiv(g.N, 192) * g.B >= 256)
    {
      // M = 128 j + 1 (SSRN's 513 channels): the last row beside the tiles of the 128 x 192 kernel (120.6 -> see DESIGN 4.6) instead of a fifth row tile
      if (SSV_NNBW_XROW && g.xrow_w && g.M % 128 == 1 && g.Kpad <= 1056 && g.scn == 1 && g.sxn == 1) return launch_nnbw_xrow(g, st, smin, span);
      return (g.M % 128 == 0) ? launch_nnbw<KT, 2, 6, 2>(g, st, smin, span) : launch_nnbw<KT, 2, 7, 4>(g, st, smin, span);
    }
  }
  static const int nts[] = {7, 6, 4, 2};
  // (round 3: 64 x 176 and 64 x 192 tiles -- 22 % less L2 -> CU operand traffic per launch at C = 256, L = 325 -- were forced per
  //  shape inside the step with SSV_NNB_FORCE, tools/sweep_force.sh: every one of ten shapes +0.03..+0.2 ms; instantiations removed)
  int wm = 2, nt = 7;
  bool forced = false;
  // LSTM wavefront with two or more layers in one launch (the steady state of the GE2E embedder: 2 x 3072 x 880 x 1536): the
  // cost model below picks 64 x 112 tiles; measured over the 122 steps of config 5, 128 x 64 tiles are 6 % faster
  // (13.8 -> 13.0 ms; 128 x 96: 13.5, 128 x 112: 14.2, 64 x 96: 14.8) as long as they still give every CU two workgroups.
  // (round 5, layer 0 riding along: three layers = 3 x 24 x 7 = 504 tiles of 128 x 128 are ONE round of two workgroups per CU, with half the weight
  //  bytes per MFMA of the 64-column tile -- the launch is bound by L2 -> CU traffic, 69 GB/s per CU measured on 1008 tiles of 128 x 64)
  if constexpr (KT == 1) {
    if (!forced && g.epi && g.lstm_D > 0 && g.B >= 2) {
      const long t128 = (long)ssv_cdiv(g.M, 128) * ssv_cdiv(g.N, 128) * g.B;
      const char* e = ssv_tuning(SSV_T_LSTM_MERGE);
      if (t128 >= 448 && t128 <= 512 && !(e && atoi(e) == 2)) {
        const int mtiles = ssv_cdiv(g.M, 128), ntiles = ssv_cdiv(g.N, 128);
        if (g.f16) hipLaunchKernelGGL((gemm_nn_bf3_kernel<1, 2, 8, 1, 1>), dim3(mtiles * ntiles, g.B), dim3(256), 0, st, g, mtiles, smin, span);
        else hipLaunchKernelGGL((gemm_nn_bf3_kernel<1, 2, 8, 1, 0>), dim3(mtiles * ntiles, g.B), dim3(256), 0, st, g, mtiles, smin, span);
        return ssv_check_launch("gemm_nn_bf3_lstm");
      }
    }
  }
  if (!forced && g.epi && g.lstm_D > 0 && g.B >= 2 && (long)ssv_cdiv(g.M, 128) * ssv_cdiv(g.N, 64) * g.B >= 512) { wm = 2; nt = 4; forced = true; }
  if (!forced) {
    // tuning aid (tools/sweep_force.sh): SSV_NNB_FORCE="kt:M:N=a,c;kt:M:N=a,c;..." forces the tile of one problem shape
    // inside a whole training step, where a tile's effect on its neighbours shows (isolated timings miss it)
    if (const char* e = ssv_tuning(SSV_T_NNB_FORCE)) {
      char key[48];
      snprintf(key, sizeof key, "%d:%d:%d=", KT, g.M, g.N);
      const char* hit = strstr(e, key);
      int a = 0, c = 0;
      if (hit && (hit == e || hit[-1] == ';') && sscanf(hit + strlen(key), "%d,%d", &a, &c) == 2 && (a == 1 || a == 2))
        for (int x : nts) if (x == c) { wm = a; nt = c; forced = true; }
    }
  }
  if (!forced) {
    double best = 1e30;
    // k=1 products carry a third of the MFMAs per weight byte: 64-row tiles (twice the weight traffic per MAC) lose at every
    // conv shape measured (256 -> 256, L=650: 28 us on 64 x 112 tiles, 20 us on 128 x 64); only the single-"batch" LSTM
    // product, short of workgroups, still wants them
    const int a_min = (KT == 1 && g.B > 1 && !g.epi && g.M > 64) ? 2 : 1;
    for (int a = a_min; a <= 2; ++a)
      for (int c : nts) {
        const long tiles = (long)ssv_cdiv(g.M, 64 * a) * ssv_cdiv(g.N, 16 * c) * g.B;
        const double per_tile = (double)a * c + 0.9 * a + 0.25 * c + 1.0;
        // The most loaded CU runs n workgroups.  For kernel-size-1 products a K chunk carries a third of the MFMAs per
        // staged byte, so co-resident workgroups are needed to cover each other's staging (measured: the LSTM step
        // 3072 x 880 x 768 runs 2x faster on 672 small tiles than on 192 large ones); k=3 chunks are long enough.
        const long n = (tiles + 255) / 256;
        const double overlap = (KT == 1) ? (n >= 3 ? 1.8 : (n == 2 ? 1.5 : 1.0)) : 1.0;
        // a single workgroup per CU (4 waves) cannot cover its own load latencies: measured +10 % on the data-gradient
        // shapes that fit in 256 large tiles (C=256, L=325: 45 -> 41 us on 768 tiles of 64 x 64)
        const double lonely = (tiles <= 256) ? 1.35 : 1.0;
        const double cost = (double)n * per_tile * lonely / overlap;
        if (cost < best) { best = cost; wm = a; nt = c; }
      }
  }
#define SSV_CASE(A_, C_) if (wm == A_ && nt == C_) return launch_nnb<KT, A_, C_>(g, st, smin, span)
  SSV_CASE(2, 7); SSV_CASE(2, 6); SSV_CASE(2, 4); SSV_CASE(2, 2);
  SSV_CASE(1, 7); SSV_CASE(1, 6); SSV_CASE(1, 4); SSV_CASE(1, 2);
#undef SSV_CASE
  return ssv_fail(SSV_UNSUPPORTED, "gemm_nn_bf3: no tile %d,%d", wm, nt);
}

int ssv_launch_gemm_nn_bf3(const GemmNNB& g, hipStream_t st) {
  SSV_CHECK(g.M > 0 && g.N > 0 && g.Kc > 0 && g.B > 0 && g.Kpad % 32 == 0 && g.Kpad >= g.Kc, SSV_BAD_SHAPE, "gemm_nn_bf3: bad problem");
  SSV_CHECK(g.KT == 1 || g.KT == 3, SSV_UNSUPPORTED, "gemm_nn_bf3: kernel_size %d", g.KT);
  SSV_CHECK(g.B <= 65535, SSV_UNSUPPORTED, "gemm_nn_bf3: batch %d exceeds grid.y", g.B);
  SSV_CHECK(g.sxn >= 1 && g.scn >= 1 && (g.scn == 1 || (!g.R && !g.epi)), SSV_BAD_SHAPE, "gemm_nn_bf3: bad column strides");
  SSV_CHECK(!g.epi || (g.KT == 1 && g.perm_h > 0 && g.M == 4 * g.perm_h && g.cstate && (g.B == 1 || g.lstm_D > 0)), SSV_BAD_SHAPE, "gemm_nn_bf3: bad LSTM epilogue request");
  SSV_CHECK(g.lstm_D == 0 || (g.epi == 1 && g.lstm_out && g.lstm_D >= 1 && g.xsplit >= 0 && g.xsplit <= g.Kpad / 32 && g.Kc == g.Kpad && g.sxn == 1), SSV_BAD_SHAPE,
            "gemm_nn_bf3: bad LSTM wavefront request");
  int smin = g.shift[0], smax = g.shift[0];
  for (int j = 1; j < g.KT; ++j) { smin = g.shift[j] < smin ? g.shift[j] : smin; smax = g.shift[j] > smax ? g.shift[j] : smax; }
  const int span = smax - smin;
  SSV_CHECK(span <= 54, SSV_UNSUPPORTED, "gemm_nn_bf3: dilation halo %d exceeds 54", span);
  // the kernels address one batch item's input rows and the weight planes with 32-bit byte offsets (buffer loads)
  SSV_CHECK(((long)g.Kpad * g.sxc + (long)g.Lx * (g.sxn > 0 ? g.sxn : 1)) * 4 < (1L << 31) && (long)g.KT * ((g.M + 15) / 16) * (g.Kpad / 32) * 1024 < (1L << 31),
            SSV_UNSUPPORTED, "gemm_nn_bf3: a batch item's input or the weight planes span 2 GiB or more");
  SSV_CHECK(!g.f16 || (g.a_inv && ((g.x_amax && g.x_namax > 0) || (g.epi && g.x_namax == 0))), SSV_BAD_SHAPE, "gemm_nn_bf3: split-fp16 needs operand scales");
  SSV_CHECK(!g.colstats || (g.M % 64 == 0 && g.scn == 1 && !g.epi && !g.perm_h && !g.R), SSV_BAD_SHAPE, "gemm_nn_bf3: column statistics need M %% 64 == 0 and a plain epilogue");
  return g.KT == 3 ? pick_nnb<3>(g, st, smin, span) : pick_nnb<1>(g, st, smin, span);
}

// ---- NT (weight gradient) -------------------------------------------------------------------------------------------------
//   C(z,m,c,j) = sum_{b = z, z+bstep, ..} sum_t A(b,m,t) * X(b,c,t+shift[j]);  rows contiguous in t for both operands.
// The reduction runs over time, so a tap's dilation shift is a misaligned shift along k: every tap needs its own staged
// copy of the input rows.  The kernel therefore steps over (batch item, 64-step chunk, tap), tap fastest:
//   * the waves split M, so a wave's A rows (dL/dH) are private: its fragments go global -> registers (two 16-byte buffer
//     loads per 8 time steps = the window's two fragment tuples), are split there in place once per chunk and reused by the
//     KT taps -- A never touches LDS (see AH / AL in the kernel);
//   * per step only ONE tap's input tile (16*NTC channels x 64 steps) is split and staged, into one of two LDS buffers:
//     the step's MFMAs read buffer s while the next step's tile is written to buffer s^1 -- one barrier per step, and
//     16 staging registers instead of 48 (staging all taps at once put the 128 x 64 x 3 tile at 256 VGPRs with spills,
//     and every scratch reload waits for vmcnt(0), i.e. for the whole prefetch in flight).
// Loads are issued raw, one step (input) or one chunk (A) ahead, with no branch in the prefetch (see load8c / split_edge).
// Tuning builds only (-DSSV_NT_STAMP): wave 0 of workgroup 0 records s_memtime at six points of every step from step 24 on
// (8 steps); read back with ssv_debug_nt_stamps().  Results are unaffected.
#ifdef SSV_NT_STAMP
__device__ unsigned long long ssv_nt_stamps[64];
__device__ unsigned long long ssv_nt_wg[4096 * 4];      // per workgroup (first 4096): s_memrealtime at entry / exit, shader clock at entry / exit
extern "C" int ssv_debug_nt_wg(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ssv_nt_wg), sizeof(ssv_nt_wg)); }
#define NT_WG(k, v) do { const unsigned w_ = blockIdx.z * gridDim.x + blockIdx.x; if (threadIdx.x == 0 && w_ < 4096u) ssv_nt_wg[w_ * 4 + (k)] = (v); } while (0)
#define NT_STAMP(k) do { if (stamp_on && (unsigned)(stamp_s - 24) < 8u) ssv_nt_stamps[(stamp_s - 24) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
extern "C" int ssv_debug_nt_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ssv_nt_stamps), sizeof(ssv_nt_stamps)); }
#else
#define NT_STAMP(k) do {} while (0)
#define NT_WG(k, v) do {} while (0)
#endif
constexpr int NT_FD = 1;      // LDS fragment groups read ahead of the MFMAs (see the step loop)
#ifndef SSV_NT_RING
#define SSV_NT_RING 1         // 0 (tuning builds): the k = 3 weight gradient on gemm_nt_bf3_kernel, as before round 4's ring kernel
#endif
// XR (k = 1): M = 128 j + 1 rows (the 513-channel layers): the tiles cover rows 0 .. M - 2 and row M - 1 of the product is added by the workgroups of
// row tile 0 as fp32 dot products of dH(M - 1, t) with the raw input values every staging thread holds before it splits them -- a fifth row tile
// of MFMAs for ONE row otherwise (30 tiles of 128 x 96 for 24).  It pays only together with RANGE slabs (p.bstep == 0: slab z reduces over the
// z-th of Z equal ranges of the launch's B x tchunks chunks, not over whole batch items): with whole items 24 x 16 workgroups do the same two
// items each as 30 x 16 did, and the launch lasts as long as its slowest workgroup.
#ifndef SSV_NT_XROW
#define SSV_NT_XROW 1        // (tuning builds: 0 = a row tile of its own for the last row and whole-item slabs, as before)
#endif
template <int KT, int WM, int NTC, int F16, int XR = 0>
__global__ __launch_bounds__(256, 2) void gemm_nt_bf3_kernel(const GemmNT p, const int mtiles) {
  constexpr int KB = 64, KG = KB / 8, KS = KB / 32;         // time steps per chunk, k-groups, MFMA k-steps
  constexpr int NCH = 16 * NTC;
  constexpr int X_SLOTS = KG * NCH;                         // 16-byte slots of one tap's tile (multiple of 256)
  constexpr int NX = X_SLOTS / 256;
  static_assert(X_SLOTS % 256 == 0, "tile slots must be a multiple of the workgroup size");
  // [buffer][hi plane | lo plane], slot = kg*NCH + (channel ^ kg).  The staging threads take kg fastest (8 lanes = 256
  // contiguous bytes of one channel row in global memory), so without the XOR the 8 lanes of a ds_write_b128 group would
  // write slots 1 KB apart -- one bank set, an 8-way conflict that cost more than the step's MFMAs.  With it they land on
  // 8 distinct 16-byte bank groups, and the fragment reads (16 consecutive channels per quarter wave) stay conflict-free.
  __shared__ uint4 lds[2][2 * X_SLOTS];
  __shared__ float amax_sm[8];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  NT_WG(0, __builtin_amdgcn_s_memrealtime()); NT_WG(2, __builtin_readcyclecounter());
  const unsigned wg = ssv_xcd_order(blockIdx.x + gridDim.x * blockIdx.z, gridDim.x * gridDim.z);   // a slab's tiles share an XCD
  const int bxx = (int)(wg % gridDim.x);
  int z = (int)(wg / gridDim.x);
  // operands: the launch's own, or those of job z / Z (several equal-shaped layers in one launch, see GemmNT)
  const float* __restrict__ Ap = p.A;
  const float* __restrict__ Xp = p.X;
  float* __restrict__ Cp = p.C;
  int shj[3] = {p.shift[0], p.shift[1], p.shift[2]};
  const float* __restrict__ a_amax = p.a_amax;
  const float* __restrict__ x_amax = p.x_amax;
  int a_namax = p.a_namax, x_namax = p.x_namax;
  if (p.jobs) {
    const int job = z / p.Z;
    z -= job * p.Z;
    // member by member: a by-value copy of the entry reaches the pointers through integer loads, and hipcc then knows nothing of
    // their address space -- every operand load of this kernel was a flat_load (see ssv_global)
    const ssv_wgrad_job* __restrict__ jb = p.jobs + job;
    Ap = ssv_global(jb->dy); Xp = ssv_global(jb->x); Cp = p.C + (long)job * p.Z * p.scz;
    shj[0] = jb->shift[0]; shj[1] = jb->shift[1]; shj[2] = jb->shift[2];
    a_amax = ssv_global(jb->dy_amax); x_amax = ssv_global(jb->x_amax); a_namax = jb->dy_namax; x_namax = jb->x_namax;
  }
  // split-fp16: one power-of-two scale per operand tensor (the reduction runs over the batch), undone in the epilogue
  float as = 1.f, xs = 1.f, us = 1.f;
  auto scales = [&]() {
    if constexpr (F16) {
      float sa, sx, ia, ix;
      ssv_pow2_scale(ssv_list_max<4>(a_amax, a_namax, amax_sm), sa, ia);       // long lists (B * tiles entries): shared among the waves
      ssv_pow2_scale(ssv_list_max<4>(x_amax, x_namax, amax_sm + 4), sx, ix);
      as = ssv_uniform(sa); xs = ssv_uniform(sx);
      us = ssv_uniform(ia * ix);
    }
  };
  const int mt = bxx % mtiles, ct = bxx / mtiles;
  const int m0 = mt * 64 * WM, c0 = ct * NCH;
  const int tchunks = (p.La + KB - 1) / KB;
  const int kq = lane >> 4, nq = lane & 15;

  f32x4 acc[WM][KT][NTC];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < KT; ++j)
#pragma unroll
      for (int q = 0; q < NTC; ++q) acc[i][j][q] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float rx[NX][8];                     // raw loads in flight: the tile of step s + 2
  int mx[NX];                          // edge windows only: validity bits (low 8) | offset clamp distance << 8
  static_assert(!XR || (KT == 1 && (256 % KG) == 0), "extra row: k = 1; a thread's slots share their k-group");
  const bool xr_on = XR && mt == 0;
  float xacc[XR ? NX : 1];
  uint4 xra[XR ? 2 : 1];               // dH(M - 1, t0 + 8 kg .. + 7) of the tile in flight (raw; what lies past the row meets masked input)
#pragma unroll
  for (int r = 0; r < (XR ? NX : 1); ++r) xacc[r] = 0.f;
  // dH fragments: two sets of (hi, lo) register tuples; set n & 1 is chunk n's.  The next chunk's windows are LOADED into the other
  // set (a window's two 16-byte loads = its two tuples) and split there in place, dword by dword (split8p's order of the time steps):
  // no staging registers.  (The split of the next set woven behind the MFMAs of a chunk's last tap -- one half-rate split instruction
  // hides behind a 16x16x32 MFMA of the same wave, tools/probe/mfma_valu.hip -- was built on this layout and measured: no change
  // in-step, 268.5 vs 268.3 us; with two waves on a SIMD the other wave's MFMAs already fill those slots.)  dH needs no masks and no clamps:
  //   * it is read by BUFFER loads whose range is the tensor (per-dword range check: what lies past the end reads 0; its offsets are
  //     never negative), so a window may run past its row -- into the next row's values, or into zeros;
  //   * time steps at or past the row length meet an input tile that is zero there (x_edge / the mask of the input include t < La).
  //   (A NaN or Inf at the head of the NEXT row would so reach this row's sums as NaN; with masks it stayed in its own row.)
  uint4 AH[2][WM][KS], AL[2][WM][KS];

  // A window = 8 consecutive time steps of one row, at any alignment.  Element offsets are 32-bit (the launcher checks
  // that both tensors span < 2^30 elements): a per-thread row offset, fixed for the whole kernel, plus a wave-uniform
  // (batch item, chunk, tap) offset.  Two kinds of chunk / step, told apart by a wave-uniform test:
  //   interior -- every window of the tile lies inside its row: plain loads, plain split; nothing else on the VALU;
  //   edge     -- a window may start before 0 or run past the row (first/last chunk, ragged tile): it is still loaded RAW
  //     by two 16-byte loads at its true offset (an edge window simply runs into the neighbouring row) and the validity
  //     bits are applied when the values are split, a step later.  Masking at load time makes hipcc branch around each
  //     load and wait for it.  Only a window that would leave the TENSOR (head of its first row, tail of its last) has
  //     its offset clamped; the clamp distance travels with the mask and the split shifts the values back into place.
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Ap), 0,
      (int)(((long)(p.B - 1) * p.sab + (long)(p.M - 1) * p.sam + p.La) * 4), 0x00020000);
  const int x_span = (int)((long)(p.B - 1) * p.sxb + (long)(p.Nc - 1) * p.sxc + p.Lx) - 8;
  int arow[WM], xrow[NX];
#pragma unroll
  for (int i = 0; i < WM; ++i) arow[i] = (min(m0 + wave * WM * 16 + i * 16 + nq, p.M - 1) * (int)p.sam + 8 * kq) * 4;     // BYTE offset of the buffer loads
  const int xra_off = ((p.M - 1) * (int)p.sam + 8 * (tid % KG)) * 4;
#pragma unroll
  for (int r = 0; r < NX; ++r) {
    const int f = tid + 256 * r;
    xrow[r] = min(c0 + f / KG, p.Nc - 1) * (int)p.sxc + 8 * (f % KG);
  }
  const bool rows_in_c = c0 + NCH <= p.Nc;

  auto load8 = [&](const float* __restrict__ base, int off, float (&v)[8]) {
    // uniform base + zero-extended 32-bit BYTE offset (operands span < 2^30 elements): the saddr form of global_load
    // address space spelled out: with operand pointers that may come from a job table hipcc emitted flat_load here (see ssv_global)
    typedef __attribute__((address_space(1))) const char gchar;
    typedef __attribute__((address_space(1))) const f4u gf4u;
    gchar* q = (gchar*)reinterpret_cast<const char*>(base) + ((unsigned)off << 2);
    const f4u a = *(gf4u*)q;
    const f4u c = *(gf4u*)(q + 16);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
  };
  // Interior and edge windows are loaded by the SAME instructions (the offset clamped by one v_med3 either way): with the loads
  // under an interior / edge branch, hipcc's waitcnt bookkeeping differed between the two paths and it drained vmcnt to 0 in front
  // of the edge path's loads.  Only the mask (VALU) is edge-only.
  auto load8c = [&](const float* __restrict__ base, int off, int span, float (&v)[8]) -> int {
    const int oc = min(max(off, 0), span);
    load8(base, oc, v);
    return off - oc;
  };
  auto edge_meta = [&](int d, int t, int len, bool row_ok, int over2 = 0) -> int {     // over2: elements cut off the window's end by a second limit
    const int sl = min(max(-t, 0), 8), sh = max(min(max(t + 8 - len, 0), 8), min(max(over2, 0), 8));
    const int m = row_ok ? (int)((0xFFu << sl) & (0xFFu >> sh) & 0xFFu) : 0;
    return m | (d << 8);
  };
  auto edge_vals = [&](const float (&raw)[8], int meta, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = raw[i];
    const int d = meta >> 8;
    // wave-uniform test: a scalar branch hipcc cannot turn into straight-line selects (128 VALU instructions per window)
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(d != 0) != 0ull, 0)) {   // v[i] must be the element at offset oc + (i + d)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float r = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) r = (i + d == k) ? raw[k] : r;
        v[i] = r;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = ((meta >> i) & 1) ? v[i] : 0.f;
  };
  auto split_edge = [&](const float (&raw)[8], int meta, float sc, uint4& h, uint4& l) {
    float v[8];
    edge_vals(raw, meta, v);
    split8p<F16>(v, sc, h, l);
  };

  // chunk cursors (wave-uniform): batch item and first time step of chunks n .. n+3
  // whole-item slabs (bstep > 0): slab z reduces over items z, z + bstep, ...;  range slabs (bstep == 0): over the z-th of Z equal ranges of the
  // item-major sequence of all B x tchunks chunks (a finer cut: the work per workgroup need not be a whole number of items)
  const int bstep = p.bstep > 0 ? p.bstep : 1;
  int total, cb[4], ct0[4];
  if (p.bstep > 0) {
    total = ((p.B - z + p.bstep - 1) / p.bstep) * tchunks;                  // chunks this workgroup reduces over
    cb[0] = z; ct0[0] = 0;
  } else {
    const int all = p.B * tchunks, per = (all + p.Z - 1) / p.Z, start = z * per;
    total = max(min(per, all - start), 0);
    cb[0] = start / tchunks; ct0[0] = (start % tchunks) * KB;
  }
  auto next_chunk = [&](int b, int t0, int& nb_, int& nt0) __attribute__((always_inline)) {
    nt0 = t0 + KB; nb_ = b;
    if (nt0 >= tchunks * KB) { nt0 = 0; nb_ = b + bstep; }
  };
#pragma unroll
  for (int k = 1; k < 4; ++k) next_chunk(cb[k - 1], ct0[k - 1], cb[k], ct0[k]);
  auto x_edge = [&](int t0, int j) __attribute__((always_inline)) -> bool {
    return !(rows_in_c && t0 + shj[j] >= 0 && t0 + KB + shj[j] <= p.Lx && t0 + KB <= p.La);       // (the last: dH is not masked, see AH / AL)
  };

  auto loadA = [&](auto set, int b, int t0) __attribute__((always_inline)) {                               // -> AH / AL[set], raw
    constexpr int SET = decltype(set)::value;
    const unsigned so = (unsigned)(b * (int)p.sab + t0) * 4u;   // uniform
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) {
        AH[SET][i][s2] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, arow[i] + s2 * 128, (int)so, 0));
        AL[SET][i][s2] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, arow[i] + s2 * 128 + 16, (int)so, 0));
      }
  };
  auto raw8 = [&](const uint4& t0_, const uint4& t1_, float (&v)[8]) __attribute__((always_inline)) {
    v[0] = __builtin_bit_cast(float, t0_.x); v[1] = __builtin_bit_cast(float, t0_.y); v[2] = __builtin_bit_cast(float, t0_.z); v[3] = __builtin_bit_cast(float, t0_.w);
    v[4] = __builtin_bit_cast(float, t1_.x); v[5] = __builtin_bit_cast(float, t1_.y); v[6] = __builtin_bit_cast(float, t1_.z); v[7] = __builtin_bit_cast(float, t1_.w);
  };
  auto splitA = [&](auto set) __attribute__((always_inline)) {                                             // AH / AL[set]: raw -> (hi, lo), in place
    constexpr int SET = decltype(set)::value;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) {
        float v[8];
        raw8(AH[SET][i][s2], AL[SET][i][s2], v);
        split8p<F16>(v, as, AH[SET][i][s2], AL[SET][i][s2]);
      }
  };
  auto loadX = [&](int b, int t0, int j) __attribute__((always_inline)) {                                  // -> rx (/ mx)
    const int base = b * (int)p.sxb + t0 + shj[j];
    int dd[NX];
#pragma unroll
    for (int r = 0; r < NX; ++r) dd[r] = load8c(Xp, base + xrow[r], x_span, rx[r]);
    if constexpr (XR) {                // every workgroup issues them (a branch around loads costs hipcc's waitcnt bookkeeping more than two L2 hits)
      const unsigned so = (unsigned)(b * (int)p.sab + t0) * 4u;
      xra[0] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, xra_off, (int)so, 0));
      xra[1] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, xra_off + 16, (int)so, 0));
    }
    if (x_edge(t0, j)) {
#pragma unroll
      for (int r = 0; r < NX; ++r) {
        const int f = tid + 256 * r;
        mx[r] = edge_meta(dd[r], t0 + shj[j] + 8 * (f % KG), p.Lx, c0 + f / KG < p.Nc, t0 + 8 * (f % KG) + 8 - p.La);
      }
    }
  };
  auto commitX = [&](auto buf, int t0, int j) __attribute__((always_inline)) {                             // rx -> LDS buffer `buf`
    constexpr int S = decltype(buf)::value;
    uint4* Xh = lds[S];
    uint4* Xl = lds[S] + X_SLOTS;
    const bool edge = x_edge(t0, j);
#pragma unroll
    for (int r = 0; r < NX; ++r) {
      const int f = tid + 256 * r;
      const int kg = f % KG, c = f / KG;
      uint4 h, l;
      if (!edge) split8p<F16>(rx[r], xs, h, l);
      else split_edge(rx[r], mx[r], xs, h, l);
      Xh[kg * NCH + (c ^ kg)] = h; Xl[kg * NCH + (c ^ kg)] = l;          // XOR swizzle, see the slot comment above
      if constexpr (XR) {
        if (xr_on) {                   // row M - 1 of the product: dH(M - 1, t) x(c, t) over this slot's 8 time steps
          float av[8], xv[8];
          raw8(xra[0], xra[1], av);
          if (!edge) {
#pragma unroll
            for (int i = 0; i < 8; ++i) xv[i] = rx[r][i];
          } else edge_vals(rx[r], mx[r], xv);
#pragma unroll
          for (int i = 0; i < 8; ++i) xacc[r] = __builtin_fmaf(av[i], xv[i], xacc[r]);
        }
      }
    }
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  const int steps = total * KT;

  // One chunk = KT steps (tap fastest); step s = n*KT + j uses LDS buffer s & 1.  In step s:
  //   MFMAs of step s on buffer s&1  |  tile s+1 (loaded a step ago) is split into buffer (s+1)&1  |  the loads of tile s+2
  //   are issued into the registers just freed  ->  a tile's loads have a barrier and a step's MFMAs (~2,000 cycles) to land.
  // (Round 2 kept two register sets and issued tile s+3: 16 more VGPRs, and hipcc's vmcnt(0) in front of every batch -- see
  // STEADY below -- made it wait for tile s+2 anyway.)
  // STEADY = every commit and load of the chunk is known to be due (no "is there a step s + 3" tests).  Not for the branches
  // saved: hipcc's s_waitcnt placement merges its bookkeeping over all paths, and the path that skips a commit leaves that
  // tile's loads "possibly in flight" -- it then drained vmcnt to 0 in front of EVERY batch of loads (their address registers
  // reuse the tile's), so each batch waited for the previous one to land: four exposed L2 round trips per chunk, 7,000 of a
  // chunk's 10,000 cycles by s_memtime stamps.  The last chunks run the tested form.
  auto chunk = [&](auto steady, auto par, int n) __attribute__((always_inline)) {
    constexpr bool STEADY = decltype(steady)::value;
    constexpr int PAR = decltype(par)::value;                               // parity of this chunk's first step
    const bool more = STEADY || n + 1 < total;
#ifdef SSV_NT_STAMP
    const bool stamp_on = blockIdx.x == 0 && blockIdx.z == 0 && tid == 0;
#endif
    using CUR = std::integral_constant<int, PAR>;                           // this chunk's dH set (n & 1 = the parity of its first step: KT is odd)
    using NXT = std::integral_constant<int, PAR ^ 1>;
    if (more) loadA(NXT{}, cb[1], ct0[1]);                                  // lands during this chunk's first steps
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      const int q_ = (PAR + j) & 1;
#ifdef SSV_NT_STAMP
      const int stamp_s = n * KT + j;
#endif
      NT_STAMP(0);
      const uint4* Xh = lds[q_];
      const uint4* Xl = lds[q_] + X_SLOTS;
      // The input fragments of group g + FD are read from LDS BEFORE the MFMAs of group g are issued (a group = one 16-channel
      // block of one k-step: 2 reads, 3 WM MFMAs).  Left to itself hipcc issues a group's reads right in front of its MFMAs and
      // parks the wave on lgkmcnt for the LDS latency eight times per step -- a third of the wave's cycles by the SQ counters.
      constexpr int G = KS * NTC;
      uint4 fb[NT_FD + 1][2];
      auto frag = [&](int g, uint4 (&f)[2]) __attribute__((always_inline)) {
        const int kg = (g / NTC) * 4 + kq;
        const int xs_ = kg * NCH + (((g % NTC) * 16 + nq) ^ kg);
        f[0] = Xh[xs_]; f[1] = Xl[xs_];
      };
#pragma unroll
      for (int g = 0; g < NT_FD; ++g) frag(g, fb[g]);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (g + NT_FD < G) frag(g + NT_FD, fb[(g + NT_FD) % (NT_FD + 1)]);
        __builtin_amdgcn_sched_barrier(0);                                   // or the scheduler sinks the reads back to their use
        const int s2 = g / NTC, q = g % NTC;
        if (s2 > 0 && q == 0 && ct0[0] + 32 * s2 >= p.La) break;           // ragged last chunk: the k-steps from here on lie past the row end (the input tile is zero there)
        const uint4 bh = fb[g % (NT_FD + 1)][0];
        const uint4 bl = fb[g % (NT_FD + 1)][1];
#pragma unroll
        for (int i = 0; i < WM; ++i) {
          const uint4 a_h = AH[CUR::value][i][s2];
          const uint4 a_l = AL[CUR::value][i][s2];
          acc[i][j][q] = mma16<F16>(a_l, bh, acc[i][j][q]);
          acc[i][j][q] = mma16<F16>(a_h, bl, acc[i][j][q]);
          acc[i][j][q] = mma16<F16>(a_h, bh, acc[i][j][q]);
        }
      }
      NT_STAMP(1);
      const int c1 = (j + 1) / KT, j1 = (j + 1) % KT;                       // step s+1: chunk n + c1, tap j1
      const int c2 = (j + 2) / KT, j2 = (j + 2) % KT;                       // step s+2
      const int s = n * KT + j;
#ifdef SSV_NT_STAMP
      asm volatile("" : "+v"(rx[NX - 1][7]));        // the tile's last load has landed
      NT_STAMP(6);
#endif
      if (STEADY || s + 1 < steps) {
        if (((PAR + j) & 1) == 0) commitX(P1{}, ct0[c1], j1); else commitX(P0{}, ct0[c1], j1);
      }
      NT_STAMP(2);
      if (STEADY || s + 2 < steps) loadX(cb[c2], ct0[c2], j2);
      NT_STAMP(3);
      if (j == KT - 1 && more) splitA(NXT{});
      NT_STAMP(4);
      __syncthreads();
      NT_STAMP(5);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { cb[k] = cb[k + 1]; ct0[k] = ct0[k + 1]; }
    next_chunk(cb[2], ct0[2], cb[3], ct0[3]);
  };

  if (total > 0) {
    // prologue: A of chunk 0 split, tile 0 staged, tile 1 in flight
    loadA(P0{}, cb[0], ct0[0]);
    loadX(cb[0], ct0[0], 0);
    scales();
    splitA(P0{});
    commitX(P0{}, ct0[0], 0);
    if (steps > 1) loadX(cb[1 / KT], ct0[1 / KT], 1 % KT);
    __syncthreads();
    using ST = std::integral_constant<bool, true>;
    using TL = std::integral_constant<bool, false>;
    const int nfull = (steps - 2) / KT;                 // chunks n < nfull have all their steps' s + 2 < steps
    static_assert((KT & 1) == 1, "chunk parities alternate");
    int n = 0;
    for (; n + 1 < nfull; n += 2) {
      chunk(ST{}, P0{}, n);
      chunk(ST{}, P1{}, n + 1);
    }
    for (; n < total; n += 2) {
      chunk(TL{}, P0{}, n);
      if (n + 1 < total) chunk(TL{}, P1{}, n + 1);
    }
  }

  NT_WG(3, __builtin_readcyclecounter());            // end of the chunk loop
  float* __restrict__ Cz = Cp + (long)z * p.scz;
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gm = m0 + wave * WM * 16 + i * 16 + kq * 4 + r;
      if (gm >= (XR ? p.M - 1 : p.M)) continue;
#pragma unroll
      for (int j = 0; j < KT; ++j)
#pragma unroll
        for (int q = 0; q < NTC; ++q) {
          const int gc = c0 + q * 16 + nq;
          if (gc < p.Nc) Cz[(long)gm * p.scm + (long)gc * p.scc + (long)j * p.scj] = F16 ? acc[i][j][q][r] * us : acc[i][j][q][r];
        }
    }
  if constexpr (XR) {
    if (xr_on) {                       // row M - 1: a channel's 8 k-groups are 8 neighbouring lanes (slot f = tid + 256 r: kg = f % 8, channel = f / 8)
#pragma unroll
      for (int r = 0; r < NX; ++r) {
        float v = xacc[r];
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
        const int f = tid + 256 * r, gc = c0 + f / KG;
        if ((f % KG) == 0 && gc < p.Nc) Cz[(long)(p.M - 1) * p.scm + (long)gc * p.scc] = v;
      }
    }
  }
#ifdef SSV_NT_STAMP
  __builtin_amdgcn_s_waitcnt(0);
#endif
  NT_WG(1, __builtin_amdgcn_s_memrealtime());
}

// ---- NT, k = 3, input staged ONCE: a ring of time-major rows read back by transposed LDS reads (round 4) --------------------------
// The kernel above re-stages the input tile for every tap (the reduction runs over time, so a tap's dilation shift is a shift along
// k): three splits, three sets of LDS stores and three barriers per 64-step chunk for the same 64 x 64 values.  Here the tile lives in
// LDS TIME-major -- row = time step, 64 channels x fp16 = 128 bytes per row, a hi plane and a lo plane -- and the MFMA B operand
// (8 consecutive time steps of one channel per lane) comes out of `ds_read_b64_tr_b16`, the gfx950 transposed read: a 16-lane group
// names four ROWS (any four: lanes 4q .. 4q+3 carry row q's address) and lane i receives column i of them.  A tap's shift is then
// only a different row address, so
//   * each input value is loaded, split and stored ONCE: the rows form a ring of four 64-row blocks (block k = the 64 time steps of
//     chunk k, ring slot k & 3); chunk n multiplies rows 64n + shift_j + [0, 64), i.e. blocks n-1 .. n+1, while block n+2 is being
//     written -- one unit of 4 channels x 4 time steps per thread and chunk (four 16-byte loads along time, eight 8-byte LDS stores)
//     against three windows of 16 before, and ONE barrier per chunk;
//   * the chunks of one slab form one virtual time line: the items follow each other in `tchunks` chunks each, with tchunks * 64 >=
//     L + the largest |shift|, so the rows a tap reaches beyond an item's ends are rows of that item's own padding region or of the
//     neighbour's -- staged as zeros (t >= Lx is masked) or met by dH values that are zero (t >= La is masked in the dH split of an
//     item's last chunk; what such a zero multiplies is input data of the tensor itself, i.e. finite).  No per-tap edge cases;
//   * rows 256 .. 287 mirror rows 0 .. 31 (written with block 4i's first rows), so a k-step (32 rows) that starts in the last rows of
//     the ring runs on linearly: every read address is lane constant + a per-(chunk, tap, k-step) SCALAR.
// The dH side: a lane's fragment of a 32-step k-step is two 16-byte tuples, steps 4 kq .. 4 kq + 3 and 16 + 4 kq .. 19 + 4 kq (so that one load
// instruction touches 64 CONTIGUOUS bytes of each of its 16 rows, not four 16-byte pieces spread over 128: -1.3 % in-step), split in place pair by
// pair: dword q of the fragment = steps (4 kq + q, 16 + 4 kq + q).  Any order is right as long as the input side uses it: read 0 of a
// fragment names rows (0, 16, 1, 17) + 4 kq of the k-step, read 1 rows (2, 18, 3, 19) + 4 kq.
// Bank conflicts (cdna_hip_programming.md section 2: 64 dword banks per 32-lane half for the transposed read): a half reads 8 row
// pieces of 32 bytes -- rows u + {0,1,4,5,16,17,20,21} (+2 for the second read) of one 16-channel block.  With 128-byte rows these would
// share two 32-byte bank groups; the 32-byte piece `cb` of row r is therefore stored at piece cb ^ (bit 2 of r | bit 4 of r << 1): the 8 pieces
// of a half land on 8 distinct bank groups for every u (checked exhaustively; no linear row stride does that for the rows a split-in-place
// fragment names, tools/tr_layout_search.py).
typedef short ssv_s4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) ssv_s4 ssv_lds_s4;
typedef unsigned ssv_u2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) ssv_u2 ssv_lds_u2;
// LDS accesses by 32-bit LDS byte address (the ring kernel computes addresses, not indices)
__device__ __forceinline__ void ssv_lds_store2(unsigned addr, unsigned a, unsigned b) {
#if defined(__HIP_DEVICE_COMPILE__)
  *(ssv_lds_u2*)(size_t)addr = (ssv_u2){a, b};
#else
  (void)addr; (void)a; (void)b;
#endif
}
__device__ __forceinline__ uint2 ssv_lds_read_tr16(unsigned addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((ssv_lds_s4*)(size_t)addr));
#else
  (void)addr; return make_uint2(0u, 0u);
#endif
}
template <int WM, int F16>
__global__ __launch_bounds__(256, 2) void gemm_nt3r_kernel(const GemmNT p, const int mtiles, const int tchunks, const int max_shift) {
  constexpr int KT = 3, NTC = 4, NCH = 64, KB = 64, KS = 2;
  constexpr int R_FD = 1;                                   // fragment groups read ahead of the MFMAs (2: no change, 8 more registers)
  constexpr unsigned ROWB = 128, PLANE = 288 * ROWB;            // bytes per row / per plane (256 ring rows + 32 mirror rows)
  __shared__ __attribute__((aligned(16))) unsigned char ring[2 * PLANE];
  __shared__ float amax_sm[8];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  NT_WG(0, __builtin_amdgcn_s_memrealtime()); NT_WG(2, __builtin_readcyclecounter());
  const unsigned wg = ssv_xcd_order(blockIdx.x + gridDim.x * blockIdx.z, gridDim.x * gridDim.z);
  const int bxx = (int)(wg % gridDim.x);
  int z = (int)(wg / gridDim.x);
  const float* __restrict__ Ap = p.A;
  const float* __restrict__ Xp = p.X;
  float* __restrict__ Cp = p.C;
  int shj[3] = {p.shift[0], p.shift[1], p.shift[2]};
  const float* __restrict__ a_amax = p.a_amax;
  const float* __restrict__ x_amax = p.x_amax;
  int a_namax = p.a_namax, x_namax = p.x_namax;
  if (p.jobs) {
    const int job = z / p.Z;
    z -= job * p.Z;
    const ssv_wgrad_job* __restrict__ jb = p.jobs + job;
    Ap = ssv_global(jb->dy); Xp = ssv_global(jb->x); Cp = p.C + (long)job * p.Z * p.scz;
    shj[0] = jb->shift[0]; shj[1] = jb->shift[1]; shj[2] = jb->shift[2];
    a_amax = ssv_global(jb->dy_amax); x_amax = ssv_global(jb->x_amax); a_namax = jb->dy_namax; x_namax = jb->x_namax;
  }
  float as = 1.f, xs = 1.f, us = 1.f;
  if constexpr (F16) {
    float sa, sx, ia, ix;
    ssv_pow2_scale(ssv_list_max<4>(a_amax, a_namax, amax_sm), sa, ia);
    ssv_pow2_scale(ssv_list_max<4>(x_amax, x_namax, amax_sm + 4), sx, ix);
    as = ssv_uniform(sa); xs = ssv_uniform(sx);
    us = ssv_uniform(ia * ix);
  }
  const int mt = bxx % mtiles, ct = bxx / mtiles;
  const int m0 = mt * 64 * WM, c0 = ct * NCH;
  const int kq = lane >> 4, nq = lane & 15;

  f32x4 acc[WM][KT][NTC];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < KT; ++j)
#pragma unroll
      for (int q = 0; q < NTC; ++q) acc[i][j][q] = (f32x4){0.f, 0.f, 0.f, 0.f};

  uint4 AH[2][WM][KS], AL[2][WM][KS];                       // dH fragments, as in gemm_nt_bf3_kernel: two sets, loaded raw, split in place
  uint4 rx[4];                                              // the thread's unit of the block in flight: channel ci, 4 time steps

  // both operands are read by buffer loads whose range is the tensor: the chunks beyond the slab's last (the look-ahead never tests
  // for them) and what lies past a tensor's end read as zeros or as other entries of the same tensor, never outside it.  The whole
  // offset travels in the VGPR: the hardware's range check covers the vector offset only, a scalar offset is added after it
  // (seen as NaN gradients: look-ahead blocks read with the batch item in the scalar offset came from beyond the tensor)
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Ap), 0,
      (int)(((long)(p.B - 1) * p.sab + (long)(p.M - 1) * p.sam + p.La) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Xp), 0,
      (int)(((long)(p.B - 1) * p.sxb + (long)(p.Nc - 1) * p.sxc + p.Lx) * 4), 0x00020000);
  // (rows / channels past a ragged tile's end are not clamped: they read other rows of the tensor or zeros, and feed accumulators nobody stores)
  const unsigned arow = (unsigned)(((m0 + wave * WM * 16 + nq) * (int)p.sam + 4 * kq) * 4);
  const unsigned a16 = 16u * (unsigned)p.sam * 4u;            // one 16-row block further (scalar)
  // staging: wave w owns the 16-channel piece w of every row; lane = cq + 4 tq: channels 16 w + 4 cq .. + 3, rows 4 tq .. 4 tq + 3 of the block
  const int cq = lane >> 4, tq = lane & 15;               // 4 consecutive lanes = 64 contiguous bytes of one channel row
  const unsigned xrow = (unsigned)(((c0 + 16 * wave + 4 * cq) * (int)p.sxc + 4 * tq) * 4);
  const unsigned x1 = (unsigned)p.sxc * 4u;                   // one channel further (scalar)
  const bool rows_in_c = c0 + NCH <= p.Nc;
  const unsigned ringb = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ring;
  const unsigned waddr = ringb + (unsigned)(4 * tq) * ROWB + (unsigned)((wave ^ ((tq & 1) | (((tq >> 2) & 1) << 1))) << 5) + (unsigned)cq * 8u;
  // fragment reads: lane (kq, 4 q + pp) of read r names row 8 kq + rsel[r][q]; the piece swizzle of that row depends on the tap's shift
  const int q4 = (lane >> 2) & 3, pp = lane & 3;
  int lrow[2];
  lrow[0] = 4 * kq + (q4 >> 1) + 16 * (q4 & 1);              // q = 0..3 -> rows 0, 16, 1, 17 of the k-step (+ 4 kq)
  lrow[1] = lrow[0] + 2;
  unsigned lbase[2], key5[KT][2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    lbase[r] = ringb + (unsigned)lrow[r] * ROWB + (unsigned)pp * 8u;
#pragma unroll
    for (int j = 0; j < KT; ++j) { const int rr = shj[j] + lrow[r]; key5[j][r] = (unsigned)(((rr >> 2) & 1) | (((rr >> 4) & 1) << 1)) << 5; }
  }

  const int nb = (p.B - z + p.bstep - 1) / p.bstep;
  const int total = nb * tchunks;
  int cb[4], ct0[4];
  cb[0] = z; ct0[0] = 0;
  auto next_chunk = [&](int b, int t0, int& nb_, int& nt0) __attribute__((always_inline)) {
    nt0 = t0 + KB; nb_ = b;
    if (nt0 >= tchunks * KB) { nt0 = 0; nb_ = b + p.bstep; }
  };
#pragma unroll
  for (int k = 1; k < 4; ++k) next_chunk(cb[k - 1], ct0[k - 1], cb[k], ct0[k]);

  auto loadA = [&](auto set, int b, int t0) __attribute__((always_inline)) {
    constexpr int SET = decltype(set)::value;
    const unsigned so = ((unsigned)b * (unsigned)p.sab + (unsigned)t0) * 4u;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) {
        AH[SET][i][s2] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)(arow + (so + i * a16)) + s2 * 128, 0, 0));
        AL[SET][i][s2] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)(arow + (so + i * a16)) + s2 * 128 + 64, 0, 0));
      }
  };
  // one of the chunk's 4 WM dH loads / 4 input loads: the chunk loop issues them one at a time between its MFMA groups
  auto loadA1 = [&](auto set, int b, int t0, int idx) __attribute__((always_inline)) {
    constexpr int SET = decltype(set)::value;
    const unsigned so = ((unsigned)b * (unsigned)p.sab + (unsigned)t0) * 4u;
    const int i = idx / (2 * KS), s2 = (idx / 2) % KS, hf = idx % 2;
    const uint4 v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)(arow + (so + i * a16)) + s2 * 128 + 64 * hf, 0, 0));
    if (hf) AL[SET][i][s2] = v; else AH[SET][i][s2] = v;
  };
  auto loadX1 = [&](int b, int t0, int ci) __attribute__((always_inline)) {
    const unsigned so = ((unsigned)b * (unsigned)p.sxb + (unsigned)t0) * 4u;
    rx[ci] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)(xrow + (so + ci * x1)), 0, 0));
  };
  auto splitA = [&](auto set, int t0) __attribute__((always_inline)) {
    constexpr int SET = decltype(set)::value;
    const bool ragged = t0 + KB > p.La;                      // the item's last chunk(s): time steps at or past the row length count as zeros
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) {
        float v[8];
        const uint4 t0_ = AH[SET][i][s2], t1_ = AL[SET][i][s2];
        v[0] = __builtin_bit_cast(float, t0_.x); v[1] = __builtin_bit_cast(float, t0_.y); v[2] = __builtin_bit_cast(float, t0_.z); v[3] = __builtin_bit_cast(float, t0_.w);
        v[4] = __builtin_bit_cast(float, t1_.x); v[5] = __builtin_bit_cast(float, t1_.y); v[6] = __builtin_bit_cast(float, t1_.z); v[7] = __builtin_bit_cast(float, t1_.w);
        if (ragged) {
          const int nv = p.La - (t0 + 32 * s2 + 4 * kq);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (e >= 4 ? e + 12 : e) < nv ? v[e] : 0.f;      // (the second tuple holds steps 16 .. 19 of the lane's window)
        }
        split8p<F16>(v, as, AH[SET][i][s2], AL[SET][i][s2]);
      }
  };
  auto loadX = [&](int b, int t0) __attribute__((always_inline)) {
    const unsigned so = ((unsigned)b * (unsigned)p.sxb + (unsigned)t0) * 4u;
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) rx[ci] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)(xrow + (so + ci * x1)), 0, 0));
  };
  auto commitX = [&](int slot, int t0) __attribute__((always_inline)) {
    float v[4][4];
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      v[ci][0] = __builtin_bit_cast(float, rx[ci].x); v[ci][1] = __builtin_bit_cast(float, rx[ci].y);
      v[ci][2] = __builtin_bit_cast(float, rx[ci].z); v[ci][3] = __builtin_bit_cast(float, rx[ci].w);
    }
    if (!(rows_in_c && t0 + KB <= p.Lx)) {                   // (wave-uniform) ragged channel tile, or the block reaches the row end
      const int nv = p.Lx - (t0 + 4 * tq);
#pragma unroll
      for (int ci = 0; ci < 4; ++ci) {
        const bool cok = c0 + 16 * wave + 4 * cq + ci < p.Nc;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[ci][i] = (cok && i < nv) ? v[ci][i] : 0.f;
      }
    }
    const unsigned a = waddr + (unsigned)slot * (64u * ROWB);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned h0, l0, h1, l1;
      split_pair<F16>(v[0][i], v[1][i], xs, h0, l0);
      split_pair<F16>(v[2][i], v[3][i], xs, h1, l1);
      ssv_lds_store2(a + i * ROWB, h0, h1);
      ssv_lds_store2(a + i * ROWB + PLANE, l0, l1);
      if (slot == 0 && tq < 8) {                             // rows 0 .. 31 again as rows 256 .. 287
        ssv_lds_store2(a + i * ROWB + 256u * ROWB, h0, h1);
        ssv_lds_store2(a + i * ROWB + 256u * ROWB + PLANE, l0, l1);
      }
    }
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;

  // chunk n: MFMAs of the three taps on ring blocks n-1 .. n+1 | block n+2 (loaded during chunk n-1) is split and stored after the first tap |
  // the 4 loads of block n+3 go into the registers just freed, one per MFMA group of the second tap | dH of chunk n+1: its 8 loads one per
  // group of the first tap, split in place after the last tap
  auto chunk = [&](auto par, int n) __attribute__((always_inline)) {
    using CUR = std::integral_constant<int, decltype(par)::value>;
    using NXT = std::integral_constant<int, decltype(par)::value ^ 1>;
#ifdef SSV_NT_STAMP
    const bool stamp_on = blockIdx.x == 0 && blockIdx.z == 0 && tid == 0;
    const int stamp_s = n + 16;                                  // chunks 8 .. 15 (NT_STAMP records "steps" 24 .. 31)
#endif
    NT_STAMP(0);
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      constexpr int G = KS * NTC;
      uint4 fb[R_FD + 1][2];
      unsigned su[KS];
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) su[s2] = (unsigned)((64 * n + shj[j] + 32 * s2) & 255) * ROWB;         // scalar
      auto frag = [&](int g, uint4 (&f)[2]) __attribute__((always_inline)) {
        const int s2 = g / NTC, cbk = g % NTC;
        const unsigned a0 = lbase[0] + su[s2] + ((unsigned)(cbk << 5) ^ key5[j][0]);
        const unsigned a1 = lbase[1] + su[s2] + ((unsigned)(cbk << 5) ^ key5[j][1]);
        const uint2 h0u = ssv_lds_read_tr16(a0), h1u = ssv_lds_read_tr16(a1);
        const uint2 l0u = ssv_lds_read_tr16(a0 + PLANE), l1u = ssv_lds_read_tr16(a1 + PLANE);
        f[0] = make_uint4(h0u.x, h0u.y, h1u.x, h1u.y);
        f[1] = make_uint4(l0u.x, l0u.y, l1u.x, l1u.y);
      };
#pragma unroll
      for (int g = 0; g < R_FD; ++g) frag(g, fb[g]);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (g + R_FD < G) frag(g + R_FD, fb[(g + R_FD) % (R_FD + 1)]);
        // one vector-memory instruction per group: among MFMAs its issue costs the wave less than in a batch (-2.7 % on the launch)
        if (j == 0 && g < 4 * WM) loadA1(NXT{}, cb[1], ct0[1], g);
        if (j == 1 && g < 4) loadX1(cb[3], ct0[3], g);
        __builtin_amdgcn_sched_barrier(0);
        const int s2 = g / NTC, q = g % NTC;
        if (ct0[0] + 32 * s2 >= p.La) continue;                  // (wave-uniform) k-steps at or past the row length: dH is zero there
        const uint4 bh = fb[g % (R_FD + 1)][0];
        const uint4 bl = fb[g % (R_FD + 1)][1];
#pragma unroll
        for (int i = 0; i < WM; ++i) {
          const uint4 a_h = AH[CUR::value][i][s2];
          const uint4 a_l = AL[CUR::value][i][s2];
          acc[i][j][q] = mma16<F16>(a_l, bh, acc[i][j][q]);
          acc[i][j][q] = mma16<F16>(a_h, bl, acc[i][j][q]);
          acc[i][j][q] = mma16<F16>(a_h, bh, acc[i][j][q]);
        }
      }
      if (j == 0) {
        NT_STAMP(2);
        commitX((n + 2) & 3, ct0[2]);
        NT_STAMP(3);
      }
      if (j == 1) NT_STAMP(5);
      if (j == KT - 1) {
        NT_STAMP(6);
        splitA(NXT{}, ct0[1]);
        NT_STAMP(7);
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) { cb[k] = cb[k + 1]; ct0[k] = ct0[k + 1]; }
    next_chunk(cb[2], ct0[2], cb[3], ct0[3]);
  };

  if (total > 0) {
    // prologue: ring block -1 (slot 3: the rows before the slab's first item) zeroed, blocks 0 and 1 staged, block 2 in flight, dH of chunk 0 split
    loadA(P0{}, cb[0], ct0[0]);
    loadX(cb[0], ct0[0]);
    {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ssv_lds_store2(waddr + 3u * 64u * ROWB + i * ROWB, 0u, 0u);
        ssv_lds_store2(waddr + 3u * 64u * ROWB + i * ROWB + PLANE, 0u, 0u);
      }
    }
    commitX(0, ct0[0]);
    loadX(cb[1], ct0[1]);
    splitA(P0{}, ct0[0]);
    commitX(1, ct0[1]);
    loadX(cb[2], ct0[2]);
    __syncthreads();
    int n = 0;
    for (; n + 1 < total; n += 2) {
      chunk(P0{}, n);
      chunk(P1{}, n + 1);
    }
    if (n < total) chunk(P0{}, n);
  }

  NT_WG(3, __builtin_readcyclecounter());
  // a job whose shifts exceed the bound the launch was planned for (the caller's max_shift) must not pass for a result
  const bool bad_shift = abs(shj[0]) > max_shift || abs(shj[1]) > max_shift || abs(shj[2]) > max_shift;
  float* __restrict__ Cz = Cp + (long)z * p.scz;
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gm = m0 + wave * WM * 16 + i * 16 + kq * 4 + r;
      if (gm >= p.M) continue;
#pragma unroll
      for (int j = 0; j < KT; ++j)
#pragma unroll
        for (int q = 0; q < NTC; ++q) {
          const int gc = c0 + q * 16 + nq;
          if (gc < p.Nc) Cz[(long)gm * p.scm + (long)gc * p.scc + (long)j * p.scj] = bad_shift ? __builtin_nanf("") : (F16 ? acc[i][j][q][r] * us : acc[i][j][q][r]);
        }
    }
#ifdef SSV_NT_STAMP
  __builtin_amdgcn_s_waitcnt(0);
#endif
  NT_WG(1, __builtin_amdgcn_s_memrealtime());
}

// Tile plan for the weight gradient.  The output (M x Nc x KT) is small, so the reduction axis (batch x time) is cut into
// Z slabs that are summed afterwards; slab traffic (Z x output, written and read back) competes with the operand reads,
// so smaller tiles with fewer slabs win when the output is small.
void ssv_nt_bf3_tile(int KT, int M, int Nc, int* wm, int* ntc) {
  // measured (round-1 sweep; in-step re-check with SSV_NT_FORCE, tools/prof_env.sh): k=3 -- the largest tile wins at every hot shape; k=1 carries a third of the MFMAs per
  // staged byte, so only large outputs (513 x 513) keep the 128 x 96 tile, smaller ones take 64 x 64 tiles with fewer slabs
  if (KT == 3) { *wm = 2; *ntc = 4; }
  else if (Nc <= 48) { *wm = 2; *ntc = 2; }
  else if ((long)M * Nc >= (1L << 18)) { *wm = 2; *ntc = 6; }
  else { *wm = 1; *ntc = 4; }
}
// co-resident workgroups per CU (4 waves each, one per SIMD): 512 / VGPRs of the instantiation, as compiled for gfx950
// (-Rpass-analysis=kernel-resource-usage: <3,2,4> 236, <3,2,2> 174, <3,1,4> 156, <3,1,2> 110, <1,2,6> 208, <1,2,4> 168, <1,2,2> 134,
// <1,1,6> 148, <1,1,4> 120, <1,1,2> 94; the two LDS buffers of the largest tile (32 KB) allow 4)
int ssv_nt_bf3_wg_per_cu(int KT, int wm, int ntc) {
  if (KT == 3) return wm == 2 ? 2 : (ntc >= 4 ? 3 : 4);
  if (wm == 2) return ntc >= 6 ? 2 : 3;
  return ntc >= 6 ? 3 : (ntc >= 4 ? 4 : 5);
}
int ssv_nt_bf3_tiles(int KT, int M, int Nc) {
  int wm, ntc;
  ssv_nt_bf3_tile(KT, M, Nc, &wm, &ntc);
  return ssv_cdiv(ssv_nt_bf3_xrow(KT, M, Nc) ? M - 1 : M, 64 * wm) * ssv_cdiv(Nc, 16 * ntc);
}
// the k = 1 weight gradient of 128 j + 1 output rows on the 128 x 96 tile: last row beside the staging, range slabs (gemm_nt_bf3_kernel<.., XR = 1>)
bool ssv_nt_bf3_xrow(int KT, int M, int Nc) {
  int wm, ntc;
  ssv_nt_bf3_tile(KT, M, Nc, &wm, &ntc);
  return SSV_NT_XROW && KT == 1 && wm == 2 && ntc == 6 && M > 128 && M % 128 == 1;
}

// the kernel addresses both operands with 32-bit element offsets
bool ssv_nt_bf3_fits(const GemmNT& g) {
  const long lim = 1L << 30;
  return (long)(g.B - 1) * g.sab + (long)(g.M - 1) * g.sam + g.La < lim && (long)(g.B - 1) * g.sxb + (long)(g.Nc - 1) * g.sxc + g.Lx < lim &&
         (long)g.B * g.sab < lim && (long)g.B * g.sxb < lim && g.La >= 8 && g.Lx >= 8;
}
int ssv_launch_gemm_nt_bf3(const GemmNT& g, hipStream_t st) {
  SSV_CHECK(g.M > 0 && g.Nc > 0 && g.La > 0 && g.B > 0 && g.Z > 0 && g.bstep >= 0, SSV_BAD_SHAPE, "gemm_nt_bf3: empty problem");      // (bstep == 0: range slabs)
  SSV_CHECK(g.KT == 1 || g.KT == 3, SSV_UNSUPPORTED, "gemm_nt_bf3: kernel_size %d", g.KT);
  SSV_CHECK(g.sat == 1 && g.sxn == 1, SSV_UNSUPPORTED, "gemm_nt_bf3: rows must be contiguous in time");
  SSV_CHECK(g.Z <= 65535, SSV_UNSUPPORTED, "gemm_nt_bf3: Z=%d exceeds grid.z", g.Z);
  SSV_CHECK(ssv_nt_bf3_fits(g), SSV_UNSUPPORTED, "gemm_nt_bf3: an operand spans 2^30 elements or more");
  int wm, ntc;
  ssv_nt_bf3_tile(g.KT, g.M, g.Nc, &wm, &ntc);
  const bool xr = ssv_nt_bf3_xrow(g.KT, g.M, g.Nc);       // (then the caller chose range slabs: bstep == 0)
  SSV_CHECK(g.bstep > 0 || xr, SSV_UNSUPPORTED, "gemm_nt_bf3: range slabs are built for the extra-row kernel only");
  const int mtiles = ssv_cdiv(xr ? g.M - 1 : g.M, 64 * wm);
  const int nz = g.jobs ? g.njobs * g.Z : g.Z;
  SSV_CHECK(nz <= 65535, SSV_UNSUPPORTED, "gemm_nt_bf3: %d slabs exceed grid.z", nz);
  const dim3 grid(mtiles * ssv_cdiv(g.Nc, 16 * ntc), 1, nz);
  SSV_CHECK(!g.f16 || g.jobs || (g.a_amax && g.x_amax && g.a_namax > 0 && g.x_namax > 0), SSV_BAD_SHAPE, "gemm_nt_bf3: split-fp16 needs both operand scales");
  // the ring kernel (k = 3): every shift within one block (64 time steps) either way.  With a job table the shifts are on the device: the caller states their bound.
  int ring_ms = -1;
#if SSV_NT_RING
  if (g.KT == 3 && wm == 2 && ntc == 4) {
    ring_ms = g.max_shift;
    if (!g.jobs) { ring_ms = 0; for (int j = 0; j < 3; ++j) ring_ms = abs(g.shift[j]) > ring_ms ? abs(g.shift[j]) : ring_ms; }
    if (ring_ms > 64) ring_ms = -1;
  }
#endif
  if (ssv_shape_log_on()) {
    char nm[96], note[96];
    const int nj = g.jobs ? g.njobs : 1;
    if (ring_ms >= 0) snprintf(nm, sizeof nm, "gemm_nt3r_kernel<%d, %d>", wm, g.f16);
    else snprintf(nm, sizeof nm, xr ? "gemm_nt_bf3_kernel<%d, %d, %d, %d, 1>" : "gemm_nt_bf3_kernel<%d, %d, %d, %d>", g.KT, wm, ntc, g.f16);
    snprintf(note, sizeof note, "jobs=%d B=%d M=%d Nc=%d L=%d k=%d Z=%d", nj, g.B, g.M, g.Nc, g.La, g.KT, g.Z);
    ssv_shape_log(nm, grid, dim3(256), 2.0 * nj * g.B * g.M * g.Nc * g.La * g.KT,
                  4.0 * nj * ((double)g.B * g.M * g.La + (double)g.B * g.Nc * g.Lx + (double)g.Z * g.M * g.Nc * g.KT), note);
  }
  if (ring_ms >= 0) {
    const int tchunks = ssv_cdiv((g.La > g.Lx ? g.La : g.Lx) + ring_ms, 64);
    if (g.f16) hipLaunchKernelGGL((gemm_nt3r_kernel<2, 1>), grid, dim3(256), 0, st, g, mtiles, tchunks, ring_ms);
    else hipLaunchKernelGGL((gemm_nt3r_kernel<2, 0>), grid, dim3(256), 0, st, g, mtiles, tchunks, ring_ms);
    return ssv_check_launch("gemm_nt3r");
  }
  if (xr) {
    if (g.f16) hipLaunchKernelGGL((gemm_nt_bf3_kernel<1, 2, 6, 1, 1>), grid, dim3(256), 0, st, g, mtiles);
    else hipLaunchKernelGGL((gemm_nt_bf3_kernel<1, 2, 6, 0, 1>), grid, dim3(256), 0, st, g, mtiles);
    return ssv_check_launch("gemm_nt_bf3 (extra row)");
  }
#define SSV_NT(K_, A_, C_) if (g.KT == K_ && wm == A_ && ntc == C_) { \
    if (g.f16) hipLaunchKernelGGL((gemm_nt_bf3_kernel<K_, A_, C_, 1>), grid, dim3(256), 0, st, g, mtiles); \
    else hipLaunchKernelGGL((gemm_nt_bf3_kernel<K_, A_, C_, 0>), grid, dim3(256), 0, st, g, mtiles); \
    return ssv_check_launch("gemm_nt_bf3"); }
  SSV_NT(3, 2, 4) SSV_NT(3, 2, 2) SSV_NT(3, 1, 4) SSV_NT(3, 1, 2)
  SSV_NT(1, 2, 6) SSV_NT(1, 2, 4) SSV_NT(1, 2, 2) SSV_NT(1, 1, 6) SSV_NT(1, 1, 4) SSV_NT(1, 1, 2)
#undef SSV_NT
  return ssv_fail(SSV_UNSUPPORTED, "gemm_nt_bf3: no tile %d,%d for kernel size %d", wm, ntc, g.KT);
}

// ---- 1x1 conv + LayerNorm over channels (+ activation) in ONE launch (round 4) ------------------------------------------------
// y = act(LN(W x + bias [+ s]))  -- models/TTSModel.py:128-131, :173-180, :218-231, :343-361.  Until round 4 every such link was two
// launches (the k = 1 GEMM, then the LayerNorm kernel re-reading its output).  A 1x1 convolution and a channel LayerNorm are both
// per-column operations, so a workgroup that owns ALL output rows of a column tile can finish the LayerNorm from its accumulators:
// no cross-workgroup step, no second pass over `pre`.  8 waves split the M axis (wave w: row blocks w * WMB .. w * WMB + WMB - 1, so
// BM = 128 * WMB >= M), every wave all 16 * NT columns; weight fragments go L2 -> registers (private rows per wave, two sets), the
// input tile is staged in LDS once per 32-channel chunk (two images, one barrier per chunk) exactly as in gemm_nn_bf3_kernel.  The
// price is the weight stream: every workgroup reads all of W (the row-tiled kernels re-read X instead).
// Epilogue: pre = acc * us + bias (+ s[b]) is stored; column sums of a lane's rows -> the 4 row-quads of the wave (cross-row shuffles)
// -> the 8 waves (LDS), mean, then the same for the squared deviations (a true two-pass variance, as ln_act_fwd_kernel); y = act(n).
// Tuning builds only (-DSSV_PW_STAMP): thread 0 of every workgroup (the first 1024) records s_memrealtime at entry and exit and the shader
// clock at six points (entry | first chunk staged | chunk loop done | pre stored + column sums | variance | y stored); ssv_debug_pw_stamps().
#ifdef SSV_PW_STAMP
__device__ unsigned long long ssv_pw_stamps[1024 * 8];
extern "C" int ssv_debug_pw_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ssv_pw_stamps), sizeof(ssv_pw_stamps)); }
#define PW_STAMP(k) do { const unsigned w_ = blockIdx.y * gridDim.x + blockIdx.x; if (threadIdx.x == 0 && w_ < 1024u) \
    ssv_pw_stamps[w_ * 8 + (k)] = ((k) >= 6) ? __builtin_amdgcn_s_memrealtime() : __builtin_readcyclecounter(); } while (0)
#else
#define PW_STAMP(k) do {} while (0)
#endif
#ifndef SSV_PWLN_XROW
#define SSV_PWLN_XROW 1      // (tuning builds: 0 = M = 513 on five row blocks per wave, as before)
#endif
#ifndef SSV_PWLN_ROLL
#define SSV_PWLN_ROLL 1      // (tuning builds: 0 = the one weight-fragment set of 4 row blocks per wave re-loaded at the end of the chunk)
#endif
#ifndef SSV_PWLN_PARK
#define SSV_PWLN_PARK 1      // (tuning builds: 0 = `pre` and `y` stored straight from the accumulator layout, as in round 4)
#endif
struct PwLn {
  GemmNNB g;                      // A planes, X, C = pre (B, M, N), bias, bias_b, f16 scales
  const float* gamma; const float* beta;
  float* y; long ybs; float* stats; float* y_amax; int namax; int act;
};
template <int WMB, int NT, int F16, int XR = 0>
__global__ __launch_bounds__(512, 2) void gemm_pwln_kernel(const PwLn q) {
  const GemmNNB& p = q.g;
  constexpr int BN = 16 * NT;
  constexpr int X_SLOTS = 4 * BN;                      // 16-byte slots of one 32-channel chunk: [k-group][column]
  static_assert(X_SLOTS <= 512, "one slot per thread");
  constexpr int IMG = 2 * X_SLOTS;
  __shared__ uint4 lds[2][IMG];
  __shared__ float red[8][BN];
  __shared__ float colv[2][BN];
  __shared__ float amx[8];
  // XR: row M - 1 (M = 128 j + 1: the 513-channel layers) is kept out of the MFMA row blocks -- a fifth row block per wave for ONE row otherwise
  // -- and comes from fp32 dot products of its weights (xw, staged once) with the raw values the staging threads hold before they split them
  __shared__ float xw[XR ? 1056 : 1];
  __shared__ float xsum[XR ? 4 * BN : 1];
  __shared__ float xrow[XR ? 2 * BN : 1];                // the row's pre-activation per column, then its normalised value
  // epilogue (round 5): every wave parks a 16-row block row-major and reads it back as 16-byte vectors along the rows, so a store instruction
  // covers four whole 256-byte row pieces instead of 64-byte pieces of 16 rows (what gemm_nn_bf3_kernel's epilogue has done since round 2)
  constexpr int LDWP = BN + 4;
  __shared__ float park[SSV_PWLN_PARK ? 8 * 16 * LDWP : 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned wg = ssv_xcd_order(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  const int ntile = (int)(wg % gridDim.x), b = (int)(wg / gridDim.x);
  const int n0 = ntile * BN;
  const int kq = lane >> 4, nq = lane & 15;
  const int nchunks = p.Kpad / 32;
  const float* __restrict__ Xb = p.X + (long)b * p.sxb;
  PW_STAMP(6); PW_STAMP(0);

  f32x4 acc[WMB][NT];
#pragma unroll
  for (int i = 0; i < WMB; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // weight fragments: two register sets (chunk c + 1 loaded while chunk c multiplies) while they fit; from 4 row blocks per wave on, one set,
  // re-loaded right after the chunk's MFMAs have been issued (accumulators + two sets would not fit the 256 registers of an 8-wave workgroup)
  constexpr int NSET = WMB >= 4 ? 1 : 2;
  uint4 Ah_[NSET][WMB], Al_[NSET][WMB];
  float rx[8];

  const int Mt = XR ? p.M - 1 : p.M;                     // rows in the MFMA row blocks
  const int MB = (p.M + 15) >> 4;
  unsigned arowb[WMB];
#pragma unroll
  for (int i = 0; i < WMB; ++i) arowb[i] = (unsigned)(((long)min(wave * WMB + i, MB - 1) * nchunks * 512 + lane * 8) * 2);
  float xacc = 0.f;
  if constexpr (XR) {
    for (int k = tid; k < nchunks * 32; k += 512) xw[k] = k < p.Kc ? p.xrow_w[(long)k * p.xrow_sk] : 0.f;      // (visible after the first barrier)
  }
  const __amdgpu_buffer_rsrc_t rsAh = ssv_buf(p.Ahi), rsAl = ssv_buf(p.Alo), rsX = ssv_buf(Xb);
  auto loadA = [&](int set, int ch) {
    // (readfirstlane: hipcc kept this offset in a vector register in some instantiations and wrapped every load in a waterfall loop)
    const unsigned ub = (unsigned)__builtin_amdgcn_readfirstlane(ch * 1024);
#pragma unroll
    for (int i = 0; i < WMB; ++i) { Ah_[set][i] = ssv_buf_u4(rsAh, arowb[i], ub); Al_[set][i] = ssv_buf_u4(rsAl, arowb[i], ub); }
  };
  const int Lrow = (int)p.sxc;
  const int skg = tid / BN, scol = tid % BN;             // this thread's staging slot (k-group, column) when tid < X_SLOTS
  const bool stager = tid < X_SLOTS;
  const bool cvs = stager && n0 + scol < p.Lx;
  const unsigned voffb = (unsigned)((stager ? 8 * skg : 0) * Lrow + min(n0 + scol, p.Lx - 1)) * 4u;
  const bool ragged = (p.Kc & 31) != 0;
  float xs = 1.f, xinv = 1.f, ainv = 1.f;
  if constexpr (F16) ainv = *p.a_inv;
  auto prefetchX = [&](int ch) {
    if (!ragged || ch + 1 < nchunks) {
#pragma unroll
      for (int i = 0; i < 8; ++i) rx[i] = ssv_buf_f32(rsX, voffb, (unsigned)((ch * 32 + i) * Lrow) * 4u);
    } else {
      const unsigned colo = voffb - (unsigned)(8 * (stager ? skg : 0) * Lrow) * 4u;
#pragma unroll
      for (int i = 0; i < 8; ++i) rx[i] = ssv_buf_f32(rsX, (unsigned)min(ch * 32 + 8 * skg + i, p.Kc - 1) * (unsigned)Lrow * 4u + colo, 0u);
    }
  };
  // the rolling loop's prefetch: ONE path (a buffer whose range is the batch item's Kc rows: channels past Kc read 0, tools/probe/buf_oob.hip),
  // because with the two-path form above hipcc must assume at the loop head that neither path ran and waits for all weight fragments at once
  const __amdgpu_buffer_rsrc_t rsXr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Xb), 0, (int)(((long)(p.Kc - 1) * Lrow + p.Lx) * 4), 0x00020000);
  auto prefetchXr = [&](int ch) {
#pragma unroll
    for (int i = 0; i < 8; ++i) rx[i] = ssv_buf_f32(rsXr, voffb, (unsigned)((ch * 32 + i) * Lrow) * 4u);
  };
  auto commitX = [&](int ch) {
    if (!stager) return;
    const bool last_ragged = ragged && ch + 1 == nchunks;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (cvs && (!last_ragged || ch * 32 + 8 * skg + i < p.Kc)) ? rx[i] : 0.f;
    if constexpr (XR) {
      const float* wq = xw + ch * 32 + 8 * skg;
#pragma unroll
      for (int i = 0; i < 8; ++i) xacc = __builtin_fmaf(wq[i], v[i], xacc);
    }
    uint4 h, l;
    split8s<F16>(v, xs, h, l);
    lds[ch & 1][tid] = h; lds[ch & 1][X_SLOTS + tid] = l;
  };
  auto tap = [&](int set, int ch) {
    const uint4* Xh = lds[ch & 1];
    const uint4* Xl = lds[ch & 1] + X_SLOTS;
    uint4 fb[2][2];
    auto frag = [&](int t, uint4 (&f)[2]) __attribute__((always_inline)) { const int s_ = kq * BN + t * 16 + nq; f[0] = Xh[s_]; f[1] = Xl[s_]; };
    frag(0, fb[0]);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t + 1 < NT) frag(t + 1, fb[(t + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      const uint4 bh = fb[t & 1][0], bl = fb[t & 1][1];
#pragma unroll
      for (int i = 0; i < WMB; ++i) {
        acc[i][t] = mma16<F16>(Al_[set][i], bh, acc[i][t]);
        acc[i][t] = mma16<F16>(Ah_[set][i], bl, acc[i][t]);
        acc[i][t] = mma16<F16>(Ah_[set][i], bh, acc[i][t]);
      }
    }
  };
  loadA(0, 0);
  prefetchX(0);
  if constexpr (F16) {
    float sc, inv;
    ssv_pow2_scale(ssv_wave_list_max(p.x_amax + (long)b * p.x_amax_bs, p.x_namax), sc, inv);
    xs = ssv_uniform(sc); xinv = ssv_uniform(inv);
  }
  if constexpr (XR) __syncthreads();                     // xw is staged
  commitX(0);
  if constexpr (SSV_PWLN_ROLL) prefetchXr(min(1, nchunks - 1));
  else if (nchunks > 1) prefetchX(1);
  if constexpr (NSET == 2 && SSV_PWLN_ROLL) {
    // two weight-fragment sets, every load unconditional (clamped chunk index, one-path prefetch: see the rolling loop below)
    const int last = nchunks - 1;
    loadA(1, min(1, last));
    __syncthreads();
    PW_STAMP(1);
    for (int ch = 0; ch < nchunks; ch += 2) {
      tap(0, ch);
      if (ch + 1 < nchunks) commitX(ch + 1);
      prefetchXr(min(ch + 2, last)); loadA(0, min(ch + 2, last));
      __syncthreads();
      if (ch + 1 < nchunks) tap(NSET - 1, ch + 1);
      if (ch + 2 < nchunks) commitX(ch + 2);
      prefetchXr(min(ch + 3, last)); loadA(NSET - 1, min(ch + 3, last));
      __syncthreads();
    }
  } else if constexpr (NSET == 2) {
    if (nchunks > 1) loadA(1, 1);
    __syncthreads();
    PW_STAMP(1);
    for (int ch = 0; ch < nchunks; ch += 2) {
      tap(0, ch);
      if (ch + 1 >= nchunks) break;
      commitX(ch + 1);
      if (ch + 2 < nchunks) { prefetchX(ch + 2); loadA(0, ch + 2); }
      __syncthreads();
      tap(NSET - 1, ch + 1);
      if (ch + 2 < nchunks) {
        commitX(ch + 2);
        if (ch + 3 < nchunks) { prefetchX(ch + 3); loadA(NSET - 1, ch + 3); }
      }
      __syncthreads();
    }
  } else if constexpr (SSV_PWLN_ROLL) {
    // One register set, re-loaded ROW BLOCK BY ROW BLOCK (round 5): the loop runs row block outermost with all NT input fragments of the chunk
    // in registers, so row block i's weight fragments are dead after its 3 NT MFMAs and chunk c + 1's are requested right there -- every
    // fragment gets (WMB - 1) / WMB of a chunk of lead time.  (Before: the whole set was re-loaded after the chunk's last MFMA, i.e. the first
    // MFMA of the next chunk waited for a full L2 round trip -- the 8 waves run in lock step, nothing else was there to cover it.)
    // The re-load is unconditional (the last chunk re-reads itself): a load under a condition makes hipcc drain the whole queue.
    __syncthreads();
    PW_STAMP(1);
    for (int ch = 0; ch < nchunks; ++ch) {
      const uint4* Xh = lds[ch & 1];
      const uint4* Xl = lds[ch & 1] + X_SLOTS;
      uint4 bh[NT], bl[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) { const int s_ = kq * BN + t * 16 + nq; bh[t] = Xh[s_]; bl[t] = Xl[s_]; }
      const unsigned ubn = (unsigned)__builtin_amdgcn_readfirstlane(min(ch + 1, nchunks - 1) * 1024);
#pragma unroll
      for (int i = 0; i < WMB; ++i) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          acc[i][t] = mma16<F16>(Al_[0][i], bh[t], acc[i][t]);
          acc[i][t] = mma16<F16>(Ah_[0][i], bl[t], acc[i][t]);
          acc[i][t] = mma16<F16>(Ah_[0][i], bh[t], acc[i][t]);
        }
        __builtin_amdgcn_sched_barrier(0);          // the re-load stays behind this row block's MFMAs, and in front of the next block's
        Ah_[0][i] = ssv_buf_u4(rsAh, arowb[i], ubn);
        Al_[0][i] = ssv_buf_u4(rsAl, arowb[i], ubn);
      }
      // (the prefetch is unconditional too -- the last chunks re-read the last one -- so that every path into the loop head has the same loads in
      //  flight and hipcc can wait for row block 0's fragments alone, vmcnt(14), instead of for the youngest count over all paths)
      if (ch + 1 < nchunks) commitX(ch + 1);
      prefetchXr(min(ch + 2, nchunks - 1));
      __syncthreads();
    }
  } else {
    __syncthreads();
    PW_STAMP(1);
    for (int ch = 0; ch < nchunks; ++ch) {
      tap(0, ch);
      __builtin_amdgcn_sched_barrier(0);            // the re-load stays behind this chunk's MFMAs
      if (ch + 1 < nchunks) {
        loadA(0, ch + 1);
        commitX(ch + 1);
        if (ch + 2 < nchunks) prefetchX(ch + 2);
      }
      __syncthreads();
    }
  }

  // ---- epilogue: pre, LayerNorm over the M rows of every column, activation
  PW_STAMP(2);
  const float us = F16 ? ssv_uniform(xinv * ainv) : 1.f;
  float* __restrict__ Cb = p.C + (long)b * p.scb;
  float* __restrict__ Yb = q.y + (long)b * q.ybs;
  float csum[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) csum[t] = 0.f;
  // the wave's parked block (rows rbase .. rbase + 15, private to the wave: its LDS operations complete in order) -> dst rows, 16 bytes per lane
  float* pk = park + (SSV_PWLN_PARK ? wave * 16 * LDWP : 0);
  auto store_block = [&](float* __restrict__ dst, long row_stride, int rbase) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      const int e = lane + 64 * it;
      const int row = e / (BN / 4), c4 = e % (BN / 4);
      const int gm = rbase + row, gn = n0 + c4 * 4;
      const f32x4 v = *reinterpret_cast<const f32x4*>(pk + row * LDWP + c4 * 4);
      if (gm < Mt && gn < p.N) {
        float* o = dst + (long)gm * row_stride + gn;
        if (gn + 3 < p.N) { f4u w = {v[0], v[1], v[2], v[3]}; *reinterpret_cast<f4u*>(o) = w; }
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (gn + j < p.N) o[j] = v[j];
        }
      }
    }
  };
#pragma unroll
  for (int i = 0; i < WMB; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gm = (wave * WMB + i) * 16 + kq * 4 + r;
      const bool rv = gm < Mt;
      const int gmc = min(gm, Mt - 1);
      float add = 0.f;
      if (p.bias) add += p.bias[gmc];
      if (p.bias_b) add += p.bias_b[(long)b * p.sbb + gmc];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float v = rv ? (F16 ? acc[i][t][r] * us : acc[i][t][r]) + add : 0.f;
        acc[i][t][r] = v;
        csum[t] += v;
        if constexpr (SSV_PWLN_PARK) pk[(kq * 4 + r) * LDWP + t * 16 + nq] = v;
        else { const int gn = n0 + t * 16 + nq; if (rv && gn < p.N) Cb[(long)gm * p.scm + gn] = v; }
      }
    }
    if constexpr (SSV_PWLN_PARK) store_block(Cb, p.scm, (wave * WMB + i) * 16);
  }
  const float invM = 1.f / (float)p.M;
  if constexpr (XR) {                                     // row M - 1: the four k-groups' partial sums of a column, bias; stored, and kept for the LayerNorm
    if (stager) xsum[tid] = xacc;                         // (slot tid = skg * BN + scol)
    __syncthreads();
    if (tid < BN) {
      const int gm = p.M - 1, gn = n0 + tid;
      float v = (xsum[tid] + xsum[BN + tid]) + (xsum[2 * BN + tid] + xsum[3 * BN + tid]);
      if (p.bias) v += p.bias[gm];
      if (p.bias_b) v += p.bias_b[(long)b * p.sbb + gm];
      if (gn >= p.N) v = 0.f;
      else Cb[(long)gm * p.scm + gn] = v;
      xrow[tid] = v;
    }
  }
  // xtra: what thread tid < BN adds to its column's sum (the extra row's term); the reduction's own barrier orders xrow before its use
  auto col_reduce = [&](float (&v)[NT], int slot) __attribute__((always_inline)) {       // sum over all rows of the tile; result in colv[slot][column]
#pragma unroll
    for (int t = 0; t < NT; ++t) { v[t] += __shfl_xor(v[t], 16); v[t] += __shfl_xor(v[t], 32); }
    if (kq == 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t) red[wave][t * 16 + nq] = v[t];
    }
    __syncthreads();
    if (tid < BN) {
      float s_ = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s_ += red[w][tid];
      if constexpr (XR) {
        if (slot == 0) s_ += xrow[tid];
        else { const float d = n0 + tid < p.N ? xrow[tid] - colv[0][tid] * invM : 0.f; s_ += d * d; }
      }
      colv[slot][tid] = s_;
    }
    __syncthreads();
  };
  col_reduce(csum, 0);
  PW_STAMP(3);
  float mean[NT], qs[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { mean[t] = colv[0][t * 16 + nq] * invM; qs[t] = 0.f; }
#pragma unroll
  for (int i = 0; i < WMB; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool rv = (wave * WMB + i) * 16 + kq * 4 + r < Mt;
#pragma unroll
      for (int t = 0; t < NT; ++t) { const float d = rv ? acc[i][t][r] - mean[t] : 0.f; qs[t] += d * d; }
    }
  col_reduce(qs, 1);
  PW_STAMP(4);
  float rstd[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) rstd[t] = rsqrtf(colv[1][t * 16 + nq] * invM + 1e-5f);
  if (q.stats && wave == 0 && kq == 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int gn = n0 + t * 16 + nq;
      if (gn < p.N) { q.stats[(long)b * 2 * p.N + gn] = mean[t]; q.stats[(long)b * 2 * p.N + p.N + gn] = rstd[t]; }
    }
  }
  float am = 0.f;
#pragma unroll
  for (int i = 0; i < WMB; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gm = (wave * WMB + i) * 16 + kq * 4 + r;
      const bool rv = gm < Mt;
      const int gmc = min(gm, Mt - 1);
      const float ga = q.gamma[gmc], be = q.beta[gmc];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int gn = n0 + t * 16 + nq;
        float n = (acc[i][t][r] - mean[t]) * rstd[t] * ga + be;
        if (q.act == 1) n = fmaxf(n, 0.f);
        else if (q.act == 2) n = 1.f / (1.f + __expf(-n));
        if (rv && gn < p.N) am = fmaxf(am, fabsf(n));
        if constexpr (SSV_PWLN_PARK) pk[(kq * 4 + r) * LDWP + t * 16 + nq] = n;
        else if (rv && gn < p.N) Yb[(long)gm * p.N + gn] = n;
      }
    }
    if constexpr (SSV_PWLN_PARK) store_block(Yb, p.N, (wave * WMB + i) * 16);
  }
  if constexpr (XR) {
    if (tid < BN && n0 + tid < p.N) {                      // row M - 1 of y (colv is final: behind the second reduction's barrier)
      const int gm = p.M - 1;
      float n = (xrow[tid] - colv[0][tid] * invM) * rsqrtf(colv[1][tid] * invM + 1e-5f) * q.gamma[gm] + q.beta[gm];
      if (q.act == 1) n = fmaxf(n, 0.f);
      else if (q.act == 2) n = 1.f / (1.f + __expf(-n));
      Yb[(long)gm * p.N + n0 + tid] = n;
      am = fmaxf(am, fabsf(n));
    }
  }
  PW_STAMP(5); PW_STAMP(7);
  if (q.y_amax) {                         // one entry per column tile, the rest of the item's list zeroed by the last tile
    am = ssv_wg_max<8>(am, amx);
    if (tid == 0) {
      float* al = q.y_amax + (long)b * q.namax;
      al[ntile] = am;
      if (ntile == (int)gridDim.x - 1) for (int e = gridDim.x; e < q.namax; ++e) al[e] = 0.f;
    }
  }
}
// WMB row blocks per wave (BM = 128 * WMB >= M), NT column blocks.  Returns SSV_UNSUPPORTED when no instantiation fits.
int ssv_launch_gemm_pwln(const GemmNNB& g, const float* gamma, const float* beta, float* y, long ybs, float* stats, float* y_amax, int namax, int act,
                         hipStream_t st) {
  SSV_CHECK(g.KT == 1 && g.sxn == 1 && g.scn == 1 && !g.epi && !g.perm_h && !g.R && !g.colstats && g.Kpad % 32 == 0, SSV_UNSUPPORTED, "gemm_pwln: plain 1x1 products only");
  SSV_CHECK(g.M <= 640 && g.B <= 65535, SSV_UNSUPPORTED, "gemm_pwln: %d output channels (max 640)", g.M);
  SSV_CHECK(!g.f16 || (g.a_inv && g.x_amax && g.x_namax > 0), SSV_BAD_SHAPE, "gemm_pwln: split-fp16 needs operand scales");
  PwLn q;
  q.g = g; q.gamma = gamma; q.beta = beta; q.y = y; q.ybs = ybs; q.stats = stats; q.y_amax = y_amax; q.namax = namax; q.act = act;
  // M = 513: four row blocks per wave for rows 0 .. 511 and the last row beside the staging (XR) instead of five row blocks
  const bool xr = SSV_PWLN_XROW && g.xrow_w && g.M == 513 && g.Kpad <= 1056;
  const int wmb = xr ? 4 : ssv_cdiv(ssv_cdiv(g.M, 16), 8);
  const int nt = 4;
  const dim3 grid(ssv_cdiv(g.N, 16 * nt), g.B);
  SSV_CHECK(!y_amax || namax >= (int)grid.x, SSV_BAD_SHAPE, "gemm_pwln: scale list shorter than the column tiles");
  if (ssv_shape_log_on()) {
    char nm[96], note[96];
    snprintf(nm, sizeof nm, xr ? "gemm_pwln_kernel<%d, %d, %d, 1>" : "gemm_pwln_kernel<%d, %d, %d>", wmb, nt, g.f16);
    snprintf(note, sizeof note, "B=%d M=%d N=%d K=%d k=1 +LN", g.B, g.M, g.N, g.Kc);
    ssv_shape_log(nm, grid, dim3(512), 2.0 * g.B * g.M * g.N * g.Kc, 4.0 * ((double)g.B * g.Kc * g.N + 2.0 * g.B * g.M * g.N + (double)g.M * g.Kc), note);
  }
#define SSV_PW(W_) if (wmb == W_) { \
    if (g.f16) hipLaunchKernelGGL((gemm_pwln_kernel<W_, 4, 1>), grid, dim3(512), 0, st, q); \
    else hipLaunchKernelGGL((gemm_pwln_kernel<W_, 4, 0>), grid, dim3(512), 0, st, q); \
    return ssv_check_launch("gemm_pwln"); }
  if (xr) {
    if (g.f16) hipLaunchKernelGGL((gemm_pwln_kernel<4, 4, 1, 1>), grid, dim3(512), 0, st, q);
    else hipLaunchKernelGGL((gemm_pwln_kernel<4, 4, 0, 1>), grid, dim3(512), 0, st, q);
    return ssv_check_launch("gemm_pwln (extra row)");
  }
  SSV_PW(1) SSV_PW(2) SSV_PW(3) SSV_PW(4) SSV_PW(5)
#undef SSV_PW
  return ssv_fail(SSV_UNSUPPORTED, "gemm_pwln: no instantiation for %d rows", g.M);
}

// ---- backward of a 1x1 conv + LayerNorm link in ONE launch (round 5): LayerNorm / activation backward, then dX = W^T dPre ------------------
// models/TTSModel.py:128-131, :173-180, :218-231, :343-361 backward.  Until now two launches per link: ln_act_bwd* (dY, pre -> dPre, parameter
// partials, scale list) and the k = 1 data-gradient GEMM re-reading dPre.  As in the forward (gemm_pwln_kernel) a workgroup owns ALL LN rows of a
// 64-column tile, so the LayerNorm backward's two column sums are local.  Phase 1 (512 threads = 32 row groups x 16 column quads; a thread holds
// 8 consecutive rows x 4 columns per unit): dPre from registers -> global (16-byte row pieces) AND, split with the TILE's own power-of-two scale,
// into an LDS image of the GEMM's input operand for every K chunk at once ([chunk][k-group][column][8 halves]: 8 rows of a column = one
// 16-byte slot).  Phase 2: the K loop runs with no staging and no barrier -- transposed weight fragments L2 -> registers (one set, re-loaded row
// block by row block as in gemm_pwln_kernel), input fragments from the image.  Phase 3: the dX tile parked in the image's memory, row-contiguous
// stores.  M = 128 j + 1 LN rows (513): the last row beside the row groups (threads 0 .. 15); Cin = 128 j + 1 output rows: the last one as fp32
// dot products of its weights with the dPre values the threads hold.
// Partial parameter-gradient rows [dgamma | dbeta | dbias] and the scale list keep the layout of the unfused kernels (ssv_ln_act_bwd_rows /
// ssv_amax_rows): this tile's row at part_q * tile, the rows up to the next tile's zeroed; scale entry 4 * tile, the next three zeroed.
#ifndef SSV_PWLN_BWD_FUSED
#define SSV_PWLN_BWD_FUSED 1
#endif
template <int WMB, int NU, int F16>
__global__ __launch_bounds__(512, 2) void pwln_bwd_kernel(const PwLnBw q) {
  constexpr int BN = 64, NT = 4;
  constexpr int NCH = NU == 1 ? 8 : 17;                 // K chunks (32 LN rows each) the image holds
  constexpr int LDWP = BN + 4;
  constexpr int IMG_U4 = NCH * 512, PARK_U4 = 8 * 16 * LDWP / 4;
  __shared__ uint4 img[IMG_U4 > PARK_U4 ? IMG_U4 : PARK_U4];     // per chunk: hi [k-group][column] (256 slots), then lo (256 slots)
  __shared__ float red[33 * 2 * BN];                    // column-sum partials of the 32 row groups (+ the extra LN row)
  __shared__ float tot[2 * BN];
  __shared__ float amx[8];
  __shared__ float xw[NU == 2 ? 544 : 1];               // weights of the extra OUTPUT row, one per LN row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned wg = ssv_xcd_order(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  const int ntile = (int)(wg % gridDim.x), b = (int)(wg / gridDim.x);
  const int n0 = ntile * BN;
  const int M = q.M, L = q.L, act = q.act;
  const int nch = (M + 31) >> 5;
  const bool xlr = (M & 7) == 1;                        // LN row M - 1 beside the row groups (M = 513)
  const int Mg = xlr ? M - 1 : M;                       // rows in the row groups
  const bool xo = NU == 2 && q.xrow_w != nullptr;       // output row Cin - 1 beside the MFMA row blocks
  const int Mt = xo ? q.Cin - 1 : q.Cin;

  // ---------------------------------------------------------------- phase 1: LayerNorm / activation backward
  const int cq = tid & 15, rgt = tid >> 4;
  const int t = n0 + 4 * cq;
  bool cv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) cv[j] = t + j < L;
  const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q.dy + (long)b * q.dy_bs), 0, M * L * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rpr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q.pre + (long)b * M * L), 0, M * L * 4, 0x00020000);
  auto ld4 = [&](__amdgpu_buffer_rsrc_t r, unsigned off, float (&v)[4]) __attribute__((always_inline)) {
    const f32x4 u = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
    v[0] = u[0]; v[1] = u[1]; v[2] = u[2]; v[3] = u[3];
  };
  float a[NU][8][4], xh[NU][8][4];                      // raw: dy, pre -> a = dn * gamma, xh; then a = dPre
  float ea[4] = {0.f, 0.f, 0.f, 0.f}, exh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int row = min((rgt + 32 * u) * 8 + r, Mg - 1);
      const unsigned o = ((unsigned)row * (unsigned)L + (unsigned)t) * 4u;
      ld4(rdy, o, a[u][r]); ld4(rpr, o, xh[u][r]);
    }
  const bool exrow = xlr && tid < 16;
  if (exrow) { const unsigned o = ((unsigned)(M - 1) * (unsigned)L + (unsigned)t) * 4u; ld4(rdy, o, ea); ld4(rpr, o, exh); }
  if constexpr (NU == 2) {
    if (xo) for (int k = tid; k < nch * 32; k += 512) xw[k] = k < M ? q.xrow_w[(long)k * q.xrow_sk] : 0.f;     // (visible after the first barrier)
  }
  float mu[4], rs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { const float* sb = q.stats + (long)b * 2 * L + min(t + j, L - 1); mu[j] = sb[0]; rs[j] = sb[L]; }
  float* pblk = q.part + ((long)b * q.part_rows + (long)q.part_q * ntile) * 3 * M;
  float sa[4] = {0.f, 0.f, 0.f, 0.f}, sah[4] = {0.f, 0.f, 0.f, 0.f};
  // one row of 4 columns: dn and the row's (dgamma, dbeta) partials; a <- dn * gamma, xh <- normalised input
  auto row_a = [&](float (&av)[4], float (&hv)[4], int row, bool rok) __attribute__((always_inline)) {
    const int rc = min(row, M - 1);
    const float gg0 = q.gamma[rc], bb = q.beta[rc];
    float q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool v = rok && cv[j];
      const float dy = v ? av[j] : 0.f, gg = v ? gg0 : 0.f;
      const float h = v ? (hv[j] - mu[j]) * rs[j] : 0.f;
      const float n = h * gg + bb;
      float dn;
      if (act == 1) dn = n > 0.f ? dy : 0.f;
      else if (act == 2) { const float s = 1.f / (1.f + __expf(-n)); dn = dy * s * (1.f - s); }
      else dn = dy;
      q0 += dn * h; q1 += dn;
      hv[j] = h; av[j] = dn * gg;
      sa[j] += av[j]; sah[j] += av[j] * h;
    }
    q0 = ssv_row16_sum(q0); q1 = ssv_row16_sum(q1);
    if (cq == 0 && rok) { pblk[rc] = q0; pblk[M + rc] = q1; }
  };
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int r = 0; r < 8; ++r) { const int row = (rgt + 32 * u) * 8 + r; row_a(a[u][r], xh[u][r], row, row < Mg); }
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[(rgt * 2 + 0) * BN + 4 * cq + j] = sa[j]; red[(rgt * 2 + 1) * BN + 4 * cq + j] = sah[j]; }
  if (xlr) {                                            // the extra LN row: threads 0 .. 15 (whose row groups are already summed above)
    if (tid < 16) {
#pragma unroll
      for (int j = 0; j < 4; ++j) sa[j] = sah[j] = 0.f;
      row_a(ea, exh, M - 1, true);
#pragma unroll
      for (int j = 0; j < 4; ++j) { red[(32 * 2 + 0) * BN + 4 * cq + j] = sa[j]; red[(32 * 2 + 1) * BN + 4 * cq + j] = sah[j]; }
    }
  }
  __syncthreads();
  if (tid < 2 * BN) {
    float sum = 0.f;
#pragma unroll 8
    for (int k = 0; k < 32; ++k) sum += red[k * 2 * BN + tid];
    if (xlr) sum += red[32 * 2 * BN + tid];
    tot[tid] = sum;
  }
  __syncthreads();
  const float invM = 1.f / (float)M;
  float m[4], mh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { m[j] = tot[4 * cq + j] * invM; mh[j] = tot[BN + 4 * cq + j] * invM; }
  float am = 0.f;
  float xacc[4] = {0.f, 0.f, 0.f, 0.f};
  float* __restrict__ dPb = q.dpre + (long)b * M * L;
  // a <- dPre = rstd (a - mean(a) - xh mean(a xh)); stored; the row's dbias partial; the extra output row's dot product
  auto row_d = [&](float (&av)[4], float (&hv)[4], int row, bool rok) __attribute__((always_inline)) {
    float q0 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float d = (rok && cv[j]) ? rs[j] * (av[j] - m[j] - hv[j] * mh[j]) : 0.f;
      av[j] = d;
      am = fmaxf(am, fabsf(d));
      q0 += d;
    }
    if (rok) {
      float* o = dPb + (long)row * L + t;
      if (cv[3]) { f4u w = {av[0], av[1], av[2], av[3]}; *reinterpret_cast<f4u*>(o) = w; }
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (cv[j]) o[j] = av[j];
      }
    }
    if constexpr (NU == 2) {
      if (xo) { const float wv = xw[min(row, NCH * 32 - 1)];
#pragma unroll
        for (int j = 0; j < 4; ++j) xacc[j] = __builtin_fmaf(wv, av[j], xacc[j]); }
    }
    q0 = ssv_row16_sum(q0);
    if (cq == 0 && rok) pblk[2 * M + min(row, M - 1)] = q0;
  };
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int r = 0; r < 8; ++r) { const int row = (rgt + 32 * u) * 8 + r; row_d(a[u][r], xh[u][r], row, row < Mg); }
  if (exrow) row_d(ea, exh, M - 1, true);
  // the tile's operand scale; its entry of the item's scale list; the partial rows between this tile's and the next one's
  am = ssv_wg_max<8>(am, amx);
  float xs = 1.f, xinv = 1.f;
  if constexpr (F16) { float sc, inv; ssv_pow2_scale(am, sc, inv); xs = ssv_uniform(sc); xinv = ssv_uniform(inv); }
  if (q.amax && tid < 4) { const int e = 4 * ntile + tid; if (e < q.namax) q.amax[(long)b * q.namax + e] = tid == 0 ? am : 0.f; }
  {
    const int r0 = q.part_q * ntile + 1, r1 = min(q.part_q * (ntile + 1), q.part_rows);
    float* z = q.part + ((long)b * q.part_rows + r0) * 3 * M;
    for (int e = tid; e < (r1 - r0) * 3 * M; e += 512) z[e] = 0.f;
  }
  // the image: 8 rows of one column -> one 16-byte slot of the hi plane and one of the lo plane
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int rb = (rgt + 32 * u) * 8, ch = rb >> 5, kg = (rb >> 3) & 3;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = a[u][r][j];
      uint4 h, l;
      split8s<F16>(v, xs, h, l);
      const int sl = ch * 512 + kg * 64 + 4 * cq + j;
      img[sl] = h; img[sl + 256] = l;
    }
  }
  if (xlr) {                                            // chunk nch - 1 holds the extra row alone: k-group 0, element 0; the rest zero
    const int base = (nch - 1) * 512;
    if (tid < 16) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[8] = {ea[j], 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        uint4 h, l;
        split8s<F16>(v, xs, h, l);
        img[base + 4 * cq + j] = h; img[base + 256 + 4 * cq + j] = l;
      }
    } else if (tid >= 64 && tid < 64 + 192) {           // k-groups 1 .. 3 of that chunk
      const int sl = base + 64 + (tid - 64);
      img[sl] = make_uint4(0, 0, 0, 0); img[sl + 256] = make_uint4(0, 0, 0, 0);
    }
  }
  if constexpr (NU == 2) {
    if (xo) {                                           // the extra output row: sum of the row groups' partial dot products
#pragma unroll
      for (int j = 0; j < 4; ++j) red[rgt * BN + 4 * cq + j] = xacc[j];          // (the extra LN row's term is in threads 0 .. 15's xacc: row_d added it)
    }
  }
  __syncthreads();
  if constexpr (NU == 2) {
    if (xo && tid < BN && n0 + tid < L) {
      float sum = 0.f;
#pragma unroll 8
      for (int k = 0; k < 32; ++k) sum += red[k * BN + tid];
      q.dx[(long)b * q.dx_bs + (long)(q.Cin - 1) * L + n0 + tid] = sum;
    }
  }

  // ---------------------------------------------------------------- phase 2: dX tile = W^T dPre, K = the LN rows, straight from the image
  const int kq = lane >> 4, nq = lane & 15;
  f32x4 acc[WMB][NT];
#pragma unroll
  for (int i = 0; i < WMB; ++i)
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) acc[i][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  uint4 Ah_[WMB], Al_[WMB];
  const int MB = (q.Cin + 15) >> 4;
  unsigned arowb[WMB];
#pragma unroll
  for (int i = 0; i < WMB; ++i) arowb[i] = (unsigned)(((long)min(wave * WMB + i, MB - 1) * nch * 512 + lane * 8) * 2);
  const __amdgpu_buffer_rsrc_t rsAh = ssv_buf(q.Ahi), rsAl = ssv_buf(q.Alo);
#pragma unroll
  for (int i = 0; i < WMB; ++i) { Ah_[i] = ssv_buf_u4(rsAh, arowb[i], 0u); Al_[i] = ssv_buf_u4(rsAl, arowb[i], 0u); }
  for (int ch = 0; ch < nch; ++ch) {
    const uint4* Xh = img + ch * 512;
    const uint4* Xl = Xh + 256;
    uint4 bh[NT], bl[NT];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) { const int s_ = kq * BN + tt * 16 + nq; bh[tt] = Xh[s_]; bl[tt] = Xl[s_]; }
    const unsigned ubn = (unsigned)__builtin_amdgcn_readfirstlane(min(ch + 1, nch - 1) * 1024);
#pragma unroll
    for (int i = 0; i < WMB; ++i) {
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        acc[i][tt] = mma16<F16>(Al_[i], bh[tt], acc[i][tt]);
        acc[i][tt] = mma16<F16>(Ah_[i], bl[tt], acc[i][tt]);
        acc[i][tt] = mma16<F16>(Ah_[i], bh[tt], acc[i][tt]);
      }
      __builtin_amdgcn_sched_barrier(0);
      Ah_[i] = ssv_buf_u4(rsAh, arowb[i], ubn);
      Al_[i] = ssv_buf_u4(rsAl, arowb[i], ubn);
    }
  }
  // ---------------------------------------------------------------- phase 3: dX rows, parked per wave, row-contiguous stores
  __syncthreads();                                      // every wave is done with the image
  const float us = F16 ? ssv_uniform(xinv * *q.a_inv) : 1.f;
  float* pk = reinterpret_cast<float*>(img) + wave * 16 * LDWP;
  float* __restrict__ Xo = q.dx + (long)b * q.dx_bs;
#pragma unroll
  for (int i = 0; i < WMB; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) pk[(kq * 4 + r) * LDWP + tt * 16 + nq] = F16 ? acc[i][tt][r] * us : acc[i][tt][r];
    const int rbase = (wave * WMB + i) * 16;
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      const int e = lane + 64 * it;
      const int row = e / (BN / 4), c4 = e % (BN / 4);
      const int gm = rbase + row, gn = n0 + c4 * 4;
      const f32x4 v = *reinterpret_cast<const f32x4*>(pk + row * LDWP + c4 * 4);
      if (gm < Mt && gn < L) {
        float* o = Xo + (long)gm * L + gn;
        if (gn + 3 < L) { f4u w = {v[0], v[1], v[2], v[3]}; *reinterpret_cast<f4u*>(o) = w; }
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (gn + j < L) o[j] = v[j];
        }
      }
    }
  }
}
// true when the link's backward can take the one-launch kernel (split-fp16 / split-bf16 planes of the TRANSPOSED weight given; dense dPre)
bool ssv_pwln_bwd_fused_ok(int B, int Cin, int Cout, int L) {
  if (!SSV_PWLN_BWD_FUSED) return false;
  // Measured in-step at B = 32 (round 5): links with up to 256 LN rows 24.3 us against 17.9 + 22.6 us in two launches; with 512 / 513 LN rows
  // (one 155 KB workgroup per CU: its three phases cannot overlap with anything) 139 against 74 + 95 us for 513 -> 513 but 139 against 74 + 67 for
  // 512 -> 513 and 123 against ~110 for 256 -> 512: no gain over the step.  Default: the small form only.  SSV_PWLN_BWD=0: never; =2: every shape.
  int mode = 1;
  if (const char* e = ssv_tuning(SSV_T_PWLN_BWD)) mode = atoi(e);
  if (mode == 0 || (mode != 2 && Cout > 256)) return false;
  const bool m_ok = (Cout % 8 == 0 && Cout >= 32 && Cout <= 512) || Cout == 513;
  const bool c_ok = (Cin >= 32 && Cin <= 512) || (Cin == 513 && Cout > 256);
  return m_ok && c_ok && B <= 65535 && L >= 16 && (long)Cout * L < (1L << 29) && (long)Cin * L < (1L << 29);
}
int ssv_launch_pwln_bwd(const PwLnBw& q, int B, int f16, hipStream_t st) {
  SSV_CHECK(ssv_pwln_bwd_fused_ok(B, q.Cin, q.M, q.L), SSV_UNSUPPORTED, "pwln_bwd: shape %d -> %d not supported", q.Cin, q.M);
  SSV_CHECK(q.dy && q.pre && q.stats && q.gamma && q.beta && q.dpre && q.part && q.Ahi && q.Alo && q.dx && (!f16 || q.a_inv), SSV_BAD_SHAPE, "pwln_bwd: null argument");
  SSV_CHECK(q.Cin % 128 != 1 || q.Cin < 128 || q.xrow_w, SSV_BAD_SHAPE, "pwln_bwd: %d output rows need the extra row's weights", q.Cin);
  const dim3 grid(ssv_cdiv(q.L, 64), B);
  SSV_CHECK(q.part_q >= 1 && q.part_rows >= q.part_q * ((int)grid.x - 1) + 1 && (!q.amax || q.namax >= 4 * (int)grid.x - 3), SSV_BAD_SHAPE, "pwln_bwd: partial rows / scale list too short");
  const int nu = q.M <= 256 ? 1 : 2;
  const int wmb = q.Cin <= 128 ? 1 : (q.Cin <= 256 ? 2 : 4);
  if (ssv_shape_log_on()) {
    char nm[96], note[96];
    snprintf(nm, sizeof nm, "pwln_bwd_kernel<%d, %d, %d>", wmb, nu, f16);
    snprintf(note, sizeof note, "B=%d Cin=%d N=%d Cout=%d LN bwd + k=1 data gradient", B, q.Cin, q.L, q.M);
    ssv_shape_log(nm, grid, dim3(512), 2.0 * B * q.Cin * q.L * q.M, 4.0 * ((double)B * q.M * q.L * 3 + (double)B * q.Cin * q.L + (double)q.M * q.Cin), note);
  }
#define SSV_PB(W_, U_) if (wmb == W_ && nu == U_) { \
    if (f16) hipLaunchKernelGGL((pwln_bwd_kernel<W_, U_, 1>), grid, dim3(512), 0, st, q); \
    else hipLaunchKernelGGL((pwln_bwd_kernel<W_, U_, 0>), grid, dim3(512), 0, st, q); \
    return ssv_check_launch("pwln_bwd"); }
  SSV_PB(1, 1) SSV_PB(2, 1) SSV_PB(4, 1) SSV_PB(1, 2) SSV_PB(2, 2) SSV_PB(4, 2)
#undef SSV_PB
  return ssv_fail(SSV_UNSUPPORTED, "pwln_bwd: no instantiation");
}
