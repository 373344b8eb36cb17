// Shared by the split-MFMA GEMM translation units (pack.hip, conv_nn.hip, wgrad_nt.hip, wgrad_nt3r.hip, pwln.hip): operand split helpers, the
// 16x16x32 MFMA wrapper, fragment types.  Until round 6 these kernels lived in one file, gemm_bf3.hip.
#pragma once
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

#include "bf3_tuning.h"
