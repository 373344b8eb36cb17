// Training attention in ONE launch per direction (round 5) -- models/TTSModel.py:266-270:
//   A = softmax_{text axis}(K^T Q / sqrt(d)),   R = V A,   decoder input = cat(R, Q)
// Until round 5: an exact-fp32 GEMM for the scores, a column-softmax kernel, a second GEMM for R and a row copy of Q -- four launches of
// ~1 GFLOP each, i.e. four launch floors (33 + 12 + 33 + 7 us).  Here a 4-wave workgroup owns a 64-column tile of one batch item: the score
// tile (N <= 192 text positions x 64 frames) lives in MFMA accumulators (v_mfma_f32_16x16x4_f32: the same exact-fp32 arithmetic as before, so
// the attention matrix keeps its 2e-5 bar against the oracle), the column softmax runs on the accumulators (lane exchanges + one LDS round
// per reduction), A is written once and parked in LDS, R = V A reads it from there, and the Q tile every wave loads for the scores is
// written into the second half of the decoder's input on the way.
// Backward the same way: dA = V^T dR in accumulators, dS = A (dA + dA_ext - sum_n A (dA + dA_ext)) / sqrt(d) written once (the two
// reductions over time, dK = Q dS^T and dV = dR A^T, stay on the weight-gradient kernel) and parked in LDS, dQ = K dS + dQ_add from there.
#include <math.h>
#include "ssv_common.h"
#include "../../include/ssv_hip.h"

typedef float f32x4a __attribute__((ext_vector_type(4)));
constexpr int AF_BN = 64, AF_PITCH = 80;        // columns per tile; LDS row pitch (the four k-groups of a B-fragment read hit disjoint banks)

// NB: score row blocks (16 text positions) per wave: ceil(ceil(N / 16) / 4); DB: output row blocks per wave of the d-row products (d / 64)
template <int NB, int DB>
__global__ __launch_bounds__(256) void attn_fwd_fused_kernel(const float* __restrict__ K, const float* __restrict__ V, long kv_bs, const float* __restrict__ Q, long q_bs,
                                                             float* __restrict__ A, float* __restrict__ RQ, long rq_bs, int copy_q, int d, int N, int T, float alpha) {
  __shared__ float sA[NB * 64 * AF_PITCH];
  __shared__ float red[4][AF_BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, nq = lane & 15;
  const int t0 = blockIdx.x * AF_BN, b = blockIdx.y;
  const float* __restrict__ Kb = K + (long)b * kv_bs;
  const float* __restrict__ Vb = V + (long)b * kv_bs;
  const float* __restrict__ Qb = Q + (long)b * q_bs;
  float* __restrict__ Ab = A + (long)b * N * T;
  float* __restrict__ Rb = RQ + (long)b * rq_bs;
  // ---- scores: S(n, t) = sum_c K(c, n) Q(c, t)
  f32x4a acc[NB][4];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[i][t] = (f32x4a){0.f, 0.f, 0.f, 0.f};
  int nrow[NB], tcol[4];
#pragma unroll
  for (int i = 0; i < NB; ++i) nrow[i] = min((wave + 4 * i) * 16 + nq, N - 1);
#pragma unroll
  for (int t = 0; t < 4; ++t) tcol[t] = min(t0 + t * 16 + nq, T - 1);
#pragma unroll 4
  for (int k0 = 0; k0 < d; k0 += 4) {
    const int c = k0 + kq;
    float a[NB], q[4];
#pragma unroll
    for (int i = 0; i < NB; ++i) a[i] = Kb[(long)c * N + nrow[i]];
#pragma unroll
    for (int t = 0; t < 4; ++t) q[t] = Qb[(long)c * T + tcol[t]];
    if (copy_q && wave == 0) {                  // cat(R, Q): every wave holds the whole Q tile once over the loop; wave 0 files it
#pragma unroll
      for (int t = 0; t < 4; ++t) if (t0 + t * 16 + nq < T) Rb[(long)(d + c) * T + t0 + t * 16 + nq] = q[t];
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], q[t], acc[i][t], 0, 0, 0);
  }
  // ---- column softmax over the N rows: this lane holds rows (wave + 4 i) * 16 + kq * 4 + r of columns t * 16 + nq
  float mx[4], sum[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = (wave + 4 * i) * 16 + kq * 4 + r;
        const float v = n < N ? acc[i][t][r] * alpha : -INFINITY;
        acc[i][t][r] = v;
        m = fmaxf(m, v);
      }
    m = fmaxf(m, __shfl_xor(m, 16));
    m = fmaxf(m, __shfl_xor(m, 32));
    mx[t] = m;
    if (kq == 0) red[wave][t * 16 + nq] = m;
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 4; ++t) mx[t] = fmaxf(fmaxf(red[0][t * 16 + nq], red[1][t * 16 + nq]), fmaxf(red[2][t * 16 + nq], red[3][t * 16 + nq]));
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float e = expf(acc[i][t][r] - mx[t]); acc[i][t][r] = e; s += e; }       // (exp(-inf) = 0 for the rows past N)
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (kq == 0) red[wave][t * 16 + nq] = s;
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 4; ++t) sum[t] = 1.f / ((red[0][t * 16 + nq] + red[1][t * 16 + nq]) + (red[2][t * 16 + nq] + red[3][t * 16 + nq]));
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = (wave + 4 * i) * 16 + kq * 4 + r;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float p = acc[i][t][r] * sum[t];
        sA[n * AF_PITCH + t * 16 + nq] = p;                              // (0 for n >= N)
        const int gt = t0 + t * 16 + nq;
        if (n < N && gt < T) Ab[(long)n * T + gt] = p;
      }
    }
  __syncthreads();
  // ---- R(c, t) = sum_n V(c, n) A(n, t): wave w owns rows w * 16 DB .. of the d output rows
  f32x4a rc[DB][4];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) rc[i][t] = (f32x4a){0.f, 0.f, 0.f, 0.f};
  const int Npad = NB * 64;
#pragma unroll 4
  for (int k0 = 0; k0 < Npad; k0 += 4) {
    const int n = k0 + kq;
    if (k0 >= N) break;                                                  // (rows past N are zero; uniform)
    float a[DB], bq[4];
#pragma unroll
    for (int i = 0; i < DB; ++i) a[i] = Vb[(long)((wave * DB + i) * 16 + nq) * N + min(n, N - 1)];
#pragma unroll
    for (int t = 0; t < 4; ++t) bq[t] = sA[n * AF_PITCH + t * 16 + nq];
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
      for (int t = 0; t < 4; ++t) rc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bq[t], rc[i][t], 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = (wave * DB + i) * 16 + kq * 4 + r;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int gt = t0 + t * 16 + nq;
        if (gt < T) Rb[(long)c * T + gt] = rc[i][t][r];
      }
    }
}

// dS (in place of nothing: written to dS), dQ.  dR: (B, d, T) with batch stride dr_bs; dA_ext may be null; dq_add may be null.
template <int NB, int DB>
__global__ __launch_bounds__(256) void attn_bwd_fused_kernel(const float* __restrict__ dR, long dr_bs, const float* __restrict__ dAext, const float* __restrict__ dQadd,
                                                             long dqa_bs, const float* __restrict__ K, const float* __restrict__ V, long kv_bs,
                                                             const float* __restrict__ A, float* __restrict__ dS, float* __restrict__ dQ, long dq_bs,
                                                             int d, int N, int T, float alpha) {
  __shared__ float sS[NB * 64 * AF_PITCH];
  __shared__ float red[4][AF_BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, nq = lane & 15;
  const int t0 = blockIdx.x * AF_BN, b = blockIdx.y;
  const float* __restrict__ Kb = K + (long)b * kv_bs;
  const float* __restrict__ Vb = V + (long)b * kv_bs;
  const float* __restrict__ dRb = dR + (long)b * dr_bs;
  const float* __restrict__ Ab = A + (long)b * N * T;
  const float* __restrict__ Eb = dAext ? dAext + (long)b * N * T : nullptr;
  float* __restrict__ dSb = dS + (long)b * N * T;
  // ---- dA(n, t) = sum_c V(c, n) dR(c, t)
  f32x4a acc[NB][4];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[i][t] = (f32x4a){0.f, 0.f, 0.f, 0.f};
  int nrow[NB], tcol[4];
#pragma unroll
  for (int i = 0; i < NB; ++i) nrow[i] = min((wave + 4 * i) * 16 + nq, N - 1);
#pragma unroll
  for (int t = 0; t < 4; ++t) tcol[t] = min(t0 + t * 16 + nq, T - 1);
#pragma unroll 4
  for (int k0 = 0; k0 < d; k0 += 4) {
    const int c = k0 + kq;
    float a[NB], q[4];
#pragma unroll
    for (int i = 0; i < NB; ++i) a[i] = Vb[(long)c * N + nrow[i]];
#pragma unroll
    for (int t = 0; t < 4; ++t) q[t] = dRb[(long)c * T + tcol[t]];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], q[t], acc[i][t], 0, 0, 0);
  }
  // ---- dS = A (g - sum_n A g) alpha,  g = dA + dA_ext
  f32x4a av[NB][4];
  float dot[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    float s = 0.f;
    const int gt = t0 + t * 16 + nq;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = (wave + 4 * i) * 16 + kq * 4 + r;
        const bool v = n < N && gt < T;
        const float p = v ? Ab[(long)n * T + gt] : 0.f;
        float g = acc[i][t][r];
        if (Eb && v) g += Eb[(long)n * T + gt];
        av[i][t][r] = p; acc[i][t][r] = g;
        s += p * g;
      }
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (kq == 0) red[wave][t * 16 + nq] = s;
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < 4; ++t) dot[t] = (red[0][t * 16 + nq] + red[1][t * 16 + nq]) + (red[2][t * 16 + nq] + red[3][t * 16 + nq]);
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = (wave + 4 * i) * 16 + kq * 4 + r;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float ds = av[i][t][r] * (acc[i][t][r] - dot[t]) * alpha;      // (0 where A was masked to 0)
        sS[n * AF_PITCH + t * 16 + nq] = ds;
        const int gt = t0 + t * 16 + nq;
        if (n < N && gt < T) dSb[(long)n * T + gt] = ds;
      }
    }
  __syncthreads();
  // ---- dQ(c, t) = sum_n K(c, n) dS(n, t) + dQ_add(c, t)
  f32x4a rc[DB][4];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) rc[i][t] = (f32x4a){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int k0 = 0; k0 < NB * 64; k0 += 4) {
    const int n = k0 + kq;
    if (k0 >= N) break;
    float a[DB], bq[4];
#pragma unroll
    for (int i = 0; i < DB; ++i) a[i] = Kb[(long)((wave * DB + i) * 16 + nq) * N + min(n, N - 1)];
#pragma unroll
    for (int t = 0; t < 4; ++t) bq[t] = sS[n * AF_PITCH + t * 16 + nq];
#pragma unroll
    for (int i = 0; i < DB; ++i)
#pragma unroll
      for (int t = 0; t < 4; ++t) rc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bq[t], rc[i][t], 0, 0, 0);
  }
  float* __restrict__ dQb = dQ + (long)b * dq_bs;
  const float* __restrict__ Db = dQadd ? dQadd + (long)b * dqa_bs : nullptr;
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = (wave * DB + i) * 16 + kq * 4 + r;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int gt = t0 + t * 16 + nq;
        if (gt < T) dQb[(long)c * T + gt] = rc[i][t][r] + (Db ? Db[(long)c * T + gt] : 0.f);
      }
    }
}

// the fused kernels cover the configured model (d = 256, N <= 192: MAX_TEXT_LEN = 186) and anything smaller in whole 64-row groups
bool ssv_attn_fused_ok(int B, int d, int N, int T) {
  return d % 64 == 0 && d <= 256 && N >= 1 && N <= 192 && T >= 1 && B <= 65535 && (long)N * T < (1L << 31) && (long)2 * d * T < (1L << 31);
}
int ssv_launch_attn_fwd_fused(const float* k, const float* v, long kv_bs, const float* q, long q_bs, float* a, float* rq, long rq_bs, int copy_q,
                              int B, int d, int N, int T, hipStream_t st) {
  const dim3 grid(ssv_cdiv(T, AF_BN), B);
  const float alpha = 1.f / sqrtf((float)d);
  const int nb = ssv_cdiv(ssv_cdiv(N, 16), 4), db = d / 64;
#define AF_CASE(NB_, DB_) if (nb == NB_ && db == DB_) { hipLaunchKernelGGL((attn_fwd_fused_kernel<NB_, DB_>), grid, dim3(256), 0, st, k, v, kv_bs, q, q_bs, a, rq, rq_bs, copy_q, d, N, T, alpha); return ssv_check_launch("attn_fwd_fused"); }
  AF_CASE(3, 4) AF_CASE(2, 4) AF_CASE(1, 4) AF_CASE(3, 2) AF_CASE(2, 2) AF_CASE(1, 2) AF_CASE(3, 1) AF_CASE(2, 1) AF_CASE(1, 1) AF_CASE(3, 3) AF_CASE(2, 3) AF_CASE(1, 3)
#undef AF_CASE
  return ssv_fail(SSV_UNSUPPORTED, "attn_fwd_fused: d = %d, N = %d", d, N);
}
int ssv_launch_attn_bwd_fused(const float* dr, long dr_bs, const float* da_ext, const float* dq_add, long dq_add_bs, const float* k, const float* v, long kv_bs,
                              const float* a, float* ds, float* dq, long dq_bs, int B, int d, int N, int T, hipStream_t st) {
  const dim3 grid(ssv_cdiv(T, AF_BN), B);
  const float alpha = 1.f / sqrtf((float)d);
  const int nb = ssv_cdiv(ssv_cdiv(N, 16), 4), db = d / 64;
#define AF_CASE(NB_, DB_) if (nb == NB_ && db == DB_) { hipLaunchKernelGGL((attn_bwd_fused_kernel<NB_, DB_>), grid, dim3(256), 0, st, dr, dr_bs, da_ext, dq_add, dq_add_bs, k, v, kv_bs, a, ds, dq, dq_bs, d, N, T, alpha); return ssv_check_launch("attn_bwd_fused"); }
  AF_CASE(3, 4) AF_CASE(2, 4) AF_CASE(1, 4) AF_CASE(3, 2) AF_CASE(2, 2) AF_CASE(1, 2) AF_CASE(3, 1) AF_CASE(2, 1) AF_CASE(1, 1) AF_CASE(3, 3) AF_CASE(2, 3) AF_CASE(1, 3)
#undef AF_CASE
  return ssv_fail(SSV_UNSUPPORTED, "attn_bwd_fused: d = %d, N = %d", d, N);
}
