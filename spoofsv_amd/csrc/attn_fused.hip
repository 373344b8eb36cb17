// Training attention in ONE launch per direction (round 5) -- models/TTSModel.py:266-270:
//   A = softmax_{text axis}(K^T Q / sqrt(d)),   R = V A,   decoder input = cat(R, Q)
// Until round 5: an exact-fp32 GEMM for the scores, a column-softmax kernel, a second GEMM for R and a row copy of Q -- four launches of
// ~1 GFLOP each, i.e. four launch floors (33 + 12 + 33 + 7 us; fused: 51 us.  Backward 33 + 19 + 33 -> 82 us: equal, two launches fewer).  Here a 4-wave workgroup owns a 64-column tile of one batch item: the score
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
// (chunks of 64 channels / 32 positions -- twice the MFMAs per staged chunk -- were measured slower: 56.6 / 85.3 us against 50.8 / 82.0)
constexpr int AF_KC = 32;                       // channels per staged chunk of the first product
constexpr int AF_NC = 16;                       // text positions per staged chunk of the second product
constexpr int AF_PV = 17;                       // row pitch of that chunk: lane (row, k) -> row * 17 + k covers 64 distinct banks

// First product of either direction: P(n, t) = sum_c L(c, n) X(c, t) for the workgroup's 64 columns, into acc[NB][4] (wave w: row blocks w, w + 4, ..).
// L is K (scores) or V (dA), (d, N) row-major; X is Q or dR, (d, T) row-major.  Both operands go through LDS in 32-channel chunks (coalesced
// loads, next chunk's loads in flight during this chunk's MFMAs); `stage` aliases the memory the second product later uses.
// copy_dst != null (forward): the X tile is also written to copy_dst(c, t) -- the decoder input's second half.
template <int NB>
__device__ __forceinline__ void attn_first_product(f32x4a (&acc)[NB][4], const float* __restrict__ Lb, const float* __restrict__ Xb, float* __restrict__ copy_dst,
                                                   float* stage, int d, int N, int T, int t0) {
  constexpr int NP = NB * 64, PK = NP + 16;      // staged L row: NP positions, pitch = NP + 16 (k-group rows 16 banks apart)
  constexpr int NL = (AF_KC * NP) / 256, NQ = (AF_KC * AF_BN) / 256;
  float* sL = stage;                             // [AF_KC][PK]
  float* sX = stage + AF_KC * PK;                // [AF_KC][AF_PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, nq = lane & 15;
  float rl[NL], rq[NQ];
  auto issue = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const int idx = tid + 256 * j, c = idx / NP, n = idx % NP;
      rl[j] = Lb[(long)(c0 + c) * N + min(n, N - 1)];
    }
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int idx = tid + 256 * j, c = idx / AF_BN, t = idx % AF_BN;
      rq[j] = Xb[(long)(c0 + c) * T + min(t0 + t, T - 1)];
    }
  };
  auto commit = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NL; ++j) { const int idx = tid + 256 * j, c = idx / NP, n = idx % NP; sL[c * PK + n] = n < N ? rl[j] : 0.f; }
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int idx = tid + 256 * j, c = idx / AF_BN, t = idx % AF_BN;
      sX[c * AF_PITCH + t] = rq[j];
      if (copy_dst && t0 + t < T) copy_dst[(long)(c0 + c) * T + t0 + t] = rq[j];
    }
  };
  issue(0);
  for (int c0 = 0; c0 < d; c0 += AF_KC) {
    __syncthreads();                             // the previous chunk's fragment reads are done
    commit(c0);
    __syncthreads();
    if (c0 + AF_KC < d) issue(c0 + AF_KC);
#pragma unroll
    for (int k0 = 0; k0 < AF_KC; k0 += 4) {
      float a[NB], q[4];
#pragma unroll
      for (int i = 0; i < NB; ++i) a[i] = sL[(k0 + kq) * PK + (wave + 4 * i) * 16 + nq];
#pragma unroll
      for (int t = 0; t < 4; ++t) q[t] = sX[(k0 + kq) * AF_PITCH + t * 16 + nq];
#pragma unroll
      for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], q[t], acc[i][t], 0, 0, 0);
    }
  }
  __syncthreads();                               // the staging memory is free again
}

// Second product: O(c, t) = sum_n L(c, n) S(n, t), S parked in LDS as [n][AF_PITCH] (zero rows past N), L = V (R) or K (dQ), (d, N) row-major,
// staged in chunks of 16 text positions as [d rows][AF_PV].  Wave w owns output row blocks w * DB .. w * DB + DB - 1.
template <int DB>
__device__ __forceinline__ void attn_second_product(f32x4a (&rc)[DB][4], const float* __restrict__ Lb, const float* sS, float* sL, int d, int N) {
  constexpr int NV = (DB * 64 * AF_NC) / 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, nq = lane & 15;
  float rv[NV];
  auto issue = [&](int n0) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NV; ++j) { const int idx = tid + 256 * j, c = idx / AF_NC, nl = idx % AF_NC; rv[j] = Lb[(long)c * N + min(n0 + nl, N - 1)]; }
  };
  issue(0);
  for (int n0 = 0; n0 < N; n0 += AF_NC) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NV; ++j) { const int idx = tid + 256 * j, c = idx / AF_NC, nl = idx % AF_NC; sL[c * AF_PV + nl] = rv[j]; }      // (positions past N meet zero rows of S)
    __syncthreads();
    if (n0 + AF_NC < N) issue(n0 + AF_NC);
#pragma unroll
    for (int k0 = 0; k0 < AF_NC; k0 += 4) {
      float a[DB], bq[4];
#pragma unroll
      for (int i = 0; i < DB; ++i) a[i] = sL[((wave * DB + i) * 16 + nq) * AF_PV + k0 + kq];
#pragma unroll
      for (int t = 0; t < 4; ++t) bq[t] = sS[(n0 + k0 + kq) * AF_PITCH + t * 16 + nq];
#pragma unroll
      for (int i = 0; i < DB; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) rc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bq[t], rc[i][t], 0, 0, 0);
    }
  }
}

// rows 0 .. N - 1 of a tile parked as [n][AF_PITCH] -> dst(n, t0 ..): 16 bytes per lane along the row (rows of T floats are 4-byte aligned only)
typedef float f4ua __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ void store_tile_rows(float* __restrict__ dst, const float* tile, int N, int T, int t0) {
  for (int e = threadIdx.x; e < N * (AF_BN / 4); e += 256) {
    const int n = e / (AF_BN / 4), c4 = e % (AF_BN / 4), gt = t0 + 4 * c4;
    if (gt >= T) continue;
    const f32x4a v = *reinterpret_cast<const f32x4a*>(tile + n * AF_PITCH + 4 * c4);
    float* o = dst + (long)n * T + gt;
    if (gt + 3 < T) { f4ua w = {v[0], v[1], v[2], v[3]}; *reinterpret_cast<f4ua*>(o) = w; }
    else for (int j = 0; j < 4; ++j) if (gt + j < T) o[j] = v[j];
  }
}

// NB: score row blocks (16 text positions) per wave: ceil(ceil(N / 16) / 4); DB: output row blocks per wave of the d-row products (d / 64)
template <int NB, int DB>
__global__ __launch_bounds__(256) void attn_fwd_fused_kernel(const float* __restrict__ K, const float* __restrict__ V, long kv_bs, const float* __restrict__ Q, long q_bs,
                                                             float* __restrict__ A, float* __restrict__ RQ, long rq_bs, int copy_q, int d, int N, int T, float alpha) {
  constexpr int NP = NB * 64;
  constexpr int STAGE1 = AF_KC * (NP + 16) + AF_KC * AF_PITCH;
  constexpr int SA = NP * AF_PITCH;
  __shared__ float sA[SA > STAGE1 ? SA : STAGE1];                        // first: the first product's staging; then the attention tile
  __shared__ float sV[DB * 64 * AF_PV];
  __shared__ float red[4][AF_BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, nq = lane & 15;
  const int t0 = blockIdx.x * AF_BN, b = blockIdx.y;
  float* __restrict__ Ab = A + (long)b * N * T;
  float* __restrict__ Rb = RQ + (long)b * rq_bs;
  f32x4a acc[NB][4];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[i][t] = (f32x4a){0.f, 0.f, 0.f, 0.f};
  attn_first_product<NB>(acc, K + (long)b * kv_bs, Q + (long)b * q_bs, copy_q ? Rb + (long)d * T : nullptr, sA, d, N, T, t0);
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
        sA[n * AF_PITCH + t * 16 + nq] = acc[i][t][r] * sum[t];         // (0 for n >= N)
      }
    }
  __syncthreads();
  store_tile_rows(Ab, sA, N, T, t0);                                     // A, 256-byte row pieces per 16 lanes
  // ---- R(c, t) = sum_n V(c, n) A(n, t)
  f32x4a rc[DB][4];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) rc[i][t] = (f32x4a){0.f, 0.f, 0.f, 0.f};
  attn_second_product<DB>(rc, V + (long)b * kv_bs, sA, sV, d, N);
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

// dS (written to dS for the dK product) and dQ.  dR: (B, d, T) with batch stride dr_bs; dA_ext may be null; dq_add may be null.
template <int NB, int DB>
__global__ __launch_bounds__(256) void attn_bwd_fused_kernel(const float* __restrict__ dR, long dr_bs, const float* __restrict__ dAext, const float* __restrict__ dQadd,
                                                             long dqa_bs, const float* __restrict__ K, const float* __restrict__ V, long kv_bs,
                                                             const float* __restrict__ A, float* __restrict__ dS, float* __restrict__ dQ, long dq_bs,
                                                             int d, int N, int T, float alpha) {
  constexpr int NP = NB * 64;
  constexpr int STAGE1 = AF_KC * (NP + 16) + AF_KC * AF_PITCH;
  constexpr int SA = NP * AF_PITCH;
  __shared__ float sS[SA > STAGE1 ? SA : STAGE1];
  __shared__ float sV[DB * 64 * AF_PV];
  __shared__ float red[4][AF_BN];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kq = lane >> 4, nq = lane & 15;
  const int t0 = blockIdx.x * AF_BN, b = blockIdx.y;
  const float* __restrict__ Ab = A + (long)b * N * T;
  const float* __restrict__ Eb = dAext ? dAext + (long)b * N * T : nullptr;
  float* __restrict__ dSb = dS + (long)b * N * T;
  // ---- dA(n, t) = sum_c V(c, n) dR(c, t)
  f32x4a acc[NB][4];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[i][t] = (f32x4a){0.f, 0.f, 0.f, 0.f};
  attn_first_product<NB>(acc, V + (long)b * kv_bs, dR + (long)b * dr_bs, nullptr, sS, d, N, T, t0);
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
        sS[n * AF_PITCH + t * 16 + nq] = av[i][t][r] * (acc[i][t][r] - dot[t]) * alpha;      // (0 where A was masked to 0)
      }
    }
  __syncthreads();
  store_tile_rows(dSb, sS, N, T, t0);
  // ---- dQ(c, t) = sum_n K(c, n) dS(n, t) + dQ_add(c, t)
  f32x4a rc[DB][4];
#pragma unroll
  for (int i = 0; i < DB; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t) rc[i][t] = (f32x4a){0.f, 0.f, 0.f, 0.f};
  attn_second_product<DB>(rc, K + (long)b * kv_bs, sS, sV, d, N);
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
