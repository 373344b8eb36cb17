// GE2E speaker-embedder kernels (gfx950): layout change for the LSTM input, the fused LSTM cell
// update, projection normalisation and the GE2E loss.  The LSTM's matrix products (input projection
// for all frames at once, and the per-frame recurrent product) run on the shared implicit-GEMM kernel
// with activations kept as [hidden][batch] so that the batch is the contiguous, coalesced axis.
#include "ssv_common.h"
#include "../../include/ssv_hip.h"

// x (Bn, T, F) row-major  ->  xt [T][F][Bn]
__global__ __launch_bounds__(256) void lstm_in_transpose_kernel(const float* __restrict__ x, float* __restrict__ xt, int Bn, int T, int F) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;   // index into xt
  if (i >= (long)Bn * T * F) return;
  const int b = (int)(i % Bn);
  const int f = (int)((i / Bn) % F), t = (int)(i / ((long)Bn * F));
  xt[i] = x[((long)b * T + t) * F + f];
}

// x (Bn, T, F) row-major  ->  the frames pre-split for layer 0's product (GemmNNB::x0_planes): frame t = a (hi, lo) pair of planes [k-group of 8 features][npad
// columns][8 halves], `plane_bytes` apart, hi = fp16(x s), lo = fp16(x s - hi) with the power-of-two scale s of the whole input (amax: 64 partial maxima,
// ssv_pow2_scale as every split-fp16 operand).  Features past F and columns past Bn are zeros.  One thread per 16-byte slot, columns fastest.
__global__ __launch_bounds__(256) void lstm_x_planes_kernel(const float* __restrict__ x, const float* __restrict__ amax, char* __restrict__ planes,
                                                            long plane_bytes, int Bn, int T, int F, int kgroups, int npad) {
  float sc, inv;
  ssv_pow2_scale(ssv_wave_list_max(amax, 64), sc, inv);
  const int e = blockIdx.x * 256 + threadIdx.x, t = blockIdx.y;
  if (e >= kgroups * npad) return;
  const int kg = e / npad, col = e % npad;
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  h8 hi, lo;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int f = 8 * kg + i;
    const float v = (col < Bn && f < F) ? x[((long)col * T + t) * F + f] * sc : 0.f;
    const _Float16 h = (_Float16)v;
    hi[i] = h;
    lo[i] = (_Float16)(v - (float)h);
  }
  char* frame = planes + (long)t * 2 * plane_bytes;
  *reinterpret_cast<h8*>(frame + (long)e * 16) = hi;
  *reinterpret_cast<h8*>(frame + plane_bytes + (long)e * 16) = lo;
}
int ssv_launch_lstm_x_planes(const float* x, const float* amax, void* planes, long plane_bytes, int Bn, int T, int F, int kgroups, int npad, hipStream_t st) {
  hipLaunchKernelGGL(lstm_x_planes_kernel, dim3(ssv_cdiv((long)kgroups * npad, 256), T), dim3(256), 0, st, x, amax, (char*)planes, plane_bytes, Bn, T, F, kgroups, npad);
  return ssv_check_launch("lstm_x_planes");
}

__device__ __forceinline__ float sigm(float v) { return 1.f / (1.f + expf(-v)); }

// g: [4H][Bn] gate pre-activations in torch order i, f, g, o; c: [H][Bn] updated in place; h: [H][Bn].
__global__ __launch_bounds__(256) void lstm_cell_kernel(const float* __restrict__ g, float* __restrict__ c, float* __restrict__ h, int H, int Bn, int first) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long n = (long)H * Bn;
  if (i >= n) return;
  const float gi = sigm(g[i]), gf = sigm(g[n + i]), gg = tanhf(g[2 * n + i]), go = sigm(g[3 * n + i]);
  const float cn = (first ? 0.f : gf * c[i]) + gi * gg;
  c[i] = cn;
  h[i] = go * tanhf(cn);
}

// training (exact-fp32 path): pre [4H][Bn] gate pre-activations in torch order -> act [4H][Bn] activated i, f, g, o (may alias pre); c_t and h_t
// [H][Bn]; cprev null at t = 0
__global__ __launch_bounds__(256) void lstm_cell_train_kernel(const float* pre, float* act, const float* __restrict__ cprev, float* __restrict__ c,
                                                              float* __restrict__ h, int H, int Bn) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long n = (long)H * Bn;
  if (i >= n) return;
  const float gi = sigm(pre[i]), gf = sigm(pre[n + i]), gg = tanhf(pre[2 * n + i]), go = sigm(pre[3 * n + i]);
  const float cn = (cprev ? gf * cprev[i] : 0.f) + gi * gg;
  act[i] = gi; act[n + i] = gf; act[2 * n + i] = gg; act[3 * n + i] = go;
  c[i] = cn;
  h[i] = go * tanhf(cn);
}
int ssv_launch_lstm_cell_train(const float* pre, float* act, const float* cprev, float* c, float* h, int H, int Bn, hipStream_t st) {
  hipLaunchKernelGGL(lstm_cell_train_kernel, dim3(ssv_cdiv((long)H * Bn, 256)), dim3(256), 0, st, pre, act, cprev, c, h, H, Bn);
  return ssv_check_launch("lstm_cell_train");
}

// h_last [H][Bn] -> (Bn, H)
__global__ __launch_bounds__(256) void transpose_out_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int Bn) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;   // index into dst (Bn, R)
  if (i >= (long)R * Bn) return;
  const int r = (int)(i % R), b = (int)(i / R);
  dst[i] = src[(long)r * Bn + b];
}

int ssv_launch_lstm_in_transpose(const float* x, float* xt, int Bn, int T, int F, hipStream_t st) {
  hipLaunchKernelGGL(lstm_in_transpose_kernel, dim3(ssv_cdiv((long)Bn * T * F, 256)), dim3(256), 0, st, x, xt, Bn, T, F);
  return ssv_check_launch("lstm_in_transpose");
}
int ssv_launch_lstm_cell(const float* g, float* c, float* h, int H, int Bn, int first, hipStream_t st) {
  hipLaunchKernelGGL(lstm_cell_kernel, dim3(ssv_cdiv((long)H * Bn, 256)), dim3(256), 0, st, g, c, h, H, Bn, first);
  return ssv_check_launch("lstm_cell");
}
int ssv_launch_transpose_out(const float* src, float* dst, int R, int Bn, hipStream_t st) {
  hipLaunchKernelGGL(transpose_out_kernel, dim3(ssv_cdiv((long)R * Bn, 256)), dim3(256), 0, st, src, dst, R, Bn);
  return ssv_check_launch("transpose_out");
}

// y [P][Bn] -> e (Bn, P), each row divided by its L2 norm.  One wave per batch item.
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* __restrict__ y, float* __restrict__ e, float* __restrict__ norms, int P, int Bn) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= Bn) return;
  float s = 0.f;
  for (int p = lane; p < P; p += 64) { const float v = y[(long)p * Bn + b]; s += v * v; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float nrm = sqrtf(s);
  if (norms && lane == 0) norms[b] = nrm;
  for (int p = lane; p < P; p += 64) e[(long)b * P + p] = y[(long)p * Bn + b] / nrm;
}
int ssv_launch_l2norm_rows(const float* y, float* e, float* norms, int P, int Bn, hipStream_t st) {
  hipLaunchKernelGGL(l2norm_rows_kernel, dim3(ssv_cdiv(Bn, 4)), dim3(256), 0, st, y, e, norms, P, Bn);
  return ssv_check_launch("l2norm_rows");
}

// ---- GE2E loss -------------------------------------------------------------------------------------
// csum[k][d] = sum_m emb[k][m][d]
__global__ __launch_bounds__(256) void ge2e_centroid_kernel(const float* __restrict__ emb, float* __restrict__ csum, int N, int M, int D) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)N * D) return;
  const int k = (int)(i / D), d = (int)(i % D);
  float s = 0.f;
  for (int m = 0; m < M; ++m) s += emb[((long)k * M + m) * D + d];
  csum[i] = s;
}
// One workgroup per embedding (j, i): wave w takes the centroids w, w + 4, ..., its lanes the D components (coalesced rows of csum; two wave sums per
// centroid, no barrier in the loop).  (Until round 6's end one wave per embedding with lane = centroid, each lane walking a row of csum: 77 us at config 5.)
//   cos_k = <e, c_k> / max(|e||c_k|, 1e-8) + 1e-6, c_k = csum_k / M, or (csum_j - e) / (M-1) for k == j
//   per = -(S_j - log(sum_k exp(S_k) + 1e-6)), S = w cos + b
__global__ __launch_bounds__(256) void ge2e_rows_kernel(const float* __restrict__ emb, const float* __restrict__ csum, const float* __restrict__ w,
                                                         const float* __restrict__ bb, float* __restrict__ per, int N, int M, int D) {
  __shared__ float s_exp[4], s_pos[4];
  const int ji = blockIdx.x, j = ji / M, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* e = emb + (long)ji * D;
  float en = 0.f;
  for (int d = lane; d < D; d += 64) en += e[d] * e[d];
  en = sqrtf(ssv_wave_sum(en));
  const float ww = w[0], b0 = bb[0];
  float sexp = 0.f, spos = 0.f;
  for (int k = wave; k < N; k += 4) {
    const float* c = csum + (long)k * D;
    const float inv = (k == j) ? 1.f / (float)(M - 1) : 1.f / (float)M;
    float dot = 0.f, cn = 0.f;
    for (int d = lane; d < D; d += 64) {
      const float ev = e[d];
      const float cv = ((k == j) ? c[d] - ev : c[d]) * inv;
      dot += ev * cv; cn += cv * cv;
    }
    dot = ssv_wave_sum(dot); cn = ssv_wave_sum(cn);
    const float cosv = dot / fmaxf(en * sqrtf(cn), 1e-8f) + 1e-6f;
    const float S = ww * cosv + b0;
    sexp += expf(S);
    if (k == j) spos = S;
  }
  if (lane == 0) { s_exp[wave] = sexp; s_pos[wave] = spos; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float se = (s_exp[0] + s_exp[1]) + (s_exp[2] + s_exp[3]), sp = (s_pos[0] + s_pos[1]) + (s_pos[2] + s_pos[3]);
    per[ji] = -(sp - logf(se + 1e-6f));
  }
}
__global__ __launch_bounds__(256) void ge2e_total_kernel(const float* __restrict__ per, float* __restrict__ loss, int n) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += per[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

extern "C" size_t ssv_ge2e_loss_fwd_workspace(int N, int M, int D) { return ((size_t)N * D + (size_t)N * M) * sizeof(float); }
extern "C" int ssv_ge2e_loss_fwd(const float* emb, const float* w, const float* b, float* loss, float* per, int N, int M, int D,
                                 void* ws, size_t ws_bytes, ssv_stream_t stream) {
  SSV_CHECK(N > 0 && M > 1 && D > 0, SSV_BAD_SHAPE, "ge2e_loss_fwd: need N>0, M>1, D>0 (N=%d M=%d D=%d)", N, M, D);
  SSV_CHECK(ws && ws_bytes >= ssv_ge2e_loss_fwd_workspace(N, M, D), SSV_BAD_SHAPE, "ge2e_loss_fwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* csum = (float*)ws;
  float* perbuf = per ? per : csum + (size_t)N * D;
  hipLaunchKernelGGL(ge2e_centroid_kernel, dim3(ssv_cdiv((long)N * D, 256)), dim3(256), 0, st, emb, csum, N, M, D);
  SSV_TRY(ssv_check_launch("ge2e_centroid"));
  hipLaunchKernelGGL(ge2e_rows_kernel, dim3(N * M), dim3(256), 0, st, emb, csum, w, b, perbuf, N, M, D);
  SSV_TRY(ssv_check_launch("ge2e_rows"));
  hipLaunchKernelGGL(ge2e_total_kernel, dim3(1), dim3(256), 0, st, perbuf, loss, N * M);
  return ssv_check_launch("ge2e_total");
}

// ---- GE2E loss backward -------------------------------------------------------------------------------
// L = sum_ji -(S_ji,j - log(sum_k exp S_ji,k + 1e-6)),  S = w cos + b,  cos_ji,k = <e_ji, c_k> / (|e_ji||c_k|) + 1e-6,
// c_k = mean_m e_km, except k = j where the centroid leaves e_ji out (GE2E/utils.py:16-46).  With
//   G_ji,k = exp(S_ji,k) / (sum_k' exp S_ji,k' + 1e-6) - [k = j]       (= dL/dS)
// dw = sum G cos, db = sum G, and an embedding receives three kinds of terms: as the query of its own row (direct),
// through the centroids of the other speakers' rows (1/M each), through the leave-one-out centroids of its own speaker.
// The problem is tiny (880 x 88 x 256); the kernels are written for clarity and a fixed summation order.
__device__ __forceinline__ float block_sum_256(float v, float* red4) {      // sum over a 256-thread block, all threads get it
  v = ssv_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red4[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red4[0] + red4[1]) + (red4[2] + red4[3]);
}
#define GE2E_MAXN 1024
// one workgroup per embedding (j, i)
__global__ __launch_bounds__(256) void ge2e_bwd_rows_kernel(const float* __restrict__ emb, const float* __restrict__ csum, const float* __restrict__ wp,
                                                            const float* __restrict__ bp, const float* __restrict__ dloss,
                                                            float* __restrict__ Hc, float* __restrict__ C0, float* __restrict__ Vn, float* __restrict__ En,
                                                            float* __restrict__ direct, float* __restrict__ dloo, float* __restrict__ part,
                                                            int N, int M, int D) {
  __shared__ float sS[GE2E_MAXN], sC[GE2E_MAXN], sV[GE2E_MAXN], sH[GE2E_MAXN];
  __shared__ float red4[4];
  const int ji = blockIdx.x, j = ji / M, tid = threadIdx.x;
  const float* e = emb + (long)ji * D;
  const float w = wp[0], b = bp[0], gs = dloss ? dloss[0] : 1.f;
  const float invM = 1.f / (float)M, invM1 = 1.f / (float)(M - 1);
  float t = 0.f;
  for (int d = tid; d < D; d += 256) t += e[d] * e[d];
  const float en = sqrtf(block_sum_256(t, red4));
  for (int k = tid >> 6; k < N; k += 4) {                     // a wave per centroid, lanes over D: two wave sums, no barrier (352 barriers before)
    const float* c = csum + (long)k * D;
    float dot = 0.f, cn = 0.f;
    for (int d = tid & 63; d < D; d += 64) {
      const float v = (k == j) ? (c[d] - e[d]) * invM1 : c[d] * invM;
      dot += e[d] * v; cn += v * v;
    }
    dot = ssv_wave_sum(dot);
    cn = ssv_wave_sum(cn);
    if ((tid & 63) == 0) {
      const float vn = sqrtf(cn), c0 = dot / fmaxf(en * vn, 1e-8f);
      sC[k] = c0; sV[k] = vn; sS[k] = w * (c0 + 1e-6f) + b;
    }
  }
  __syncthreads();
  float se = 0.f;
  for (int k = tid; k < N; k += 256) se += expf(sS[k]);
  se = block_sum_256(se, red4);
  float pw = 0.f, pb = 0.f;
  for (int k = tid; k < N; k += 256) {
    const float G = expf(sS[k]) / (se + 1e-6f) - (k == j ? 1.f : 0.f);
    pw += G * (sC[k] + 1e-6f); pb += G;
    sH[k] = gs * w * G;
    Hc[(long)ji * N + k] = sH[k]; C0[(long)ji * N + k] = sC[k]; Vn[(long)ji * N + k] = sV[k];
  }
  pw = block_sum_256(pw, red4);
  pb = block_sum_256(pb, red4);
  if (tid == 0) { part[2 * ji] = pw; part[2 * ji + 1] = pb; En[ji] = en; }
  __syncthreads();
  for (int d = tid; d < D; d += 256) {
    float acc = 0.f;
    for (int k = 0; k < N; ++k) {
      const float v = (k == j) ? (csum[(long)k * D + d] - e[d]) * invM1 : csum[(long)k * D + d] * invM;
      const float den = fmaxf(en * sV[k], 1e-8f);
      acc += sH[k] * (v / den - sC[k] * e[d] / (en * en));
    }
    direct[(long)ji * D + d] = acc;
    const float vj = (csum[(long)j * D + d] - e[d]) * invM1, denj = fmaxf(en * sV[j], 1e-8f);
    dloo[(long)ji * D + d] = sH[j] * (e[d] / denj - sC[j] * vj / fmaxf(sV[j] * sV[j], 1e-16f)) * invM1;
  }
}
// one workgroup per speaker k: gradient that reaches the centroid c_k from the rows of the OTHER speakers.  16 waves: thread = (component d, one of four
// interleaved row ranges), four rows in flight per thread, the four ranges added in turn (one thread per component walking all N M rows one load
// at a time took 0.32 ms at config 5).
__global__ __launch_bounds__(1024) void ge2e_bwd_centroid_kernel(const float* __restrict__ emb, const float* __restrict__ csum, const float* __restrict__ Hc,
                                                                 const float* __restrict__ C0, const float* __restrict__ En, float* __restrict__ dcent,
                                                                 int N, int M, int D) {
  __shared__ float part[4][256];
  const int k = blockIdx.x, tid = threadIdx.x, dl = tid & 255, g = tid >> 8, lane = tid & 63;
  const float invM = 1.f / (float)M;
  float t = 0.f;
  for (int d = lane; d < D; d += 64) { const float v = csum[(long)k * D + d] * invM; t += v * v; }
  const float vk = sqrtf(ssv_wave_sum(t));                                   // (every wave on its own)
  const float vk2 = fmaxf(vk * vk, 1e-16f);
  const int NM = N * M;
  for (int d0 = 0; d0 < D; d0 += 256) {
    const int d = d0 + dl;
    const bool dok = d < D;
    const float v = dok ? csum[(long)k * D + d] * invM : 0.f;
    float acc = 0.f;
    auto term = [&](int ji, float h, float ev, float c0, float en) -> float { return (ji / M == k) ? 0.f : h * (ev / fmaxf(en * vk, 1e-8f) - c0 * v / vk2); };
    int ji = g;
    for (; ji + 12 < NM; ji += 16) {
      float h[4], ev[4], c0[4], en[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = ji + 4 * q;
        h[q] = Hc[(long)r * N + k]; c0[q] = C0[(long)r * N + k]; en[q] = En[r]; ev[q] = dok ? emb[(long)r * D + d] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) acc += term(ji + 4 * q, h[q], ev[q], c0[q], en[q]);
    }
    for (; ji < NM; ji += 4) acc += term(ji, Hc[(long)ji * N + k], dok ? emb[(long)ji * D + d] : 0.f, C0[(long)ji * N + k], En[ji]);
    __syncthreads();
    part[g][dl] = acc;
    __syncthreads();
    if (g == 0 && dok) dcent[(long)k * D + d] = (((part[0][dl] + part[1][dl]) + part[2][dl]) + part[3][dl]) * invM;
  }
}
__global__ __launch_bounds__(256) void ge2e_bwd_finish_kernel(const float* __restrict__ direct, const float* __restrict__ dcent, const float* __restrict__ dloo,
                                                              float* __restrict__ demb, int N, int M, int D) {
  const int jm = blockIdx.x, j = jm / M;
  for (int d = threadIdx.x; d < D; d += 256) {
    float s = 0.f;
    for (int i = 0; i < M; ++i) s += dloo[((long)j * M + i) * D + d];
    demb[(long)jm * D + d] = direct[(long)jm * D + d] + dcent[(long)j * D + d] + (s - dloo[(long)jm * D + d]);
  }
}
__global__ __launch_bounds__(256) void ge2e_bwd_wb_kernel(const float* __restrict__ part, const float* __restrict__ dloss, float* __restrict__ dw, float* __restrict__ db, int n) {
  __shared__ float red4[4];
  float a = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) { a += part[2 * i]; c += part[2 * i + 1]; }
  a = block_sum_256(a, red4);
  c = block_sum_256(c, red4);
  const float gs = dloss ? dloss[0] : 1.f;
  if (threadIdx.x == 0) { dw[0] = gs * a; db[0] = gs * c; }
}
extern "C" size_t ssv_ge2e_loss_bwd_workspace(int N, int M, int D) {
  const size_t nm = (size_t)N * M;
  return (3 * nm * N + nm + 2 * nm * D + 2 * (size_t)N * D + 2 * nm) * sizeof(float);
}
extern "C" int ssv_ge2e_loss_bwd(const float* emb, const float* w, const float* b, const float* dloss, float* demb, float* dw, float* db,
                                 int N, int M, int D, void* ws, size_t ws_bytes, ssv_stream_t stream) {
  SSV_CHECK(emb && w && b && demb && dw && db && N > 0 && M > 1 && D > 0, SSV_BAD_SHAPE, "ge2e_loss_bwd: need N>0, M>1, D>0 (N=%d M=%d D=%d)", N, M, D);
  SSV_CHECK(N <= GE2E_MAXN, SSV_UNSUPPORTED, "ge2e_loss_bwd: N=%d > %d speakers", N, GE2E_MAXN);
  SSV_CHECK(ws && ws_bytes >= ssv_ge2e_loss_bwd_workspace(N, M, D), SSV_BAD_SHAPE, "ge2e_loss_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const size_t nm = (size_t)N * M;
  float* Hc = (float*)ws;
  float* C0 = Hc + nm * N;
  float* Vn = C0 + nm * N;
  float* En = Vn + nm * N;
  float* direct = En + nm;
  float* dloo = direct + nm * D;
  float* dcent = dloo + nm * D;
  float* csum = dcent + (size_t)N * D;
  float* part = csum + (size_t)N * D;
  hipLaunchKernelGGL(ge2e_centroid_kernel, dim3(ssv_cdiv((long)N * D, 256)), dim3(256), 0, st, emb, csum, N, M, D);
  SSV_TRY(ssv_check_launch("ge2e_centroid"));
  hipLaunchKernelGGL(ge2e_bwd_rows_kernel, dim3(N * M), dim3(256), 0, st, emb, (const float*)csum, w, b, dloss, Hc, C0, Vn, En, direct, dloo, part, N, M, D);
  SSV_TRY(ssv_check_launch("ge2e_bwd_rows"));
  hipLaunchKernelGGL(ge2e_bwd_centroid_kernel, dim3(N), dim3(1024), 0, st, emb, (const float*)csum, (const float*)Hc, (const float*)C0, (const float*)En, dcent, N, M, D);
  SSV_TRY(ssv_check_launch("ge2e_bwd_centroid"));
  hipLaunchKernelGGL(ge2e_bwd_finish_kernel, dim3(N * M), dim3(256), 0, st, (const float*)direct, (const float*)dcent, (const float*)dloo, demb, N, M, D);
  SSV_TRY(ssv_check_launch("ge2e_bwd_finish"));
  hipLaunchKernelGGL(ge2e_bwd_wb_kernel, dim3(1), dim3(256), 0, st, (const float*)part, dloss, dw, db, N * M);
  return ssv_check_launch("ge2e_bwd_wb");
}


// ---- LSTM backward: cell ----------------------------------------------------------------------------------
// Layer l = lo + blockIdx.y at frame t = s - l (reverse wavefront step s).  dh_t collects the gradient from the layer above
// (rows [0, H) of that layer's data-gradient product at the same frame), from the layer's own next frame (rows [H, 2H) of its own
// product at t + 1) and, for the top layer's last frame, from the projection.  Both products ran at step s + 1 and lie under that step's
// parity in dxa = [K range z][parity][layer][2H][Bn] (api.hip, lstm_bwd_ws); nz partial products are added here, in a fixed order.
// Writes the pre-activation gate gradients dgates[l][t] in torch row order (gate*H + u) and carries dc_{t-1} = dc_t * f_t in dcarry[l].
// One workgroup per hidden unit u (a row of Bn utterances) and layer: besides the cells it leaves the row sums of the four gate gradients --
// the bias gradient's terms of this frame, dbp[l][t][gate * H + u] -- in a fixed order (per thread over its columns, the 64 lanes of a wave by
// ssv_wave_sum, the four waves in turn): the separate pass over all of dgates (ssv_rowsum: 3 x 0.22 ms for config 5) is gone.
// V = 4: Bn % 4 == 0, a thread takes four neighbouring columns as 16-byte vectors (every row then starts 16-byte aligned); V = 1: any Bn.
template <int V>
__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(const float* __restrict__ gates, const float* __restrict__ cs, const float* __restrict__ dxa,
                                                            const long zstride, const int nz, const float* __restrict__ dh_top, float* __restrict__ dgates,
                                                            float* __restrict__ dcarry, float* __restrict__ dbp, int H, int Bn, int T, int layers, int s, int lo) {
  typedef float vec __attribute__((ext_vector_type(V)));
  __shared__ float red[4][4];
  const int u = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long HN = (long)H * Bn;
  const int l = lo + blockIdx.y, t = s - l;
  const float* __restrict__ d = dxa + (long)((s + 1) & 1) * layers * 2 * HN;
  const long gt = ((long)l * T + t) * 4 * HN;
  const bool above = l + 1 < layers, nextf = t + 1 < T, top = l == layers - 1 && t == T - 1;
  auto ld = [&](const float* q) -> vec { return *reinterpret_cast<const vec*>(q); };
  auto st = [&](float* q, vec v) { *reinterpret_cast<vec*>(q) = v; };
  vec sum[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) sum[g] = (vec)0.f;
  // every load of a pass in front of its first store (a load issued behind a store waits for it: one vmcnt)
  for (int b = tid * V; b < Bn; b += 256 * V) {
    const long i = (long)u * Bn + b;
    vec dh = (vec)0.f;
    for (int z = 0; z < nz; ++z) {
      if (above) dh += ld(d + z * zstride + (long)(l + 1) * 2 * HN + i);
      if (nextf) dh += ld(d + z * zstride + (long)l * 2 * HN + HN + i);
    }
    if (top) dh += ld(dh_top + i);
    const vec gi = ld(gates + gt + i), gf = ld(gates + gt + HN + i), gg = ld(gates + gt + 2 * HN + i), go = ld(gates + gt + 3 * HN + i);
    const vec c = ld(cs + ((long)l * T + t) * HN + i);
    const vec cprev = t > 0 ? ld(cs + ((long)l * T + t - 1) * HN + i) : (vec)0.f;
    const vec dcin = nextf ? ld(dcarry + (long)l * HN + i) : (vec)0.f;
    vec tc;
#pragma unroll
    for (int k = 0; k < V; ++k) tc[k] = tanhf(c[k]);
    const vec dc = dh * go * (1.f - tc * tc) + dcin;
    const vec di = dc * gg * gi * (1.f - gi), df = dc * cprev * gf * (1.f - gf), dg = dc * gi * (1.f - gg * gg), d_o = dh * tc * go * (1.f - go);
    st(dcarry + (long)l * HN + i, dc * gf);
    st(dgates + gt + i, di);
    st(dgates + gt + HN + i, df);
    st(dgates + gt + 2 * HN + i, dg);
    st(dgates + gt + 3 * HN + i, d_o);
    sum[0] += di; sum[1] += df; sum[2] += dg; sum[3] += d_o;
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float v = sum[g][0];
    if constexpr (V == 4) v = (sum[g][0] + sum[g][1]) + (sum[g][2] + sum[g][3]);
    v = ssv_wave_sum(v);
    if (lane == 0) red[wave][g] = v;
  }
  __syncthreads();
  if (tid < 4) dbp[((long)l * T + t) * 4 * H + (long)tid * H + u] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
}
int ssv_launch_lstm_cell_bwd(const float* gates, const float* cs, const float* dxa, long zstride, int nz, const float* dh_top, float* dgates, float* dcarry,
                             float* dbp, int H, int Bn, int T, int layers, int s, int lo, int nl, hipStream_t st) {
  // (the vector form also wants 16-byte aligned bases: the workspace sections are 256-byte aligned, the saved tensors come from the caller)
  const bool v4 = Bn % 4 == 0 && zstride % 4 == 0 && (((uintptr_t)gates | (uintptr_t)cs | (uintptr_t)dxa | (uintptr_t)dh_top | (uintptr_t)dgates | (uintptr_t)dcarry) & 15) == 0;
  if (v4) hipLaunchKernelGGL(lstm_cell_bwd_kernel<4>, dim3(H, nl), dim3(256), 0, st, gates, cs, dxa, zstride, nz, dh_top, dgates, dcarry, dbp, H, Bn, T, layers, s, lo);
  else hipLaunchKernelGGL(lstm_cell_bwd_kernel<1>, dim3(H, nl), dim3(256), 0, st, gates, cs, dxa, zstride, nz, dh_top, dgates, dcarry, dbp, H, Bn, T, layers, s, lo);
  return ssv_check_launch("lstm_cell_bwd");
}

// ---- projection + L2 normalisation backward ---------------------------------------------------------------
// e = y / |y|:  dy = (de - e <e, de>) / |y|.  One wave per batch row.
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ de, const float* __restrict__ e, const float* __restrict__ norms,
                                                         float* __restrict__ dy, int P, int Bn) {
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (b >= Bn) return;
  float s = 0.f;
  for (int p = lane; p < P; p += 64) s += e[(long)b * P + p] * de[(long)b * P + p];
  s = ssv_wave_sum(s);
  const float inv = 1.f / norms[b];
  for (int p = lane; p < P; p += 64) dy[(long)b * P + p] = (de[(long)b * P + p] - e[(long)b * P + p] * s) * inv;
}
// out[p] = sum_b x[b][p]   (x (Bn, P) row-major), fixed order: 64 columns per workgroup of 16 waves, wave w adds rows w, w + 16, ... (four loads in
// flight), then the sixteen waves in turn (one thread per column walking all Bn rows took 0.2 ms of config 5's iteration)
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, int P, int Bn) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = blockIdx.x * 64 + lane;
  const int pc = p < P ? p : P - 1;
  float s = 0.f;
  int b = wave;
  for (; b + 48 < Bn; b += 64) {
    const float v0 = x[(long)b * P + pc], v1 = x[(long)(b + 16) * P + pc], v2 = x[(long)(b + 32) * P + pc], v3 = x[(long)(b + 48) * P + pc];
    s += v0; s += v1; s += v2; s += v3;
  }
  for (; b < Bn; b += 16) s += x[(long)b * P + pc];
  red[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && p < P) {
    float r = red[0][lane];
#pragma unroll
    for (int w = 1; w < 16; ++w) r += red[w][lane];
    out[p] = r;
  }
}
int ssv_launch_l2norm_bwd(const float* de, const float* e, const float* norms, float* dy, int P, int Bn, hipStream_t st) {
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(ssv_cdiv(Bn, 4)), dim3(256), 0, st, de, e, norms, dy, P, Bn);
  return ssv_check_launch("l2norm_bwd");
}
int ssv_launch_colsum(const float* x, float* out, int P, int Bn, hipStream_t st) {
  hipLaunchKernelGGL(colsum_kernel, dim3(ssv_cdiv(P, 64)), dim3(1024), 0, st, x, out, P, Bn);
  return ssv_check_launch("colsum");
}
