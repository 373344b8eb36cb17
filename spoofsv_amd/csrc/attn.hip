// Attention softmax kernels (column softmax over the text axis N of a (B, N, T) score tensor) and the
// fused single-step kernel used during synthesis (gfx950).  The two GEMMs around the softmax run on
// the shared implicit-GEMM kernels (gemm_nn / gemm_nt); these kernels are HBM/L2-bound glue.
#include "ssv_common.h"
#include "../../include/ssv_hip.h"

// One thread per (b, t) column; consecutive threads walk consecutive t, so every row read is coalesced.
__global__ __launch_bounds__(256) void softmax_cols_kernel(float* __restrict__ s, int B, int N, int T) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * T) return;
  const int b = (int)(i / T), t = (int)(i % T);
  float* p = s + (long)b * N * T + t;
  float mx = -INFINITY;
  for (int n = 0; n < N; ++n) mx = fmaxf(mx, p[(long)n * T]);
  float sum = 0.f;
  for (int n = 0; n < N; ++n) { const float e = expf(p[(long)n * T] - mx); p[(long)n * T] = e; sum += e; }
  const float inv = 1.f / sum;
  for (int n = 0; n < N; ++n) p[(long)n * T] *= inv;
}

// ds = a * (da + da_ext - sum_n a*(da + da_ext)) * scale, in place on da.
__global__ __launch_bounds__(256) void softmax_cols_bwd_kernel(const float* __restrict__ a, float* __restrict__ da,
                                                               const float* __restrict__ da_ext, float scale, int B, int N, int T) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * T) return;
  const int b = (int)(i / T), t = (int)(i % T);
  const long base = (long)b * N * T + t;
  float dot = 0.f;
  for (int n = 0; n < N; ++n) {
    const long k = base + (long)n * T;
    float g = da[k];
    if (da_ext) g += da_ext[k];
    dot += a[k] * g;
  }
  for (int n = 0; n < N; ++n) {
    const long k = base + (long)n * T;
    float g = da[k];
    if (da_ext) g += da_ext[k];
    da[k] = a[k] * (g - dot) * scale;
  }
}

// Tiled versions for N <= 256 text positions (the configured maximum is 186): a workgroup owns 32 columns of one batch
// item, its 8 row groups hold rows rg, rg+8, ... in registers, and the column max / sums are combined through LDS in a fixed
// order.  B*ceil(T/32) workgroups instead of B*T/256, one read and one write per element.
template <int NR>
__global__ __launch_bounds__(256) void softmax_cols_tile_kernel(float* __restrict__ s, int N, int T) {
  __shared__ float red[8][32];
  const int col = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int t = blockIdx.x * 32 + col, b = blockIdx.y;
  const bool tv = t < T;
  float* __restrict__ p = s + (long)b * N * T;
  const unsigned o0 = (unsigned)rg * T + t, ostep = 8u * T;
  float v[NR];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    v[i] = (tv && rg + 8 * i < N) ? p[o0 + i * ostep] : -INFINITY;
    mx = fmaxf(mx, v[i]);
  }
  red[rg][col] = mx;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 8; ++r) mx = fmaxf(mx, red[r][col]);
  __syncthreads();
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    v[i] = (tv && rg + 8 * i < N) ? expf(v[i] - mx) : 0.f;
    sum += v[i];
  }
  red[rg][col] = sum;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r) tot += red[r][col];
  const float inv = 1.f / tot;
#pragma unroll
  for (int i = 0; i < NR; ++i)
    if (tv && rg + 8 * i < N) p[o0 + i * ostep] = v[i] * inv;
}
template <int NR>
__global__ __launch_bounds__(256) void softmax_cols_bwd_tile_kernel(const float* __restrict__ a, float* __restrict__ da,
                                                                    const float* __restrict__ da_ext, float scale, int N, int T) {
  __shared__ float red[8][32];
  const int col = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int t = blockIdx.x * 32 + col, b = blockIdx.y;
  const bool tv = t < T;
  const long base = (long)b * N * T;
  const float* __restrict__ ab = a + base;
  float* __restrict__ db = da + base;
  const float* __restrict__ eb = da_ext ? da_ext + base : nullptr;
  const unsigned o0 = (unsigned)rg * T + t, ostep = 8u * T;
  float av[NR], g[NR];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const bool v = tv && rg + 8 * i < N;
    av[i] = v ? ab[o0 + i * ostep] : 0.f;
    g[i] = v ? db[o0 + i * ostep] : 0.f;
    if (eb && v) g[i] += eb[o0 + i * ostep];
    dot += av[i] * g[i];
  }
  red[rg][col] = dot;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r) tot += red[r][col];
#pragma unroll
  for (int i = 0; i < NR; ++i)
    if (tv && rg + 8 * i < N) db[o0 + i * ostep] = av[i] * (g[i] - tot) * scale;
}

int ssv_launch_softmax_cols(float* s, int B, int N, int T, hipStream_t st) {
  if (N <= 256 && B <= 65535 && (long)N * T < (1L << 31)) {
    dim3 grid(ssv_cdiv(T, 32), B);
    if (N <= 64) hipLaunchKernelGGL(softmax_cols_tile_kernel<8>, grid, dim3(256), 0, st, s, N, T);
    else if (N <= 192) hipLaunchKernelGGL(softmax_cols_tile_kernel<24>, grid, dim3(256), 0, st, s, N, T);
    else hipLaunchKernelGGL(softmax_cols_tile_kernel<32>, grid, dim3(256), 0, st, s, N, T);
    return ssv_check_launch("softmax_cols_tile");
  }
  hipLaunchKernelGGL(softmax_cols_kernel, dim3(ssv_cdiv((long)B * T, 256)), dim3(256), 0, st, s, B, N, T);
  return ssv_check_launch("softmax_cols");
}
int ssv_launch_softmax_cols_bwd(const float* a, float* da, const float* da_ext, float scale, int B, int N, int T, hipStream_t st) {
  if (N <= 256 && B <= 65535 && (long)N * T < (1L << 31)) {
    dim3 grid(ssv_cdiv(T, 32), B);
    if (N <= 64) hipLaunchKernelGGL(softmax_cols_bwd_tile_kernel<8>, grid, dim3(256), 0, st, a, da, da_ext, scale, N, T);
    else if (N <= 192) hipLaunchKernelGGL(softmax_cols_bwd_tile_kernel<24>, grid, dim3(256), 0, st, a, da, da_ext, scale, N, T);
    else hipLaunchKernelGGL(softmax_cols_bwd_tile_kernel<32>, grid, dim3(256), 0, st, a, da, da_ext, scale, N, T);
    return ssv_check_launch("softmax_cols_bwd_tile");
  }
  hipLaunchKernelGGL(softmax_cols_bwd_kernel, dim3(ssv_cdiv((long)B * T, 256)), dim3(256), 0, st, a, da, da_ext, scale, B, N, T);
  return ssv_check_launch("softmax_cols_bwd");
}

// ---- synthesis step: one workgroup per batch item --------------------------------------------------
// logits[n] = (sum_c k[c][n] q[c]) / sqrt(d); positions outside [pma, pma+2] are set to -2^32 exactly
// as the reference does before its softmax (their exp underflows to 0); the new attention column and
// the first arg-max index are written.
#define STEP_MAXN 1024
__global__ __launch_bounds__(256) void attention_step_kernel(const float* __restrict__ k, long kv_bs, const float* __restrict__ q, long q_bs, long q_cs,
                                                             const int64_t* __restrict__ pma_in, float* __restrict__ a, int a_T, int col,
                                                             const int* __restrict__ col_dev, int64_t* __restrict__ pma_out, int d, int N, float scale) {
  __shared__ float logit[STEP_MAXN];
  __shared__ float red[4];
  __shared__ int redi[4];
  if (col_dev) { col = col_dev[0]; q += col; }       // device-side frame counter: q is column 0 of Q, the step's frame is added here
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* kb = k + (long)b * kv_bs;
  const float* qb = q + (long)b * q_bs;
  const int64_t pma = pma_in[b];
  for (int n = tid; n < N; n += 256) {
    float s = 0.f;
    for (int c = 0; c < d; ++c) s = fmaf(kb[(long)c * N + n], qb[(long)c * q_cs], s);
    s *= scale;
    if (n < pma || n >= pma + 3) s = -4294967296.f;
    logit[n] = s;
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int n = tid; n < N; n += 256) mx = fmaxf(mx, logit[n]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int n = tid; n < N; n += 256) { const float e = expf(logit[n] - mx); logit[n] = e; sum += e; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  if ((tid & 63) == 0) red[tid >> 6] = sum;
  __syncthreads();
  sum = (red[0] + red[1]) + (red[2] + red[3]);
  const float inv = 1.f / sum;
  // arg-max over the normalised column (first index wins ties, as torch.argmax on CPU does)
  float best = -1.f; int bi = N;
  for (int n = tid; n < N; n += 256) {
    const float p = logit[n] * inv;
    a[((long)b * N + n) * a_T + col] = p;
    if (p > best) { best = p; bi = n; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  __syncthreads();
  if ((tid & 63) == 0) { red[tid >> 6] = best; redi[tid >> 6] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (red[w] > best || (red[w] == best && redi[w] < bi)) { best = red[w]; bi = redi[w]; }
    pma_out[b] = bi;
  }
}

extern "C" int ssv_attention_step(const float* k, long kv_bs, const float* q_last, long q_bs, long q_cs, const int64_t* pma_in,
                                  float* a, int a_T, int col, const int* col_dev, int64_t* pma_out, int B, int d, int N, ssv_stream_t stream) {
  SSV_CHECK(B > 0 && d > 0 && N > 0 && (col_dev || (col >= 0 && col < a_T)), SSV_BAD_SHAPE, "attention_step: bad shape B=%d d=%d N=%d col=%d a_T=%d", B, d, N, col, a_T);
  SSV_CHECK(N <= STEP_MAXN, SSV_UNSUPPORTED, "attention_step: N=%d > %d", N, STEP_MAXN);
  hipLaunchKernelGGL(attention_step_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, k, kv_bs, q_last, q_bs, q_cs, pma_in, a, a_T, col, col_dev,
                     pma_out, d, N, 1.f / sqrtf((float)d));
  return ssv_check_launch("attention_step");
}

// ---- synthesis loop bookkeeping (hipGraph replay of one fixed-shape step) ------------------------------------------
// mel_in[b][f][col+1] = y[b][f][col] (the frame just synthesised becomes the next input, synthesize.py:108-109), then col += 1.
__global__ __launch_bounds__(256) void synth_feed_kernel(const float* __restrict__ y, float* __restrict__ mel_in, const int* __restrict__ col_dev,
                                                         int n, int T) {
  const int i = blockIdx.x * 256 + threadIdx.x;            // (b, f) row
  const int col = col_dev[0];
  if (i < n && col + 1 < T) mel_in[(long)i * T + col + 1] = y[(long)i * T + col];
}
__global__ void synth_inc_kernel(int* col_dev) { col_dev[0] += 1; }
extern "C" int ssv_synth_advance(const float* y, float* mel_in, int* col_dev, int B, int F, int T, ssv_stream_t stream) {
  SSV_CHECK(y && mel_in && col_dev && B > 0 && F > 0 && T > 0, SSV_BAD_SHAPE, "synth_advance: bad argument");
  hipLaunchKernelGGL(synth_feed_kernel, dim3(ssv_cdiv((long)B * F, 256)), dim3(256), 0, (hipStream_t)stream, y, mel_in, (const int*)col_dev, B * F, T);
  SSV_TRY(ssv_check_launch("synth_feed"));
  hipLaunchKernelGGL(synth_inc_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, col_dev);
  return ssv_check_launch("synth_inc");
}
