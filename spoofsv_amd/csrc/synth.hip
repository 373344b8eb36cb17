// Column-incremental synthesis step (models/TTSModel.py:275-300 as driven by synthesize.py:103-109), gfx950.
//
// The reference re-encodes the whole prefix at every step (O(T^2)).  Everything on the audio side is causal, so the
// columns of every layer for frames < t never change: only column t of each layer is new.  These kernels compute that
// one column.  Activations of a step are (B, C) matrices with the channels contiguous; the input history of every causal
// k = 3 convolution is a (B, Tmax, C) ring of such columns.  The frame counter t lives on the device (t_dev), so one
// captured hipGraph of the whole step can be replayed for every frame.
#include "ssv_common.h"

#define CMV_BG 8          // batch items per workgroup (their input vectors are staged in LDS)
#define CMV_KMAX 1536     // C * k

// out[b][m] = bias[m] + bias_b[b][m] + sum_{j, c} w[m][j][c] * x_j[b][c]        (w TAP-MAJOR: (M, k, C), see ssv_hip.h)
//   k = 1: x_0 = cur.   k = 3 (causal, dilation d): x_2 = cur (column t), x_1 = hist[t - d], x_0 = hist[t - 2d], zero before 0.
// By-product (row block 0): cur is stored as column t of hist, for the steps to come.
// Workgroup (x, y): output rows 4x..4x+3 (one per wave) for batch items 8y..8y+7.  The staged inputs are the k columns
// one after the other, like a tap-major weight row, so a row is one contiguous dot product of length K = k*C.  In a wave, lane = 8*bb + ks:
// the 8 lanes of a batch item take every 8th float4 of the row (a weight read is 128 contiguous bytes, shared by the 8
// batch groups of lanes) and are summed with three DPP steps.  The kernel is one latency chain (frame counter -> history
// columns -> LDS -> dot product); it is launched ~25 times per frame, so it is kept short rather than wide.
template <int KT>
__global__ __launch_bounds__(256) void column_matvec_kernel(
    const float* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ bias_b, long bb_bs,
    const float* __restrict__ cur, long cur_bs, float* __restrict__ hist, long hist_bs, int Tmax,
    const int* __restrict__ t_dev, int dil, float* __restrict__ out, long out_bs, int B, int C, int M) {
  __shared__ __attribute__((aligned(16))) float xs[CMV_BG * (CMV_KMAX + 4)];
  const int K = C * KT, KP = K + 4, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m = blockIdx.x * 4 + wave, b0 = blockIdx.y * CMV_BG;
  const int t = (KT > 1) ? t_dev[0] : 0;
  // this lane's share of the weight row (every 8th float4, the first 24 of them = K up to 768) is requested before the inputs
  // are staged, so that its latency overlaps the history reads and the barrier
  constexpr int NPRE = 24;
  const int bbl = lane >> 3, ks = lane & 7, b = b0 + bbl, nq = K / 4;
  const float* wr = w + (long)(m < M ? m : M - 1) * K;
  f32x4 wp[NPRE];
#pragma unroll
  for (int u = 0; u < NPRE; ++u) {
    const int q = ks + 8 * u;
    wp[u] = *reinterpret_cast<const f32x4*>(wr + 4 * (q < nq ? q : 0));
  }
  // staging: the K = k*C inputs of a batch item are its k columns one after the other (tap-major, like the weights), each
  // a contiguous run of C floats in memory: 16-byte copies
  const int c4 = C / 4;
#pragma unroll
  for (int bb = 0; bb < CMV_BG; ++bb) {
    const int bi = b0 + bb;
    for (int i = tid; i < nq; i += 256) {
      const int j = i / c4, c = 4 * (i - j * c4);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (bi < B) {
        if (j == KT - 1) {
          v = *reinterpret_cast<const f32x4*>(cur + (long)bi * cur_bs + c);
          if (KT > 1 && blockIdx.x == 0 && t < Tmax) *reinterpret_cast<f32x4*>(hist + (long)bi * hist_bs + (long)t * C + c) = v;
        } else {
          const int col = t - (KT - 1 - j) * dil;
          if (col >= 0) v = *reinterpret_cast<const f32x4*>(hist + (long)bi * hist_bs + (long)col * C + c);
        }
      }
      *reinterpret_cast<f32x4*>(xs + bb * KP + 4 * i) = v;
    }
  }
  __syncthreads();
  if (m >= M) return;                          // wave-uniform, after the only barrier
  const float* xr = xs + bbl * KP;
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int u = 0; u < NPRE; ++u) {
    const int q = ks + 8 * u;
    if (q < nq) {
      const f32x4 x0 = *reinterpret_cast<const f32x4*>(xr + 4 * q), w0 = wp[u];
      if (u & 1) a1 = fmaf(w0.x, x0.x, fmaf(w0.y, x0.y, fmaf(w0.z, x0.z, fmaf(w0.w, x0.w, a1))));
      else a0 = fmaf(w0.x, x0.x, fmaf(w0.y, x0.y, fmaf(w0.z, x0.z, fmaf(w0.w, x0.w, a0))));
    }
  }
  for (int q = ks + 8 * NPRE; q < nq; q += 8) {    // K > 768
    const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr + 4 * q), x0 = *reinterpret_cast<const f32x4*>(xr + 4 * q);
    a0 = fmaf(w0.x, x0.x, fmaf(w0.y, x0.y, fmaf(w0.z, x0.z, fmaf(w0.w, x0.w, a0))));
  }
  float s = a0 + a1;
  s += ssv_dpp_mov<0xB1>(s);                   // the 8 lanes of a batch item: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror
  s += ssv_dpp_mov<0x4E>(s);
  s += ssv_dpp_mov<0x141>(s);
  if (ks == 0 && b < B) {
    float o = s + (bias ? bias[m] : 0.f);
    if (bias_b) o += bias_b[(long)b * bb_bs + m];
    out[(long)b * out_bs + m] = o;
  }
}

extern "C" int ssv_column_matvec(const float* w, const float* bias, const float* bias_b, long bb_bs, const float* cur, long cur_bs,
                                 float* hist, long hist_bs, int Tmax, const int* t_dev, int dilation, float* out, long out_bs,
                                 int B, int C, int M, int k, ssv_stream_t stream) {
  SSV_CHECK(w && cur && out && B > 0 && B <= 8 * 65535 && C > 0 && M > 0 && (k == 1 || k == 3), SSV_BAD_SHAPE, "column_matvec: bad argument B=%d C=%d M=%d k=%d", B, C, M, k);
  SSV_CHECK(C % 4 == 0 && C * k <= CMV_KMAX, SSV_UNSUPPORTED, "column_matvec: C = %d must be a multiple of 4 and C*k at most %d", C, CMV_KMAX);
  SSV_CHECK(cur_bs % 4 == 0 && (k == 1 || hist_bs % 4 == 0), SSV_BAD_SHAPE, "column_matvec: batch strides must be multiples of 4 floats");
  SSV_CHECK(k == 1 || (hist && t_dev && Tmax > 0 && dilation > 0), SSV_BAD_SHAPE, "column_matvec: kernel size 3 needs a history, its length and the frame counter");
  dim3 grid(ssv_cdiv(M, 4), ssv_cdiv(B, CMV_BG));
  if (k == 1)
    hipLaunchKernelGGL(column_matvec_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, w, bias, bias_b, bb_bs, cur, cur_bs, hist, hist_bs, Tmax, t_dev, dilation,
                       out, out_bs, B, C, M);
  else
    hipLaunchKernelGGL(column_matvec_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, w, bias, bias_b, bb_bs, cur, cur_bs, hist, hist_bs, Tmax, t_dev, dilation,
                       out, out_bs, B, C, M);
  return ssv_check_launch("column_matvec");
}

// ---- LayerNorm (+ activation) and the highway gate on ONE column: a workgroup per batch item, a thread per channel ---------
// Same arithmetic as ln_act_fwd_kernel / ln_gate_fwd_kernel of norm.hip (two-pass mean and variance, eps 1e-5), which at a
// sequence length of 1 keep 16 of their 256 threads busy (7-9 us per launch, ~26 launches per frame).
#define COL_EPS 1e-5f
__device__ __forceinline__ float col_sigmoid(float v) { return 1.f / (1.f + __expf(-v)); }
__device__ __forceinline__ float block_sum(float v, float* red) {        // sum over the 256 threads, result in every thread
  v = ssv_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
#define COL_CPT 4         // channels per thread: C <= 1024
__global__ __launch_bounds__(256) void column_ln_act_kernel(const float* __restrict__ x, long x_bs, const float* __restrict__ gam,
                                                            const float* __restrict__ bet, float* __restrict__ y, long y_bs, int C, int act) {
  __shared__ float red[4];
  const float* xb = x + (long)blockIdx.x * x_bs;
  float v[COL_CPT], s = 0.f;
#pragma unroll
  for (int i = 0; i < COL_CPT; ++i) { const int c = threadIdx.x + 256 * i; v[i] = c < C ? xb[c] : 0.f; s += v[i]; }
  const float inv = 1.f / (float)C, mu = block_sum(s, red) * inv;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < COL_CPT; ++i) { const float dlt = (threadIdx.x + 256 * i) < C ? v[i] - mu : 0.f; q += dlt * dlt; }
  const float r = rsqrtf(block_sum(q, red) * inv + COL_EPS);
  float* yb = y + (long)blockIdx.x * y_bs;
#pragma unroll
  for (int i = 0; i < COL_CPT; ++i) {
    const int c = threadIdx.x + 256 * i;
    if (c < C) {
      float n = (v[i] - mu) * r * gam[c] + bet[c];
      if (act == 1) n = fmaxf(n, 0.f);
      else if (act == 2) n = col_sigmoid(n);
      yb[c] = n;
    }
  }
}
__global__ __launch_bounds__(256) void column_gate_kernel(const float* __restrict__ h, const float* __restrict__ x, long x_bs,
                                                          const float* __restrict__ g1, const float* __restrict__ b1, const float* __restrict__ g2,
                                                          const float* __restrict__ b2, float* __restrict__ y, long y_bs, int C) {
  __shared__ float red[4];
  const float* h1 = h + (long)blockIdx.x * 2 * C;
  const float* h2 = h1 + C;
  float u[COL_CPT], v[COL_CPT], s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < COL_CPT; ++i) {
    const int c = threadIdx.x + 256 * i;
    u[i] = c < C ? h1[c] : 0.f; v[i] = c < C ? h2[c] : 0.f;
    s1 += u[i]; s2 += v[i];
  }
  const float inv = 1.f / (float)C;
  const float mu1 = block_sum(s1, red) * inv, mu2 = block_sum(s2, red) * inv;
  float q1 = 0.f, q2 = 0.f;
#pragma unroll
  for (int i = 0; i < COL_CPT; ++i) {
    const bool ok = (threadIdx.x + 256 * i) < C;
    const float d1 = ok ? u[i] - mu1 : 0.f, d2 = ok ? v[i] - mu2 : 0.f;
    q1 += d1 * d1; q2 += d2 * d2;
  }
  const float r1 = rsqrtf(block_sum(q1, red) * inv + COL_EPS), r2 = rsqrtf(block_sum(q2, red) * inv + COL_EPS);
  const float* xb = x + (long)blockIdx.x * x_bs;
  float* yb = y + (long)blockIdx.x * y_bs;
#pragma unroll
  for (int i = 0; i < COL_CPT; ++i) {
    const int c = threadIdx.x + 256 * i;
    if (c < C) {
      const float n1 = (u[i] - mu1) * r1 * g1[c] + b1[c], n2 = (v[i] - mu2) * r2 * g2[c] + b2[c];
      const float sg = col_sigmoid(n1);
      yb[c] = sg * n2 + (1.f - sg) * xb[c];
    }
  }
}
extern "C" int ssv_column_ln_act(const float* x, long x_bs, const float* gamma, const float* beta, float* y, long y_bs, int B, int C, int act,
                                 ssv_stream_t stream) {
  SSV_CHECK(x && gamma && beta && y && B > 0 && C > 0 && act >= 0 && act <= 2, SSV_BAD_SHAPE, "column_ln_act: bad argument B=%d C=%d", B, C);
  SSV_CHECK(C <= 256 * COL_CPT, SSV_UNSUPPORTED, "column_ln_act: %d channels (max %d)", C, 256 * COL_CPT);
  hipLaunchKernelGGL(column_ln_act_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, x_bs, gamma, beta, y, y_bs, C, act);
  return ssv_check_launch("column_ln_act");
}
extern "C" int ssv_column_gate(const float* h, const float* x, long x_bs, const float* g1, const float* b1, const float* g2, const float* b2,
                               float* y, long y_bs, int B, int C, ssv_stream_t stream) {
  SSV_CHECK(h && x && g1 && b1 && g2 && b2 && y && B > 0 && C > 0, SSV_BAD_SHAPE, "column_gate: bad argument B=%d C=%d", B, C);
  SSV_CHECK(C <= 256 * COL_CPT, SSV_UNSUPPORTED, "column_gate: %d channels (max %d)", C, 256 * COL_CPT);
  hipLaunchKernelGGL(column_gate_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, h, x, x_bs, g1, b1, g2, b2, y, y_bs, C);
  return ssv_check_launch("column_gate");
}

// ---- attention for one new frame (models/TTSModel.py:281-295), one workgroup per batch item ---------------------------
// logits[n] = <K[:, n], q> / sqrt(d), positions outside [pma, pma+2] set to -2^32 (as the reference does before its softmax),
// softmax over n -> column t of A, first arg-max -> pma, r = V a, and rq = [r ; q] is the decoder's input column.
// Softmax and arg-max as in attention_step_kernel (attn.hip); the logits are summed in four channel quarters.
#define ACOL_MAXN 1024
__global__ __launch_bounds__(256) void attention_column_kernel(const float* __restrict__ kv, long kv_bs, const float* __restrict__ q,
                                                               int64_t* __restrict__ pma, float* __restrict__ a, int a_T,
                                                               const int* __restrict__ t_dev, float* __restrict__ rq, int d, int N, float scale) {
  __shared__ float logit[ACOL_MAXN];
  __shared__ float part[4][ACOL_MAXN];
  __shared__ float red[4];
  __shared__ int redi[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = t_dev[0];
  const float* kb = kv + (long)b * kv_bs;
  const float* vb = kb + (long)d * N;
  const float* qb = q + (long)b * d;
  const int64_t p0 = pma[b];
  // logits: wave w sums channels [w*d/4, (w+1)*d/4) for every n (lanes along n: coalesced K reads); the four partial sums
  // are combined in a fixed order
  {
    const int c0 = (int)((long)wave * d / 4), c1 = (int)((long)(wave + 1) * d / 4);
    for (int n = lane; n < N; n += 64) {
      float s0 = 0.f, s1 = 0.f;
      int c = c0;
      for (; c + 1 < c1; c += 2) { s0 = fmaf(kb[(long)c * N + n], qb[c], s0); s1 = fmaf(kb[(long)(c + 1) * N + n], qb[c + 1], s1); }
      if (c < c1) s0 = fmaf(kb[(long)c * N + n], qb[c], s0);
      part[wave][n] = s0 + s1;
    }
  }
  __syncthreads();
  for (int n = tid; n < N; n += 256) {
    float s = ((part[0][n] + part[1][n]) + (part[2][n] + part[3][n])) * scale;
    if (n < p0 || n >= p0 + 3) s = -4294967296.f;
    logit[n] = s;
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int n = tid; n < N; n += 256) mx = fmaxf(mx, logit[n]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float sum = 0.f;
  for (int n = tid; n < N; n += 256) { const float e = expf(logit[n] - mx); logit[n] = e; sum += e; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  sum = (red[0] + red[1]) + (red[2] + red[3]);
  const float inv = 1.f / sum;
  float best = -1.f; int bi = N;
  for (int n = tid; n < N; n += 256) {
    const float p = logit[n] * inv;
    logit[n] = p;
    if (col < a_T) a[((long)b * N + n) * a_T + col] = p;
    if (p > best) { best = p; bi = n; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  __syncthreads();
  if (lane == 0) { red[wave] = best; redi[wave] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (red[w] > best || (red[w] == best && redi[w] < bi)) { best = red[w]; bi = redi[w]; }
    pma[b] = bi;
  }
  float* rqb = rq + (long)b * 2 * d;
  __syncthreads();                                   // every thread's share of the column is in logit[]
  for (int c = tid; c < d; c += 256) {               // r[c] = sum_n V[c][n] a[n]: a thread per channel, a[] broadcast from LDS
    const float* vr = vb + (long)c * N;
    float s0 = 0.f, s1 = 0.f;
    int n = 0;
    for (; n + 1 < N; n += 2) { s0 = fmaf(vr[n], logit[n], s0); s1 = fmaf(vr[n + 1], logit[n + 1], s1); }
    if (n < N) s0 = fmaf(vr[n], logit[n], s0);
    rqb[c] = s0 + s1;
  }
  for (int c = tid; c < d; c += 256) rqb[d + c] = qb[c];
}

extern "C" int ssv_attention_column(const float* kv, long kv_bs, const float* q, int64_t* pma, float* a, int a_T, const int* t_dev,
                                    float* rq, int B, int d, int N, ssv_stream_t stream) {
  SSV_CHECK(kv && q && pma && a && t_dev && rq && B > 0 && d > 0 && N > 0 && a_T > 0, SSV_BAD_SHAPE, "attention_column: bad argument B=%d d=%d N=%d", B, d, N);
  SSV_CHECK(N <= ACOL_MAXN, SSV_UNSUPPORTED, "attention_column: N=%d > %d", N, ACOL_MAXN);
  hipLaunchKernelGGL(attention_column_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, kv, kv_bs, q, pma, a, a_T, t_dev, rq, d, N, 1.f / sqrtf((float)d));
  return ssv_check_launch("attention_column");
}

// ---- end of a step: Y[:, :, t] = y_cur, mel_cur = y_cur (the frame just synthesised is the next input), t += 1 -------------
__global__ __launch_bounds__(256) void synth_column_feed_kernel(const float* __restrict__ y_cur, float* __restrict__ Y, float* __restrict__ mel_cur,
                                                                const int* __restrict__ t_dev, int n, int T) {
  const int i = blockIdx.x * 256 + threadIdx.x;            // (b, f)
  const int t = t_dev[0];
  if (i < n) {
    const float v = y_cur[i];
    if (t < T) Y[(long)i * T + t] = v;
    mel_cur[i] = v;
  }
}
__global__ void synth_column_inc_kernel(int* t_dev) { t_dev[0] += 1; }
extern "C" int ssv_synth_column_advance(const float* y_cur, float* Y, float* mel_cur, int* t_dev, int B, int F, int T, ssv_stream_t stream) {
  SSV_CHECK(y_cur && Y && mel_cur && t_dev && B > 0 && F > 0 && T > 0, SSV_BAD_SHAPE, "synth_column_advance: bad argument");
  hipLaunchKernelGGL(synth_column_feed_kernel, dim3(ssv_cdiv((long)B * F, 256)), dim3(256), 0, (hipStream_t)stream, y_cur, Y, mel_cur, (const int*)t_dev, B * F, T);
  SSV_TRY(ssv_check_launch("synth_column_feed"));
  hipLaunchKernelGGL(synth_column_inc_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, t_dev);
  return ssv_check_launch("synth_column_inc");
}
