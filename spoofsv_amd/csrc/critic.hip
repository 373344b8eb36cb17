// Elementwise / pooling / penalty kernels of the WGAN-GP critics (gfx950): what models/discriminator.py:24-41 does between its
// convolutions and LayerNorms -- dropout (p = 0.05, always active: the reference never calls disc.eval()), leaky-ReLU(0.05),
// AvgPool1d -- and the gradient-penalty reduction of train/adversarial_wasserstein_gp.py:305-308.  All of them are
// piecewise-linear or linear maps, so each is closed under differentiation with ONE extra kernel (multiply by the saved
// derivative factor / pool <-> un-pool), which is what the double backward of the penalty needs.  HBM-bound, tiny tensors.
#include "ssv_common.h"

// ---- Philox4x32-10 (counter-based: no RNG state to carry; key = the device-side call counter, counter = element index / 4)
__device__ __forceinline__ uint4 philox4x32_10(uint4 c, uint2 k) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
    k.x += 0x9E3779B9u; k.y += 0xBB67AE85u;
  }
  return c;
}

// y = leaky_relu(x, slope) * keep / (1 - p),  keep ~ Bernoulli(1 - p) per element;  d = the factor y / x (saved: the op's
// derivative, constant in x almost everywhere).  slope = 1: plain dropout; p = 0: plain leaky-ReLU.
__global__ __launch_bounds__(256) void act_dropout_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ d, long n,
                                                              float slope, float p, const unsigned long long* __restrict__ ctr, unsigned seed) {
  const long i4 = (long)blockIdx.x * 256 + threadIdx.x;       // group of 4 consecutive elements
  const long i0 = i4 * 4;
  if (i0 >= n) return;
  float keep[4] = {1.f, 1.f, 1.f, 1.f};
  if (p > 0.f) {
    const unsigned long long call = ctr ? ctr[0] : 0ull;
    const uint4 r = philox4x32_10(make_uint4((unsigned)i4, (unsigned)(i4 >> 32), (unsigned)call, (unsigned)(call >> 32)), make_uint2(seed, 0x5f3759dfu));
    const float inv = 1.f / (1.f - p);
    const unsigned thr = (unsigned)((double)p * 4294967296.0);
    keep[0] = r.x >= thr ? inv : 0.f; keep[1] = r.y >= thr ? inv : 0.f; keep[2] = r.z >= thr ? inv : 0.f; keep[3] = r.w >= thr ? inv : 0.f;
  }
  if (i0 + 3 < n && ((((size_t)x) | ((size_t)y) | ((size_t)d)) & 15) == 0) {       // the group as one 16-byte access per array
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + i0);
    f32x4 f, o;
#pragma unroll
    for (int q = 0; q < 4; ++q) { f[q] = (v[q] > 0.f ? 1.f : slope) * keep[q]; o[q] = v[q] * f[q]; }
    *reinterpret_cast<f32x4*>(d + i0) = f;
    *reinterpret_cast<f32x4*>(y + i0) = o;
    return;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const long i = i0 + q;
    if (i < n) {
      const float v = x[i];
      const float f = (v > 0.f ? 1.f : slope) * keep[q];
      d[i] = f;
      y[i] = v * f;
    }
  }
}
__global__ void counter_inc_kernel(unsigned long long* ctr) { ctr[0] += 1ull; }
extern "C" int ssv_act_dropout_fwd(const float* x, float* y, float* d, long n, float slope, float p, unsigned long long* ctr_dev, unsigned seed,
                                   ssv_stream_t stream) {
  SSV_CHECK(x && y && d && n > 0 && p >= 0.f && p < 1.f, SSV_BAD_SHAPE, "act_dropout_fwd: bad argument n=%ld p=%f", n, p);
  SSV_CHECK(p == 0.f || ctr_dev, SSV_BAD_SHAPE, "act_dropout_fwd: dropout needs the device-side call counter");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(act_dropout_fwd_kernel, dim3(ssv_cdiv(ssv_cdiv(n, 4), 256)), dim3(256), 0, st, x, y, d, n, slope, p, ctr_dev, seed);
  SSV_TRY(ssv_check_launch("act_dropout_fwd"));
  if (p > 0.f) {      // the next call draws a new mask -- also when this call is replayed from a hipGraph
    hipLaunchKernelGGL(counter_inc_kernel, dim3(1), dim3(1), 0, st, ctr_dev);
    return ssv_check_launch("counter_inc");
  }
  return 0;
}
// y = x * d (the backward of the op above, and the backward of that)
// (four elements per thread as one 16-byte access when the three pointers allow it: 32 launches per pair of critic iterations, most of
//  them over 10 M elements, ran one 4-byte element per thread)
__global__ __launch_bounds__(256) void mul_kernel(const float* __restrict__ x, const float* __restrict__ d, float* __restrict__ y, long n) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n && ((((size_t)x) | ((size_t)d) | ((size_t)y)) & 15) == 0) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + i), b = *reinterpret_cast<const f32x4*>(d + i);
    *reinterpret_cast<f32x4*>(y + i) = a * b;
  } else {
    for (long e = i; e < n && e < i + 4; ++e) y[e] = x[e] * d[e];
  }
}
extern "C" int ssv_mul(const float* x, const float* d, float* y, long n, ssv_stream_t stream) {
  SSV_CHECK(x && d && y && n > 0, SSV_BAD_SHAPE, "mul: bad argument");
  hipLaunchKernelGGL(mul_kernel, dim3(ssv_cdiv(ssv_cdiv(n, 4), 256)), dim3(256), 0, (hipStream_t)stream, x, d, y, n);
  return ssv_check_launch("mul");
}

// ---- nn.AvgPool1d(kernel_size = k) (stride k, no padding: Lo = L / k, a tail of L % k columns is dropped) and its adjoint
__global__ __launch_bounds__(256) void avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long rows, int L, int k, int Lo) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * Lo) return;
  const long r = i / Lo;
  const int j = (int)(i % Lo);
  const float* p = x + r * L + (long)j * k;
  float s = 0.f;
  for (int q = 0; q < k; ++q) s += p[q];
  y[i] = s / (float)k;
}
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long rows, int L, int k, int Lo) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * L) return;
  const long r = i / L;
  const int t = (int)(i % L), j = t / k;
  dx[i] = j < Lo ? dy[r * Lo + j] / (float)k : 0.f;
}
extern "C" int ssv_avgpool1d_fwd(const float* x, float* y, long rows, int L, int k, ssv_stream_t stream) {
  SSV_CHECK(x && y && rows > 0 && L > 0 && k > 0 && L / k > 0, SSV_BAD_SHAPE, "avgpool1d_fwd: bad argument rows=%ld L=%d k=%d", rows, L, k);
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(ssv_cdiv(rows * (L / k), 256)), dim3(256), 0, (hipStream_t)stream, x, y, rows, L, k, L / k);
  return ssv_check_launch("avgpool1d_fwd");
}
extern "C" int ssv_avgpool1d_bwd(const float* dy, float* dx, long rows, int L, int k, ssv_stream_t stream) {
  SSV_CHECK(dy && dx && rows > 0 && L > 0 && k > 0 && L / k > 0, SSV_BAD_SHAPE, "avgpool1d_bwd: bad argument rows=%ld L=%d k=%d", rows, L, k);
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(ssv_cdiv(rows * L, 256)), dim3(256), 0, (hipStream_t)stream, dy, dx, rows, L, k, L / k);
  return ssv_check_launch("avgpool1d_bwd");
}

// ---- gradient penalty, train/adversarial_wasserstein_gp.py:305-308: loss = mean_b lam * (||g_b||_2 - 1)^2 over the per-sample
// gradients g (B, n).  Two launches, fixed summation order: partial sums of squares per (sample, 64 K-element chunk), then per
// sample the norm, its loss term and the coefficient c_b = 2 lam (||g_b|| - 1) / (B ||g_b||) of the backward  d loss / d g = c_b g.
#define GP_CHUNK 65536
__global__ __launch_bounds__(256) void gp_partial_kernel(const float* __restrict__ g, float* __restrict__ part, long n, int nchunk) {
  __shared__ float red[4];
  const int b = blockIdx.y, ck = blockIdx.x;
  const float* p = g + (long)b * n;
  const long lo = (long)ck * GP_CHUNK, hi = lo + GP_CHUNK < n ? lo + GP_CHUNK : n;
  float s = 0.f;
  for (long i = lo + threadIdx.x; i < hi; i += 256) s += p[i] * p[i];
  s = ssv_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[(long)b * nchunk + ck] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(64) void gp_finish_kernel(const float* __restrict__ part, float* __restrict__ loss, float* __restrict__ coef, int B, int nchunk, float lam) {
  float acc = 0.f;                                  // one wave; lane b < B handles sample b, samples beyond 64 in further rounds
  for (int b0 = 0; b0 < B; b0 += 64) {
    const int b = b0 + threadIdx.x;
    float term = 0.f;
    if (b < B) {
      float s = 0.f;
      for (int c = 0; c < nchunk; ++c) s += part[(long)b * nchunk + c];
      const float nrm = sqrtf(s);
      term = lam * (nrm - 1.f) * (nrm - 1.f);
      coef[b] = 2.f * lam * (nrm - 1.f) / ((float)B * fmaxf(nrm, 1e-30f));
    }
    acc += ssv_wave_sum(term);
  }
  if (threadIdx.x == 0) loss[0] = acc / (float)B;
}
extern "C" size_t ssv_grad_penalty_workspace(int B, long n) { return (size_t)B * ssv_cdiv(n, GP_CHUNK) * sizeof(float) + 256; }
extern "C" int ssv_grad_penalty_fwd(const float* g, float* loss, float* coef, int B, long n, float lam, void* ws, size_t ws_bytes, ssv_stream_t stream) {
  SSV_CHECK(g && loss && coef && B > 0 && B <= 65535 && n > 0, SSV_BAD_SHAPE, "grad_penalty_fwd: bad argument B=%d n=%ld", B, n);
  SSV_CHECK(ws && ws_bytes >= ssv_grad_penalty_workspace(B, n), SSV_BAD_SHAPE, "grad_penalty_fwd: workspace too small");
  const int nchunk = ssv_cdiv(n, GP_CHUNK);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gp_partial_kernel, dim3(nchunk, B), dim3(256), 0, st, g, (float*)ws, n, nchunk);
  SSV_TRY(ssv_check_launch("gp_partial"));
  hipLaunchKernelGGL(gp_finish_kernel, dim3(1), dim3(64), 0, st, (const float*)ws, loss, coef, B, nchunk, lam);
  return ssv_check_launch("gp_finish");
}
// dg(b, i) = gout[0] * coef[b] * g(b, i)
__global__ __launch_bounds__(256) void gp_bwd_kernel(const float* __restrict__ g, const float* __restrict__ coef, const float* __restrict__ gout,
                                                     float* __restrict__ dg, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dg[(long)blockIdx.y * n + i] = gout[0] * coef[blockIdx.y] * g[(long)blockIdx.y * n + i];
}
extern "C" int ssv_grad_penalty_bwd(const float* g, const float* coef, const float* gout, float* dg, int B, long n, ssv_stream_t stream) {
  SSV_CHECK(g && coef && gout && dg && B > 0 && B <= 65535 && n > 0, SSV_BAD_SHAPE, "grad_penalty_bwd: bad argument");
  hipLaunchKernelGGL(gp_bwd_kernel, dim3(ssv_cdiv(n, 256), B), dim3(256), 0, (hipStream_t)stream, g, coef, gout, dg, n);
  return ssv_check_launch("grad_penalty_bwd");
}
