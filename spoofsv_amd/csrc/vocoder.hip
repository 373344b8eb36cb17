// Vocoder / spectrogram front-end kernels, gfx950.  All HBM-bound elementwise / gather work; the DFTs
// themselves are 1x1 convolutions with windowed Fourier bases on the split-bf16 GEMM (api.hip, ssv_conv1d_fwd).
//
// Spectra are (B, 2F, T) float32: rows [0, F) real parts, rows [F, 2F) imaginary parts, T frames (fastest).
// Frame matrices are (B, N, T): row c = sample c of every frame, so they are the (B, C, T) operands of the 1x1 convs.
#include "ssv_common.h"

// ---- Griffin-Lim phase step: a = reb - alpha * tprev;  proj = mag * a / (|a| + 1e-16) --------------------------------
// librosa 0.7.0 core.griffinlim loop body (`angles[:] = rebuilt - momentum/(1+momentum)*tprev; angles /= abs(angles)+1e-16`)
// fused with the `S * angles` product of the next inverse transform.  tprev == NULL: zero (first iteration).
__global__ __launch_bounds__(256) void gl_project_kernel(const float* __restrict__ mag, const float* __restrict__ reb,
                                                         const float* __restrict__ tprev, float alpha,
                                                         float* __restrict__ proj, long FT) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= FT) return;
  const long o = (long)blockIdx.y * 2 * FT + i;
  float ar = reb[o], ai = reb[o + FT];
  if (tprev) { ar -= alpha * tprev[o]; ai -= alpha * tprev[o + FT]; }
  const float m = mag[(long)blockIdx.y * FT + i] / (sqrtf(ar * ar + ai * ai) + 1e-16f);
  proj[o] = ar * m;
  proj[o + FT] = ai * m;
}

__global__ __launch_bounds__(256) void complex_abs_kernel(const float* __restrict__ spec, float* __restrict__ mag, long FT) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= FT) return;
  const long o = (long)blockIdx.y * 2 * FT + i;
  const float r = spec[o], q = spec[o + FT];
  mag[(long)blockIdx.y * FT + i] = sqrtf(r * r + q * q);
}

// ---- overlap-add -------------------------------------------------------------------------------------------------
// ola(m) = sum over frames t' with 0 <= m - t'*hop < N of fr[m - t'*hop][t'], frames added in increasing t' (the order of
// librosa's istft loop), times inv_env[m] = 1 / window_sumsquare (1 where the envelope is ~0).
__device__ __forceinline__ float ola_at(const float* __restrict__ frb, const float* __restrict__ inv_env, int m, int N, int T, int hop) {
  int t1 = m / hop;
  if (t1 > T - 1) t1 = T - 1;
  int t0 = (m - N + hop) / hop;          // ceil((m - N + 1) / hop) for m - N + 1 > 0
  if (m - N + 1 <= 0) t0 = 0;
  float acc = 0.f;
  for (int t = t0; t <= t1; ++t) acc += frb[(long)(m - t * hop) * T + t];
  return acc * inv_env[m];
}

// One Griffin-Lim round trip between the two DFT GEMMs: overlap-add the inverse frames, normalise, drop the N/2 centre
// padding (istft), reflect-pad by N/2 and cut into frames again (stft) -- without materialising the waveform.
// out[b][c][t] = ypad[t*hop + c].  Interior reads are fr[c - d*hop][t + d], d = 0..N/hop-1: coalesced along t.
__global__ __launch_bounds__(256) void ola_frames_kernel(const float* __restrict__ fr, const float* __restrict__ inv_env,
                                                         float* __restrict__ out, int N, int T, int hop) {
  const int t = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
  if (t >= T) return;
  const int len = hop * (T - 1);
  int j = t * hop + c - N / 2;           // index into the trimmed waveform
  if (j < 0) j = -j;
  else if (j >= len) j = 2 * (len - 1) - j;
  const long bo = (long)blockIdx.z * N * T;
  out[bo + (long)c * T + t] = ola_at(fr + bo, inv_env, j + N / 2, N, T, hop);
}

__global__ __launch_bounds__(256) void ola_signal_kernel(const float* __restrict__ fr, const float* __restrict__ inv_env,
                                                         float* __restrict__ y, int N, int T, int hop) {
  const int len = hop * (T - 1);
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= len) return;
  y[(long)blockIdx.y * len + n] = ola_at(fr + (long)blockIdx.y * N * T, inv_env, n + N / 2, N, T, hop);
}

// stft framing of a waveform: fr[b][c][t] = reflect_pad(y, N/2)[t*hop + c]
__global__ __launch_bounds__(256) void frame_signal_kernel(const float* __restrict__ y, float* __restrict__ fr, int n, int N, int T, int hop) {
  const int t = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
  if (t >= T) return;
  int j = t * hop + c - N / 2;
  if (j < 0) j = -j;
  else if (j >= n) j = 2 * (n - 1) - j;
  fr[((long)blockIdx.z * N + c) * T + t] = y[(long)blockIdx.z * n + j];
}

// ---- per-row max, (x / max)^p * s ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rowmax_kernel(const float* __restrict__ x, float* __restrict__ out, long n) {
  __shared__ float red[256];
  const float* xb = x + (long)blockIdx.x * n;
  float m = -INFINITY;
  for (long i = threadIdx.x; i < n; i += 256) m = fmaxf(m, xb[i]);
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void scale_pow_kernel(const float* __restrict__ x, const float* __restrict__ rowmax,
                                                        float* __restrict__ y, float p, float s, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long o = (long)blockIdx.y * n + i;
  float v = x[o] / rowmax[blockIdx.y];
  if (p != 1.f) v = powf(v, p);
  y[o] = v * s;
}

// ---- de-emphasis: y[n] = x[n] + a*y[n-1]  (scipy.signal.lfilter([1], [1, -a], x), synthesize.py:145) -----------------
// One workgroup per utterance; thread j owns a contiguous chunk.  Pass 1: chunk-local recurrence from zero.  Thread 0
// chains the 256 chunk ends (end_j + a^len_j * carry_j).  Pass 2: the recurrence again from the true carry-in.  The
// recurrence runs in double, as scipy's does for the reference's float64 coefficients.
__global__ __launch_bounds__(256) void deemphasis_kernel(const float* __restrict__ x, float* __restrict__ y, double a, int n) {
  __shared__ double ends[256];
  __shared__ double carry[256];
  const float* xb = x + (long)blockIdx.x * n;
  float* yb = y + (long)blockIdx.x * n;
  const int chunk = (n + 255) / 256;
  const int lo = threadIdx.x * chunk, hi = min(n, lo + chunk);
  double v = 0.0;
  for (int i = lo; i < hi; ++i) v = (double)xb[i] + a * v;
  ends[threadIdx.x] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    double c = 0.0;
    for (int j = 0; j < 256; ++j) {
      carry[j] = c;
      const int l = j * chunk, h = min(n, l + chunk);
      if (h > l) c = ends[j] + pow(a, (double)(h - l)) * c;
    }
  }
  __syncthreads();
  v = carry[threadIdx.x];
  for (int i = lo; i < hi; ++i) { v = (double)xb[i] + a * v; yb[i] = (float)v; }
}

// pre-emphasis y[0] = x[0], y[n] = x[n] - a*x[n-1]  (data/dataset.py:96)
__global__ __launch_bounds__(256) void preemphasis_kernel(const float* __restrict__ x, float* __restrict__ y, float a, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long o = (long)blockIdx.y * n + i;
  y[o] = i ? x[o] - a * x[o - 1] : x[o];
}

// LOG_FEATURE spectrograms.  exp_affine: y = exp(a*x + b) -- the dB de-normalisation, 10^(dB/20) and the reconstruction
// power of synthesize.py:133-135,142 in one pass.  log_norm: y = clip((20 log10(max(1e-5, x)) - ref_db + max_db) / max_db,
// 1e-8, 1), data/dataset.py:101-105.
__global__ __launch_bounds__(256) void exp_affine_kernel(const float* __restrict__ x, float* __restrict__ y, float a, float b, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = expf(a * x[i] + b);
}
__global__ __launch_bounds__(256) void log_norm_kernel(const float* __restrict__ x, float* __restrict__ y, float ref_db, float max_db, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = fminf(fmaxf((20.f * log10f(fmaxf(1e-5f, x[i])) - ref_db + max_db) / max_db, 1e-8f), 1.f);
}

// ---- C ABI ---------------------------------------------------------------------------------------------------------
static bool frames_ok(int B, int N, int T, int hop) {
  return B > 0 && B <= 65535 && N >= 2 && N <= 65535 && N % 2 == 0 && hop > 0 && hop <= N && T >= 2 &&
         (long)hop * (T - 1) > N / 2 && (long)B * N * T < (1L << 40) && (long)hop * (T - 1) + N < (1L << 30);
}

extern "C" int ssv_gl_project(const float* mag, const float* reb, const float* tprev, float alpha, float* proj,
                              int B, int F, int T, ssv_stream_t stream) {
  SSV_CHECK(mag && reb && proj && B > 0 && B <= 65535 && F > 0 && T > 0, SSV_BAD_SHAPE, "gl_project: bad argument B=%d F=%d T=%d", B, F, T);
  const long FT = (long)F * T;
  hipLaunchKernelGGL(gl_project_kernel, dim3(ssv_cdiv(FT, 256), B), dim3(256), 0, (hipStream_t)stream, mag, reb, tprev, alpha, proj, FT);
  return ssv_check_launch("gl_project");
}

extern "C" int ssv_complex_abs(const float* spec, float* mag, int B, int F, int T, ssv_stream_t stream) {
  SSV_CHECK(spec && mag && B > 0 && B <= 65535 && F > 0 && T > 0, SSV_BAD_SHAPE, "complex_abs: bad argument B=%d F=%d T=%d", B, F, T);
  const long FT = (long)F * T;
  hipLaunchKernelGGL(complex_abs_kernel, dim3(ssv_cdiv(FT, 256), B), dim3(256), 0, (hipStream_t)stream, spec, mag, FT);
  return ssv_check_launch("complex_abs");
}

extern "C" int ssv_ola_frames(const float* fr, const float* inv_env, float* out, int B, int N, int T, int hop, ssv_stream_t stream) {
  SSV_CHECK(fr && inv_env && out && fr != out && frames_ok(B, N, T, hop), SSV_BAD_SHAPE,
            "ola_frames: bad argument B=%d N=%d T=%d hop=%d (need hop*(T-1) > N/2)", B, N, T, hop);
  hipLaunchKernelGGL(ola_frames_kernel, dim3(ssv_cdiv(T, 256), N, B), dim3(256), 0, (hipStream_t)stream, fr, inv_env, out, N, T, hop);
  return ssv_check_launch("ola_frames");
}

extern "C" int ssv_ola_signal(const float* fr, const float* inv_env, float* y, int B, int N, int T, int hop, ssv_stream_t stream) {
  SSV_CHECK(fr && inv_env && y && frames_ok(B, N, T, hop), SSV_BAD_SHAPE,
            "ola_signal: bad argument B=%d N=%d T=%d hop=%d (need hop*(T-1) > N/2)", B, N, T, hop);
  hipLaunchKernelGGL(ola_signal_kernel, dim3(ssv_cdiv((long)hop * (T - 1), 256), B), dim3(256), 0, (hipStream_t)stream, fr, inv_env, y, N, T, hop);
  return ssv_check_launch("ola_signal");
}

extern "C" int ssv_frame_signal(const float* y, float* fr, int B, int n, int N, int T, int hop, ssv_stream_t stream) {
  SSV_CHECK(y && fr && B > 0 && B <= 65535 && N >= 2 && N <= 65535 && N % 2 == 0 && hop > 0 && n > N / 2 && T == 1 + n / hop,
            SSV_BAD_SHAPE, "frame_signal: bad argument B=%d n=%d N=%d T=%d hop=%d (need n > N/2, T = 1 + n/hop)", B, n, N, T, hop);
  hipLaunchKernelGGL(frame_signal_kernel, dim3(ssv_cdiv(T, 256), N, B), dim3(256), 0, (hipStream_t)stream, y, fr, n, N, T, hop);
  return ssv_check_launch("frame_signal");
}

extern "C" int ssv_rowmax(const float* x, float* out, int B, long n, ssv_stream_t stream) {
  SSV_CHECK(x && out && B > 0 && n > 0, SSV_BAD_SHAPE, "rowmax: bad argument B=%d n=%ld", B, n);
  hipLaunchKernelGGL(rowmax_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, out, n);
  return ssv_check_launch("rowmax");
}

extern "C" int ssv_scale_pow(const float* x, const float* rowmax, float* y, float p, float s, int B, long n, ssv_stream_t stream) {
  SSV_CHECK(x && rowmax && y && B > 0 && B <= 65535 && n > 0, SSV_BAD_SHAPE, "scale_pow: bad argument B=%d n=%ld", B, n);
  hipLaunchKernelGGL(scale_pow_kernel, dim3(ssv_cdiv(n, 256), B), dim3(256), 0, (hipStream_t)stream, x, rowmax, y, p, s, n);
  return ssv_check_launch("scale_pow");
}

extern "C" int ssv_deemphasis(const float* x, float* y, double a, int B, int n, ssv_stream_t stream) {
  SSV_CHECK(x && y && B > 0 && n > 0, SSV_BAD_SHAPE, "deemphasis: bad argument B=%d n=%d", B, n);
  hipLaunchKernelGGL(deemphasis_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, y, a, n);
  return ssv_check_launch("deemphasis");
}

extern "C" int ssv_preemphasis(const float* x, float* y, float a, int B, int n, ssv_stream_t stream) {
  SSV_CHECK(x && y && x != y && B > 0 && B <= 65535 && n > 0, SSV_BAD_SHAPE, "preemphasis: bad argument B=%d n=%d", B, n);
  hipLaunchKernelGGL(preemphasis_kernel, dim3(ssv_cdiv(n, 256), B), dim3(256), 0, (hipStream_t)stream, x, y, a, n);
  return ssv_check_launch("preemphasis");
}

extern "C" int ssv_exp_affine(const float* x, float* y, float a, float b, long n, ssv_stream_t stream) {
  SSV_CHECK(x && y && n > 0 && n < (1L << 39), SSV_BAD_SHAPE, "exp_affine: bad argument n=%ld", n);
  hipLaunchKernelGGL(exp_affine_kernel, dim3(ssv_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, a, b, n);
  return ssv_check_launch("exp_affine");
}

extern "C" int ssv_log_norm(const float* x, float* y, float ref_db, float max_db, long n, ssv_stream_t stream) {
  SSV_CHECK(x && y && n > 0 && n < (1L << 39) && max_db > 0.f, SSV_BAD_SHAPE, "log_norm: bad argument n=%ld max_db=%g", n, (double)max_db);
  hipLaunchKernelGGL(log_norm_kernel, dim3(ssv_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, ref_db, max_db, n);
  return ssv_check_launch("log_norm");
}
