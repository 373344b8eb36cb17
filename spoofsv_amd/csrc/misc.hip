// Small HBM-bound kernels around the GEMMs: weight transposition, slab reduction, row sums, the
// text embedding, the spectrogram / guided-attention losses and multi-tensor Adam (gfx950).
#include "ssv_common.h"
#include "../../include/ssv_hip.h"

// ---- out[i] = sum_z slabs[z*stride + i] (fixed order) ---------------------------------------------
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ s, float* __restrict__ out, long n, int Z, long stride) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float a[4] = {0.f, 0.f, 0.f, 0.f};                         // four slabs in flight
  int z = 0;
  for (; z + 3 < Z; z += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] += s[(long)(z + u) * stride + i];
  }
  for (; z < Z; ++z) a[0] += s[(long)z * stride + i];
  out[i] = (a[0] + a[1]) + (a[2] + a[3]);
}
int ssv_launch_reduce_slabs(const float* slabs, float* out, long n, int Z, long stride, hipStream_t st) {
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3(ssv_cdiv(n, 256)), dim3(256), 0, st, slabs, out, n, Z, stride);
  return ssv_check_launch("reduce_slabs");
}

// Slabs stored [z][m][j][c] (channels contiguous: the weight-gradient kernel's lanes write 64-byte runs), output in the
// weight layout out[m][c][j] = sum_z slab[z][m][j][c].  A thread owns one (m, c) and all KT taps: the Z reads per tap are
// coalesced along c, and the workgroup's 256 * KT results are one contiguous run of `out`, written as 16-byte vectors
// through LDS.  (One thread per slab element wrote 4 bytes at a 12-byte stride, the three taps of a line from three
// different waves: WRITE_SIZE was 3x the gradient.)  Summation order per element: pairs of slabs, as before.
__device__ __forceinline__ void reduce_perm_block(const float* __restrict__ s, float* __restrict__ out, int Nc, int KT, long n, int Z, int block,
                                                  float* __restrict__ stage /* [256 * 3] */) {
  const long P = n / KT;                                      // (m, c) pairs
  const long q = (long)block * 256 + threadIdx.x;
  if (KT == 1) {
    if (q >= P) return;
    float a[8];                                               // eight slabs in flight (two before: Z / 2 memory round trips in a row, Z = 16 .. 32 for the k = 1 layers)
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = 0.f;
    int z = 0;
    for (; z + 7 < Z; z += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += s[(long)(z + u) * n + q];
    }
    for (; z < Z; ++z) a[0] += s[(long)z * n + q];
    out[q] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    return;
  }
  if (q < P) {
    const long m = q / Nc;
    const int c = (int)(q % Nc);
    if (KT == 3) {                                            // the three taps side by side: three independent chains of loads in flight
      const long i0 = (m * 3) * Nc + c, i1 = i0 + Nc, i2 = i1 + Nc;
      float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f, c0 = 0.f, c1 = 0.f;
      int z = 0;
#pragma unroll 2
      for (; z + 1 < Z; z += 2) {
        const float* __restrict__ s0 = s + (long)z * n;
        const float* __restrict__ s1 = s0 + n;
        a0 += s0[i0]; a1 += s1[i0]; b0 += s0[i1]; b1 += s1[i1]; c0 += s0[i2]; c1 += s1[i2];
      }
      if (z < Z) { const float* __restrict__ s0 = s + (long)z * n; a0 += s0[i0]; b0 += s0[i1]; c0 += s0[i2]; }
      stage[threadIdx.x * 3] = a0 + a1; stage[threadIdx.x * 3 + 1] = b0 + b1; stage[threadIdx.x * 3 + 2] = c0 + c1;
    } else {
      for (int j = 0; j < KT; ++j) {
        const long i = (m * KT + j) * Nc + c;
        float a0 = 0.f, a1 = 0.f;
        int z = 0;
        for (; z + 1 < Z; z += 2) { a0 += s[(long)z * n + i]; a1 += s[(long)(z + 1) * n + i]; }
        if (z < Z) a0 += s[(long)z * n + i];
        stage[threadIdx.x * KT + j] = a0 + a1;
      }
    }
  }
  __syncthreads();
  const long o0 = (long)block * 256 * KT;                      // first output element of this workgroup
  const long cnt = (P - (long)block * 256 < 256 ? P - (long)block * 256 : 256) * KT;
  float* __restrict__ dst = out + o0;
  if ((reinterpret_cast<size_t>(dst) & 15) == 0) {
    for (long f = threadIdx.x * 4; f + 3 < cnt; f += 1024) *reinterpret_cast<float4*>(dst + f) = *reinterpret_cast<const float4*>(stage + f);
    for (long f = (cnt & ~3L) + threadIdx.x; f < cnt; f += 256) dst[f] = stage[f];
  } else {
    for (long f = threadIdx.x; f < cnt; f += 256) dst[f] = stage[f];
  }
}
__global__ __launch_bounds__(256) void reduce_slabs_perm_kernel(const float* __restrict__ s, float* __restrict__ out, int Nc, int KT, long n, int Z) {
  __shared__ __attribute__((aligned(16))) float stage[256 * 3];
  reduce_perm_block(s, out, Nc, KT, n, Z, (int)blockIdx.x, stage);
}
int ssv_launch_reduce_slabs_perm(const float* slabs, float* out, int M, int Nc, int KT, int Z, hipStream_t st) {
  const long n = (long)M * Nc * KT;
  hipLaunchKernelGGL(reduce_slabs_perm_kernel, dim3(ssv_cdiv((long)M * Nc, 256)), dim3(256), 0, st, slabs, out, Nc, KT, n, Z);
  return ssv_check_launch("reduce_slabs_perm");
}

// Rows rg, rg + 8, rg + 16, ... of column i of the partial-row matrix part[nblk][n2], sixteen loads in flight per trip (four before: up to
// 768 rows are 24 memory round trips in a row per thread at 8 row groups -- the row half was what these launches waited for), fixed order.
template <typename P>
__device__ __forceinline__ float ssv_fold_rows(P part, int n2, int nblk, int i, int rg) {
  float a[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) a[u] = 0.f;
  int k = rg;
  for (; k + 120 < nblk; k += 128) {
#pragma unroll
    for (int u = 0; u < 16; ++u) a[u] += part[(long)(k + 8 * u) * n2 + i];
  }
  for (; k < nblk; k += 8) a[0] += part[(long)k * n2 + i];
  return (((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]))) + (((a[8] + a[9]) + (a[10] + a[11])) + ((a[12] + a[13]) + (a[14] + a[15])));
}
// Both parameter-gradient reductions of one highwayConv backward in ONE launch: workgroups [0, nA) sum the weight-gradient
// slabs (exactly reduce_slabs_perm_kernel), workgroups [nA, nA + nB) sum the LayerNorm / bias partial rows (exactly
// reduce_partials_1_kernel: a workgroup owns 32 outputs, 8 row groups, fixed order).  Same arithmetic per element as the two
// separate kernels -- results are bit-identical -- one launch less per layer (36 highway layers per Text2Mel + SSRN step).
__global__ __launch_bounds__(256) void reduce_pair_kernel(const float* __restrict__ s, float* __restrict__ out, int Nc, int KT, long n, int Z, int nA,
                                                          const float* __restrict__ part, float* __restrict__ pout, int n2, int nblk) {
  __shared__ float red[8][32];
  __shared__ __attribute__((aligned(16))) float stage[256 * 3];
  if ((int)blockIdx.x < nA) {
    reduce_perm_block(s, out, Nc, KT, n, Z, (int)blockIdx.x, stage);
    return;
  }
  const int li = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int i = ((int)blockIdx.x - nA) * 32 + li;
  red[rg][li] = i < n2 ? ssv_fold_rows(part, n2, nblk, i, rg) : 0.f;
  __syncthreads();
  if (rg == 0 && i < n2) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) t += red[r][li];
    pout[i] = t;
  }
}
int ssv_launch_reduce_pair(const float* slabs, float* out, int M, int Nc, int KT, int Z, const float* part, float* pout, int n2, int nblk, hipStream_t st) {
  const long n = (long)M * Nc * KT;
  const int nA = ssv_cdiv((long)M * Nc, 256), nB = ssv_cdiv(n2, 32);
  hipLaunchKernelGGL(reduce_pair_kernel, dim3(nA + nB), dim3(256), 0, st, slabs, out, Nc, KT, n, Z, nA, part, pout, n2, nblk);
  return ssv_check_launch("reduce_pair");
}

// reduce_pair_kernel for every job of a batched weight-gradient launch: blockIdx.y = job.
__global__ __launch_bounds__(256) void reduce_pair_multi_kernel(const ssv_wgrad_job* __restrict__ jobs, const float* __restrict__ slabs, int Nc, int KT, long n,
                                                                int Z, int nA, int n2, int nblk) {
  __shared__ float red[8][32];
  const ssv_wgrad_job jb = jobs[blockIdx.y];
  const float* __restrict__ s = slabs + (long)blockIdx.y * Z * n;
  __shared__ __attribute__((aligned(16))) float stage[256 * 3];
  if ((int)blockIdx.x < nA) {
    reduce_perm_block(s, jb.dw, Nc, KT, n, Z, (int)blockIdx.x, stage);
    return;
  }
  if (!jb.part) return;
  const __attribute__((address_space(1))) float* __restrict__ part = (const __attribute__((address_space(1))) float*)jb.part;   // (table pointer: see ssv_global)
  const int li = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int i = ((int)blockIdx.x - nA) * 32 + li;
  red[rg][li] = i < n2 ? ssv_fold_rows(part, n2, nblk, i, rg) : 0.f;
  __syncthreads();
  if (rg == 0 && i < n2) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) t += red[r][li];
    jb.pgrads[i] = t;
  }
}
int ssv_launch_reduce_pair_multi(const ssv_wgrad_job* jobs, int njobs, const float* slabs, int M, int Nc, int KT, int Z, int n2, int nblk, hipStream_t st) {
  const long n = (long)M * Nc * KT;
  const int nA = ssv_cdiv((long)M * Nc, 256), nB = n2 > 0 ? ssv_cdiv(n2, 32) : 0;
  hipLaunchKernelGGL(reduce_pair_multi_kernel, dim3(nA + nB, njobs), dim3(256), 0, st, jobs, slabs, Nc, KT, n, Z, nA, n2, nblk);
  return ssv_check_launch("reduce_pair_multi");
}

// ---- wt[c][o][j] = w[o][c][j]: weights for the data gradient ---------------------------------------
__global__ __launch_bounds__(256) void pack_wt_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int Cin, int KT) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;   // index into wt
  const long n = (long)Cout * Cin * KT;
  if (i >= n) return;
  const int j = (int)(i % KT);
  const long r = i / KT;
  const int o = (int)(r % Cout), c = (int)(r / Cout);
  wt[i] = w[((long)o * Cin + c) * KT + j];
}
int ssv_launch_pack_wt(const float* w, float* wt, int Cout, int Cin, int KT, hipStream_t st) {
  const long n = (long)Cout * Cin * KT;
  hipLaunchKernelGGL(pack_wt_kernel, dim3(ssv_cdiv(n, 256)), dim3(256), 0, st, w, wt, Cout, Cin, KT);
  return ssv_check_launch("pack_wt");
}

// ---- 1x1 convolution over a length-1 sequence (= nn.Linear on a (B, K) matrix: the speaker-code layers audioEncoder.fc1 / fc2,
// models/TTSModel.py:148-151, 159-160).  The tiled GEMM kernels spend a serial chain of per-item round trips on these 3 MFLOP
// (forward 21 us, weight gradient 44 us per launch in-step); plain fp32 dot products from LDS take a few microseconds.
//   y(b, m) = sum_c w(m, c) x(b, c) + bias(m) + bias_b(b, m):  a thread per output, m fastest (x is one address per item, w a row per lane
//   read 16 bytes at a time and served by the cache: 205 KB in all), four independent partial sums.
//   (First form: 8 x 32 outputs per workgroup with w and x staged in LDS in chunks of 64 -- 32 workgroups, four load / barrier round trips
//   each: 23.7 us per launch in-step against 21 us for the tiled GEMM it replaced.)
// (round 5) EIGHT lanes per output: lane j takes the 16-byte pieces j, j + 8, .. of the weight row (128 contiguous bytes per group and trip) and the
// eight partial sums meet through three lane exchanges.  With a thread per output the launch was 32 workgroups whose every thread walked its 800-byte
// row alone, 50 dependent-issue trips: 29 us for 1.6 MFLOP.
__global__ __launch_bounds__(256) void linear_len1_fwd_kernel(const float* __restrict__ x, long x_bs, const float* __restrict__ w, const float* __restrict__ bias,
                                                              const float* __restrict__ bias_b, long sbb, float* __restrict__ y, long y_bs, int B, int K, int M) {
  const long t = ((long)blockIdx.x * 256 + threadIdx.x) >> 3;
  const int j = threadIdx.x & 7;
  if (t >= (long)B * M) return;                          // (a group of eight leaves or stays together)
  const int m = (int)(t % M), b = (int)(t / M);
  const float* __restrict__ wr = w + (long)m * K;
  const float* __restrict__ xr = x + (long)b * x_bs;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int kdone = 0;
  if ((K & 3) == 0 && (((size_t)w | (size_t)x) & 15) == 0 && (x_bs & 3) == 0) {
    for (int k = 4 * j; k + 3 < K; k += 32) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + k), xv = *reinterpret_cast<const f32x4*>(xr + k);
      a0 = __builtin_fmaf(wv[0], xv[0], a0); a1 = __builtin_fmaf(wv[1], xv[1], a1); a2 = __builtin_fmaf(wv[2], xv[2], a2); a3 = __builtin_fmaf(wv[3], xv[3], a3);
    }
    kdone = K;
  }
  for (int k = kdone + j; k < K; k += 8) a0 = __builtin_fmaf(wr[k], xr[k], a0);
  float acc = (a0 + a1) + (a2 + a3);
  acc += __shfl_xor(acc, 1);
  acc += __shfl_xor(acc, 2);
  acc += __shfl_xor(acc, 4);
  if (j == 0) {
    if (bias) acc += bias[m];
    if (bias_b) acc += bias_b[(long)b * sbb + m];
    y[(long)b * y_bs + m] = acc;
  }
}
int ssv_launch_linear_len1_fwd(const float* x, long x_bs, const float* w, const float* bias, const float* bias_b, long sbb, float* y, long y_bs,
                               int B, int K, int M, hipStream_t st) {
  hipLaunchKernelGGL(linear_len1_fwd_kernel, dim3(ssv_cdiv((long)B * M * 8, 256)), dim3(256), 0, st, x, x_bs, w, bias, bias_b, sbb, y, y_bs, B, K, M);
  return ssv_check_launch("linear_len1_fwd");
}
//   dw(m, c) = sum_b dy(b, m) x(b, c): a thread per entry, c fastest (x coalesced, dy one address per 64 lanes or two).
__global__ __launch_bounds__(256) void linear_len1_wgrad_kernel(const float* __restrict__ dy, long dy_bs, const float* __restrict__ x, long x_bs,
                                                                float* __restrict__ dw, int B, int K, int M) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)M * K) return;
  const int m = (int)(i / K), c = (int)(i % K);
  float acc = 0.f;
  int b = 0;
  for (; b + 3 < B; b += 4) {
    const float d0 = dy[(long)b * dy_bs + m], d1 = dy[(long)(b + 1) * dy_bs + m], d2 = dy[(long)(b + 2) * dy_bs + m], d3 = dy[(long)(b + 3) * dy_bs + m];
    const float x0 = x[(long)b * x_bs + c], x1 = x[(long)(b + 1) * x_bs + c], x2 = x[(long)(b + 2) * x_bs + c], x3 = x[(long)(b + 3) * x_bs + c];
    acc = __builtin_fmaf(d0, x0, acc); acc = __builtin_fmaf(d1, x1, acc); acc = __builtin_fmaf(d2, x2, acc); acc = __builtin_fmaf(d3, x3, acc);
  }
  for (; b < B; ++b) acc = __builtin_fmaf(dy[(long)b * dy_bs + m], x[(long)b * x_bs + c], acc);
  dw[i] = acc;
}
int ssv_launch_linear_len1_wgrad(const float* dy, long dy_bs, const float* x, long x_bs, float* dw, int B, int K, int M, hipStream_t st) {
  hipLaunchKernelGGL(linear_len1_wgrad_kernel, dim3(ssv_cdiv((long)M * K, 256)), dim3(256), 0, st, dy, dy_bs, x, x_bs, dw, B, K, M);
  return ssv_check_launch("linear_len1_wgrad");
}

__global__ __launch_bounds__(256) void fill_kernel(float* p, float v, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}
int ssv_launch_fill(float* p, float v, long n, hipStream_t st) {
  hipLaunchKernelGGL(fill_kernel, dim3(ssv_cdiv(n, 256)), dim3(256), 0, st, p, v, n);
  return ssv_check_launch("fill");
}

// ---- out(b,c) = sum_t x(b,c,t): one wave per row ---------------------------------------------------
__global__ __launch_bounds__(256) void rowsum_kernel(const float* __restrict__ x, long x_bs, float* __restrict__ out, int B, int C, int L) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= B * C) return;
  const int b = row / C, c = row % C;
  const float* p = x + (long)b * x_bs + (long)c * L;
  float s = 0.f;
  int t = lane;
  for (; t + 192 < L; t += 256) {                    // four loads in flight (one at a time, each behind a wait, as a plain loop); same order of adds
    const float v0 = p[t], v1 = p[t + 64], v2 = p[t + 128], v3 = p[t + 192];
    s += v0; s += v1; s += v2; s += v3;
  }
  for (; t < L; t += 64) s += p[t];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) out[row] = s;
}
extern "C" int ssv_rowsum(const float* x, long x_bs, float* out, int B, int C, int L, ssv_stream_t stream) {
  SSV_CHECK(B > 0 && C > 0 && L > 0, SSV_BAD_SHAPE, "rowsum: bad shape B=%d C=%d L=%d", B, C, L);
  hipLaunchKernelGGL(rowsum_kernel, dim3(ssv_cdiv((long)B * C, 4)), dim3(256), 0, (hipStream_t)stream, x, x_bs, out, B, C, L);
  return ssv_check_launch("rowsum");
}

__device__ __forceinline__ float block_sum256(float v, float* red);        // (defined with the loss kernels below)
// ---- out(c) = sum_{b,t} x(b,c,t): a conv layer's bias gradient in ONE launch (round 6; before: ssv_rowsum + ssv_sum_slabs) -------------------
// One workgroup per channel; thread i adds the elements (b, t) with b * L + t = i, i + 256, ... in that order (eight loads in flight), then the 256
// partial sums are added in a fixed tree: bitwise reproducible, no atomics.
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ x, long x_bs, float* __restrict__ out, int B, int C, int L) {
  __shared__ float red[4];
  const int c = blockIdx.x;
  const float* __restrict__ p = x + (long)c * L;
  const long n = (long)B * L;
  float s = 0.f;
  long i = threadIdx.x;
  auto at = [&](long k) { const int b = (int)(k / L); return p[(long)b * x_bs + (k - (long)b * L)]; };
  for (; i + 1792 < n; i += 2048) {                          // eight loads in flight (a channel's B x L values are one workgroup's: 40-160 per thread)
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = at(i + 256 * u);
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; i < n; i += 256) s += at(i);
  s = block_sum256(s, red);
  if (threadIdx.x == 0) out[c] = s;
}
extern "C" int ssv_bias_grad(const float* x, long x_bs, float* out, int B, int C, int L, ssv_stream_t stream) {
  SSV_CHECK(x && out && B > 0 && C > 0 && L > 0, SSV_BAD_SHAPE, "bias_grad: bad shape B=%d C=%d L=%d", B, C, L);
  hipLaunchKernelGGL(bias_grad_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, x, x_bs, out, B, C, L);
  return ssv_check_launch("bias_grad");
}

extern "C" int ssv_sum_slabs(const float* slabs, float* out, long n, int Z, long stride, ssv_stream_t stream) {
  SSV_CHECK(slabs && out && n > 0 && Z > 0, SSV_BAD_SHAPE, "sum_slabs: bad argument n=%ld Z=%d", n, Z);
  return ssv_launch_reduce_slabs(slabs, out, n, Z, stride, (hipStream_t)stream);
}
__global__ __launch_bounds__(256) void copy_rows_kernel(const float* __restrict__ src, long src_bs, float* __restrict__ dst, long dst_bs, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[(long)blockIdx.y * dst_bs + i] = src[(long)blockIdx.y * src_bs + i];
}
extern "C" int ssv_copy_rows(const float* src, long src_bs, float* dst, long dst_bs, int B, long n, ssv_stream_t stream) {
  SSV_CHECK(src && dst && B > 0 && B <= 65535 && n > 0, SSV_BAD_SHAPE, "copy_rows: bad argument B=%d n=%ld", B, n);
  hipLaunchKernelGGL(copy_rows_kernel, dim3(ssv_cdiv(n, 256), B), dim3(256), 0, (hipStream_t)stream, src, src_bs, dst, dst_bs, n);
  return ssv_check_launch("copy_rows");
}

// ---- layout changes that also deliver the result's operand scale list (round 6: each replaced one torch copy kernel AND one ssv_absmax launch) ----
// Teacher forcing (train/ordinary.py:226, train/adversarial_wasserstein_gp.py:277: torch.cat((zeros, mel[:, :, :-1]), -1)):
// y(b, c, 0) = 0, y(b, c, t) = x(b, c, t - 1); amax[b * npb + i] = max |y| over the i-th of npb equal ROW ranges of item b (may be null; 0 for an empty range).
__global__ __launch_bounds__(256) void shift_right_amax_kernel(const float* __restrict__ x, long x_bs, float* __restrict__ y, int C, int T,
                                                               float* __restrict__ amax, int npb) {
  __shared__ float sm[4];
  const float* __restrict__ xb = x + (long)blockIdx.y * x_bs;
  float* __restrict__ yb = y + (long)blockIdx.y * C * T;
  const int rows = (C + npb - 1) / npb;                             // a piece = a range of rows (any partition gives a valid list)
  const int r0 = blockIdx.x * rows, r1 = min(r0 + rows, C);
  float m = 0.f;
  for (int r = r0; r < r1; ++r)
    for (int t = threadIdx.x; t < T; t += 256) {
      const float v = t ? xb[(long)r * T + t - 1] : 0.f;
      yb[(long)r * T + t] = v;
      m = fmaxf(m, fabsf(v));
    }
  m = ssv_wg_max<4>(m, sm);
  if (amax && threadIdx.x == 0) amax[(long)blockIdx.y * npb + blockIdx.x] = m;
}
extern "C" int ssv_shift_right_amax(const float* x, long x_bs, float* y, int B, int C, int T, float* amax, int namax, ssv_stream_t stream) {
  SSV_CHECK(x && y && B > 0 && B <= 65535 && C > 0 && T > 0 && namax > 0 && namax <= 65535, SSV_BAD_SHAPE, "shift_right_amax: bad argument");
  hipLaunchKernelGGL(shift_right_amax_kernel, dim3(namax, B), dim3(256), 0, (hipStream_t)stream, x, x_bs, y, C, T, amax, namax);
  return ssv_check_launch("shift_right_amax");
}
// De-interleave: x (B items of 2 n floats, item stride x_bs) -> out[j][b][i] = x[b][2 i + j], j = 0, 1 (the two taps of the transposed
// convolution's output gradient, models/TTSModel.py:309,314 backward); amax[b * npb + p] = max |x| over the p-th of npb equal PAIR ranges (may be null).
__global__ __launch_bounds__(256) void deinterleave2_amax_kernel(const float* __restrict__ x, long x_bs, float* __restrict__ out, long n, int B,
                                                                 float* __restrict__ amax, int npb) {
  __shared__ float sm[4];
  const float* __restrict__ xb = x + (long)blockIdx.y * x_bs;
  float* __restrict__ o0 = out + (long)blockIdx.y * n;
  float* __restrict__ o1 = out + ((long)B + blockIdx.y) * n;
  const long piece = (((n + npb - 1) / npb) + 1) & ~1L;            // pairs per workgroup, even: two pairs = one 16-byte load
  const long lo = (long)blockIdx.x * piece, hi = min(lo + piece, n);
  float m = 0.f;
  const bool vec = ((((size_t)xb) & 15) == 0) && ((((size_t)o0) & 7) == 0) && ((((size_t)o1) & 7) == 0);
  auto one = [&](long k) { const float a = xb[2 * k], c = xb[2 * k + 1]; o0[k] = a; o1[k] = c; m = fmaxf(m, fmaxf(fabsf(a), fabsf(c))); };
  if (vec) {
    long i = lo + 2L * threadIdx.x;                                 // (lo is even: 2 lo floats = a multiple of 16 bytes)
    for (; i + 1 < hi; i += 512) {
      const f32x4 q = *reinterpret_cast<const f32x4*>(xb + 2 * i);
      *reinterpret_cast<float2*>(o0 + i) = make_float2(q[0], q[2]);
      *reinterpret_cast<float2*>(o1 + i) = make_float2(q[1], q[3]);
      m = fmaxf(m, fmaxf(fmaxf(fabsf(q[0]), fabsf(q[1])), fmaxf(fabsf(q[2]), fabsf(q[3]))));
    }
    if (i < hi) one(i);                                             // the odd last pair of the last range: exactly one thread arrives at i == hi - 1
  } else {
    for (long i = lo + threadIdx.x; i < hi; i += 256) one(i);
  }
  m = ssv_wg_max<4>(m, sm);
  if (amax && threadIdx.x == 0) amax[(long)blockIdx.y * npb + blockIdx.x] = m;
}
// The same split ROW by row (round 6): out(b, 2 r + j, t) = x(b, r, 2 t + j) -- the output gradient of a ConvTranspose1d(k = 2, s = 2) as the (B, 2 Cout, L)
// output gradient of the 1x1 convolution it is (y(b, o, 2t + j) = u(b, 2o + j, t), u = W2 x, W2 = w.view(Cin, 2 Cout)): dx and dw are then ONE dense
// k = 1 product each over 2 Cout channels instead of two per tap, and dw lands in the weight's own (Cin, Cout, 2) layout.
__global__ __launch_bounds__(256) void deinterleave2_rows_amax_kernel(const float* __restrict__ x, long x_bs, float* __restrict__ out, int rows, int L,
                                                                      float* __restrict__ amax, int npb) {
  __shared__ float sm[4];
  const float* __restrict__ xb = x + (long)blockIdx.y * x_bs;
  float* __restrict__ ob = out + (long)blockIdx.y * 2 * rows * L;
  const long n = (long)rows * L;
  const long piece = (n + npb - 1) / npb;
  const long lo = (long)blockIdx.x * piece, hi = min(lo + piece, n);
  const bool vec = (((size_t)xb) & 7) == 0;
  float m = 0.f;
  for (long i = lo + threadIdx.x; i < hi; i += 256) {
    const int r = (int)(i / L), t = (int)(i - (long)r * L);
    float a, c;
    if (vec) { const float2 v = *reinterpret_cast<const float2*>(xb + 2 * i); a = v.x; c = v.y; }
    else { a = xb[2 * i]; c = xb[2 * i + 1]; }
    ob[(long)(2 * r) * L + t] = a;
    ob[(long)(2 * r + 1) * L + t] = c;
    m = fmaxf(m, fmaxf(fabsf(a), fabsf(c)));
  }
  m = ssv_wg_max<4>(m, sm);
  if (amax && threadIdx.x == 0) amax[(long)blockIdx.y * npb + blockIdx.x] = m;
}
extern "C" int ssv_deinterleave2_rows_amax(const float* x, long x_bs, float* out, int B, int rows, int L, float* amax, int namax, ssv_stream_t stream) {
  SSV_CHECK(x && out && B > 0 && B <= 65535 && rows > 0 && L > 0 && namax > 0 && namax <= 65535 && (long)rows * L < (1L << 31), SSV_BAD_SHAPE,
            "deinterleave2_rows_amax: bad argument");
  hipLaunchKernelGGL(deinterleave2_rows_amax_kernel, dim3(namax, B), dim3(256), 0, (hipStream_t)stream, x, x_bs, out, rows, L, amax, namax);
  return ssv_check_launch("deinterleave2_rows_amax");
}
extern "C" int ssv_deinterleave2_amax(const float* x, long x_bs, float* out, int B, long n, float* amax, int namax, ssv_stream_t stream) {
  SSV_CHECK(x && out && B > 0 && B <= 65535 && n > 0 && namax > 0 && namax <= 65535, SSV_BAD_SHAPE, "deinterleave2_amax: bad argument");
  hipLaunchKernelGGL(deinterleave2_amax_kernel, dim3(namax, B), dim3(256), 0, (hipStream_t)stream, x, x_bs, out, n, B, amax, namax);
  return ssv_check_launch("deinterleave2_amax");
}

// ---- text embedding ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, int B, int N, int E, int V) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * E * N) return;
  const int n = (int)(i % N);
  const int e = (int)((i / N) % E), b = (int)(i / ((long)N * E));
  const int64_t id = ids[(long)b * N + n];
  float v = bias ? bias[e] : 0.f;
  if (id >= 0 && id < V) v += w[(long)e * V + id];
  y[i] = v;
}
// One workgroup per embedding channel e.  Each of the 128 threads owns a private row of vocabulary
// bins in LDS and a strided share of the (b, n) positions; the rows are then summed in thread order,
// so the result does not depend on timing (no atomics).
#define EMB_MAXV 64
__global__ __launch_bounds__(128) void embed_bwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dy,
                                                        float* __restrict__ dw, float* __restrict__ dbias, int B, int N, int E, int V) {
  __shared__ float bins[128 * (EMB_MAXV + 1)];
  const int e = blockIdx.x, tid = threadIdx.x;
  float* mine = bins + tid * (EMB_MAXV + 1);
  for (int v = 0; v <= V; ++v) mine[v] = 0.f;     // bin V collects every position (bias gradient)
  const long tot = (long)B * N;
  long i = tid;
  for (; i + 7 * 128 < tot; i += 8 * 128) {          // eight positions' loads in flight (one at a time before: 47 memory round trips in a row, 28 us)
    float g[8];
    int64_t id[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long q = i + 128 * u;
      g[u] = dy[((q / N) * E + e) * N + q % N];
      id[u] = ids[q];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {                      // same order of additions as the plain loop
      if (id[u] >= 0 && id[u] < V) mine[id[u]] += g[u];
      mine[V] += g[u];
    }
  }
  for (; i < tot; i += 128) {
    const int b = (int)(i / N), n = (int)(i % N);
    const float g = dy[((long)b * E + e) * N + n];
    const int64_t id = ids[i];
    if (id >= 0 && id < V) mine[id] += g;
    mine[V] += g;
  }
  __syncthreads();
  if (tid <= V) {
    float s = 0.f;
    for (int k = 0; k < 128; ++k) s += bins[k * (EMB_MAXV + 1) + tid];
    if (tid < V) dw[(long)e * V + tid] = s; else if (dbias) dbias[e] = s;
  }
}
extern "C" int ssv_text_embed_fwd(const int64_t* ids, const float* w, const float* bias, float* y, int B, int N, int E, int vocab, ssv_stream_t stream) {
  SSV_CHECK(B > 0 && N > 0 && E > 0 && vocab > 0, SSV_BAD_SHAPE, "text_embed_fwd: bad shape");
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(ssv_cdiv((long)B * E * N, 256)), dim3(256), 0, (hipStream_t)stream, ids, w, bias, y, B, N, E, vocab);
  return ssv_check_launch("text_embed_fwd");
}
extern "C" int ssv_text_embed_bwd(const int64_t* ids, const float* dy, float* dw, float* dbias, int B, int N, int E, int vocab, ssv_stream_t stream) {
  SSV_CHECK(B > 0 && N > 0 && E > 0 && vocab > 0, SSV_BAD_SHAPE, "text_embed_bwd: bad shape");
  SSV_CHECK(vocab <= EMB_MAXV, SSV_UNSUPPORTED, "text_embed_bwd: vocabulary %d > %d", vocab, EMB_MAXV);
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(E), dim3(128), 0, (hipStream_t)stream, ids, dy, dw, dbias, B, N, E, vocab);
  return ssv_check_launch("text_embed_bwd");
}

// ---- spectrogram losses --------------------------------------------------------------------------------
#define LOSS_BLOCKS 1024
__device__ __forceinline__ float block_sum256(float v, float* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void spec_loss_part_kernel(const float* __restrict__ y, const float* __restrict__ gt, long n, float* __restrict__ part) {
  __shared__ float red[4];
  float s1 = 0.f, s2 = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float a = y[i], g = gt[i];
    s1 += fabsf(g - a);
    s2 += -g * logf(a + 1e-8f) - (1.f - g) * logf(1.f - a + 1e-8f);
  }
  s1 = block_sum256(s1, red);
  s2 = block_sum256(s2, red);
  if (threadIdx.x == 0) { part[blockIdx.x] = s1; part[gridDim.x + blockIdx.x] = s2; }
}
// out[k] = scale * sum_i part[k*nblk + i], k < nout; one workgroup, fixed order
__global__ __launch_bounds__(256) void finish_sums_kernel(const float* __restrict__ part, int nblk, int nout, float scale, float* __restrict__ out) {
  __shared__ float red[4];
  for (int k = 0; k < nout; ++k) {
    float s = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) s += part[(long)k * nblk + i];
    s = block_sum256(s, red);
    if (threadIdx.x == 0) out[k] = s * scale;
  }
}
__global__ __launch_bounds__(256) void spec_loss_bwd_kernel(const float* __restrict__ y, const float* __restrict__ gt, long n,
                                                            const float* __restrict__ gscale, float* __restrict__ dy) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float inv = 1.f / (float)n, w1 = gscale[0] * inv, w2 = gscale[1] * inv;
  const float a = y[i], g = gt[i];
  const float d = a - g;
  const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
  dy[i] = w1 * sgn + w2 * (-g / (a + 1e-8f) + (1.f - g) / (1.f - a + 1e-8f));
}
extern "C" size_t ssv_spec_losses_workspace(long n) { (void)n; return 2 * LOSS_BLOCKS * sizeof(float); }
extern "C" int ssv_spec_losses_fwd(const float* y, const float* gt, long n, float* out, void* ws, size_t ws_bytes, ssv_stream_t stream) {
  SSV_CHECK(n > 0, SSV_BAD_SHAPE, "spec_losses_fwd: n=%ld", n);
  SSV_CHECK(ws && ws_bytes >= ssv_spec_losses_workspace(n), SSV_BAD_SHAPE, "spec_losses_fwd: workspace too small");
  const int nblk = (int)((n + 255) / 256 < LOSS_BLOCKS ? (n + 255) / 256 : LOSS_BLOCKS);
  hipLaunchKernelGGL(spec_loss_part_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, y, gt, n, (float*)ws);
  SSV_TRY(ssv_check_launch("spec_loss_part"));
  hipLaunchKernelGGL(finish_sums_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)ws, nblk, 2, 1.f / (float)n, out);
  return ssv_check_launch("finish_sums");
}
// Forward and backward of the two spectrogram losses in ONE pass over (y, gt) (round 6): the partial sums of spec_loss_part_kernel and the dy of
// spec_loss_bwd_kernel from the same 16-byte loads.  A training step knows the gradient of the loss vector before the forward runs (a constant seed),
// and the two-kernel form read the 85 MB prediction and its target twice (67 + 46 us at (32, 513, 1300)); the forward kernel alone was bound by its two
// logf per element, not by memory.  Same formulas and the same per-element arithmetic as the two kernels; the sums are taken in another order.
__global__ __launch_bounds__(256) void spec_loss_fused_kernel(const float* __restrict__ y, const float* __restrict__ gt, long n, const float* __restrict__ gscale,
                                                              float* __restrict__ dy, float* __restrict__ part) {
  __shared__ float red[4];
  const float inv = 1.f / (float)n, w1 = gscale[0] * inv, w2 = gscale[1] * inv;
  float s1 = 0.f, s2 = 0.f;
  auto one = [&](float a, float g, float& d) __attribute__((always_inline)) {
    s1 += fabsf(g - a);
    s2 += -g * logf(a + 1e-8f) - (1.f - g) * logf(1.f - a + 1e-8f);
    const float df = a - g;
    const float sgn = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
    d = w1 * sgn + w2 * (-g / (a + 1e-8f) + (1.f - g) / (1.f - a + 1e-8f));
  };
  const long n4 = n >> 2;
  const float4* __restrict__ y4 = reinterpret_cast<const float4*>(y);
  const float4* __restrict__ g4 = reinterpret_cast<const float4*>(gt);
  float4* __restrict__ d4 = reinterpret_cast<float4*>(dy);
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + stride < n4; i += 2 * stride) {                 // two vectors of each operand in flight
    const float4 a0 = y4[i], b0 = g4[i], a1 = y4[i + stride], b1 = g4[i + stride];
    float4 d0, d1;
    one(a0.x, b0.x, d0.x); one(a0.y, b0.y, d0.y); one(a0.z, b0.z, d0.z); one(a0.w, b0.w, d0.w);
    one(a1.x, b1.x, d1.x); one(a1.y, b1.y, d1.y); one(a1.z, b1.z, d1.z); one(a1.w, b1.w, d1.w);
    d4[i] = d0; d4[i + stride] = d1;
  }
  if (i < n4) {
    const float4 a0 = y4[i], b0 = g4[i];
    float4 d0;
    one(a0.x, b0.x, d0.x); one(a0.y, b0.y, d0.y); one(a0.z, b0.z, d0.z); one(a0.w, b0.w, d0.w);
    d4[i] = d0;
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) {        // the last n % 4 elements
    const long e = (n4 << 2) + threadIdx.x;
    float d;
    one(y[e], gt[e], d);
    dy[e] = d;
  }
  s1 = block_sum256(s1, red);
  s2 = block_sum256(s2, red);
  if (threadIdx.x == 0) { part[blockIdx.x] = s1; part[gridDim.x + blockIdx.x] = s2; }
}
extern "C" int ssv_spec_losses_fwd_bwd(const float* y, const float* gt, long n, const float* gscale, float* out, float* dy, void* ws, size_t ws_bytes,
                                       ssv_stream_t stream) {
  SSV_CHECK(n > 0 && y && gt && gscale && out && dy, SSV_BAD_SHAPE, "spec_losses_fwd_bwd: bad argument (n=%ld)", n);
  SSV_CHECK(ws && ws_bytes >= ssv_spec_losses_workspace(n), SSV_BAD_SHAPE, "spec_losses_fwd_bwd: workspace too small");
  if ((((size_t)y | (size_t)gt | (size_t)dy) & 15) != 0) {      // 16-byte vectors need aligned tensors; otherwise the two passes
    SSV_TRY(ssv_spec_losses_fwd(y, gt, n, out, ws, ws_bytes, stream));
    return ssv_spec_losses_bwd(y, gt, n, gscale, dy, stream);
  }
  const long v = (n + 3) / 4;
  const int nblk = (int)((v + 511) / 512 < LOSS_BLOCKS ? (v + 511) / 512 : LOSS_BLOCKS);
  hipLaunchKernelGGL(spec_loss_fused_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, y, gt, n, gscale, dy, (float*)ws);
  SSV_TRY(ssv_check_launch("spec_loss_fused"));
  hipLaunchKernelGGL(finish_sums_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)ws, nblk, 2, 1.f / (float)n, out);
  return ssv_check_launch("finish_sums");
}
extern "C" int ssv_spec_losses_bwd(const float* y, const float* gt, long n, const float* gscale, float* dy, ssv_stream_t stream) {
  SSV_CHECK(n > 0, SSV_BAD_SHAPE, "spec_losses_bwd: n=%ld", n);
  hipLaunchKernelGGL(spec_loss_bwd_kernel, dim3(ssv_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, y, gt, n, gscale, dy);
  return ssv_check_launch("spec_loss_bwd");
}

// ---- guided attention loss ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gatt_part_kernel(const float* __restrict__ a, const float* __restrict__ gaw, int gaw_T,
                                                        long tot, int N, int T, float* __restrict__ part) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < tot; i += (long)gridDim.x * 256) {
    const int t = (int)(i % T), n = (int)((i / T) % N);
    s += a[i] * gaw[(long)n * gaw_T + t];
  }
  s = block_sum256(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void gatt_bwd_kernel(const float* __restrict__ gaw, int gaw_T, const float* __restrict__ gscale,
                                                       float* __restrict__ da, long tot, int N, int T) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= tot) return;
  const int t = (int)(i % T), n = (int)((i / T) % N);
  da[i] = gscale[0] * gaw[(long)n * gaw_T + t] / (float)tot;
}
extern "C" size_t ssv_guided_att_loss_workspace(int B, int N, int T) { (void)B; (void)N; (void)T; return LOSS_BLOCKS * sizeof(float); }
extern "C" int ssv_guided_att_loss_fwd(const float* a, const float* gaw, int gaw_T, float* out, int B, int N, int T,
                                       void* ws, size_t ws_bytes, ssv_stream_t stream) {
  SSV_CHECK(B > 0 && N > 0 && T > 0 && gaw_T >= T, SSV_BAD_SHAPE, "guided_att_loss_fwd: bad shape B=%d N=%d T=%d gaw_T=%d", B, N, T, gaw_T);
  SSV_CHECK(ws && ws_bytes >= LOSS_BLOCKS * sizeof(float), SSV_BAD_SHAPE, "guided_att_loss_fwd: workspace too small");
  const long tot = (long)B * N * T;
  const int nblk = (int)((tot + 255) / 256 < LOSS_BLOCKS ? (tot + 255) / 256 : LOSS_BLOCKS);
  hipLaunchKernelGGL(gatt_part_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, a, gaw, gaw_T, tot, N, T, (float*)ws);
  SSV_TRY(ssv_check_launch("gatt_part"));
  hipLaunchKernelGGL(finish_sums_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)ws, nblk, 1, 1.f / (float)tot, out);
  return ssv_check_launch("finish_sums");
}
extern "C" int ssv_guided_att_loss_bwd(const float* gaw, int gaw_T, const float* gscale, float* da, int B, int N, int T, ssv_stream_t stream) {
  SSV_CHECK(B > 0 && N > 0 && T > 0 && gaw_T >= T, SSV_BAD_SHAPE, "guided_att_loss_bwd: bad shape");
  const long tot = (long)B * N * T;
  hipLaunchKernelGGL(gatt_bwd_kernel, dim3(ssv_cdiv(tot, 256)), dim3(256), 0, (hipStream_t)stream, gaw, gaw_T, gscale, da, tot, N, T);
  return ssv_check_launch("gatt_bwd");
}

// ---- multi-tensor Adam -----------------------------------------------------------------------------------
// Same arithmetic as torch.optim.Adam's single-tensor path: m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// denom = sqrt(v)/sqrt(1-b2^t) + eps; p -= (lr/(1-b1^t)) * m / denom.
__global__ __launch_bounds__(256) void adam_multi_kernel(const ssv_adam_chunk* __restrict__ chunks, float lr, float b1, float b2, float eps,
                                                         int step_host, const int* __restrict__ step_dev) {
  // step count: host value, or (device counter + 1) so that a captured hipGraph advances on replay
  const int step = step_dev ? step_dev[0] + 1 : step_host;
  const double bc1 = 1.0 - pow((double)b1, (double)step), bc2 = 1.0 - pow((double)b2, (double)step);
  const float step_size = (float)((double)lr / bc1), inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  const ssv_adam_chunk ch = chunks[blockIdx.x];
  auto upd = [&](float& p, float g, float& m, float& v) {
    m = b1 * m + (1.f - b1) * g;
    v = b2 * v + (1.f - b2) * g * g;
    p -= step_size * (m / (sqrtf(v) * inv_sqrt_bc2 + eps));
  };
  long i0 = 0;
  // 16-byte accesses when the four pointers allow it (28 B/param of pure streaming: this kernel is HBM-bound)
  if ((((uintptr_t)ch.p | (uintptr_t)ch.g | (uintptr_t)ch.m | (uintptr_t)ch.v) & 15) == 0) {
    const long n4 = ch.n >> 2;
    // The pointers come from the chunk table, i.e. from memory: spelled out as global, or hipcc emits flat_load / flat_store
    // (see ssv_global in ssv_common.h).  Two elements per thread in flight: eight 16-byte loads before the first use.
    typedef float vf4 __attribute__((ext_vector_type(4)));
    auto upd4 = [&](vf4& p, const vf4 g, vf4& m, vf4& v) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float pp = p[k], mm = m[k], vv = v[k];
        upd(pp, g[k], mm, vv);
        p[k] = pp; m[k] = mm; v[k] = vv;
      }
    };
    typedef __attribute__((address_space(1))) vf4 gf4;
    gf4* __restrict__ p4 = (gf4*)ch.p; const gf4* __restrict__ g4 = (const gf4*)ch.g;
    gf4* __restrict__ m4 = (gf4*)ch.m; gf4* __restrict__ v4 = (gf4*)ch.v;
    long i = threadIdx.x;
    for (; i + 256 < n4; i += 512) {
      vf4 p = p4[i], m = m4[i], v = v4[i], q = p4[i + 256], n = m4[i + 256], w = v4[i + 256];
      const vf4 g = g4[i], h = g4[i + 256];
      upd4(p, g, m, v);
      upd4(q, h, n, w);
      p4[i] = p; m4[i] = m; v4[i] = v;
      p4[i + 256] = q; m4[i + 256] = n; v4[i + 256] = w;
    }
    for (; i < n4; i += 256) {
      vf4 p = p4[i], m = m4[i], v = v4[i];
      const vf4 g = g4[i];
      upd4(p, g, m, v);
      p4[i] = p; m4[i] = m; v4[i] = v;
    }
    i0 = n4 << 2;
  }
  for (long i = i0 + threadIdx.x; i < ch.n; i += 256) {
    float p = ch.p[i], m = ch.m[i], v = ch.v[i];
    upd(p, ch.g[i], m, v);
    ch.p[i] = p; ch.m[i] = m; ch.v[i] = v;
  }
}
__global__ void step_inc_kernel(int* step_dev) { step_dev[0] += 1; }
extern "C" int ssv_adam_multi(const ssv_adam_chunk* chunks, int nchunks, float lr, float beta1, float beta2, float eps,
                              int step, int* step_dev, ssv_stream_t stream) {
  SSV_CHECK(nchunks > 0 && (step_dev || step > 0), SSV_BAD_SHAPE, "adam_multi: nchunks=%d step=%d", nchunks, step);
  hipLaunchKernelGGL(adam_multi_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, chunks, lr, beta1, beta2, eps, step, (const int*)step_dev);
  SSV_TRY(ssv_check_launch("adam_multi"));
  if (step_dev) {
    hipLaunchKernelGGL(step_inc_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
    return ssv_check_launch("step_inc");
  }
  return 0;
}
