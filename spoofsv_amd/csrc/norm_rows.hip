// Channel-axis LayerNorm for long-sequence (B, C, T) tensors, gfx950: "row-chunk" decomposition.
//
// The column-tile kernels of norm.hip read 64-byte pieces of rows that lie T*4 bytes apart and reach ~2.6 TB/s (forward) /
// 1.8 TB/s (backward) where a contiguous stream reaches 5.5-7 TB/s on the same device.  Here every workgroup streams whole
// ROWS: it owns LNR_ROWS consecutive channels of one batch item and all (up to 256*KC) columns, thread t owning columns
// t, t+256, ...  Sums over channels are accumulated per column in registers across the rows of the chunk and written as
// per-chunk partial rows; a tiny kernel adds the chunks in a fixed order.  Sums over columns (parameter gradients) are block
// reductions per row.  The price is a second pass over the tensors (sums, then apply; the re-read is mostly served by the
// 256 MB Infinity Cache).  Measured (tools/bench_ln.py, B=32): a win only for the BACKWARD pass of long rows (C=513,
// L=1300: 124 us vs 144 us); forward passes and short rows stay on the column-tile kernels, which touch every byte once.
// Both paths share the stats layout, so forward and backward choose independently (SSV_LNR=all|off overrides).
//
//   forward : colsum (sum, sum of squares of x - x[row 0], i.e. shifted by a sample to avoid cancellation)
//             -> finalize (mean, rstd) -> apply (normalise [+ highway gate | activation])
//   backward: sums (column sums of a = dn*gamma and a*xhat; row sums -> dgamma, dbeta) -> finalize -> apply (dX, row sums -> dbias)
#include <stdlib.h>
#include <string.h>
#include "ssv_common.h"

#define LNR_ROWS 8
#define LN_EPS 1e-5f
__device__ __forceinline__ float sigm_(float v) { return 1.f / (1.f + __expf(-v)); }

// block-wide sums of NQ per-thread values for one row; rowred is [LNR_ROWS][NQ][4 waves]
template <int NQ>
__device__ __forceinline__ void row_reduce(const float (&v)[NQ], float* rowred, int r) {
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const float s = ssv_wave_sum(v[q]);
    if ((threadIdx.x & 63) == 0) rowred[(r * NQ + q) * 4 + (threadIdx.x >> 6)] = s;
  }
}

// ---- forward: column partial sums ---------------------------------------------------------------------------------
// X: (B, nln*C, L) rows; LayerNorm group g covers rows [g*C, (g+1)*C).  part[((b*nln+g)*cpl + ck)*2 + {0,1}][L]
template <int KC>
__global__ __launch_bounds__(256) void lnr_colsum_kernel(const float* __restrict__ X, long x_bs, float* __restrict__ part,
                                                          int C, int L, int nln, int cpl) {
  const int g = blockIdx.x / cpl, ck = blockIdx.x % cpl, b = blockIdx.z;
  const int c0 = ck * LNR_ROWS;
  const float* Xg = X + (long)b * x_bs + (long)g * C * L;
  float s[KC], q[KC], K[KC];
  int t[KC];
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    t[k] = (blockIdx.y * KC + k) * 256 + threadIdx.x;
    K[k] = t[k] < L ? Xg[t[k]] : 0.f;
    s[k] = 0.f; q[k] = 0.f;
  }
#pragma unroll
  for (int r = 0; r < LNR_ROWS; ++r) {
    const int c = c0 + r;
    if (c < C) {
#pragma unroll
      for (int k = 0; k < KC; ++k)
        if (t[k] < L) { const float d = Xg[(long)c * L + t[k]] - K[k]; s[k] += d; q[k] += d * d; }
    }
  }
  float* pb = part + (((long)b * nln + g) * cpl + ck) * 2 * L;
#pragma unroll
  for (int k = 0; k < KC; ++k)
    if (t[k] < L) { pb[t[k]] = s[k]; pb[L + t[k]] = q[k]; }
}
// stats (B, 2*nln, L): mean, rstd per group
__global__ __launch_bounds__(256) void lnr_stats_kernel(const float* __restrict__ X, long x_bs, const float* __restrict__ part,
                                                        float* __restrict__ stats, int C, int L, int nln, int cpl) {
  const int t = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y, b = blockIdx.z;
  if (t >= L) return;
  const float* pb = part + (((long)b * nln + g) * cpl) * 2 * L;
  float S = 0.f, Q = 0.f;
  for (int ck = 0; ck < cpl; ++ck) { S += pb[(long)ck * 2 * L + t]; Q += pb[(long)ck * 2 * L + L + t]; }
  const float K = X[(long)b * x_bs + (long)g * C * L + t];
  const float inv = 1.f / (float)C, ms = S * inv;
  stats[((long)b * 2 * nln + 2 * g) * L + t] = K + ms;
  stats[((long)b * 2 * nln + 2 * g + 1) * L + t] = rsqrtf(fmaxf(Q * inv - ms * ms, 0.f) + LN_EPS);
}

// ---- forward: apply --------------------------------------------------------------------------------------------------
template <int KC>
__global__ __launch_bounds__(256) void lnr_gate_apply_kernel(const float* __restrict__ H, long h_bs, const float* __restrict__ X, long x_bs,
                                                              const float* __restrict__ stats, const float* __restrict__ g1, const float* __restrict__ b1,
                                                              const float* __restrict__ g2, const float* __restrict__ b2,
                                                              float* __restrict__ Y, long y_bs, int C, int L) {
  const int c0 = blockIdx.x * LNR_ROWS, b = blockIdx.z;
  const float* Hb = H + (long)b * h_bs;
  const float* Xb = X + (long)b * x_bs;
  float* Yb = Y + (long)b * y_bs;
  const float* sb = stats + (long)b * 4 * L;
  int t[KC];
  float mu1[KC], r1[KC], mu2[KC], r2[KC];
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    t[k] = (blockIdx.y * KC + k) * 256 + threadIdx.x;
    const int tc = min(t[k], L - 1);
    mu1[k] = sb[tc]; r1[k] = sb[L + tc]; mu2[k] = sb[2L * L + tc]; r2[k] = sb[3L * L + tc];
  }
#pragma unroll
  for (int r = 0; r < LNR_ROWS; ++r) {
    const int c = c0 + r;
    if (c < C) {
      const float ga1 = g1[c], be1 = b1[c], ga2 = g2[c], be2 = b2[c];
#pragma unroll
      for (int k = 0; k < KC; ++k)
        if (t[k] < L) {
          const long o = (long)c * L + t[k];
          const float n1 = (Hb[o] - mu1[k]) * r1[k] * ga1 + be1;
          const float n2 = (Hb[(long)C * L + o] - mu2[k]) * r2[k] * ga2 + be2;
          const float s = sigm_(n1);
          Yb[o] = s * n2 + (1.f - s) * Xb[o];
        }
    }
  }
}
template <int KC>
__global__ __launch_bounds__(256) void lnr_act_apply_kernel(const float* __restrict__ X, long x_bs, const float* __restrict__ stats,
                                                             const float* __restrict__ gam, const float* __restrict__ bet,
                                                             float* __restrict__ Y, long y_bs, int C, int L, int act) {
  const int c0 = blockIdx.x * LNR_ROWS, b = blockIdx.z;
  const float* Xb = X + (long)b * x_bs;
  float* Yb = Y + (long)b * y_bs;
  const float* sb = stats + (long)b * 2 * L;
  int t[KC];
  float mu[KC], rs[KC];
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    t[k] = (blockIdx.y * KC + k) * 256 + threadIdx.x;
    const int tc = min(t[k], L - 1);
    mu[k] = sb[tc]; rs[k] = sb[L + tc];
  }
#pragma unroll
  for (int r = 0; r < LNR_ROWS; ++r) {
    const int c = c0 + r;
    if (c < C) {
      const float ga = gam[c], be = bet[c];
#pragma unroll
      for (int k = 0; k < KC; ++k)
        if (t[k] < L) {
          const long o = (long)c * L + t[k];
          float n = (Xb[o] - mu[k]) * rs[k] * ga + be;
          if (act == 1) n = fmaxf(n, 0.f); else if (act == 2) n = sigm_(n);
          Yb[o] = n;
        }
    }
  }
}

// ---- backward, highway gate ---------------------------------------------------------------------------------------------
// part[((b*cpl + ck)*4 + q)][L] (q: sum a1, sum a1*xh1, sum a2, sum a2*xh2); rowp[((b*ncb + cb)*4 + q)][C] (dgamma1, dbeta1, dgamma2, dbeta2)
template <int KC>
__global__ __launch_bounds__(256) void lnr_gate_bwd_sums_kernel(const float* __restrict__ dY, long dy_bs, const float* __restrict__ H,
                                                                 const float* __restrict__ X, long x_bs, const float* __restrict__ stats,
                                                                 const float* __restrict__ g1, const float* __restrict__ b1,
                                                                 const float* __restrict__ g2, const float* __restrict__ b2,
                                                                 float* __restrict__ dXres, long dx_bs, float* __restrict__ part,
                                                                 float* __restrict__ rowp, int C, int L, int cpl) {
  __shared__ float rowred[LNR_ROWS * 4 * 4];
  const int ck = blockIdx.x, c0 = ck * LNR_ROWS, b = blockIdx.z;
  const float* Hb = H + (long)b * 2 * C * L;
  const float* sb = stats + (long)b * 4 * L;
  int t[KC];
  float mu1[KC], r1[KC], mu2[KC], r2[KC], sa1[KC], sh1[KC], sa2[KC], sh2[KC];
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    t[k] = (blockIdx.y * KC + k) * 256 + threadIdx.x;
    const int tc = min(t[k], L - 1);
    mu1[k] = sb[tc]; r1[k] = sb[L + tc]; mu2[k] = sb[2L * L + tc]; r2[k] = sb[3L * L + tc];
    sa1[k] = sh1[k] = sa2[k] = sh2[k] = 0.f;
  }
#pragma unroll
  for (int r = 0; r < LNR_ROWS; ++r) {
    const int c = c0 + r;
    float pr[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
      const float ga1 = g1[c], be1 = b1[c], ga2 = g2[c], be2 = b2[c];
#pragma unroll
      for (int k = 0; k < KC; ++k)
        if (t[k] < L) {
          const long o = (long)c * L + t[k];
          const float dy = dY[(long)b * dy_bs + o], x = X[(long)b * x_bs + o];
          const float xh1 = (Hb[o] - mu1[k]) * r1[k], xh2 = (Hb[(long)C * L + o] - mu2[k]) * r2[k];
          const float n1 = xh1 * ga1 + be1, n2 = xh2 * ga2 + be2;
          const float s = sigm_(n1);
          const float dn2 = dy * s, dn1 = dy * (n2 - x) * s * (1.f - s);
          dXres[(long)b * dx_bs + o] = dy * (1.f - s);
          pr[0] += dn1 * xh1; pr[1] += dn1; pr[2] += dn2 * xh2; pr[3] += dn2;
          const float a1 = dn1 * ga1, a2 = dn2 * ga2;
          sa1[k] += a1; sh1[k] += a1 * xh1; sa2[k] += a2; sh2[k] += a2 * xh2;
        }
    }
    row_reduce<4>(pr, rowred, r);
  }
  float* pb = part + ((long)b * cpl + ck) * 4 * L;
#pragma unroll
  for (int k = 0; k < KC; ++k)
    if (t[k] < L) { pb[t[k]] = sa1[k]; pb[L + t[k]] = sh1[k]; pb[2L * L + t[k]] = sa2[k]; pb[3L * L + t[k]] = sh2[k]; }
  __syncthreads();
  if (threadIdx.x < LNR_ROWS * 4) {
    const int r = threadIdx.x / 4, q = threadIdx.x % 4, c = c0 + r;
    const float* rr = rowred + (r * 4 + q) * 4;
    if (c < C) rowp[(((long)b * gridDim.y + blockIdx.y) * 4 + q) * C + c] = (rr[0] + rr[1]) + (rr[2] + rr[3]);
  }
}
// m[b][nq][L] = sum over chunks / C
__global__ __launch_bounds__(256) void lnr_bwd_means_kernel(const float* __restrict__ part, float* __restrict__ m, int C, int L, int cpl, int nq) {
  const int t = blockIdx.x * 256 + threadIdx.x, q = blockIdx.y, b = blockIdx.z;
  if (t >= L) return;
  float s = 0.f;
  for (int ck = 0; ck < cpl; ++ck) s += part[(((long)b * cpl + ck) * nq + q) * L + t];
  m[((long)b * nq + q) * L + t] = s / (float)C;
}
// rowp2[((b*ncb + cb)*2 + q)][C]: sums over columns of dH (bias gradient of the conv)
template <int KC>
__global__ __launch_bounds__(256) void lnr_gate_bwd_apply_kernel(const float* __restrict__ dY, long dy_bs, const float* __restrict__ H,
                                                                  const float* __restrict__ X, long x_bs, const float* __restrict__ stats,
                                                                  const float* __restrict__ m, const float* __restrict__ g1, const float* __restrict__ b1,
                                                                  const float* __restrict__ g2, const float* __restrict__ b2,
                                                                  float* __restrict__ dH, float* __restrict__ rowp2, int C, int L) {
  __shared__ float rowred[LNR_ROWS * 2 * 4];
  const int c0 = blockIdx.x * LNR_ROWS, b = blockIdx.z;
  const float* Hb = H + (long)b * 2 * C * L;
  float* dHb = dH + (long)b * 2 * C * L;
  const float* sb = stats + (long)b * 4 * L;
  const float* mb = m + (long)b * 4 * L;
  int t[KC];
  float mu1[KC], r1[KC], mu2[KC], r2[KC], m1[KC], mh1[KC], m2[KC], mh2[KC];
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    t[k] = (blockIdx.y * KC + k) * 256 + threadIdx.x;
    const int tc = min(t[k], L - 1);
    mu1[k] = sb[tc]; r1[k] = sb[L + tc]; mu2[k] = sb[2L * L + tc]; r2[k] = sb[3L * L + tc];
    m1[k] = mb[tc]; mh1[k] = mb[L + tc]; m2[k] = mb[2L * L + tc]; mh2[k] = mb[3L * L + tc];
  }
#pragma unroll
  for (int r = 0; r < LNR_ROWS; ++r) {
    const int c = c0 + r;
    float pr[2] = {0.f, 0.f};
    if (c < C) {
      const float ga1 = g1[c], be1 = b1[c], ga2 = g2[c], be2 = b2[c];
#pragma unroll
      for (int k = 0; k < KC; ++k)
        if (t[k] < L) {
          const long o = (long)c * L + t[k];
          const float dy = dY[(long)b * dy_bs + o], x = X[(long)b * x_bs + o];
          const float xh1 = (Hb[o] - mu1[k]) * r1[k], xh2 = (Hb[(long)C * L + o] - mu2[k]) * r2[k];
          const float n1 = xh1 * ga1 + be1, n2 = xh2 * ga2 + be2;
          const float s = sigm_(n1);
          const float a2 = dy * s * ga2, a1 = dy * (n2 - x) * s * (1.f - s) * ga1;
          const float d1 = r1[k] * (a1 - m1[k] - xh1 * mh1[k]), d2 = r2[k] * (a2 - m2[k] - xh2 * mh2[k]);
          dHb[o] = d1; dHb[(long)C * L + o] = d2;
          pr[0] += d1; pr[1] += d2;
        }
    }
    row_reduce<2>(pr, rowred, r);
  }
  __syncthreads();
  if (threadIdx.x < LNR_ROWS * 2) {
    const int r = threadIdx.x / 2, q = threadIdx.x % 2, c = c0 + r;
    const float* rr = rowred + (r * 2 + q) * 4;
    if (c < C) rowp2[(((long)b * gridDim.y + blockIdx.y) * 2 + q) * C + c] = (rr[0] + rr[1]) + (rr[2] + rr[3]);
  }
}

// ---- backward, LayerNorm + activation ---------------------------------------------------------------------------------
// part[((b*cpl+ck)*2 + q)][L] (sum a, sum a*xh); rowp[((b*ncb+cb)*2 + q)][C] (dgamma, dbeta)
template <int KC>
__global__ __launch_bounds__(256) void lnr_act_bwd_sums_kernel(const float* __restrict__ dY, long dy_bs, const float* __restrict__ X, long x_bs,
                                                                const float* __restrict__ stats, const float* __restrict__ gam, const float* __restrict__ bet,
                                                                float* __restrict__ part, float* __restrict__ rowp, int C, int L, int cpl, int act) {
  __shared__ float rowred[LNR_ROWS * 2 * 4];
  const int ck = blockIdx.x, c0 = ck * LNR_ROWS, b = blockIdx.z;
  const float* sb = stats + (long)b * 2 * L;
  int t[KC];
  float mu[KC], rs[KC], sa[KC], sh[KC];
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    t[k] = (blockIdx.y * KC + k) * 256 + threadIdx.x;
    const int tc = min(t[k], L - 1);
    mu[k] = sb[tc]; rs[k] = sb[L + tc]; sa[k] = 0.f; sh[k] = 0.f;
  }
#pragma unroll
  for (int r = 0; r < LNR_ROWS; ++r) {
    const int c = c0 + r;
    float pr[2] = {0.f, 0.f};
    if (c < C) {
      const float ga = gam[c], be = bet[c];
#pragma unroll
      for (int k = 0; k < KC; ++k)
        if (t[k] < L) {
          const long o = (long)c * L + t[k];
          const float dy = dY[(long)b * dy_bs + o];
          const float xh = (X[(long)b * x_bs + o] - mu[k]) * rs[k];
          const float n = xh * ga + be;
          float dn;
          if (act == 1) dn = n > 0.f ? dy : 0.f;
          else if (act == 2) { const float s = sigm_(n); dn = dy * s * (1.f - s); }
          else dn = dy;
          pr[0] += dn * xh; pr[1] += dn;
          const float a = dn * ga;
          sa[k] += a; sh[k] += a * xh;
        }
    }
    row_reduce<2>(pr, rowred, r);
  }
  float* pb = part + ((long)b * cpl + ck) * 2 * L;
#pragma unroll
  for (int k = 0; k < KC; ++k)
    if (t[k] < L) { pb[t[k]] = sa[k]; pb[L + t[k]] = sh[k]; }
  __syncthreads();
  if (threadIdx.x < LNR_ROWS * 2) {
    const int r = threadIdx.x / 2, q = threadIdx.x % 2, c = c0 + r;
    const float* rr = rowred + (r * 2 + q) * 4;
    if (c < C) rowp[(((long)b * gridDim.y + blockIdx.y) * 2 + q) * C + c] = (rr[0] + rr[1]) + (rr[2] + rr[3]);
  }
}
template <int KC>
__global__ __launch_bounds__(256) void lnr_act_bwd_apply_kernel(const float* __restrict__ dY, long dy_bs, const float* __restrict__ X, long x_bs,
                                                                 const float* __restrict__ stats, const float* __restrict__ m,
                                                                 const float* __restrict__ gam, const float* __restrict__ bet,
                                                                 float* __restrict__ dX, long dx_bs, float* __restrict__ rowp2, int C, int L, int act) {
  __shared__ float rowred[LNR_ROWS * 1 * 4];
  const int c0 = blockIdx.x * LNR_ROWS, b = blockIdx.z;
  const float* sb = stats + (long)b * 2 * L;
  const float* mb = m + (long)b * 2 * L;
  int t[KC];
  float mu[KC], rs[KC], m0[KC], mh[KC];
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    t[k] = (blockIdx.y * KC + k) * 256 + threadIdx.x;
    const int tc = min(t[k], L - 1);
    mu[k] = sb[tc]; rs[k] = sb[L + tc]; m0[k] = mb[tc]; mh[k] = mb[L + tc];
  }
#pragma unroll
  for (int r = 0; r < LNR_ROWS; ++r) {
    const int c = c0 + r;
    float pr[1] = {0.f};
    if (c < C) {
      const float ga = gam[c], be = bet[c];
#pragma unroll
      for (int k = 0; k < KC; ++k)
        if (t[k] < L) {
          const long o = (long)c * L + t[k];
          const float dy = dY[(long)b * dy_bs + o];
          const float xh = (X[(long)b * x_bs + o] - mu[k]) * rs[k];
          const float n = xh * ga + be;
          float dn;
          if (act == 1) dn = n > 0.f ? dy : 0.f;
          else if (act == 2) { const float s = sigm_(n); dn = dy * s * (1.f - s); }
          else dn = dy;
          const float d = rs[k] * (dn * ga - m0[k] - xh * mh[k]);
          dX[(long)b * dx_bs + o] = d;
          pr[0] += d;
        }
    }
    row_reduce<1>(pr, rowred, r);
  }
  __syncthreads();
  if (threadIdx.x < LNR_ROWS) {
    const int r = threadIdx.x, c = c0 + r;
    const float* rr = rowred + r * 4;
    if (c < C) rowp2[((long)b * gridDim.y + blockIdx.y) * C + c] = (rr[0] + rr[1]) + (rr[2] + rr[3]);
  }
}

// out[q][c] = sum over n partial rows of src[(i*nq + q)*C + c]
__global__ __launch_bounds__(256) void lnr_rowpart_reduce_kernel(const float* __restrict__ src, float* __restrict__ out, int C, int nq, int n) {
  const int c = blockIdx.x * 256 + threadIdx.x, q = blockIdx.y;
  if (c >= C) return;
  float s = 0.f;
  for (int i = 0; i < n; ++i) s += src[((long)i * nq + q) * C + c];
  out[(long)q * C + c] = s;
}

// ---- host side --------------------------------------------------------------------------------------------------------------
static inline int lnr_kc(int L) { const int k = ssv_cdiv(L, 256); return k <= 1 ? 1 : k <= 2 ? 2 : k <= 3 ? 3 : k <= 4 ? 4 : 6; }
static inline int lnr_ncb(int L) { return ssv_cdiv(L, 256 * lnr_kc(L)); }
static inline int lnr_cpl(int C) { return ssv_cdiv(C, LNR_ROWS); }
static inline size_t a256(size_t n) { return (n + 255) & ~(size_t)255; }

// Measured (tools/bench_ln.py, B=32): this path only pays in the BACKWARD pass of long sequences (C=513, L=1300: 124 us
// vs 144 us for the column-tile kernel); forward and short rows are faster on the column-tile kernels, which touch every
// byte once.  SSV_LNR=all forces it everywhere (tests), SSV_LNR=off disables it.
static int lnr_mode() {
  static int m = -1;
  if (m < 0) { const char* e = getenv("SSV_LNR"); m = !e ? 0 : (!strcmp(e, "all") ? 1 : (!strcmp(e, "off") ? 2 : 0)); }
  return m;
}
bool ssv_lnr_use_fwd(int B, int C, int L) { return lnr_mode() == 1 && L >= 32; }
bool ssv_lnr_use(int B, int C, int L) {
  if (lnr_mode() == 2) return false;
  if (lnr_mode() == 1) return L >= 32;
  return L >= 1024 && (long)B * lnr_cpl(C) * lnr_ncb(L) >= 256;
}

size_t ssv_lnr_fwd_ws(int B, int C, int L, int nln) { return a256((size_t)B * nln * lnr_cpl(C) * 2 * L * sizeof(float)) + a256((size_t)B * 2 * nln * L * sizeof(float)); }
size_t ssv_lnr_bwd_ws(int B, int C, int L, int nq /*4 gate, 2 act*/) {
  return a256((size_t)B * lnr_cpl(C) * nq * L * sizeof(float)) + a256((size_t)B * nq * L * sizeof(float)) +
         a256((size_t)B * lnr_ncb(L) * nq * C * sizeof(float)) + a256((size_t)B * lnr_ncb(L) * (nq / 2) * C * sizeof(float));
}

#define LNR_KC(L, K)                                                                                  \
  do {                                                                                                \
    switch (lnr_kc(L)) { case 1: K(1); break; case 2: K(2); break; case 3: K(3); break; case 4: K(4); break; default: K(6); } \
  } while (0)

static int lnr_stats(const float* X, long x_bs, float* stats_out, float* ws, int B, int C, int L, int nln, hipStream_t st) {
  const int cpl = lnr_cpl(C), ncb = lnr_ncb(L);
  float* part = ws;
  dim3 grid(cpl * nln, ncb, B);
#define K(N) hipLaunchKernelGGL(lnr_colsum_kernel<N>, grid, dim3(256), 0, st, X, x_bs, part, C, L, nln, cpl)
  LNR_KC(L, K);
#undef K
  SSV_TRY(ssv_check_launch("lnr_colsum"));
  hipLaunchKernelGGL(lnr_stats_kernel, dim3(ssv_cdiv(L, 256), nln, B), dim3(256), 0, st, X, x_bs, (const float*)part, stats_out, C, L, nln, cpl);
  return ssv_check_launch("lnr_stats");
}

int ssv_lnr_gate_fwd(const float* H, long h_bs, const float* X, long x_bs, const float* g1, const float* b1, const float* g2,
                     const float* b2, float* Y, long y_bs, float* stats, void* ws, int B, int C, int L, hipStream_t st) {
  float* wsf = (float*)ws;
  float* st_buf = stats ? stats : (float*)((char*)ws + a256((size_t)B * 2 * lnr_cpl(C) * 2 * L * sizeof(float)));
  SSV_TRY(lnr_stats(H, h_bs, st_buf, wsf, B, C, L, 2, st));
  dim3 grid(lnr_cpl(C), lnr_ncb(L), B);
#define K(N) hipLaunchKernelGGL(lnr_gate_apply_kernel<N>, grid, dim3(256), 0, st, H, h_bs, X, x_bs, (const float*)st_buf, g1, b1, g2, b2, Y, y_bs, C, L)
  LNR_KC(L, K);
#undef K
  return ssv_check_launch("lnr_gate_apply");
}
int ssv_lnr_act_fwd(const float* X, long x_bs, const float* gam, const float* bet, float* Y, long y_bs, float* stats, void* ws,
                    int B, int C, int L, int act, hipStream_t st) {
  float* wsf = (float*)ws;
  float* st_buf = stats ? stats : (float*)((char*)ws + a256((size_t)B * lnr_cpl(C) * 2 * L * sizeof(float)));
  SSV_TRY(lnr_stats(X, x_bs, st_buf, wsf, B, C, L, 1, st));
  dim3 grid(lnr_cpl(C), lnr_ncb(L), B);
#define K(N) hipLaunchKernelGGL(lnr_act_apply_kernel<N>, grid, dim3(256), 0, st, X, x_bs, (const float*)st_buf, gam, bet, Y, y_bs, C, L, act)
  LNR_KC(L, K);
#undef K
  return ssv_check_launch("lnr_act_apply");
}

int ssv_lnr_gate_bwd(const float* dY, long dy_bs, const float* H, const float* X, long x_bs, const float* stats, const float* g1,
                     const float* b1, const float* g2, const float* b2, float* dH, float* dXres, long dx_bs, void* ws,
                     float* pgrads /* [6][C] */, int B, int C, int L, hipStream_t st) {
  const int cpl = lnr_cpl(C), ncb = lnr_ncb(L);
  char* base = (char*)ws;
  float* part = (float*)base;
  float* m = (float*)(base + a256((size_t)B * cpl * 4 * L * sizeof(float)));
  float* rowp = (float*)((char*)m + a256((size_t)B * 4 * L * sizeof(float)));
  float* rowp2 = (float*)((char*)rowp + a256((size_t)B * ncb * 4 * C * sizeof(float)));
  dim3 grid(cpl, ncb, B);
#define K(N) hipLaunchKernelGGL(lnr_gate_bwd_sums_kernel<N>, grid, dim3(256), 0, st, dY, dy_bs, H, X, x_bs, stats, g1, b1, g2, b2, dXres, dx_bs, part, rowp, C, L, cpl)
  LNR_KC(L, K);
#undef K
  SSV_TRY(ssv_check_launch("lnr_gate_bwd_sums"));
  hipLaunchKernelGGL(lnr_bwd_means_kernel, dim3(ssv_cdiv(L, 256), 4, B), dim3(256), 0, st, (const float*)part, m, C, L, cpl, 4);
  SSV_TRY(ssv_check_launch("lnr_bwd_means"));
#define K(N) hipLaunchKernelGGL(lnr_gate_bwd_apply_kernel<N>, grid, dim3(256), 0, st, dY, dy_bs, H, X, x_bs, stats, (const float*)m, g1, b1, g2, b2, dH, rowp2, C, L)
  LNR_KC(L, K);
#undef K
  SSV_TRY(ssv_check_launch("lnr_gate_bwd_apply"));
  hipLaunchKernelGGL(lnr_rowpart_reduce_kernel, dim3(ssv_cdiv(C, 256), 4), dim3(256), 0, st, (const float*)rowp, pgrads, C, 4, B * ncb);
  hipLaunchKernelGGL(lnr_rowpart_reduce_kernel, dim3(ssv_cdiv(C, 256), 2), dim3(256), 0, st, (const float*)rowp2, pgrads + 4 * C, C, 2, B * ncb);
  return ssv_check_launch("lnr_rowpart_reduce");
}

int ssv_lnr_act_bwd(const float* dY, long dy_bs, const float* X, long x_bs, const float* stats, const float* gam, const float* bet,
                    float* dX, long dx_bs, void* ws, float* pgrads /* [3][C] */, int B, int C, int L, int act, hipStream_t st) {
  const int cpl = lnr_cpl(C), ncb = lnr_ncb(L);
  char* base = (char*)ws;
  float* part = (float*)base;
  float* m = (float*)(base + a256((size_t)B * cpl * 2 * L * sizeof(float)));
  float* rowp = (float*)((char*)m + a256((size_t)B * 2 * L * sizeof(float)));
  float* rowp2 = (float*)((char*)rowp + a256((size_t)B * ncb * 2 * C * sizeof(float)));
  dim3 grid(cpl, ncb, B);
#define K(N) hipLaunchKernelGGL(lnr_act_bwd_sums_kernel<N>, grid, dim3(256), 0, st, dY, dy_bs, X, x_bs, stats, gam, bet, part, rowp, C, L, cpl, act)
  LNR_KC(L, K);
#undef K
  SSV_TRY(ssv_check_launch("lnr_act_bwd_sums"));
  hipLaunchKernelGGL(lnr_bwd_means_kernel, dim3(ssv_cdiv(L, 256), 2, B), dim3(256), 0, st, (const float*)part, m, C, L, cpl, 2);
  SSV_TRY(ssv_check_launch("lnr_bwd_means"));
#define K(N) hipLaunchKernelGGL(lnr_act_bwd_apply_kernel<N>, grid, dim3(256), 0, st, dY, dy_bs, X, x_bs, stats, (const float*)m, gam, bet, dX, dx_bs, rowp2, C, L, act)
  LNR_KC(L, K);
#undef K
  SSV_TRY(ssv_check_launch("lnr_act_bwd_apply"));
  hipLaunchKernelGGL(lnr_rowpart_reduce_kernel, dim3(ssv_cdiv(C, 256), 2), dim3(256), 0, st, (const float*)rowp, pgrads, C, 2, B * ncb);
  hipLaunchKernelGGL(lnr_rowpart_reduce_kernel, dim3(ssv_cdiv(C, 256), 1), dim3(256), 0, st, (const float*)rowp2, pgrads + 2 * C, C, 1, B * ncb);
  return ssv_check_launch("lnr_rowpart_reduce");
}
