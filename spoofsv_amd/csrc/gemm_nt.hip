// Weight-gradient implicit GEMM for gfx950 (fp32 MFMA 16x16x4): the reduction runs over time
// (and over the batch items a workgroup owns), both operands are time-contiguous:
//   C(z,m,c,j) = sum_{b = z, z+bstep, ..} sum_t A(b,m,t) * X(b,c,t+shift[j])
// For Conv1d: A = dL/dH (rows = output channels), X = the layer input, C = dL/dW in the layout of
// nn.Conv1d.weight (m, c, j).  One workgroup (4 waves, split over M) owns a (64*WM) x (16*NTC
// channels x KT taps) tile; the KT taps reuse ONE staged copy of the input rows at column offsets
// shift[j]-smin.  With grid.z = Z the batch is split Z ways into partial slabs that a second kernel
// sums in a fixed order (bitwise reproducible, no float atomics).
//
// LDS strides (ds_read_b32, groups of 32 lanes):
//   A fragment lane l reads As[(m = l&15)][t + (l>>4)]   -> AS = 34 (2 mod 4)
//   B fragment lane l reads Xs[(c = l&15)][t + (l>>4) + off] -> XS = 2 (mod 32): banks 2c + k
#include "ssv_common.h"

template <int KT, int WM, int NTC>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const GemmNT p, const int mtiles, const int smin, const int span) {
  constexpr int KB = 32;                        // time steps per K chunk
  constexpr int BM = 64 * WM, NCH = 16 * NTC;
  constexpr int AS = KB + 2;
  constexpr int HALO = (KT == 1) ? 0 : 54;
  constexpr int XS = ((KB + HALO + 29) / 32) * 32 + 2;
  static_assert(XS >= KB + HALO, "input tile stride too small");
  constexpr int NA = BM * KB / 256;
  constexpr int NX = (NCH * XS + 255) / 256;
  __shared__ float lds[BM * AS + NCH * XS];
  float* As = lds;
  float* Xs = lds + BM * AS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned wg = ssv_xcd_order(blockIdx.x + gridDim.x * blockIdx.z, gridDim.x * gridDim.z);   // a slab's tiles share an XCD
  const int bxx = (int)(wg % gridDim.x), z = (int)(wg / gridDim.x);
  const int mt = bxx % mtiles, ct = bxx / mtiles;
  const int m0 = mt * BM, c0 = ct * NCH;
  const int W = KB + span;
  const int tchunks = (p.La + KB - 1) / KB;

  f32x4 acc[WM][KT][NTC];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < KT; ++j)
#pragma unroll
      for (int q = 0; q < NTC; ++q) acc[i][j][q] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float ra[NA], rx[NX];
  auto prefetch = [&](int b, int tc) {
    const float* __restrict__ Ab = p.A + (long)b * p.sab;
    const float* __restrict__ Xb = p.X + (long)b * p.sxb;
    const int t0 = tc * KB;
#pragma unroll
    for (int r = 0; r < NA; ++r) {
      const int e = tid + 256 * r;
      const int row = e / KB, tt = e % KB;
      const int gm = m0 + row, gt = t0 + tt;
      ra[r] = Ab[(long)min(gm, p.M - 1) * p.sam + (long)min(gt, p.La - 1) * p.sat];   // raw; masked in commit()
    }
#pragma unroll
    for (int r = 0; r < NX; ++r) {
      const int e = tid + 256 * r;
      const int cr = e / XS, col = e % XS;
      const int gc = c0 + cr, gcol = t0 + smin + col;
      rx[r] = Xb[(long)min(gc, p.Nc - 1) * p.sxc + (long)min(max(gcol, 0), p.Lx - 1) * p.sxn];
    }
  };
  auto commit = [&](int tc) {       // validity masks are applied here, one chunk after the loads were issued
    const int t0 = tc * KB;
#pragma unroll
    for (int r = 0; r < NA; ++r) {
      const int e = tid + 256 * r;
      As[(e / KB) * AS + (e % KB)] = (m0 + e / KB < p.M && t0 + e % KB < p.La) ? ra[r] : 0.f;
    }
#pragma unroll
    for (int r = 0; r < NX; ++r) {
      const int e = tid + 256 * r;
      const int cr = e / XS, col = e % XS;
      const int gcol = t0 + smin + col;
      if (e < NCH * XS) Xs[e] = (col < W && c0 + cr < p.Nc && gcol >= 0 && gcol < p.Lx) ? rx[r] : 0.f;
    }
  };

  int offj[KT];
#pragma unroll
  for (int j = 0; j < KT; ++j) offj[j] = p.shift[j] - smin;
  const int kq = lane >> 4, nq = lane & 15;
  const int arow = (wave * WM * 16 + nq) * AS;

  // flattened (batch item, time chunk) iteration space owned by this workgroup
  const int nb = (p.B - z + p.bstep - 1) / p.bstep;
  const int total = nb * tchunks;
  if (total > 0) prefetch(z, 0);
  for (int it = 0; it < total; ++it) {
    __syncthreads();
    commit(it % tchunks);
    __syncthreads();
    if (it + 1 < total) {
      const int nx = it + 1;
      prefetch(z + (nx / tchunks) * p.bstep, nx % tchunks);
    }
#pragma unroll
    for (int s = 0; s < KB / 4; ++s) {
      const int kk = s * 4 + kq;
      float a[WM];
#pragma unroll
      for (int i = 0; i < WM; ++i) a[i] = As[arow + i * 16 * AS + kk];
#pragma unroll
      for (int j = 0; j < KT; ++j)
#pragma unroll
        for (int q = 0; q < NTC; ++q) {
          const float bf = Xs[(q * 16 + nq) * XS + kk + offj[j]];
#pragma unroll
          for (int i = 0; i < WM; ++i) acc[i][j][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bf, acc[i][j][q], 0, 0, 0);
        }
    }
  }

  float* __restrict__ Cz = p.C + (long)z * p.scz;
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gm = m0 + wave * WM * 16 + i * 16 + kq * 4 + r;
      if (gm >= p.M) continue;
#pragma unroll
      for (int j = 0; j < KT; ++j)
#pragma unroll
        for (int q = 0; q < NTC; ++q) {
          const int gc = c0 + q * 16 + nq;
          if (gc < p.Nc) Cz[(long)gm * p.scm + (long)gc * p.scc + (long)j * p.scj] = acc[i][j][q][r];
        }
    }
}

template <int KT, int WM, int NTC>
static int launch_cfg(const GemmNT& g, hipStream_t st, int smin, int span) {
  const int mtiles = ssv_cdiv(g.M, 64 * WM), ctiles = ssv_cdiv(g.Nc, 16 * NTC);
  dim3 grid(mtiles * ctiles, 1, g.Z);
  hipLaunchKernelGGL((gemm_nt_kernel<KT, WM, NTC>), grid, dim3(256), 0, st, g, mtiles, smin, span);
  return ssv_check_launch("gemm_nt");
}

int ssv_launch_gemm_nt(const GemmNT& g, hipStream_t st) {
  SSV_CHECK(!g.jobs, SSV_UNSUPPORTED, "gemm_nt: job tables need the split-bf16 kernel");
  SSV_CHECK(g.M > 0 && g.Nc > 0 && g.La > 0 && g.B > 0 && g.Z > 0 && g.bstep > 0, SSV_BAD_SHAPE,
            "gemm_nt: empty problem M=%d Nc=%d La=%d B=%d Z=%d", g.M, g.Nc, g.La, g.B, g.Z);
  SSV_CHECK(g.KT == 1 || g.KT == 3, SSV_UNSUPPORTED, "gemm_nt: kernel_size %d not supported (1 or 3)", g.KT);
  SSV_CHECK(g.Z <= 65535, SSV_UNSUPPORTED, "gemm_nt: Z=%d exceeds grid.z", g.Z);
  int smin = g.shift[0], smax = g.shift[0];
  for (int j = 1; j < g.KT; ++j) { smin = g.shift[j] < smin ? g.shift[j] : smin; smax = g.shift[j] > smax ? g.shift[j] : smax; }
  const int span = smax - smin;
  SSV_CHECK(span <= 54, SSV_UNSUPPORTED, "gemm_nt: dilation halo %d exceeds 54", span);
  if (g.KT == 3) return launch_cfg<3, 2, 2>(g, st, smin, span);
  if (g.Nc > 48) return launch_cfg<1, 2, 6>(g, st, smin, span);
  return launch_cfg<1, 2, 2>(g, st, smin, span);
}
