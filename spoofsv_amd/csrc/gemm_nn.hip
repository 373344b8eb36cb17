// Implicit-GEMM Conv1d kernel for gfx950 (MI355X), fp32 in / fp32 accumulate on the matrix cores
// (v_mfma_f32_16x16x4_f32: exact fp32 fma chain, so results match the fp32 CPU oracle to rounding).
//
// One workgroup = 4 waves (256 threads) computes a (64*WM) x (16*NT) tile of C for one batch item.
// Waves split the M (output-channel) axis; every wave sweeps all NT column tiles, so one B fragment
// read from LDS feeds WM MFMAs and one A fragment feeds NT.
//
// K loop: chunks of KCH input channels x KT taps.  The chunk's weight tile (BM x KCH*KT, K order in
// memory = (channel, tap), exactly the layout of nn.Conv1d.weight) and ONE copy of the input tile
// (KCH rows x BN+halo columns) are staged in LDS; the KT taps read the same input rows at column
// offsets shift[j]-smin, so the dilation halo costs no extra global traffic and zero padding (causal
// or "same") is a predicate on the staging load, never a materialised padded tensor.
//
// LDS strides are chosen for conflict-free ds_read_b32 fragment reads (64 lanes = 2 groups of 32):
//   A fragment lane l reads As[(m = l&15)][(k = l>>4)*KT + j]  -> AS = 2 (mod 4), so m*AS covers the
//                                                                even banks and the +KT (odd) the odd
//   B fragment lane l reads Xs[(k = l>>4)][(n = l&15) + off]   -> XS = 16 (mod 32)
// Global -> LDS goes through registers (issue the next chunk's loads, run this chunk's MFMAs, then
// write), which hides HBM/L2 latency under the 32-cycle MFMAs with 2 workgroups per CU.
#include <stdio.h>
#include <stdlib.h>
#include "ssv_common.h"

template <int KT> struct NNCfg {
  static constexpr int KCH = (KT == 3) ? 16 : 32;   // input channels per K chunk
  static constexpr int KC = KCH * KT;                // K depth per chunk
  static constexpr int AS = KC + 2;                  // LDS row stride of the weight tile
  static constexpr int HALO = (KT == 1) ? 0 : 54;    // max (smax - smin) = 2 * 27
};

template <int KT, int WM, int NT, bool AVEC>
__global__ __launch_bounds__(256, (WM == 1 ? 3 : 2)) void gemm_nn_kernel(const GemmNN p, const int mtiles, const int smin, const int span) {
  using Cfg = NNCfg<KT>;
  constexpr int KCH = Cfg::KCH, KC = Cfg::KC, AS = Cfg::AS;
  constexpr int BM = 64 * WM, BN = 16 * NT;
  constexpr int XS = ((BN + Cfg::HALO + 15) / 32) * 32 + 16;
  static_assert(XS >= BN + Cfg::HALO, "input tile stride too small");
  constexpr int NA4 = BM * KC / 4 / 256;       // float4 prefetch registers (vector path)
  constexpr int NA1 = BM * KC / 256;           // scalar prefetch registers
  constexpr int NX = (KCH * XS + 255) / 256;
  __shared__ float lds[BM * AS + KCH * XS];
  float* As = lds;
  float* Xs = lds + BM * AS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned wg = ssv_xcd_order(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);   // see ssv_xcd_order
  const int bxx = (int)(wg % gridDim.x), b = (int)(wg / gridDim.x);
  const int mt = bxx % mtiles, ntile = bxx / mtiles;
  const int m0 = mt * BM, n0 = ntile * BN;
  const float* __restrict__ Ab = p.A + (long)b * p.sab;
  const float* __restrict__ Xb = p.X + (long)b * p.sxb;
  const int W = BN + span;                    // staged input columns: [n0+smin, n0+smin+W)
  const int nchunks = (p.Kc + KCH - 1) / KCH;
  const bool a_mfast = (p.sam == 1);          // "TN" operand: consecutive lanes walk m

  f32x4 acc[WM][NT];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float4 ra4[AVEC ? NA4 : 1];
  float ra1[AVEC ? 1 : NA1];
  float rx[NX];

  auto prefetch = [&](int ch) {
    const int c0 = ch * KCH;
    if constexpr (AVEC) {
#pragma unroll
      for (int r = 0; r < NA4; ++r) {
        const int f = tid + 256 * r;
        const int row = f / (KC / 4), q = f % (KC / 4);
        const int gm = m0 + row, gk = c0 * KT + 4 * q;
        // raw load on a clamped, always-legal address; the validity mask is applied in commit(), one chunk later
        // (masking here would make hipcc wait for -- or branch around -- every load)
        ra4[r] = *reinterpret_cast<const float4*>(Ab + (long)min(gm, p.M - 1) * p.sam + min(gk, p.Kc * KT - 4));
      }
    } else {
#pragma unroll
      for (int r = 0; r < NA1; ++r) {
        const int e = tid + 256 * r;
        int row, kk;
        if (a_mfast) { kk = e / BM; row = e % BM; } else { row = e / KC; kk = e % KC; }
        const int c = c0 + kk / KT, j = kk % KT, gm = m0 + row;
        ra1[r] = Ab[(long)min(gm, p.M - 1) * p.sam + (long)min(c, p.Kc - 1) * p.sac + (long)j * p.saj];
      }
    }
#pragma unroll
    for (int r = 0; r < NX; ++r) {
      const int e = tid + 256 * r;
      const int kr = e / XS, col = e % XS;
      const int c = c0 + kr, gcol = n0 + smin + col;
      rx[r] = Xb[(long)min(c, p.Kc - 1) * p.sxc + (long)min(max(gcol, 0), p.Lx - 1) * p.sxn];
    }
  };

  auto commit = [&](int ch) {
    const int c0 = ch * KCH;
    if constexpr (AVEC) {
#pragma unroll
      for (int r = 0; r < NA4; ++r) {
        const int f = tid + 256 * r;
        const int row = f / (KC / 4), q = f % (KC / 4);
        const bool ok = m0 + row < p.M && c0 * KT + 4 * q < p.Kc * KT;
        float2* dst = reinterpret_cast<float2*>(As + row * AS + 4 * q);   // AS even -> 8-byte aligned
        dst[0] = ok ? make_float2(ra4[r].x, ra4[r].y) : make_float2(0.f, 0.f);
        dst[1] = ok ? make_float2(ra4[r].z, ra4[r].w) : make_float2(0.f, 0.f);
      }
    } else {
#pragma unroll
      for (int r = 0; r < NA1; ++r) {
        const int e = tid + 256 * r;
        int row, kk;
        if (a_mfast) { kk = e / BM; row = e % BM; } else { row = e / KC; kk = e % KC; }
        As[row * AS + kk] = (m0 + row < p.M && c0 + kk / KT < p.Kc) ? ra1[r] : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < NX; ++r) {
      const int e = tid + 256 * r;
      const int kr = e / XS, col = e % XS;
      const int gcol = n0 + smin + col;
      if (e < KCH * XS) Xs[e] = (col < W && c0 + kr < p.Kc && gcol >= 0 && gcol < p.Lx) ? rx[r] : 0.f;
    }
  };

  int offj[KT];
#pragma unroll
  for (int j = 0; j < KT; ++j) offj[j] = p.shift[j] - smin;

  const int arow = (wave * WM * 16 + (lane & 15)) * AS;
  const int kq = lane >> 4, nq = lane & 15;

  prefetch(0);
  for (int ch = 0; ch < nchunks; ++ch) {
    __syncthreads();            // every wave is done reading the previous chunk
    commit(ch);
    __syncthreads();
    if (ch + 1 < nchunks) prefetch(ch + 1);
#pragma unroll
    for (int j = 0; j < KT; ++j) {
#pragma unroll
      for (int s = 0; s < KCH / 4; ++s) {
        const int kk = s * 4 + kq;
        float a[WM], bf[NT];
#pragma unroll
        for (int i = 0; i < WM; ++i) a[i] = As[arow + i * 16 * AS + kk * KT + j];
#pragma unroll
        for (int t = 0; t < NT; ++t) bf[t] = Xs[kk * XS + t * 16 + nq + offj[j]];
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bf[t], acc[i][t], 0, 0, 0);
      }
    }
  }

  // Epilogue.  C/D layout of the 16x16 tile: column = lane & 15, row = (lane >> 4) * 4 + r.
  float* __restrict__ Cb = p.C + (long)b * p.scb;
  const float* __restrict__ Rb = p.R ? p.R + (long)b * p.srb : nullptr;
  // every load of the epilogue -- bias terms, the residual (the in-place data gradient: R = C) -- in front of the first store (round 6): a load behind a
  // store is waited for together with it (one in-order vmcnt), and with the residual read inside the store loop the WM x 4 x NT elements of a lane
  // ran one memory round trip apart (see gemm_nn_bf3_kernel's epilogue, conv_nn.hip)
  float addv[WM][4], resv[WM][4][NT];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gm = m0 + wave * WM * 16 + i * 16 + kq * 4 + r;
      const bool rok = gm < p.M;
      float add = 0.f;
      if (rok && p.bias) add += p.bias[gm];
      if (rok && p.bias_b) add += p.bias_b[(long)b * p.sbb + gm];
      addv[i][r] = add;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int gn = n0 + t * 16 + nq;
        resv[i][r][t] = (Rb && rok && gn < p.N) ? Rb[(long)gm * p.srm + (long)gn * p.srn] : 0.f;
      }
    }
#pragma unroll
  for (int i = 0; i < WM; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gm = m0 + wave * WM * 16 + i * 16 + kq * 4 + r;
      if (gm >= p.M) continue;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int gn = n0 + t * 16 + nq;
        if (gn >= p.N) continue;
        Cb[(long)gm * p.scm + (long)gn * p.scn] = acc[i][t][r] * p.alpha + addv[i][r] + resv[i][r][t];
      }
    }
  }
}

template <int KT, int WM, int NT, bool AVEC>
static int launch_cfg(const GemmNN& g, hipStream_t st, int smin, int span) {
  constexpr int BM = 64 * WM, BN = 16 * NT;
  const int mtiles = ssv_cdiv(g.M, BM), ntiles = ssv_cdiv(g.N, BN);
  dim3 grid(mtiles * ntiles, g.B, 1);
  hipLaunchKernelGGL((gemm_nn_kernel<KT, WM, NT, AVEC>), grid, dim3(256), 0, st, g, mtiles, smin, span);
  return ssv_check_launch("gemm_nn");
}

// Tile choice.  Every workgroup of these grids is resident at once (<= 4 per CU), so the kernel lasts as long
// as the most loaded CU: ceil(tiles / 256) workgroups, each costing about its MFMA count plus a per-chunk
// staging/barrier overhead that grows with the tile perimeter.
static void pick_tile(const GemmNN& g, int* wm_out, int* nt_out) {
  static const int nts[] = {8, 7, 6, 4, 2};
  double best = 1e30;
  for (int wm = 1; wm <= 2; ++wm)
    for (int nt : nts) {
      const long tiles = (long)ssv_cdiv(g.M, 64 * wm) * ssv_cdiv(g.N, 16 * nt) * g.B;
      const double per_tile = (double)wm * nt + 0.35 * wm + 0.12 * nt + 0.6;
      const double cost = (double)((tiles + 255) / 256) * per_tile;
      if (cost < best) { best = cost; *wm_out = wm; *nt_out = nt; }
    }
}

template <int KT, bool AVEC>
static int launch_nt(const GemmNN& g, hipStream_t st, int smin, int span) {
  int wm = 2, nt = 8;
  pick_tile(g, &wm, &nt);
#define SSV_CASE(WM_, NT_) if (wm == WM_ && nt == NT_) return launch_cfg<KT, WM_, NT_, AVEC>(g, st, smin, span)
  SSV_CASE(2, 8); SSV_CASE(2, 7); SSV_CASE(2, 6); SSV_CASE(2, 4); SSV_CASE(2, 2);
  SSV_CASE(1, 8); SSV_CASE(1, 7); SSV_CASE(1, 6); SSV_CASE(1, 4); SSV_CASE(1, 2);
#undef SSV_CASE
  return ssv_fail(SSV_UNSUPPORTED, "gemm_nn: no tile %d,%d", wm, nt);
}

int ssv_launch_gemm_nn(const GemmNN& g, hipStream_t st) {
  SSV_CHECK(g.M > 0 && g.N > 0 && g.Kc > 0 && g.B > 0, SSV_BAD_SHAPE, "gemm_nn: empty problem M=%d N=%d Kc=%d B=%d", g.M, g.N, g.Kc, g.B);
  SSV_CHECK(g.KT == 1 || g.KT == 3, SSV_UNSUPPORTED, "gemm_nn: kernel_size %d not supported (1 or 3)", g.KT);
  SSV_CHECK(g.B <= 65535, SSV_UNSUPPORTED, "gemm_nn: batch %d exceeds grid.y", g.B);
  int smin = g.shift[0], smax = g.shift[0];
  for (int j = 1; j < g.KT; ++j) { smin = g.shift[j] < smin ? g.shift[j] : smin; smax = g.shift[j] > smax ? g.shift[j] : smax; }
  const int span = smax - smin;
  SSV_CHECK(span <= NNCfg<3>::HALO, SSV_UNSUPPORTED, "gemm_nn: dilation halo %d exceeds %d", span, NNCfg<3>::HALO);
  // 16-byte vector path for the weight operand: K-contiguous rows, 16-byte aligned
  const bool avec = (g.KT == 1 || g.saj == 1) && g.sac == g.KT && (g.sam % 4) == 0 && (g.sab % 4) == 0 &&
                    ((long)g.Kc * g.KT) % 4 == 0 && (reinterpret_cast<uintptr_t>(g.A) & 15) == 0;
  if (g.KT == 3) return avec ? launch_nt<3, true>(g, st, smin, span) : launch_nt<3, false>(g, st, smin, span);
  return avec ? launch_nt<1, true>(g, st, smin, span) : launch_nt<1, false>(g, st, smin, span);
}
