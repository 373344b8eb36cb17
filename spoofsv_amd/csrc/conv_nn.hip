// Implicit-GEMM Conv1d forward / data gradient on split MFMAs: gemm_nn_bf3_kernel (4 waves) and gemm_nn_bf3w_kernel (8-16 waves), their tile choice and launchers.
#include "bf3_common.h"

// ---- NN ---------------------------------------------------------------------------------------------------------------
// Waves split the M axis, so a weight row is only ever used by ONE wave: weight fragments go straight from global memory
// (L2-resident, pre-split, fragment-shaped 16-byte loads) into MFMA operand registers, one K chunk ahead (two register
// sets, the chunk loop is unrolled by two).  Only the input tile, which all four waves share, is staged in LDS -- this
// removes 2/3 of the LDS writes and 1/4 of the LDS reads of a version that staged both operands.
// (third waves per SIMD for the small k = 1 tile: 168 VGPRs in the split-bf16 form, 174 in the split-fp16 one without the bound -- +22 % time)
// Tuning builds only (-DSSV_NN_STAMP): thread 0 of workgroup (0, 0) records s_memtime at five points of every K chunk; ssv_debug_nn_stamps().
#ifdef SSV_NN_STAMP
#ifndef SSV_NN_STAMP_WG
#define SSV_NN_STAMP_WG 0     // (stamp builds) 1: the launch's last workgroup instead of its first
#endif
__device__ unsigned long long ssv_nn_stamps[128];
__device__ unsigned long long ssv_nn_rt[4096];        // s_memrealtime (100 MHz, one clock for the device) at entry / exit of the first 2048 workgroups
extern "C" int ssv_debug_nn_realtime(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ssv_nn_rt), sizeof(ssv_nn_rt)); }
#define NN_RT(which) do { const unsigned w_ = blockIdx.y * gridDim.x + blockIdx.x; if (threadIdx.x == 0 && w_ < 2048u) ssv_nn_rt[2 * w_ + (which)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define NN_STAMP(k) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && (unsigned)ch < 15u) ssv_nn_stamps[ch * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#define NN_STAMP_AT(slot) do { if ((SSV_NN_STAMP_WG ? (blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1) : (blockIdx.x == 0 && blockIdx.y == 0)) && threadIdx.x == 0) \
    ssv_nn_stamps[120 + (slot)] = __builtin_readcyclecounter(); } while (0)       /* SSV_NN_STAMP_WG=1: the launch's last workgroup instead of its first */
extern "C" int ssv_debug_nn_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ssv_nn_stamps), sizeof(ssv_nn_stamps)); }
#else
#define NN_STAMP(k) do {} while (0)
#define NN_STAMP_AT(slot) do {} while (0)
#define NN_RT(which) do {} while (0)
#endif

// HW: the halo (columns beyond the tile that the taps reach) the instantiation stages for -- 54 (dilation 27, any form) or 16: most layers of the
// models have dilation 1 or 3, and with the 54-column halo a third of the loads, splits and LDS stores of their chunks went into columns no tap reads
// (112 + 54 -> 176 staged columns = 3 slots per thread; 112 + 16 -> 128 = 2).
template <int KT, int WM, int NT, int EPI, int F16, int HW = 54>
__global__ __launch_bounds__(256, SSV_NNB_WAVES(KT, WM, NT, EPI, HW)) void gemm_nn_bf3_kernel(const GemmNNB p, const int mtiles, const int smin, const int span) {
  constexpr int BM = 64 * WM, BN = 16 * NT;
  constexpr int HALO = (KT == 1) ? 0 : HW;
  constexpr int WX = ((BN + HALO + 15) / 16) * 16;         // staged columns, plane = WX*16 B = multiple of 256 B
  constexpr int X_SLOTS = 4 * WX;
  constexpr int NX = (X_SLOTS + 255) / 256;
  // two images of the staged input tile: the MFMAs of chunk c read image c & 1 while chunk c+1 is split into the other
  // one -- one barrier per chunk, and the split (VALU) runs under the MFMAs instead of between two barriers
  // (the epilogue re-uses the memory to turn the accumulator tiles into row-contiguous stores: 4 waves x 16 rows x (BN + 4))
  // (+ 4 * WM * BN (mean, M2) pairs behind the parked tiles when the caller wants the output's column statistics)
  constexpr int IMG = 2 * X_SLOTS, EPI_U4 = (4 * 16 * (BN + 4) + 4 * WM * BN * 2) / 4;
  constexpr int LDS_U4 = 2 * IMG > EPI_U4 ? 2 * IMG : EPI_U4;
  __shared__ uint4 lds_all[LDS_U4];
  uint4 (*lds)[IMG] = reinterpret_cast<uint4 (*)[IMG]>(lds_all);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned wg = ssv_xcd_order(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);   // see ssv_xcd_order
  const int bxx = (int)(wg % gridDim.x);
  int b = (int)(wg / gridDim.x), kz = 0;
  const int mt = bxx % mtiles, ntile = bxx / mtiles;
  const int m0 = mt * BM, n0 = ntile * BN;
  if constexpr (EPI == 0 && KT == 1) {                 // split reduction (GemmNNB::ksplit): grid.y entry = (batch item, K range)
    if (p.ksplit > 1) {
      kz = b % p.ksplit;
      b /= p.ksplit;
      if (b == 0 && m0 + BM <= p.skip_rows) return;    // (whole workgroup, before any barrier)
    }
  }
  const float* __restrict__ Xb = p.X + (long)b * p.sxb + (long)kz * p.Kc * p.sxc;
  // LSTM wavefront (see GemmNNB): layer / frame of this grid.y entry, the second K segment, the chunks to run
  const float* __restrict__ X2b = nullptr;
  int lstm_layer = 0, lstm_t = 0;
  bool lstm_l0 = false;                       // this entry is layer 0 riding in a wavefront launch (GemmNNB::A0hi)
  if constexpr (EPI == 1) {
    if (p.lstm_D > 0) {
      lstm_layer = p.lstm_lo + b;
      lstm_t = p.lstm_s - lstm_layer;
      const long HN = (long)p.perm_h * p.N;
      lstm_l0 = p.A0hi != nullptr && lstm_layer == 0;
      Xb = p.lstm_out + ((long)max(lstm_layer - 1, 0) * p.lstm_D + lstm_t % p.lstm_D) * HN;
      X2b = p.lstm_out + ((long)lstm_layer * p.lstm_D + (lstm_t + p.lstm_D - 1) % p.lstm_D) * HN - (long)p.xsplit * 32 * (long)p.sxc;
      if (lstm_l0) { Xb = X2b + (long)p.xsplit * 32 * (long)p.sxc; X2b = nullptr; }      // one segment: the layer's own h_{t-1}
    }
  }
  // XS (HW = 3 marks the instantiation; GemmNNB::hs_planes): the same two K segments read from the pre-split planes of (layer - 1, slot t % 2) and
  // (layer, slot (t - 1) % 2); the second is pre-offset by its first chunk so that one chunk formula serves both, as X2b above.
  constexpr bool XS = EPI == 1 && KT == 1 && NT == 8 && HW == 3;
  const char* Pb = nullptr;
  const char* P2b = nullptr;
  int xsp = p.xsplit;                           // first chunk of this entry's second K segment
  bool x0fold = false;
  if constexpr (XS) {
    const long pl = p.hs_plane_bytes, chunkb = (long)4 * p.hs_npad * 16;         // bytes of one chunk (4 k-groups) in a plane
    const char* base = reinterpret_cast<const char*>(p.hs_planes);
    Pb = base + ((long)max(lstm_layer - 1, 0) * 2 + (lstm_t & 1)) * 2 * pl;
    P2b = base + ((long)lstm_layer * 2 + ((lstm_t + 1) & 1)) * 2 * pl - (long)p.xsplit * chunkb;
    if (lstm_l0) {
      Pb = P2b + (long)p.xsplit * chunkb; P2b = nullptr;
      if (p.x0_planes) {                        // layer 0's input frame as the first K segment (GemmNNB::x0_planes), its own h_{t-1} as the second
        x0fold = true; xsp = p.xsplit0;
        P2b = Pb - (long)xsp * chunkb;
        Pb = reinterpret_cast<const char*>(p.x0_planes) + (long)lstm_t * 2 * pl;
      }
    }
  }
  // chunks to run: all of K, except that an LSTM entry at its first frame has no h_{t-1} segment (layer 0 riding along: nothing but that segment)
  const int nchunks_all = lstm_l0 ? (x0fold ? xsp + p.perm_h / 32 : p.xsplit) : p.Kpad / 32;
  const int nchunks_ = (EPI == 1 && p.lstm_D > 0 && lstm_t == 0) ? (lstm_l0 ? (x0fold ? xsp : 0) : p.xsplit) : (EPI == 0 && KT == 1 && p.ksplit > 1) ? p.Kc / 32 : nchunks_all;
  // (tuning builds, -DSSV_LSTM_ABL: bit 0 = the pre-split wavefront's cells store nothing, bit 1 = its tiles run two chunks; profiles/round6_ge2e_cell_epilogue.txt (6))
  const int nchunks = (SSV_LSTM_ABL & 2) && XS ? (nchunks_ < 2 ? nchunks_ : 2) : nchunks_;
  const int W = BN + span;
  const int kq = lane >> 4, nq = lane & 15;

  f32x4 acc[WM][NT];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Weight fragments.  KT == 3: one register set; tap j of the NEXT chunk is re-loaded into its registers right after
  // tap j's MFMAs of the current chunk have been issued (2/3 of a chunk of lead time).  KT == 1: two sets, alternating.
  constexpr int NSET = (KT == 1) ? 2 : 1;
  uint4 Ah_[NSET][KT][WM], Al_[NSET][KT][WM];
  float rx[NX][8];

  // this lane's weight fragments: row block (m0 + wave*WM*16 + i*16) / 16, 16 bytes at lane*16 of each 1 KB (mb, chunk) block
  const int MB = (p.M + 15) >> 4;
  long arow[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int mb = (m0 + wave * WM * 16 + i * 16) >> 4;
    arow[i] = (long)min(mb, MB - 1) * nchunks_all * 512 + lane * 8;     // blocks past M re-read the last one: never stored
  }
  const long aplane = (long)MB * nchunks_all * 512;

  // uniform byte base (batch item / layer, tap, chunk) + a per-lane 32-bit byte offset: the saddr form of global_load
  unsigned arowb[WM];
#pragma unroll
  for (int i = 0; i < WM; ++i) arowb[i] = (unsigned)(arow[i] * 2);
  // (LSTM wavefront with layer 0 riding along: entry 0 reads the planes A0hi / A0lo, entry b >= 1 the planes of layer b at (b - 1) * sab)
  const long aent = (EPI == 1 && p.A0hi) ? (long)(b > 0 ? b - 1 : 0) * p.sab : (long)b * p.sab + (long)kz * (p.Kc / 32) * 512;
  const __amdgpu_buffer_rsrc_t rsAh = ssv_buf(lstm_l0 ? p.A0hi : p.Ahi + aent), rsAl = ssv_buf(lstm_l0 ? p.A0lo : p.Alo + aent);   // (see ssv_buf)
  auto loadA = [&](int set, int j, int ch) {
    // wave-uniform byte offset, said so: in the LSTM instantiations hipcc kept it in a vector register and wrapped every weight load in a waterfall loop
    // (readfirstlane + compare + branch per load, and its wait counts no longer counted: vmcnt(0) in front of each chunk pair's first MFMA; round 6)
    const unsigned ub = (unsigned)__builtin_amdgcn_readfirstlane((int)((j * aplane + (long)ch * 512) * 2));
    // (buffer loads everywhere: equal or 1-3 % faster than loads through pointers, measured in-step per tile)
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      Ah_[set][j][i] = ssv_buf_u4(rsAh, arowb[i], ub);
      Al_[set][j][i] = ssv_buf_u4(rsAl, arowb[i], ub);
    }
  };
  // Input staging, two halves.  prefetchX only ISSUES loads (raw values, addresses clamped into the batch item so every
  // load is legal); validity masks are applied in commitX one chunk later, right before the split.  A mask applied at
  // load time would make hipcc wait for each load (or branch around it), serialising 24 L2 round trips per chunk.
  // Address arithmetic is hoisted: element (chunk ch, slot channel 8*kg+i, column) lives at
  //   [Xb + (ch*32 + i)*L]  (wave-uniform, scalar ALU)  +  [8*kg*L + column]  (per thread, computed once),
  // and the column mask is computed once; only a ragged last chunk (Kc % 32 != 0) needs per-channel clamps and masks.
  const int Lrow = (int)p.sxc;
  const __amdgpu_buffer_rsrc_t rsX = ssv_buf(Xb), rsX2 = ssv_buf(X2b ? X2b : Xb);        // (see ssv_buf)
  const __amdgpu_buffer_rsrc_t rsP = ssv_buf(XS ? Pb : nullptr), rsP2 = ssv_buf(XS ? (P2b ? P2b : Pb) : nullptr);
  uint4 rxs[XS ? NX : 1][2];                                                     // XS: the slot's hi and lo vectors as they lie in the planes
  unsigned voff[NX], voffb[NX];
  bool cvs[NX];
#pragma unroll
  for (int r = 0; r < NX; ++r) {
    const int e = tid + 256 * r;
    const int kg = e / WX, col = e % WX;
    const int gcol = n0 + smin + col;
    cvs[r] = e < X_SLOTS && col < W && gcol >= 0 && gcol < p.Lx;
    voff[r] = (unsigned)((e < X_SLOTS ? 8 * kg : 0) * Lrow + min(max(gcol, 0), p.Lx - 1) * p.sxn);
    voffb[r] = voff[r] * 4u;           // BYTE offset of the buffer load (a row's offset is added as its scalar operand)
  }
  const bool ragged = (p.Kc & 31) != 0;
  // split-fp16: xs = 2^ex scales the input while it is split, us = 2^-(ea + ex) the accumulators in the epilogue.  The weights'
  // inverse scale is requested here (a scalar load) and first USED in the epilogue: nothing in the prologue waits for it.  The input's
  // scale is needed before the first split; x_namax == 0 (the LSTM products: |h| < 1 by construction) means the constant 2^14, no list.
  float xs = 1.f, xinv = 1.f, ainv = 1.f, x0fac = 1.f;
  if constexpr (F16) ainv = *p.a_inv;
  auto scales = [&]() {
    if constexpr (F16) {
      if (p.x_namax == 0) {
        xs = 16384.f; xinv = 1.f / 16384.f;
        if constexpr (XS) {
          if (x0fold) {                         // the input frames were split with their own scale 2^e: segment 1's sums, times 2^(14 - e), join segment 2's
            float sc, inv;
            ssv_pow2_scale(ssv_wave_list_max(p.x0_amax, 64), sc, inv);
            x0fac = ssv_uniform(16384.f * inv);
          }
        }
      }
      else {
        float sc, inv;
        ssv_pow2_scale(ssv_wave_list_max(p.x_amax + (long)b * p.x_amax_bs, p.x_namax), sc, inv);
        xs = ssv_uniform(sc);
        xinv = ssv_uniform(inv);
      }
    }
  };
  // Convolutions (EPI == 0) prefetch on ONE path: a buffer whose range is the batch item's Kc rows, so the channels of a ragged last chunk past
  // Kc read 0 (tools/probe/buf_oob.hip: voffset + soffset is checked against the range, per dword) and no "last, partial chunk" form is needed.
  // Not for the branch: hipcc lays an if / else out as two tests in a row, its s_waitcnt bookkeeping then sees a path on which NEITHER form ran, and
  // in front of every chunk's first MFMA it waited for all but the weight fragments' own loads -- i.e. for the input loads issued a few hundred
  // cycles earlier, one exposed round trip per chunk (round 5; the steady / tail split below had removed only the "is there a chunk c + 2" tests).
  const __amdgpu_buffer_rsrc_t rsXr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Xb), 0,
      (int)(((long)(p.Kc - 1) * Lrow + (long)(p.Lx - 1) * p.sxn + 1) * 4), 0x00020000);
  auto prefetchX = [&](int ch) {
    if constexpr (EPI == 0 && SSV_NN_XONE && (KT == 1 || (WM == 2 && NT == 7 && HW == 16))) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const unsigned so = (unsigned)((ch * 32 + i) * Lrow) * 4u;                // uniform row offset: scalar arithmetic
#pragma unroll
        for (int r = 0; r < NX; ++r) rx[r][i] = ssv_buf_f32(rsXr, voffb[r], so);
      }
    } else if constexpr (XS) {
      const bool seg2 = P2b && ch >= xsp;
      const __amdgpu_buffer_rsrc_t rs = seg2 ? rsP2 : rsP;
      const unsigned so = (unsigned)ch * (unsigned)(4 * p.hs_npad * 16);           // uniform: the chunk's four k-groups
#pragma unroll
      for (int r = 0; r < NX; ++r) {
        const unsigned vo = (unsigned)(((tid + 256 * r) / WX) * p.hs_npad + n0 + (tid + 256 * r) % WX) * 16u;
        rxs[r][0] = ssv_buf_u4(rs, vo, so);
        rxs[r][1] = ssv_buf_u4(rs, vo, so + (unsigned)p.hs_plane_bytes);
      }
    } else if constexpr (EPI == 1) {
      // LSTM products: ONE path too (round 6).  K is a whole number of chunks there (checked by the launcher), so no ragged form exists, and the two K
      // segments differ only in the buffer descriptor -- a scalar select.  With the two forms below hipcc's wait counts for this loop collapsed to
      // vmcnt(0) in front of every chunk pair's first MFMA, i.e. a wait for the input loads issued just before the barrier.
      const bool seg2 = X2b && ch >= p.xsplit;                                   // (the h_{t-1} segment of K)
      const __amdgpu_buffer_rsrc_t rs = seg2 ? rsX2 : rsX;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const unsigned so = (unsigned)((ch * 32 + i) * Lrow) * 4u;
#pragma unroll
        for (int r = 0; r < NX; ++r) rx[r][i] = ssv_buf_f32(rs, voffb[r], so);
      }
    } else if (!ragged || ch + 1 < nchunks) {
      const bool seg2 = EPI == 1 && X2b && ch >= p.xsplit;                       // (LSTM: the h_{t-1} segment of K)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if constexpr (SSV_NN_XBUF(KT, WM, NT)) {
          const unsigned so = (unsigned)((ch * 32 + i) * Lrow) * 4u;              // uniform row offset: scalar arithmetic
#pragma unroll
          for (int r = 0; r < NX; ++r) rx[r][i] = (EPI == 1 && seg2) ? ssv_buf_f32(rsX2, voffb[r], so) : ssv_buf_f32(rsX, voffb[r], so);
        } else {
          const char* __restrict__ rowp = (const char*)((seg2 ? X2b : Xb) + (long)(ch * 32 + i) * Lrow);     // uniform
#pragma unroll
          for (int r = 0; r < NX; ++r) rx[r][i] = *reinterpret_cast<const float*>(rowp + voffb[r]);
        }
      }
    } else {                                                                   // last, partial chunk: clamp channels
#pragma unroll
      for (int r = 0; r < NX; ++r) {
        const int e = tid + 256 * r;
        const int kg = (e < X_SLOTS) ? e / WX : 0;
        const unsigned colo = voff[r] - (unsigned)(8 * kg * Lrow);
#pragma unroll
        for (int i = 0; i < 8; ++i) rx[r][i] = ssv_buf_f32(rsX, ((unsigned)min(ch * 32 + 8 * kg + i, p.Kc - 1) * (unsigned)Lrow + colo) * 4u, 0u);
      }
    }
  };
  auto commitX = [&](int ch) {
    uint4* Xh = lds[ch & 1];
    uint4* Xl = lds[ch & 1] + X_SLOTS;
    const bool last_ragged = EPI == 0 && ragged && ch + 1 == nchunks;           // (the LSTM products have no ragged chunk)
    if constexpr (XS) {                                                          // already split, already in slot order: straight into the image
#pragma unroll
      for (int r = 0; r < NX; ++r) { Xh[tid + 256 * r] = rxs[r][0]; Xl[tid + 256 * r] = rxs[r][1]; }
      return;
    }
#pragma unroll
    for (int r = 0; r < NX; ++r) {
      const int e = tid + 256 * r;
      if (e < X_SLOTS) {
        float v[8];
        if (!last_ragged) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = cvs[r] ? rx[r][i] : 0.f;
        } else {
          const int kg = e / WX;
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = (cvs[r] && ch * 32 + 8 * kg + i < p.Kc) ? rx[r][i] : 0.f;
        }
        uint4 h, l;
        split8s<F16>(v, xs, h, l);
        Xh[e] = h; Xl[e] = l;            // slot index = kg*WX + col = e
      }
    }
  };

  int offj[KT];
#pragma unroll
  for (int j = 0; j < KT; ++j) offj[j] = p.shift[j] - smin;

  // The input fragments of column block t + FD are read from LDS before the MFMAs of block t are issued (2 reads, 3 WM MFMAs per
  // block): hipcc on its own issues a block's reads right in front of its MFMAs and parks the wave for the LDS latency NT times per tap.
  constexpr int FD = NT > 1 ? 1 : 0;
  auto tap = [&](int set, int j, int ch) {
    const uint4* Xh = lds[ch & 1];
    const uint4* Xl = lds[ch & 1] + X_SLOTS;
    uint4 fb[FD + 1][2];
    auto frag = [&](int t, uint4 (&f)[2]) __attribute__((always_inline)) {
      const int xs_ = kq * WX + t * 16 + nq + offj[j];
      f[0] = Xh[xs_]; f[1] = Xl[xs_];
    };
#pragma unroll
    for (int t = 0; t < FD; ++t) frag(t, fb[t]);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t + FD < NT) frag(t + FD, fb[(t + FD) % (FD + 1)]);
      if (FD > 0) __builtin_amdgcn_sched_barrier(0);                          // or the scheduler sinks the reads back to their use
      const uint4 bh = fb[t % (FD + 1)][0];
      const uint4 bl = fb[t % (FD + 1)][1];
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        acc[i][t] = mma16<F16>(Al_[set][j][i], bh, acc[i][t]);
        acc[i][t] = mma16<F16>(Ah_[set][j][i], bl, acc[i][t]);
        acc[i][t] = mma16<F16>(Ah_[set][j][i], bh, acc[i][t]);
      }
    }
  };

  // Prologue: chunk 0 staged, chunk 1 in flight.  Chunk c: MFMAs on image c & 1 (weight fragments of chunk c+1 re-loaded
  // tap by tap behind them), then chunk c+1 is split into the other image and the loads of chunk c+2 are issued.
  // The loop comes in two forms: STEADY, for the chunks whose every load / commit is known to be due, has no "is there a chunk
  // c + 2" tests -- not for the branches: hipcc's s_waitcnt bookkeeping merges over all paths, and with the prefetch under a test it
  // waited for vmcnt(0) in front of each chunk's first MFMA, i.e. for the input loads issued a few hundred cycles earlier (one
  // exposed L2 round trip per chunk; the weight-gradient kernel has the full story at its STEADY).  The last chunks run the tested form.
  using ST_ = std::integral_constant<bool, true>;
  using TL_ = std::integral_constant<bool, false>;
  NN_STAMP_AT(0);
  NN_RT(0);
#ifdef SSV_NN_STAMP
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) ssv_nn_stamps[127] = __builtin_readcyclecounter();     // the first workgroup's entry, for the ramp
#endif
  if constexpr (KT == 1) {
    NN_STAMP_AT(1);
    if (nchunks > 0) {
      loadA(0, 0, 0);
      prefetchX(0);
      scales();
      commitX(0);
      if (nchunks > 1) { prefetchX(1); loadA(1, 0, 1); }
    }
    __syncthreads();
    auto pair = [&](auto steady, int ch) __attribute__((always_inline)) -> bool {
      constexpr bool ST = decltype(steady)::value;
      NN_STAMP(0);
      tap(0, 0, ch);
      NN_STAMP(1);
      if (!ST && ch + 1 >= nchunks) return false;
      commitX(ch + 1);
      NN_STAMP(2);
      if (ST || ch + 2 < nchunks) { prefetchX(ch + 2); loadA(0, 0, ch + 2); }
      NN_STAMP(3);
      __syncthreads();
      NN_STAMP(4);
      tap(1, 0, ch + 1);
      if (ST || ch + 2 < nchunks) {
        commitX(ch + 2);
        if (ST || ch + 3 < nchunks) { prefetchX(ch + 3); loadA(1, 0, ch + 3); }
      }
      __syncthreads();
      NN_STAMP(5);
      return true;
    };
    auto x0_rescale = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[i][t] *= x0fac;
    };
    int ch = 0;
    for (; ch + 3 < nchunks; ch += 2) {
      if constexpr (XS) if (x0fold && ch == xsp) x0_rescale();       // (xsp is even: a pair never straddles the segments; VALU only on the branch)
      pair(ST_{}, ch);
    }
    for (; ch < nchunks; ch += 2) {
      if constexpr (XS) if (x0fold && ch == xsp) x0_rescale();
      if (!pair(TL_{}, ch)) break;
    }
    if constexpr (XS) if (x0fold && nchunks == xsp) x0_rescale();      // frame 0: no second segment
  } else {
#pragma unroll
    for (int j = 0; j < KT; ++j) loadA(0, j, 0);
    prefetchX(0);
    scales();
    commitX(0);
    if (nchunks > 1) prefetchX(1);
    __syncthreads();
    NN_STAMP_AT(1);
    auto chunk = [&](auto steady, int ch) __attribute__((always_inline)) {
      constexpr bool ST = decltype(steady)::value;
      const bool more = ST || ch + 1 < nchunks;
      NN_STAMP(0);
#pragma unroll
      for (int j = 0; j < KT; ++j) {
        tap(0, j, ch);
        __builtin_amdgcn_sched_barrier(0);      // keep the re-load behind this tap's MFMAs, and later taps' LDS reads behind it
        if (more) loadA(0, j, ch + 1);
      }
      NN_STAMP(1);
      if (more) {
        commitX(ch + 1);
        NN_STAMP(2);
        if (ST || ch + 2 < nchunks) prefetchX(ch + 2);
      }
      NN_STAMP(3);
      __syncthreads();
      NN_STAMP(4);
    };
    int ch = 0;
    // (the steady form holds more values live: 140 -> 190 VGPRs for the 64 x 112 tile, whose hot launches are 768 workgroups and need
    // three per CU to run in one round -- +20 % on it; the 64 x 96 tile's launches are 512 workgroups and gain 7 % from it)
    for (; ch + 2 < nchunks; ++ch) chunk(ST_{}, ch);
    for (; ch < nchunks; ++ch) chunk(TL_{}, ch);
  }

  NN_STAMP_AT(2);
  const float us = F16 ? ssv_uniform(xinv * ainv) : 1.f;
  float* __restrict__ Cb = p.C + (long)b * p.scb + (long)kz * p.scz;
  const float* __restrict__ Rb = p.R ? p.R + (long)b * p.srb : nullptr;
  if constexpr (EPI == 1) {
    if (p.A0hi && !lstm_l0) Rb = nullptr;         // the input projection in R belongs to layer 0 alone
    // Fused LSTM cell (torch gate order i, f, g, o).  Rows were packed gate-interleaved, so the four accumulator rows a
    // lane holds for a 16-row tile (rows kq*4 .. kq*4+3) are the four gates of ONE hidden unit at column nq.
    const int H = p.perm_h;
    float* cst = p.cstate;
    const float* __restrict__ bia = p.bias ? p.bias + (long)b * p.sbb : nullptr;
    const float* __restrict__ bib = p.bias_b ? p.bias_b + (long)b * p.sbb : nullptr;
    bool first = p.first != 0;
    float* cnew = p.cstate;                       // where c_t goes (same place as c_{t-1} unless every frame is kept)
    float* __restrict__ gsave = nullptr;
    if (p.lstm_D > 0) {
      const long HN = (long)H * p.N;
      cst += (long)lstm_layer * HN;
      cnew = cst;
      Cb = p.lstm_out + ((long)lstm_layer * p.lstm_D + lstm_t % p.lstm_D) * HN;
      first = lstm_t == 0;
      if (p.gates_out) {
        cnew = p.cstate + ((long)lstm_layer * p.lstm_D + lstm_t) * HN;
        cst = cnew - HN;                            // c_{t-1}: the previous frame of the same layer (not read at t = 0)
        gsave = p.gates_out + ((long)lstm_layer * p.lstm_D + lstm_t) * 4 * HN;
      }
    }
    // Every load of the cell -- the biases, c_{t-1}, layer 0's input projection -- in front of the first store (round 6).  c_t overwrites c_{t-1} in
    // place at inference, so hipcc must keep each cell's load behind the previous cell's stores, and on gfx9 a load behind a store is waited for
    // together with it (one in-order vmcnt): the WM x NT cells of a lane ran one memory round trip apart, 16 in a row on the 128 x 128 tile.
    // A lane reads and writes only its own cells, so reading them all first is the same computation.
    float addv[WM][4], cprev[WM][NT], rproj[WM][NT][4];
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const int row0 = m0 + wave * WM * 16 + i * 16 + kq * 4;       // = 4 * unit
      const int u = row0 >> 2;
      const bool uok = u < H;
#pragma unroll
      for (int r = 0; r < 4; ++r) addv[i][r] = uok ? (bia ? bia[r * H + u] : 0.f) + (bib ? bib[r * H + u] : 0.f) : 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int gn = n0 + t * 16 + nq;
        const bool ok = uok && gn < p.N;
        cprev[i][t] = (ok && !first) ? cst[(long)u * p.N + gn] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) rproj[i][t][r] = (ok && Rb) ? Rb[(long)(row0 + r) * p.srm + gn] : 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const int row0 = m0 + wave * WM * 16 + i * 16 + kq * 4;       // = 4 * unit
      const int u = row0 >> 2;
      if (u >= H) continue;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int gn = n0 + t * 16 + nq;
        if (gn >= p.N) continue;
        float gte[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) gte[r] = (F16 ? acc[i][t][r] * us : acc[i][t][r]) + addv[i][r] + rproj[i][t][r];
#if SSV_LSTM_FAST_CELL
        // (tuning builds) v_exp_f32 / v_rcp_f32 forms, as the LayerNorm / gate kernels' sigmoid: ~8 instructions per gate instead of ~25 / ~40
        const float gi = __builtin_amdgcn_rcpf(1.f + __expf(-gte[0])), gf = __builtin_amdgcn_rcpf(1.f + __expf(-gte[1]));
        const float gg = 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * gte[2])), go = __builtin_amdgcn_rcpf(1.f + __expf(-gte[3]));
#else
        const float gi = 1.f / (1.f + expf(-gte[0])), gf = 1.f / (1.f + expf(-gte[1]));
        const float gg = tanhf(gte[2]), go = 1.f / (1.f + expf(-gte[3]));
#endif
        const long ci = (long)u * p.N + gn;
        const float cn = (first ? 0.f : gf * cprev[i][t]) + gi * gg;
#if SSV_LSTM_ABL & 1
        if (cn == 1234.5678f) cnew[ci] = cn;
#else
        cnew[ci] = cn;
#endif
        if (gsave) {
          const long HN = (long)H * p.N;
          gsave[ci] = gi; gsave[HN + ci] = gf; gsave[2 * HN + ci] = gg; gsave[3 * HN + ci] = go;
        }
#if SSV_LSTM_FAST_CELL
        const float hval = go * (1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * cn)));
#else
        const float hval = go * tanhf(cn);
#endif
        if (!XS || p.hs_keep_h) Cb[(long)u * p.scm + gn] = hval;
        if constexpr (XS) {
          // the same h, split as its consumers would split it (scale 2^14: |h| < 1), into the planes of (layer, slot t % 2)
          const float hs_ = hval * 16384.f;
          const _Float16 hh = (_Float16)hs_;
          const _Float16 hl = (_Float16)(hs_ - (float)hh);
          _Float16* ph = reinterpret_cast<_Float16*>(reinterpret_cast<char*>(p.hs_planes) + ((long)lstm_layer * 2 + (lstm_t & 1)) * 2 * p.hs_plane_bytes);
          const long at = ((long)(u >> 3) * p.hs_npad + gn) * 8 + (u & 7);
#if SSV_LSTM_ABL & 1
          if (hval == 1234.5678f) { ph[at] = hh; ph[at + p.hs_plane_bytes / 2] = hl; }
#else
          ph[at] = hh;
          ph[at + p.hs_plane_bytes / 2] = hl;
#endif
        }
      }
    }
    return;
  }
  if constexpr (KT == 1 && EPI == 0) if (p.row_pair) {
    // The transposed convolution's forward (ConvTranspose1d(k = 2, s = 2), GemmNNB::row_pair): rows 2 o and 2 o + 1 of the product are the even and
    // the odd columns of output row o.  A wave parks its 16 x BN block as the other epilogues do and reads it back PAIRWISE: a 16-byte vector of
    // output row o is two 8-byte pieces of the parked rows 2 o and 2 o + 1 -- 8 output rows of 2 BN contiguous floats per block, whole row pieces per
    // store instruction (the two-launch form wrote every other float: each 128-byte line in two half-filled passes).
    constexpr int LDW = BN + 4;
    float* stage = reinterpret_cast<float*>(lds_all) + wave * 16 * LDW;
    __shared__ float amx_rp[4];
    __syncthreads();                                            // every wave is done reading the last chunk's image
    float addv[WM][4];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gmc = min(m0 + wave * WM * 16 + i * 16 + kq * 4 + r, p.M - 1);
        addv[i][r] = p.bias ? p.bias[gmc >> 1] : 0.f;
      }
    float am = 0.f;
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const int rbase = m0 + wave * WM * 16 + i * 16;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) stage[(kq * 4 + r) * LDW + t * 16 + nq] = (F16 ? acc[i][t][r] * us : acc[i][t][r]) + addv[i][r];
#pragma unroll
      for (int it = 0; it < NT; ++it) {
        const int e = lane + 64 * it;                           // 16-byte vector index in the 8 x (2 BN) block of output rows
        const int orow = e / (BN / 2), c2 = e % (BN / 2);       // the vector covers parked columns 2 c2, 2 c2 + 1 of rows 2 orow, 2 orow + 1
        const float2 ev = *reinterpret_cast<const float2*>(stage + (2 * orow) * LDW + 2 * c2);
        const float2 od = *reinterpret_cast<const float2*>(stage + (2 * orow + 1) * LDW + 2 * c2);
        const int gm = rbase + 2 * orow, gn = n0 + 2 * c2;
        if (gm < p.M && gn < p.N) {
          float* dst = Cb + (long)(gm >> 1) * p.scm + 2 * gn;
          if (gn + 1 < p.N) {
            const f4u o = {ev.x, od.x, ev.y, od.y};
            *reinterpret_cast<f4u*>(dst) = o;
            am = fmaxf(fmaxf(am, fmaxf(fabsf(ev.x), fabsf(od.x))), fmaxf(fabsf(ev.y), fabsf(od.y)));
          } else {
            dst[0] = ev.x; dst[1] = od.x;
            am = fmaxf(am, fmaxf(fabsf(ev.x), fabsf(od.x)));
          }
        }
      }
    }
    if (p.c_amax) {
      am = ssv_wg_max<4>(am, amx_rp);
      if (tid == 0) {
        float* al = p.c_amax + (long)b * p.c_namax;
        al[bxx] = am;
        if (bxx == (int)gridDim.x - 1) for (int e = gridDim.x; e < p.c_namax; ++e) al[e] = 0.f;
      }
    }
    return;
  }
  if (p.scn == 1) {
    // Row-contiguous stores.  An MFMA accumulator holds 4 rows x 1 column per lane, so storing it directly writes 64-byte
    // pieces of 4 different rows per instruction (measured: the epilogue was 7.3 of 36.6 us at C = 256, L = 325).  The tile
    // goes through LDS instead (free after the K loop): every wave parks its 16 x BN block row-major and reads it back as
    // 16-byte vectors along the row -- a store instruction then covers up to 448 contiguous bytes of one or two rows.
    constexpr int LDW = BN + 4;                                 // row pitch in floats: 16-byte aligned, bank-conflict free
    float* stage = reinterpret_cast<float*>(lds_all) + wave * 16 * LDW;
    __syncthreads();                                            // every wave is done reading the last chunk's image
    // Every load of the epilogue sits in front of its first store (round 6): the bias terms of all the wave's rows and the residual's vectors (the data
    // gradient of a highway layer adds into dx, which the gate backward has seeded: R = C).  On gfx9 loads and stores share ONE in-order counter (vmcnt),
    // so a load issued behind a store can only be waited for together with that store: with the residual load inside the store loop every 16-byte
    // store waited for the acknowledgement of the one before it, and row block 1's bias loads for row block 0's stores (tools/isa_store_waits.py
    // lists such waits).  In-step, same box: the 64 x 96 data-gradient tiles 46.9 -> 44.9 and 50.4 -> 48.6 us, 128 x 112 at L = 650 73.2 -> 70.3.
    float addv[WM][4];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gmc = min(m0 + wave * WM * 16 + i * 16 + kq * 4 + r, p.M - 1);
        const int gb = p.perm_h ? (gmc & 3) * p.perm_h + (gmc >> 2) : gmc;
        float add = 0.f;
        if (p.bias) add += p.bias[gb];
        if (p.bias_b) add += p.bias_b[(long)b * p.sbb + gb];
        addv[i][r] = add;
      }
    f4u rres[WM][NT];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int it = 0; it < NT; ++it) rres[i][it] = (f4u){0.f, 0.f, 0.f, 0.f};
    if (Rb) {
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int it = 0; it < NT; ++it) {
          const int e = lane + 64 * it;
          const int gm = m0 + wave * WM * 16 + i * 16 + e / (BN / 4), gn = n0 + (e % (BN / 4)) * 4;
          if (gm < p.M && gn < p.N) {
            const float* src = Rb + (long)gm * p.srm + gn;
            if (gn + 3 < p.N) rres[i][it] = *reinterpret_cast<const f4u*>(src);
            else {
#pragma unroll
              for (int q = 0; q < 4; ++q) if (gn + q < p.N) rres[i][it][q] = src[q];
            }
          }
        }
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const int rbase = m0 + wave * WM * 16 + i * 16;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float add = addv[i][r];
#pragma unroll
        for (int t = 0; t < NT; ++t) stage[(kq * 4 + r) * LDW + t * 16 + nq] = (F16 ? acc[i][t][r] * us : acc[i][t][r]) + add;
      }
      // the block is private to the wave: no workgroup barrier, the LDS operations of one wave complete in order
#pragma unroll
      for (int it = 0; it < NT; ++it) {
        const int e = lane + 64 * it;                           // 16-byte vector index in the 16 x BN block
        const int row = e / (BN / 4), c4 = e % (BN / 4);
        const int gm = rbase + row, gn = n0 + c4 * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * LDW + c4 * 4);
        if (gm < p.M && gn < p.N) {
          float* dst = Cb + (long)gm * p.scm + gn;
          f4u o = {v[0], v[1], v[2], v[3]};
          o += rres[i][it];
          if (gn + 3 < p.N) *reinterpret_cast<f4u*>(dst) = o;
          else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (gn + q < p.N) dst[q] = o[q];
          }
        }
      }
      if (p.colstats) {
        // (after the row stores: they are asynchronous and the kernel cannot end before they drain, so they go first)
        // column statistics of this 16-row block: a lane reads ITS column of the parked block (consecutive lanes, consecutive
        // words: conflict-free), two passes over 16 values in registers
        float* cst = reinterpret_cast<float*>(lds_all) + 4 * 16 * LDW + (wave * WM + i) * BN * 2;
#pragma unroll
        for (int c0 = 0; c0 < BN; c0 += 64) {
          const int c = c0 + lane;
          if (c < BN) {
            float v[16], sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { v[r] = stage[r * LDW + c]; sum += v[r]; }
            const float mean = sum * (1.f / 16.f);
            float m2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float d = v[r] - mean; m2 += d * d; }
            cst[2 * c] = mean; cst[2 * c + 1] = m2;
          }
        }
      }
    }
    if (p.colstats) {
      // four 16-row blocks -> one 64-row group (Chan's merge of equal counts), WM groups per tile
      __syncthreads();
      const float* cst = reinterpret_cast<const float*>(lds_all) + 4 * 16 * LDW;
      for (int e = tid; e < WM * BN; e += 256) {
        const int grp = e / BN, c = e % BN, gn = n0 + c;
        if (gn >= p.N) continue;
        float mu[4], mean = 0.f, m2 = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) { mu[q] = cst[((grp * 4 + q) * BN + c) * 2]; mean += mu[q]; m2 += cst[((grp * 4 + q) * BN + c) * 2 + 1]; }
        mean *= 0.25f;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const float d = mu[q] - mean; m2 += 16.f * d * d; }
        float* dst = p.colstats + (((long)b * (p.M >> 6) + (m0 >> 6) + grp) * p.N + gn) * 2;
        dst[0] = mean; dst[1] = m2;
      }
    }
    NN_STAMP_AT(3);
    NN_RT(1);
    return;
  }
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gm = m0 + wave * WM * 16 + i * 16 + kq * 4 + r;
      if (gm >= p.M) continue;
      const int gb = p.perm_h ? (gm & 3) * p.perm_h + (gm >> 2) : gm;   // bias index in the caller's (torch) row order
      float add = 0.f;
      if (p.bias) add += p.bias[gb];
      if (p.bias_b) add += p.bias_b[(long)b * p.sbb + gb];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int gn = n0 + t * 16 + nq;
        if (gn >= p.N) continue;
        float v = (F16 ? acc[i][t][r] * us : acc[i][t][r]) + add;
        if (Rb) v += Rb[(long)gm * p.srm + gn];
        Cb[(long)gm * p.scm + (long)gn * p.scn] = v;
      }
    }
}

// ---- NN, wide workgroup ------------------------------------------------------------------------------------------------
// The 4-wave kernel above moves (A tile + X tile) bytes from L2 per chunk for only 128 x 112 outputs: at the bf16 MFMA rate
// that is ~34 B/clk/CU, and an ablation (MFMAs removed: 356 of 503 us remain; loads removed: 167 us) shows it is bound by
// L2->CU traffic, not by the matrix cores.  Here a workgroup has 4 x NWN waves (M x N): the NWN wave columns share one
// weight tile, so both operands are staged in LDS once per workgroup and the L2 bytes per MAC drop by ~2x (128 x 336 tile:
// 16 B/clk/CU at the full MFMA rate).  Same LDS image layout [k-group][row][8 x bf16] (conflict-free b128 reads), same
// "issue raw loads, mask at commit" staging and hoisted addressing as above.
// (two 8-wave workgroups per CU = 4 waves per SIMD need <= 128 VGPRs: the split-bf16 form of the 128 x 192 tile has 122, the split-fp16 one 134
// without the bound -- +42 % time; the second __launch_bounds__ argument is waves per SIMD in HIP)
template <int KT, int WM, int NT, int NWN, int F16, int XR = 0>
__global__ __launch_bounds__(256 * NWN, (NWN == 2 && KT == 1) ? 4 : 1) void gemm_nn_bf3w_kernel(const GemmNNB p, const int mtiles, const int smin, const int span) {
  constexpr int T = 256 * NWN;
  constexpr int BM = 64 * WM, BN = 16 * NT * NWN;
  constexpr int HALO = (KT == 1) ? 0 : 54;
  constexpr int WX = ((BN + HALO + 15) / 16) * 16;
  constexpr int A_SLOTS = KT * 4 * BM, X_SLOTS = 4 * WX;
  constexpr int NA = (A_SLOTS + T - 1) / T, NX = (X_SLOTS + T - 1) / T;
  // (the epilogue re-uses the staging memory to park every wave's 16 x (16 NT) blocks for row-contiguous stores, as gemm_nn_bf3_kernel does)
  constexpr int LDWP = 16 * NT + 4, PARK_U4 = SSV_NNBW_PARK ? (4 * NWN * 16 * LDWP) / 4 : 0;
  constexpr int STAGE_U4 = 2 * A_SLOTS + 2 * X_SLOTS;
  __shared__ uint4 lds[STAGE_U4 > PARK_U4 ? STAGE_U4 : PARK_U4];
  uint4* Ah = lds;
  uint4* Al = lds + A_SLOTS;
  uint4* Xh = lds + 2 * A_SLOTS;
  uint4* Xl = Xh + X_SLOTS;
  // XR: the launch covers rows 0 .. M - 2 with its tiles (p.M is the FULL row count); row M - 1 is added by the workgroups of row tile 0 as fp32
  // dot products of its weights (xw, staged once) with the raw input values every staging thread holds before it splits them
  constexpr int XW_MAX = XR ? 1056 : 1;
  __shared__ float xw[XW_MAX];
  __shared__ float xsum[XR ? 4 * WX : 1];
  static_assert(!XR || (KT == 1 && X_SLOTS <= 2 * T), "extra row: k = 1, at most two slots per thread");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 3, wn = wave >> 2;
  const unsigned wg = ssv_xcd_order(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);   // see ssv_xcd_order
  const int bxx = (int)(wg % gridDim.x), b = (int)(wg / gridDim.x);
  const int mt = bxx % mtiles, ntile = bxx / mtiles;
  const int m0 = mt * BM, n0 = ntile * BN;
  const float* __restrict__ Xb = p.X + (long)b * p.sxb;
  const int W = BN + span;
  const int nchunks = p.Kpad / 32;
  const int kq = lane >> 4, nq = lane & 15;
  const int Mt = XR ? p.M - 1 : p.M;                      // rows the tiles cover
  const bool xr_on = XR && mt == 0;
  float xacc[NX];
#pragma unroll
  for (int r = 0; r < NX; ++r) xacc[r] = 0.f;
  if constexpr (XR) {
    if (xr_on) for (int k = tid; k < nchunks * 32; k += T) xw[k] = k < p.Kc ? p.xrow_w[(long)k * p.xrow_sk] : 0.f;    // (visible after the loop's first barrier)
  }

  f32x4 acc[WM][NT];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  uint4 rah[NA], ral[NA];
  float rx[NX][8];

  // weight staging: slot f -> (tap j, row block, k-group kg, row%16) in the packed fragment order, so a wave reads 1 KB
  // of contiguous global memory per load
  const int MB = (p.M + 15) >> 4;
  auto a_off = [&](int r) -> long {
    const int f = min(tid + T * r, A_SLOTS - 1);
    const int r16 = f & 15, kg = (f >> 4) & 3, mbl = (f >> 6) % (BM / 16), j = f / (4 * BM);
    return (((long)j * MB + min((m0 >> 4) + mbl, MB - 1)) * nchunks) * 512 + (kg * 16 + r16) * 8;
  };
  auto a_slot = [&](int r) -> int {
    const int f = min(tid + T * r, A_SLOTS - 1);
    const int r16 = f & 15, kg = (f >> 4) & 3, mbl = (f >> 6) % (BM / 16), j = f / (4 * BM);
    return (j * 4 + kg) * BM + mbl * 16 + r16;
  };
  const int Lrow = (int)p.sxc;
  unsigned voff[NX];
  unsigned cvmask = 0;
#pragma unroll
  for (int r = 0; r < NX; ++r) {
    const int e = tid + T * r;
    const int kg = e / WX, col = e % WX;
    const int gcol = n0 + smin + col;
    if (e < X_SLOTS && col < W && gcol >= 0 && gcol < p.Lx) cvmask |= 1u << r;
    voff[r] = (unsigned)((e < X_SLOTS ? 8 * kg : 0) * Lrow + min(max(gcol, 0), p.Lx - 1));
  }
  const bool ragged = (p.Kc & 31) != 0;
  float xs = 1.f, us = 1.f;           // split-fp16 scales, see gemm_nn_bf3_kernel

  // buffer loads (see ssv_buf): per-thread byte offsets fixed for the kernel, the chunk / row offset a scalar operand
  const __amdgpu_buffer_rsrc_t rsAh = ssv_buf(p.Ahi), rsAl = ssv_buf(p.Alo), rsX = ssv_buf(Xb);
  unsigned aoffb[NA];
#pragma unroll
  for (int r = 0; r < NA; ++r) aoffb[r] = (unsigned)(a_off(r) * 2);
  auto prefetch = [&](int ch) {
#pragma unroll
    for (int r = 0; r < NA; ++r) {
      rah[r] = ssv_buf_u4(rsAh, aoffb[r], (unsigned)__builtin_amdgcn_readfirstlane(ch * 1024));
      ral[r] = ssv_buf_u4(rsAl, aoffb[r], (unsigned)__builtin_amdgcn_readfirstlane(ch * 1024));
    }
    if (!ragged || ch + 1 < nchunks) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const unsigned so = (unsigned)((ch * 32 + i) * Lrow) * 4u;                                     // wave-uniform row offset
#pragma unroll
        for (int r = 0; r < NX; ++r) rx[r][i] = ssv_buf_f32(rsX, voff[r] * 4u, so);
      }
    } else {
#pragma unroll
      for (int r = 0; r < NX; ++r) {
        const int e = tid + T * r;
        const int kg = (e < X_SLOTS) ? e / WX : 0;
        const unsigned colo = voff[r] - (unsigned)(8 * kg * Lrow);
#pragma unroll
        for (int i = 0; i < 8; ++i) rx[r][i] = ssv_buf_f32(rsX, ((unsigned)min(ch * 32 + 8 * kg + i, p.Kc - 1) * (unsigned)Lrow + colo) * 4u, 0u);
      }
    }
  };
  auto commit = [&](int ch) {
#pragma unroll
    for (int r = 0; r < NA; ++r)
      if (tid + T * r < A_SLOTS) { const int sl = a_slot(r); Ah[sl] = rah[r]; Al[sl] = ral[r]; }
    const bool last_ragged = ragged && ch + 1 == nchunks;
#pragma unroll
    for (int r = 0; r < NX; ++r) {
      const int e = tid + T * r;
      if (e < X_SLOTS) {
        float v[8];
        if (!last_ragged) {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = ((cvmask >> r) & 1) ? rx[r][i] : 0.f;
        } else {
          const int kg = e / WX;
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = (((cvmask >> r) & 1) && ch * 32 + 8 * kg + i < p.Kc) ? rx[r][i] : 0.f;
        }
        if constexpr (XR) {
          if (xr_on) {
            const float* wq = xw + ch * 32 + 8 * (e / WX);
#pragma unroll
            for (int i = 0; i < 8; ++i) xacc[r] = __builtin_fmaf(wq[i], v[i], xacc[r]);
          }
        }
        uint4 h, l;
        split8s<F16>(v, xs, h, l);
        Xh[e] = h; Xl[e] = l;
      }
    }
  };

  int offj[KT];
#pragma unroll
  for (int j = 0; j < KT; ++j) offj[j] = p.shift[j] - smin;
  const int arow = kq * BM + wm * WM * 16 + nq;
  const int xcol = kq * WX + wn * NT * 16 + nq;

  prefetch(0);
  if constexpr (F16) {
    float sc, inv;
    ssv_pow2_scale(ssv_wave_list_max(p.x_amax + (long)b * p.x_amax_bs, p.x_namax), sc, inv);
    xs = ssv_uniform(sc);
    us = ssv_uniform(inv * *p.a_inv);
  }
  for (int ch = 0; ch < nchunks; ++ch) {
    __syncthreads();
    commit(ch);
    __syncthreads();
    if (ch + 1 < nchunks) prefetch(ch + 1);
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      uint4 ah[WM], al[WM];
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        ah[i] = Ah[j * 4 * BM + arow + i * 16];
        al[i] = Al[j * 4 * BM + arow + i * 16];
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int xs_ = xcol + t * 16 + offj[j];
        const uint4 bh = Xh[xs_];
        const uint4 bl = Xl[xs_];
#pragma unroll
        for (int i = 0; i < WM; ++i) {
          acc[i][t] = mma16<F16>(al[i], bh, acc[i][t]);
          acc[i][t] = mma16<F16>(ah[i], bl, acc[i][t]);
          acc[i][t] = mma16<F16>(ah[i], bh, acc[i][t]);
        }
      }
    }
  }

  float* __restrict__ Cb = p.C + (long)b * p.scb;
  const float* __restrict__ Rb = p.R ? p.R + (long)b * p.srb : nullptr;
  if constexpr (SSV_NNBW_PARK) {
    if (p.scn == 1) {
      __syncthreads();                                            // every wave is done reading the last chunk's image
      float* pk = reinterpret_cast<float*>(lds) + wave * 16 * LDWP;
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const int rbase = m0 + wm * WM * 16 + i * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int gmc = min(rbase + kq * 4 + r, Mt - 1);
          float add = 0.f;
          if (p.bias) add += p.bias[gmc];
          if (p.bias_b) add += p.bias_b[(long)b * p.sbb + gmc];
#pragma unroll
          for (int t = 0; t < NT; ++t) pk[(kq * 4 + r) * LDWP + t * 16 + nq] = (F16 ? acc[i][t][r] * us : acc[i][t][r]) + add;
        }
        // (the block is private to the wave: its LDS operations complete in order)
#pragma unroll
        for (int it = 0; it < NT; ++it) {
          const int e = lane + 64 * it;
          const int row = e / (4 * NT), c4 = e % (4 * NT);
          const int gm = rbase + row, gn = n0 + wn * NT * 16 + c4 * 4;
          const f32x4 v = *reinterpret_cast<const f32x4*>(pk + row * LDWP + c4 * 4);
          if (gm < Mt && gn < p.N) {
            float* dst = Cb + (long)gm * p.scm + gn;
            if (gn + 3 < p.N) {
              f4u o = {v[0], v[1], v[2], v[3]};
              if (Rb) { const f4u rr = *reinterpret_cast<const f4u*>(Rb + (long)gm * p.srm + gn); o += rr; }
              *reinterpret_cast<f4u*>(dst) = o;
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j) if (gn + j < p.N) dst[j] = v[j] + (Rb ? Rb[(long)gm * p.srm + gn + j] : 0.f);
            }
          }
        }
      }
    }
  }
  if (!SSV_NNBW_PARK || p.scn != 1) {
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gm = m0 + wm * WM * 16 + i * 16 + kq * 4 + r;
      if (gm >= Mt) continue;
      float add = 0.f;
      if (p.bias) add += p.bias[gm];
      if (p.bias_b) add += p.bias_b[(long)b * p.sbb + gm];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int gn = n0 + wn * NT * 16 + t * 16 + nq;
        if (gn >= p.N) continue;
        float v = (F16 ? acc[i][t][r] * us : acc[i][t][r]) + add;
        if (Rb) v += Rb[(long)gm * p.srm + gn];
        Cb[(long)gm * p.scm + gn] = v;
      }
    }
  }
  if constexpr (XR) {
    if (xr_on) {                                            // (workgroup-uniform) row M - 1: the four k-groups' partial sums of a column, then bias and residual
#pragma unroll
      for (int r = 0; r < NX; ++r) { const int e = tid + T * r; if (e < X_SLOTS) xsum[e] = xacc[r]; }
      __syncthreads();
      for (int c = tid; c < WX; c += T) {
        const int gn = n0 + c, gm = p.M - 1;
        if (c < BN && gn < p.N) {
          float v = (xsum[c] + xsum[WX + c]) + (xsum[2 * WX + c] + xsum[3 * WX + c]);
          if (p.bias) v += p.bias[gm];
          if (p.bias_b) v += p.bias_b[(long)b * p.sbb + gm];
          if (Rb) v += Rb[(long)gm * p.srm + gn];
          Cb[(long)gm * p.scm + gn] = v;
        }
      }
    }
  }
}

// rows 0 .. M - 2 in tiles, row M - 1 beside the staging (GemmNNB::xrow_w)
static int launch_nnbw_xrow(const GemmNNB& g, hipStream_t st, int smin, int span) {
  const int mtiles = ssv_cdiv(g.M - 1, 128), ntiles = ssv_cdiv(g.N, 192);
  if (ssv_shape_log_on()) {
    char nm[96], note[96];
    snprintf(nm, sizeof nm, "gemm_nn_bf3w_kernel<1, 2, 6, 2, %d, 1>", g.f16);
    snprintf(note, sizeof note, "B=%d M=%d N=%d K=%d k=1 (last row beside the tiles)", g.B, g.M, g.N, g.Kc);
    ssv_shape_log(nm, dim3(mtiles * ntiles, g.B), dim3(512), 2.0 * g.B * g.M * g.N * g.Kc,
                  4.0 * ((double)g.B * g.Kc * g.N + (double)g.B * g.M * g.N + (double)g.M * g.Kc), note);
  }
  if (g.f16) hipLaunchKernelGGL((gemm_nn_bf3w_kernel<1, 2, 6, 2, 1, 1>), dim3(mtiles * ntiles, g.B), dim3(512), 0, st, g, mtiles, smin, span);
  else hipLaunchKernelGGL((gemm_nn_bf3w_kernel<1, 2, 6, 2, 0, 1>), dim3(mtiles * ntiles, g.B), dim3(512), 0, st, g, mtiles, smin, span);
  return ssv_check_launch("gemm_nn_bf3w (extra row)");
}
template <int KT, int WM, int NT, int NWN>
static int launch_nnbw(const GemmNNB& g, hipStream_t st, int smin, int span) {
  const int mtiles = ssv_cdiv(g.M, 64 * WM), ntiles = ssv_cdiv(g.N, 16 * NT * NWN);
  if (ssv_shape_log_on()) {
    char nm[96], note[96];
    snprintf(nm, sizeof nm, "gemm_nn_bf3w_kernel<%d, %d, %d, %d, %d>", KT, WM, NT, NWN, g.f16);
    snprintf(note, sizeof note, "B=%d M=%d N=%d K=%d k=%d", g.B, g.M, g.N, g.Kc, KT);
    ssv_shape_log(nm, dim3(mtiles * ntiles, g.B), dim3(256 * NWN), 2.0 * g.B * g.M * g.N * g.Kc * KT,
                  4.0 * ((double)g.B * g.Kc * g.N + (double)g.B * g.M * g.N + (double)g.M * g.Kc * KT), note);
  }
  if (g.f16) hipLaunchKernelGGL((gemm_nn_bf3w_kernel<KT, WM, NT, NWN, 1>), dim3(mtiles * ntiles, g.B), dim3(256 * NWN), 0, st, g, mtiles, smin, span);
  else hipLaunchKernelGGL((gemm_nn_bf3w_kernel<KT, WM, NT, NWN, 0>), dim3(mtiles * ntiles, g.B), dim3(256 * NWN), 0, st, g, mtiles, smin, span);
  return ssv_check_launch("gemm_nn_bf3w");
}

template <int KT, int WM, int NT>
static int launch_nnb(const GemmNNB& g, hipStream_t st, int smin, int span) {
  const int mtiles = ssv_cdiv(g.M, 64 * WM), ntiles = ssv_cdiv(g.N, 16 * NT);
  SSV_CHECK(!g.c_amax || g.c_namax >= mtiles * ntiles, SSV_UNSUPPORTED, "gemm_nn_bf3: %d tiles per item, the output's scale list has %d entries", mtiles * ntiles, g.c_namax);
  if constexpr (KT == 1) {
    if (g.epi == 1) {
      if (g.f16) hipLaunchKernelGGL((gemm_nn_bf3_kernel<KT, WM, NT, 1, 1>), dim3(mtiles * ntiles, g.B), dim3(256), 0, st, g, mtiles, smin, span);
      else hipLaunchKernelGGL((gemm_nn_bf3_kernel<KT, WM, NT, 1, 0>), dim3(mtiles * ntiles, g.B), dim3(256), 0, st, g, mtiles, smin, span);
      return ssv_check_launch("gemm_nn_bf3_lstm");
    }
  }
  if (ssv_shape_log_on()) {
    char nm[96], note[96];
    snprintf(nm, sizeof nm, "gemm_nn_bf3_kernel<%d, %d, %d, 0, %d, %d>", KT, WM, NT, g.f16, (KT == 3 && span <= SSV_NN_HALO_SMALL) ? SSV_NN_HALO_SMALL : 54);
    snprintf(note, sizeof note, "B=%d M=%d N=%d K=%d k=%d%s", g.B, g.M, g.N, g.Kc, KT, g.colstats ? " +colstats" : "");
    ssv_shape_log(nm, dim3(mtiles * ntiles, g.B), dim3(256), 2.0 * g.B * g.M * g.N * g.Kc * KT,
                  4.0 * ((double)g.B * g.Kc * g.N + (double)g.B * g.M * g.N * (g.R ? 2 : 1) + (double)g.M * g.Kc * KT), note);
  }
  if constexpr (KT == 3) {
    if (span <= SSV_NN_HALO_SMALL) {
      if (g.f16) hipLaunchKernelGGL((gemm_nn_bf3_kernel<KT, WM, NT, 0, 1, SSV_NN_HALO_SMALL>), dim3(mtiles * ntiles, g.B), dim3(256), 0, st, g, mtiles, smin, span);
      else hipLaunchKernelGGL((gemm_nn_bf3_kernel<KT, WM, NT, 0, 0, SSV_NN_HALO_SMALL>), dim3(mtiles * ntiles, g.B), dim3(256), 0, st, g, mtiles, smin, span);
      return ssv_check_launch("gemm_nn_bf3");
    }
  }
  if (g.f16) hipLaunchKernelGGL((gemm_nn_bf3_kernel<KT, WM, NT, 0, 1>), dim3(mtiles * ntiles, g.B), dim3(256), 0, st, g, mtiles, smin, span);
  else hipLaunchKernelGGL((gemm_nn_bf3_kernel<KT, WM, NT, 0, 0>), dim3(mtiles * ntiles, g.B), dim3(256), 0, st, g, mtiles, smin, span);
  return ssv_check_launch("gemm_nn_bf3");
}

template <int KT>
static int pick_nnb(const GemmNNB& g, hipStream_t st, int smin, int span) {
  {
    // measured (round-1 tile sweep; in-step re-check: tools/sweep_force.sh): the wide workgroup wins for kernel-size-1 convolutions over long sequences
    // (SSRN's 513-channel layers: 150 -> 205 TFLOP/s); the k=3 layers are as fast or faster on the 4-wave kernel.
    // 128 x 192 tiles (8 waves) are the faster wide shape (513 -> 512 channels: 104 -> 87 us, 256 -> 512: 60 -> 49 us) except
    // when M leaves a nearly empty last row tile (M = 513), where the 16-wave 128 x 448 tile loses less (round-1 sweep)
    // ... and only when the 128 x 192 tiling still gives every CU a workgroup: a single long utterance (the vocoder's DFT
    // at B = 1: 1026 x 1300 x 1024) is 27-56 wide tiles, a fifth of the chip; the cost model below then picks small tiles.
    if (KT == 1 && !g.epi && !g.perm_h && !g.colstats && !g.row_pair && g.sxn == 1 && g.scn == 1 && g.N >= 1024 && g.M >= 256 && g.Kc >= 256 &&
        (long)ssv_cdiv(g.M, 128) * ssv_cdiv(g.N, 192) * g.B >= 256)
    {
      // M = 128 j + 1 (SSRN's 513 channels): the last row beside the tiles of the 128 x 192 kernel (120.6 -> see DESIGN 4.6) instead of a fifth row tile
      if (SSV_NNBW_XROW && g.xrow_w && g.M % 128 == 1 && g.Kpad <= 1056 && g.scn == 1 && g.sxn == 1) return launch_nnbw_xrow(g, st, smin, span);
      return (g.M % 128 == 0) ? launch_nnbw<KT, 2, 6, 2>(g, st, smin, span) : launch_nnbw<KT, 2, 7, 4>(g, st, smin, span);
    }
  }
  static const int nts[] = {7, 6, 4, 2};
  // (round 3: 64 x 176 and 64 x 192 tiles -- 22 % less L2 -> CU operand traffic per launch at C = 256, L = 325 -- were forced per
  //  shape inside the step with SSV_NNB_FORCE, tools/sweep_force.sh: every one of ten shapes +0.03..+0.2 ms; instantiations removed)
  int wm = 2, nt = 7;
  bool forced = false;
  // LSTM wavefront with two or more layers in one launch (the steady state of the GE2E embedder: 2 x 3072 x 880 x 1536): the
  // cost model below picks 64 x 112 tiles; measured over the 122 steps of config 5, 128 x 64 tiles are 6 % faster
  // (13.8 -> 13.0 ms; 128 x 96: 13.5, 128 x 112: 14.2, 64 x 96: 14.8) as long as they still give every CU two workgroups.
  // (round 5, layer 0 riding along: three layers = 3 x 24 x 7 = 504 tiles of 128 x 128 are ONE round of two workgroups per CU, with half the weight
  //  bytes per MFMA of the 64-column tile -- the launch is bound by L2 -> CU traffic, 69 GB/s per CU measured on 1008 tiles of 128 x 64)
  if constexpr (KT == 1) {
    if (g.ksplit > 1) {                     // split reduction (the LSTM backward's merged data-gradient product): the forward wavefront's 128 x 128 tile
      const int mtiles = ssv_cdiv(g.M, 128), ntiles = ssv_cdiv(g.N, 128);
      SSV_CHECK((long)g.B * g.ksplit <= 65535, SSV_UNSUPPORTED, "gemm_nn_bf3: batch %d x %d K ranges exceeds grid.y", g.B, g.ksplit);
      if (g.f16) hipLaunchKernelGGL((gemm_nn_bf3_kernel<1, 2, 8, 0, 1>), dim3(mtiles * ntiles, g.B * g.ksplit), dim3(256), 0, st, g, mtiles, smin, span);
      else hipLaunchKernelGGL((gemm_nn_bf3_kernel<1, 2, 8, 0, 0>), dim3(mtiles * ntiles, g.B * g.ksplit), dim3(256), 0, st, g, mtiles, smin, span);
      return ssv_check_launch("gemm_nn_bf3 (split reduction)");
    }
    if (g.hs_planes) {                      // pre-split recurrent activations: the 128 x 128 tile for every step of the wavefront (also its first and last, partly filled ones)
      const int mtiles = ssv_cdiv(g.M, 128), ntiles = ssv_cdiv(g.N, 128);
      hipLaunchKernelGGL((gemm_nn_bf3_kernel<1, 2, 8, 1, 1, 3>), dim3(mtiles * ntiles, g.B), dim3(256), 0, st, g, mtiles, smin, span);
      return ssv_check_launch("gemm_nn_bf3_lstm (pre-split h)");
    }
    if (!forced && g.epi && g.lstm_D > 0 && g.B >= 2) {
      const long t128 = (long)ssv_cdiv(g.M, 128) * ssv_cdiv(g.N, 128) * g.B;
      const char* e = ssv_tuning(SSV_T_LSTM_MERGE);
      if (t128 >= 448 && t128 <= 512 && !(e && atoi(e) == 2)) {
        const int mtiles = ssv_cdiv(g.M, 128), ntiles = ssv_cdiv(g.N, 128);
        if (g.f16) hipLaunchKernelGGL((gemm_nn_bf3_kernel<1, 2, 8, 1, 1>), dim3(mtiles * ntiles, g.B), dim3(256), 0, st, g, mtiles, smin, span);
        else hipLaunchKernelGGL((gemm_nn_bf3_kernel<1, 2, 8, 1, 0>), dim3(mtiles * ntiles, g.B), dim3(256), 0, st, g, mtiles, smin, span);
        return ssv_check_launch("gemm_nn_bf3_lstm");
      }
    }
  }
  if (!forced && g.epi && g.lstm_D > 0 && g.B >= 2 && (long)ssv_cdiv(g.M, 128) * ssv_cdiv(g.N, 64) * g.B >= 512) { wm = 2; nt = 4; forced = true; }
  if (!forced) {
    // tuning aid (tools/sweep_force.sh): SSV_NNB_FORCE="kt:M:N=a,c;kt:M:N=a,c;..." forces the tile of one problem shape
    // inside a whole training step, where a tile's effect on its neighbours shows (isolated timings miss it)
    if (const char* e = ssv_tuning(SSV_T_NNB_FORCE)) {
      char key[48];
      snprintf(key, sizeof key, "%d:%d:%d=", KT, g.M, g.N);
      const char* hit = strstr(e, key);
      int a = 0, c = 0;
      if (hit && (hit == e || hit[-1] == ';') && sscanf(hit + strlen(key), "%d,%d", &a, &c) == 2 && (a == 1 || a == 2))
        for (int x : nts) if (x == c) { wm = a; nt = c; forced = true; }
    }
  }
  if (!forced) {
    double best = 1e30;
    // k=1 products carry a third of the MFMAs per weight byte: 64-row tiles (twice the weight traffic per MAC) lose at every
    // conv shape measured (256 -> 256, L=650: 28 us on 64 x 112 tiles, 20 us on 128 x 64); only the single-"batch" LSTM
    // product, short of workgroups, still wants them
    const int a_min = (KT == 1 && g.B > 1 && !g.epi && g.M > 64) ? 2 : 1;
    for (int a = a_min; a <= 2; ++a)
      for (int c : nts) {
        const long tiles = (long)ssv_cdiv(g.M, 64 * a) * ssv_cdiv(g.N, 16 * c) * g.B;
        const double per_tile = (double)a * c + 0.9 * a + 0.25 * c + 1.0;
        // The most loaded CU runs n workgroups.  For kernel-size-1 products a K chunk carries a third of the MFMAs per
        // staged byte, so co-resident workgroups are needed to cover each other's staging (measured: the LSTM step
        // 3072 x 880 x 768 runs 2x faster on 672 small tiles than on 192 large ones); k=3 chunks are long enough.
        const long n = (tiles + 255) / 256;
        const double overlap = (KT == 1) ? (n >= 3 ? 1.8 : (n == 2 ? 1.5 : 1.0)) : 1.0;
        // a single workgroup per CU (4 waves) cannot cover its own load latencies: measured +10 % on the data-gradient
        // shapes that fit in 256 large tiles (C=256, L=325: 45 -> 41 us on 768 tiles of 64 x 64)
        const double lonely = (tiles <= 256) ? 1.35 : 1.0;
        const double cost = (double)n * per_tile * lonely / overlap;
        if (cost < best) { best = cost; wm = a; nt = c; }
      }
  }
#define SSV_CASE(A_, C_) if (wm == A_ && nt == C_) return launch_nnb<KT, A_, C_>(g, st, smin, span)
  SSV_CASE(2, 7); SSV_CASE(2, 6); SSV_CASE(2, 4); SSV_CASE(2, 2);
  SSV_CASE(1, 7); SSV_CASE(1, 6); SSV_CASE(1, 4); SSV_CASE(1, 2);
#undef SSV_CASE
  return ssv_fail(SSV_UNSUPPORTED, "gemm_nn_bf3: no tile %d,%d", wm, nt);
}

int ssv_launch_gemm_nn_bf3(const GemmNNB& g, hipStream_t st) {
  SSV_CHECK(g.M > 0 && g.N > 0 && g.Kc > 0 && g.B > 0 && g.Kpad % 32 == 0 && g.Kpad >= g.Kc, SSV_BAD_SHAPE, "gemm_nn_bf3: bad problem");
  SSV_CHECK(g.KT == 1 || g.KT == 3, SSV_UNSUPPORTED, "gemm_nn_bf3: kernel_size %d", g.KT);
  SSV_CHECK(g.B <= 65535, SSV_UNSUPPORTED, "gemm_nn_bf3: batch %d exceeds grid.y", g.B);
  SSV_CHECK(g.sxn >= 1 && g.scn >= 1 && (g.scn == 1 || (!g.R && !g.epi)), SSV_BAD_SHAPE, "gemm_nn_bf3: bad column strides");
  SSV_CHECK(!g.epi || (g.KT == 1 && g.perm_h > 0 && g.M == 4 * g.perm_h && g.cstate && (g.B == 1 || g.lstm_D > 0) && g.Kc % 32 == 0), SSV_BAD_SHAPE,
            "gemm_nn_bf3: bad LSTM epilogue request");
  SSV_CHECK(g.lstm_D == 0 || (g.epi == 1 && g.lstm_out && g.lstm_D >= 1 && g.xsplit >= 0 && g.xsplit <= g.Kpad / 32 && g.Kc == g.Kpad && g.sxn == 1), SSV_BAD_SHAPE,
            "gemm_nn_bf3: bad LSTM wavefront request");
  int smin = g.shift[0], smax = g.shift[0];
  for (int j = 1; j < g.KT; ++j) { smin = g.shift[j] < smin ? g.shift[j] : smin; smax = g.shift[j] > smax ? g.shift[j] : smax; }
  const int span = smax - smin;
  SSV_CHECK(span <= 54, SSV_UNSUPPORTED, "gemm_nn_bf3: dilation halo %d exceeds 54", span);
  // the kernels address one batch item's input rows and the weight planes with 32-bit byte offsets (buffer loads)
  SSV_CHECK(((long)g.Kpad * g.sxc + (long)g.Lx * (g.sxn > 0 ? g.sxn : 1)) * 4 < (1L << 31) && (long)g.KT * ((g.M + 15) / 16) * (g.Kpad / 32) * 1024 < (1L << 31),
            SSV_UNSUPPORTED, "gemm_nn_bf3: a batch item's input or the weight planes span 2 GiB or more");
  SSV_CHECK(!g.f16 || (g.a_inv && ((g.x_amax && g.x_namax > 0) || (g.epi && g.x_namax == 0))), SSV_BAD_SHAPE, "gemm_nn_bf3: split-fp16 needs operand scales");
  SSV_CHECK(!g.colstats || (g.M % 64 == 0 && g.scn == 1 && !g.epi && !g.perm_h && !g.R), SSV_BAD_SHAPE, "gemm_nn_bf3: column statistics need M %% 64 == 0 and a plain epilogue");
  SSV_CHECK(!g.row_pair || (g.KT == 1 && g.M % 2 == 0 && g.scn == 1 && !g.R && !g.epi && !g.perm_h && !g.colstats && !g.bias_b), SSV_BAD_SHAPE,
            "gemm_nn_bf3: paired output rows need a plain k = 1 product over an even number of rows");
  SSV_CHECK(g.row_pair || !g.c_amax, SSV_BAD_SHAPE, "gemm_nn_bf3: the output's scale list comes with paired output rows only");
  SSV_CHECK(!g.hs_planes || (g.epi == 1 && g.lstm_D >= 1 && g.f16 && g.x_namax == 0 && g.perm_h % 32 == 0 && g.hs_npad % 128 == 0 && g.hs_npad >= g.N &&
                             g.hs_plane_bytes == (long)(g.perm_h / 8) * g.hs_npad * 16 && (long)(g.perm_h / 8) * g.hs_npad * 16 < (1L << 31)), SSV_BAD_SHAPE,
            "gemm_nn_bf3: pre-split recurrent activations need the merged split-fp16 inference wavefront and whole column tiles of planes");
  SSV_CHECK(!g.x0_planes || (g.hs_planes && g.A0hi && g.x0_amax && g.xsplit0 > 0 && g.xsplit0 % 2 == 0 && !g.R), SSV_BAD_SHAPE,
            "gemm_nn_bf3: pre-split input frames need the pre-split wavefront with layer 0 riding along and an even number of input chunks");
  SSV_CHECK(g.ksplit == 1 || (g.ksplit > 1 && g.KT == 1 && !g.epi && !g.R && !g.bias && !g.bias_b && !g.colstats && !g.row_pair && !g.perm_h && g.sxn == 1 && g.scn == 1 &&
                              g.Kc % 32 == 0 && g.Kpad == g.ksplit * g.Kc && g.skip_rows >= 0), SSV_BAD_SHAPE,
            "gemm_nn_bf3: a split reduction needs a plain k = 1 product whose planes hold ksplit ranges of Kc (a multiple of 32) rows");
  return g.KT == 3 ? pick_nnb<3>(g, st, smin, span) : pick_nnb<1>(g, st, smin, span);
}

