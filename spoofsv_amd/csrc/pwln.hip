// 1x1 conv + LayerNorm (+ activation) in one launch, forward (gemm_pwln_kernel) and backward (pwln_bwd_kernel), on split MFMAs.
#include "bf3_common.h"

// ---- 1x1 conv + LayerNorm over channels (+ activation) in ONE launch (round 4) ------------------------------------------------
// y = act(LN(W x + bias [+ s]))  -- models/TTSModel.py:128-131, :173-180, :218-231, :343-361.  Until round 4 every such link was two
// launches (the k = 1 GEMM, then the LayerNorm kernel re-reading its output).  A 1x1 convolution and a channel LayerNorm are both
// per-column operations, so a workgroup that owns ALL output rows of a column tile can finish the LayerNorm from its accumulators:
// no cross-workgroup step, no second pass over `pre`.  8 waves split the M axis (wave w: row blocks w * WMB .. w * WMB + WMB - 1, so
// BM = 128 * WMB >= M), every wave all 16 * NT columns; weight fragments go L2 -> registers (private rows per wave, two sets), the
// input tile is staged in LDS once per 32-channel chunk (two images, one barrier per chunk) exactly as in gemm_nn_bf3_kernel.  The
// price is the weight stream: every workgroup reads all of W (the row-tiled kernels re-read X instead).
// Epilogue: pre = acc * us + bias (+ s[b]) is stored; column sums of a lane's rows -> the 4 row-quads of the wave (cross-row shuffles)
// -> the 8 waves (LDS), mean, then the same for the squared deviations (a true two-pass variance, as ln_act_fwd_kernel); y = act(n).
// Tuning builds only (-DSSV_PW_STAMP): thread 0 of every workgroup (the first 1024) records s_memrealtime at entry and exit and the shader
// clock at six points (entry | first chunk staged | chunk loop done | pre stored + column sums | variance | y stored); ssv_debug_pw_stamps().
#ifdef SSV_PW_STAMP
__device__ unsigned long long ssv_pw_stamps[1024 * 8];
extern "C" int ssv_debug_pw_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ssv_pw_stamps), sizeof(ssv_pw_stamps)); }
#define PW_STAMP(k) do { const unsigned w_ = blockIdx.y * gridDim.x + blockIdx.x; if (threadIdx.x == 0 && w_ < 1024u) \
    ssv_pw_stamps[w_ * 8 + (k)] = ((k) >= 6) ? __builtin_amdgcn_s_memrealtime() : __builtin_readcyclecounter(); } while (0)
#else
#define PW_STAMP(k) do {} while (0)
#endif
struct PwLn {
  GemmNNB g;                      // A planes, X, C = pre (B, M, N), bias, bias_b, f16 scales
  const float* gamma; const float* beta;
  float* y; long ybs; float* stats; float* y_amax; int namax; int act;
};
template <int WMB, int NT, int F16, int XR = 0>
__global__ __launch_bounds__(512, 2) void gemm_pwln_kernel(const PwLn q) {
  const GemmNNB& p = q.g;
  constexpr int BN = 16 * NT;
  // K64 (round 6, the 4-row-block instantiations): a staged chunk is 64 channels -- two 32-channel sub-chunks of MFMAs per barrier, ALL 512 threads
  // stage one slot each (with 32-channel chunks only waves 0-3 staged while waves 4-7 waited at the barrier), and the staging of chunk c + 1 runs
  // in pieces behind the row blocks' MFMAs of chunk c (tools/probe/ldsdma_ring.hip, profiles/round6_ldsdma_ring_probe.txt: 1.12 -> 0.91 us per
  // 32 channels in the probe's model of this loop; the weight stream itself is not the limit: 0.74 us with nothing else in the loop)
  constexpr bool K64 = SSV_PWLN_BK64 && WMB == 4 && NT == 4;
  constexpr int KG = K64 ? 8 : 4;                      // k-groups (8 channels each) of a staged chunk
  constexpr int X_SLOTS = KG * BN;                     // 16-byte slots of one chunk: [k-group][column]
  static_assert(X_SLOTS <= 512, "one slot per thread");
  constexpr int IMG = 2 * X_SLOTS;
  __shared__ uint4 lds[2][IMG];
  __shared__ float red[8][BN];
  __shared__ float colv[2][BN];
  __shared__ float amx[8];
  // XR: row M - 1 (M = 128 j + 1: the 513-channel layers) is kept out of the MFMA row blocks -- a fifth row block per wave for ONE row otherwise
  // -- and comes from fp32 dot products of its weights (xw, staged once) with the raw values the staging threads hold before they split them
  __shared__ float xw[XR ? 1056 : 1];
  __shared__ float xsum[XR ? KG * BN : 1];
  __shared__ float xrow[XR ? 2 * BN : 1];                // the row's pre-activation per column, then its normalised value
  // epilogue (round 5): every wave parks a 16-row block row-major and reads it back as 16-byte vectors along the rows, so a store instruction
  // covers four whole 256-byte row pieces instead of 64-byte pieces of 16 rows (what gemm_nn_bf3_kernel's epilogue has done since round 2)
  constexpr int LDWP = BN + 4;
  __shared__ float park[SSV_PWLN_PARK ? 8 * 16 * LDWP : 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned wg = ssv_xcd_order(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  const int ntile = (int)(wg % gridDim.x), b = (int)(wg / gridDim.x);
  const int n0 = ntile * BN;
  const int kq = lane >> 4, nq = lane & 15;
  const int nchunks = p.Kpad / 32;
  const float* __restrict__ Xb = p.X + (long)b * p.sxb;
  PW_STAMP(6); PW_STAMP(0);

  f32x4 acc[WMB][NT];
#pragma unroll
  for (int i = 0; i < WMB; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // weight fragments: two register sets (chunk c + 1 loaded while chunk c multiplies) while they fit; from 4 row blocks per wave on, one set,
  // re-loaded right after the chunk's MFMAs have been issued (accumulators + two sets would not fit the 256 registers of an 8-wave workgroup)
  constexpr int NSET = WMB >= 4 ? 1 : 2;
  uint4 Ah_[NSET][WMB], Al_[NSET][WMB];
  float rx[8];

  const int Mt = XR ? p.M - 1 : p.M;                     // rows in the MFMA row blocks
  const int MB = (p.M + 15) >> 4;
  unsigned arowb[WMB];
#pragma unroll
  for (int i = 0; i < WMB; ++i) arowb[i] = (unsigned)(((long)min(wave * WMB + i, MB - 1) * nchunks * 512 + lane * 8) * 2);
  float xacc = 0.f;
  if constexpr (XR) {
    for (int k = tid; k < (K64 ? ((nchunks + 1) >> 1) * 64 : nchunks * 32); k += 512) xw[k] = k < p.Kc ? p.xrow_w[(long)k * p.xrow_sk] : 0.f;      // (visible after the first barrier)
  }
  const __amdgpu_buffer_rsrc_t rsAh = ssv_buf(p.Ahi), rsAl = ssv_buf(p.Alo), rsX = ssv_buf(Xb);
  auto loadA = [&](int set, int ch) {
    // (readfirstlane: hipcc kept this offset in a vector register in some instantiations and wrapped every load in a waterfall loop)
    const unsigned ub = (unsigned)__builtin_amdgcn_readfirstlane(ch * 1024);
#pragma unroll
    for (int i = 0; i < WMB; ++i) { Ah_[set][i] = ssv_buf_u4(rsAh, arowb[i], ub); Al_[set][i] = ssv_buf_u4(rsAl, arowb[i], ub); }
  };
  const int Lrow = (int)p.sxc;
  const int skg = tid / BN, scol = tid % BN;             // this thread's staging slot (k-group, column) when tid < X_SLOTS
  const bool stager = tid < X_SLOTS;
  const bool cvs = stager && n0 + scol < p.Lx;
  const unsigned voffb = (unsigned)((stager ? 8 * skg : 0) * Lrow + min(n0 + scol, p.Lx - 1)) * 4u;
  const bool ragged = (p.Kc & 31) != 0;
  float xs = 1.f, xinv = 1.f, ainv = 1.f;
  if constexpr (F16) ainv = *p.a_inv;
  auto prefetchX = [&](int ch) {
    if (!ragged || ch + 1 < nchunks) {
#pragma unroll
      for (int i = 0; i < 8; ++i) rx[i] = ssv_buf_f32(rsX, voffb, (unsigned)((ch * 32 + i) * Lrow) * 4u);
    } else {
      const unsigned colo = voffb - (unsigned)(8 * (stager ? skg : 0) * Lrow) * 4u;
#pragma unroll
      for (int i = 0; i < 8; ++i) rx[i] = ssv_buf_f32(rsX, (unsigned)min(ch * 32 + 8 * skg + i, p.Kc - 1) * (unsigned)Lrow * 4u + colo, 0u);
    }
  };
  // the rolling loop's prefetch: ONE path (a buffer whose range is the batch item's Kc rows: channels past Kc read 0, tools/probe/buf_oob.hip),
  // because with the two-path form above hipcc must assume at the loop head that neither path ran and waits for all weight fragments at once
  const __amdgpu_buffer_rsrc_t rsXr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Xb), 0, (int)(((long)(p.Kc - 1) * Lrow + p.Lx) * 4), 0x00020000);
  auto prefetchXr = [&](int ch) {
#pragma unroll
    for (int i = 0; i < 8; ++i) rx[i] = ssv_buf_f32(rsXr, voffb, (unsigned)((ch * 32 + i) * Lrow) * 4u);
  };
  auto commitX = [&](int ch) {
    if (!stager) return;
    const bool last_ragged = ragged && ch + 1 == nchunks;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (cvs && (!last_ragged || ch * 32 + 8 * skg + i < p.Kc)) ? rx[i] : 0.f;
    if constexpr (XR) {
      const float* wq = xw + ch * 32 + 8 * skg;
#pragma unroll
      for (int i = 0; i < 8; ++i) xacc = __builtin_fmaf(wq[i], v[i], xacc);
    }
    uint4 h, l;
    split8s<F16>(v, xs, h, l);
    lds[ch & 1][tid] = h; lds[ch & 1][X_SLOTS + tid] = l;
  };
  auto tap = [&](int set, int ch) {
    const uint4* Xh = lds[ch & 1];
    const uint4* Xl = lds[ch & 1] + X_SLOTS;
    uint4 fb[2][2];
    auto frag = [&](int t, uint4 (&f)[2]) __attribute__((always_inline)) { const int s_ = kq * BN + t * 16 + nq; f[0] = Xh[s_]; f[1] = Xl[s_]; };
    frag(0, fb[0]);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t + 1 < NT) frag(t + 1, fb[(t + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      const uint4 bh = fb[t & 1][0], bl = fb[t & 1][1];
#pragma unroll
      for (int i = 0; i < WMB; ++i) {
        acc[i][t] = mma16<F16>(Al_[set][i], bh, acc[i][t]);
        acc[i][t] = mma16<F16>(Ah_[set][i], bl, acc[i][t]);
        acc[i][t] = mma16<F16>(Ah_[set][i], bh, acc[i][t]);
      }
    }
  };
  loadA(0, 0);
  auto prefetch64 = [&](int c) {                         // (K64) chunk c = channels 64 c + 8 skg + i; rows past Kc read 0 through the ranged buffer
#pragma unroll
    for (int i = 0; i < 8; ++i) rx[i] = ssv_buf_f32(rsXr, voffb, (unsigned)((c * 64 + i) * Lrow) * 4u);
  };
  if constexpr (K64) prefetch64(0);
  else prefetchX(0);
  if constexpr (F16) {
    float sc, inv;
    ssv_pow2_scale(ssv_wave_list_max(p.x_amax + (long)b * p.x_amax_bs, p.x_namax), sc, inv);
    xs = ssv_uniform(sc); xinv = ssv_uniform(inv);
  }
  if constexpr (XR) __syncthreads();                     // xw is staged
  if constexpr (K64) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = cvs ? rx[i] : 0.f;
    if constexpr (XR) {
      const float* wq = xw + 8 * skg;
#pragma unroll
      for (int i = 0; i < 8; ++i) xacc = __builtin_fmaf(wq[i], v[i], xacc);
    }
    uint4 h, l;
    split8s<F16>(v, xs, h, l);
    lds[0][tid] = h; lds[0][X_SLOTS + tid] = l;
    prefetch64(min(1, ((nchunks + 1) >> 1) - 1));
  } else {
    commitX(0);
    if constexpr (SSV_PWLN_ROLL) prefetchXr(min(1, nchunks - 1));
    else if (nchunks > 1) prefetchX(1);
  }
  if constexpr (NSET == 2 && SSV_PWLN_ROLL) {
    // two weight-fragment sets, every load unconditional (clamped chunk index, one-path prefetch: see the rolling loop below)
    const int last = nchunks - 1;
    loadA(1, min(1, last));
    __syncthreads();
    PW_STAMP(1);
    for (int ch = 0; ch < nchunks; ch += 2) {
      tap(0, ch);
      if (ch + 1 < nchunks) commitX(ch + 1);
      prefetchXr(min(ch + 2, last)); loadA(0, min(ch + 2, last));
      __syncthreads();
      if (ch + 1 < nchunks) tap(NSET - 1, ch + 1);
      if (ch + 2 < nchunks) commitX(ch + 2);
      prefetchXr(min(ch + 3, last)); loadA(NSET - 1, min(ch + 3, last));
      __syncthreads();
    }
  } else if constexpr (NSET == 2) {
    if (nchunks > 1) loadA(1, 1);
    __syncthreads();
    PW_STAMP(1);
    for (int ch = 0; ch < nchunks; ch += 2) {
      tap(0, ch);
      if (ch + 1 >= nchunks) break;
      commitX(ch + 1);
      if (ch + 2 < nchunks) { prefetchX(ch + 2); loadA(0, ch + 2); }
      __syncthreads();
      tap(NSET - 1, ch + 1);
      if (ch + 2 < nchunks) {
        commitX(ch + 2);
        if (ch + 3 < nchunks) { prefetchX(ch + 3); loadA(NSET - 1, ch + 3); }
      }
      __syncthreads();
    }
  } else if constexpr (K64) {
    // 64-channel chunks: chunk c = the 32-channel weight chunks 2c and 2c + 1 (the last one may have only the first: Kpad / 32 odd).  Weight fragments
    // as in the rolling form below (one set, row block i re-loaded right behind its MFMAs, now with the NEXT 32-channel chunk's).  Staging of chunk
    // c + 1 from the raw values in rx, one piece behind each of the 8 row-block slots of chunk c: 4 x (mask, extra row's dot product, split one pair),
    // the two LDS stores, 2 x 4 loads of chunk c + 2 -- all unconditional (past the end the last chunk is re-read and lands in the image nobody
    // reads; its extra-row products are masked), so every path into the loop head has the same loads in flight.
    const int nc64 = (nchunks + 1) >> 1, last32 = nchunks - 1;
    unsigned ph[4], pl[4];
    auto piece = [&](int c, int slot8) __attribute__((always_inline)) {
      if (slot8 < 4) {
        const bool live = cvs && c + 1 < nc64;
        const float v0 = live ? rx[2 * slot8] : 0.f, v1 = live ? rx[2 * slot8 + 1] : 0.f;
        if constexpr (XR) {
          const float* wq = xw + (c + 1) * 64 + 8 * skg + 2 * slot8;
          xacc = __builtin_fmaf(wq[0], v0, xacc); xacc = __builtin_fmaf(wq[1], v1, xacc);
        }
        split_pair<F16>(v0, v1, xs, ph[slot8], pl[slot8]);
      } else if (slot8 == 4) lds[(c + 1) & 1][tid] = make_uint4(ph[0], ph[1], ph[2], ph[3]);
      else if (slot8 == 5) lds[(c + 1) & 1][X_SLOTS + tid] = make_uint4(pl[0], pl[1], pl[2], pl[3]);
      else {
        const int cn = min(c + 2, nc64 - 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) rx[4 * (slot8 - 6) + j] = ssv_buf_f32(rsXr, voffb, (unsigned)((cn * 64 + 4 * (slot8 - 6) + j) * Lrow) * 4u);
      }
    };
    auto sub = [&](int c, int sb, bool staging) __attribute__((always_inline)) {
      const uint4* Xh = lds[c & 1];
      const uint4* Xl = lds[c & 1] + X_SLOTS;
      uint4 bh[NT], bl[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) { const int s_ = (sb * 4 + kq) * BN + t * 16 + nq; bh[t] = Xh[s_]; bl[t] = Xl[s_]; }
      const unsigned ubn = (unsigned)__builtin_amdgcn_readfirstlane(min(2 * c + sb + 1, last32) * 1024);
#pragma unroll
      for (int i = 0; i < WMB; ++i) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          acc[i][t] = mma16<F16>(Al_[0][i], bh[t], acc[i][t]);
          acc[i][t] = mma16<F16>(Ah_[0][i], bl[t], acc[i][t]);
          acc[i][t] = mma16<F16>(Ah_[0][i], bh[t], acc[i][t]);
        }
        __builtin_amdgcn_sched_barrier(0);          // re-load and staging piece stay behind this row block's MFMAs, in front of the next block's
        Ah_[0][i] = ssv_buf_u4(rsAh, arowb[i], ubn);
        Al_[0][i] = ssv_buf_u4(rsAl, arowb[i], ubn);
        if (staging) piece(c, sb * WMB + i);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    __syncthreads();
    PW_STAMP(1);
    const int nfull = nchunks >> 1;                 // chunks with both halves
    for (int c = 0; c < nfull; ++c) {
      sub(c, 0, true);
      sub(c, 1, true);
      __syncthreads();
    }
    if (nchunks & 1) sub(nfull, 0, false);          // the odd last 32 channels (its image was staged by chunk nfull - 1's pieces, or by the prologue)
  } else if constexpr (SSV_PWLN_ROLL) {
    // One register set, re-loaded ROW BLOCK BY ROW BLOCK (round 5): the loop runs row block outermost with all NT input fragments of the chunk
    // in registers, so row block i's weight fragments are dead after its 3 NT MFMAs and chunk c + 1's are requested right there -- every
    // fragment gets (WMB - 1) / WMB of a chunk of lead time.  (Before: the whole set was re-loaded after the chunk's last MFMA, i.e. the first
    // MFMA of the next chunk waited for a full L2 round trip -- the 8 waves run in lock step, nothing else was there to cover it.)
    // The re-load is unconditional (the last chunk re-reads itself): a load under a condition makes hipcc drain the whole queue.
    __syncthreads();
    PW_STAMP(1);
    for (int ch = 0; ch < nchunks; ++ch) {
      const uint4* Xh = lds[ch & 1];
      const uint4* Xl = lds[ch & 1] + X_SLOTS;
      uint4 bh[NT], bl[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) { const int s_ = kq * BN + t * 16 + nq; bh[t] = Xh[s_]; bl[t] = Xl[s_]; }
      const unsigned ubn = (unsigned)__builtin_amdgcn_readfirstlane(min(ch + 1, nchunks - 1) * 1024);
#pragma unroll
      for (int i = 0; i < WMB; ++i) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          acc[i][t] = mma16<F16>(Al_[0][i], bh[t], acc[i][t]);
          acc[i][t] = mma16<F16>(Ah_[0][i], bl[t], acc[i][t]);
          acc[i][t] = mma16<F16>(Ah_[0][i], bh[t], acc[i][t]);
        }
        __builtin_amdgcn_sched_barrier(0);          // the re-load stays behind this row block's MFMAs, and in front of the next block's
        Ah_[0][i] = ssv_buf_u4(rsAh, arowb[i], ubn);
        Al_[0][i] = ssv_buf_u4(rsAl, arowb[i], ubn);
      }
      // (the prefetch is unconditional too -- the last chunks re-read the last one -- so that every path into the loop head has the same loads in
      //  flight and hipcc can wait for row block 0's fragments alone, vmcnt(14), instead of for the youngest count over all paths)
      if (ch + 1 < nchunks) commitX(ch + 1);
      prefetchXr(min(ch + 2, nchunks - 1));
      __syncthreads();
    }
  } else {
    __syncthreads();
    PW_STAMP(1);
    for (int ch = 0; ch < nchunks; ++ch) {
      tap(0, ch);
      __builtin_amdgcn_sched_barrier(0);            // the re-load stays behind this chunk's MFMAs
      if (ch + 1 < nchunks) {
        loadA(0, ch + 1);
        commitX(ch + 1);
        if (ch + 2 < nchunks) prefetchX(ch + 2);
      }
      __syncthreads();
    }
  }

  // ---- epilogue: pre, LayerNorm over the M rows of every column, activation
  PW_STAMP(2);
  const float us = F16 ? ssv_uniform(xinv * ainv) : 1.f;
  float* __restrict__ Cb = p.C + (long)b * p.scb;
  float* __restrict__ Yb = q.y + (long)b * q.ybs;
  float csum[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) csum[t] = 0.f;
  // the wave's parked block (rows rbase .. rbase + 15, private to the wave: its LDS operations complete in order) -> dst rows, 16 bytes per lane
  float* pk = park + (SSV_PWLN_PARK ? wave * 16 * LDWP : 0);
  auto store_block = [&](float* __restrict__ dst, long row_stride, int rbase) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      const int e = lane + 64 * it;
      const int row = e / (BN / 4), c4 = e % (BN / 4);
      const int gm = rbase + row, gn = n0 + c4 * 4;
      const f32x4 v = *reinterpret_cast<const f32x4*>(pk + row * LDWP + c4 * 4);
      if (gm < Mt && gn < p.N) {
        float* o = dst + (long)gm * row_stride + gn;
        if (gn + 3 < p.N) { f4u w = {v[0], v[1], v[2], v[3]}; *reinterpret_cast<f4u*>(o) = w; }
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (gn + j < p.N) o[j] = v[j];
        }
      }
    }
  };
  // The per-row parameters of the wave's WMB x 4 rows -- bias (+ the item's broadcast term), gamma, beta -- are ALL loaded here, in front of the
  // first store (round 6).  On gfx9 loads and stores share one in-order counter (vmcnt): a load issued after a row block's stores can only be
  // waited for together with them, so with the loads inside the row-block loops (as until round 6) every row block of `pre` and of `y` began
  // with the acknowledgement of the block before it (tools/isa_store_waits.py lists such waits; the bias / gamma / beta pointers may alias the outputs
  // as far as hipcc knows, so it keeps the order written).  Stamps at 513 x 1300: the y phase 15.1 k -> 10.9 k cycles of a workgroup's ~70 k, the pre phase
  // unchanged; in-step, same box: 116.7 -> 112.3 us.
  float addv[WMB][4], gav[WMB][4], bev[WMB][4];
#pragma unroll
  for (int i = 0; i < WMB; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gmc = min((wave * WMB + i) * 16 + kq * 4 + r, Mt - 1);
      float add = 0.f;
      if (p.bias) add += p.bias[gmc];
      if (p.bias_b) add += p.bias_b[(long)b * p.sbb + gmc];
      addv[i][r] = add; gav[i][r] = q.gamma[gmc]; bev[i][r] = q.beta[gmc];
    }
  float xadd = 0.f, xga = 0.f, xbe = 0.f;                   // the same for row M - 1 (XR; threads tid < BN)
  if constexpr (XR) {
    if (tid < BN) {
      if (p.bias) xadd += p.bias[p.M - 1];
      if (p.bias_b) xadd += p.bias_b[(long)b * p.sbb + p.M - 1];
      xga = q.gamma[p.M - 1]; xbe = q.beta[p.M - 1];
    }
  }
#pragma unroll
  for (int i = 0; i < WMB; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gm = (wave * WMB + i) * 16 + kq * 4 + r;
      const bool rv = gm < Mt;
      const float add = addv[i][r];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float v = rv ? (F16 ? acc[i][t][r] * us : acc[i][t][r]) + add : 0.f;
        acc[i][t][r] = v;
        csum[t] += v;
        if constexpr (SSV_PWLN_PARK) pk[(kq * 4 + r) * LDWP + t * 16 + nq] = v;
        else { const int gn = n0 + t * 16 + nq; if (rv && gn < p.N) Cb[(long)gm * p.scm + gn] = v; }
      }
    }
    if constexpr (SSV_PWLN_PARK) store_block(Cb, p.scm, (wave * WMB + i) * 16);
  }
  const float invM = 1.f / (float)p.M;
  if constexpr (XR) {                                     // row M - 1: the four k-groups' partial sums of a column, bias; stored, and kept for the LayerNorm
    if (stager) xsum[tid] = xacc;                         // (slot tid = skg * BN + scol)
    __syncthreads();
    if (tid < BN) {
      const int gm = p.M - 1, gn = n0 + tid;
      float v = (xsum[tid] + xsum[BN + tid]) + (xsum[2 * BN + tid] + xsum[3 * BN + tid]);
      if constexpr (K64) v += (xsum[4 * BN + tid] + xsum[5 * BN + tid]) + (xsum[6 * BN + tid] + xsum[7 * BN + tid]);
      v += xadd;
      if (gn >= p.N) v = 0.f;
      else Cb[(long)gm * p.scm + gn] = v;
      xrow[tid] = v;
    }
  }
  // xtra: what thread tid < BN adds to its column's sum (the extra row's term); the reduction's own barrier orders xrow before its use
  auto col_reduce = [&](float (&v)[NT], int slot) __attribute__((always_inline)) {       // sum over all rows of the tile; result in colv[slot][column]
#pragma unroll
    for (int t = 0; t < NT; ++t) { v[t] += __shfl_xor(v[t], 16); v[t] += __shfl_xor(v[t], 32); }
    if (kq == 0) {
#pragma unroll
      for (int t = 0; t < NT; ++t) red[wave][t * 16 + nq] = v[t];
    }
    __syncthreads();
    if (tid < BN) {
      float s_ = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s_ += red[w][tid];
      if constexpr (XR) {
        if (slot == 0) s_ += xrow[tid];
        else { const float d = n0 + tid < p.N ? xrow[tid] - colv[0][tid] * invM : 0.f; s_ += d * d; }
      }
      colv[slot][tid] = s_;
    }
    __syncthreads();
  };
  col_reduce(csum, 0);
  PW_STAMP(3);
  float mean[NT], qs[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) { mean[t] = colv[0][t * 16 + nq] * invM; qs[t] = 0.f; }
#pragma unroll
  for (int i = 0; i < WMB; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool rv = (wave * WMB + i) * 16 + kq * 4 + r < Mt;
#pragma unroll
      for (int t = 0; t < NT; ++t) { const float d = rv ? acc[i][t][r] - mean[t] : 0.f; qs[t] += d * d; }
    }
  col_reduce(qs, 1);
  PW_STAMP(4);
  float rstd[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) rstd[t] = rsqrtf(colv[1][t * 16 + nq] * invM + 1e-5f);
  if (q.stats && wave == 0 && kq == 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int gn = n0 + t * 16 + nq;
      if (gn < p.N) { q.stats[(long)b * 2 * p.N + gn] = mean[t]; q.stats[(long)b * 2 * p.N + p.N + gn] = rstd[t]; }
    }
  }
  float am = 0.f;
#pragma unroll
  for (int i = 0; i < WMB; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gm = (wave * WMB + i) * 16 + kq * 4 + r;
      const bool rv = gm < Mt;
      const float ga = gav[i][r], be = bev[i][r];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int gn = n0 + t * 16 + nq;
        float n = (acc[i][t][r] - mean[t]) * rstd[t] * ga + be;
        if (q.act == 1) n = fmaxf(n, 0.f);
        else if (q.act == 2) n = 1.f / (1.f + __expf(-n));
        if (rv && gn < p.N) am = fmaxf(am, fabsf(n));
        if constexpr (SSV_PWLN_PARK) pk[(kq * 4 + r) * LDWP + t * 16 + nq] = n;
        else if (rv && gn < p.N) Yb[(long)gm * p.N + gn] = n;
      }
    }
    if constexpr (SSV_PWLN_PARK) store_block(Yb, p.N, (wave * WMB + i) * 16);
  }
  if constexpr (XR) {
    if (tid < BN && n0 + tid < p.N) {                      // row M - 1 of y (colv is final: behind the second reduction's barrier)
      const int gm = p.M - 1;
      float n = (xrow[tid] - colv[0][tid] * invM) * rsqrtf(colv[1][tid] * invM + 1e-5f) * xga + xbe;
      if (q.act == 1) n = fmaxf(n, 0.f);
      else if (q.act == 2) n = 1.f / (1.f + __expf(-n));
      Yb[(long)gm * p.N + n0 + tid] = n;
      am = fmaxf(am, fabsf(n));
    }
  }
  PW_STAMP(5); PW_STAMP(7);
  if (q.y_amax) {                         // one entry per column tile, the rest of the item's list zeroed by the last tile
    am = ssv_wg_max<8>(am, amx);
    if (tid == 0) {
      float* al = q.y_amax + (long)b * q.namax;
      al[ntile] = am;
      if (ntile == (int)gridDim.x - 1) for (int e = gridDim.x; e < q.namax; ++e) al[e] = 0.f;
    }
  }
}
// WMB row blocks per wave (BM = 128 * WMB >= M), NT column blocks.  Returns SSV_UNSUPPORTED when no instantiation fits.
int ssv_launch_gemm_pwln(const GemmNNB& g, const float* gamma, const float* beta, float* y, long ybs, float* stats, float* y_amax, int namax, int act,
                         hipStream_t st) {
  SSV_CHECK(g.KT == 1 && g.sxn == 1 && g.scn == 1 && !g.epi && !g.perm_h && !g.R && !g.colstats && g.Kpad % 32 == 0, SSV_UNSUPPORTED, "gemm_pwln: plain 1x1 products only");
  SSV_CHECK(g.M <= 640 && g.B <= 65535, SSV_UNSUPPORTED, "gemm_pwln: %d output channels (max 640)", g.M);
  SSV_CHECK(!g.f16 || (g.a_inv && g.x_amax && g.x_namax > 0), SSV_BAD_SHAPE, "gemm_pwln: split-fp16 needs operand scales");
  PwLn q;
  q.g = g; q.gamma = gamma; q.beta = beta; q.y = y; q.ybs = ybs; q.stats = stats; q.y_amax = y_amax; q.namax = namax; q.act = act;
  // M = 513: four row blocks per wave for rows 0 .. 511 and the last row beside the staging (XR) instead of five row blocks
  const bool xr = SSV_PWLN_XROW && g.xrow_w && g.M == 513 && g.Kpad <= 1056;
  const int wmb = xr ? 4 : ssv_cdiv(ssv_cdiv(g.M, 16), 8);
  const int nt = 4;
  const dim3 grid(ssv_cdiv(g.N, 16 * nt), g.B);
  SSV_CHECK(!y_amax || namax >= (int)grid.x, SSV_BAD_SHAPE, "gemm_pwln: scale list shorter than the column tiles");
  if (ssv_shape_log_on()) {
    char nm[96], note[96];
    snprintf(nm, sizeof nm, xr ? "gemm_pwln_kernel<%d, %d, %d, 1>" : "gemm_pwln_kernel<%d, %d, %d>", wmb, nt, g.f16);
    snprintf(note, sizeof note, "B=%d M=%d N=%d K=%d k=1 +LN", g.B, g.M, g.N, g.Kc);
    ssv_shape_log(nm, grid, dim3(512), 2.0 * g.B * g.M * g.N * g.Kc, 4.0 * ((double)g.B * g.Kc * g.N + 2.0 * g.B * g.M * g.N + (double)g.M * g.Kc), note);
  }
#define SSV_PW(W_) if (wmb == W_) { \
    if (g.f16) hipLaunchKernelGGL((gemm_pwln_kernel<W_, 4, 1>), grid, dim3(512), 0, st, q); \
    else hipLaunchKernelGGL((gemm_pwln_kernel<W_, 4, 0>), grid, dim3(512), 0, st, q); \
    return ssv_check_launch("gemm_pwln"); }
  if (xr) {
    if (g.f16) hipLaunchKernelGGL((gemm_pwln_kernel<4, 4, 1, 1>), grid, dim3(512), 0, st, q);
    else hipLaunchKernelGGL((gemm_pwln_kernel<4, 4, 0, 1>), grid, dim3(512), 0, st, q);
    return ssv_check_launch("gemm_pwln (extra row)");
  }
  SSV_PW(1) SSV_PW(2) SSV_PW(3) SSV_PW(4) SSV_PW(5)
#undef SSV_PW
  return ssv_fail(SSV_UNSUPPORTED, "gemm_pwln: no instantiation for %d rows", g.M);
}

// ---- backward of a 1x1 conv + LayerNorm link in ONE launch (round 5): LayerNorm / activation backward, then dX = W^T dPre ------------------
// models/TTSModel.py:128-131, :173-180, :218-231, :343-361 backward.  Until now two launches per link: ln_act_bwd* (dY, pre -> dPre, parameter
// partials, scale list) and the k = 1 data-gradient GEMM re-reading dPre.  As in the forward (gemm_pwln_kernel) a workgroup owns ALL LN rows of a
// 64-column tile, so the LayerNorm backward's two column sums are local.  Phase 1 (512 threads = 32 row groups x 16 column quads; a thread holds
// 8 consecutive rows x 4 columns per unit): dPre from registers -> global (16-byte row pieces) AND, split with the TILE's own power-of-two scale,
// into an LDS image of the GEMM's input operand for every K chunk at once ([chunk][k-group][column][8 halves]: 8 rows of a column = one
// 16-byte slot).  Phase 2: the K loop runs with no staging and no barrier -- transposed weight fragments L2 -> registers (one set, re-loaded row
// block by row block as in gemm_pwln_kernel), input fragments from the image.  Phase 3: the dX tile parked in the image's memory, row-contiguous
// stores.  M = 128 j + 1 LN rows (513): the last row beside the row groups (threads 0 .. 15); Cin = 128 j + 1 output rows: the last one as fp32
// dot products of its weights with the dPre values the threads hold.
// Partial parameter-gradient rows [dgamma | dbeta | dbias] and the scale list keep the layout of the unfused kernels (ssv_ln_act_bwd_rows /
// ssv_amax_rows): this tile's row at part_q * tile, the rows up to the next tile's zeroed; scale entry 4 * tile, the next three zeroed.
template <int WMB, int NU, int F16>
__global__ __launch_bounds__(512, 2) void pwln_bwd_kernel(const PwLnBw q) {
  constexpr int BN = 64, NT = 4;
  constexpr int NCH = NU == 1 ? 8 : 17;                 // K chunks (32 LN rows each) the image holds
  constexpr int LDWP = BN + 4;
  constexpr int IMG_U4 = NCH * 512, PARK_U4 = 8 * 16 * LDWP / 4;
  __shared__ uint4 img[IMG_U4 > PARK_U4 ? IMG_U4 : PARK_U4];     // per chunk: hi [k-group][column] (256 slots), then lo (256 slots)
  __shared__ float red[33 * 2 * BN];                    // column-sum partials of the 32 row groups (+ the extra LN row)
  __shared__ float tot[2 * BN];
  __shared__ float amx[8];
  __shared__ float xw[NU == 2 ? 544 : 1];               // weights of the extra OUTPUT row, one per LN row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned wg = ssv_xcd_order(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
  const int ntile = (int)(wg % gridDim.x), b = (int)(wg / gridDim.x);
  const int n0 = ntile * BN;
  const int M = q.M, L = q.L, act = q.act;
  const int nch = (M + 31) >> 5;
  const bool xlr = (M & 7) == 1;                        // LN row M - 1 beside the row groups (M = 513)
  const int Mg = xlr ? M - 1 : M;                       // rows in the row groups
  const bool xo = NU == 2 && q.xrow_w != nullptr;       // output row Cin - 1 beside the MFMA row blocks
  const int Mt = xo ? q.Cin - 1 : q.Cin;

  // ---------------------------------------------------------------- phase 1: LayerNorm / activation backward
  const int cq = tid & 15, rgt = tid >> 4;
  const int t = n0 + 4 * cq;
  bool cv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) cv[j] = t + j < L;
  const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q.dy + (long)b * q.dy_bs), 0, M * L * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rpr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q.pre + (long)b * M * L), 0, M * L * 4, 0x00020000);
  auto ld4 = [&](__amdgpu_buffer_rsrc_t r, unsigned off, float (&v)[4]) __attribute__((always_inline)) {
    const f32x4 u = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
    v[0] = u[0]; v[1] = u[1]; v[2] = u[2]; v[3] = u[3];
  };
  float a[NU][8][4], xh[NU][8][4];                      // raw: dy, pre -> a = dn * gamma, xh; then a = dPre
  float ea[4] = {0.f, 0.f, 0.f, 0.f}, exh[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int row = min((rgt + 32 * u) * 8 + r, Mg - 1);
      const unsigned o = ((unsigned)row * (unsigned)L + (unsigned)t) * 4u;
      ld4(rdy, o, a[u][r]); ld4(rpr, o, xh[u][r]);
    }
  const bool exrow = xlr && tid < 16;
  if (exrow) { const unsigned o = ((unsigned)(M - 1) * (unsigned)L + (unsigned)t) * 4u; ld4(rdy, o, ea); ld4(rpr, o, exh); }
  if constexpr (NU == 2) {
    if (xo) for (int k = tid; k < 544; k += 512) xw[k] = k < M ? q.xrow_w[(long)k * q.xrow_sk] : 0.f;     // every slot row_d can read, not only the first nch * 32 (visible after the first barrier)
  }
  float mu[4], rs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { const float* sb = q.stats + (long)b * 2 * L + min(t + j, L - 1); mu[j] = sb[0]; rs[j] = sb[L]; }
  float* pblk = q.part + ((long)b * q.part_rows + (long)q.part_q * ntile) * 3 * M;
  float sa[4] = {0.f, 0.f, 0.f, 0.f}, sah[4] = {0.f, 0.f, 0.f, 0.f};
  // one row of 4 columns: dn and the row's (dgamma, dbeta) partials; a <- dn * gamma, xh <- normalised input
  auto row_a = [&](float (&av)[4], float (&hv)[4], int row, bool rok) __attribute__((always_inline)) {
    const int rc = min(row, M - 1);
    const float gg0 = q.gamma[rc], bb = q.beta[rc];
    float q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool v = rok && cv[j];
      const float dy = v ? av[j] : 0.f, gg = v ? gg0 : 0.f;
      const float h = v ? (hv[j] - mu[j]) * rs[j] : 0.f;
      const float n = h * gg + bb;
      float dn;
      if (act == 1) dn = n > 0.f ? dy : 0.f;
      else if (act == 2) { const float s = 1.f / (1.f + __expf(-n)); dn = dy * s * (1.f - s); }
      else dn = dy;
      q0 += dn * h; q1 += dn;
      hv[j] = h; av[j] = dn * gg;
      sa[j] += av[j]; sah[j] += av[j] * h;
    }
    q0 = ssv_row16_sum(q0); q1 = ssv_row16_sum(q1);
    if (cq == 0 && rok) { pblk[rc] = q0; pblk[M + rc] = q1; }
  };
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int r = 0; r < 8; ++r) { const int row = (rgt + 32 * u) * 8 + r; row_a(a[u][r], xh[u][r], row, row < Mg); }
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[(rgt * 2 + 0) * BN + 4 * cq + j] = sa[j]; red[(rgt * 2 + 1) * BN + 4 * cq + j] = sah[j]; }
  if (xlr) {                                            // the extra LN row: threads 0 .. 15 (whose row groups are already summed above)
    if (tid < 16) {
#pragma unroll
      for (int j = 0; j < 4; ++j) sa[j] = sah[j] = 0.f;
      row_a(ea, exh, M - 1, true);
#pragma unroll
      for (int j = 0; j < 4; ++j) { red[(32 * 2 + 0) * BN + 4 * cq + j] = sa[j]; red[(32 * 2 + 1) * BN + 4 * cq + j] = sah[j]; }
    }
  }
  __syncthreads();
  if (tid < 2 * BN) {
    float sum = 0.f;
#pragma unroll 8
    for (int k = 0; k < 32; ++k) sum += red[k * 2 * BN + tid];
    if (xlr) sum += red[32 * 2 * BN + tid];
    tot[tid] = sum;
  }
  __syncthreads();
  const float invM = 1.f / (float)M;
  float m[4], mh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { m[j] = tot[4 * cq + j] * invM; mh[j] = tot[BN + 4 * cq + j] * invM; }
  float am = 0.f;
  float xacc[4] = {0.f, 0.f, 0.f, 0.f};
  float* __restrict__ dPb = q.dpre + (long)b * M * L;
  // a <- dPre = rstd (a - mean(a) - xh mean(a xh)); stored; the row's dbias partial; the extra output row's dot product
  auto row_d = [&](float (&av)[4], float (&hv)[4], int row, bool rok) __attribute__((always_inline)) {
    float q0 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float d = (rok && cv[j]) ? rs[j] * (av[j] - m[j] - hv[j] * mh[j]) : 0.f;
      av[j] = d;
      am = fmaxf(am, fabsf(d));
      q0 += d;
    }
    if (rok) {
      float* o = dPb + (long)row * L + t;
      if (cv[3]) { f4u w = {av[0], av[1], av[2], av[3]}; *reinterpret_cast<f4u*>(o) = w; }
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (cv[j]) o[j] = av[j];
      }
    }
    if constexpr (NU == 2) {
      if (xo) { const float wv = xw[min(row, NCH * 32 - 1)];
#pragma unroll
        for (int j = 0; j < 4; ++j) xacc[j] = __builtin_fmaf(wv, av[j], xacc[j]); }
    }
    q0 = ssv_row16_sum(q0);
    if (cq == 0 && rok) pblk[2 * M + min(row, M - 1)] = q0;
  };
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int r = 0; r < 8; ++r) { const int row = (rgt + 32 * u) * 8 + r; row_d(a[u][r], xh[u][r], row, row < Mg); }
  if (exrow) row_d(ea, exh, M - 1, true);
  // the tile's operand scale; its entry of the item's scale list; the partial rows between this tile's and the next one's
  am = ssv_wg_max<8>(am, amx);
  float xs = 1.f, xinv = 1.f;
  if constexpr (F16) { float sc, inv; ssv_pow2_scale(am, sc, inv); xs = ssv_uniform(sc); xinv = ssv_uniform(inv); }
  if (q.amax && tid < 4) { const int e = 4 * ntile + tid; if (e < q.namax) q.amax[(long)b * q.namax + e] = tid == 0 ? am : 0.f; }
  {
    const int r0 = q.part_q * ntile + 1, r1 = min(q.part_q * (ntile + 1), q.part_rows);
    float* z = q.part + ((long)b * q.part_rows + r0) * 3 * M;
    for (int e = tid; e < (r1 - r0) * 3 * M; e += 512) z[e] = 0.f;
  }
  // the image: 8 rows of one column -> one 16-byte slot of the hi plane and one of the lo plane
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int rb = (rgt + 32 * u) * 8, ch = rb >> 5, kg = (rb >> 3) & 3;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = a[u][r][j];
      uint4 h, l;
      split8s<F16>(v, xs, h, l);
      const int sl = ch * 512 + kg * 64 + 4 * cq + j;
      img[sl] = h; img[sl + 256] = l;
    }
  }
  if (xlr) {                                            // chunk nch - 1 holds the extra row alone: k-group 0, element 0; the rest zero
    const int base = (nch - 1) * 512;
    if (tid < 16) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[8] = {ea[j], 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        uint4 h, l;
        split8s<F16>(v, xs, h, l);
        img[base + 4 * cq + j] = h; img[base + 256 + 4 * cq + j] = l;
      }
    } else if (tid >= 64 && tid < 64 + 192) {           // k-groups 1 .. 3 of that chunk
      const int sl = base + 64 + (tid - 64);
      img[sl] = make_uint4(0, 0, 0, 0); img[sl + 256] = make_uint4(0, 0, 0, 0);
    }
  }
  if constexpr (NU == 2) {
    if (xo) {                                           // the extra output row: sum of the row groups' partial dot products
#pragma unroll
      for (int j = 0; j < 4; ++j) red[rgt * BN + 4 * cq + j] = xacc[j];          // (the extra LN row's term is in threads 0 .. 15's xacc: row_d added it)
    }
  }
  __syncthreads();
  if constexpr (NU == 2) {
    if (xo && tid < BN && n0 + tid < L) {
      float sum = 0.f;
#pragma unroll 8
      for (int k = 0; k < 32; ++k) sum += red[k * BN + tid];
      q.dx[(long)b * q.dx_bs + (long)(q.Cin - 1) * L + n0 + tid] = sum;
    }
  }

  // ---------------------------------------------------------------- phase 2: dX tile = W^T dPre, K = the LN rows, straight from the image
  const int kq = lane >> 4, nq = lane & 15;
  f32x4 acc[WMB][NT];
#pragma unroll
  for (int i = 0; i < WMB; ++i)
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) acc[i][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  uint4 Ah_[WMB], Al_[WMB];
  const int MB = (q.Cin + 15) >> 4;
  unsigned arowb[WMB];
#pragma unroll
  for (int i = 0; i < WMB; ++i) arowb[i] = (unsigned)(((long)min(wave * WMB + i, MB - 1) * nch * 512 + lane * 8) * 2);
  const __amdgpu_buffer_rsrc_t rsAh = ssv_buf(q.Ahi), rsAl = ssv_buf(q.Alo);
#pragma unroll
  for (int i = 0; i < WMB; ++i) { Ah_[i] = ssv_buf_u4(rsAh, arowb[i], 0u); Al_[i] = ssv_buf_u4(rsAl, arowb[i], 0u); }
  for (int ch = 0; ch < nch; ++ch) {
    const uint4* Xh = img + ch * 512;
    const uint4* Xl = Xh + 256;
    uint4 bh[NT], bl[NT];
#pragma unroll
    for (int tt = 0; tt < NT; ++tt) { const int s_ = kq * BN + tt * 16 + nq; bh[tt] = Xh[s_]; bl[tt] = Xl[s_]; }
    const unsigned ubn = (unsigned)__builtin_amdgcn_readfirstlane(min(ch + 1, nch - 1) * 1024);
#pragma unroll
    for (int i = 0; i < WMB; ++i) {
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        acc[i][tt] = mma16<F16>(Al_[i], bh[tt], acc[i][tt]);
        acc[i][tt] = mma16<F16>(Ah_[i], bl[tt], acc[i][tt]);
        acc[i][tt] = mma16<F16>(Ah_[i], bh[tt], acc[i][tt]);
      }
      __builtin_amdgcn_sched_barrier(0);
      Ah_[i] = ssv_buf_u4(rsAh, arowb[i], ubn);
      Al_[i] = ssv_buf_u4(rsAl, arowb[i], ubn);
    }
  }
  // ---------------------------------------------------------------- phase 3: dX rows, parked per wave, row-contiguous stores
  __syncthreads();                                      // every wave is done with the image
  const float us = F16 ? ssv_uniform(xinv * *q.a_inv) : 1.f;
  float* pk = reinterpret_cast<float*>(img) + wave * 16 * LDWP;
  float* __restrict__ Xo = q.dx + (long)b * q.dx_bs;
#pragma unroll
  for (int i = 0; i < WMB; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) pk[(kq * 4 + r) * LDWP + tt * 16 + nq] = F16 ? acc[i][tt][r] * us : acc[i][tt][r];
    const int rbase = (wave * WMB + i) * 16;
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      const int e = lane + 64 * it;
      const int row = e / (BN / 4), c4 = e % (BN / 4);
      const int gm = rbase + row, gn = n0 + c4 * 4;
      const f32x4 v = *reinterpret_cast<const f32x4*>(pk + row * LDWP + c4 * 4);
      if (gm < Mt && gn < L) {
        float* o = Xo + (long)gm * L + gn;
        if (gn + 3 < L) { f4u w = {v[0], v[1], v[2], v[3]}; *reinterpret_cast<f4u*>(o) = w; }
        else {
#pragma unroll
          for (int j = 0; j < 4; ++j) if (gn + j < L) o[j] = v[j];
        }
      }
    }
  }
}
// true when the link's backward can take the one-launch kernel (split-fp16 / split-bf16 planes of the TRANSPOSED weight given; dense dPre)
bool ssv_pwln_bwd_fused_ok(int B, int Cin, int Cout, int L) {
  if (!SSV_PWLN_BWD_FUSED) return false;
  // Measured in-step at B = 32 (round 5): links with up to 256 LN rows 24.3 us against 17.9 + 22.6 us in two launches; with 512 / 513 LN rows
  // (one 155 KB workgroup per CU: its three phases cannot overlap with anything) 139 against 74 + 95 us for 513 -> 513 but 139 against 74 + 67 for
  // 512 -> 513 and 123 against ~110 for 256 -> 512: no gain over the step.  Default: the small form only.  SSV_PWLN_BWD=0: never; =2: every shape.
  int mode = 1;
  if (const char* e = ssv_tuning(SSV_T_PWLN_BWD)) mode = atoi(e);
  if (mode == 0 || (mode != 2 && Cout > 256)) return false;
  const bool m_ok = (Cout % 8 == 0 && Cout >= 32 && Cout <= 512) || Cout == 513;
  const bool c_ok = (Cin >= 32 && Cin <= 512) || (Cin == 513 && Cout > 256);
  return m_ok && c_ok && B <= 65535 && L >= 16 && (long)Cout * L < (1L << 29) && (long)Cin * L < (1L << 29);
}
int ssv_launch_pwln_bwd(const PwLnBw& q, int B, int f16, hipStream_t st) {
  SSV_CHECK(ssv_pwln_bwd_fused_ok(B, q.Cin, q.M, q.L), SSV_UNSUPPORTED, "pwln_bwd: shape %d -> %d not supported", q.Cin, q.M);
  SSV_CHECK(q.dy && q.pre && q.stats && q.gamma && q.beta && q.dpre && q.part && q.Ahi && q.Alo && q.dx && (!f16 || q.a_inv), SSV_BAD_SHAPE, "pwln_bwd: null argument");
  SSV_CHECK(q.Cin % 128 != 1 || q.Cin < 128 || q.xrow_w, SSV_BAD_SHAPE, "pwln_bwd: %d output rows need the extra row's weights", q.Cin);
  const dim3 grid(ssv_cdiv(q.L, 64), B);
  SSV_CHECK(q.part_q >= 1 && q.part_rows >= q.part_q * ((int)grid.x - 1) + 1 && (!q.amax || q.namax >= 4 * (int)grid.x - 3), SSV_BAD_SHAPE, "pwln_bwd: partial rows / scale list too short");
  const int nu = q.M <= 256 ? 1 : 2;
  const int wmb = q.Cin <= 128 ? 1 : (q.Cin <= 256 ? 2 : 4);
  if (ssv_shape_log_on()) {
    char nm[96], note[96];
    snprintf(nm, sizeof nm, "pwln_bwd_kernel<%d, %d, %d>", wmb, nu, f16);
    snprintf(note, sizeof note, "B=%d Cin=%d N=%d Cout=%d LN bwd + k=1 data gradient", B, q.Cin, q.L, q.M);
    ssv_shape_log(nm, grid, dim3(512), 2.0 * B * q.Cin * q.L * q.M, 4.0 * ((double)B * q.M * q.L * 3 + (double)B * q.Cin * q.L + (double)q.M * q.Cin), note);
  }
#define SSV_PB(W_, U_) if (wmb == W_ && nu == U_) { \
    if (f16) hipLaunchKernelGGL((pwln_bwd_kernel<W_, U_, 1>), grid, dim3(512), 0, st, q); \
    else hipLaunchKernelGGL((pwln_bwd_kernel<W_, U_, 0>), grid, dim3(512), 0, st, q); \
    return ssv_check_launch("pwln_bwd"); }
  SSV_PB(1, 1) SSV_PB(2, 1) SSV_PB(4, 1) SSV_PB(1, 2) SSV_PB(2, 2) SSV_PB(4, 2)
#undef SSV_PB
  return ssv_fail(SSV_UNSUPPORTED, "pwln_bwd: no instantiation");
}

