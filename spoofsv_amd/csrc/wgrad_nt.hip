// Conv1d weight gradient on split MFMAs: gemm_nt_bf3_kernel (k = 1, and k = 3 beyond the ring kernel's reach), the tile / slab plan and the launcher of both weight-gradient kernels.
#include "bf3_common.h"

// ---- NT (weight gradient) -------------------------------------------------------------------------------------------------
//   C(z,m,c,j) = sum_{b = z, z+bstep, ..} sum_t A(b,m,t) * X(b,c,t+shift[j]);  rows contiguous in t for both operands.
// The reduction runs over time, so a tap's dilation shift is a misaligned shift along k: every tap needs its own staged
// copy of the input rows.  The kernel therefore steps over (batch item, 64-step chunk, tap), tap fastest:
//   * the waves split M, so a wave's A rows (dL/dH) are private: its fragments go global -> registers (two 16-byte buffer
//     loads per 8 time steps = the window's two fragment tuples), are split there in place once per chunk and reused by the
//     KT taps -- A never touches LDS (see AH / AL in the kernel);
//   * per step only ONE tap's input tile (16*NTC channels x 64 steps) is split and staged, into one of two LDS buffers:
//     the step's MFMAs read buffer s while the next step's tile is written to buffer s^1 -- one barrier per step, and
//     16 staging registers instead of 48 (staging all taps at once put the 128 x 64 x 3 tile at 256 VGPRs with spills,
//     and every scratch reload waits for vmcnt(0), i.e. for the whole prefetch in flight).
// Loads are issued raw, one step (input) or one chunk (A) ahead, with no branch in the prefetch (see load8c / split_edge).
// Tuning builds only (-DSSV_NT_STAMP): wave 0 of workgroup 0 records s_memtime at six points of every step from step 24 on
// (8 steps); read back with ssv_debug_nt_stamps().  Results are unaffected.
#ifdef SSV_NT_STAMP
__device__ unsigned long long ssv_nt_stamps[64];
__device__ unsigned long long ssv_nt_wg[4096 * 4];      // per workgroup (first 4096): s_memrealtime at entry / exit, shader clock at entry / exit
extern "C" int ssv_debug_nt_wg(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ssv_nt_wg), sizeof(ssv_nt_wg)); }
#define NT_WG(k, v) do { const unsigned w_ = blockIdx.z * gridDim.x + blockIdx.x; if (threadIdx.x == 0 && w_ < 4096u) ssv_nt_wg[w_ * 4 + (k)] = (v); } while (0)
#define NT_STAMP(k) do { if (stamp_on && (unsigned)(stamp_s - 24) < 8u) ssv_nt_stamps[(stamp_s - 24) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
extern "C" int ssv_debug_nt_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ssv_nt_stamps), sizeof(ssv_nt_stamps)); }
#else
#define NT_STAMP(k) do {} while (0)
#define NT_WG(k, v) do {} while (0)
#endif
constexpr int NT_FD = 1;      // LDS fragment groups read ahead of the MFMAs (see the step loop)
// WV: waves per workgroup -- 4, or 8 (round 6): a 256-row tile whose waves share the staged input tile, one workgroup per CU
template <int KT, int WM, int NTC, int F16, int XR = 0, int WV = 4>
__global__ __launch_bounds__(64 * WV, WV == 4 ? 2 : 1) void gemm_nt_bf3_kernel(const GemmNT p, const int mtiles) {
  constexpr int NTH = 64 * WV;
  constexpr int KB = 64, KG = KB / 8, KS = KB / 32;         // time steps per chunk, k-groups, MFMA k-steps
  constexpr int NCH = 16 * NTC;
  constexpr int X_SLOTS = KG * NCH;                         // 16-byte slots of one tap's tile (multiple of 256)
  constexpr int NX = X_SLOTS / NTH;
  static_assert(X_SLOTS % NTH == 0, "tile slots must be a multiple of the workgroup size");
  // [buffer][hi plane | lo plane], slot = kg*NCH + (channel ^ kg).  The staging threads take kg fastest (8 lanes = 256
  // contiguous bytes of one channel row in global memory), so without the XOR the 8 lanes of a ds_write_b128 group would
  // write slots 1 KB apart -- one bank set, an 8-way conflict that cost more than the step's MFMAs.  With it they land on
  // 8 distinct 16-byte bank groups, and the fragment reads (16 consecutive channels per quarter wave) stay conflict-free.
  __shared__ uint4 lds[2][2 * X_SLOTS];
  __shared__ float amax_sm[2 * WV];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  NT_WG(0, __builtin_amdgcn_s_memrealtime()); NT_WG(2, __builtin_readcyclecounter());
  const unsigned wg = ssv_xcd_order(blockIdx.x + gridDim.x * blockIdx.z, gridDim.x * gridDim.z);   // a slab's tiles share an XCD
  const int bxx = (int)(wg % gridDim.x);
  int z = (int)(wg / gridDim.x);
  // operands: the launch's own, or those of job z / Z (several equal-shaped layers in one launch, see GemmNT)
  const float* __restrict__ Ap = p.A;
  const float* __restrict__ Xp = p.X;
  float* __restrict__ Cp = p.C;
  int shj[3] = {p.shift[0], p.shift[1], p.shift[2]};
  const float* __restrict__ a_amax = p.a_amax;
  const float* __restrict__ x_amax = p.x_amax;
  int a_namax = p.a_namax, x_namax = p.x_namax;
  if (p.jobs) {
    const int job = z / p.Z;
    z -= job * p.Z;
    // member by member: a by-value copy of the entry reaches the pointers through integer loads, and hipcc then knows nothing of
    // their address space -- every operand load of this kernel was a flat_load (see ssv_global)
    const ssv_wgrad_job* __restrict__ jb = p.jobs + job;
    Ap = ssv_global(jb->dy); Xp = ssv_global(jb->x); Cp = p.C + (long)job * p.Z * p.scz;
    shj[0] = jb->shift[0]; shj[1] = jb->shift[1]; shj[2] = jb->shift[2];
    a_amax = ssv_global(jb->dy_amax); x_amax = ssv_global(jb->x_amax); a_namax = jb->dy_namax; x_namax = jb->x_namax;
  }
  // split-fp16: one power-of-two scale per operand tensor (the reduction runs over the batch), undone in the epilogue
  float as = 1.f, xs = 1.f, us = 1.f;
  auto scales = [&]() {
    if constexpr (F16) {
      float sa, sx, ia, ix;
      ssv_pow2_scale(ssv_list_max<WV>(a_amax, a_namax, amax_sm), sa, ia);      // long lists (B * tiles entries): shared among the waves
      ssv_pow2_scale(ssv_list_max<WV>(x_amax, x_namax, amax_sm + WV), sx, ix);
      as = ssv_uniform(sa); xs = ssv_uniform(sx);
      us = ssv_uniform(ia * ix);
    }
  };
  // Tile order inside a slab.  Few tiles (every tile of the slab resident on its XCD at once): any order.  Many (the LSTM's 3072 x 768 outputs: 72-192
  // tiles on 32-64 slots): the COLUMN tiles of a row tile next to each other -- they read the same dH rows, the larger operand (4H rows against H), at the
  // same time, so it comes from HBM once per slab instead of once per round of column tiles (config 5's weight gradients 9.6 -> 8.5 ms).
  const int ctiles_ = (int)gridDim.x / mtiles;
  const bool ct_fast = gridDim.x > 64u;
  const int mt = ct_fast ? bxx / ctiles_ : bxx % mtiles, ct = ct_fast ? bxx % ctiles_ : bxx / mtiles;
  const int m0 = mt * 16 * WV * WM, c0 = ct * NCH;
  const int tchunks = (p.La + KB - 1) / KB;
  const int kq = lane >> 4, nq = lane & 15;

  f32x4 acc[WM][KT][NTC];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < KT; ++j)
#pragma unroll
      for (int q = 0; q < NTC; ++q) acc[i][j][q] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float rx[NX][8];                     // raw loads in flight: the tile of step s + 2
  int mx[NX];                          // edge windows only: validity bits (low 8) | offset clamp distance << 8
  static_assert(!XR || (KT == 1 && WV == 4 && (256 % KG) == 0), "extra row: k = 1; a thread's slots share their k-group");
  const bool xr_on = XR && mt == 0;
  float xacc[XR ? NX : 1];
  uint4 xra[XR ? 2 : 1];               // dH(M - 1, t0 + 8 kg .. + 7) of the tile in flight (raw; what lies past the row meets masked input)
#pragma unroll
  for (int r = 0; r < (XR ? NX : 1); ++r) xacc[r] = 0.f;
  // dH fragments: two sets of (hi, lo) register tuples; set n & 1 is chunk n's.  The next chunk's windows are LOADED into the other
  // set (a window's two 16-byte loads = its two tuples) and split there in place, dword by dword (split8p's order of the time steps):
  // no staging registers.  (The split of the next set woven behind the MFMAs of a chunk's last tap -- one half-rate split instruction
  // hides behind a 16x16x32 MFMA of the same wave, tools/probe/mfma_valu.hip -- was built on this layout and measured: no change
  // in-step, 268.5 vs 268.3 us; with two waves on a SIMD the other wave's MFMAs already fill those slots.)  dH needs no masks and no clamps:
  //   * it is read by BUFFER loads whose range is the tensor (per-dword range check: what lies past the end reads 0; its offsets are
  //     never negative), so a window may run past its row -- into the next row's values, or into zeros;
  //   * time steps at or past the row length meet an input tile that is zero there (x_edge / the mask of the input include t < La).
  //   (A NaN or Inf at the head of the NEXT row would so reach this row's sums as NaN; with masks it stayed in its own row.)
  uint4 AH[2][WM][KS], AL[2][WM][KS];

  // A window = 8 consecutive time steps of one row, at any alignment.  Element offsets are 32-bit (the launcher checks
  // that both tensors span < 2^30 elements): a per-thread row offset, fixed for the whole kernel, plus a wave-uniform
  // (batch item, chunk, tap) offset.  Two kinds of chunk / step, told apart by a wave-uniform test:
  //   interior -- every window of the tile lies inside its row: plain loads, plain split; nothing else on the VALU;
  //   edge     -- a window may start before 0 or run past the row (first/last chunk, ragged tile): it is still loaded RAW
  //     by two 16-byte loads at its true offset (an edge window simply runs into the neighbouring row) and the validity
  //     bits are applied when the values are split, a step later.  Masking at load time makes hipcc branch around each
  //     load and wait for it.  Only a window that would leave the TENSOR (head of its first row, tail of its last) has
  //     its offset clamped; the clamp distance travels with the mask and the split shifts the values back into place.
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Ap), 0,
      (int)(((long)(p.B - 1) * p.sab + (long)(p.M - 1) * p.sam + p.La) * 4), 0x00020000);
  const int x_span = (int)((long)(p.B - 1) * p.sxb + (long)(p.Nc - 1) * p.sxc + p.Lx) - 8;
  int arow[WM], xrow[NX];
#pragma unroll
  for (int i = 0; i < WM; ++i) arow[i] = (min(m0 + wave * WM * 16 + i * 16 + nq, p.M - 1) * (int)p.sam + 8 * kq) * 4;     // BYTE offset of the buffer loads
  const int xra_off = ((p.M - 1) * (int)p.sam + 8 * (tid % KG)) * 4;
#pragma unroll
  for (int r = 0; r < NX; ++r) {
    const int f = tid + NTH * r;
    xrow[r] = min(c0 + f / KG, p.Nc - 1) * (int)p.sxc + 8 * (f % KG);
  }
  const bool rows_in_c = c0 + NCH <= p.Nc;

  auto load8 = [&](const float* __restrict__ base, int off, float (&v)[8]) {
    // uniform base + zero-extended 32-bit BYTE offset (operands span < 2^30 elements): the saddr form of global_load
    // address space spelled out: with operand pointers that may come from a job table hipcc emitted flat_load here (see ssv_global)
    typedef __attribute__((address_space(1))) const char gchar;
    typedef __attribute__((address_space(1))) const f4u gf4u;
    gchar* q = (gchar*)reinterpret_cast<const char*>(base) + ((unsigned)off << 2);
    const f4u a = *(gf4u*)q;
    const f4u c = *(gf4u*)(q + 16);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = c.x; v[5] = c.y; v[6] = c.z; v[7] = c.w;
  };
  // Interior and edge windows are loaded by the SAME instructions (the offset clamped by one v_med3 either way): with the loads
  // under an interior / edge branch, hipcc's waitcnt bookkeeping differed between the two paths and it drained vmcnt to 0 in front
  // of the edge path's loads.  Only the mask (VALU) is edge-only.
  auto load8c = [&](const float* __restrict__ base, int off, int span, float (&v)[8]) -> int {
    const int oc = min(max(off, 0), span);
    load8(base, oc, v);
    return off - oc;
  };
  auto edge_meta = [&](int d, int t, int len, bool row_ok, int over2 = 0) -> int {     // over2: elements cut off the window's end by a second limit
    const int sl = min(max(-t, 0), 8), sh = max(min(max(t + 8 - len, 0), 8), min(max(over2, 0), 8));
    const int m = row_ok ? (int)((0xFFu << sl) & (0xFFu >> sh) & 0xFFu) : 0;
    return m | (d << 8);
  };
  auto edge_vals = [&](const float (&raw)[8], int meta, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = raw[i];
    const int d = meta >> 8;
    // wave-uniform test: a scalar branch hipcc cannot turn into straight-line selects (128 VALU instructions per window)
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(d != 0) != 0ull, 0)) {   // v[i] must be the element at offset oc + (i + d)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float r = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) r = (i + d == k) ? raw[k] : r;
        v[i] = r;
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = ((meta >> i) & 1) ? v[i] : 0.f;
  };
  auto split_edge = [&](const float (&raw)[8], int meta, float sc, uint4& h, uint4& l) {
    float v[8];
    edge_vals(raw, meta, v);
    split8p<F16>(v, sc, h, l);
  };

  // chunk cursors (wave-uniform): batch item and first time step of chunks n .. n+3
  // whole-item slabs (bstep > 0): slab z reduces over items z, z + bstep, ...;  range slabs (bstep == 0): over the z-th of Z equal ranges of the
  // item-major sequence of all B x tchunks chunks (a finer cut: the work per workgroup need not be a whole number of items)
  const int bstep = p.bstep > 0 ? p.bstep : 1;
  int total, cb[4], ct0[4];
  if (p.bstep > 0) {
    total = ((p.B - z + p.bstep - 1) / p.bstep) * tchunks;                  // chunks this workgroup reduces over
    cb[0] = z; ct0[0] = 0;
  } else {
    const int all = p.B * tchunks, per = (all + p.Z - 1) / p.Z, start = z * per;
    total = max(min(per, all - start), 0);
    cb[0] = start / tchunks; ct0[0] = (start % tchunks) * KB;
  }
  auto next_chunk = [&](int b, int t0, int& nb_, int& nt0) __attribute__((always_inline)) {
    nt0 = t0 + KB; nb_ = b;
    if (nt0 >= tchunks * KB) { nt0 = 0; nb_ = b + bstep; }
  };
#pragma unroll
  for (int k = 1; k < 4; ++k) next_chunk(cb[k - 1], ct0[k - 1], cb[k], ct0[k]);
  auto x_edge = [&](int t0, int j) __attribute__((always_inline)) -> bool {
    return !(rows_in_c && t0 + shj[j] >= 0 && t0 + KB + shj[j] <= p.Lx && t0 + KB <= p.La);       // (the last: dH is not masked, see AH / AL)
  };

  auto loadA = [&](auto set, int b, int t0) __attribute__((always_inline)) {                               // -> AH / AL[set], raw
    constexpr int SET = decltype(set)::value;
    const unsigned so = (unsigned)(b * (int)p.sab + t0) * 4u;   // uniform
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) {
        AH[SET][i][s2] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, arow[i] + s2 * 128, (int)so, 0));
        AL[SET][i][s2] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, arow[i] + s2 * 128 + 16, (int)so, 0));
      }
  };
  auto raw8 = [&](const uint4& t0_, const uint4& t1_, float (&v)[8]) __attribute__((always_inline)) {
    v[0] = __builtin_bit_cast(float, t0_.x); v[1] = __builtin_bit_cast(float, t0_.y); v[2] = __builtin_bit_cast(float, t0_.z); v[3] = __builtin_bit_cast(float, t0_.w);
    v[4] = __builtin_bit_cast(float, t1_.x); v[5] = __builtin_bit_cast(float, t1_.y); v[6] = __builtin_bit_cast(float, t1_.z); v[7] = __builtin_bit_cast(float, t1_.w);
  };
  auto splitA = [&](auto set) __attribute__((always_inline)) {                                             // AH / AL[set]: raw -> (hi, lo), in place
    constexpr int SET = decltype(set)::value;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) {
        float v[8];
        raw8(AH[SET][i][s2], AL[SET][i][s2], v);
        split8p<F16>(v, as, AH[SET][i][s2], AL[SET][i][s2]);
      }
  };
  auto loadX = [&](int b, int t0, int j) __attribute__((always_inline)) {                                  // -> rx (/ mx)
    const int base = b * (int)p.sxb + t0 + shj[j];
    int dd[NX];
#pragma unroll
    for (int r = 0; r < NX; ++r) dd[r] = load8c(Xp, base + xrow[r], x_span, rx[r]);
    if constexpr (XR) {                // every workgroup issues them (a branch around loads costs hipcc's waitcnt bookkeeping more than two L2 hits)
      const unsigned so = (unsigned)(b * (int)p.sab + t0) * 4u;
      xra[0] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, xra_off, (int)so, 0));
      xra[1] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, xra_off + 16, (int)so, 0));
    }
    if (x_edge(t0, j)) {
#pragma unroll
      for (int r = 0; r < NX; ++r) {
        const int f = tid + NTH * r;
        mx[r] = edge_meta(dd[r], t0 + shj[j] + 8 * (f % KG), p.Lx, c0 + f / KG < p.Nc, t0 + 8 * (f % KG) + 8 - p.La);
      }
    }
  };
  auto commitX = [&](auto buf, int t0, int j) __attribute__((always_inline)) {                             // rx -> LDS buffer `buf`
    constexpr int S = decltype(buf)::value;
    uint4* Xh = lds[S];
    uint4* Xl = lds[S] + X_SLOTS;
    const bool edge = x_edge(t0, j);
#pragma unroll
    for (int r = 0; r < NX; ++r) {
      const int f = tid + NTH * r;
      const int kg = f % KG, c = f / KG;
      uint4 h, l;
      if (!edge) split8p<F16>(rx[r], xs, h, l);
      else split_edge(rx[r], mx[r], xs, h, l);
      Xh[kg * NCH + (c ^ kg)] = h; Xl[kg * NCH + (c ^ kg)] = l;          // XOR swizzle, see the slot comment above
      if constexpr (XR) {
        if (xr_on) {                   // row M - 1 of the product: dH(M - 1, t) x(c, t) over this slot's 8 time steps
          float av[8], xv[8];
          raw8(xra[0], xra[1], av);
          if (!edge) {
#pragma unroll
            for (int i = 0; i < 8; ++i) xv[i] = rx[r][i];
          } else edge_vals(rx[r], mx[r], xv);
#pragma unroll
          for (int i = 0; i < 8; ++i) xacc[r] = __builtin_fmaf(av[i], xv[i], xacc[r]);
        }
      }
    }
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  const int steps = total * KT;

  // One chunk = KT steps (tap fastest); step s = n*KT + j uses LDS buffer s & 1.  In step s:
  //   MFMAs of step s on buffer s&1  |  tile s+1 (loaded a step ago) is split into buffer (s+1)&1  |  the loads of tile s+2
  //   are issued into the registers just freed  ->  a tile's loads have a barrier and a step's MFMAs (~2,000 cycles) to land.
  // (Round 2 kept two register sets and issued tile s+3: 16 more VGPRs, and hipcc's vmcnt(0) in front of every batch -- see
  // STEADY below -- made it wait for tile s+2 anyway.)
  // STEADY = every commit and load of the chunk is known to be due (no "is there a step s + 3" tests).  Not for the branches
  // saved: hipcc's s_waitcnt placement merges its bookkeeping over all paths, and the path that skips a commit leaves that
  // tile's loads "possibly in flight" -- it then drained vmcnt to 0 in front of EVERY batch of loads (their address registers
  // reuse the tile's), so each batch waited for the previous one to land: four exposed L2 round trips per chunk, 7,000 of a
  // chunk's 10,000 cycles by s_memtime stamps.  The last chunks run the tested form.
  auto chunk = [&](auto steady, auto par, int n) __attribute__((always_inline)) {
    constexpr bool STEADY = decltype(steady)::value;
    constexpr int PAR = decltype(par)::value;                               // parity of this chunk's first step
    const bool more = STEADY || n + 1 < total;
#ifdef SSV_NT_STAMP
    const bool stamp_on = blockIdx.x == 0 && blockIdx.z == 0 && tid == 0;
#endif
    using CUR = std::integral_constant<int, PAR>;                           // this chunk's dH set (n & 1 = the parity of its first step: KT is odd)
    using NXT = std::integral_constant<int, PAR ^ 1>;
    if (more) loadA(NXT{}, cb[1], ct0[1]);                                  // lands during this chunk's first steps
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      const int q_ = (PAR + j) & 1;
#ifdef SSV_NT_STAMP
      const int stamp_s = n * KT + j;
#endif
      NT_STAMP(0);
      const uint4* Xh = lds[q_];
      const uint4* Xl = lds[q_] + X_SLOTS;
      // The input fragments of group g + FD are read from LDS BEFORE the MFMAs of group g are issued (a group = one 16-channel
      // block of one k-step: 2 reads, 3 WM MFMAs).  Left to itself hipcc issues a group's reads right in front of its MFMAs and
      // parks the wave on lgkmcnt for the LDS latency eight times per step -- a third of the wave's cycles by the SQ counters.
      constexpr int G = KS * NTC;
      uint4 fb[NT_FD + 1][2];
      auto frag = [&](int g, uint4 (&f)[2]) __attribute__((always_inline)) {
        const int kg = (g / NTC) * 4 + kq;
        const int xs_ = kg * NCH + (((g % NTC) * 16 + nq) ^ kg);
        f[0] = Xh[xs_]; f[1] = Xl[xs_];
      };
#pragma unroll
      for (int g = 0; g < NT_FD; ++g) frag(g, fb[g]);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (g + NT_FD < G) frag(g + NT_FD, fb[(g + NT_FD) % (NT_FD + 1)]);
        __builtin_amdgcn_sched_barrier(0);                                   // or the scheduler sinks the reads back to their use
        const int s2 = g / NTC, q = g % NTC;
#if !SSV_NT_NOBREAK
        if (s2 > 0 && q == 0 && ct0[0] + 32 * s2 >= p.La) break;           // ragged last chunk: the k-steps from here on lie past the row end (the input tile is zero there)
#endif
        const uint4 bh = fb[g % (NT_FD + 1)][0];
        const uint4 bl = fb[g % (NT_FD + 1)][1];
#pragma unroll
        for (int i = 0; i < WM; ++i) {
          const uint4 a_h = AH[CUR::value][i][s2];
          const uint4 a_l = AL[CUR::value][i][s2];
          acc[i][j][q] = mma16<F16>(a_l, bh, acc[i][j][q]);
          acc[i][j][q] = mma16<F16>(a_h, bl, acc[i][j][q]);
          acc[i][j][q] = mma16<F16>(a_h, bh, acc[i][j][q]);
        }
      }
      NT_STAMP(1);
      const int c1 = (j + 1) / KT, j1 = (j + 1) % KT;                       // step s+1: chunk n + c1, tap j1
      const int c2 = (j + 2) / KT, j2 = (j + 2) % KT;                       // step s+2
      const int s = n * KT + j;
#ifdef SSV_NT_STAMP
      asm volatile("" : "+v"(rx[NX - 1][7]));        // the tile's last load has landed
      NT_STAMP(6);
#endif
      if (STEADY || s + 1 < steps) {
        if (((PAR + j) & 1) == 0) commitX(P1{}, ct0[c1], j1); else commitX(P0{}, ct0[c1], j1);
      }
      NT_STAMP(2);
#if SSV_NT_SPLIT_FIRST
      // the next chunk's dH is split BEFORE the loads of tile s + 2 are issued (round 6): behind them, hipcc's waits for the dH registers came out as
      // vmcnt(5) .. vmcnt(0) right after those loads, i.e. every step ended by waiting for the tile it had just requested
      if (j == KT - 1 && more) splitA(NXT{});
      NT_STAMP(3);
      if (STEADY || s + 2 < steps) loadX(cb[c2], ct0[c2], j2);
#else
      if (STEADY || s + 2 < steps) loadX(cb[c2], ct0[c2], j2);
      NT_STAMP(3);
      if (j == KT - 1 && more) splitA(NXT{});
#endif
      NT_STAMP(4);
      __syncthreads();
      NT_STAMP(5);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { cb[k] = cb[k + 1]; ct0[k] = ct0[k + 1]; }
    next_chunk(cb[2], ct0[2], cb[3], ct0[3]);
  };

  if (total > 0) {
    // prologue: A of chunk 0 split, tile 0 staged, tile 1 in flight
    loadA(P0{}, cb[0], ct0[0]);
    loadX(cb[0], ct0[0], 0);
    scales();
    splitA(P0{});
    commitX(P0{}, ct0[0], 0);
    if (steps > 1) loadX(cb[1 / KT], ct0[1 / KT], 1 % KT);
    __syncthreads();
    using ST = std::integral_constant<bool, true>;
    using TL = std::integral_constant<bool, false>;
    const int nfull = (steps - 2) / KT;                 // chunks n < nfull have all their steps' s + 2 < steps
    static_assert((KT & 1) == 1, "chunk parities alternate");
    int n = 0;
    for (; n + 1 < nfull; n += 2) {
      chunk(ST{}, P0{}, n);
      chunk(ST{}, P1{}, n + 1);
    }
    for (; n < total; n += 2) {
      chunk(TL{}, P0{}, n);
      if (n + 1 < total) chunk(TL{}, P1{}, n + 1);
    }
  }

  NT_WG(3, __builtin_readcyclecounter());            // end of the chunk loop
  float* __restrict__ Cz = Cp + (long)z * p.scz;
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gm = m0 + wave * WM * 16 + i * 16 + kq * 4 + r;
      if (gm >= (XR ? p.M - 1 : p.M)) continue;
#pragma unroll
      for (int j = 0; j < KT; ++j)
#pragma unroll
        for (int q = 0; q < NTC; ++q) {
          const int gc = c0 + q * 16 + nq;
          if (gc < p.Nc) Cz[(long)gm * p.scm + (long)gc * p.scc + (long)j * p.scj] = F16 ? acc[i][j][q][r] * us : acc[i][j][q][r];
        }
    }
  if constexpr (XR) {
    if (xr_on) {                       // row M - 1: a channel's 8 k-groups are 8 neighbouring lanes (slot f = tid + 256 r: kg = f % 8, channel = f / 8)
#pragma unroll
      for (int r = 0; r < NX; ++r) {
        float v = xacc[r];
        v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
        const int f = tid + NTH * r, gc = c0 + f / KG;
        if ((f % KG) == 0 && gc < p.Nc) Cz[(long)(p.M - 1) * p.scm + (long)gc * p.scc] = v;
      }
    }
  }
#ifdef SSV_NT_STAMP
  __builtin_amdgcn_s_waitcnt(0);
#endif
  NT_WG(1, __builtin_amdgcn_s_memrealtime());
}

// Tile plan for the weight gradient.  The output (M x Nc x KT) is small, so the reduction axis (batch x time) is cut into
// Z slabs that are summed afterwards; slab traffic (Z x output, written and read back) competes with the operand reads,
// so smaller tiles with fewer slabs win when the output is small.
void ssv_nt_bf3_tile(int KT, int M, int Nc, int* wm, int* ntc) {
  // measured (round-1 sweep; in-step re-check with SSV_NT_FORCE, tools/prof_env.sh): k=3 -- the largest tile wins at every hot shape; k=1 carries a third of the MFMAs per
  // staged byte, so only large outputs (513 x 513) keep the 128 x 96 tile, smaller ones take 64 x 64 tiles with fewer slabs
  if (KT == 3) { *wm = 2; *ntc = 4; }
  else if (Nc <= 48) { *wm = 2; *ntc = 2; }
  // (round 6) the LSTM's weight gradients (3072 x 768 over 120 frames x 880 utterances): 128 x 128.  The k = 1 kernel is bound by its dH path -- the
  // fragments each wave loads for itself, 32 bytes per lane and 8 time steps: with 64-column tiles (twice the dH bytes per MFMA) the five launches of
  // config 5 take +31 %, with 64-row tiles (twice the staged input per MFMA) +5 %, with 128 columns -14 % (11.2 -> 9.6 ms).  Not for the convolutions'
  // shapes: their M = 513 outputs lose the extra-row form (115 -> 148 us in-step).
  // ... and 256 x 128 on 8 waves (wm = 4 stands for 8 waves x 32 rows; one workgroup per CU) where the rows allow: the staged input tile serves twice the
  // rows and half as many workgroups walk the rows of dH (8.5 -> 7.9 ms)
  else if (M % 256 == 0 && Nc % 128 == 0 && (long)M * Nc >= (1L << 21)) { *wm = 4; *ntc = 8; }
  else if (M % 128 == 0 && Nc % 128 == 0 && (long)M * Nc >= (1L << 21)) { *wm = 2; *ntc = 8; }
  else if ((long)M * Nc >= (1L << 18)) { *wm = 2; *ntc = 6; }
  else { *wm = 1; *ntc = 4; }
}
// co-resident workgroups per CU (4 waves each, one per SIMD): 512 / VGPRs of the instantiation, as compiled for gfx950
// (-Rpass-analysis=kernel-resource-usage: <3,2,4> 236, <3,2,2> 174, <3,1,4> 156, <3,1,2> 110, <1,2,6> 208, <1,2,4> 168, <1,2,2> 134,
// <1,1,6> 148, <1,1,4> 120, <1,1,2> 94; the two LDS buffers of the largest tile (32 KB) allow 4)
int ssv_nt_bf3_wg_per_cu(int KT, int wm, int ntc) {
  if (wm == 4) return 1;                            // 8 waves, 64 KB of LDS, 2 waves per SIMD as the others
  if (KT == 3) return wm == 2 ? 2 : (ntc >= 4 ? 3 : 4);
  if (wm == 2) return ntc >= 6 ? 2 : 3;            // (<1,2,8>: 244 VGPRs, 64 KB of LDS)
  return ntc >= 6 ? 3 : (ntc >= 4 ? 4 : 5);
}
int ssv_nt_bf3_tiles(int KT, int M, int Nc) {
  int wm, ntc;
  ssv_nt_bf3_tile(KT, M, Nc, &wm, &ntc);
  return ssv_cdiv(ssv_nt_bf3_xrow(KT, M, Nc) ? M - 1 : M, 64 * wm) * ssv_cdiv(Nc, 16 * ntc);
}
// the k = 1 weight gradient of 128 j + 1 output rows on the 128 x 96 tile: last row beside the staging, range slabs (gemm_nt_bf3_kernel<.., XR = 1>)
bool ssv_nt_bf3_xrow(int KT, int M, int Nc) {
  int wm, ntc;
  ssv_nt_bf3_tile(KT, M, Nc, &wm, &ntc);
  return SSV_NT_XROW && KT == 1 && wm == 2 && ntc == 6 && M > 128 && M % 128 == 1;
}

// the kernel addresses both operands with 32-bit element offsets
bool ssv_nt_bf3_fits(const GemmNT& g) {
  const long lim = 1L << 30;
  return (long)(g.B - 1) * g.sab + (long)(g.M - 1) * g.sam + g.La < lim && (long)(g.B - 1) * g.sxb + (long)(g.Nc - 1) * g.sxc + g.Lx < lim &&
         (long)g.B * g.sab < lim && (long)g.B * g.sxb < lim && g.La >= 8 && g.Lx >= 8;
}
int ssv_launch_gemm_nt_bf3(const GemmNT& g, hipStream_t st) {
  SSV_CHECK(g.M > 0 && g.Nc > 0 && g.La > 0 && g.B > 0 && g.Z > 0 && g.bstep >= 0, SSV_BAD_SHAPE, "gemm_nt_bf3: empty problem");      // (bstep == 0: range slabs)
  SSV_CHECK(g.KT == 1 || g.KT == 3, SSV_UNSUPPORTED, "gemm_nt_bf3: kernel_size %d", g.KT);
  SSV_CHECK(g.sat == 1 && g.sxn == 1, SSV_UNSUPPORTED, "gemm_nt_bf3: rows must be contiguous in time");
  SSV_CHECK(g.Z <= 65535, SSV_UNSUPPORTED, "gemm_nt_bf3: Z=%d exceeds grid.z", g.Z);
  SSV_CHECK(ssv_nt_bf3_fits(g), SSV_UNSUPPORTED, "gemm_nt_bf3: an operand spans 2^30 elements or more");
  int wm, ntc;
  ssv_nt_bf3_tile(g.KT, g.M, g.Nc, &wm, &ntc);
  const bool xr = ssv_nt_bf3_xrow(g.KT, g.M, g.Nc);       // (then the caller chose range slabs: bstep == 0)
  SSV_CHECK(g.bstep > 0 || xr, SSV_UNSUPPORTED, "gemm_nt_bf3: range slabs are built for the extra-row kernel only");
  const int mtiles = ssv_cdiv(xr ? g.M - 1 : g.M, 64 * wm);
  const int nz = g.jobs ? g.njobs * g.Z : g.Z;
  SSV_CHECK(nz <= 65535, SSV_UNSUPPORTED, "gemm_nt_bf3: %d slabs exceed grid.z", nz);
  const dim3 grid(mtiles * ssv_cdiv(g.Nc, 16 * ntc), 1, nz);
  SSV_CHECK(!g.f16 || g.jobs || (g.a_amax && g.x_amax && g.a_namax > 0 && g.x_namax > 0), SSV_BAD_SHAPE, "gemm_nt_bf3: split-fp16 needs both operand scales");
  // the ring kernel (k = 3): every shift within one block (64 time steps) either way.  With a job table the shifts are on the device: the caller states their bound.
  int ring_ms = -1;
#if SSV_NT_RING
  if (g.KT == 3 && wm == 2 && ntc == 4) {
    ring_ms = g.max_shift;
    if (!g.jobs) { ring_ms = 0; for (int j = 0; j < 3; ++j) ring_ms = abs(g.shift[j]) > ring_ms ? abs(g.shift[j]) : ring_ms; }
    if (ring_ms > 64) ring_ms = -1;
  }
#endif
  if (ssv_shape_log_on()) {
    char nm[96], note[96];
    const int nj = g.jobs ? g.njobs : 1;
    if (ring_ms >= 0) snprintf(nm, sizeof nm, "gemm_nt3r_kernel<%d, %d>", wm, g.f16);
    else if (wm == 4) snprintf(nm, sizeof nm, "gemm_nt_bf3_kernel<%d, 2, %d, %d, 0, 8>", g.KT, ntc, g.f16);
    else snprintf(nm, sizeof nm, "gemm_nt_bf3_kernel<%d, %d, %d, %d, %d, 4>", g.KT, wm, ntc, g.f16, xr ? 1 : 0);       // (as rocprofv3 prints the instantiation)
    snprintf(note, sizeof note, "jobs=%d B=%d M=%d Nc=%d L=%d k=%d Z=%d", nj, g.B, g.M, g.Nc, g.La, g.KT, g.Z);
    ssv_shape_log(nm, grid, dim3(wm == 4 ? 512 : 256), 2.0 * nj * g.B * g.M * g.Nc * g.La * g.KT,
                  4.0 * nj * ((double)g.B * g.M * g.La + (double)g.B * g.Nc * g.Lx + (double)g.Z * g.M * g.Nc * g.KT), note);
  }
  if (ring_ms >= 0) {
    const int tchunks = ssv_cdiv((g.La > g.Lx ? g.La : g.Lx) + ring_ms, 64);
    return ssv_launch_gemm_nt3r(g, grid, mtiles, tchunks, ring_ms, st);
  }
  if (xr) {
    if (g.f16) hipLaunchKernelGGL((gemm_nt_bf3_kernel<1, 2, 6, 1, 1>), grid, dim3(256), 0, st, g, mtiles);
    else hipLaunchKernelGGL((gemm_nt_bf3_kernel<1, 2, 6, 0, 1>), grid, dim3(256), 0, st, g, mtiles);
    return ssv_check_launch("gemm_nt_bf3 (extra row)");
  }
  if (g.KT == 1 && wm == 4 && ntc == 8) {
    if (g.f16) hipLaunchKernelGGL((gemm_nt_bf3_kernel<1, 2, 8, 1, 0, 8>), grid, dim3(512), 0, st, g, mtiles);
    else hipLaunchKernelGGL((gemm_nt_bf3_kernel<1, 2, 8, 0, 0, 8>), grid, dim3(512), 0, st, g, mtiles);
    return ssv_check_launch("gemm_nt_bf3 (8 waves)");
  }
#define SSV_NT(K_, A_, C_) if (g.KT == K_ && wm == A_ && ntc == C_) { \
    if (g.f16) hipLaunchKernelGGL((gemm_nt_bf3_kernel<K_, A_, C_, 1>), grid, dim3(256), 0, st, g, mtiles); \
    else hipLaunchKernelGGL((gemm_nt_bf3_kernel<K_, A_, C_, 0>), grid, dim3(256), 0, st, g, mtiles); \
    return ssv_check_launch("gemm_nt_bf3"); }
  SSV_NT(3, 2, 4) SSV_NT(3, 2, 2) SSV_NT(3, 1, 4) SSV_NT(3, 1, 2)
  SSV_NT(1, 2, 8) SSV_NT(1, 2, 6) SSV_NT(1, 2, 4) SSV_NT(1, 2, 2) SSV_NT(1, 1, 6) SSV_NT(1, 1, 4) SSV_NT(1, 1, 2)
#undef SSV_NT
  return ssv_fail(SSV_UNSUPPORTED, "gemm_nt_bf3: no tile %d,%d for kernel size %d", wm, ntc, g.KT);
}

