// Conv1d weight gradient, k = 3, on split MFMAs with the input staged ONCE per chunk into a ring of time-major LDS rows (gemm_nt3r_kernel).
#include "bf3_common.h"

#ifdef SSV_NT_STAMP
// (tuning builds: the ring kernel's own stamp buffers -- ssv_debug_nt3r_stamps / ssv_debug_nt3r_wg, tools/nt3r_stamps.py)
__device__ unsigned long long ssv_nt3r_stamps[64];
__device__ unsigned long long ssv_nt3r_wg[4096 * 4];
extern "C" int ssv_debug_nt3r_wg(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ssv_nt3r_wg), sizeof(ssv_nt3r_wg)); }
extern "C" int ssv_debug_nt3r_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ssv_nt3r_stamps), sizeof(ssv_nt3r_stamps)); }
#define NT_WG(k, v) do { const unsigned w_ = blockIdx.z * gridDim.x + blockIdx.x; if (threadIdx.x == 0 && w_ < 4096u) ssv_nt3r_wg[w_ * 4 + (k)] = (v); } while (0)
#define NT_STAMP(k) do { if (stamp_on && (unsigned)(stamp_s - 24) < 8u) ssv_nt3r_stamps[(stamp_s - 24) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define NT_STAMP(k) do {} while (0)
#define NT_WG(k, v) do {} while (0)
#endif
// ---- NT, k = 3, input staged ONCE: a ring of time-major rows read back by transposed LDS reads (round 4) --------------------------
// The kernel above re-stages the input tile for every tap (the reduction runs over time, so a tap's dilation shift is a shift along
// k): three splits, three sets of LDS stores and three barriers per 64-step chunk for the same 64 x 64 values.  Here the tile lives in
// LDS TIME-major -- row = time step, 64 channels x fp16 = 128 bytes per row, a hi plane and a lo plane -- and the MFMA B operand
// (8 consecutive time steps of one channel per lane) comes out of `ds_read_b64_tr_b16`, the gfx950 transposed read: a 16-lane group
// names four ROWS (any four: lanes 4q .. 4q+3 carry row q's address) and lane i receives column i of them.  A tap's shift is then
// only a different row address, so
//   * each input value is loaded, split and stored ONCE: the rows form a ring of four 64-row blocks (block k = the 64 time steps of
//     chunk k, ring slot k & 3); chunk n multiplies rows 64n + shift_j + [0, 64), i.e. blocks n-1 .. n+1, while block n+2 is being
//     written -- one unit of 4 channels x 4 time steps per thread and chunk (four 16-byte loads along time, eight 8-byte LDS stores)
//     against three windows of 16 before, and ONE barrier per chunk;
//   * the chunks of one slab form one virtual time line: the items follow each other in `tchunks` chunks each, with tchunks * 64 >=
//     L + the largest |shift|, so the rows a tap reaches beyond an item's ends are rows of that item's own padding region or of the
//     neighbour's -- staged as zeros (t >= Lx is masked) or met by dH values that are zero (t >= La is masked in the dH split of an
//     item's last chunk; what such a zero multiplies is input data of the tensor itself, i.e. finite).  No per-tap edge cases;
//   * rows 256 .. 287 mirror rows 0 .. 31 (written with block 4i's first rows), so a k-step (32 rows) that starts in the last rows of
//     the ring runs on linearly: every read address is lane constant + a per-(chunk, tap, k-step) SCALAR.
// The dH side: a lane's fragment of a 32-step k-step is two 16-byte tuples, steps 4 kq .. 4 kq + 3 and 16 + 4 kq .. 19 + 4 kq (so that one load
// instruction touches 64 CONTIGUOUS bytes of each of its 16 rows, not four 16-byte pieces spread over 128: -1.3 % in-step), split in place pair by
// pair: dword q of the fragment = steps (4 kq + q, 16 + 4 kq + q).  Any order is right as long as the input side uses it: read 0 of a
// fragment names rows (0, 16, 1, 17) + 4 kq of the k-step, read 1 rows (2, 18, 3, 19) + 4 kq.
// Bank conflicts (cdna_hip_programming.md section 2: 64 dword banks per 32-lane half for the transposed read): a half reads 8 row
// pieces of 32 bytes -- rows u + {0,1,4,5,16,17,20,21} (+2 for the second read) of one 16-channel block.  With 128-byte rows these would
// share two 32-byte bank groups; the 32-byte piece `cb` of row r is therefore stored at piece cb ^ (bit 2 of r | bit 4 of r << 1): the 8 pieces
// of a half land on 8 distinct bank groups for every u (checked exhaustively; no linear row stride does that for the rows a split-in-place
// fragment names, tools/tr_layout_search.py).
typedef short ssv_s4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) ssv_s4 ssv_lds_s4;
typedef unsigned ssv_u2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) ssv_u2 ssv_lds_u2;
// LDS accesses by 32-bit LDS byte address (the ring kernel computes addresses, not indices)
__device__ __forceinline__ void ssv_lds_store2(unsigned addr, unsigned a, unsigned b) {
#if defined(__HIP_DEVICE_COMPILE__)
  *(ssv_lds_u2*)(size_t)addr = (ssv_u2){a, b};
#else
  (void)addr; (void)a; (void)b;
#endif
}
__device__ __forceinline__ uint2 ssv_lds_read_tr16(unsigned addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((ssv_lds_s4*)(size_t)addr));
#else
  (void)addr; return make_uint2(0u, 0u);
#endif
}
template <int WM, int F16>
__global__ __launch_bounds__(256, 2) void gemm_nt3r_kernel(const GemmNT p, const int mtiles, const int tchunks, const int max_shift) {
  constexpr int KT = 3, NTC = 4, NCH = 64, KB = 64, KS = 2;
  constexpr int R_FD = 1;                                   // fragment groups read ahead of the MFMAs (2: no change, 8 more registers)
  constexpr unsigned ROWB = 128, PLANE = 288 * ROWB;            // bytes per row / per plane (256 ring rows + 32 mirror rows)
  __shared__ __attribute__((aligned(16))) unsigned char ring[2 * PLANE];
  __shared__ float amax_sm[8];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  NT_WG(0, __builtin_amdgcn_s_memrealtime()); NT_WG(2, __builtin_readcyclecounter());
  const unsigned wg = ssv_xcd_order(blockIdx.x + gridDim.x * blockIdx.z, gridDim.x * gridDim.z);
  const int bxx = (int)(wg % gridDim.x);
  int z = (int)(wg / gridDim.x);
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
    const ssv_wgrad_job* __restrict__ jb = p.jobs + job;
    Ap = ssv_global(jb->dy); Xp = ssv_global(jb->x); Cp = p.C + (long)job * p.Z * p.scz;
    shj[0] = jb->shift[0]; shj[1] = jb->shift[1]; shj[2] = jb->shift[2];
    a_amax = ssv_global(jb->dy_amax); x_amax = ssv_global(jb->x_amax); a_namax = jb->dy_namax; x_namax = jb->x_namax;
  }
  float as = 1.f, xs = 1.f, us = 1.f;
  if constexpr (F16) {
    float sa, sx, ia, ix;
    ssv_pow2_scale(ssv_list_max<4>(a_amax, a_namax, amax_sm), sa, ia);
    ssv_pow2_scale(ssv_list_max<4>(x_amax, x_namax, amax_sm + 4), sx, ix);
    as = ssv_uniform(sa); xs = ssv_uniform(sx);
    us = ssv_uniform(ia * ix);
  }
  const int mt = bxx % mtiles, ct = bxx / mtiles;
  const int m0 = mt * 64 * WM, c0 = ct * NCH;
  const int kq = lane >> 4, nq = lane & 15;

  f32x4 acc[WM][KT][NTC];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < KT; ++j)
#pragma unroll
      for (int q = 0; q < NTC; ++q) acc[i][j][q] = (f32x4){0.f, 0.f, 0.f, 0.f};

  uint4 AH[2][WM][KS], AL[2][WM][KS];                       // dH fragments, as in gemm_nt_bf3_kernel: two sets, loaded raw, split in place
  uint4 rx[4];                                              // the thread's unit of the block in flight: channel ci, 4 time steps

  // both operands are read by buffer loads whose range is the tensor: the chunks beyond the slab's last (the look-ahead never tests
  // for them) and what lies past a tensor's end read as zeros or as other entries of the same tensor, never outside it.  The whole
  // offset travels in the VGPR: the hardware's range check covers the vector offset only, a scalar offset is added after it
  // (seen as NaN gradients: look-ahead blocks read with the batch item in the scalar offset came from beyond the tensor)
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Ap), 0,
      (int)(((long)(p.B - 1) * p.sab + (long)(p.M - 1) * p.sam + p.La) * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(Xp), 0,
      (int)(((long)(p.B - 1) * p.sxb + (long)(p.Nc - 1) * p.sxc + p.Lx) * 4), 0x00020000);
  // (rows / channels past a ragged tile's end are not clamped: they read other rows of the tensor or zeros, and feed accumulators nobody stores)
  const unsigned arow = (unsigned)(((m0 + wave * WM * 16 + nq) * (int)p.sam + 4 * kq) * 4);
  const unsigned a16 = 16u * (unsigned)p.sam * 4u;            // one 16-row block further (scalar)
  // staging: wave w owns the 16-channel piece w of every row; lane = cq + 4 tq: channels 16 w + 4 cq .. + 3, rows 4 tq .. 4 tq + 3 of the block
  const int cq = lane >> 4, tq = lane & 15;               // 4 consecutive lanes = 64 contiguous bytes of one channel row
  const unsigned xrow = (unsigned)(((c0 + 16 * wave + 4 * cq) * (int)p.sxc + 4 * tq) * 4);
  const unsigned x1 = (unsigned)p.sxc * 4u;                   // one channel further (scalar)
  const bool rows_in_c = c0 + NCH <= p.Nc;
  const unsigned ringb = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ring;
  const unsigned waddr = ringb + (unsigned)(4 * tq) * ROWB + (unsigned)((wave ^ ((tq & 1) | (((tq >> 2) & 1) << 1))) << 5) + (unsigned)cq * 8u;
  // fragment reads: lane (kq, 4 q + pp) of read r names row 8 kq + rsel[r][q]; the piece swizzle of that row depends on the tap's shift
  const int q4 = (lane >> 2) & 3, pp = lane & 3;
  int lrow[2];
  lrow[0] = 4 * kq + (q4 >> 1) + 16 * (q4 & 1);              // q = 0..3 -> rows 0, 16, 1, 17 of the k-step (+ 4 kq)
  lrow[1] = lrow[0] + 2;
  unsigned lbase[2], key5[KT][2];
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    lbase[r] = ringb + (unsigned)lrow[r] * ROWB + (unsigned)pp * 8u;
#pragma unroll
    for (int j = 0; j < KT; ++j) { const int rr = shj[j] + lrow[r]; key5[j][r] = (unsigned)(((rr >> 2) & 1) | (((rr >> 4) & 1) << 1)) << 5; }
  }

  const int nb = (p.B - z + p.bstep - 1) / p.bstep;
  const int total = nb * tchunks;
  int cb[4], ct0[4];
  cb[0] = z; ct0[0] = 0;
  auto next_chunk = [&](int b, int t0, int& nb_, int& nt0) __attribute__((always_inline)) {
    nt0 = t0 + KB; nb_ = b;
    if (nt0 >= tchunks * KB) { nt0 = 0; nb_ = b + p.bstep; }
  };
#pragma unroll
  for (int k = 1; k < 4; ++k) next_chunk(cb[k - 1], ct0[k - 1], cb[k], ct0[k]);

  auto loadA = [&](auto set, int b, int t0) __attribute__((always_inline)) {
    constexpr int SET = decltype(set)::value;
    const unsigned so = ((unsigned)b * (unsigned)p.sab + (unsigned)t0) * 4u;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) {
        AH[SET][i][s2] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)(arow + (so + i * a16)) + s2 * 128, 0, 0));
        AL[SET][i][s2] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)(arow + (so + i * a16)) + s2 * 128 + 64, 0, 0));
      }
  };
  // one of the chunk's 4 WM dH loads / 4 input loads: the chunk loop issues them one at a time between its MFMA groups
  auto loadA1 = [&](auto set, int b, int t0, int idx) __attribute__((always_inline)) {
    constexpr int SET = decltype(set)::value;
    const unsigned so = ((unsigned)b * (unsigned)p.sab + (unsigned)t0) * 4u;
    const int i = idx / (2 * KS), s2 = (idx / 2) % KS, hf = idx % 2;
    const uint4 v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)(arow + (so + i * a16)) + s2 * 128 + 64 * hf, 0, 0));
    if (hf) AL[SET][i][s2] = v; else AH[SET][i][s2] = v;
  };
  auto loadX1 = [&](int b, int t0, int ci) __attribute__((always_inline)) {
    const unsigned so = ((unsigned)b * (unsigned)p.sxb + (unsigned)t0) * 4u;
    rx[ci] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)(xrow + (so + ci * x1)), 0, 0));
  };
  auto splitA = [&](auto set, int t0) __attribute__((always_inline)) {
    constexpr int SET = decltype(set)::value;
    const bool ragged = t0 + KB > p.La;                      // the item's last chunk(s): time steps at or past the row length count as zeros
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) {
        float v[8];
        const uint4 t0_ = AH[SET][i][s2], t1_ = AL[SET][i][s2];
        v[0] = __builtin_bit_cast(float, t0_.x); v[1] = __builtin_bit_cast(float, t0_.y); v[2] = __builtin_bit_cast(float, t0_.z); v[3] = __builtin_bit_cast(float, t0_.w);
        v[4] = __builtin_bit_cast(float, t1_.x); v[5] = __builtin_bit_cast(float, t1_.y); v[6] = __builtin_bit_cast(float, t1_.z); v[7] = __builtin_bit_cast(float, t1_.w);
        if (ragged) {
          const int nv = p.La - (t0 + 32 * s2 + 4 * kq);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (e >= 4 ? e + 12 : e) < nv ? v[e] : 0.f;      // (the second tuple holds steps 16 .. 19 of the lane's window)
        }
#if SSV_NT3R_ABL & 1
        (void)v;                                              // ablation build: the raw fp32 bits stand in for the hi / lo fragments (garbage values, same loads)
#else
        split8p<F16>(v, as, AH[SET][i][s2], AL[SET][i][s2]);
#endif
      }
  };
  auto loadX = [&](int b, int t0) __attribute__((always_inline)) {
    const unsigned so = ((unsigned)b * (unsigned)p.sxb + (unsigned)t0) * 4u;
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) rx[ci] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsX, (int)(xrow + (so + ci * x1)), 0, 0));
  };
  auto commitX = [&](int slot, int t0) __attribute__((always_inline)) {
    float v[4][4];
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) {
      v[ci][0] = __builtin_bit_cast(float, rx[ci].x); v[ci][1] = __builtin_bit_cast(float, rx[ci].y);
      v[ci][2] = __builtin_bit_cast(float, rx[ci].z); v[ci][3] = __builtin_bit_cast(float, rx[ci].w);
    }
    if (!(rows_in_c && t0 + KB <= p.Lx)) {                   // (wave-uniform) ragged channel tile, or the block reaches the row end
      const int nv = p.Lx - (t0 + 4 * tq);
#pragma unroll
      for (int ci = 0; ci < 4; ++ci) {
        const bool cok = c0 + 16 * wave + 4 * cq + ci < p.Nc;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[ci][i] = (cok && i < nv) ? v[ci][i] : 0.f;
      }
    }
    const unsigned a = waddr + (unsigned)slot * (64u * ROWB);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned h0, l0, h1, l1;
#if SSV_NT3R_ABL & 2
      h0 = __builtin_bit_cast(unsigned, v[0][i]); l0 = __builtin_bit_cast(unsigned, v[1][i]);      // ablation build: no split, same LDS stores
      h1 = __builtin_bit_cast(unsigned, v[2][i]); l1 = __builtin_bit_cast(unsigned, v[3][i]);
#else
      split_pair<F16>(v[0][i], v[1][i], xs, h0, l0);
      split_pair<F16>(v[2][i], v[3][i], xs, h1, l1);
#endif
      ssv_lds_store2(a + i * ROWB, h0, h1);
      ssv_lds_store2(a + i * ROWB + PLANE, l0, l1);
      if (slot == 0 && tq < 8) {                             // rows 0 .. 31 again as rows 256 .. 287
        ssv_lds_store2(a + i * ROWB + 256u * ROWB, h0, h1);
        ssv_lds_store2(a + i * ROWB + 256u * ROWB + PLANE, l0, l1);
      }
    }
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;

  // chunk n: MFMAs of the three taps on ring blocks n-1 .. n+1 | block n+2 (loaded during chunk n-1) is split and stored after the first tap |
  // the 4 loads of block n+3 go into the registers just freed, one per MFMA group of the second tap | dH of chunk n+1: its 8 loads one per
  // group of the first tap, split in place after the last tap
  auto chunk = [&](auto par, int n) __attribute__((always_inline)) {
    using CUR = std::integral_constant<int, decltype(par)::value>;
    using NXT = std::integral_constant<int, decltype(par)::value ^ 1>;
#ifdef SSV_NT_STAMP
    const bool stamp_on = blockIdx.x == 0 && blockIdx.z == 0 && tid == 0;
    const int stamp_s = n + 16;                                  // chunks 8 .. 15 (NT_STAMP records "steps" 24 .. 31)
#endif
    NT_STAMP(0);
#pragma unroll
    for (int j = 0; j < KT; ++j) {
      constexpr int G = KS * NTC;
      uint4 fb[R_FD + 1][2];
      unsigned su[KS];
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) su[s2] = (unsigned)((64 * n + shj[j] + 32 * s2) & 255) * ROWB;         // scalar
      auto frag = [&](int g, uint4 (&f)[2]) __attribute__((always_inline)) {
        const int s2 = g / NTC, cbk = g % NTC;
        const unsigned a0 = lbase[0] + su[s2] + ((unsigned)(cbk << 5) ^ key5[j][0]);
        const unsigned a1 = lbase[1] + su[s2] + ((unsigned)(cbk << 5) ^ key5[j][1]);
        const uint2 h0u = ssv_lds_read_tr16(a0), h1u = ssv_lds_read_tr16(a1);
        const uint2 l0u = ssv_lds_read_tr16(a0 + PLANE), l1u = ssv_lds_read_tr16(a1 + PLANE);
        f[0] = make_uint4(h0u.x, h0u.y, h1u.x, h1u.y);
        f[1] = make_uint4(l0u.x, l0u.y, l1u.x, l1u.y);
      };
#pragma unroll
      for (int g = 0; g < R_FD; ++g) frag(g, fb[g]);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (g + R_FD < G) frag(g + R_FD, fb[(g + R_FD) % (R_FD + 1)]);
        // one vector-memory instruction per group: among MFMAs its issue costs the wave less than in a batch (-2.7 % on the launch)
        if (j == 0 && g < 4 * WM) loadA1(NXT{}, cb[1], ct0[1], g);
        if (j == 1 && g < 4) loadX1(cb[3], ct0[3], g);
        __builtin_amdgcn_sched_barrier(0);
        const int s2 = g / NTC, q = g % NTC;
        if (ct0[0] + 32 * s2 >= p.La) continue;                  // (wave-uniform) k-steps at or past the row length: dH is zero there
        const uint4 bh = fb[g % (R_FD + 1)][0];
        const uint4 bl = fb[g % (R_FD + 1)][1];
#pragma unroll
        for (int i = 0; i < WM; ++i) {
          const uint4 a_h = AH[CUR::value][i][s2];
          const uint4 a_l = AL[CUR::value][i][s2];
          acc[i][j][q] = mma16<F16>(a_l, bh, acc[i][j][q]);
          acc[i][j][q] = mma16<F16>(a_h, bl, acc[i][j][q]);
          acc[i][j][q] = mma16<F16>(a_h, bh, acc[i][j][q]);
        }
      }
      if (j == 0) {
        NT_STAMP(2);
        commitX((n + 2) & 3, ct0[2]);
        NT_STAMP(3);
      }
      if (j == 1) NT_STAMP(5);
      if (j == KT - 1) {
        NT_STAMP(6);
        splitA(NXT{}, ct0[1]);
        NT_STAMP(7);
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) { cb[k] = cb[k + 1]; ct0[k] = ct0[k + 1]; }
    next_chunk(cb[2], ct0[2], cb[3], ct0[3]);
  };

  if (total > 0) {
    // prologue: ring block -1 (slot 3: the rows before the slab's first item) zeroed, blocks 0 and 1 staged, block 2 in flight, dH of chunk 0 split
    loadA(P0{}, cb[0], ct0[0]);
    loadX(cb[0], ct0[0]);
    {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        ssv_lds_store2(waddr + 3u * 64u * ROWB + i * ROWB, 0u, 0u);
        ssv_lds_store2(waddr + 3u * 64u * ROWB + i * ROWB + PLANE, 0u, 0u);
      }
    }
    commitX(0, ct0[0]);
    loadX(cb[1], ct0[1]);
    splitA(P0{}, ct0[0]);
    commitX(1, ct0[1]);
    loadX(cb[2], ct0[2]);
    __syncthreads();
    int n = 0;
    for (; n + 1 < total; n += 2) {
      chunk(P0{}, n);
      chunk(P1{}, n + 1);
    }
    if (n < total) chunk(P0{}, n);
  }

  NT_WG(3, __builtin_readcyclecounter());
  // a job whose shifts exceed the bound the launch was planned for (the caller's max_shift) must not pass for a result
  const bool bad_shift = abs(shj[0]) > max_shift || abs(shj[1]) > max_shift || abs(shj[2]) > max_shift;
  float* __restrict__ Cz = Cp + (long)z * p.scz;
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
          if (gc < p.Nc) Cz[(long)gm * p.scm + (long)gc * p.scc + (long)j * p.scj] = bad_shift ? __builtin_nanf("") : (F16 ? acc[i][j][q][r] * us : acc[i][j][q][r]);
        }
    }
#ifdef SSV_NT_STAMP
  __builtin_amdgcn_s_waitcnt(0);
#endif
  NT_WG(1, __builtin_amdgcn_s_memrealtime());
}


// launched by ssv_launch_gemm_nt_bf3 (wgrad_nt.hip), which owns the tile / slab plan and the argument checks
int ssv_launch_gemm_nt3r(const GemmNT& g, dim3 grid, int mtiles, int tchunks, int ring_ms, hipStream_t st) {
  if (g.f16) hipLaunchKernelGGL((gemm_nt3r_kernel<2, 1>), grid, dim3(256), 0, st, g, mtiles, tchunks, ring_ms);
  else hipLaunchKernelGGL((gemm_nt3r_kernel<2, 0>), grid, dim3(256), 0, st, g, mtiles, tchunks, ring_ms);
  return ssv_check_launch("gemm_nt3r");
}
