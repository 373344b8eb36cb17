// Shared declarations for the libssv_hip kernels (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/ssv_hip.h"

// ---- error reporting (thread-local, see ssv_last_error in api.hip) -------------------
int ssv_fail(int code, const char* fmt, ...);
#define SSV_BAD_SHAPE (-1)
#define SSV_UNSUPPORTED (-2)
#define SSV_CHECK(cond, code, ...) do { if (!(cond)) return ssv_fail((code), __VA_ARGS__); } while (0)
#define SSV_TRY(expr) do { int _e = (expr); if (_e) return _e; } while (0)
int ssv_check_launch(const char* what);   // hipGetLastError() -> 0 or -(int)hipError_t

typedef __attribute__((ext_vector_type(4))) float f32x4;

// ---- implicit GEMM, "NN" form --------------------------------------------------------
//   C(b,m,n) = bias[m] + bias_b[b][m] + R(b,m,n) + sum_{c<Kc} sum_{j<KT} A(b,m,c,j) * X(b,c,n+shift[j])
// with X(b,c,col) = 0 outside 0 <= col < Lx.  Used for: dilated/causal Conv1d forward,
// its data gradient (transposed weights, negated shifts), 1x1 convs, the two halves of
// ConvTranspose1d(k=2,s=2), attention V*A / K*dS / K^T*Q and the LSTM projections.
struct GemmNN {
  const float* A; long sab, sam, sac, saj;
  const float* X; long sxb, sxc, sxn; int Lx;
  float* C; long scb, scm, scn;
  const float* bias;
  const float* bias_b; long sbb;
  const float* R; long srb, srm, srn;
  int M, N, Kc, KT, B;
  int shift[3];
  float alpha;               // scales the accumulated product only (attention 1/sqrt(d))
};
int ssv_launch_gemm_nn(const GemmNN& g, hipStream_t st);

// ---- implicit GEMM, "NT" form (reduction over time) ----------------------------------
//   C(z,m,c,j) = sum_{b = z, z+bstep, ... < B} sum_{t<La} A(b,m,t) * X(b,c,t+shift[j])
// Used for: Conv1d weight gradient (split over batches into slabs), ConvTranspose1d
// weight gradient, attention dV = dR*A^T and dK = Q*dS^T (one output per batch).
struct GemmNT {
  const float* A; long sab, sam, sat; int La;
  const float* X; long sxb, sxc, sxn; int Lx;
  float* C; long scz, scm, scc, scj;
  int M, Nc, KT, B, Z, bstep;   // bstep > 0: slab z reduces over batch items z, z + bstep, ...; bstep == 0 (gemm_nt_bf3's extra-row kernel): over the z-th of Z equal chunk ranges
  int shift[3];
  // several problems of one shape in one launch (split-bf16 kernel only): grid.z = njobs * Z, entry z belongs to job z / Z and
  // takes A = jobs[job].dy, X = jobs[job].x, the job's shifts, and slab (z % Z) of the job's slab region C + job * Z * scz
  const ssv_wgrad_job* jobs; int njobs;
  int max_shift;           // with a job table: the caller's bound on |shift| over all jobs (picks the k = 3 ring kernel when <= 64); < 0 = unknown
  // split-fp16 arithmetic (split-bf16 kernel only): both operands are scaled while they are split, by the power of two that
  // the maximum over ALL entries of their partial-maxima lists gives (the reduction runs over the batch, so one scale per
  // tensor); with a job table the lists are the job's (dy_amax / x_amax, n_amax entries each side as stored in the job).
  int f16; const float* a_amax; int a_namax; const float* x_amax; int x_namax;
};
int ssv_launch_gemm_nt(const GemmNT& g, hipStream_t st);

// ---- split-MFMA variants (conv_nn.hip, wgrad_nt.hip, wgrad_nt3r.hip, pwln.hip; helpers in bf3_common.h): same contracts, unit column strides, weights pre-split --------------
struct GemmNNB {
  const unsigned short* Ahi; const unsigned short* Alo; int Kpad;   // bf16 planes (hi, lo) in fragment order, see pack_split_kernel
  const float* X; long sxb, sxc; int Lx;
  float* C; long scb, scm;
  const float* bias;
  const float* bias_b; long sbb;
  const float* R; long srb, srm;
  int M, N, Kc, KT, B;
  int shift[3];
  int sxn, scn;            // column strides of X and C (1 everywhere except the stride-2 deconvolution halves; R needs scn == 1)
  // row_pair = 1 (round 6; the transposed convolution's forward in ONE product over 2 Cout rows): output row m, column n goes to
  // C(b, m >> 1, 2 n + (m & 1)) -- scm is the stride of those M / 2 rows of 2 N floats -- and bias is indexed by m >> 1.  Plain k = 1 products only
  // (M even, scn == 1, no R / LSTM / statistics).  c_amax != null: the launch also leaves the output's operand-scale list, c_namax entries per
  // item at c_amax + b * c_namax: entry = the workgroup's tile (max |C| over it), the entries past the tiles zeroed by the item's last tile.
  int row_pair; float* c_amax; int c_namax;
  // LSTM wavefront at inference with PRE-SPLIT recurrent activations (round 6; hs_planes != null, split-fp16 mode, lstm_D == 2): |h| < 1, so its
  // operand scale is the constant 2^14 and the cell epilogue can write h already split -- fp16 hi / lo planes in the consumer's own staging order,
  // [k-group of 8 units][column][8 halves] over hs_npad columns (a multiple of the column tile; the pad columns stay zero) -- for (layer, ring slot) at
  // hs_planes + ((layer * 2 + slot) * 2 + {0: hi, 1: lo}) * hs_plane_bytes.  The products then stage their input with 16-byte loads and no VALU split.
  unsigned short* hs_planes; long hs_plane_bytes; int hs_npad;
  int hs_keep_h;            // (with hs_planes) 1: this launch also stores the fp32 h -- only the wavefront's last step needs it, for the caller's h_last
  // ksplit > 1 (round 6; plain k = 1 products whose K is long and whose output is small -- the LSTM backward's [W_ih | W_hh]^T dgates): grid.y = B * ksplit,
  // entry (b, z) reduces over input rows [z Kc, (z + 1) Kc) against weight chunks [z Kc / 32, ...) of planes whose row length is Kpad = ksplit * Kc and
  // writes its partial product at C + b scb + z scz (the consumer adds the slabs).  skip_rows: entry b = 0 has no use for its row tiles below this row.
  int ksplit; long scz; int skip_rows;
  // x0_planes (with hs_planes, layer 0 riding along): layer 0's input frames pre-split as the recurrent activations are -- frame t at x0_planes + t * 2 *
  // hs_plane_bytes, the first 4 * xsplit0 k-groups of the hi and of the lo plane, split with the frames' own power-of-two scale (x0_amax: 64 partial maxima) --
  // so that W_ih x_t is the FIRST K segment (xsplit0 chunks, an even number) of layer 0's product instead of a projection of all frames written to memory
  // and read back through R: A0hi / A0lo are then the planes of [W_ih | W_hh] (K = 32 xsplit0 + H).  The accumulators are rescaled by 2^14 / (the frames'
  // scale) between the segments, exactly (powers of two): h's scale is the constant 2^14.
  const unsigned short* x0_planes; const float* x0_amax; int xsplit0;
  // LSTM support.  perm_h = H > 0: output row m is gate (m % 4) of hidden unit (m / 4), i.e. row (m % 4) * H + m / 4 of
  // the torch layout -- the weights were packed in that order and the bias vectors are indexed through the same map.
  // epi = 1: fused cell epilogue -- the 4 accumulator rows of a lane are the (i, f, g, o) pre-activations of one unit;
  // the kernel adds R (the input projection) and the biases, updates cstate [H][N] in place and writes h to C [H][N].
  int perm_h, epi, first;
  float* cstate;
  // LSTM wavefront (epi == 1 with lstm_D > 0): grid.y entry b is layer lstm_lo + b at frame t = lstm_s - layer.  The K axis
  // has two segments: chunks [0, xsplit) read the lower layer's h_t, chunks [xsplit, Kpad/32) the layer's own h_{t-1}
  // (skipped at t = 0).  All h live in lstm_out[layer][slot = frame % lstm_D][H][N]; the kernel derives X, X2 and C from
  // (layer, t), and offsets the weight planes by b*sab, the biases by b*sbb and cstate by layer*H*N.
  float* lstm_out; int lstm_s, lstm_lo, lstm_D, xsplit; long sab;
  // Layer 0 in the same launch (round 5; A0hi != null, lstm_lo == 0): entry 0 is layer 0 at frame lstm_s -- its K axis is the layer's own
  // h_{t-1} alone (xsplit chunks of the planes A0hi / A0lo; none at t = 0), its input projection comes in through R (no batch stride), and
  // the entries b >= 1 are layers 1 .. with planes Ahi + (b - 1) * sab.  One launch per wavefront step instead of two.
  const unsigned short* A0hi; const unsigned short* A0lo;
  // training (gates_out != null, lstm_D = number of frames): the activated gates i, f, g, o are saved as
  // gates_out[layer][frame][gate*H + u][N] (torch row order) and cstate is [layer][frame][H][N] (c_{t-1} read, c_t written)
  float* gates_out;
  // split-fp16 arithmetic (f16 = 1, see "split-fp16" below; with the LSTM epilogue one weight scale serves all layers of a launch and
  // the activations, |h| < 1, take the fixed scale 2^14 from a one-entry list holding 1.0): the planes hold fp16
  // hi / lo of A * 2^ea and *a_inv = 2^-ea (written by the pack kernels); X is scaled by 2^ex while it is split, with ex from
  // the maximum of the x_namax partial maxima |X| at x_amax + b * x_amax_bs (x_amax_bs = 0: one list for every batch item).
  int f16; const float* a_inv; const float* x_amax; int x_namax; long x_amax_bs;
  // Column statistics of the OUTPUT for the LayerNorm that follows (highwayConv: the reduction over channels runs across the
  // GEMM's M axis, i.e. across workgroups; this makes the LayerNorm / gate forward a reduction-free streaming kernel):
  // colstats[((b * (M / 64) + m / 64) * N + n) * 2 + {0, 1}] = mean and sum of squared deviations of C(b, 64-row group, n)
  // over the group's 64 rows, bias included.  Needs M % 64 == 0, unit column stride, no LSTM epilogue.  Null: not wanted.
  float* colstats;
  // One output row beyond the last full 128-row tile, kept out of the MFMA tiles (gemm_nn_bf3w_kernel<.., XR = 1>, k = 1): M = 128 j + 1 output
  // rows -- the 513-channel layers of SSRN -- cost a whole extra row tile of MFMAs for ONE row otherwise.  xrow_w[k * xrow_sk], k < Kc, is that
  // row of the fp32 weight; the kernel launches over M - 1 rows and the workgroups of row tile 0 add the row as plain fp32 dot products
  // from the values they stage anyway.  Null: not used.
  const float* xrow_w; long xrow_sk;
};
int ssv_launch_gemm_nn_bf3(const GemmNNB& g, hipStream_t st);
// 1x1 product (g: KT = 1, unit strides, no residual / LSTM epilogue, M <= 640) that finishes LayerNorm over its M rows and the activation in
// the same launch: g.C receives the LayerNorm's input `pre`; y (B, M, N) dense, stats (B, 2, N) or null, y_amax: namax entries per item or null
int ssv_launch_gemm_pwln(const GemmNNB& g, const float* gamma, const float* beta, float* y, long ybs, float* stats, float* y_amax, int namax, int act,
                         hipStream_t st);
int ssv_launch_gemm_nt_bf3(const GemmNT& g, hipStream_t st);
int ssv_launch_gemm_nt3r(const GemmNT& g, dim3 grid, int mtiles, int tchunks, int ring_ms, hipStream_t st);   // (wgrad_nt3r.hip; called by the line above)
// backward of a 1x1 conv + LayerNorm link in one launch (pwln.hip, pwln_bwd_kernel): dy, pre (B, M, L) with M = Cout LN rows (pre dense),
// stats (B, 2, L); dpre (B, M, L) dense, this tile's partial row [dgamma | dbeta | dbias] at part_q * tile of the item's part_rows rows, scale
// list entry 4 * tile; Ahi / Alo: the TRANSPOSED weight's planes (rows = Cin, K = M); dx (B, Cin, L); xrow_w[o * xrow_sk]: the weights of
// output row Cin - 1 when Cin = 128 j + 1
struct PwLnBw {
  const float* dy; long dy_bs; const float* pre; const float* stats; const float* gamma; const float* beta;
  float* dpre; float* part; int part_rows; int part_q; float* amax; int namax;
  const unsigned short* Ahi; const unsigned short* Alo; const float* a_inv;
  float* dx; long dx_bs; const float* xrow_w; long xrow_sk;
  int M, Cin, L, act;
};
bool ssv_pwln_bwd_fused_ok(int B, int Cin, int Cout, int L);
int ssv_launch_pwln_bwd(const PwLnBw& q, int B, int f16, hipStream_t st);
bool ssv_nt_bf3_xrow(int KT, int M, int Nc);    // this shape runs the extra-row kernel: its callers cut RANGE slabs (GemmNT::bstep = 0)
bool ssv_nt_bf3_fits(const GemmNT& g);          // operands addressable with the kernel's 32-bit element offsets
// nch_total / ch_off: this source fills K chunks [ch_off, ch_off + Kpad/32) of planes that have nch_total chunks per row block
// (two matrices side by side along K, e.g. [W_ih | W_hh]); 0, 0 = the plane holds this source only.
int ssv_launch_pack_split(const float* w, void* hi, void* lo, int M, int K, int Kpad, int KT, long sm, long sk, long sj, int perm_h, hipStream_t st,
                          int nch_total = 0, int ch_off = 0);
// split-fp16 planes of one dense weight (w_elems floats at w): partial maxima -> aux[0..63], 2^-ea -> aux[64]; then the planes
int ssv_launch_pack_split_f16(const float* w, long w_elems, void* hi, void* lo, int M, int K, int Kpad, int KT, long sm, long sk, long sj, float* aux, hipStream_t st);
// ... with the scale taken from a list the caller filled (several weights sharing one scale; LSTM row order / side-by-side planes as in ssv_launch_pack_split)
int ssv_launch_pack_split_f16_list(const float* w, void* hi, void* lo, int M, int K, int Kpad, int KT, long sm, long sk, long sj, int perm_h,
                                   const float* list, int nlist, float* inv_out, hipStream_t st, int nch_total = 0, int ch_off = 0);
#define SSV_F16_AUX_FLOATS 128          // floats of aux the call above needs (64 partial maxima, the inverse scale, padding)
#define SSV_AMAX_FALLBACK 64            // partial maxima an internally computed |x| list has (one list for the whole tensor)
// partial maxima of |x| for B items of n dense floats each (item stride x_bs): out[b * npb + i], i < npb
int ssv_launch_absmax(const float* x, long x_bs, int B, long n, float* out, int npb, hipStream_t st);
int ssv_nt_bf3_tiles(int KT, int M, int Nc);
void ssv_nt_bf3_tile(int KT, int M, int Nc, int* wm, int* ntc);
int ssv_nt_bf3_wg_per_cu(int KT, int wm, int ntc);        // co-resident workgroups per CU of that instantiation (its register count)
int ssv_precision();      // 0 = exact fp32 MFMA, 1 = split-bf16 MFMA, 2 = split-fp16 MFMA with power-of-two operand scales (default)
// The three tuning knobs that remain (per-shape overrides for in-step sweeps: SSV_NNB_FORCE="kt:M:N=wm,nt;...", SSV_NT_FORCE="M:Nc:k=Z;...",
// SSV_LN_GROUPS for tools/bench_ln.py): read from the environment ONCE at first use -- a launch must not cost getenv() scans -- and
// again only when a tuning script calls ssv_reload_tuning() (exported, not part of include/ssv_hip.h).
enum { SSV_T_NT_FORCE, SSV_T_NNB_FORCE, SSV_T_LN_GROUPS, SSV_T_LN_PERSIST, SSV_T_LSTM_MERGE, SSV_T_PWLN_BWD, SSV_T_COUNT };
const char* ssv_tuning(int knob);          // value of the knob or nullptr
int ssv_pack_job_blocks(const ssv_pack_job& j);           // workgroups one job of ssv_conv_pack_multi takes
int ssv_launch_pack_multi(const ssv_pack_job* jobs_dev, int njobs, int nblocks, float* amax_ws, hipStream_t st);   // amax_ws != null: split-fp16 planes
#define SSV_PACK_AMAX_PER_WEIGHT 32     // partial maxima per weight in amax_ws (njobs / 2 weights)

// ---- small helpers (misc.hip) ---------------------------------------------------------
int ssv_launch_reduce_slabs(const float* slabs, float* out, long n, int Z, long slab_stride, hipStream_t st);
int ssv_launch_reduce_slabs_perm(const float* slabs, float* out, int M, int Nc, int KT, int Z, hipStream_t st);
int ssv_launch_reduce_pair(const float* slabs, float* out, int M, int Nc, int KT, int Z, const float* part, float* pout, int n2, int nblk, hipStream_t st);
int ssv_launch_reduce_pair_multi(const ssv_wgrad_job* jobs, int njobs, const float* slabs, int M, int Nc, int KT, int Z, int n2, int nblk, hipStream_t st);
int ssv_launch_pack_wt(const float* w, float* wt, int Cout, int Cin, int KT, hipStream_t st);
int ssv_launch_fill(float* p, float v, long n, hipStream_t st);
int ssv_launch_linear_len1_fwd(const float* x, long x_bs, const float* w, const float* bias, const float* bias_b, long sbb, float* y, long y_bs,
                               int B, int K, int M, hipStream_t st);
int ssv_launch_linear_len1_wgrad(const float* dy, long dy_bs, const float* x, long x_bs, float* dw, int B, int K, int M, hipStream_t st);

static inline int ssv_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
// Profiling aid (env SSV_SHAPE_LOG=<file>): every distinct (kernel, grid) a launcher issues is appended once as
//   kernel-name \t gridX x gridY x gridZ (in threads, as rocprofv3 prints it) \t algorithmic FLOP \t algorithmic HBM bytes \t note
// so that tools/summarize_prof.py can put "achieved / roof" beside each row of a kernel trace.  Off (one predictable branch) otherwise.
bool ssv_shape_log_on();
void ssv_shape_log(const char* kernel, dim3 grid, dim3 block, double flops, double bytes, const char* note);
// entries per batch item of the scale lists the LayerNorm / gate kernels write (include/ssv_hip.h, ssv_amax_rows): the 16-column
// kernels fill one per tile and zero the rest, the streaming forward (64 columns x 4 channel quarters per tile) fills four per tile
// (eight per tile with 64-channel chunks was measured: the kernel gains 2 %, every consumer's longer list costs more: +0.09 ms a step)
__host__ __device__ static inline int ssv_amax_rows_(int L) { return 4 * ((L + 63) / 64); }

// ---- XCD-aware workgroup order ----------------------------------------------------------------
// Workgroups are dealt round-robin to the 8 XCDs (linear id % 8, x fastest), each with a private 4 MB L2.  In launch
// order, the workgroups that share an operand tile (the M tiles of one column tile, the tiles of one batch slab) are
// neighbours, i.e. they land on 8 different XCDs and each L2 fetches that tile from HBM again.  Re-numbered so that
// XCD k walks the contiguous range [k*total/8, (k+1)*total/8), neighbours in the problem are neighbours in one L2.
__device__ __forceinline__ unsigned ssv_xcd_order(unsigned lin, unsigned total) {
  const unsigned xcd = lin & 7u, idx = lin >> 3, q = total >> 3, r = total & 7u;
  return xcd * q + (xcd < r ? xcd : r) + idx;
}

// ---- split-fp16 ("f16x2") arithmetic -------------------------------------------------------------
// An fp32 operand x is scaled by a power of two s = 2^e chosen per tensor (or per batch item) so that max |x| s lies in
// [2^14, 2^15), then split as hi = fp16(x s), lo = fp16(x s - hi): 22 significand bits for every element within 2^-17 of
// the maximum (lo stays a normal fp16 number), an absolute error of 2^-40 of the maximum below that.  A product is
// a_lo*b_hi + a_hi*b_lo + a_hi*b_hi on v_mfma_f32_16x16x32_f16 (each fp16 x fp16 product is exact in fp32; the dropped
// lo*lo term is 2^-22 relative), fp32 accumulate, and the result is multiplied by 2^-(ea + eb).  Same three MFMAs per
// product as the split-bf16 mode at ~2^-22 instead of ~2^-16: fp32-grade results at the bf16 MFMA rate.
// amax -> (2^e, 2^-e): exponent field E of amax (value in [2^(E-127), 2^(E-126))) -> e = 141 - E.
__device__ __forceinline__ void ssv_pow2_scale(float amax, float& scale, float& inv) {
  int E = (int)((__float_as_uint(amax) >> 23) & 0xffu);
  E = min(max(E, 15), 254);
  scale = __uint_as_float((unsigned)(268 - E) << 23);
  inv = __uint_as_float((unsigned)(E - 14) << 23);
}
// a value every lane of the wave holds (hipcc cannot prove it after an LDS round trip) -> a scalar register
// Buffer loads: uniform 128-bit resource (base address in SGPRs) + 32-bit VGPR byte offset + 32-bit SGPR byte offset.  Used where a
// loop loads many streams off one base: with plain pointers hipcc's strength reduction keeps a 64-bit VGPR pointer PER STREAM
// (24 input-row streams of the conv kernel = 48 VGPRs and 48 VALU adds per chunk); here a stream costs nothing -- its row offset is
// scalar arithmetic.  num_records = 2^32 - 1: no range check is relied on, callers clamp as before.
typedef unsigned ssv_u32x4g __attribute__((vector_size(16)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ssv_buf(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000);
}
__device__ __forceinline__ float ssv_buf_f32(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
}
__device__ __forceinline__ uint4 ssv_buf_u4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
// A pointer READ FROM MEMORY (a job table) is a generic pointer to hipcc: it emits flat_load / flat_store, which count on vmcnt AND
// lgkmcnt and return out of order, so every wait on them -- and every wait on an LDS read issued after them -- becomes
// vmcnt(0) lgkmcnt(0): no load stays in flight across a use.  Kernel arguments are known to be global; this says so for the rest.
template <class T> __device__ __forceinline__ T* ssv_global(T* p) {
  return (T*)(__attribute__((address_space(1))) T*)p;
}
__device__ __forceinline__ float ssv_uniform(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
// maximum of a per-thread value over the workgroup (NW waves); sm: NW floats of LDS nobody else uses around the call
template <int NW>
__device__ __forceinline__ float ssv_wg_max(float v, float* sm) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) v = fmaxf(v, __shfl_xor(v, o));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
  __syncthreads();
  float r = sm[0];
#pragma unroll
  for (int i = 1; i < NW; ++i) r = fmaxf(r, sm[i]);
  return r;
}
template <int CTRL> __device__ __forceinline__ float ssv_dpp_mov(float v);
// maximum of the n non-negative floats at p, computed by EVERY wave on its own (no LDS, no barrier: a GEMM prologue cannot
// afford a workgroup round trip per tile -- measured 3-4 % of the large conv kernels): each lane takes entries lane, lane + 64,
// ..., a DPP butterfly inside the rows of 16 lanes, then the four row results through scalar registers.  Returns a scalar.
template <int CTRL>
__device__ __forceinline__ float ssv_dpp_max(float v) { return fmaxf(v, ssv_dpp_mov<CTRL>(v)); }
__device__ __forceinline__ float ssv_wave_list_max(const float* __restrict__ p, int n) {
  float v = 0.f;
  for (int i = threadIdx.x & 63; i < n; i += 64) v = fmaxf(v, p[i]);
  v = ssv_dpp_max<0xB1>(v);      // quad_perm [1,0,3,2]
  v = ssv_dpp_max<0x4E>(v);      // quad_perm [2,3,0,1]
  v = ssv_dpp_max<0x141>(v);     // row_half_mirror
  v = ssv_dpp_max<0x140>(v);     // row_mirror
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// maximum of the n floats at p (every thread of an NW-wave workgroup gets it)
template <int NW>
__device__ __forceinline__ float ssv_list_max(const float* __restrict__ p, int n, float* sm) {
  float v = 0.f;
  for (int i = threadIdx.x; i < n; i += 64 * NW) v = fmaxf(v, p[i]);
  return ssv_wg_max<NW>(v, sm);
}

// ---- cross-lane sums on the VALU (DPP) ---------------------------------------------------
// hipcc lowers __shfl_xor to ds_bpermute_b32 (an LDS-pipe instruction); inside a DPP row of 16 lanes the same
// butterfly is four v_add_f32_dpp.  Pairing (hence rounding) is that of a xor-butterfly over 1, 2, 4, 8.
template <int CTRL>
__device__ __forceinline__ float ssv_dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float ssv_row16_sum(float v) {   // every lane gets the sum over its aligned group of 16 lanes
  v += ssv_dpp_mov<0xB1>(v);     // quad_perm [1,0,3,2]
  v += ssv_dpp_mov<0x4E>(v);     // quad_perm [2,3,0,1]
  v += ssv_dpp_mov<0x141>(v);    // row_half_mirror: the other quad of this half row
  v += ssv_dpp_mov<0x140>(v);    // row_mirror: the other half row
  return v;
}
__device__ __forceinline__ float ssv_wave_sum(float v) {    // every lane gets the sum over the 64 lanes
  v = ssv_row16_sum(v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
