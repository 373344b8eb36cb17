/* libssv_hip.so -- C ABI of the MI355X (gfx950) hot path of SpoofSV.
 *
 * The reference (MingruiYuan/SpoofSV) is pure Python on torch and has NO plugin / operator / FFI
 * interface: its hot path is the nn.Module surface of models/TTSModel.py and
 * GE2E/speech_embedder_net.py.  Each entry point below therefore replaces a *module forward (or its
 * autograd backward)* of the reference, cited as file:line; a binding is a ctypes stub
 * (INTEGRATION.md shows the one a maintainer of the reference would add).
 *
 * Conventions (all entries):
 *   - every pointer is a DEVICE pointer to caller-owned memory; fp32 unless stated; text ids and
 *     attention indices are int64.  Activations are (B, C, T) with T fastest and rows contiguous
 *     (channel stride = T); `*_bs` arguments are the batch stride in floats (C*T when dense), which
 *     lets a caller pass channel slices of a wider tensor (K|V halves, the R|Q concatenation).
 *   - no allocation, no ownership transfer, no host synchronisation: work is enqueued on `stream`
 *     (a hipStream_t; NULL = default stream) and the call returns.  Safe under hipGraph capture.
 *   - scratch memory is passed as (ws, ws_bytes); the matching *_workspace() query gives the size.
 *   - return value: 0 on success, -1 bad shape/argument, -2 unsupported configuration, otherwise
 *     -(int)hipError_t.  ssv_last_error() returns a thread-local message for the last failure.
 *   - re-entrant; callable from any host thread (torch's autograd thread calls the *_bwd entries).
 *   - outputs listed as "saved" are the tensors the matching backward needs; the caller owns them.
 */
#ifndef SSV_HIP_H
#define SSV_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* ssv_stream_t; /* hipStream_t */

int ssv_version(void);            /* ABI version, currently 7 (7 = ssv_spec_losses_fwd_bwd, ssv_deinterleave2_rows_amax, w_packed / y_amax of ssv_deconv1d_k2s2_fwd, ssv_bias_grad; 6 = ssv_shift_right_amax, ssv_deinterleave2_amax, ssv_lstm_fwd_cached, GE2E training without shape / mode limits; 5 = ssv_attention_train_fwd_rq; 2 = split-fp16 operand scales; 3 = max_shift of ssv_conv1d_bwd_weight_multi, ssv_pointwise_conv_ln_act_fwd;
                                    4 = compact partial rows: the nblk of a weight-gradient job is ssv_ln_bwd_partial_rows(...), not ssv_ln_partial_rows(B, L)) */
const char* ssv_arch(void);       /* "gfx950" */
const char* ssv_last_error(void); /* thread-local, valid until the next failing call on this thread */
/* Arithmetic of the conv GEMMs (the reference computes in fp32: requirements.txt:5, nn.Conv1d at models/TTSModel.py:59):
 *   0 = fp32-in MFMA: bit-exact fp32 fma chains (157 TFLOP/s peak);
 *   1 = split-bf16 MFMA: each fp32 operand as bf16 hi+lo, three bf16 MFMAs per product, fp32 accumulate; ~2^-16 per
 *       product -- NARROWER than the reference's arithmetic, opt-in only;
 *   2 = split-fp16 MFMA (default): each operand is scaled by a power of two (per tensor / per batch item, from its
 *       maximum magnitude) and split into fp16 hi+lo = 22 significand bits, three fp16 MFMAs per product (every
 *       fp16 x fp16 product is exact in fp32), fp32 accumulate, result rescaled; ~2^-22 per product = fp32-grade at
 *       the bf16/fp16 MFMA rate.  The LSTM products of the GE2E embedder run in this arithmetic too (one power-of-two scale for all
 *       weight matrices of a launch; the recurrent activations, |h| < 1, take the fixed scale 2^14).
 * Env SSV_PRECISION = fp32 | bf16x3 | f16x2 selects the mode at load.  Returns the previous mode.  Process-wide.
 *
 * Operand scales (mode 2).  A kernel that reads an fp32 tensor as an MFMA operand needs max |x| BEFORE it starts.  The
 * convention: a "scale list" of x (B, C, L) is `namax` floats per batch item, items consecutive, whose maximum per item
 * is max |x(b)| -- partial maxima, in any partition.  The LayerNorm / gate kernels write such a list for their output as
 * a by-product (ssv_amax_rows(L) entries per item: one per kernel tile, unused ones zeroed), ssv_absmax computes one for any tensor, and
 * every entry below that takes `*_amax, *_namax` arguments accepts NULL, 0: it then computes the list itself in its
 * workspace (one extra small launch).  Outside mode 2 the arguments are ignored (amax outputs are still written). */
int ssv_set_precision(int mode);
int ssv_get_precision(void);
int ssv_amax_rows(int L);         /* entries per batch item of the lists the LayerNorm / gate kernels write: 4 * ceil(L / 64) */
/* amax[b * namax + i] = max |x| over the i-th of namax equal pieces of item b (n dense floats at x + b * x_bs). */
int ssv_absmax(const float* x, long x_bs, int B, long n, float* amax, int namax, ssv_stream_t stream);
/* Two layout changes of the trainers that also deliver the result's scale list (one launch instead of a copy kernel + ssv_absmax; amax may be
 * NULL).  Teacher forcing, train/ordinary.py:226 = train/adversarial_wasserstein_gp.py:277 (`torch.cat((zeros, mel[:, :, :-1]), -1)`):
 * y (B, C, T) dense, y(b, c, 0) = 0, y(b, c, t) = x(b, c, t - 1); x items x_bs apart. */
int ssv_shift_right_amax(const float* x, long x_bs, float* y, int B, int C, int T, float* amax, int namax, ssv_stream_t stream);
/* out[j][b][i] = x[b][2 i + j] for j = 0, 1, i < n: the two taps of a ConvTranspose1d(k = 2, s = 2) output gradient (models/TTSModel.py:309,314,
 * backward), each then a dense (B, n) operand of a k = 1 weight gradient; amax: namax partial maxima of |x| per item. */
int ssv_deinterleave2_amax(const float* x, long x_bs, float* out, int B, long n, float* amax, int namax, ssv_stream_t stream);
/* The same, row by row (ABI 7): out(b, 2 r + j, t) = x(b, r, 2 t + j), x (B, rows, 2 L) with items x_bs apart, out (B, 2 rows, L) dense.  A
 * ConvTranspose1d(k = 2, s = 2) is the 1x1 convolution u = W2 x, W2 = w.view(Cin, 2 Cout), followed by y(b, o, 2 t + j) = u(b, 2 o + j, t): with its
 * output gradient in this layout, dx is ONE ssv_conv1d_fwd of it with the weight w.view(Cin, 2 Cout, 1) and dw ONE ssv_conv1d_bwd_weight that lands in
 * the weight's own (Cin, Cout, 2) layout (spoofsv_amd/ops.py, DeconvK2S2Fn.backward). */
int ssv_deinterleave2_rows_amax(const float* x, long x_bs, float* out, int B, int rows, int L, float* amax, int namax, ssv_stream_t stream);

/* ---- Conv1d (stride 1, kernel 1 or 3, dilated, "same" or causal zero padding) -------------------
 * Replaces nn.Conv1d as used at models/TTSModel.py:59,78 (highway), :115-117, :154-158, :203-214,
 * :323-339 (kernel 1) and nn.Linear on (B, D, 1) speaker codes (:150-151, :174, :179).
 * y(b,o,t) = bias[o] + bias_b[b*Cout+o] + sum_{c,j} w[o,c,j] * x(b,c,t + (j-j0)*dilation),
 * j0 = (k-1)/2 ("same", :57-59) or k-1 (causal: 2*pad zeros on the left, :72-74).
 * bias and bias_b (the broadcast speaker term of :175,:180) may be NULL. */
/* w_packed (every entry that takes it): NULL, or the weight's resident pre-split planes written by
 * ssv_conv_pack_multi (see "Resident pre-split weights" below) -- then the call skips its own weight split.  The
 * caller vouches that the planes are current for w. */
size_t ssv_conv1d_fwd_workspace(int Cin, int Cout, int k);   /* holds the pre-split weights */
/* y_colstats (may be NULL; split-MFMA modes, Cout % 64 == 0, dense y): per batch item, 64-row group of output channels and
 * column, the mean and the sum of squared deviations of y over the group's 64 channels, (B, Cout/64, L, 2) floats -- the
 * channel-axis LayerNorm that follows merges the groups instead of reducing over y again (ssv_highway_conv1d_fwd does). */
int ssv_conv1d_fwd(const float* x, long x_bs, const float* x_amax, int x_namax, const float* w, const void* w_packed, const float* bias,
                   const float* bias_b, float* y, long y_bs, float* y_colstats,
                   int B, int Cin, int Cout, int L, int k, int dilation, int causal,
                   void* ws, size_t ws_bytes, ssv_stream_t stream);
/* dx = conv1d_transpose(dy, w) [+ dx_add if non-NULL, same layout as dx]; ws holds w transposed. */
size_t ssv_conv1d_bwd_data_workspace(int Cin, int Cout, int k);
int ssv_conv1d_bwd_data(const float* dy, long dy_bs, const float* dy_amax, int dy_namax, const float* w, const void* w_packed,
                        const float* dx_add, float* dx, long dx_bs,
                        int B, int Cin, int Cout, int L, int k, int dilation, int causal,
                        void* ws, size_t ws_bytes, ssv_stream_t stream);
/* dw(o,c,j) = sum_{b,t} dy(b,o,t) x(b,c,t+(j-j0)*dilation); split over the batch into slabs, summed
 * in a fixed order (bitwise reproducible). */
size_t ssv_conv1d_bwd_weight_workspace(int B, int Cin, int Cout, int L, int k);
int ssv_conv1d_bwd_weight(const float* dy, long dy_bs, const float* dy_amax, int dy_namax, const float* x, long x_bs, const float* x_amax, int x_namax,
                          float* dw, int B, int Cin, int Cout, int L, int k, int dilation, int causal,
                          void* ws, size_t ws_bytes, ssv_stream_t stream);
/* out(b,c) = sum_t x(b,c,t): gradient of a (B,C,1) broadcast term / of a bias per batch item. */
int ssv_rowsum(const float* x, long x_bs, float* out, int B, int C, int L, ssv_stream_t stream);
/* out(c) = sum_{b,t} x(b,c,t): the bias gradient of a conv layer in one launch (ABI 7; ssv_rowsum + ssv_sum_slabs before), fixed summation order. */
int ssv_bias_grad(const float* x, long x_bs, float* out, int B, int C, int L, ssv_stream_t stream);
/* out[i] = sum_{z<Z} slabs[z*stride + i], i < n, summed in index order (bitwise reproducible). */
int ssv_sum_slabs(const float* slabs, float* out, long n, int Z, long stride, ssv_stream_t stream);
/* dst(b, 0:n) = src(b, 0:n) for B rows with independent row strides (the Q half of torch.cat((R, Q), 1),
 * models/TTSModel.py:270). */
int ssv_copy_rows(const float* src, long src_bs, float* dst, long dst_bs, int B, long n, ssv_stream_t stream);

/* ---- LayerNorm over channels (+ activation) ----------------------------------------------------
 * Replaces `ln(x.permute(0,2,1)).permute(0,2,1)` followed by F.relu / F.sigmoid,
 * models/TTSModel.py:129-131, :175-180, :219-231, :344-361.  act: 0 none, 1 relu, 2 sigmoid.
 * stats (B,2,L) = mean, rstd: saved for backward (may be NULL for inference). */
size_t ssv_channel_ln_act_fwd_workspace(int B, int C, int L);
/* y_amax (may be NULL): scale list of y, ssv_amax_rows(L) entries per item, for the convolution that reads y next. */
int ssv_channel_ln_act_fwd(const float* x, long x_bs, const float* gamma, const float* beta,
                           float* y, long y_bs, float* y_amax, float* stats, int B, int C, int L, int act,
                           void* ws, size_t ws_bytes, ssv_stream_t stream);
/* The whole forward of such a link: y = act(LN(conv1x1(x) + bias [+ s])) -- models/TTSModel.py:128-131, :173-180 (s = fc(spk), (B,Cout)),
 * :218-231, :343-361.  pre (B,Cout,L) dense = the LayerNorm's input and stats (B,2,L) (may be NULL) are kept for
 * ssv_pointwise_conv_ln_act_bwd.  One launch where the library has the fused kernel for the shape (a workgroup owns all output rows of
 * its column tile), otherwise the product followed by the LayerNorm kernel; x_amax / w_packed / y_amax as in ssv_conv1d_fwd /
 * ssv_channel_ln_act_fwd. */
size_t ssv_pointwise_conv_ln_act_fwd_workspace(int Cin, int Cout);
int ssv_pointwise_conv_ln_act_fwd(const float* x, long x_bs, const float* x_amax, int x_namax, const float* w, const void* w_packed, const float* bias,
                                  const float* s, const float* gamma, const float* beta, float* pre, float* stats, float* y, long y_bs, float* y_amax,
                                  int B, int Cin, int Cout, int L, int act, void* ws, size_t ws_bytes, ssv_stream_t stream);
size_t ssv_channel_ln_act_bwd_workspace(int B, int C, int L);
/* dx: gradient w.r.t. the pre-LN input; pgrads (3,C) = dgamma, dbeta, sum_{b,t} dx (bias gradient of
 * the producing conv).   pgrads may be NULL: the parameter gradients are not wanted
 * (the partial rows stay unsummed in ws) -- the gradient penalty's input-gradient pass, train/adversarial_wasserstein_gp.py:303-304. */
int ssv_channel_ln_act_bwd(const float* dy, long dy_bs, const float* x, long x_bs, const float* stats,
                           const float* gamma, const float* beta, float* dx, long dx_bs, float* pgrads,
                           int B, int C, int L, int act, void* ws, size_t ws_bytes, ssv_stream_t stream);

/* The whole backward of y = act(LN(conv1x1(x) [+ s])) (models/TTSModel.py:128-131, :173-180, :218-231, :343-361) in one call,
 * so that the LayerNorm partial sums and the weight-gradient slabs share one reduction launch.  pre (B,Cout,L) dense and stats
 * (B,2,L) are what the forward saved; dx may be NULL (first layer: the input needs no gradient); ds (B,Cout) = gradient of
 * the broadcast term s, or NULL; pgrads (3,Cout) = dgamma, dbeta, dbias. */
size_t ssv_pointwise_conv_ln_act_bwd_workspace(int B, int Cin, int Cout, int L);
int ssv_pointwise_conv_ln_act_bwd(const float* dy, long dy_bs, const float* x, long x_bs, const float* x_amax, int x_namax,
                                  const float* w, const void* w_packed,
                                  const float* gamma, const float* beta, const float* pre, const float* stats,
                                  float* dx, long dx_bs, float* dw, float* pgrads, float* ds,
                                  int B, int Cin, int Cout, int L, int act, void* ws, size_t ws_bytes, ssv_stream_t stream);

/* ---- Weight gradients of several equal-shaped conv layers in ONE launch ------------------------------------------
 * The weight gradient is off the critical path of backward (nothing downstream reads dW), and a single layer's reduction over
 * (batch, time) must be cut into Z slabs only to fill the chip (C = 256, L = 325: 16 output tiles x 32 slabs = 49 MiB of partial
 * sums written and read back per layer).  A trainer can therefore run backward with the *_bwd_data entries below (LayerNorm /
 * gate backward + data gradient only; dH and the LayerNorm partial rows are left in caller-owned buffers), collect one job per
 * layer, and hand all layers of one shape to ssv_conv1d_bwd_weight_multi: njobs x tiles x Z workgroups with a Z that is njobs
 * times smaller, then one reduction launch for all jobs (slabs -> dw, partial rows -> pgrads).  Same arithmetic per product;
 * only the slab boundaries (hence the fp32 summation order over the batch) differ from the one-layer entry.
 * jobs_dev: device array of njobs jobs.  Per job: dy (B,Cout,L) = dH, x (B,Cin,L), dw (Cout,Cin,k) out, part (nblk, n2) partial
 * rows or NULL, pgrads (n2) out or NULL, shift[j] = (j - j0) * dilation as ssv_conv_shifts returns them.
 * max_shift: the caller's bound on |shift[j]| over all jobs (the table lives on the device; the entry plans its kernel from the
 * bound), or -1 for "unknown" (the general kernel).  A job whose shifts exceed a stated bound gets NaN gradients, not wrong ones.
 * Split-bf16 mode only, B*L >= 256 (ssv_conv1d_bwd_weight_multi_ok). */
typedef struct {
  const float* dy; const float* x; float* dw; const float* part; float* pgrads;
  int shift[3]; int pad_;
  const float* dy_amax; const float* x_amax;   /* split-fp16 (mode 2): the two operands' scale lists, WHOLE lists (all batch items) ... */
  int dy_namax, x_namax;                       /* ... and their total entry counts (B * entries per item); ignored in the other modes */
} ssv_wgrad_job;
int ssv_conv_shifts(int k, int dilation, int causal, int* shift3);
int ssv_conv1d_bwd_weight_multi_ok(int B, int Cin, int Cout, int L, int k);
int ssv_conv1d_bwd_weight_multi_splits(int njobs, int B, int Cin, int Cout, int L, int k);   /* the Z the entry will use */
size_t ssv_conv1d_bwd_weight_multi_workspace(int njobs, int B, int Cin, int Cout, int L, int k);
int ssv_conv1d_bwd_weight_multi(const ssv_wgrad_job* jobs_dev, int njobs, long dy_bs, long x_bs, int B, int Cin, int Cout, int L, int k, int max_shift,
                                int n2, int nblk, void* ws, size_t ws_bytes, ssv_stream_t stream);
/* highwayConv backward without the weight gradient: dx, plus dh (B,2C,L) dense and the partial rows `part` for the job.
 * `part` has room for ssv_ln_partial_rows(B, L) rows (of 6C floats; 3 Cout for the pointwise entry); the launch WRITES the first
 * ssv_ln_bwd_partial_rows(gate, B, C, L, with_amax) of them -- one per column tile of the kernel it picks for the shape (gate = 1: highway
 * gate over C channels, 0: LayerNorm + activation over C = Cout channels; with_amax = whether the call passes a scale-list output) -- and
 * that count is the `nblk` of the job (the rest of the buffer is not read). */
int ssv_ln_partial_rows(int B, int L);
int ssv_ln_bwd_partial_rows(int gate, int B, int C, int L, int with_amax);
size_t ssv_highway_conv1d_bwd_data_workspace(int B, int C, int L, int k);
int ssv_highway_conv1d_bwd_data(const float* dy, long dy_bs, const float* x, long x_bs, const float* w, const void* w_packed,
                                const float* g1, const float* b1, const float* g2, const float* b2, const float* h, const float* stats,
                                float* dx, long dx_bs, float* dh, float* dh_amax, float* part, int B, int C, int L, int k, int dilation, int causal,
                                void* ws, size_t ws_bytes, ssv_stream_t stream);   /* dh_amax (B * ssv_amax_rows(L), may be NULL): scale list of dh for the job */
/* The same for y = act(LN(conv1x1(x) [+ s])): dx (may be NULL), ds (may be NULL), dpre (B,Cout,L) dense and part (rows, 3 Cout). */
size_t ssv_pointwise_conv_ln_act_bwd_data_workspace(int B, int Cin, int Cout, int L);
int ssv_pointwise_conv_ln_act_bwd_data(const float* dy, long dy_bs, const float* w, const void* w_packed, const float* gamma, const float* beta,
                                       const float* pre, const float* stats, float* dx, long dx_bs, float* ds, float* dpre, float* dpre_amax,
                                       float* part, int B, int Cin, int Cout, int L, int act, void* ws, size_t ws_bytes, ssv_stream_t stream);

/* ---- highwayConv ---------------------------------------------------------------------------------
 * Replaces highwayConv.forward, models/TTSModel.py:63-84:
 *   h = conv(x) (2C channels); y = sigmoid(LN1(h[:C])) * LN2(h[C:]) + (1 - sigmoid(LN1(h[:C]))) * x.
 * h (B,2C,L) dense and stats (B,4,L) = mean1, rstd1, mean2, rstd2 are saved for backward. */
size_t ssv_highway_conv1d_fwd_workspace(int B, int C, int L, int k);
int ssv_highway_conv1d_fwd(const float* x, long x_bs, const float* x_amax, int x_namax, const float* w, const void* w_packed, const float* bias,
                           const float* g1, const float* b1, const float* g2, const float* b2,
                           float* h, float* stats, float* y, long y_bs, float* y_amax,
                           int B, int C, int L, int k, int dilation, int causal,
                           void* ws, size_t ws_bytes, ssv_stream_t stream);   /* y_amax: as in ssv_channel_ln_act_fwd */
size_t ssv_highway_conv1d_bwd_workspace(int B, int C, int L, int k);
/* Outputs: dx (B,C,L), dw (2C,C,k), pgrads (6,C) = dgamma1, dbeta1, dgamma2, dbeta2, dbias[:C], dbias[C:]. */
int ssv_highway_conv1d_bwd(const float* dy, long dy_bs, const float* x, long x_bs, const float* x_amax, int x_namax, const float* w, const void* w_packed,
                           const float* g1, const float* b1, const float* g2, const float* b2,
                           const float* h, const float* stats, float* dx, long dx_bs, float* dw, float* pgrads,
                           int B, int C, int L, int k, int dilation, int causal,
                           void* ws, size_t ws_bytes, ssv_stream_t stream);

/* Building block of the above: only the gate + two LayerNorm backward.  dh (B,2C,L) dense = dL/dh (the conv output
 * gradient), dxres = dy*(1-gate) (the residual-path part of dL/dx), pgrads (6,C) as above.  A caller can then run
 * ssv_conv1d_bwd_data(dh, ..., dx_add = dxres) and ssv_conv1d_bwd_weight(dh, x, ...) on two different streams.  pgrads may be NULL (as in ssv_channel_ln_act_bwd). */
size_t ssv_highway_gate_bwd_workspace(int B, int C, int L);
int ssv_highway_gate_bwd(const float* dy, long dy_bs, const float* x, long x_bs,
                         const float* g1, const float* b1, const float* g2, const float* b2,
                         const float* h, const float* stats, float* dh, float* dxres, long dx_bs, float* pgrads,
                         int B, int C, int L, void* ws, size_t ws_bytes, ssv_stream_t stream);

/* ---- textEmbedding -------------------------------------------------------------------------------
 * Replaces textEmbedding.forward, models/TTSModel.py:25-35 (one-hot scatter + Linear):
 * y(b,e,n) = w[e, ids(b,0,n)] + bias[e]; w is the nn.Linear weight (E, vocab).  ids int64 (B,1,N);
 * an id outside [0, vocab) contributes the bias only (the reference would raise in scatter_). */
int ssv_text_embed_fwd(const int64_t* ids, const float* w, const float* bias, float* y,
                       int B, int N, int E, int vocab, ssv_stream_t stream);
int ssv_text_embed_bwd(const int64_t* ids, const float* dy, float* dw, float* dbias,
                       int B, int N, int E, int vocab, ssv_stream_t stream);

/* ---- attention (training) ------------------------------------------------------------------------
 * Replaces models/TTSModel.py:266-270: A = softmax_N(K^T Q / sqrt(d)); R = V A; cat(R, Q).
 * k, v: (B,d,N) with batch stride kv_bs; q: (B,d,T); a: (B,N,T) dense (returned attention, saved);
 * r: (B,d,T) with batch stride r_bs (pass the first half of the (B,2d,T) decoder input). */
int ssv_attention_train_fwd(const float* k, const float* v, long kv_bs, const float* q, long q_bs,
                            float* a, float* r, long r_bs, int B, int d, int N, int T, ssv_stream_t stream);
/* ABI 5: the decoder's input cat(R, Q) (models/TTSModel.py:270) in the same call -- rq (B, 2d, T): R in rows [0, d), a copy of Q in rows [d, 2d).
 * For d % 64 == 0, d <= 256, N <= 192 scores, softmax and V A run in ONE launch (exact-fp32 MFMA, attn_fused.hip). */
int ssv_attention_train_fwd_rq(const float* k, const float* v, long kv_bs, const float* q, long q_bs, float* a, float* rq, long rq_bs,
                               int B, int d, int N, int T, ssv_stream_t stream);
size_t ssv_attention_train_bwd_workspace(int B, int d, int N, int T);
/* dr: grad of R; da_ext: extra dL/dA (guided-attention loss), may be NULL; dq_add: gradient reaching
 * Q through the concatenation, added into dq (may be NULL).  dk, dv have batch stride dkv_bs. */
int ssv_attention_train_bwd(const float* dr, long dr_bs, const float* da_ext, const float* dq_add, long dq_add_bs,
                            const float* k, const float* v, long kv_bs, const float* q, long q_bs, const float* a,
                            float* dk, float* dv, long dkv_bs, float* dq, long dq_bs,
                            int B, int d, int N, int T, void* ws, size_t ws_bytes, ssv_stream_t stream);

/* ---- attention (one synthesis step) ---------------------------------------------------------------
 * Replaces models/TTSModel.py:281-291 for the newest frame: logits of the last query column, masked
 * outside the text window [pma, pma+2] with -2^32 (:282-286), softmax over N, argmax -> pma_out.
 * q_last: (B,d) column t of Q (element stride q_cs between channels, batch stride q_bs);
 * a: (B,N,a_T) attention buffer, column `col` is written.  pma_in/pma_out int64 (B). */
int ssv_attention_step(const float* k, long kv_bs, const float* q_last, long q_bs, long q_cs,
                       const int64_t* pma_in, float* a, int a_T, int col, const int* col_dev, int64_t* pma_out,
                       int B, int d, int N, ssv_stream_t stream);
/* col_dev (DEVICE int*, may be NULL): when given, the frame index is *col_dev instead of `col` and q_last must point at
 * column 0 of Q -- one captured hipGraph of a fixed-shape synthesis step can then be replayed for every frame.
 * ssv_synth_advance closes such a step: mel_in[b][f][*col_dev + 1] = y[b][f][*col_dev] (both (B,F,T) dense; the frame just
 * synthesised is the next input, synthesize.py:108-109), then *col_dev += 1. */
int ssv_synth_advance(const float* y, float* mel_in, int* col_dev, int B, int F, int T, ssv_stream_t stream);
/* r(b,c,t) = sum_n v(b,c,n) a(b,n,t) for t < T (a has row stride a_T). */
int ssv_attention_apply(const float* v, long kv_bs, const float* a, int a_T, float* r, long r_bs,
                        int B, int d, int N, int T, ssv_stream_t stream);

/* ---- ConvTranspose1d(kernel 2, stride 2) ----------------------------------------------------------
 * Replaces upsampling.deconv, models/TTSModel.py:309,314.  w: (Cin, Cout, 2) as nn.ConvTranspose1d.
 * y(b,o,2t+j) = bias[o] + sum_c w[c,o,j] x(b,c,t). */
size_t ssv_deconv1d_k2s2_fwd_workspace(int Cin, int Cout);   /* pre-split weights (when no resident planes are given) */
/* ABI 7: w_packed = the resident planes of the 1x1 weight w.view(Cin, 2 Cout, 1) (ssv_conv_pack_multi; NULL: split here) -- the forward is ONE product over
 * 2 Cout rows whose epilogue interleaves the row pairs --, and y_amax (may be NULL) receives y's operand-scale list, y_namax entries per item. */
int ssv_deconv1d_k2s2_fwd(const float* x, long x_bs, const float* x_amax, int x_namax, const float* w, const void* w_packed, const float* bias,
                          float* y, long y_bs, float* y_amax, int y_namax, int B, int Cin, int Cout, int L, void* ws, size_t ws_bytes, ssv_stream_t stream);
size_t ssv_deconv1d_k2s2_bwd_workspace(int B, int Cin, int Cout);
/* dw may be NULL: the weight gradient is then left to the caller -- dw[:, :, j] is the k = 1 conv weight gradient of
 * (dy' = x, x' = dy[:, :, j::2]) and runs on the split-precision kernel through ssv_conv1d_bwd_weight once dy is de-interleaved,
 * which the stride-2 operand of this entry's own product (an exact-fp32 MFMA kernel) cannot; spoofsv_amd/ops.py does so. */
int ssv_deconv1d_k2s2_bwd(const float* dy, long dy_bs, const float* dy_amax, int dy_namax, const float* x, long x_bs, const float* w,
                          float* dx, long dx_bs, float* dw, float* dbias,
                          int B, int Cin, int Cout, int L, void* ws, size_t ws_bytes, ssv_stream_t stream);

/* ---- losses ----------------------------------------------------------------------------------------
 * Replaces train/ordinary.py:230-231 / :249-250: out[0] = mean|gt - y|,
 * out[1] = mean(-gt*log(y+1e-8) - (1-gt)*log(1-y+1e-8)) over n elements (dense tensors). */
size_t ssv_spec_losses_workspace(long n);
int ssv_spec_losses_fwd(const float* y, const float* gt, long n, float* out, void* ws, size_t ws_bytes, ssv_stream_t stream);
/* dy = gscale[0] * dl1/dy + gscale[1] * dbd/dy; gscale is a DEVICE pointer to 2 floats. */
int ssv_spec_losses_bwd(const float* y, const float* gt, long n, const float* gscale, float* dy, ssv_stream_t stream);
/* Both of the above in one pass over (y, gt) (ABI 7): `out` as ssv_spec_losses_fwd, `dy` as ssv_spec_losses_bwd for the gradient seed `gscale` (DEVICE pointer
 * to 2 floats) that the caller will hand to the backward -- a trainer knows it before the forward runs (train/ordinary.py:237,253: loss.backward()
 * seeds every term with 1).  The sums are taken in another order than ssv_spec_losses_fwd's (same value to fp32 rounding of a mean over n). */
int ssv_spec_losses_fwd_bwd(const float* y, const float* gt, long n, const float* gscale, float* out, float* dy, void* ws, size_t ws_bytes,
                            ssv_stream_t stream);
/* Replaces train/ordinary.py:232-234: out[0] = sum(a * gaw[:N,:T]) / (B*N*T); gaw row stride gaw_T. */
size_t ssv_guided_att_loss_workspace(int B, int N, int T);
int ssv_guided_att_loss_fwd(const float* a, const float* gaw, int gaw_T, float* out, int B, int N, int T,
                            void* ws, size_t ws_bytes, ssv_stream_t stream);
int ssv_guided_att_loss_bwd(const float* gaw, int gaw_T, const float* gscale, float* da, int B, int N, int T,
                            ssv_stream_t stream);

/* ---- Resident pre-split weights ------------------------------------------------------------------
 * The split-bf16 conv kernels read weights as bf16 hi/lo planes in MFMA fragment order.  Splitting a weight costs one
 * small launch per conv call (forward order for the forward, transposed order for the data gradient): ~100 launches
 * per training step.  Instead a trainer can keep one caller-owned buffer of ssv_conv_pack_bytes() per weight and
 * refresh ALL of them with one launch after each optimizer step; the conv entry points then take that buffer as
 * w_packed.  A job table is planned once on the host (ssv_conv_pack_plan), copied to the device by the caller, and
 * replayed by ssv_conv_pack_multi (safe inside hipGraph capture). */
typedef struct { const float* w; void* planes; int M, K, Kpad, KT; long sm, sk; int first_block, pad_; float* inv_out; } ssv_pack_job;
size_t ssv_conv_pack_bytes(int Cout, int Cin, int k);   /* forward + transposed planes of one weight (Cout,Cin,k) + their two inverse scales (mode 2) */
/* Fills 2*n HOST jobs (forward, transposed per weight) for weights w[i] (DEVICE pointers, torch layout (Cout,Cin,k))
 * and their DEVICE buffers planes[i]; returns the number of workgroups ssv_conv_pack_multi must launch, or < 0. */
int ssv_conv_pack_plan(int n, const float* const* w, void* const* planes, const int* Cout, const int* Cin, const int* k,
                       ssv_pack_job* jobs_host);
/* The planes are written in the arithmetic mode in force at the call (bf16 or scaled fp16 halves): re-pack after ssv_set_precision.
 * ws (mode 2 only): ssv_conv_pack_multi_workspace(njobs) bytes for the weights' partial maxima. */
size_t ssv_conv_pack_multi_workspace(int njobs);
int ssv_conv_pack_multi(const ssv_pack_job* jobs_dev, int njobs, int nblocks, void* ws, size_t ws_bytes, ssv_stream_t stream);

/* ---- Adam -------------------------------------------------------------------------------------------
 * Replaces optim.Adam(...).step(), train/ordinary.py:182,238 (config.json:41-46), for many tensors in
 * one launch.  `chunks` is a DEVICE array describing contiguous pieces of parameters. */
typedef struct { float* p; const float* g; float* m; float* v; long n; } ssv_adam_chunk;
/* step: 1-based step count.  If step_dev (DEVICE int*) is non-NULL the count is *step_dev + 1 and the
 * counter is incremented on the stream after the update, so a captured hipGraph advances on replay. */
int ssv_adam_multi(const ssv_adam_chunk* chunks, int nchunks, float lr, float beta1, float beta2, float eps,
                   int step, int* step_dev, ssv_stream_t stream);

/* ---- GE2E speaker embedder --------------------------------------------------------------------------
 * Replaces SpeechEmbedder.forward, GE2E/speech_embedder_net.py:27-33: 3-layer nn.LSTM(batch_first) ->
 * last frame -> Linear -> x / ||x||.  x: (Bn, T, F) row-major as the reference feeds it.
 * Weights in torch layout: w_ih[l] (4H, F or H), w_hh[l] (4H, H), b_ih[l], b_hh[l] (4H), gate order
 * i, f, g, o.  h_last: (Bn, H) output of the top layer at the last frame. */
size_t ssv_lstm_fwd_workspace(int Bn, int T, int F, int H, int layers);
int ssv_lstm_fwd(const float* x, const float* const* w_ih, const float* const* w_hh,
                 const float* const* b_ih, const float* const* b_hh, float* h_last,
                 int Bn, int T, int F, int H, int layers, void* ws, size_t ws_bytes, ssv_stream_t stream);
/* The same forward for a caller that runs it again and again on FIXED weights (d-vector extraction, GE2E/dvector_create.py:100, batch after
 * batch): weights_packed != 0 promises that `ws` is the buffer of an earlier call with the same shape, arithmetic mode and weight VALUES, not
 * written by anyone since -- the call then skips the weights' scale scans and the split into fp16 / bf16 planes.  0 = ssv_lstm_fwd. */
int ssv_lstm_fwd_cached(const float* x, const float* const* w_ih, const float* const* w_hh,
                        const float* const* b_ih, const float* const* b_hh, float* h_last,
                        int Bn, int T, int F, int H, int layers, void* ws, size_t ws_bytes, int weights_packed, ssv_stream_t stream);
/* e = normalize(h W^T + b): h (Bn,H), w (P,H), e (Bn,P).  norms (Bn, may be NULL): |h W^T + b| per row, kept for the backward. */
size_t ssv_proj_l2norm_fwd_workspace(int Bn, int P);
int ssv_proj_l2norm_fwd(const float* h, const float* w, const float* bias, float* e, float* norms, int Bn, int H, int P,
                        void* ws, size_t ws_bytes, ssv_stream_t stream);
/* Backward: de (Bn,P) -> dh (Bn,H), dw (P,H), dbias (P). */
size_t ssv_proj_l2norm_bwd_workspace(int Bn, int P);
int ssv_proj_l2norm_bwd(const float* de, const float* e, const float* norms, const float* h, const float* w,
                        float* dh, float* dw, float* dbias, int Bn, int H, int P, void* ws, size_t ws_bytes, ssv_stream_t stream);
/* Training (GE2E/train_speech_embedder.py:77-83): the same LSTM forward, keeping every frame of h, c and
 * the activated gates in `saved` (caller-owned, ssv_lstm_saved_bytes), and backpropagation through time from dh_last
 * (Bn,H) to the gradients of all weights and biases (arrays of `layers` DEVICE pointers, torch layouts).  No shape or mode
 * limits (nn.LSTM + autograd have none, GE2E/speech_embedder_net.py:19): in a split mode with hidden % 32 == 0 and
 * batch >= 8 the (layer, frame) wavefront kernels run in that mode's arithmetic; every other case -- mode 0, any other
 * hidden size, fewer than 8 utterances -- runs frame by frame on the exact-fp32 MFMA GEMMs.  One `saved` layout for both;
 * forward and backward of one iteration must run in the same mode. */
size_t ssv_lstm_saved_bytes(int Bn, int T, int F, int H, int layers);
size_t ssv_lstm_train_fwd_workspace(int Bn, int T, int F, int H, int layers);
int ssv_lstm_train_fwd(const float* x, const float* const* w_ih, const float* const* w_hh,
                       const float* const* b_ih, const float* const* b_hh, float* h_last, void* saved,
                       int Bn, int T, int F, int H, int layers, void* ws, size_t ws_bytes, ssv_stream_t stream);
size_t ssv_lstm_bwd_workspace(int Bn, int T, int F, int H, int layers);
int ssv_lstm_bwd(const float* dh_last, const void* saved, const float* const* w_ih, const float* const* w_hh,
                 float* const* dw_ih, float* const* dw_hh, float* const* db_ih, float* const* db_hh,
                 int Bn, int T, int F, int H, int layers, void* ws, size_t ws_bytes, ssv_stream_t stream);
/* Replaces GE2ELoss.forward, GE2E/speech_embedder_net.py:43-49 with GE2E/utils.py:16-55.
 * emb (N,M,D); w, b device scalars; loss[0] = total; per (N,M) per-embedding losses (may be NULL). */
size_t ssv_ge2e_loss_fwd_workspace(int N, int M, int D);
int ssv_ge2e_loss_fwd(const float* emb, const float* w, const float* b, float* loss, float* per,
                      int N, int M, int D, void* ws, size_t ws_bytes, ssv_stream_t stream);

/* Backward of the above (autograd of GE2ELoss.forward; GE2E/train_speech_embedder.py:82): dloss is the upstream gradient
 * of the scalar loss (DEVICE float, NULL = 1).  demb (N,M,D), dw, db (DEVICE scalars). */
size_t ssv_ge2e_loss_bwd_workspace(int N, int M, int D);
int ssv_ge2e_loss_bwd(const float* emb, const float* w, const float* b, const float* dloss, float* demb, float* dw, float* db,
                      int N, int M, int D, void* ws, size_t ws_bytes, ssv_stream_t stream);

/* ---- Vocoder and spectrogram front end ------------------------------------------------
 * Replaces the CPU tail of synthesis, synthesize.py:138-147 / generate_test_utterances.py:128-139
 * (`librosa.core.griffinlim(S, n_iter=64, hop_length, win_length)`, `signal.lfilter([1], [1, -PREEMPH], y)`, peak
 * normalisation) and the STFT front end of data/dataset.py:96-110, for a batch of equal-length utterances.
 * The DFTs are ssv_conv1d_fwd calls (k = 1) with windowed Fourier bases built by the host; these entries are the
 * steps between them.  Spectra are (B, 2F, T): rows [0,F) real, [F,2F) imaginary.  Frame matrices are (B, N, T):
 * row c holds sample c of every frame.  N = n_fft = 2(F-1), centred frames (librosa center=True, reflect padding),
 * so a T-frame spectrum is a waveform of hop*(T-1) samples.  inv_env: (N + hop*(T-1)) reciprocals of librosa's
 * window_sumsquare envelope (1 where the envelope vanishes). */
/* proj = mag * a / (|a| + 1e-16), a = reb - alpha*tprev (tprev NULL = 0): the Griffin-Lim phase update fused with S*angles. */
int ssv_gl_project(const float* mag, const float* reb, const float* tprev, float alpha, float* proj,
                   int B, int F, int T, ssv_stream_t stream);
/* mag = |spec| */
int ssv_complex_abs(const float* spec, float* mag, int B, int F, int T, ssv_stream_t stream);
/* istft overlap-add + envelope + centre trim, then stft reflect padding + framing, in one pass: windowed inverse frames
 * fr (B,N,T) -> analysis frames out (B,N,T) of the same waveform (out must not alias fr).  Needs hop*(T-1) > N/2. */
int ssv_ola_frames(const float* fr, const float* inv_env, float* out, int B, int N, int T, int hop, ssv_stream_t stream);
/* istft overlap-add + envelope + centre trim: fr (B,N,T) -> y (B, hop*(T-1)). */
int ssv_ola_signal(const float* fr, const float* inv_env, float* y, int B, int N, int T, int hop, ssv_stream_t stream);
/* stft framing: y (B,n) -> fr (B,N,T), T = 1 + n/hop, reflect padding by N/2 (n > N/2). */
int ssv_frame_signal(const float* y, float* fr, int B, int n, int N, int T, int hop, ssv_stream_t stream);
/* out[b] = max_i x[b][i];   y = (x / rowmax[b])^p * s   (synthesize.py:141-142,147; data/dataset.py:107-111). */
int ssv_rowmax(const float* x, float* out, int B, long n, ssv_stream_t stream);
int ssv_scale_pow(const float* x, const float* rowmax, float* y, float p, float s, int B, long n, ssv_stream_t stream);
/* y[n] = x[n] + a*y[n-1] per row, recurrence in double (scipy.signal.lfilter([1], [1, -a], x), synthesize.py:145). */
int ssv_deemphasis(const float* x, float* y, double a, int B, int n, ssv_stream_t stream);
/* y[0] = x[0], y[n] = x[n] - a*x[n-1] per row (data/dataset.py:96); y must not alias x. */
int ssv_preemphasis(const float* x, float* y, float a, int B, int n, ssv_stream_t stream);
/* LOG_FEATURE spectrograms (config.json:26-28).  y = exp(a*x + b): dB de-normalisation, 10^(dB/20) and the reconstruction
 * power of synthesize.py:133-135,142 folded into one pass.  y = clip((20 log10(max(1e-5, x)) - ref_db + max_db) / max_db,
 * 1e-8, 1): data/dataset.py:101-105. */
int ssv_exp_affine(const float* x, float* y, float a, float b, long n, ssv_stream_t stream);
int ssv_log_norm(const float* x, float* y, float ref_db, float max_db, long n, ssv_stream_t stream);

/* ---- Second order, for the critics' gradient penalty ------------------------------------------
 * train/adversarial_wasserstein_gp.py:300-308 differentiates the critic's input gradient
 * (`autograd.grad(..., create_graph=True)` then `loss.backward()`), so the LayerNorm / highway-gate BACKWARD kernels need
 * gradients of their own.  Critics only: at most 256 channels, no activation inside the LayerNorm.
 * ssv_channel_ln_bwd2: for dx = ssv_channel_ln_act_bwd(act = 0)(gn; x) and an upstream v (same shape as dx), the
 * gradients of <v, dx> w.r.t. gn (d_gn), x (d_x) and gamma (dgamma (C)); beta does not enter. */
size_t ssv_channel_ln_bwd2_workspace(int B, int C, int L);
int ssv_channel_ln_bwd2(const float* v, long v_bs, const float* gn, long gn_bs, const float* x, long x_bs, const float* stats,
                        const float* gamma, float* d_gn, long dgn_bs, float* d_x, long dx_bs, float* dgamma,
                        int B, int C, int L, void* ws, size_t ws_bytes, ssv_stream_t stream);
/* The highway gate alone (models/TTSModel_dropout.py:63-84 without the conv and the dropout, which the critics keep as
 * separate differentiable ops): h (B,2C,L) dense, y = sigmoid(LN1(h[:C])) * LN2(h[C:]) + (1 - sigmoid(..)) * x;
 * stats (B,4,L) saved for ssv_highway_gate_bwd. */
int ssv_highway_gate_fwd(const float* h, const float* x, long x_bs, const float* g1, const float* b1, const float* g2, const float* b2,
                         float* stats, float* y, long y_bs, float* y_amax, int B, int C, int L, ssv_stream_t stream);
/* For (dh, dxres) = ssv_highway_gate_bwd(gy; h, x) and upstreams vh (B,2C,L dense), vx: gradients of <vh, dh> + <vx, dxres>
 * w.r.t. gy (d_gy), h (d_h, dense), x (d_x) and pgrads (4,C) = dgamma1, dbeta1, dgamma2, dbeta2. */
size_t ssv_highway_gate_bwd2_workspace(int B, int C, int L);
int ssv_highway_gate_bwd2(const float* vh, const float* vx, long vx_bs, const float* gy, long gy_bs, const float* h, const float* x, long x_bs,
                          const float* stats, const float* g1, const float* b1, const float* g2, const float* b2,
                          float* d_gy, long dgy_bs, float* d_h, float* d_x, long dx_bs, float* pgrads,
                          int B, int C, int L, void* ws, size_t ws_bytes, ssv_stream_t stream);

/* ---- Column-incremental synthesis ------------------------------------------------------
 * The reference's free-running loop (synthesize.py:103-109, models/TTSModel.py:275-300) re-encodes the whole prefix at
 * every step.  The audio encoder and decoder are causal, so only column t of every layer is new at step t; these entries
 * compute that column.  Step activations are (B, C) matrices, channels contiguous; the input history of a causal k = 3
 * convolution is (B, Tmax, C); the frame counter is a DEVICE int so a captured step can be replayed for every frame.
 * ssv_column_matvec: out[b][m] = bias[m] + bias_b[b][m] + sum_{j,c} w[m][j][c] x_j[b][c] with w TAP-MAJOR, (M, k, C) =
 * nn.Conv1d's (M, C, k) weight with its last two axes swapped (the weights are frozen during synthesis: permute once; for
 * k = 1 the layouts coincide).  C a multiple of 4, buffers 16-byte aligned.  k = 1: x_0 = cur.  k = 3 (causal, `dilation`): x_2 = cur (frame t = *t_dev), x_1 = hist[t-d], x_0 = hist[t-2d]
 * (zero before frame 0); cur is also stored as column t of hist.  bias, bias_b, and (k = 1) hist / t_dev may be NULL. */
int ssv_column_matvec(const float* w, const float* bias, const float* bias_b, long bb_bs, const float* cur, long cur_bs,
                      float* hist, long hist_bs, int Tmax, const int* t_dev, int dilation, float* out, long out_bs,
                      int B, int C, int M, int k, ssv_stream_t stream);
/* LayerNorm over channels (+ activation, as ssv_channel_ln_act_fwd) and the highway gate (as ssv_highway_gate_fwd, h (B,2C)
 * dense) on one column per batch item. */
int ssv_column_ln_act(const float* x, long x_bs, const float* gamma, const float* beta, float* y, long y_bs, int B, int C, int act,
                      ssv_stream_t stream);
int ssv_column_gate(const float* h, const float* x, long x_bs, const float* g1, const float* b1, const float* g2, const float* b2,
                    float* y, long y_bs, int B, int C, ssv_stream_t stream);
/* One new attention frame, models/TTSModel.py:281-295: kv (B,2d,N) = K | V; q (B,d); pma (B) int64 is read and replaced by
 * the arg-max of the new column; column *t_dev of a (B,N,a_T) is written; rq (B,2d) = [V a ; q]. */
int ssv_attention_column(const float* kv, long kv_bs, const float* q, int64_t* pma, float* a, int a_T, const int* t_dev,
                         float* rq, int B, int d, int N, ssv_stream_t stream);
/* End of a step: Y[:, :, t] = y_cur (B,F), mel_cur = y_cur (the next step's input frame, synthesize.py:108-109), t += 1. */
int ssv_synth_column_advance(const float* y_cur, float* Y, float* mel_cur, int* t_dev, int B, int F, int T, ssv_stream_t stream);

/* ---- Critic glue: what models/discriminator.py:24-41 does between its convolutions and LayerNorms ----------
 * Dropout(p = 0.05) (always active: the reference never calls disc.eval()), leaky-ReLU(0.05), AvgPool1d, and the reduction of
 * the gradient penalty (train/adversarial_wasserstein_gp.py:305-308).  Each is (piecewise) linear, hence differentiable to any
 * order with ssv_mul / the pool adjoint.
 * ssv_act_dropout_fwd: y = leaky_relu(x, slope) * keep / (1 - p); d = y / x's factor, saved for ssv_mul.  slope = 1: plain
 * dropout, p = 0: plain leaky-ReLU.  The mask is Philox4x32-10 keyed by (seed, *ctr_dev); the call increments *ctr_dev on the
 * stream, so a replayed hipGraph draws a fresh mask every iteration. */
int ssv_act_dropout_fwd(const float* x, float* y, float* d, long n, float slope, float p, unsigned long long* ctr_dev, unsigned seed,
                        ssv_stream_t stream);
int ssv_mul(const float* x, const float* d, float* y, long n, ssv_stream_t stream);
/* nn.AvgPool1d(kernel_size = k) on rows of length L (rows = B * C): Lo = L / k outputs per row; the adjoint spreads dy / k. */
int ssv_avgpool1d_fwd(const float* x, float* y, long rows, int L, int k, ssv_stream_t stream);
int ssv_avgpool1d_bwd(const float* dy, float* dx, long rows, int L, int k, ssv_stream_t stream);
/* loss[0] = mean_b lam * (||g_b||_2 - 1)^2 for g (B, n); coef (B) = 2 lam (||g_b|| - 1) / (B ||g_b||) is saved for the backward
 * dg = gout[0] * coef[b] * g.  Fixed summation order. */
size_t ssv_grad_penalty_workspace(int B, long n);
int ssv_grad_penalty_fwd(const float* g, float* loss, float* coef, int B, long n, float lam, void* ws, size_t ws_bytes, ssv_stream_t stream);
int ssv_grad_penalty_bwd(const float* g, const float* coef, const float* gout, float* dg, int B, long n, ssv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SSV_HIP_H */
