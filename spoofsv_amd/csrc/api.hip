// extern "C" entry points of libssv_hip.so (see include/ssv_hip.h): argument checking and the
// composition of kernel launches for each replaced module of the reference.  No allocation, no
// synchronisation: everything is enqueued on the caller's stream.
#include <stdarg.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "ssv_common.h"
#include "bf3_tuning.h"
#include "../../include/ssv_hip.h"
#define SSV_HIP(expr) do { hipError_t _he = (expr); if (_he != hipSuccess) { ssv_fail(0, "%s: %s", #expr, hipGetErrorString(_he)); return -(int)_he; } } while (0)

// ---- error state ---------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
int ssv_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
int ssv_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return 0;
  ssv_fail(0, "%s: launch failed: %s", what, hipGetErrorString(e));
  return -(int)e;
}
static int g_precision = -1;
// SSV_PRECISION, strictly parsed: -> 0 | 1 | 2, or -1 when the variable names no mode (a typo must not silently select another arithmetic)
static int precision_from_env() {
  const char* e = getenv("SSV_PRECISION");
  if (!e || !*e || !strcmp(e, "f16x2") || !strcmp(e, "2")) return 2;
  if (!strcmp(e, "fp32") || !strcmp(e, "0")) return 0;
  if (!strcmp(e, "bf16x3") || !strcmp(e, "1")) return 1;
  return -1;
}
int ssv_precision() {
  if (g_precision < 0) {
    g_precision = precision_from_env();
    if (g_precision < 0) {
      // Reached only by a host that neither asked ssv_get_precision() (which reports the bad value as an error) nor chose a mode with
      // ssv_set_precision() before its first compute call: no arithmetic at all rather than another one than was asked for.
      fprintf(stderr, "libssv_hip: SSV_PRECISION=%s is not one of fp32|0, bf16x3|1, f16x2|2 (call ssv_set_precision to choose a mode)\n", getenv("SSV_PRECISION"));
      abort();
    }
  }
  return g_precision;
}
static const char* const g_knob_names[SSV_T_COUNT] = {"SSV_NT_FORCE", "SSV_NNB_FORCE", "SSV_LN_GROUPS", "SSV_LN_PERSIST", "SSV_LSTM_MERGE", "SSV_PWLN_BWD"};
static char g_knob_val[SSV_T_COUNT][512];
static const char* g_knob[SSV_T_COUNT];
static int g_knobs_loaded = 0;
static void load_knobs() {
  for (int i = 0; i < SSV_T_COUNT; ++i) {
    const char* e = getenv(g_knob_names[i]);
    if (e) { strncpy(g_knob_val[i], e, sizeof g_knob_val[i] - 1); g_knob_val[i][sizeof g_knob_val[i] - 1] = 0; g_knob[i] = g_knob_val[i]; }
    else g_knob[i] = nullptr;
  }
  __atomic_store_n(&g_knobs_loaded, 1, __ATOMIC_RELEASE);
}
const char* ssv_tuning(int knob) {
  if (!__atomic_load_n(&g_knobs_loaded, __ATOMIC_ACQUIRE)) load_knobs();
  return g_knob[knob];
}
extern "C" void ssv_reload_tuning(void) { load_knobs(); }
// ---- shape log (see ssv_common.h) ---------------------------------------------------------------
#include <mutex>
#include <string>
#include <map>
static int g_shape_log = -1;
static std::string g_shape_path;                                 // cached when the log is enabled: the environment may change before exit
static std::mutex g_shape_mu;
static std::map<std::string, long>* g_shape_seen = nullptr;      // line -> host-side launches (eager and capture passes; replays do not come here)
static void shape_log_flush() {
  std::lock_guard<std::mutex> lk(g_shape_mu);
  if (!g_shape_seen) return;
  if (g_shape_path.empty()) return;
  if (FILE* f = fopen(g_shape_path.c_str(), "w")) {
    for (const auto& kv : *g_shape_seen) fprintf(f, "%s\t%ld\n", kv.first.c_str(), kv.second);
    fclose(f);
  }
}
bool ssv_shape_log_on() {
  if (g_shape_log < 0) { const char* e = getenv("SSV_SHAPE_LOG"); g_shape_log = (e && *e) ? 1 : 0; if (g_shape_log) { g_shape_path = e; atexit(shape_log_flush); } }
  return g_shape_log == 1;
}
void ssv_shape_log(const char* kernel, dim3 grid, dim3 block, double flops, double bytes, const char* note) {
  if (!ssv_shape_log_on()) return;
  char line[768];
  snprintf(line, sizeof line, "%s\t%ux%ux%u\t%.6g\t%.6g\t%s", kernel, grid.x * block.x, grid.y * block.y, grid.z * block.z, flops, bytes, note ? note : "");
  std::lock_guard<std::mutex> lk(g_shape_mu);
  if (!g_shape_seen) g_shape_seen = new std::map<std::string, long>();
  ++(*g_shape_seen)[line];
}
// An unknown SSV_PRECISION value is an ERROR of these two entries (SSV_UNSUPPORTED + ssv_last_error), not the end of the process: a C host
// that asks for the mode first, or sets one explicitly (which overrides the variable), never reaches the abort in ssv_precision().
extern "C" int ssv_set_precision(int mode) {
  int prev = g_precision >= 0 ? g_precision : precision_from_env();
  if (prev < 0) prev = ssv_fail(SSV_UNSUPPORTED, "SSV_PRECISION=%s is not one of fp32|0, bf16x3|1, f16x2|2; mode %d set explicitly", getenv("SSV_PRECISION"), mode);
  g_precision = mode <= 0 ? 0 : (mode == 1 ? 1 : 2);
  return prev;
}
extern "C" int ssv_get_precision(void) {
  if (g_precision < 0 && precision_from_env() < 0)
    return ssv_fail(SSV_UNSUPPORTED, "SSV_PRECISION=%s is not one of fp32|0, bf16x3|1, f16x2|2", getenv("SSV_PRECISION"));
  return ssv_precision();
}
extern "C" int ssv_version(void) { return 7; }
extern "C" const char* ssv_arch(void) { return "gfx950"; }
extern "C" const char* ssv_last_error(void) { return g_err; }

// launchers defined in the other translation units
// last argument of the four LayerNorm launchers: where the kernel leaves its tiles' max |output| (ssv_amax_rows(L) per batch item), or null
int ssv_launch_ln_gate_fwd(const float*, long, const float*, long, const float*, const float*, const float*, const float*, float*, long, float*, int, int, int, hipStream_t, float* = nullptr);
int ssv_launch_ln_gate_bwd(const float*, long, const float*, const float*, long, const float*, const float*, const float*, const float*, const float*, float*, float*, long, float*, float*, int, int, int, hipStream_t, float* = nullptr);
int ssv_ln_gate_bwd_nblk(int B, int L);
int ssv_ln_gate_bwd_rows(int B, int C, int L, bool has_amax);      // partial rows the backward launch of this shape writes (norm.hip)
int ssv_ln_act_bwd_rows(int B, int C, int L, bool has_amax);
int ssv_ln_act_bwd_vec(int C, int L, bool has_amax);
int ssv_launch_ln_gate_fwd_stream(const float* H, const float* X, long x_bs, const float* colstats, const float* g1, const float* b1, const float* g2, const float* b2,
                                  float* Y, long y_bs, float* stats, float* amax, int B, int C, int L, hipStream_t st);
int ssv_launch_ln_bwd2(const float*, long, const float*, long, const float*, long, const float*, const float*, float*, long, float*, long, float*, float*, int, int, int, hipStream_t);
int ssv_reduce_partial_rows(const float* part, float* out, int n, int nblk, hipStream_t st);
int ssv_launch_ln_gate_bwd2(const float*, const float*, long, const float*, long, const float*, const float*, long, const float*, const float*, const float*, const float*,
                            const float*, float*, long, float*, float*, long, float*, float*, int, int, int, hipStream_t);
int ssv_launch_ln_act_fwd(const float*, long, const float*, const float*, float*, long, float*, int, int, int, int, hipStream_t, float* = nullptr);
int ssv_launch_ln_act_bwd(const float*, long, const float*, long, const float*, const float*, const float*, float*, long, float*, float*, int, int, int, int, hipStream_t, float* = nullptr);
int ssv_launch_softmax_cols(float*, int, int, int, hipStream_t);
int ssv_launch_softmax_cols_bwd(const float*, float*, const float*, float, int, int, int, hipStream_t);
int ssv_launch_lstm_in_transpose(const float*, float*, int, int, int, hipStream_t);
int ssv_launch_lstm_x_planes(const float*, const float*, void*, long, int, int, int, int, int, hipStream_t);
int ssv_launch_lstm_cell(const float*, float*, float*, int, int, int, hipStream_t);
int ssv_launch_lstm_cell_train(const float* pre, float* act, const float* cprev, float* c, float* h, int H, int Bn, hipStream_t st);
int ssv_launch_transpose_out(const float*, float*, int, int, hipStream_t);
int ssv_launch_l2norm_rows(const float*, float*, float*, int, int, hipStream_t);
int ssv_launch_l2norm_bwd(const float*, const float*, const float*, float*, int, int, hipStream_t);
int ssv_launch_colsum(const float*, float*, int, int, hipStream_t);
int ssv_launch_lstm_cell_bwd(const float*, const float*, const float*, long, int, const float*, float*, float*, float*, int, int, int, int, int, int, int, hipStream_t);

static inline size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

static inline size_t zmax(size_t a, size_t b) { return a > b ? a : b; }

static int conv_shifts(int k, int dilation, int causal, int* shift) {
  SSV_CHECK(k == 1 || k == 3, SSV_UNSUPPORTED, "conv1d: kernel_size %d not supported (1 or 3)", k);
  SSV_CHECK(dilation >= 1 && dilation * (k - 1) <= 54, SSV_UNSUPPORTED, "conv1d: dilation %d not supported (k=%d)", dilation, k);
  const int j0 = causal ? k - 1 : (k - 1) / 2;
  for (int j = 0; j < 3; ++j) shift[j] = j < k ? (j - j0) * dilation : 0;
  return 0;
}

static GemmNN nn_zero() {
  GemmNN g;
  g.A = nullptr; g.sab = g.sam = g.sac = g.saj = 0;
  g.X = nullptr; g.sxb = g.sxc = 0; g.sxn = 1; g.Lx = 0;
  g.C = nullptr; g.scb = g.scm = 0; g.scn = 1;
  g.bias = nullptr; g.bias_b = nullptr; g.sbb = 0;
  g.R = nullptr; g.srb = g.srm = 0; g.srn = 1;
  g.M = g.N = g.Kc = 0; g.KT = 1; g.B = 1;
  g.shift[0] = g.shift[1] = g.shift[2] = 0;
  g.alpha = 1.f;
  return g;
}
static GemmNT nt_zero() {
  GemmNT g;
  g.A = nullptr; g.sab = g.sam = 0; g.sat = 1; g.La = 0;
  g.X = nullptr; g.sxb = g.sxc = 0; g.sxn = 1; g.Lx = 0;
  g.C = nullptr; g.scz = g.scm = 0; g.scc = 1; g.scj = 0;
  g.M = g.Nc = 0; g.KT = 1; g.B = 1; g.Z = 1; g.bstep = 1;
  g.shift[0] = g.shift[1] = g.shift[2] = 0;
  g.jobs = nullptr; g.njobs = 0; g.max_shift = -1;
  g.f16 = 0; g.a_amax = g.x_amax = nullptr; g.a_namax = g.x_namax = 0;
  return g;
}

// ---- Conv1d ----------------------------------------------------------------------------------------
static inline int pad32(int n) { return (n + 31) & ~31; }
static inline size_t split_bytes(int rows, int K, int k) { return align256((size_t)k * ((rows + 15) / 16 * 16) * pad32(K) * sizeof(unsigned short)); }
// Channel counts below 32 (the tail of the WGAN-GP critics: 64 -> 16 -> 8 -> 1 channels, models/discriminator.py:31-38) take the exact-fp32
// kernels in all three products of a convolution: such a launch is 5 us either way, and the split-MFMA path would need a scale list per
// operand -- one ssv_absmax launch each for tensors whose producers (pooling, dropout, second-order LayerNorm) emit none (ops._tiny_conv).
#define SSV_MIN_SPLIT_CHANNELS 32
static inline bool use_bf3(int B, int L, int Cin, int Cout) {
  return ssv_precision() >= 1 && (long)B * L >= 128 && Cin >= SSV_MIN_SPLIT_CHANNELS && Cout >= SSV_MIN_SPLIT_CHANNELS;
}
static inline bool use_f16() { return ssv_precision() == 2; }

// ---- split-fp16 operand scales (ssv_common.h, "split-fp16") ------------------------------------------------------------
// A list of partial maxima of |x|: n entries per batch item, items consecutive.  Either the caller's (written by the kernel
// that produced x, or by ssv_absmax) or computed here into `fb`, SSV_AMAX_FB_FLOATS floats of the call's workspace.
#define SSV_AMAX_FB_FLOATS 4096
#define SSV_F16_AUX_BYTES (SSV_F16_AUX_FLOATS * sizeof(float))
static const size_t AMAX_FB_BYTES = SSV_AMAX_FB_FLOATS * sizeof(float);
struct AmaxList { const float* p; int n; };
static int amax_of(const float* x, long x_bs, int B, long n_item, const float* given, int ngiven, float* fb, AmaxList* out, hipStream_t st) {
  if (given) {
    SSV_CHECK(ngiven > 0, SSV_BAD_SHAPE, "operand scale list given with %d entries per item", ngiven);
    out->p = given; out->n = ngiven;
    return 0;
  }
  SSV_CHECK(fb, SSV_BAD_SHAPE, "split-fp16: no operand scales given and no workspace to compute them in");
  SSV_CHECK(B <= SSV_AMAX_FB_FLOATS, SSV_UNSUPPORTED, "split-fp16: batch %d needs caller-provided operand scales (ssv_absmax)", B);
  int npb = SSV_AMAX_FB_FLOATS / B;
  if (npb > 64) npb = 64;
  const long pieces = (n_item + 4095) / 4096;
  if (npb > pieces) npb = (int)(pieces > 0 ? pieces : 1);
  SSV_TRY(ssv_launch_absmax(x, x_bs, B, n_item, fb, npb, st));
  out->p = fb; out->n = npb;
  return 0;
}
extern "C" int ssv_amax_rows(int L) { return ssv_amax_rows_(L); }
extern "C" int ssv_absmax(const float* x, long x_bs, int B, long n, float* amax, int namax, ssv_stream_t stream) {
  SSV_CHECK(x && amax && B > 0 && B <= 65535 && n > 0 && namax > 0 && namax <= 65535, SSV_BAD_SHAPE, "absmax: bad argument");
  return ssv_launch_absmax(x, x_bs, B, n, amax, namax, (hipStream_t)stream);
}

static GemmNNB nnb_zero() {
  GemmNNB g;
  g.Ahi = g.Alo = nullptr; g.Kpad = 0;
  g.X = nullptr; g.sxb = g.sxc = 0; g.Lx = 0;
  g.C = nullptr; g.scb = g.scm = 0;
  g.bias = g.bias_b = nullptr; g.sbb = 0;
  g.R = nullptr; g.srb = g.srm = 0;
  g.M = g.N = g.Kc = 0; g.KT = 1; g.B = 1;
  g.shift[0] = g.shift[1] = g.shift[2] = 0;
  g.sxn = g.scn = 1;
  g.row_pair = 0; g.c_amax = nullptr; g.c_namax = 0;
  g.hs_planes = nullptr; g.hs_plane_bytes = 0; g.hs_npad = 0; g.hs_keep_h = 1;
  g.ksplit = 1; g.scz = 0; g.skip_rows = 0;
  g.x0_planes = nullptr; g.x0_amax = nullptr; g.xsplit0 = 0;
  g.perm_h = g.epi = g.first = 0; g.cstate = nullptr;
  g.lstm_out = nullptr; g.lstm_s = g.lstm_lo = g.lstm_D = g.xsplit = 0; g.sab = 0; g.A0hi = g.A0lo = nullptr;
  g.gates_out = nullptr;
  g.f16 = 0; g.a_inv = nullptr; g.x_amax = nullptr; g.x_namax = 0; g.x_amax_bs = 0;
  g.colstats = nullptr;
  g.xrow_w = nullptr; g.xrow_sk = 0;
  return g;
}

// y = conv(x, w): shared by forward (rows = Cout) and data gradient (rows = Cin, transposed weights, negated shifts)
// `packed`: resident pre-split planes of this operand (hi plane, then lo plane), or null -> split into ws here.
// split-fp16: `a_inv` = where the resident planes keep 2^-ea (packed only); xa_given / xa_n = the caller's scale list of x or
// null; the tail of ws (conv_aux_bytes: after `ws_main` bytes) holds the pack kernel's aux floats and the fallback list.
static inline size_t conv_aux_bytes() { return SSV_F16_AUX_BYTES + AMAX_FB_BYTES; }
// (pw != null: the 1x1 product finishes LayerNorm + activation in its own launch, gemm_pwln_kernel; y is then `pre`)
struct PwLnArgs { const float* gamma; const float* beta; float* y; long ybs; float* stats; float* y_amax; int namax; int act; };
static int conv_nn(const float* x, long x_bs, const float* w, const void* packed, long w_sm, long w_sk, const float* bias, const float* bias_b,
                   const float* r, long r_bs, float* y, long y_bs, int B, int K, int M, int L, int k, const int* shift,
                   bool bf3, void* ws, hipStream_t st, const float* a_inv = nullptr, const float* xa_given = nullptr, int xa_n = 0, size_t ws_main = 0,
                   float* colstats = nullptr, const PwLnArgs* pw = nullptr) {
  if (bf3) {
    const int Kpad = pad32(K);
    const unsigned short* hi = (const unsigned short*)(packed ? packed : ws);
    const unsigned short* lo = (const unsigned short*)((const char*)hi + split_bytes(M, K, k));
    const bool f16 = use_f16();
    float* aux = ws ? (float*)((char*)ws + ws_main) : nullptr;
    if (!packed) {
      if (f16) { SSV_TRY(ssv_launch_pack_split_f16(w, (long)M * K * k, (void*)hi, (void*)lo, M, K, Kpad, k, w_sm, w_sk, 1, aux, st)); a_inv = aux + 64; }
      else SSV_TRY(ssv_launch_pack_split(w, (void*)hi, (void*)lo, M, K, Kpad, k, w_sm, w_sk, 1, 0, st));
    }
    GemmNNB g = nnb_zero();
    if (f16) {
      AmaxList xa;
      SSV_TRY(amax_of(x, x_bs, B, (long)K * L, xa_given, xa_n, aux ? aux + SSV_F16_AUX_FLOATS : nullptr, &xa, st));
      g.f16 = 1; g.a_inv = a_inv; g.x_amax = xa.p; g.x_namax = xa.n; g.x_amax_bs = xa.n;
    }
    g.colstats = colstats;
    g.Ahi = hi; g.Alo = lo; g.Kpad = Kpad;
    g.X = x; g.sxb = x_bs; g.sxc = L; g.Lx = L;
    g.C = y; g.scb = y_bs; g.scm = L;
    g.bias = bias; g.bias_b = bias_b; g.sbb = M;
    g.R = r; g.srb = r_bs; g.srm = L;
    g.M = M; g.N = L; g.Kc = K; g.KT = k; g.B = B;
    for (int j = 0; j < 3; ++j) g.shift[j] = shift[j];
    if (k == 1 && M > 128 && M % 128 == 1) { g.xrow_w = w + (long)(M - 1) * w_sm; g.xrow_sk = w_sk; }     // (GemmNNB::xrow_w; the launchers decide)
    if (pw) return ssv_launch_gemm_pwln(g, pw->gamma, pw->beta, pw->y, pw->ybs, pw->stats, pw->y_amax, pw->namax, pw->act, st);
    return ssv_launch_gemm_nn_bf3(g, st);
  }
  if (L == 1 && k == 1 && w_sk == 1 && w_sm == K && !r)          // nn.Linear on a (B, K) matrix (the speaker-code layers): see linear_len1_fwd_kernel
    return ssv_launch_linear_len1_fwd(x, x_bs, w, bias, bias_b, M, y, y_bs, B, K, M, st);
  GemmNN g = nn_zero();
  for (int j = 0; j < 3; ++j) g.shift[j] = shift[j];
  const float* a = w;
  if (w_sk != k) {                                   // transposed operand for the data gradient: wt[c][o][j] = w[o][c][j]
    SSV_TRY(ssv_launch_pack_wt(w, (float*)ws, K, M, k, st));
    a = (const float*)ws;
  }
  g.A = a; g.sam = (long)K * k; g.sac = k; g.saj = 1;
  g.X = x; g.sxb = x_bs; g.sxc = L; g.Lx = L;
  g.C = y; g.scb = y_bs; g.scm = L;
  g.bias = bias; g.bias_b = bias_b; g.sbb = M;
  if (r) { g.R = r; g.srb = r_bs; g.srm = L; }
  g.M = M; g.N = L; g.Kc = K; g.KT = k; g.B = B;
  return ssv_launch_gemm_nn(g, st);
}

// where the resident planes of a (Cout, Cin, k) weight keep 2^-ea of the forward / transposed planes (split-fp16)
static size_t pack_bytes(int Cout, int Cin, int k) { return 2 * split_bytes(Cout, Cin, k) + 2 * split_bytes(Cin, Cout, k) + 256; }
static const float* packed_inv(const void* w_packed, int Cout, int Cin, int k, int transposed) {
  return w_packed ? (const float*)((const char*)w_packed + pack_bytes(Cout, Cin, k) - 256 + (transposed ? 128 : 0)) : nullptr;
}
extern "C" size_t ssv_conv1d_fwd_workspace(int Cin, int Cout, int k) { return 2 * split_bytes(Cout, Cin, k) + conv_aux_bytes(); }
extern "C" int ssv_conv1d_fwd(const float* x, long x_bs, const float* x_amax, int x_namax, const float* w, const void* w_packed, const float* bias,
                              const float* bias_b, float* y, long y_bs, float* y_colstats,
                              int B, int Cin, int Cout, int L, int k, int dilation, int causal, void* ws, size_t ws_bytes,
                              ssv_stream_t stream) {
  SSV_CHECK(x && w && y && B > 0 && Cin > 0 && Cout > 0 && L > 0, SSV_BAD_SHAPE, "conv1d_fwd: bad argument B=%d Cin=%d Cout=%d L=%d", B, Cin, Cout, L);
  SSV_CHECK(x_bs >= (long)Cin * L && y_bs >= (long)Cout * L, SSV_BAD_SHAPE, "conv1d_fwd: batch stride smaller than C*L");
  int shift[3];
  SSV_TRY(conv_shifts(k, dilation, causal, shift));
  const bool bf3 = use_bf3(B, L, Cin, Cout);
  if (bf3 && (!w_packed || (use_f16() && !x_amax)))
    SSV_CHECK(ws && ws_bytes >= ssv_conv1d_fwd_workspace(Cin, Cout, k), SSV_BAD_SHAPE, "conv1d_fwd: workspace too small");
  SSV_CHECK(!y_colstats || (bf3 && Cout % 64 == 0 && y_bs == (long)Cout * L), SSV_UNSUPPORTED,
            "conv1d_fwd: column statistics need a split-MFMA mode, Cout %% 64 == 0 and a dense output (Cout=%d)", Cout);
  return conv_nn(x, x_bs, w, w_packed, (long)Cin * k, k, bias, bias_b, nullptr, 0, y, y_bs, B, Cin, Cout, L, k, shift, bf3, ws, (hipStream_t)stream,
                 packed_inv(w_packed, Cout, Cin, k, 0), x_amax, x_namax, 2 * split_bytes(Cout, Cin, k), y_colstats);
}

static size_t bwd_data_main(int Cin, int Cout, int k) {
  const size_t a = align256((size_t)Cin * Cout * k * sizeof(float)), b = 2 * split_bytes(Cin, Cout, k);
  return a > b ? a : b;
}
extern "C" size_t ssv_conv1d_bwd_data_workspace(int Cin, int Cout, int k) { return bwd_data_main(Cin, Cout, k) + conv_aux_bytes(); }
extern "C" int ssv_conv1d_bwd_data(const float* dy, long dy_bs, const float* dy_amax, int dy_namax, const float* w, const void* w_packed,
                                   const float* dx_add, float* dx, long dx_bs,
                                   int B, int Cin, int Cout, int L, int k, int dilation, int causal,
                                   void* ws, size_t ws_bytes, ssv_stream_t stream) {
  SSV_CHECK(dy && w && dx && B > 0 && Cin > 0 && Cout > 0 && L > 0, SSV_BAD_SHAPE, "conv1d_bwd_data: bad argument");
  SSV_CHECK((w_packed && !(use_f16() && !dy_amax)) || (ws && ws_bytes >= ssv_conv1d_bwd_data_workspace(Cin, Cout, k)), SSV_BAD_SHAPE,
            "conv1d_bwd_data: workspace too small");
  int shift[3];
  SSV_TRY(conv_shifts(k, dilation, causal, shift));
  for (int j = 0; j < 3; ++j) shift[j] = -shift[j];
  // rows = input channels c, reduction over output channels o: element (c, o, j) = w[o][c][j]
  // the transposed planes follow the forward planes in a resident buffer (ssv_conv_pack_bytes)
  const void* pk = w_packed ? (const char*)w_packed + 2 * split_bytes(Cout, Cin, k) : nullptr;
  return conv_nn(dy, dy_bs, w, pk, k, (long)Cin * k, nullptr, nullptr, dx_add, dx_bs, dx, dx_bs, B, Cout, Cin, L, k, shift,
                 use_bf3(B, L, Cout, Cin), ws, (hipStream_t)stream, packed_inv(w_packed, Cout, Cin, k, 1), dy_amax, dy_namax, bwd_data_main(Cin, Cout, k));
}

// ---- resident pre-split weights ------------------------------------------------------------------------------------
extern "C" size_t ssv_conv_pack_bytes(int Cout, int Cin, int k) { return pack_bytes(Cout, Cin, k); }
extern "C" int ssv_conv_pack_plan(int n, const float* const* w, void* const* planes, const int* Cout, const int* Cin, const int* k,
                                  ssv_pack_job* jobs) {
  SSV_CHECK(n > 0 && w && planes && Cout && Cin && k && jobs, SSV_BAD_SHAPE, "conv_pack_plan: bad argument");
  long blocks = 0;
  for (int i = 0; i < n; ++i) {
    SSV_CHECK(w[i] && planes[i] && Cout[i] > 0 && Cin[i] > 0 && (k[i] == 1 || k[i] == 3), SSV_BAD_SHAPE, "conv_pack_plan: weight %d: bad shape", i);
    for (int tr = 0; tr < 2; ++tr) {
      ssv_pack_job& j = jobs[2 * i + tr];
      const int M = tr ? Cin[i] : Cout[i], K = tr ? Cout[i] : Cin[i];
      j.w = w[i];
      j.planes = (char*)planes[i] + (tr ? 2 * split_bytes(Cout[i], Cin[i], k[i]) : 0);
      j.M = M; j.K = K; j.Kpad = pad32(K); j.KT = k[i];
      j.sm = tr ? k[i] : (long)Cin[i] * k[i];            // element (m, kk, tap) = w[m*sm + kk*sk + tap]
      j.sk = tr ? (long)Cin[i] * k[i] : k[i];
      j.first_block = (int)blocks; j.pad_ = 0;
      j.inv_out = (float*)((char*)planes[i] + pack_bytes(Cout[i], Cin[i], k[i]) - 256 + (tr ? 128 : 0));
      blocks += ssv_pack_job_blocks(j);
      SSV_CHECK(blocks < (1L << 30), SSV_UNSUPPORTED, "conv_pack_plan: too many elements");
    }
  }
  return (int)blocks;
}
extern "C" size_t ssv_conv_pack_multi_workspace(int njobs) { return align256((size_t)(njobs / 2) * SSV_PACK_AMAX_PER_WEIGHT * sizeof(float)); }
extern "C" int ssv_conv_pack_multi(const ssv_pack_job* jobs_dev, int njobs, int nblocks, void* ws, size_t ws_bytes, ssv_stream_t stream) {
  SSV_CHECK(jobs_dev && njobs > 0 && njobs % 2 == 0 && nblocks > 0, SSV_BAD_SHAPE, "conv_pack_multi: bad argument");
  const bool f16 = use_f16();
  SSV_CHECK(!f16 || (ws && ws_bytes >= ssv_conv_pack_multi_workspace(njobs)), SSV_BAD_SHAPE, "conv_pack_multi: workspace too small");
  return ssv_launch_pack_multi(jobs_dev, njobs, nblocks, f16 ? (float*)ws : nullptr, (hipStream_t)stream);
}

// Number of batch slabs Z a weight-gradient launch is cut into (njobs layers x output tiles x Z workgroups, each reducing over
// ceil(B / Z) batch items).  The launch lasts  rounds x (items per workgroup) x (time per item)  +  Z x (slab write + read back),
// rounds = ceil(workgroups / co-resident slots): the slots are few (2 workgroups per CU for the 128 x 64 x 3 tile = 512), so the
// count is a matter of wave quantisation -- 640 workgroups take TWO rounds of which the second runs a quarter full (the ten
// C = 512 / L = 186 layers of the text encoder with Z = 1: 905 us; Z = 4 -> 2,560 workgroups, 5 full rounds of 8 items: 660 us by
// this model).  Round 2 aimed at "about 512 workgroups" whatever the remainder.  L = 0: length unknown (325 assumed).
static int nt_slabs(long tiles_all, int njobs, int B, int L, int kt, int M, int Nc) {
  int wm, ntc;
  ssv_nt_bf3_tile(kt, M, Nc, &wm, &ntc);
  const int per_cu = ssv_nt_bf3_wg_per_cu(kt, wm, ntc);
  const long slots = 256L * per_cu;
  // per workgroup and batch item: 2 x (64 wm) x (16 ntc) x kt x L flop at ~0.55 TFLOP/s per resident workgroup (2 per CU; scaled
  // when more fit); per slab and job: the output written and read back at ~4 TB/s
  const double t_item = 2.0 * 64 * wm * 16 * ntc * kt * L / (0.55e12 * 2.0 / per_cu);
  const double t_slab = 8.0 * (double)M * Nc * kt / 4e12 * njobs;
  int best = 1;
  double best_t = 1e30;
  for (int z = 1; z <= B && z <= 64; ++z) {
    const long rounds = (tiles_all * z + slots - 1) / slots;
    const double t = (double)rounds * ssv_cdiv(B, z) * t_item + (z > 1 ? z * t_slab : 0.0);
    if (t < best_t * 0.98) { best_t = t; best = z; }          // ties and near-ties: the smaller count
  }
  return best;
}
// range slabs of the extra-row kernel (ssv_nt_bf3_xrow): as many slabs as fill the co-resident slots once -- every workgroup then reduces over
// the same number of 64-step chunks, not over a whole number of batch items
static int xrow_slabs(long tiles_all, int B, int L) {
  const long chunks = (long)B * ssv_cdiv(L, 64);
  long z = 512 / (tiles_all > 0 ? tiles_all : 1);
  if (z > chunks) z = chunks;
  if (z > 64) z = 64;
  return (int)(z < 1 ? 1 : z);
}
// the predicate of the split-MFMA weight-gradient launch (conv1d_bwd_weight_impl), for dense operands: what the workspace queries can know
static bool nt_bf3_runs(int B, int M, int Nc, int L) {
  if (ssv_precision() < 1 || (long)B * L < 256 || Nc < SSV_MIN_SPLIT_CHANNELS || M < SSV_MIN_SPLIT_CHANNELS) return false;
  GemmNT g = nt_zero();
  g.sab = (long)M * L; g.sam = L; g.La = L; g.sxb = (long)Nc * L; g.sxc = L; g.Lx = L;
  g.M = M; g.Nc = Nc; g.KT = 1; g.B = B; g.Z = 1; g.bstep = 1;
  return ssv_nt_bf3_fits(g);
}
static int nt_force(int z, int M, int Nc, int k) {
  if (const char* e = ssv_tuning(SSV_T_NT_FORCE)) {      // "M:Nc:k=Z;..." -- one shape's slab count inside a whole step (tools/sweep_force.sh)
    char key[48];
    snprintf(key, sizeof key, "%d:%d:%d=", M, Nc, k);
    const char* hit = strstr(e, key);
    if (hit && (hit == e || hit[-1] == ';')) { const int v = atoi(hit + strlen(key)); if (v > 0) z = v; }
  }
  return z;
}
// L: the reduction length per batch item.  Required: a caller whose workspace query has no length (the transposed conv) passes
// SSV_NOMINAL_L so that query and launch agree by construction.
#define SSV_NOMINAL_L 325
static int dw_splits(int B, int M, int Nc, int k, int L) {
  const int tiles = ssv_nt_bf3_tiles(k == 3 ? 3 : 1, M, Nc);
  // range slabs only when the extra-row kernel will really run (same predicate as the launch): the fp32 fallback cuts whole-item slabs, Z <= B
  if (k != 3 && ssv_nt_bf3_xrow(1, M, Nc) && nt_bf3_runs(B, M, Nc, L)) {
    int z = nt_force(xrow_slabs(tiles, B, L), M, Nc, k);
    const long chunks = (long)B * ssv_cdiv(L, 64);
    if (z > chunks) z = (int)chunks;
    return z < 1 ? 1 : z;
  }
  int z = nt_force(nt_slabs(tiles, 1, B, L, k == 3 ? 3 : 1, M, Nc), M, Nc, k);
  if (z > B) z = B;
  if (z < 1) z = 1;
  return z;
}
static size_t bwd_weight_main(int B, int Cin, int Cout, int L, int k) { return align256((size_t)dw_splits(B, Cout, Cin, k, L) * Cout * Cin * k * sizeof(float)); }
extern "C" size_t ssv_conv1d_bwd_weight_workspace(int B, int Cin, int Cout, int L, int k) { return bwd_weight_main(B, Cin, Cout, L, k) + 2 * AMAX_FB_BYTES; }
// part / pgrads / n2 / nblk: partial rows of another reduction (the LayerNorm / bias gradients of the same layer) summed by the
// SAME launch that sums the weight-gradient slabs (highwayConv backward); part == nullptr: weight gradient only.
// dy_amax / x_amax (n entries per batch item each): the operands' scale lists for the split-fp16 arithmetic, or null (computed here).
static int conv1d_bwd_weight_impl(const float* dy, long dy_bs, const float* x, long x_bs, float* dw, int B, int Cin, int Cout, int L, int k, int dilation,
                                  int causal, void* ws, size_t ws_bytes, ssv_stream_t stream, const float* part, float* pgrads, int n2, int nblk,
                                  const float* dy_amax, int dy_namax, const float* x_amax, int x_namax);
extern "C" int ssv_conv1d_bwd_weight(const float* dy, long dy_bs, const float* dy_amax, int dy_namax, const float* x, long x_bs, const float* x_amax, int x_namax,
                                     float* dw, int B, int Cin, int Cout, int L, int k, int dilation, int causal,
                                     void* ws, size_t ws_bytes, ssv_stream_t stream) {
  return conv1d_bwd_weight_impl(dy, dy_bs, x, x_bs, dw, B, Cin, Cout, L, k, dilation, causal, ws, ws_bytes, stream, nullptr, nullptr, 0, 0,
                                dy_amax, dy_namax, x_amax, x_namax);
}
static int conv1d_bwd_weight_impl(const float* dy, long dy_bs, const float* x, long x_bs, float* dw, int B, int Cin, int Cout, int L, int k, int dilation,
                                  int causal, void* ws, size_t ws_bytes, ssv_stream_t stream, const float* part, float* pgrads, int n2, int nblk,
                                  const float* dy_amax, int dy_namax, const float* x_amax, int x_namax) {
  SSV_CHECK(dy && x && dw && B > 0 && Cin > 0 && Cout > 0 && L > 0, SSV_BAD_SHAPE, "conv1d_bwd_weight: bad argument");
  SSV_CHECK(ws && ws_bytes >= ssv_conv1d_bwd_weight_workspace(B, Cin, Cout, L, k), SSV_BAD_SHAPE, "conv1d_bwd_weight: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  GemmNT g = nt_zero();
  SSV_TRY(conv_shifts(k, dilation, causal, g.shift));
  int Z = dw_splits(B, Cout, Cin, k, L);
  const long n = (long)Cout * Cin * k;
  g.A = dy; g.sab = dy_bs; g.sam = L; g.La = L;
  g.X = x; g.sxb = x_bs; g.sxc = L; g.Lx = L;
  g.M = Cout; g.Nc = Cin; g.KT = k; g.B = B;
  const bool bf3 = ssv_precision() >= 1 && (long)B * L >= 256 && Cin >= SSV_MIN_SPLIT_CHANNELS && Cout >= SSV_MIN_SPLIT_CHANNELS && ssv_nt_bf3_fits(g);
  if (!bf3 && Z > B) Z = B;        // (range-slab count chosen for dense operands, strided ones do not fit the split kernel: whole-item slabs, no empty ones)
  if (Z == 1) { g.C = dw; g.scz = n; g.scm = (long)Cin * k; g.scc = k; g.scj = 1; }
  else { g.C = (float*)ws; g.scz = n; g.scm = (long)Cin * k; g.scc = 1; g.scj = Cin; }     // slabs [z][m][j][c]
  g.Z = Z; g.bstep = Z;
  if (bf3) {
    if (k != 3 && ssv_nt_bf3_xrow(1, Cout, Cin)) g.bstep = 0;            // range slabs (dw_splits chose Z for them; any Z <= chunks is valid)
    if (use_f16()) {
      float* fb = (float*)((char*)ws + bwd_weight_main(B, Cin, Cout, L, k));
      AmaxList la, lx;
      SSV_TRY(amax_of(dy, dy_bs, B, (long)Cout * L, dy_amax, dy_namax, fb, &la, st));
      SSV_TRY(amax_of(x, x_bs, B, (long)Cin * L, x_amax, x_namax, fb + SSV_AMAX_FB_FLOATS, &lx, st));
      g.f16 = 1; g.a_amax = la.p; g.a_namax = la.n * B; g.x_amax = lx.p; g.x_namax = lx.n * B;
    }
    SSV_TRY(ssv_launch_gemm_nt_bf3(g, st));
  } else if (L == 1 && k == 1) {                       // see linear_len1_wgrad_kernel; writes dw itself, whatever Z says
    SSV_TRY(ssv_launch_linear_len1_wgrad(dy, dy_bs, x, x_bs, dw, B, Cin, Cout, st));
    if (part) SSV_TRY(ssv_reduce_partial_rows(part, pgrads, n2, nblk, st));
    return 0;
  } else SSV_TRY(ssv_launch_gemm_nt(g, st));
  if (part) {
    if (Z > 1 && nblk <= 768) return ssv_launch_reduce_pair((const float*)ws, dw, Cout, Cin, k, Z, part, pgrads, n2, nblk, st);
    SSV_TRY(ssv_reduce_partial_rows(part, pgrads, n2, nblk, st));
  }
  if (Z > 1) SSV_TRY(ssv_launch_reduce_slabs_perm((const float*)ws, dw, Cout, Cin, k, Z, st));
  return 0;
}

// ---- several equal-shaped weight gradients in one launch (see include/ssv_hip.h) ------------------------------------------
extern "C" int ssv_conv_shifts(int k, int dilation, int causal, int* shift3) { return conv_shifts(k, dilation, causal, shift3); }
extern "C" int ssv_conv1d_bwd_weight_multi_ok(int B, int Cin, int Cout, int L, int k) {
  if (ssv_precision() < 1 || (k != 1 && k != 3) || (long)B * L < 256 || L < 8 || Cin < SSV_MIN_SPLIT_CHANNELS || Cout < SSV_MIN_SPLIT_CHANNELS) return 0;
  GemmNT g = nt_zero();
  g.sab = (long)Cout * L; g.sam = L; g.La = L; g.sxb = (long)Cin * L; g.sxc = L; g.Lx = L;
  g.M = Cout; g.Nc = Cin; g.KT = k; g.B = B; g.Z = 1; g.bstep = 1;
  return ssv_nt_bf3_fits(g) ? 1 : 0;
}
extern "C" int ssv_conv1d_bwd_weight_multi_splits(int njobs, int B, int Cin, int Cout, int L, int k) {
  const int kt = k == 3 ? 3 : 1;
  if (njobs < 1) njobs = 1;
  const long tiles = (long)ssv_nt_bf3_tiles(kt, Cout, Cin) * njobs;
  if (kt == 1 && ssv_nt_bf3_xrow(1, Cout, Cin)) return xrow_slabs(tiles, B, L);
  int z = nt_slabs(tiles, njobs, B, L, kt, Cout, Cin);
  if (z > B) z = B;
  if (z < 1) z = 1;
  return z;
}
extern "C" size_t ssv_conv1d_bwd_weight_multi_workspace(int njobs, int B, int Cin, int Cout, int L, int k) {
  return align256((size_t)njobs * ssv_conv1d_bwd_weight_multi_splits(njobs, B, Cin, Cout, L, k) * Cout * Cin * k * sizeof(float));
}
extern "C" int ssv_conv1d_bwd_weight_multi(const ssv_wgrad_job* jobs_dev, int njobs, long dy_bs, long x_bs, int B, int Cin, int Cout, int L, int k, int max_shift,
                                           int n2, int nblk, void* ws, size_t ws_bytes, ssv_stream_t stream) {
  SSV_CHECK(jobs_dev && njobs > 0 && B > 0 && Cin > 0 && Cout > 0 && L > 0, SSV_BAD_SHAPE, "conv1d_bwd_weight_multi: bad argument");
  SSV_CHECK(ssv_conv1d_bwd_weight_multi_ok(B, Cin, Cout, L, k), SSV_UNSUPPORTED, "conv1d_bwd_weight_multi: shape or arithmetic mode not supported");
  SSV_CHECK(n2 == 0 || nblk <= 768, SSV_UNSUPPORTED, "conv1d_bwd_weight_multi: %d partial rows (max 768)", nblk);
  SSV_CHECK(ws && ws_bytes >= ssv_conv1d_bwd_weight_multi_workspace(njobs, B, Cin, Cout, L, k), SSV_BAD_SHAPE, "conv1d_bwd_weight_multi: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int Z = ssv_conv1d_bwd_weight_multi_splits(njobs, B, Cin, Cout, L, k);
  const long n = (long)Cout * Cin * k;
  GemmNT g = nt_zero();
  g.A = nullptr; g.sab = dy_bs; g.sam = L; g.La = L;
  g.X = nullptr; g.sxb = x_bs; g.sxc = L; g.Lx = L;
  g.C = (float*)ws; g.scz = n; g.scm = (long)Cin * k; g.scc = 1; g.scj = Cin;                 // slabs [job][z][m][j][c]
  g.M = Cout; g.Nc = Cin; g.KT = k; g.B = B; g.Z = Z; g.bstep = (k != 3 && ssv_nt_bf3_xrow(1, Cout, Cin)) ? 0 : Z;      // (0: range slabs)
  g.jobs = jobs_dev; g.njobs = njobs; g.max_shift = max_shift;
  g.f16 = use_f16() ? 1 : 0;                     // the jobs carry their operands' scale lists (the caller saw to that)
  SSV_TRY(ssv_launch_gemm_nt_bf3(g, st));
  return ssv_launch_reduce_pair_multi(jobs_dev, njobs, (const float*)ws, Cout, Cin, k, Z, n2, nblk, st);
}

// ---- LayerNorm over channels ------------------------------------------------------------------------
extern "C" size_t ssv_channel_ln_act_fwd_workspace(int B, int C, int L) { (void)B; (void)C; (void)L; return 256; }   // none needed; kept in the ABI
extern "C" int ssv_channel_ln_act_fwd(const float* x, long x_bs, const float* gamma, const float* beta, float* y, long y_bs, float* y_amax, float* stats,
                                      int B, int C, int L, int act, void* ws, size_t ws_bytes, ssv_stream_t stream) {
  SSV_CHECK(x && gamma && beta && y && B > 0 && C > 0 && L > 0 && act >= 0 && act <= 2, SSV_BAD_SHAPE, "channel_ln_act_fwd: bad argument");
  SSV_CHECK(B <= 65535, SSV_UNSUPPORTED, "channel_ln_act_fwd: batch %d exceeds grid.y", B);
  (void)ws; (void)ws_bytes;
  return ssv_launch_ln_act_fwd(x, x_bs, gamma, beta, y, y_bs, stats, B, C, L, act, (hipStream_t)stream, y_amax);
}
// ---- 1x1 conv + LayerNorm (+ activation), forward ---------------------------------------------------------------------------
// y = act(LN(conv1x1(x) + bias [+ s])) with pre = the LayerNorm's input and stats (B,2,L) = mean / rstd per column kept for the backward.
// One launch (gemm_pwln_kernel: a workgroup owns all output rows of its column tile and finishes the LayerNorm from its accumulators)
// where that form is the faster one in-step, else the product followed by the LayerNorm kernel -- same results up to summation order.
// (round 4, in-step and same box, fused against product + LayerNorm kernel: 513 x 1300 139.8 against 111.7 + 43.6 us; M = 256 / N = 325 29.1 against 19.5 + 12.5;
//  M = 512 / N = 186 33.9 against 29.1 + ~13; M = 512 / N = 1300 90.9 against 70.2 + 24: the fused form everywhere the split-MFMA kernels run)
static bool pwln_fused(int B, int Cin, int Cout, int L) { return use_bf3(B, L, Cin, Cout) && Cout <= 640 && B <= 65535; }
extern "C" size_t ssv_pointwise_conv_ln_act_fwd_workspace(int Cin, int Cout) { return ssv_conv1d_fwd_workspace(Cin, Cout, 1); }
extern "C" int ssv_pointwise_conv_ln_act_fwd(const float* x, long x_bs, const float* x_amax, int x_namax, const float* w, const void* w_packed, const float* bias,
                                             const float* s, const float* gamma, const float* beta, float* pre, float* stats, float* y, long y_bs, float* y_amax,
                                             int B, int Cin, int Cout, int L, int act, void* ws, size_t ws_bytes, ssv_stream_t stream) {
  SSV_CHECK(x && w && gamma && beta && pre && y && B > 0 && Cin > 0 && Cout > 0 && L > 0 && act >= 0 && act <= 2, SSV_BAD_SHAPE, "pointwise_conv_ln_act_fwd: bad argument");
  SSV_CHECK(x_bs >= (long)Cin * L && y_bs >= (long)Cout * L, SSV_BAD_SHAPE, "pointwise_conv_ln_act_fwd: batch stride smaller than C*L");
  if (pwln_fused(B, Cin, Cout, L) && y_bs == (long)Cout * L && (!y_amax || ssv_amax_rows_(L) >= ssv_cdiv(L, 64))) {
    SSV_CHECK((w_packed && !(use_f16() && !x_amax)) || (ws && ws_bytes >= ssv_pointwise_conv_ln_act_fwd_workspace(Cin, Cout)), SSV_BAD_SHAPE,
              "pointwise_conv_ln_act_fwd: workspace too small");
    int shift[3] = {0, 0, 0};
    PwLnArgs pw = {gamma, beta, y, y_bs, stats, y_amax, ssv_amax_rows_(L), act};
    return conv_nn(x, x_bs, w, w_packed, (long)Cin, 1, bias, s, nullptr, 0, pre, (long)Cout * L, B, Cin, Cout, L, 1, shift, true, ws, (hipStream_t)stream,
                   packed_inv(w_packed, Cout, Cin, 1, 0), x_amax, x_namax, 2 * split_bytes(Cout, Cin, 1), nullptr, &pw);
  }
  SSV_TRY(ssv_conv1d_fwd(x, x_bs, x_amax, x_namax, w, w_packed, bias, s, pre, (long)Cout * L, nullptr, B, Cin, Cout, L, 1, 1, 0, ws, ws_bytes, stream));
  return ssv_channel_ln_act_fwd(pre, (long)Cout * L, gamma, beta, y, y_bs, y_amax, stats, B, Cout, L, act, nullptr, 0, stream);
}
extern "C" size_t ssv_channel_ln_act_bwd_workspace(int B, int C, int L) {
  return align256((size_t)ssv_ln_gate_bwd_nblk(B, L) * 3 * C * sizeof(float));
}
extern "C" int ssv_channel_ln_act_bwd(const float* dy, long dy_bs, const float* x, long x_bs, const float* stats, const float* gamma, const float* beta,
                                      float* dx, long dx_bs, float* pgrads, int B, int C, int L, int act, void* ws, size_t ws_bytes, ssv_stream_t stream) {
  SSV_CHECK(dy && x && stats && gamma && beta && dx && B > 0 && C > 0 && L > 0 && act >= 0 && act <= 2, SSV_BAD_SHAPE, "channel_ln_act_bwd: bad argument");   // pgrads may be NULL
  SSV_CHECK(ws && ws_bytes >= ssv_channel_ln_act_bwd_workspace(B, C, L), SSV_BAD_SHAPE, "channel_ln_act_bwd: workspace too small");
  return ssv_launch_ln_act_bwd(dy, dy_bs, x, x_bs, stats, gamma, beta, dx, dx_bs, (float*)ws, pgrads, B, C, L, act, (hipStream_t)stream);
}

// ---- highwayConv ---------------------------------------------------------------------------------------
// Column statistics of h come out of the conv kernel's epilogue (64-row groups) when the split-MFMA kernel runs and the two
// halves are whole groups; the LayerNorm / gate forward is then a streaming kernel without reductions (norm.hip).
static inline bool hw_colstats(int B, int C, int L) { return use_bf3(B, L, C, 2 * C) && C % 64 == 0 && C <= 512; }
static inline size_t hw_colstats_bytes(int B, int C, int L) { return align256((size_t)B * (2 * C / 64) * L * 2 * sizeof(float)); }
extern "C" size_t ssv_highway_conv1d_fwd_workspace(int B, int C, int L, int k) {
  return ssv_conv1d_fwd_workspace(C, 2 * C, k) + (hw_colstats(B, C, L) ? hw_colstats_bytes(B, C, L) : 0);
}
extern "C" int ssv_highway_conv1d_fwd(const float* x, long x_bs, const float* x_amax, int x_namax, const float* w, const void* w_packed, const float* bias,
                                      const float* g1, const float* b1, const float* g2, const float* b2, float* h, float* stats, float* y, long y_bs,
                                      float* y_amax, int B, int C, int L, int k, int dilation, int causal, void* ws, size_t ws_bytes, ssv_stream_t stream) {
  SSV_CHECK(x && w && g1 && b1 && g2 && b2 && h && y, SSV_BAD_SHAPE, "highway_conv1d_fwd: null argument");
  SSV_CHECK(B > 0 && C > 0 && L > 0 && B <= 65535, SSV_BAD_SHAPE, "highway_conv1d_fwd: bad shape B=%d C=%d L=%d", B, C, L);
  if (hw_colstats(B, C, L)) {
    const size_t conv_ws = ssv_conv1d_fwd_workspace(C, 2 * C, k);
    SSV_CHECK(ws && ws_bytes >= conv_ws + hw_colstats_bytes(B, C, L), SSV_BAD_SHAPE, "highway_conv1d_fwd: workspace too small");
    SSV_CHECK(x_bs >= (long)C * L && y_bs >= (long)C * L, SSV_BAD_SHAPE, "highway_conv1d_fwd: batch stride smaller than C*L");
    float* cs = (float*)((char*)ws + conv_ws);
    int shift[3];
    SSV_TRY(conv_shifts(k, dilation, causal, shift));
    SSV_TRY(conv_nn(x, x_bs, w, w_packed, (long)C * k, k, bias, nullptr, nullptr, 0, h, (long)2 * C * L, B, C, 2 * C, L, k, shift, true, ws, (hipStream_t)stream,
                    packed_inv(w_packed, 2 * C, C, k, 0), x_amax, x_namax, 2 * split_bytes(2 * C, C, k), cs));
    return ssv_launch_ln_gate_fwd_stream(h, x, x_bs, cs, g1, b1, g2, b2, y, y_bs, stats, y_amax, B, C, L, (hipStream_t)stream);
  }
  SSV_TRY(ssv_conv1d_fwd(x, x_bs, x_amax, x_namax, w, w_packed, bias, nullptr, h, (long)2 * C * L, nullptr, B, C, 2 * C, L, k, dilation, causal, ws, ws_bytes, stream));
  return ssv_launch_ln_gate_fwd(h, (long)2 * C * L, x, x_bs, g1, b1, g2, b2, y, y_bs, stats, B, C, L, (hipStream_t)stream, y_amax);
}

// ---- 1x1 conv + LayerNorm (+ activation), whole backward ------------------------------------------------------------------
// y = act(LN(conv1x1(x) [+ s])) -- models/TTSModel.py:128-131, :173-180, :218-231, :343-361.  One entry for the backward so that
// the LayerNorm partial rows and the weight-gradient slabs are summed by ONE launch (as in ssv_highway_conv1d_bwd).
// LayerNorm / activation backward + the k = 1 data gradient of a link: ONE launch (round 5, pwln_bwd_kernel) when the transposed weight's planes
// are resident and the shape fits, else ln_act_bwd*, then the data-gradient GEMM.  dpre (B, Cout, L) dense; part: ssv_ln_act_bwd_rows rows.
static int pw_bwd_ln_and_data(const float* dy, long dy_bs, const float* w, const void* w_packed, const float* gamma, const float* beta, const float* pre,
                              const float* stats, float* dx, long dx_bs, float* dpre, float* dpre_amax, float* part, int B, int Cin, int Cout, int L, int act,
                              void* ws, size_t ws_bytes, ssv_stream_t stream) {
  const long pbs = (long)Cout * L;
  if (dx && w_packed && use_bf3(B, L, Cout, Cin) && (!use_f16() || dpre_amax) && ssv_pwln_bwd_fused_ok(B, Cin, Cout, L)) {
    const bool has_amax = dpre_amax != nullptr;
    PwLnBw q;
    q.dy = dy; q.dy_bs = dy_bs; q.pre = pre; q.stats = stats; q.gamma = gamma; q.beta = beta;
    q.dpre = dpre; q.part = part; q.part_rows = ssv_ln_act_bwd_rows(1, Cout, L, has_amax); q.part_q = 4 / ssv_ln_act_bwd_vec(Cout, L, has_amax);
    q.amax = dpre_amax; q.namax = ssv_amax_rows_(L);
    q.Ahi = (const unsigned short*)((const char*)w_packed + 2 * split_bytes(Cout, Cin, 1));
    q.Alo = (const unsigned short*)((const char*)q.Ahi + split_bytes(Cin, Cout, 1));
    q.a_inv = packed_inv(w_packed, Cout, Cin, 1, 1);
    q.dx = dx; q.dx_bs = dx_bs;
    q.xrow_w = (Cin > 128 && Cin % 128 == 1) ? w + (Cin - 1) : nullptr; q.xrow_sk = Cin;       // w[o][Cin - 1], o < Cout
    q.M = Cout; q.Cin = Cin; q.L = L; q.act = act;
    return ssv_launch_pwln_bwd(q, B, use_f16() ? 1 : 0, (hipStream_t)stream);
  }
  SSV_TRY(ssv_launch_ln_act_bwd(dy, dy_bs, pre, pbs, stats, gamma, beta, dpre, pbs, part, nullptr, B, Cout, L, act, (hipStream_t)stream, dpre_amax));
  if (dx) SSV_TRY(ssv_conv1d_bwd_data(dpre, pbs, dpre_amax, ssv_amax_rows_(L), w, w_packed, nullptr, dx, dx_bs, B, Cin, Cout, L, 1, 1, 0, ws, ws_bytes, stream));
  return 0;
}
struct PwWs { size_t dpre, part, amax, wt, slabs, total; };
static PwWs pw_ws(int B, int Cin, int Cout, int L) {
  PwWs s;
  s.dpre = 0;
  s.part = s.dpre + align256((size_t)B * Cout * L * sizeof(float));
  s.amax = s.part + align256((size_t)ssv_ln_gate_bwd_nblk(B, L) * 3 * Cout * sizeof(float));
  s.wt = s.amax + align256((size_t)B * ssv_amax_rows_(L) * sizeof(float));     // max |dpre| per LayerNorm tile (split-fp16 scales)
  s.slabs = s.wt + ssv_conv1d_bwd_data_workspace(Cin, Cout, 1);
  s.total = s.slabs + ssv_conv1d_bwd_weight_workspace(B, Cin, Cout, L, 1);
  return s;
}
extern "C" size_t ssv_pointwise_conv_ln_act_bwd_workspace(int B, int Cin, int Cout, int L) { return pw_ws(B, Cin, Cout, L).total; }
extern "C" int ssv_pointwise_conv_ln_act_bwd(const float* dy, long dy_bs, const float* x, long x_bs, const float* x_amax, int x_namax, const float* w,
                                             const void* w_packed, const float* gamma,
                                             const float* beta, const float* pre, const float* stats, float* dx, long dx_bs, float* dw, float* pgrads,
                                             float* ds, int B, int Cin, int Cout, int L, int act, void* ws, size_t ws_bytes, ssv_stream_t stream) {
  SSV_CHECK(dy && x && w && gamma && beta && pre && stats && dw && pgrads, SSV_BAD_SHAPE, "pointwise_conv_ln_act_bwd: null argument");
  SSV_CHECK(B > 0 && B <= 65535 && Cin > 0 && Cout > 0 && L > 0 && act >= 0 && act <= 2, SSV_BAD_SHAPE, "pointwise_conv_ln_act_bwd: bad shape");
  const PwWs s = pw_ws(B, Cin, Cout, L);
  SSV_CHECK(ws && ws_bytes >= s.total, SSV_BAD_SHAPE, "pointwise_conv_ln_act_bwd: workspace too small (%zu < %zu)", ws_bytes, s.total);
  char* base = (char*)ws;
  float* dpre = (float*)(base + s.dpre);
  const long pbs = (long)Cout * L;
  float* da = use_f16() ? (float*)(base + s.amax) : nullptr;
  const int dn = ssv_amax_rows_(L);
  SSV_TRY(pw_bwd_ln_and_data(dy, dy_bs, w, w_packed, gamma, beta, pre, stats, dx, dx_bs, dpre, da, (float*)(base + s.part), B, Cin, Cout, L, act,
                             base + s.wt, s.slabs - s.wt, stream));
  if (ds) SSV_TRY(ssv_rowsum(dpre, pbs, ds, B, Cout, L, stream));               // gradient of the broadcast (B, Cout, 1) term
  return conv1d_bwd_weight_impl(dpre, pbs, x, x_bs, dw, B, Cin, Cout, L, 1, 1, 0, base + s.slabs, s.total - s.slabs, stream,
                                (const float*)(base + s.part), pgrads, 3 * Cout, ssv_ln_act_bwd_rows(B, Cout, L, da != nullptr), da, dn, x_amax, x_namax);
}

// ---- second order (gradient penalty through the critics) and the gate forward alone ------------------------------------
extern "C" size_t ssv_channel_ln_bwd2_workspace(int B, int C, int L) { return align256((size_t)ssv_ln_gate_bwd_nblk(B, L) * C * sizeof(float)); }
extern "C" int ssv_channel_ln_bwd2(const float* v, long v_bs, const float* gn, long gn_bs, const float* x, long x_bs, const float* stats,
                                   const float* gamma, float* d_gn, long dgn_bs, float* d_x, long dx_bs, float* dgamma,
                                   int B, int C, int L, void* ws, size_t ws_bytes, ssv_stream_t stream) {
  SSV_CHECK(v && gn && x && stats && gamma && d_gn && d_x && dgamma && B > 0 && B <= 65535 && C > 0 && L > 0, SSV_BAD_SHAPE, "channel_ln_bwd2: bad argument");
  SSV_CHECK(ws && ws_bytes >= ssv_channel_ln_bwd2_workspace(B, C, L), SSV_BAD_SHAPE, "channel_ln_bwd2: workspace too small");
  return ssv_launch_ln_bwd2(v, v_bs, gn, gn_bs, x, x_bs, stats, gamma, d_gn, dgn_bs, d_x, dx_bs, (float*)ws, dgamma, B, C, L, (hipStream_t)stream);
}
extern "C" int ssv_highway_gate_fwd(const float* h, const float* x, long x_bs, const float* g1, const float* b1, const float* g2, const float* b2,
                                    float* stats, float* y, long y_bs, float* y_amax, int B, int C, int L, ssv_stream_t stream) {
  SSV_CHECK(h && x && g1 && b1 && g2 && b2 && y && B > 0 && B <= 65535 && C > 0 && L > 0, SSV_BAD_SHAPE, "highway_gate_fwd: bad argument");
  return ssv_launch_ln_gate_fwd(h, (long)2 * C * L, x, x_bs, g1, b1, g2, b2, y, y_bs, stats, B, C, L, (hipStream_t)stream, y_amax);
}
extern "C" size_t ssv_highway_gate_bwd2_workspace(int B, int C, int L) { return align256((size_t)ssv_ln_gate_bwd_nblk(B, L) * 4 * C * sizeof(float)); }
extern "C" int ssv_highway_gate_bwd2(const float* vh, const float* vx, long vx_bs, const float* gy, long gy_bs, const float* h, const float* x, long x_bs,
                                     const float* stats, const float* g1, const float* b1, const float* g2, const float* b2,
                                     float* d_gy, long dgy_bs, float* d_h, float* d_x, long dx_bs, float* pgrads,
                                     int B, int C, int L, void* ws, size_t ws_bytes, ssv_stream_t stream) {
  SSV_CHECK(vh && vx && gy && h && x && stats && g1 && b1 && g2 && b2 && d_gy && d_h && d_x && pgrads && B > 0 && B <= 65535 && C > 0 && L > 0,
            SSV_BAD_SHAPE, "highway_gate_bwd2: bad argument");
  SSV_CHECK(ws && ws_bytes >= ssv_highway_gate_bwd2_workspace(B, C, L), SSV_BAD_SHAPE, "highway_gate_bwd2: workspace too small");
  return ssv_launch_ln_gate_bwd2(vh, vx, vx_bs, gy, gy_bs, h, x, x_bs, stats, g1, b1, g2, b2, d_gy, dgy_bs, d_h, d_x, dx_bs, (float*)ws, pgrads,
                                 B, C, L, (hipStream_t)stream);
}

// ---- highway gate alone (building block: lets a caller overlap the two conv gradients on different streams) ---------------
extern "C" size_t ssv_highway_gate_bwd_workspace(int B, int C, int L) {
  return align256((size_t)ssv_ln_gate_bwd_nblk(B, L) * 6 * C * sizeof(float));
}
extern "C" int ssv_highway_gate_bwd(const float* dy, long dy_bs, const float* x, long x_bs, const float* g1, const float* b1,
                                    const float* g2, const float* b2, const float* h, const float* stats, float* dh, float* dxres,
                                    long dx_bs, float* pgrads, int B, int C, int L, void* ws, size_t ws_bytes, ssv_stream_t stream) {
  SSV_CHECK(dy && x && g1 && b1 && g2 && b2 && h && stats && dh && dxres && B > 0 && C > 0 && L > 0 && B <= 65535, SSV_BAD_SHAPE, "highway_gate_bwd: bad argument");   // pgrads may be NULL
  SSV_CHECK(ws && ws_bytes >= ssv_highway_gate_bwd_workspace(B, C, L), SSV_BAD_SHAPE, "highway_gate_bwd: workspace too small");
  return ssv_launch_ln_gate_bwd(dy, dy_bs, h, x, x_bs, stats, g1, b1, g2, b2, dh, dxres, dx_bs, (float*)ws, pgrads, B, C, L, (hipStream_t)stream);
}

struct HwWs { size_t dh, part, amax, wt, slabs, total; };
static HwWs hw_ws(int B, int C, int L, int k) {
  HwWs s;
  s.dh = 0;
  s.part = s.dh + align256((size_t)B * 2 * C * L * sizeof(float));
  s.amax = s.part + align256((size_t)ssv_ln_gate_bwd_nblk(B, L) * 6 * C * sizeof(float));
  s.wt = s.amax + align256((size_t)B * ssv_amax_rows_(L) * sizeof(float));      // max |dH| per LayerNorm tile (split-fp16 scales)
  s.slabs = s.wt + ssv_conv1d_bwd_data_workspace(C, 2 * C, k);
  s.total = s.slabs + ssv_conv1d_bwd_weight_workspace(B, C, 2 * C, L, k);
  return s;
}
extern "C" size_t ssv_highway_conv1d_bwd_workspace(int B, int C, int L, int k) { return hw_ws(B, C, L, k).total; }
extern "C" int ssv_highway_conv1d_bwd(const float* dy, long dy_bs, const float* x, long x_bs, const float* x_amax, int x_namax, const float* w, const void* w_packed,
                                      const float* g1, const float* b1,
                                      const float* g2, const float* b2, const float* h, const float* stats, float* dx, long dx_bs, float* dw,
                                      float* pgrads, int B, int C, int L, int k, int dilation, int causal, void* ws, size_t ws_bytes,
                                      ssv_stream_t stream) {
  SSV_CHECK(dy && x && w && g1 && b1 && g2 && b2 && h && stats && dx && dw && pgrads, SSV_BAD_SHAPE, "highway_conv1d_bwd: null argument");
  SSV_CHECK(B > 0 && C > 0 && L > 0 && B <= 65535, SSV_BAD_SHAPE, "highway_conv1d_bwd: bad shape B=%d C=%d L=%d", B, C, L);
  const HwWs s = hw_ws(B, C, L, k);
  SSV_CHECK(ws && ws_bytes >= s.total, SSV_BAD_SHAPE, "highway_conv1d_bwd: workspace too small (%zu < %zu)", ws_bytes, s.total);
  char* base = (char*)ws;
  float* dH = (float*)(base + s.dh);
  // gate + both LayerNorms backward: dH (B,2C,L), the residual-path gradient dy*(1-g) into dx, parameter partials
  // (its partial rows are summed at the end, by the launch that also sums the weight-gradient slabs)
  float* da = use_f16() ? (float*)(base + s.amax) : nullptr;
  const int dn = ssv_amax_rows_(L);
  SSV_TRY(ssv_launch_ln_gate_bwd(dy, dy_bs, h, x, x_bs, stats, g1, b1, g2, b2, dH, dx, dx_bs, (float*)(base + s.part), nullptr, B, C, L, (hipStream_t)stream, da));
  // dx += conv^T(dH)
  SSV_TRY(ssv_conv1d_bwd_data(dH, (long)2 * C * L, da, dn, w, w_packed, dx, dx, dx_bs, B, C, 2 * C, L, k, dilation, causal, base + s.wt, s.slabs - s.wt, stream));
  return conv1d_bwd_weight_impl(dH, (long)2 * C * L, x, x_bs, dw, B, C, 2 * C, L, k, dilation, causal, base + s.slabs, s.total - s.slabs, stream,
                                (const float*)(base + s.part), pgrads, 6 * C, ssv_ln_gate_bwd_rows(B, C, L, da != nullptr), da, dn, x_amax, x_namax);
}

extern "C" int ssv_ln_partial_rows(int B, int L) { return ssv_ln_gate_bwd_nblk(B, L); }
extern "C" int ssv_ln_bwd_partial_rows(int gate, int B, int C, int L, int with_amax) {
  return gate ? ssv_ln_gate_bwd_rows(B, C, L, with_amax != 0) : ssv_ln_act_bwd_rows(B, C, L, with_amax != 0);
}
extern "C" size_t ssv_highway_conv1d_bwd_data_workspace(int B, int C, int L, int k) { (void)B; (void)L; return ssv_conv1d_bwd_data_workspace(C, 2 * C, k); }
extern "C" int ssv_highway_conv1d_bwd_data(const float* dy, long dy_bs, const float* x, long x_bs, const float* w, const void* w_packed,
                                           const float* g1, const float* b1, const float* g2, const float* b2, const float* h, const float* stats,
                                           float* dx, long dx_bs, float* dh, float* dh_amax, float* part, int B, int C, int L, int k, int dilation, int causal,
                                           void* ws, size_t ws_bytes, ssv_stream_t stream) {
  SSV_CHECK(dy && x && w && g1 && b1 && g2 && b2 && h && stats && dx && dh && part, SSV_BAD_SHAPE, "highway_conv1d_bwd_data: null argument");
  SSV_CHECK(B > 0 && C > 0 && L > 0 && B <= 65535, SSV_BAD_SHAPE, "highway_conv1d_bwd_data: bad shape B=%d C=%d L=%d", B, C, L);
  SSV_TRY(ssv_launch_ln_gate_bwd(dy, dy_bs, h, x, x_bs, stats, g1, b1, g2, b2, dh, dx, dx_bs, part, nullptr, B, C, L, (hipStream_t)stream, dh_amax));
  return ssv_conv1d_bwd_data(dh, (long)2 * C * L, dh_amax, ssv_amax_rows_(L), w, w_packed, dx, dx, dx_bs, B, C, 2 * C, L, k, dilation, causal, ws, ws_bytes, stream);
}
extern "C" size_t ssv_pointwise_conv_ln_act_bwd_data_workspace(int B, int Cin, int Cout, int L) { (void)B; (void)L; return ssv_conv1d_bwd_data_workspace(Cin, Cout, 1); }
extern "C" int ssv_pointwise_conv_ln_act_bwd_data(const float* dy, long dy_bs, const float* w, const void* w_packed, const float* gamma, const float* beta,
                                                  const float* pre, const float* stats, float* dx, long dx_bs, float* ds, float* dpre, float* dpre_amax,
                                                  float* part, int B, int Cin, int Cout, int L, int act, void* ws, size_t ws_bytes, ssv_stream_t stream) {
  SSV_CHECK(dy && w && gamma && beta && pre && stats && dpre && part, SSV_BAD_SHAPE, "pointwise_conv_ln_act_bwd_data: null argument");
  SSV_CHECK(B > 0 && B <= 65535 && Cin > 0 && Cout > 0 && L > 0 && act >= 0 && act <= 2, SSV_BAD_SHAPE, "pointwise_conv_ln_act_bwd_data: bad shape");
  const long pbs = (long)Cout * L;
  SSV_TRY(pw_bwd_ln_and_data(dy, dy_bs, w, w_packed, gamma, beta, pre, stats, dx, dx_bs, dpre, dpre_amax, part, B, Cin, Cout, L, act, ws, ws_bytes, stream));
  if (ds) SSV_TRY(ssv_rowsum(dpre, pbs, ds, B, Cout, L, stream));
  return 0;
}

// ---- attention -------------------------------------------------------------------------------------------
extern "C" int ssv_attention_apply(const float* v, long kv_bs, const float* a, int a_T, float* r, long r_bs, int B, int d, int N, int T, ssv_stream_t stream) {
  SSV_CHECK(v && a && r && B > 0 && d > 0 && N > 0 && T > 0 && a_T >= T, SSV_BAD_SHAPE, "attention_apply: bad argument");
  GemmNN g = nn_zero();
  g.A = v; g.sab = kv_bs; g.sam = N; g.sac = 1; g.saj = 0;
  g.X = a; g.sxb = (long)N * a_T; g.sxc = a_T; g.Lx = T;
  g.C = r; g.scb = r_bs; g.scm = T;
  g.M = d; g.N = T; g.Kc = N; g.B = B;
  return ssv_launch_gemm_nn(g, (hipStream_t)stream);
}
bool ssv_attn_fused_ok(int B, int d, int N, int T);        // attn_fused.hip: scores, softmax and V A (backward: dA, dS, dQ) in one launch
int ssv_launch_attn_fwd_fused(const float* k, const float* v, long kv_bs, const float* q, long q_bs, float* a, float* rq, long rq_bs, int copy_q,
                              int B, int d, int N, int T, hipStream_t st);
int ssv_launch_attn_bwd_fused(const float* dr, long dr_bs, const float* da_ext, const float* dq_add, long dq_add_bs, const float* k, const float* v, long kv_bs,
                              const float* a, float* ds, float* dq, long dq_bs, int B, int d, int N, int T, hipStream_t st);
#ifndef SSV_ATTN_FUSED
#define SSV_ATTN_FUSED 1     // (tuning builds: 0 = two GEMM launches, the softmax kernel and the row copy, as before round 5)
#endif
extern "C" int ssv_attention_train_fwd(const float* k, const float* v, long kv_bs, const float* q, long q_bs, float* a, float* r, long r_bs,
                                       int B, int d, int N, int T, ssv_stream_t stream) {
  SSV_CHECK(k && v && q && a && r && B > 0 && d > 0 && N > 0 && T > 0, SSV_BAD_SHAPE, "attention_train_fwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  if (SSV_ATTN_FUSED && ssv_attn_fused_ok(B, d, N, T)) return ssv_launch_attn_fwd_fused(k, v, kv_bs, q, q_bs, a, r, r_bs, 0, B, d, N, T, st);
  GemmNN g = nn_zero();                       // scores(b,n,t) = sum_c k(b,c,n) q(b,c,t) / sqrt(d)
  g.A = k; g.sab = kv_bs; g.sam = 1; g.sac = N; g.saj = 0;
  g.X = q; g.sxb = q_bs; g.sxc = T; g.Lx = T;
  g.C = a; g.scb = (long)N * T; g.scm = T;
  g.M = N; g.N = T; g.Kc = d; g.B = B; g.alpha = 1.f / sqrtf((float)d);
  SSV_TRY(ssv_launch_gemm_nn(g, st));
  SSV_TRY(ssv_launch_softmax_cols(a, B, N, T, st));
  return ssv_attention_apply(v, kv_bs, a, T, r, r_bs, B, d, N, T, stream);
}
// The decoder's input cat(R, Q) (models/TTSModel.py:270) in the same call: rq (B, 2d, T) receives R in rows [0, d) and a copy of Q in rows [d, 2d).
extern "C" int ssv_attention_train_fwd_rq(const float* k, const float* v, long kv_bs, const float* q, long q_bs, float* a, float* rq, long rq_bs,
                                          int B, int d, int N, int T, ssv_stream_t stream) {
  SSV_CHECK(k && v && q && a && rq && B > 0 && d > 0 && N > 0 && T > 0 && rq_bs >= (long)2 * d * T, SSV_BAD_SHAPE, "attention_train_fwd_rq: bad argument");
  if (SSV_ATTN_FUSED && ssv_attn_fused_ok(B, d, N, T)) return ssv_launch_attn_fwd_fused(k, v, kv_bs, q, q_bs, a, rq, rq_bs, 1, B, d, N, T, (hipStream_t)stream);
  SSV_TRY(ssv_attention_train_fwd(k, v, kv_bs, q, q_bs, a, rq, rq_bs, B, d, N, T, stream));
  return ssv_copy_rows(q, q_bs, rq + (long)d * T, rq_bs, B, (long)d * T, stream);
}
extern "C" size_t ssv_attention_train_bwd_workspace(int B, int d, int N, int T) { (void)d; return align256((size_t)B * N * T * sizeof(float)) + 2 * AMAX_FB_BYTES; }
// Per-batch-item products reduced over time (attention dV, dK): the split-bf16 weight-gradient kernel with one slab per batch
// item and no slab sum (21 -> ~100 TFLOP/s at d = 256, N = 186, T = 325; the fp32 kernel's 128 x 96 tiles leave the chip idle).
// fb: 2 * SSV_AMAX_FB_FLOATS floats of workspace for the operands' scale lists (split-fp16)
static int nt_per_batch(GemmNT& g, int T, hipStream_t st, float* fb) {
  g.KT = 1;
  g.scj = 1;
  if (ssv_precision() >= 1 && (long)g.B * T >= 256 && ssv_nt_bf3_fits(g)) {
    if (use_f16()) {
      AmaxList la, lx;
      SSV_TRY(amax_of(g.A, g.sab, g.B, (long)g.M * T, nullptr, 0, fb, &la, st));
      SSV_TRY(amax_of(g.X, g.sxb, g.B, (long)g.Nc * T, nullptr, 0, fb + SSV_AMAX_FB_FLOATS, &lx, st));
      g.f16 = 1; g.a_amax = la.p; g.a_namax = la.n * g.B; g.x_amax = lx.p; g.x_namax = lx.n * g.B;
    }
    return ssv_launch_gemm_nt_bf3(g, st);
  }
  return ssv_launch_gemm_nt(g, st);
}

extern "C" int ssv_attention_train_bwd(const float* dr, long dr_bs, const float* da_ext, const float* dq_add, long dq_add_bs,
                                       const float* k, const float* v, long kv_bs, const float* q, long q_bs, const float* a,
                                       float* dk, float* dv, long dkv_bs, float* dq, long dq_bs, int B, int d, int N, int T,
                                       void* ws, size_t ws_bytes, ssv_stream_t stream) {
  SSV_CHECK(dr && k && v && q && a && dk && dv && dq && B > 0 && d > 0 && N > 0 && T > 0, SSV_BAD_SHAPE, "attention_train_bwd: bad argument");
  SSV_CHECK(ws && ws_bytes >= ssv_attention_train_bwd_workspace(B, d, N, T), SSV_BAD_SHAPE, "attention_train_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* dA = (float*)ws;
  float* fb = (float*)((char*)ws + align256((size_t)B * N * T * sizeof(float)));
  if (SSV_ATTN_FUSED && ssv_attn_fused_ok(B, d, N, T)) {
    // dA, dS (left in ws for dk) and dq in ONE launch; the two reductions over time stay on the weight-gradient kernel
    SSV_TRY(ssv_launch_attn_bwd_fused(dr, dr_bs, da_ext, dq_add, dq_add_bs, k, v, kv_bs, a, dA, dq, dq_bs, B, d, N, T, st));
    {  // dv(b,c,n) = sum_t dr(b,c,t) a(b,n,t)
      GemmNT g = nt_zero();
      g.A = dr; g.sab = dr_bs; g.sam = T; g.La = T;
      g.X = a; g.sxb = (long)N * T; g.sxc = T; g.Lx = T;
      g.C = dv; g.scz = dkv_bs; g.scm = N; g.scc = 1;
      g.M = d; g.Nc = N; g.B = B; g.Z = B; g.bstep = B;
      SSV_TRY(nt_per_batch(g, T, st, fb));
    }
    {  // dk(b,c,n) = sum_t q(b,c,t) ds(b,n,t)
      GemmNT g = nt_zero();
      g.A = q; g.sab = q_bs; g.sam = T; g.La = T;
      g.X = dA; g.sxb = (long)N * T; g.sxc = T; g.Lx = T;
      g.C = dk; g.scz = dkv_bs; g.scm = N; g.scc = 1;
      g.M = d; g.Nc = N; g.B = B; g.Z = B; g.bstep = B;
      SSV_TRY(nt_per_batch(g, T, st, fb));
    }
    return 0;
  }
  {  // dA(b,n,t) = sum_c v(b,c,n) dr(b,c,t)
    GemmNN g = nn_zero();
    g.A = v; g.sab = kv_bs; g.sam = 1; g.sac = N;
    g.X = dr; g.sxb = dr_bs; g.sxc = T; g.Lx = T;
    g.C = dA; g.scb = (long)N * T; g.scm = T;
    g.M = N; g.N = T; g.Kc = d; g.B = B;
    SSV_TRY(ssv_launch_gemm_nn(g, st));
  }
  {  // dv(b,c,n) = sum_t dr(b,c,t) a(b,n,t)
    GemmNT g = nt_zero();
    g.A = dr; g.sab = dr_bs; g.sam = T; g.La = T;
    g.X = a; g.sxb = (long)N * T; g.sxc = T; g.Lx = T;
    g.C = dv; g.scz = dkv_bs; g.scm = N; g.scc = 1;
    g.M = d; g.Nc = N; g.B = B; g.Z = B; g.bstep = B;
    SSV_TRY(nt_per_batch(g, T, st, fb));
  }
  SSV_TRY(ssv_launch_softmax_cols_bwd(a, dA, da_ext, 1.f / sqrtf((float)d), B, N, T, st));   // dA now holds dScores
  {  // dk(b,c,n) = sum_t q(b,c,t) ds(b,n,t)
    GemmNT g = nt_zero();
    g.A = q; g.sab = q_bs; g.sam = T; g.La = T;
    g.X = dA; g.sxb = (long)N * T; g.sxc = T; g.Lx = T;
    g.C = dk; g.scz = dkv_bs; g.scm = N; g.scc = 1;
    g.M = d; g.Nc = N; g.B = B; g.Z = B; g.bstep = B;
    SSV_TRY(nt_per_batch(g, T, st, fb));
  }
  {  // dq(b,c,t) = sum_n k(b,c,n) ds(b,n,t) + dq_add
    GemmNN g = nn_zero();
    g.A = k; g.sab = kv_bs; g.sam = N; g.sac = 1;
    g.X = dA; g.sxb = (long)N * T; g.sxc = T; g.Lx = T;
    g.C = dq; g.scb = dq_bs; g.scm = T;
    if (dq_add) { g.R = dq_add; g.srb = dq_add_bs; g.srm = T; }
    g.M = d; g.N = T; g.Kc = N; g.B = B;
    SSV_TRY(ssv_launch_gemm_nn(g, st));
  }
  return 0;
}

// ---- ConvTranspose1d(k=2, s=2) -----------------------------------------------------------------------------
// Split-bf16 path of the two deconvolution halves: both taps' weights are split by ONE pack launch (tap-major planes);
// tap j is a k=1 product whose output (forward) or input (data gradient) columns have stride 2.
static size_t deconv_pack_bytes(int rows, int K) { return 2 * split_bytes(rows, K, 2); }
extern "C" size_t ssv_deconv1d_k2s2_fwd_workspace(int Cin, int Cout) { return deconv_pack_bytes(Cout, Cin) + conv_aux_bytes(); }
extern "C" int ssv_deconv1d_k2s2_fwd(const float* x, long x_bs, const float* x_amax, int x_namax, const float* w, const void* w_packed, const float* bias,
                                     float* y, long y_bs, float* y_amax, int y_namax, int B, int Cin, int Cout, int L, void* ws, size_t ws_bytes, ssv_stream_t stream) {
  SSV_CHECK(x && w && y && B > 0 && Cin > 0 && Cout > 0 && L > 0, SSV_BAD_SHAPE, "deconv1d_k2s2_fwd: bad argument");
  SSV_CHECK(!y_amax || y_namax > 0, SSV_BAD_SHAPE, "deconv1d_k2s2_fwd: scale list of %d entries", y_namax);
  hipStream_t st = (hipStream_t)stream;
  if (use_bf3(B, L, Cin, Cout)) {
    // ONE product over 2 Cout rows (round 6): u = W2^T x with W2 = w.view(Cin, 2 Cout) -- row 2 o + j of u is tap j of output channel o -- whose
    // epilogue interleaves row pairs into y(b, o, 2 t + j) (GemmNNB::row_pair).  The planes are the TRANSPOSED planes of the 1x1 weight
    // w.view(Cin, 2 Cout, 1): resident ones when the caller keeps them (w_packed, ssv_conv_pack_multi), else split here.
    // (Before: one stride-2 product per tap behind a per-call scan + split of the weight, and an ssv_absmax launch over y for the next layer.)
    SSV_CHECK(ws && ws_bytes >= ssv_deconv1d_k2s2_fwd_workspace(Cin, Cout), SSV_BAD_SHAPE, "deconv1d_k2s2_fwd: workspace too small");
    const int M2 = 2 * Cout, Kpad = pad32(Cin);
    const bool f16 = use_f16();
    float* aux = (float*)((char*)ws + deconv_pack_bytes(Cout, Cin));
    const unsigned short* hi; const unsigned short* lo; const float* a_inv = nullptr;
    if (w_packed) {
      hi = (const unsigned short*)((const char*)w_packed + 2 * split_bytes(Cin, M2, 1));       // behind the forward planes of the (Cin, 2 Cout, 1) weight
      lo = (const unsigned short*)((const char*)hi + split_bytes(M2, Cin, 1));
      if (f16) a_inv = packed_inv(w_packed, Cin, M2, 1, 1);
    } else {
      unsigned short* phi = (unsigned short*)ws;
      unsigned short* plo = (unsigned short*)((char*)ws + split_bytes(M2, Cin, 1));
      // (m = 2 o + j, k = c) = w[c][o][j] = w[c * 2 Cout + m]: row stride 1, column stride 2 Cout
      if (f16) { SSV_TRY(ssv_launch_pack_split_f16(w, (long)Cin * M2, phi, plo, M2, Cin, Kpad, 1, 1, (long)M2, 1, aux, st)); a_inv = aux + 64; }
      else SSV_TRY(ssv_launch_pack_split(w, phi, plo, M2, Cin, Kpad, 1, 1, (long)M2, 1, 0, st));
      hi = phi; lo = plo;
    }
    GemmNNB g = nnb_zero();
    if (f16) {
      AmaxList xa = {nullptr, 0};
      SSV_TRY(amax_of(x, x_bs, B, (long)Cin * L, x_amax, x_namax, aux + SSV_F16_AUX_FLOATS, &xa, st));
      g.f16 = 1; g.a_inv = a_inv; g.x_amax = xa.p; g.x_namax = xa.n; g.x_amax_bs = xa.n;
    }
    g.Ahi = hi; g.Alo = lo; g.Kpad = Kpad;
    g.X = x; g.sxb = x_bs; g.sxc = L; g.Lx = L;
    g.C = y; g.scb = y_bs; g.scm = (long)2 * L; g.row_pair = 1;
    g.bias = bias;
    g.M = M2; g.N = L; g.Kc = Cin; g.B = B;
    if (y_amax && f16) { g.c_amax = y_amax; g.c_namax = y_namax; }
    const int rc = ssv_launch_gemm_nn_bf3(g, st);
    if (rc == SSV_UNSUPPORTED && g.c_amax) {                   // more tiles per item than list entries: the product without the list, then a scan
      g.c_amax = nullptr; g.c_namax = 0;
      SSV_TRY(ssv_launch_gemm_nn_bf3(g, st));
      return ssv_launch_absmax(y, y_bs, B, (long)Cout * 2 * L, y_amax, y_namax, st);
    }
    return rc;
  }
  for (int j = 0; j < 2; ++j) {               // y(b,o,2t+j) = bias[o] + sum_c w[c,o,j] x(b,c,t)
    GemmNN g = nn_zero();
    g.A = w + j; g.sam = 2; g.sac = (long)2 * Cout;
    g.X = x; g.sxb = x_bs; g.sxc = L; g.Lx = L;
    g.C = y + j; g.scb = y_bs; g.scm = (long)2 * L; g.scn = 2;
    g.bias = bias;
    g.M = Cout; g.N = L; g.Kc = Cin; g.B = B;
    SSV_TRY(ssv_launch_gemm_nn(g, st));
  }
  if (y_amax && use_f16()) return ssv_launch_absmax(y, y_bs, B, (long)Cout * 2 * L, y_amax, y_namax, st);
  return 0;
}
static int deconv_splits(int B, int Cin, int Cout) { return dw_splits(B, Cin, Cout, 1, SSV_NOMINAL_L); }     // (ssv_deconv1d_k2s2_bwd_workspace has no length)
extern "C" size_t ssv_deconv1d_k2s2_bwd_workspace(int B, int Cin, int Cout) {
  return align256((size_t)deconv_splits(B, Cin, Cout) * Cin * Cout * 2 * sizeof(float)) + align256((size_t)B * Cout * sizeof(float)) +
         deconv_pack_bytes(Cin, Cout) + conv_aux_bytes();
}
extern "C" int ssv_deconv1d_k2s2_bwd(const float* dy, long dy_bs, const float* dy_amax, int dy_namax, const float* x, long x_bs, const float* w, float* dx, long dx_bs,
                                     float* dw, float* dbias, int B, int Cin, int Cout, int L, void* ws, size_t ws_bytes, ssv_stream_t stream) {
  SSV_CHECK(dy && x && w && dx && B > 0 && Cin > 0 && Cout > 0 && L > 0, SSV_BAD_SHAPE, "deconv1d_k2s2_bwd: bad argument");      // dw may be NULL: see the header
  SSV_CHECK(ws && ws_bytes >= ssv_deconv1d_k2s2_bwd_workspace(B, Cin, Cout), SSV_BAD_SHAPE, "deconv1d_k2s2_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int Z = deconv_splits(B, Cin, Cout);
  const long n = (long)Cin * Cout * 2;
  float* slabs = (float*)ws;
  float* rs = (float*)((char*)ws + align256((size_t)Z * n * sizeof(float)));
  const bool bf3 = use_bf3(B, L, Cout, Cin);
  unsigned short* hi = (unsigned short*)((char*)rs + align256((size_t)B * Cout * sizeof(float)));
  unsigned short* lo = (unsigned short*)((char*)hi + split_bytes(Cin, Cout, 2));
  const int Kpad = pad32(Cout);
  const size_t tap = (size_t)((Cin + 15) / 16 * 16) * Kpad;
  const bool f16 = bf3 && use_f16();
  float* aux = (float*)((char*)hi + deconv_pack_bytes(Cin, Cout));
  AmaxList ya = {nullptr, 0};
  if (f16) {
    SSV_TRY(ssv_launch_pack_split_f16(w, (long)Cin * Cout * 2, hi, lo, Cin, Cout, Kpad, 2, (long)2 * Cout, 2, 1, aux, st));
    SSV_TRY(amax_of(dy, dy_bs, B, (long)Cout * 2 * L, dy_amax, dy_namax, aux + SSV_F16_AUX_FLOATS, &ya, st));
  } else if (bf3) SSV_TRY(ssv_launch_pack_split(w, hi, lo, Cin, Cout, Kpad, 2, (long)2 * Cout, 2, 1, 0, st));   // (m=c, k=o, tap j) = w[c][o][j]
  for (int j = 0; j < 2; ++j) {
    if (bf3) {                                 // dx(b,c,t) (+)= sum_o w[c,o,j] dy(b,o,2t+j)
      GemmNNB g = nnb_zero();
      if (f16) { g.f16 = 1; g.a_inv = aux + 64; g.x_amax = ya.p; g.x_namax = ya.n; g.x_amax_bs = ya.n; }
      g.Ahi = hi + j * tap; g.Alo = lo + j * tap; g.Kpad = Kpad;
      g.X = dy + j; g.sxb = dy_bs; g.sxc = (long)2 * L; g.sxn = 2; g.Lx = L;
      g.C = dx; g.scb = dx_bs; g.scm = L;
      if (j == 1) { g.R = dx; g.srb = dx_bs; g.srm = L; }
      g.M = Cin; g.N = L; g.Kc = Cout; g.B = B;
      SSV_TRY(ssv_launch_gemm_nn_bf3(g, st));
    } else {
      GemmNN g = nn_zero();
      g.A = w + j; g.sam = (long)2 * Cout; g.sac = 2;
      g.X = dy + j; g.sxb = dy_bs; g.sxc = (long)2 * L; g.sxn = 2; g.Lx = L;
      g.C = dx; g.scb = dx_bs; g.scm = L;
      if (j == 1) { g.R = dx; g.srb = dx_bs; g.srm = L; }
      g.M = Cin; g.N = L; g.Kc = Cout; g.B = B;
      SSV_TRY(ssv_launch_gemm_nn(g, st));
    }
    if (!dw) continue;
    GemmNT t = nt_zero();                      // dw[c,o,j] = sum_{b,t} x(b,c,t) dy(b,o,2t+j)
    t.A = x; t.sab = x_bs; t.sam = L; t.La = L;
    t.X = dy + j; t.sxb = dy_bs; t.sxc = (long)2 * L; t.sxn = 2; t.Lx = L;
    t.C = ((Z == 1) ? dw : slabs) + j; t.scz = n; t.scm = (long)2 * Cout; t.scc = 2;
    t.M = Cin; t.Nc = Cout; t.B = B; t.Z = Z; t.bstep = Z;
    SSV_TRY(ssv_launch_gemm_nt(t, st));
  }
  if (dw && Z > 1) SSV_TRY(ssv_launch_reduce_slabs(slabs, dw, n, Z, n, st));
  if (dbias) {
    SSV_TRY(ssv_rowsum(dy, dy_bs, rs, B, Cout, 2 * L, stream));
    SSV_TRY(ssv_launch_reduce_slabs(rs, dbias, Cout, B, Cout, st));
  }
  return 0;
}

// ---- GE2E speaker embedder ---------------------------------------------------------------------------------
struct LstmWs { size_t xt, xp, seq0, seq1, g, c, wih, whh, total; };
static LstmWs lstm_ws(int Bn, int T, int F, int H) {
  LstmWs s;
  s.xt = 0;
  s.xp = s.xt + align256((size_t)T * F * Bn * sizeof(float));
  s.seq0 = s.xp + align256((size_t)T * 4 * H * Bn * sizeof(float));
  s.seq1 = s.seq0 + align256((size_t)T * H * Bn * sizeof(float));
  s.g = s.seq1 + align256((size_t)T * H * Bn * sizeof(float));
  s.c = s.g + align256((size_t)4 * H * Bn * sizeof(float));
  s.wih = s.c + align256((size_t)H * Bn * sizeof(float));
  s.whh = s.wih + 2 * split_bytes(4 * H, F > H ? F : H, 1);        // pre-split weights (hi, lo planes)
  s.total = s.whh + 2 * split_bytes(4 * H, H, 1);
  return s;
}
// C = A X (+ bias + bias_b + R) with A (M x K) row-major weights and X, C as [rows][Bn] activations; "batch" of nb
// independent problems strided by sxb / scb.  fp32 MFMA path (the split-bf16 path is spelled out in ssv_lstm_fwd).
static int lstm_gemm_f32(const float* A, const float* X, long sxb, float* C, long scb, const float* bias, const float* bias_b,
                         const float* R, int M, int K, int Bn, int nb, hipStream_t st) {
  GemmNN g = nn_zero();
  g.A = A; g.sam = K; g.sac = 1; g.saj = 1;
  g.X = X; g.sxb = sxb; g.sxc = Bn; g.Lx = Bn;
  g.C = C; g.scb = scb; g.scm = Bn;
  g.bias = bias; g.bias_b = bias_b; g.sbb = 0;
  if (R) { g.R = R; g.srm = Bn; }
  g.M = M; g.N = Bn; g.Kc = K; g.B = nb;
  return ssv_launch_gemm_nn(g, st);
}
// Wavefront (split-bf16) layout: h of every layer lives in a 2-frame ring, weights of layer l >= 1 are [W_ih | W_hh] side by side.
struct LstmWave { size_t xt, xp, out, c, bias, ih0, hh0, comb, comb_stride, aux, hp, hp_plane, l0c, x0p, total; int npad, xsplit0; };
// split-fp16 scales of the wavefront (floats at `aux`): [0, 64) partial maxima over ALL weight matrices (one scale for every layer: a
// launch batches layers over grid.y and has one epilogue factor), [64] its inverse scale, [128, 192) partial maxima of the input frames
// (layer 0's projection).  The recurrent activations need no list: |h| = |o tanh c| < 1, their scale is the constant 2^14 (x_namax = 0).
#define LSTM_AUX_FLOATS 192
static LstmWave lstm_wave_ws(int Bn, int T, int F, int H, int layers) {
  LstmWave s;
  s.xt = 0;
  s.xp = s.xt + align256((size_t)T * F * Bn * sizeof(float));
  s.out = s.xp + align256((size_t)T * 4 * H * Bn * sizeof(float));
  s.c = s.out + align256((size_t)layers * 2 * H * Bn * sizeof(float));
  s.bias = s.c + align256((size_t)layers * H * Bn * sizeof(float));
  s.ih0 = s.bias + align256((size_t)layers * 8 * H * sizeof(float));
  s.hh0 = s.ih0 + 2 * split_bytes(4 * H, F, 1);
  s.comb = s.hh0 + 2 * split_bytes(4 * H, H, 1);
  s.comb_stride = 2 * split_bytes(4 * H, 2 * H, 1);
  s.aux = s.comb + (size_t)(layers > 1 ? layers - 1 : 0) * s.comb_stride;
  // pre-split recurrent activations (GemmNNB::hs_planes): hi and lo planes of [H / 8][npad][8 halves] per (layer, ring slot); npad = whole 128-column tiles
  s.npad = (Bn + 127) / 128 * 128;
  s.hp = s.aux + align256(LSTM_AUX_FLOATS * sizeof(float));
  s.hp_plane = (H % 8 == 0) ? (size_t)(H / 8) * s.npad * 16 : 0;
  // layer 0's input as the first K segment of its product (GemmNNB::x0_planes): the planes of [W_ih (F padded to whole chunk pairs) | W_hh] and the
  // input frames pre-split, a (hi, lo) plane pair of the recurrent activations' size per frame (only its first 4 * xsplit0 k-groups are used)
  s.xsplit0 = 2 * ((F + 63) / 64);
  s.l0c = s.hp + align256((size_t)layers * 2 * 2 * s.hp_plane);
  s.x0p = s.l0c + 2 * split_bytes(4 * H, 32 * s.xsplit0 + H, 1);
  s.total = s.x0p + align256((size_t)T * 2 * s.hp_plane);
  return s;
}
static bool lstm_wave_ok(int Bn, int H) { return Bn >= 64 && H >= 32 && H % 32 == 0; }
extern "C" size_t ssv_lstm_fwd_workspace(int Bn, int T, int F, int H, int layers) {
  return zmax(lstm_ws(Bn, T, F, H).total, lstm_wave_ws(Bn, T, F, H, layers).total);
}
// LSTM forward as a wavefront over (layer, frame): in step s layer l computes frame s - l, so the layers' recurrent products
// (each too small to fill the chip: 672 workgroups of 24 K-chunks) run side by side in ONE launch, and a layer's input
// projection rides along as the first K segment of the same product instead of a separate pass over all frames.
// T + layers - 1 steps of two launches (layer 0, whose input projection W_ih x_t is precomputed for all frames, and layers
// 1.. batched over grid.y) instead of layers * T sequential products.
// Training (keep != null): every frame of h, c and the activated gates is kept in the caller's buffers (D = T instead of the 2-frame ring).
static int lstm_fwd_wave(const float* x, const float* const* w_ih, const float* const* w_hh, const float* const* b_ih,
                         const float* const* b_hh, float* h_last, int Bn, int T, int F, int H, int layers, char* base, hipStream_t st,
                         float* keep_xt = nullptr, float* keep_hs = nullptr, float* keep_cs = nullptr, float* keep_gates = nullptr, bool packed = false) {
  const LstmWave s = lstm_wave_ws(Bn, T, F, H, layers);
  const int D = keep_hs ? T : 2;
  float* xt = keep_xt ? keep_xt : (float*)(base + s.xt);
  float* xp = (float*)(base + s.xp);
  float* out = keep_hs ? keep_hs : (float*)(base + s.out);
  float* cbuf = keep_cs ? keep_cs : (float*)(base + s.c);
  float* bias = (float*)(base + s.bias);
  const long HN = (long)H * Bn;
  // Arithmetic of the products: split-fp16 in the default mode (the reference's nn.LSTM computes in fp32,
  // GE2E/speech_embedder_net.py:19,28), split-bf16 when that mode is selected.
  const bool f16 = use_f16() && 2 * layers <= 64;
  // One launch per wavefront step (SSV_LSTM_MERGE=0 keeps the two launches: tuning), and in the split-fp16 mode the cells write h already split into the
  // consumers' staging order (GemmNNB::hs_planes): no split, no masks and a quarter of the load instructions in the products' input staging.  The input
  // frames are then pre-split the same way and layer 0's W_ih x_t is the first K segment of its product (GemmNNB::x0_planes): no projection of all frames
  // (0.42 ms and 1.3 GB written, then read back by the cells, at config 5's shape).
  const char* mk = ssv_tuning(SSV_T_LSTM_MERGE);
  const bool merge = !(mk && atoi(mk) == 0);
  const bool presplit = SSV_LSTM_PRESPLIT && f16 && merge && layers >= 2 && (SSV_LSTM_PRESPLIT_TRAIN || (!keep_hs && D == 2)) && H % 32 == 0 && s.hp_plane > 0 && s.hp_plane < ((size_t)1 << 31) &&
                        !(mk && atoi(mk) == 2);
  const bool x0fold = presplit && SSV_LSTM_X0FOLD && 4 * s.xsplit0 <= H / 8;
  if (!x0fold || keep_xt) SSV_TRY(ssv_launch_lstm_in_transpose(x, xt, Bn, T, F, st));    // [T][F][Bn]  (training keeps it for W_ih[0]'s gradient)
  // packed: the workspace still holds what a previous call of the same shape and arithmetic mode prepared from the SAME weight values -- bias
  // rows, split weight planes, the weights' scale (ssv_lstm_fwd_cached: d-vector extraction runs batch after batch on fixed weights; the six
  // absmax scans over 48 MB of weights and the six packs were ~0.35 ms of an 11.6 ms forward)
  for (int l = 0; !packed && l < layers; ++l) {                   // biases side by side: [layer][b_ih (4H) | b_hh (4H)]
    SSV_HIP(hipMemcpyAsync(bias + (long)l * 8 * H, b_ih[l], (size_t)4 * H * sizeof(float), hipMemcpyDeviceToDevice, st));
    SSV_HIP(hipMemcpyAsync(bias + (long)l * 8 * H + 4 * H, b_hh[l], (size_t)4 * H * sizeof(float), hipMemcpyDeviceToDevice, st));
  }
  // weights, rows gate-interleaved (row 4u + gate) so that the product can finish the cell in its epilogue
  unsigned short* ih0_hi = (unsigned short*)(base + s.ih0);
  unsigned short* ih0_lo = (unsigned short*)(base + s.ih0 + split_bytes(4 * H, F, 1));
  unsigned short* hh0_hi = (unsigned short*)(base + s.hh0);
  unsigned short* hh0_lo = (unsigned short*)(base + s.hh0 + split_bytes(4 * H, H, 1));
  unsigned short* l0c_hi = (unsigned short*)(base + s.l0c);
  unsigned short* l0c_lo = (unsigned short*)(base + s.l0c + split_bytes(4 * H, 32 * s.xsplit0 + H, 1));
  float* aux = (float*)(base + s.aux);
  if (f16) {
    const int npb = 64 / (2 * layers);                            // partial maxima per weight matrix
    if (!packed) {
      SSV_HIP(hipMemsetAsync(aux, 0, LSTM_AUX_FLOATS * sizeof(float), st));
      for (int l = 0; l < layers; ++l) {
        SSV_TRY(ssv_launch_absmax(w_ih[l], 0, 1, (long)4 * H * (l == 0 ? F : H), aux + (2 * l) * npb, npb, st));
        SSV_TRY(ssv_launch_absmax(w_hh[l], 0, 1, (long)4 * H * H, aux + (2 * l + 1) * npb, npb, st));
      }
    }
    SSV_TRY(ssv_launch_absmax(x0fold ? x : xt, 0, 1, (long)T * F * Bn, aux + 128, 64, st));       // (writes all 64 entries of the input's list; the same values either way)
  }
  auto pack = [&](const float* w, unsigned short* hi, unsigned short* lo, int K, int Kpad, int nch_total, int ch_off) -> int {
    if (f16) return ssv_launch_pack_split_f16_list(w, hi, lo, 4 * H, K, Kpad, 1, K, 1, 1, H, aux, 64, aux + 64, st, nch_total, ch_off);
    return ssv_launch_pack_split(w, hi, lo, 4 * H, K, Kpad, 1, K, 1, 1, H, st, nch_total, ch_off);
  };
  const int hch = H / 32;
  if (!packed && !x0fold) {
    SSV_TRY(pack(w_ih[0], ih0_hi, ih0_lo, F, pad32(F), 0, 0));
    SSV_TRY(pack(w_hh[0], hh0_hi, hh0_lo, H, H, 0, 0));
  }
  if (!packed && x0fold) {                                       // [W_ih (zero-padded to xsplit0 chunks) | W_hh], one row of chunks per 16 output rows
    SSV_TRY(pack(w_ih[0], l0c_hi, l0c_lo, F, 32 * s.xsplit0, s.xsplit0 + hch, 0));
    SSV_TRY(pack(w_hh[0], l0c_hi, l0c_lo, H, H, s.xsplit0 + hch, s.xsplit0));
  }
  for (int l = 1; !packed && l < layers; ++l) {
    unsigned short* hi = (unsigned short*)(base + s.comb + (size_t)(l - 1) * s.comb_stride);
    unsigned short* lo = (unsigned short*)((char*)hi + split_bytes(4 * H, 2 * H, 1));
    SSV_TRY(pack(w_ih[l], hi, lo, H, H, 2 * hch, 0));
    SSV_TRY(pack(w_hh[l], hi, lo, H, H, 2 * hch, hch));
  }
  // layer 0's input projection for every frame at once (biases are left to the cell): xp[t] = W_ih x_t
  if (x0fold) SSV_TRY(ssv_launch_lstm_x_planes(x, aux + 128, base + s.x0p, (long)s.hp_plane, Bn, T, F, 4 * s.xsplit0, s.npad, st));
  else {
    GemmNNB g = nnb_zero();
    g.Ahi = ih0_hi; g.Alo = ih0_lo; g.Kpad = pad32(F); g.Kc = F;
    g.X = xt; g.sxb = (long)F * Bn; g.sxc = Bn; g.Lx = Bn;
    g.C = xp; g.scb = (long)4 * H * Bn; g.scm = Bn;
    g.M = 4 * H; g.N = Bn; g.B = T; g.perm_h = H;
    if (f16) { g.f16 = 1; g.a_inv = aux + 64; g.x_amax = aux + 128; g.x_namax = 64; g.x_amax_bs = 0; }
    SSV_TRY(ssv_launch_gemm_nn_bf3(g, st));
  }
  GemmNNB g = nnb_zero();
  g.sxc = Bn; g.Lx = Bn; g.scm = Bn; g.srm = Bn;
  g.M = 4 * H; g.N = Bn; g.perm_h = H; g.epi = 1; g.cstate = cbuf;
  if (f16) { g.f16 = 1; g.a_inv = aux + 64; g.x_amax = nullptr; g.x_namax = 0; g.x_amax_bs = 0; }     // activations: |h| < 1, the fixed scale 2^14
  g.lstm_out = out; g.lstm_D = D; g.sbb = (long)8 * H; g.gates_out = keep_gates;
  g.X = out; g.C = out;                        // placeholders: the kernel derives X, X2 and C from (layer, frame)
  // One launch per wavefront step (round 5): layer 0 (K = H: its own h_{t-1}; the input projection xp[t] through R) rides in the launch of the
  // layers above it (K = 2 H) as entry 0.  Before, a step was two launches -- 336 workgroups with 24 chunks, then 672 with 48 -- each with a
  // half-empty last round; together they are 1008 workgroups = two full rounds of 512.  SSV_LSTM_MERGE=0 keeps the two launches (tuning).
  if (presplit) {
    SSV_HIP(hipMemsetAsync(base + s.hp, 0, (size_t)layers * 2 * 2 * s.hp_plane, st));
    g.hs_planes = (unsigned short*)(base + s.hp); g.hs_plane_bytes = (long)s.hp_plane; g.hs_npad = s.npad;
  }
  for (int step = 0; merge && layers >= 2 && step < T + layers - 1; ++step) {
    g.lstm_s = step;
    g.hs_keep_h = !presplit || keep_hs || step == T + layers - 2;   // (pre-split h at inference: the fp32 copy is read by nobody but the caller, from the last step)
    const int lo = step - T + 1 > 0 ? step - T + 1 : 0, hi = step < layers - 1 ? step : layers - 1;
    const int lo1 = lo > 1 ? lo : 1;           // the first layer >= 1 of the launch: its planes are the launch's Ahi
    g.Ahi = (unsigned short*)(base + s.comb + (size_t)(lo1 - 1) * s.comb_stride);
    g.Alo = (unsigned short*)((char*)g.Ahi + split_bytes(4 * H, 2 * H, 1));
    g.sab = (long)(s.comb_stride / sizeof(unsigned short));
    g.Kpad = 2 * H; g.Kc = 2 * H;
    g.xsplit = hch; g.lstm_lo = lo; g.B = hi - lo + 1;
    g.bias = bias + (long)lo * 8 * H; g.bias_b = g.bias + 4 * H;
    g.x0_planes = nullptr;
    if (lo == 0 && x0fold) {
      g.A0hi = l0c_hi; g.A0lo = l0c_lo; g.R = nullptr;
      g.x0_planes = (const unsigned short*)(base + s.x0p); g.x0_amax = aux + 128; g.xsplit0 = s.xsplit0;
    }
    else if (lo == 0) { g.A0hi = hh0_hi; g.A0lo = hh0_lo; g.R = xp + (long)step * 4 * H * Bn; g.srb = 0; }
    else { g.A0hi = g.A0lo = nullptr; g.R = nullptr; }
    SSV_TRY(ssv_launch_gemm_nn_bf3(g, st));
  }
  for (int step = 0; !(merge && layers >= 2) && step < T + layers - 1; ++step) {
    g.lstm_s = step;
    if (step < T) {                            // layer 0: gates = W_hh h_{t-1} + xp[t] + b
      g.Ahi = hh0_hi; g.Alo = hh0_lo; g.Kpad = H; g.Kc = H; g.sab = 0;
      g.xsplit = 0; g.lstm_lo = 0; g.B = 1;
      g.R = xp + (long)step * 4 * H * Bn;
      g.bias = bias; g.bias_b = bias + 4 * H;
      SSV_TRY(ssv_launch_gemm_nn_bf3(g, st));
    }
    const int lo = step - T + 1 > 1 ? step - T + 1 : 1, hi = step < layers - 1 ? step : layers - 1;
    if (lo <= hi) {                            // layers lo..hi: gates = [W_ih | W_hh] [h^{l-1}_t ; h^l_{t-1}] + b
      g.Ahi = (unsigned short*)(base + s.comb + (size_t)(lo - 1) * s.comb_stride);
      g.Alo = (unsigned short*)((char*)g.Ahi + split_bytes(4 * H, 2 * H, 1));
      g.sab = (long)(s.comb_stride / sizeof(unsigned short));
      g.Kpad = 2 * H; g.Kc = 2 * H;
      g.xsplit = hch; g.lstm_lo = lo; g.B = hi - lo + 1;
      g.R = nullptr;
      g.bias = bias + (long)lo * 8 * H; g.bias_b = g.bias + 4 * H;
      SSV_TRY(ssv_launch_gemm_nn_bf3(g, st));
    }
  }
  (void)HN;
  return ssv_launch_transpose_out(out + ((long)(layers - 1) * D + (T - 1) % D) * H * Bn, h_last, H, Bn, st);
}

static int lstm_fwd_impl(const float* x, const float* const* w_ih, const float* const* w_hh, const float* const* b_ih, const float* const* b_hh, float* h_last,
                         int Bn, int T, int F, int H, int layers, void* ws, size_t ws_bytes, ssv_stream_t stream, bool packed);
extern "C" int ssv_lstm_fwd(const float* x, const float* const* w_ih, const float* const* w_hh, const float* const* b_ih,
                            const float* const* b_hh, float* h_last, int Bn, int T, int F, int H, int layers,
                            void* ws, size_t ws_bytes, ssv_stream_t stream) {
  return lstm_fwd_impl(x, w_ih, w_hh, b_ih, b_hh, h_last, Bn, T, F, H, layers, ws, ws_bytes, stream, false);
}
extern "C" int ssv_lstm_fwd_cached(const float* x, const float* const* w_ih, const float* const* w_hh, const float* const* b_ih,
                                   const float* const* b_hh, float* h_last, int Bn, int T, int F, int H, int layers,
                                   void* ws, size_t ws_bytes, int weights_packed, ssv_stream_t stream) {
  return lstm_fwd_impl(x, w_ih, w_hh, b_ih, b_hh, h_last, Bn, T, F, H, layers, ws, ws_bytes, stream, weights_packed != 0);
}
static int lstm_fwd_impl(const float* x, const float* const* w_ih, const float* const* w_hh, const float* const* b_ih, const float* const* b_hh, float* h_last,
                         int Bn, int T, int F, int H, int layers, void* ws, size_t ws_bytes, ssv_stream_t stream, bool packed) {
  SSV_CHECK(x && w_ih && w_hh && b_ih && b_hh && h_last && Bn > 0 && T > 0 && F > 0 && H > 0 && layers > 0, SSV_BAD_SHAPE, "lstm_fwd: bad argument");
  SSV_CHECK(T <= 65535, SSV_UNSUPPORTED, "lstm_fwd: T=%d exceeds grid.y", T);
  const LstmWs s = lstm_ws(Bn, T, F, H);
  SSV_CHECK(ws && ws_bytes >= ssv_lstm_fwd_workspace(Bn, T, F, H, layers), SSV_BAD_SHAPE, "lstm_fwd: workspace too small (%zu < %zu)", ws_bytes,
            ssv_lstm_fwd_workspace(Bn, T, F, H, layers));
  hipStream_t st = (hipStream_t)stream;
  char* base = (char*)ws;
  if (ssv_precision() >= 1 && lstm_wave_ok(Bn, H))
    return lstm_fwd_wave(x, w_ih, w_hh, b_ih, b_hh, h_last, Bn, T, F, H, layers, base, st, nullptr, nullptr, nullptr, nullptr, packed);
  // (the layer-by-layer paths below re-pack per layer into ONE buffer: nothing to keep)
  float* xt = (float*)(base + s.xt);
  float* xp = (float*)(base + s.xp);
  float* seq[2] = {(float*)(base + s.seq0), (float*)(base + s.seq1)};
  float* gbuf = (float*)(base + s.g);
  float* cbuf = (float*)(base + s.c);
  const bool bf3 = ssv_precision() >= 1 && Bn >= 64 && H >= 32;
  SSV_TRY(ssv_launch_lstm_in_transpose(x, xt, Bn, T, F, st));    // [T][F][Bn]
  if (bf3) SSV_TRY(ssv_launch_fill(gbuf, 0.f, (long)H * Bn, st));
  const float* in = xt;
  int Fin = F;
  float* out = nullptr;
  for (int l = 0; l < layers; ++l) {
    out = seq[l & 1];
    if (!bf3) {
      // input projection for every frame at once: xp[t] = W_ih in[t] + b_ih + b_hh    ("batch" = frame)
      SSV_TRY(lstm_gemm_f32(w_ih[l], in, (long)Fin * Bn, xp, (long)4 * H * Bn, b_ih[l], b_hh[l], nullptr, 4 * H, Fin, Bn, T, st));
      for (int t = 0; t < T; ++t) {
        const float* gates = xp + (long)t * 4 * H * Bn;
        if (t > 0) {  // gates = W_hh h_{t-1} + xp[t]
          SSV_TRY(lstm_gemm_f32(w_hh[l], out + (long)(t - 1) * H * Bn, 0, gbuf, 0, nullptr, nullptr, gates, 4 * H, H, Bn, 1, st));
          gates = gbuf;
        }
        SSV_TRY(ssv_launch_lstm_cell(gates, cbuf, out + (long)t * H * Bn, H, Bn, t == 0, st));
      }
    } else {
      // Split-bf16 path.  The layer's weights are used by T + 1 products: split them once, with the 4H output rows
      // re-ordered gate-interleaved (row 4u + gate) so that the recurrent product can finish the cell in its epilogue.
      unsigned short* ih_hi = (unsigned short*)(base + s.wih);
      unsigned short* ih_lo = (unsigned short*)(base + s.wih + split_bytes(4 * H, Fin, 1));
      unsigned short* hh_hi = (unsigned short*)(base + s.whh);
      unsigned short* hh_lo = (unsigned short*)(base + s.whh + split_bytes(4 * H, H, 1));
      SSV_TRY(ssv_launch_pack_split(w_ih[l], ih_hi, ih_lo, 4 * H, Fin, pad32(Fin), 1, Fin, 1, 1, H, st));
      SSV_TRY(ssv_launch_pack_split(w_hh[l], hh_hi, hh_lo, 4 * H, H, pad32(H), 1, H, 1, 1, H, st));
      GemmNNB g = nnb_zero();
      g.X = in; g.sxb = (long)Fin * Bn; g.sxc = Bn; g.Lx = Bn;
      g.bias_b = nullptr; g.sbb = 0; g.R = nullptr; g.srb = 0; g.srm = Bn;
      g.M = 4 * H; g.N = Bn; g.KT = 1;
      g.shift[0] = g.shift[1] = g.shift[2] = 0;
      g.perm_h = H; g.first = 0; g.cstate = nullptr;
      // input projection for every frame at once, biases left to the cell: xp[t] = W_ih in[t]  (gate-interleaved rows)
      g.Ahi = ih_hi; g.Alo = ih_lo; g.Kpad = pad32(Fin); g.Kc = Fin;
      g.C = xp; g.scb = (long)4 * H * Bn; g.scm = Bn; g.bias = nullptr; g.B = T; g.epi = 0;
      SSV_TRY(ssv_launch_gemm_nn_bf3(g, st));
      // recurrent product with the cell finished in its epilogue: h_t, c_t from W_hh h_{t-1} + xp[t] + b_ih + b_hh.
      // t = 0 has no recurrent term; it runs the same kernel on an all-zero h_{-1} (gbuf, zeroed above; one product in T).
      g.Ahi = hh_hi; g.Alo = hh_lo; g.Kpad = pad32(H); g.Kc = H;
      g.sxb = 0; g.scb = 0; g.scm = Bn;
      g.bias = b_ih[l]; g.bias_b = b_hh[l]; g.sbb = 0;
      g.srb = 0; g.srm = Bn;
      g.B = 1; g.epi = 1; g.cstate = cbuf;
      for (int t = 0; t < T; ++t) {
        g.X = (t > 0) ? out + (long)(t - 1) * H * Bn : gbuf;
        g.C = out + (long)t * H * Bn;
        g.R = xp + (long)t * 4 * H * Bn;
        g.first = (t == 0);
        SSV_TRY(ssv_launch_gemm_nn_bf3(g, st));
      }
    }
    in = out;
    Fin = H;
  }
  return ssv_launch_transpose_out(out + (long)(T - 1) * H * Bn, h_last, H, Bn, st);
}

extern "C" size_t ssv_proj_l2norm_fwd_workspace(int Bn, int P) { return align256((size_t)Bn * P * sizeof(float)); }
extern "C" int ssv_proj_l2norm_fwd(const float* h, const float* w, const float* bias, float* e, float* norms, int Bn, int H, int P,
                                   void* ws, size_t ws_bytes, ssv_stream_t stream) {
  SSV_CHECK(h && w && e && Bn > 0 && H > 0 && P > 0, SSV_BAD_SHAPE, "proj_l2norm_fwd: bad argument");
  SSV_CHECK(ws && ws_bytes >= ssv_proj_l2norm_fwd_workspace(Bn, P), SSV_BAD_SHAPE, "proj_l2norm_fwd: workspace too small");
  GemmNN g = nn_zero();                        // y[p][b] = sum_c w[p][c] h[b][c] + bias[p]
  g.A = w; g.sam = H; g.sac = 1; g.saj = 1;
  g.X = h; g.sxc = 1; g.sxn = H; g.Lx = Bn;
  g.C = (float*)ws; g.scm = Bn;
  g.bias = bias;
  g.M = P; g.N = Bn; g.Kc = H; g.B = 1;
  SSV_TRY(ssv_launch_gemm_nn(g, (hipStream_t)stream));
  return ssv_launch_l2norm_rows((const float*)ws, e, norms, P, Bn, (hipStream_t)stream);
}
// Backward of the above: dy = (de - e <e,de>) / |y|;  dh = dy W,  dW = dy^T h,  dbias = column sums of dy.
extern "C" size_t ssv_proj_l2norm_bwd_workspace(int Bn, int P) { return align256((size_t)Bn * P * sizeof(float)); }
extern "C" int ssv_proj_l2norm_bwd(const float* de, const float* e, const float* norms, const float* h, const float* w, float* dh, float* dw,
                                   float* dbias, int Bn, int H, int P, void* ws, size_t ws_bytes, ssv_stream_t stream) {
  SSV_CHECK(de && e && norms && h && w && dh && dw && dbias && Bn > 0 && H > 0 && P > 0, SSV_BAD_SHAPE, "proj_l2norm_bwd: bad argument");
  SSV_CHECK(ws && ws_bytes >= ssv_proj_l2norm_bwd_workspace(Bn, P), SSV_BAD_SHAPE, "proj_l2norm_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  float* dy = (float*)ws;                      // (Bn, P)
  SSV_TRY(ssv_launch_l2norm_bwd(de, e, norms, dy, P, Bn, st));
  SSV_TRY(ssv_launch_colsum(dy, dbias, P, Bn, st));
  GemmNN g = nn_zero();                        // dh (Bn,H): rows b, reduction over p
  g.A = dy; g.sam = P; g.sac = 1; g.saj = 1;
  g.X = w; g.sxc = H; g.Lx = H;
  g.C = dh; g.scm = H;
  g.M = Bn; g.N = H; g.Kc = P; g.B = 1;
  SSV_TRY(ssv_launch_gemm_nn(g, st));
  GemmNN q = nn_zero();                        // dW (P,H) = dy^T h: rows p, reduction over b
  q.A = dy; q.sam = 1; q.sac = P; q.saj = 1;
  q.X = h; q.sxc = H; q.Lx = H;
  q.C = dw; q.scm = H;
  q.M = P; q.N = H; q.Kc = Bn; q.B = 1;
  return ssv_launch_gemm_nn(q, st);
}

// ---- LSTM training: forward that keeps every frame, and backpropagation through time ------------------------------------
// saved (caller-owned, ssv_lstm_saved_bytes): xt [T][F][Bn] | hs [layers][T][H][Bn] | cs [layers][T][H][Bn] |
// gates [layers][T][4H][Bn] (activated i, f, g, o; torch row order).
struct LstmSaved { size_t xt, hs, cs, gates, total; };
static LstmSaved lstm_saved(int Bn, int T, int F, int H, int layers) {
  LstmSaved s;
  s.xt = 0;
  s.hs = s.xt + align256((size_t)T * F * Bn * sizeof(float));
  s.cs = s.hs + align256((size_t)layers * T * H * Bn * sizeof(float));
  s.gates = s.cs + align256((size_t)layers * T * H * Bn * sizeof(float));
  s.total = s.gates + align256((size_t)layers * T * 4 * H * Bn * sizeof(float));
  return s;
}
extern "C" size_t ssv_lstm_saved_bytes(int Bn, int T, int F, int H, int layers) { return lstm_saved(Bn, T, F, H, layers).total; }
// The wavefront kernels exist in the split modes, for hidden sizes that are multiples of 32 and at least 8 utterances.  Everything else -- the
// exact-fp32 mode (ssv_set_precision(0)), any hidden size, any batch -- trains on the exact-fp32 MFMA GEMMs below: layer by layer and frame by
// frame (nn.LSTM + autograd of the reference have no such limits: GE2E/speech_embedder_net.py:19, GE2E/train_speech_embedder.py:82-86).
// Same saved-tensor layout, same cell backward kernel; only the products differ.
static bool lstm_train_split_ok(int Bn, int H) { return ssv_precision() >= 1 && H % 32 == 0 && Bn >= 8; }
static int lstm_train_fwd_f32(const float* x, const float* const* w_ih, const float* const* w_hh, const float* const* b_ih, const float* const* b_hh,
                              float* h_last, int Bn, int T, int F, int H, int layers, char* base, hipStream_t st,
                              float* xt, float* hs, float* cs, float* gates) {
  const LstmWave s = lstm_wave_ws(Bn, T, F, H, layers);
  float* xp = (float*)(base + s.xp);                               // [T][4H][Bn]: the layer's input projection of every frame, biases included
  const long HN = (long)H * Bn;
  SSV_TRY(ssv_launch_lstm_in_transpose(x, xt, Bn, T, F, st));     // [T][F][Bn]
  for (int l = 0; l < layers; ++l) {
    const float* in = l == 0 ? xt : hs + (long)(l - 1) * T * HN;
    const int Fin = l == 0 ? F : H;
    SSV_TRY(lstm_gemm_f32(w_ih[l], in, (long)Fin * Bn, xp, 4 * HN, b_ih[l], b_hh[l], nullptr, 4 * H, Fin, Bn, T, st));
    for (int t = 0; t < T; ++t) {
      float* gt = gates + ((long)l * T + t) * 4 * HN;
      const float* pre = xp + (long)t * 4 * HN;
      if (t > 0) {                                                  // pre-activations = W_hh h_{t-1} + xp[t], into the saved slot (activated in place)
        SSV_TRY(lstm_gemm_f32(w_hh[l], hs + ((long)l * T + t - 1) * HN, 0, gt, 0, nullptr, nullptr, pre, 4 * H, H, Bn, 1, st));
        pre = gt;
      }
      SSV_TRY(ssv_launch_lstm_cell_train(pre, gt, t > 0 ? cs + ((long)l * T + t - 1) * HN : nullptr, cs + ((long)l * T + t) * HN,
                                         hs + ((long)l * T + t) * HN, H, Bn, st));
    }
  }
  return ssv_launch_transpose_out(hs + ((long)(layers - 1) * T + (T - 1)) * HN, h_last, H, Bn, st);
}
extern "C" size_t ssv_lstm_train_fwd_workspace(int Bn, int T, int F, int H, int layers) { return lstm_wave_ws(Bn, T, F, H, layers).total; }
extern "C" int ssv_lstm_train_fwd(const float* x, const float* const* w_ih, const float* const* w_hh, const float* const* b_ih,
                                  const float* const* b_hh, float* h_last, void* saved, int Bn, int T, int F, int H, int layers,
                                  void* ws, size_t ws_bytes, ssv_stream_t stream) {
  SSV_CHECK(x && w_ih && w_hh && b_ih && b_hh && h_last && saved && Bn > 0 && T > 0 && F > 0 && H > 0 && layers > 0, SSV_BAD_SHAPE, "lstm_train_fwd: bad argument");
  SSV_CHECK(ws && ws_bytes >= ssv_lstm_train_fwd_workspace(Bn, T, F, H, layers), SSV_BAD_SHAPE, "lstm_train_fwd: workspace too small");
  const LstmSaved sv = lstm_saved(Bn, T, F, H, layers);
  char* sb = (char*)saved;
  if (!lstm_train_split_ok(Bn, H))
    return lstm_train_fwd_f32(x, w_ih, w_hh, b_ih, b_hh, h_last, Bn, T, F, H, layers, (char*)ws, (hipStream_t)stream,
                              (float*)(sb + sv.xt), (float*)(sb + sv.hs), (float*)(sb + sv.cs), (float*)(sb + sv.gates));
  return lstm_fwd_wave(x, w_ih, w_hh, b_ih, b_hh, h_last, Bn, T, F, H, layers, (char*)ws, (hipStream_t)stream,
                       (float*)(sb + sv.xt), (float*)(sb + sv.hs), (float*)(sb + sv.cs), (float*)(sb + sv.gates));
}

struct LstmBwdWs { size_t dgates, dxa, dxa_slab, dcarry, dhtop, rs, wta, wta_stride, slabs, total; };
static size_t lstm_dw_slab_bytes(int Bn, int T, int H, int Fin) {
  return align256((size_t)dw_splits(T, 4 * H, Fin, 1, Bn) * 4 * H * Fin * sizeof(float));        // "batch" = frames, reduction length = utterances
}
// dxa: the data-gradient products of one reverse wavefront step, [K range z][step parity][layer][2H][Bn] -- layer l's product at frame t is
// [dh^{l-1}_t ; dh^l_{t-1}] (layer 0: only the second half is used), written at step s = l + t under parity s & 1 and read by the cells of step s - 1:
// two parities are the whole life of these values, so the buffer stays in the last-level cache instead of walking through T frames of HBM.
static LstmBwdWs lstm_bwd_ws(int Bn, int T, int F, int H, int layers) {
  LstmBwdWs s;
  s.dgates = 0;
  s.dxa = s.dgates + align256((size_t)layers * T * 4 * H * Bn * sizeof(float));
  s.dxa_slab = align256((size_t)2 * layers * 2 * H * Bn * sizeof(float));
  s.dcarry = s.dxa + 2 * s.dxa_slab;
  s.dhtop = s.dcarry + align256((size_t)layers * H * Bn * sizeof(float));
  s.rs = s.dhtop + align256((size_t)H * Bn * sizeof(float));
  s.wta = s.rs + align256((size_t)layers * T * 4 * H * sizeof(float));      // the bias gradients' per-frame terms, [layer][frame][4H] (lstm_cell_bwd_kernel)
  s.wta_stride = 2 * split_bytes(2 * H, 4 * H, 1);                 // [W_ih | W_hh]^T of a layer (layer 0: the W_ih half stays zero)
  s.slabs = s.wta + (size_t)layers * s.wta_stride;
  // every (items, M, Nc) lstm_weight_grad is called with: W_ih over T frames (Fin = F or H), W_hh over T - 1
  s.total = s.slabs + zmax(zmax(lstm_dw_slab_bytes(Bn, T, H, H), lstm_dw_slab_bytes(Bn, T > 1 ? T - 1 : 1, H, H)), lstm_dw_slab_bytes(Bn, T, H, F));
  return s;
}
extern "C" size_t ssv_lstm_bwd_workspace(int Bn, int T, int F, int H, int layers) { return lstm_bwd_ws(Bn, T, F, H, layers).total; }
// dW (M x Nc) = sum over `items` frames of A_item (M x Bn) X_item^T (Nc x Bn): the conv weight-gradient kernel with time = batch
static int lstm_weight_grad(const float* A, long sab, const float* X, long sxb, float* dw, int M, int Nc, int Bn, int items, void* slabs, hipStream_t st,
                            bool f32 = false) {
  GemmNT g = nt_zero();
  const int Z = dw_splits(items, M, Nc, 1, Bn);
  const long n = (long)M * Nc;
  g.A = A; g.sab = sab; g.sam = Bn; g.La = Bn;
  g.X = X; g.sxb = sxb; g.sxc = Bn; g.Lx = Bn;
  if (Z == 1) { g.C = dw; g.scz = n; g.scm = Nc; g.scc = 1; g.scj = 0; }
  else { g.C = (float*)slabs; g.scz = n; g.scm = Nc; g.scc = 1; g.scj = 0; }
  g.M = M; g.Nc = Nc; g.KT = 1; g.B = items; g.Z = Z; g.bstep = Z;
  if (f32) SSV_TRY(ssv_launch_gemm_nt(g, st));
  else {
    SSV_CHECK(ssv_nt_bf3_fits(g), SSV_UNSUPPORTED, "lstm_bwd: sequence buffers exceed the weight-gradient kernel's 32-bit offsets");
    SSV_TRY(ssv_launch_gemm_nt_bf3(g, st));
  }
  if (Z > 1) SSV_TRY(ssv_launch_reduce_slabs((const float*)slabs, dw, n, Z, n, st));
  return 0;
}
extern "C" int ssv_lstm_bwd(const float* dh_last, const void* saved, const float* const* w_ih, const float* const* w_hh,
                            float* const* dw_ih, float* const* dw_hh, float* const* db_ih, float* const* db_hh,
                            int Bn, int T, int F, int H, int layers, void* ws, size_t ws_bytes, ssv_stream_t stream) {
  SSV_CHECK(dh_last && saved && w_ih && w_hh && dw_ih && dw_hh && db_ih && db_hh && Bn > 0 && T > 0 && F > 0 && H > 0 && layers > 0, SSV_BAD_SHAPE, "lstm_bwd: bad argument");
  const bool f32 = !lstm_train_split_ok(Bn, H);                    // the exact-fp32 products (see lstm_train_fwd_f32)
  const LstmBwdWs s = lstm_bwd_ws(Bn, T, F, H, layers);
  SSV_CHECK(ws && ws_bytes >= s.total, SSV_BAD_SHAPE, "lstm_bwd: workspace too small (%zu < %zu)", ws_bytes, s.total);
  hipStream_t st = (hipStream_t)stream;
  const LstmSaved sv = lstm_saved(Bn, T, F, H, layers);
  const char* sb = (const char*)saved;
  const float* xt = (const float*)(sb + sv.xt);
  const float* hs = (const float*)(sb + sv.hs);
  const float* cs = (const float*)(sb + sv.cs);
  const float* gates = (const float*)(sb + sv.gates);
  char* base = (char*)ws;
  float* dgates = (float*)(base + s.dgates);
  float* dxa = (float*)(base + s.dxa);
  float* dcarry = (float*)(base + s.dcarry);
  float* dhtop = (float*)(base + s.dhtop);
  float* rs = (float*)(base + s.rs);
  const long HN = (long)H * Bn;
  const long zstride = (long)(s.dxa_slab / sizeof(float));
  SSV_TRY(ssv_launch_transpose_out(dh_last, dhtop, Bn, H, st));               // (Bn, H) -> [H][Bn]
  // the transposed product dX [H][Bn] = W^T dG with W (4H x H) row-major, on the exact-fp32 kernel: A(m = q, c = r) = W[r][q]
  auto wt_gemm_f32 = [&](const float* W, const float* dg, float* out) -> int {
    GemmNN q = nn_zero();
    q.A = W; q.sam = 1; q.sac = H; q.saj = 1;
    q.X = dg; q.sxc = Bn; q.Lx = Bn;
    q.C = out; q.scm = Bn;
    q.M = H; q.N = Bn; q.Kc = 4 * H; q.B = 1;
    return ssv_launch_gemm_nn(q, st);
  };
  for (int step = T + layers - 2; f32 && step >= 0; --step) {
    const int lo = step - T + 1 > 0 ? step - T + 1 : 0, hi = step < layers - 1 ? step : layers - 1;
    SSV_TRY(ssv_launch_lstm_cell_bwd(gates, cs, dxa, zstride, 1, dhtop, dgates, dcarry, rs, H, Bn, T, layers, step, lo, hi - lo + 1, st));
    float* outp = dxa + (long)(step & 1) * layers * 2 * HN;                   // this step's parity
    if (lo == 0 && step >= 1) SSV_TRY(wt_gemm_f32(w_hh[0], dgates + (long)step * 4 * HN, outp + HN));
    for (int l = lo > 1 ? lo : 1; l <= hi; ++l) {                            // [dh^{l-1}_t ; dh^l_{t-1}] = [W_ih | W_hh]^T dgates^l_t
      const float* dg = dgates + ((long)l * T + (step - l)) * 4 * HN;
      SSV_TRY(wt_gemm_f32(w_ih[l], dg, outp + (long)l * 2 * HN));
      SSV_TRY(wt_gemm_f32(w_hh[l], dg, outp + (long)l * 2 * HN + HN));
    }
  }
  // transposed weights for the data-gradient products: rows = inputs of the layer, reduction over the 4H gate rows
  const size_t rows_h = (size_t)(H / 16) * (4 * H / 32) * 512;               // elements of the first H rows of a [2H x 4H] plane
  for (int l = 0; !f32 && l < layers; ++l) {
    unsigned short* hi = (unsigned short*)(base + s.wta + (size_t)l * s.wta_stride);
    unsigned short* lo = (unsigned short*)((char*)hi + split_bytes(2 * H, 4 * H, 1));
    if (l == 0) {                                                            // no data gradient of the utterance itself: zero rows (mostly skipped, see skip_rows)
      SSV_HIP(hipMemsetAsync(hi, 0, rows_h * sizeof(unsigned short), st));
      SSV_HIP(hipMemsetAsync(lo, 0, rows_h * sizeof(unsigned short), st));
    } else SSV_TRY(ssv_launch_pack_split(w_ih[l], hi, lo, H, 4 * H, 4 * H, 1, 1, H, 1, 0, st));               // (m=q, k=r) = W_ih[r][q]
    SSV_TRY(ssv_launch_pack_split(w_hh[l], hi + rows_h, lo + rows_h, H, 4 * H, 4 * H, 1, 1, H, 1, 0, st));
  }
  // ONE product launch per reverse wavefront step: every active layer (layer 0 included) x two K ranges of 2H gate rows, on the forward wavefront's
  // 128 x 128 tile -- 3 layers x 12 x 7 x 2 = 504 tiles less layer 0's 84 skipped ones for config 5 (before round 6's end: a 768-row and a 1536-row product
  // on 128 x 32 tiles, 47 + 87 us per step)
  GemmNNB g = nnb_zero();
  g.Kpad = 4 * H; g.Kc = 2 * H; g.ksplit = 2; g.sxc = Bn; g.Lx = Bn; g.scm = Bn; g.N = Bn; g.M = 2 * H;
  g.sab = (long)(s.wta_stride / sizeof(unsigned short)); g.sxb = (long)(T - 1) * 4 * HN; g.scb = 2 * HN; g.scz = zstride;
  for (int step = T + layers - 2; !f32 && step >= 0; --step) {
    const int lo = step - T + 1 > 0 ? step - T + 1 : 0, hi = step < layers - 1 ? step : layers - 1;
    SSV_TRY(ssv_launch_lstm_cell_bwd(gates, cs, dxa, zstride, 2, dhtop, dgates, dcarry, rs, H, Bn, T, layers, step, lo, hi - lo + 1, st));
    if (step == 0) break;                                                    // frame 0 of layer 0: its product would be the gradient of the initial state
    g.Ahi = (unsigned short*)(base + s.wta + (size_t)lo * s.wta_stride);
    g.Alo = (unsigned short*)((char*)g.Ahi + split_bytes(2 * H, 4 * H, 1));
    g.X = dgates + ((long)lo * T + (step - lo)) * 4 * HN;
    g.C = dxa + ((long)(step & 1) * layers + lo) * 2 * HN;
    g.B = hi - lo + 1;
    g.skip_rows = lo == 0 ? H : 0;
    SSV_TRY(ssv_launch_gemm_nn_bf3(g, st));
  }
  // parameter gradients: one reduction over all frames per matrix
  for (int l = 0; l < layers; ++l) {
    const float* dg = dgates + (long)l * T * 4 * HN;
    const int Fin = l == 0 ? F : H;
    const float* in = l == 0 ? xt : hs + (long)(l - 1) * T * HN;
    SSV_TRY(lstm_weight_grad(dg, 4 * HN, in, (long)Fin * Bn, dw_ih[l], 4 * H, Fin, Bn, T, base + s.slabs, st, f32));
    if (T > 1) SSV_TRY(lstm_weight_grad(dg + 4 * HN, 4 * HN, hs + (long)l * T * HN, HN, dw_hh[l], 4 * H, H, Bn, T - 1, base + s.slabs, st, f32));
    else SSV_TRY(ssv_launch_fill(dw_hh[l], 0.f, (long)4 * H * H, st));
    SSV_TRY(ssv_launch_reduce_slabs(rs + (long)l * T * 4 * H, db_ih[l], 4 * H, T, 4 * H, st));      // sum over frames of sum_b dgates[l][t][r][b] (the cell kernel's row sums)
    SSV_HIP(hipMemcpyAsync(db_hh[l], db_ih[l], (size_t)4 * H * sizeof(float), hipMemcpyDeviceToDevice, st));
  }
  return 0;
}
