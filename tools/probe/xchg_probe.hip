// Hardware probe (GPU box) for the distributed-statistics fused conv + LayerNorm / gate (DESIGN: "fused K1"): what does it cost the eight row-tile
// workgroups of one column tile to exchange their partial column sums through memory, and do they always run at the same time?
//
// Every workgroup publishes NW 64-bit words (tag << 32 | payload) with agent-scope relaxed atomic stores -- no release fence, no L2 write-back:
// each word validates itself -- and then reads the NW words of each of its 7 peers with agent-scope atomic loads, polling a word until its tag is
// this launch's epoch.  Every spin is BOUNDED: after `limit` polls the thread gives up and the workgroup reports a failure (so a wrong
// co-residency guess is a number in this table, never a hung GPU).
//   mode 0: peers are consecutive in the XCD-aware order (ssv_xcd_order: the eight of a group share an XCD, as the conv kernel's row tiles do)
//   mode 1: peers are consecutive launch ids (dealt round-robin: eight different XCDs)
//   mode 2: publish only (the kernel's fixed cost)
// The epoch comes from device memory (incremented by a one-thread kernel in front of every launch), so the pair can be captured in a hipGraph and
// replayed: question (a) of the review -- do <= resident-capacity workgroups always co-reside under replay -- is "failures == 0 over all replays".
// build: hipcc -O3 --offload-arch=gfx950 -Wno-unused-value tools/probe/xchg_probe.hip -o tools/probe/xchg_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

struct Rec { unsigned long long t_enter, t_pub, t_done; unsigned xcc, polls, fail, pad; };

__device__ __forceinline__ unsigned xcd_order(unsigned lin, unsigned total) {
  const unsigned xcd = lin & 7u, idx = lin >> 3, q = total >> 3, r = total & 7u;
  return xcd * q + (xcd < r ? xcd : r) + idx;
}
__global__ void bump(unsigned* epoch) { *epoch += 1; }

template <int REGS>
__global__ __launch_bounds__(256) void xchg(unsigned long long* words, const unsigned* epoch_dev, int mode, Rec* rec, unsigned* fails, int nw, int limit, float* sink) {
  extern __shared__ float sm[];
  __shared__ unsigned s_polls, s_fail;
  float r[REGS];
#pragma unroll
  for (int i = 0; i < REGS; ++i) r[i] = threadIdx.x * 0.5f + i;     // register footprint of the conv kernel: residency as in the real launch
  if (threadIdx.x == 0) { s_polls = 0; s_fail = 0; }
  sm[threadIdx.x] = r[0];
  __syncthreads();
  const unsigned lin = blockIdx.x, total = gridDim.x;
  const unsigned id = mode == 1 ? lin : xcd_order(lin, total);
  const unsigned group = id >> 3, me = id & 7u;
  const unsigned epoch = __hip_atomic_load(epoch_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long* mine = words + (size_t)(group * 8 + me) * nw;
  for (int i = threadIdx.x; i < nw; i += 256)
    __hip_atomic_store(mine + i, ((unsigned long long)epoch << 32) | (id * 4096u + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  unsigned polls = 0, fail = 0, acc = 0;
  if (mode != 2) {
    for (unsigned p = 0; p < 8; ++p) {
      if (p == me) continue;
      const unsigned long long* theirs = words + (size_t)(group * 8 + p) * nw;
      for (int i = threadIdx.x; i < nw; i += 256) {
        unsigned long long v;
        int tries = 0;
        for (;;) {
          v = __hip_atomic_load(theirs + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((unsigned)(v >> 32) == epoch) break;
          if (++tries > limit) { fail = 1; break; }
        }
        polls += tries;
        if (!fail && (unsigned)v != (group * 8 + p) * 4096u + i) fail = 2;
        acc += (unsigned)v;
      }
    }
  }
  atomicAdd(&s_polls, polls);
  atomicMax(&s_fail, fail);
  __syncthreads();
  const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
  float s = sm[(threadIdx.x + 1) & 255] + (float)acc;
#pragma unroll
  for (int i = 0; i < REGS; ++i) s += r[i] * s;
  if (nw < 0) sink[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    Rec q;
    q.t_enter = t0; q.t_pub = t1; q.t_done = t2; q.polls = s_polls; q.fail = s_fail; q.pad = 0;
    q.xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf;      // HW_REG_XCC_ID[3:0]
    rec[lin] = q;
    if (s_fail) atomicAdd(fails, 1u);
  }
}

static double pct(std::vector<double>& v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; }

template <int REGS>
static void run(const char* what, int wgs, int lds, int mode, int nw, bool graph) {
  unsigned long long* words; Rec* rec; unsigned *epoch, *fails; float* sink;
  const int groups = (wgs + 7) / 8;
  hipMalloc(&words, (size_t)groups * 8 * nw * 8); hipMemset(words, 0, (size_t)groups * 8 * nw * 8);
  hipMalloc(&rec, sizeof(Rec) * wgs); hipMalloc(&epoch, 4); hipMalloc(&fails, 4); hipMalloc(&sink, 1 << 20);
  hipMemset(epoch, 0, 4); hipMemset(fails, 0, 4);
  hipStream_t st; hipStreamCreate(&st);
  const int limit = 1 << 16, reps = 200;
  auto pair = [&]() {
    hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, st, epoch);
    hipLaunchKernelGGL((xchg<REGS>), dim3(wgs), dim3(256), lds, st, words, epoch, mode, rec, fails, nw, limit, sink);
  };
  hipGraph_t g; hipGraphExec_t ge;
  if (graph) {
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    pair();
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  }
  auto once = [&]() { if (graph) hipGraphLaunch(ge, st); else pair(); };
  for (int i = 0; i < 10; ++i) once();
  hipStreamSynchronize(st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, st);
  for (int i = 0; i < reps; ++i) once();
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<Rec> h(wgs);
  hipMemcpy(h.data(), rec, sizeof(Rec) * wgs, hipMemcpyDeviceToHost);
  unsigned nf; hipMemcpy(&nf, fails, 4, hipMemcpyDeviceToHost);
  std::vector<double> resid, pub, enter;
  unsigned long long tmin = ~0ull;
  unsigned long long polls = 0;
  for (auto& q : h) tmin = std::min(tmin, q.t_enter);
  int mixed = 0;
  for (int w = 0; w < wgs; ++w) {
    resid.push_back((h[w].t_done - h[w].t_enter) * 0.01); pub.push_back((h[w].t_pub - h[w].t_enter) * 0.01); enter.push_back((h[w].t_enter - tmin) * 0.01);
    polls += h[w].polls;
  }
  // XCDs per group of eight peers (last launch)
  for (int gi = 0; gi + 1 <= groups && gi * 8 + 7 < wgs; ++gi) {
    unsigned mask = 0;
    for (int w = 0; w < wgs; ++w) {
      const unsigned lin = w, total = wgs, xcd = lin & 7u, idx = lin >> 3, q = total >> 3, r = total & 7u;
      const unsigned id = mode == 1 ? lin : xcd * q + (xcd < r ? xcd : r) + idx;
      if ((int)(id >> 3) == gi) mask |= 1u << h[w].xcc;
    }
    if (__builtin_popcount(mask) > 1) ++mixed;
  }
  printf("%-34s %5d wg %2d KB LDS ~%3d VGPR nw=%4d %s: %6.2f us/launch (pair)  | enter 50/90/max %.2f/%.2f/%.2f us | publish %.2f | residence 10/50/90/max %.2f/%.2f/%.2f/%.2f us | polls/wg %.1f | groups on >1 XCD %d/%d | FAILED workgroups over all launches: %u\n",
         what, wgs, lds / 1024, REGS + 8, nw, graph ? "graph" : "eager", ms * 1000 / reps, pct(enter, 0.5), pct(enter, 0.9), pct(enter, 1.0), pct(pub, 0.5),
         pct(resid, 0.1), pct(resid, 0.5), pct(resid, 0.9), pct(resid, 1.0), (double)polls / wgs, mixed, groups, nf);
  fflush(stdout);
  if (graph) { hipGraphExecDestroy(ge); hipGraphDestroy(g); }
  hipFree(words); hipFree(rec); hipFree(epoch); hipFree(fails); hipFree(sink); hipStreamDestroy(st);
}

int main() {
  // the C = 256 / L = 325 forward launch: 768 workgroups, 33 KB LDS, ~60 VGPRs, 112 columns x 4 partial sums per row tile = 448 words
  run<52>("publish only", 768, 33 * 1024, 2, 448, false);
  run<52>("peers share an XCD", 768, 33 * 1024, 0, 448, false);
  run<52>("peers share an XCD", 768, 33 * 1024, 0, 448, true);
  run<52>("peers on eight XCDs", 768, 33 * 1024, 1, 448, false);
  run<52>("peers on eight XCDs", 768, 33 * 1024, 1, 448, true);
  // the C = 512 / L = 186 launch: 512 workgroups of the 128-row tile (~110 VGPRs, 42 KB LDS), 96 columns x 4
  run<104>("C=512 L=186 shape, one XCD", 512, 42 * 1024, 0, 384, true);
  // more workgroups than can be resident (L = 1300: 3072): peers are dispatched in order; does the bounded spin ever trip?
  run<52>("3072 wg (> residency), one XCD", 3072, 33 * 1024, 0, 448, true);
  run<52>("3072 wg (> residency), 8 XCDs", 3072, 33 * 1024, 1, 448, true);
  return 0;
}
