// Hardware probe (GPU box): does what one launch leaves in the XCDs' L2 serve the NEXT launch?  The split-MFMA kernels of a training step and the 122 launches of the
// LSTM wavefront each re-read what the launch before them wrote or read (weight planes, activations), so the answer prices every kernel boundary.
// A reader kernel streams a buffer of S bytes with the same workgroup -> address map every time (each XCD re-reads its own share), 16-byte loads, grid = 2048 workgroups.
//   cold:   after a 1 GiB scrub buffer was streamed (nothing of the buffer in L2 or in the memory-side cache)
//   warm:   the same launch again, back to back                      (L2 hit if lines survive the kernel boundary and S fits the 8 x 4 MB; else MALL / HBM)
//   twice:  ONE launch that reads the buffer twice, second pass timed by difference  (what an L2 hit is worth inside a kernel)
//   w -> r: a writer launch fills the buffer (same map), then the reader  (producer / consumer across a boundary)
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/l2_persist.hip -o tools/probe/l2_persist
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));

// stamps: s_memrealtime (100 MHz) at a workgroup's entry and exit, [2 w] / [2 w + 1]: the host takes max(exit) - min(entry), the launch floor left out
__global__ __launch_bounds__(256) void reader(const f4* __restrict__ src, float* __restrict__ out, long n16, int passes, unsigned long long* __restrict__ stamps) {
  if (threadIdx.x == 0) stamps[2 * blockIdx.x] = wall_clock64();
  // workgroup w owns the contiguous range [w, w + 1) * n16 / gridDim.x: the hardware deals workgroups to XCDs by w % 8, the same way every launch
  const long per = n16 / gridDim.x;
  const f4* p = src + (long)blockIdx.x * per;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int ps = 0; ps < passes; ++ps)
    for (long i = threadIdx.x; i < per; i += 1024) {
      f4 a = p[i], b = i + 256 < per ? p[i + 256] : acc, c = i + 512 < per ? p[i + 512] : acc, d = i + 768 < per ? p[i + 768] : acc;
      acc += a; acc += b; acc += c; acc += d;
    }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 1234.5f) out[blockIdx.x] = acc[0];
  __syncthreads();
  if (threadIdx.x == 0) stamps[2 * blockIdx.x + 1] = wall_clock64();
}
__global__ __launch_bounds__(256) void writer(f4* __restrict__ dst, long n16, float v) {
  const long per = n16 / gridDim.x;
  f4* p = dst + (long)blockIdx.x * per;
  const f4 x = {v, v, v, v};
  for (long i = threadIdx.x; i < per; i += 256) p[i] = x;
}
static unsigned long long* g_stamps;
static float timed(hipEvent_t e0, hipEvent_t e1) {            // device-side span of the launch just recorded, in ms
  hipEventSynchronize(e1);
  static std::vector<unsigned long long> h(2 * 2048);
  hipMemcpy(h.data(), g_stamps, h.size() * 8, hipMemcpyDeviceToHost);
  unsigned long long lo = ~0ull, hi = 0;
  for (int w = 0; w < 2048; ++w) { lo = std::min(lo, h[2 * w]); hi = std::max(hi, h[2 * w + 1]); }
  return (float)((hi - lo) * 1e-5);                          // 100 MHz ticks -> ms
}

int main() {
  const long scrub_bytes = 1L << 30;
  f4 *buf, *scrub; float* out;
  hipMalloc(&buf, 1L << 30); hipMalloc(&scrub, scrub_bytes); hipMalloc(&out, 1 << 16); hipMalloc(&g_stamps, 2 * 2048 * 8);
  hipMemset(buf, 0, 1L << 30); hipMemset(scrub, 0, scrub_bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int G = 2048;
  auto med = [](std::vector<float>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  printf("%10s %12s %12s %12s %12s   (GB/s over the device-side span of the launch: first workgroup entry to last exit)\n", "bytes", "cold", "warm", "2nd pass", "after write");
  for (long S : {1L << 20, 4L << 20, 16L << 20, 32L << 20, 64L << 20, 128L << 20, 256L << 20, 512L << 20}) {
    const long n16 = S / 16;
    std::vector<float> cold, warm, twice, once, wr;
    for (int rep = 0; rep < 7; ++rep) {
      hipLaunchKernelGGL(reader, dim3(G), dim3(256), 0, 0, scrub, out, scrub_bytes / 16, 1, g_stamps);
      hipEventRecord(e0); hipLaunchKernelGGL(reader, dim3(G), dim3(256), 0, 0, buf, out, n16, 1, g_stamps); hipEventRecord(e1); cold.push_back(timed(e0, e1));
      hipEventRecord(e0); hipLaunchKernelGGL(reader, dim3(G), dim3(256), 0, 0, buf, out, n16, 1, g_stamps); hipEventRecord(e1); warm.push_back(timed(e0, e1));
      hipLaunchKernelGGL(reader, dim3(G), dim3(256), 0, 0, scrub, out, scrub_bytes / 16, 1, g_stamps);
      hipEventRecord(e0); hipLaunchKernelGGL(reader, dim3(G), dim3(256), 0, 0, buf, out, n16, 2, g_stamps); hipEventRecord(e1); twice.push_back(timed(e0, e1));
      hipLaunchKernelGGL(reader, dim3(G), dim3(256), 0, 0, scrub, out, scrub_bytes / 16, 1, g_stamps);
      hipLaunchKernelGGL(writer, dim3(G), dim3(256), 0, 0, buf, n16, 1.f);
      hipEventRecord(e0); hipLaunchKernelGGL(reader, dim3(G), dim3(256), 0, 0, buf, out, n16, 1, g_stamps); hipEventRecord(e1); wr.push_back(timed(e0, e1));
    }
    const float c = med(cold), w = med(warm), t2 = med(twice) - c, a = med(wr);
    printf("%10ld %12.0f %12.0f %12.0f %12.0f   (ms: %.4f %.4f %.4f %.4f)\n", S, S / (c * 1e6), S / (w * 1e6), t2 > 0 ? S / (t2 * 1e6) : 0.0, S / (a * 1e6), c, w, t2, a);
  }
  return 0;
}
