// Hardware probe (GPU box): what rate does L2 -> CU operand traffic reach for the access shapes the GEMM kernels use?
// 16 workgroups of a group read the same rows at the same time (15/16 of the requests are L2 hits), four groups per XCD.
//   mode 0: dword per lane, 256 B contiguous per wave-instruction (input rows of the forward conv)
//   mode 1: dwordx4, 16-byte pieces with 16-byte gaps, two instructions per 32-byte window (weight-gradient windows)
//   mode 2: dwordx4, 1 KB contiguous per wave-instruction
//   mode 3: dwordx4, lanes 8 per 128-byte line segment, rows 1300 B apart (window rows, contiguous pieces)
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/l2_stream.hip -o tools/probe/l2_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MODE, int DEPTH>
__global__ __launch_bounds__(256, 2) void probe(const float* __restrict__ src, float* __restrict__ out, long group_bytes, int iters, int row_floats) {
  const int w = blockIdx.x;
  const int xcd = w & 7, slot = w >> 3;
  const int group = xcd * (gridDim.x / 128) + slot / 16;            // 16 workgroups per group, groups of one XCD adjacent
  const char* base = reinterpret_cast<const char*>(src) + (long)group * group_bytes;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float acc = 0.f;
  // one "tile" = 256 threads x DEPTH x 16 bytes (dword mode: DEPTH x 4 loads of 4 bytes)
  const long tile_bytes = 256L * DEPTH * 16;
  const long ntiles = group_bytes / tile_bytes;
  for (int it = 0; it < iters; ++it) {
    const char* tb = base + (long)(it % ntiles) * tile_bytes;
    if constexpr (MODE == 0) {
      float v[DEPTH * 4];
#pragma unroll
      for (int i = 0; i < DEPTH * 4; ++i) v[i] = *reinterpret_cast<const float*>(tb + ((long)(i * 4 + wave) * 256 + lane * 4));
#pragma unroll
      for (int i = 0; i < DEPTH * 4; ++i) acc += v[i];
    } else {
      float4 v[DEPTH];
#pragma unroll
      for (int i = 0; i < DEPTH; ++i) {
        long off;
        if (MODE == 1) {          // window w_ = tid + 256 * (i / 2), piece i & 1: [32 w_ + 16 (i&1))
          off = (long)(tid + 256 * (i / 2)) * 32 + 16 * (i & 1);
        } else if (MODE == 2) {
          off = (long)(i * 256 + tid) * 16;
        } else {                  // rows of 256 B taken from a matrix with row stride row_floats: 8 lanes per row piece of 128 B
          const int f = tid + 256 * (i / 2);
          off = ((long)(f / 8) * row_floats * 4 + (f % 8) * 16 + 128 * (i & 1)) % tile_bytes / 16 * 16;
        }
        v[i] = *reinterpret_cast<const float4*>(tb + off);
      }
#pragma unroll
      for (int i = 0; i < DEPTH; ++i) acc += v[i].x + v[i].y + v[i].z + v[i].w;
    }
  }
  if (acc == 12345.678f) out[w] = acc;
}

template <int MODE, int DEPTH>
static void run(const float* src, float* out, long group_bytes, int nwg, const char* name) {
  if (nwg % 128 != 0 || nwg / 16 > 48) { printf("bad grid\n"); exit(1); }
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<MODE, DEPTH>), dim3(nwg), dim3(256), 0, 0, src, out, group_bytes, iters, 325);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)nwg * iters * 256 * DEPTH * 16;
  printf("%-44s depth %2d  wgs %4d  %.2f TB/s into the CUs (%.1f GB/s per CU)\n", name, DEPTH, nwg, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
  fflush(stdout);
}

int main() {
  const int nwg = 512;
  const long group_bytes = 8L << 20;
  const int ngroups = 768 / 16;             // the largest grid below
  float *src, *out;
  hipMalloc(&src, ngroups * group_bytes);
  hipMalloc(&out, 4096);
  hipMemset(src, 0, ngroups * group_bytes);
  run<0, 2>(src, out, group_bytes, nwg, "dword, 256 B rows");
  run<0, 4>(src, out, group_bytes, nwg, "dword, 256 B rows");
  run<0, 8>(src, out, group_bytes, nwg, "dword, 256 B rows");
  run<1, 4>(src, out, group_bytes, nwg, "dwordx4, 16 B pieces with 16 B gaps");
  run<1, 8>(src, out, group_bytes, nwg, "dwordx4, 16 B pieces with 16 B gaps");
  run<1, 16>(src, out, group_bytes, nwg, "dwordx4, 16 B pieces with 16 B gaps");
  run<2, 4>(src, out, group_bytes, nwg, "dwordx4, 1 KB contiguous");
  run<2, 8>(src, out, group_bytes, nwg, "dwordx4, 1 KB contiguous");
  run<2, 16>(src, out, group_bytes, nwg, "dwordx4, 1 KB contiguous");
  run<3, 4>(src, out, group_bytes, nwg, "dwordx4, 128 B row pieces");
  run<3, 8>(src, out, group_bytes, nwg, "dwordx4, 128 B row pieces");
  run<2, 8>(src, out, group_bytes, 768, "dwordx4, 1 KB contiguous");
  run<0, 8>(src, out, group_bytes, 768, "dword, 256 B rows");
  return 0;
}
