// Hardware probe (GPU box): what a raw buffer load (stride 0, num_records = the tensor's bytes) returns outside its range --
// a 16-byte load straddling the end, one starting past it, one at a "negative" (wrapped) offset, and a negative scalar offset.
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/buf_oob.hip -o tools/probe/buf_oob
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u4 __attribute__((vector_size(16)));
__global__ void k(const float* x, int n, float* out) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, n * 4, 0x00020000);
  const int offs[6] = {(n - 4) * 4, (n - 2) * 4, n * 4, (n + 100) * 4, -8, -16};
  for (int c = 0; c < 6; ++c) {
    const uint4 v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, offs[c], 0, 0));
    out[c * 4 + 0] = __builtin_bit_cast(float, v.x); out[c * 4 + 1] = __builtin_bit_cast(float, v.y); out[c * 4 + 2] = __builtin_bit_cast(float, v.z); out[c * 4 + 3] = __builtin_bit_cast(float, v.w);
  }
  uint4 v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, 32, -48, 0));      // voffset 32 + soffset -48 = -16
  out[24] = __builtin_bit_cast(float, v.x); out[25] = __builtin_bit_cast(float, v.y); out[26] = __builtin_bit_cast(float, v.z); out[27] = __builtin_bit_cast(float, v.w);
  v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, 64, -48, 0));         // = +16: in range
  out[28] = __builtin_bit_cast(float, v.x); out[29] = __builtin_bit_cast(float, v.y); out[30] = __builtin_bit_cast(float, v.z); out[31] = __builtin_bit_cast(float, v.w);
  v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, 16, 48, 0));          // voffset 16 + soffset 48 = 64: in range
  out[32] = __builtin_bit_cast(float, v.x); out[33] = __builtin_bit_cast(float, v.y); out[34] = __builtin_bit_cast(float, v.z); out[35] = __builtin_bit_cast(float, v.w);
  v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, 16, (n - 6) * 4, 0)); // voffset 16 + soffset (n-6)*4: straddles by 2
  out[36] = __builtin_bit_cast(float, v.x); out[37] = __builtin_bit_cast(float, v.y); out[38] = __builtin_bit_cast(float, v.z); out[39] = __builtin_bit_cast(float, v.w);
}
int main() {
  const int n = 1000, pad = 64;
  float *buf, *out, h[40];
  hipMalloc(&buf, (n + 2 * pad) * 4); hipMalloc(&out, 160);
  float* hb = new float[n + 2 * pad];
  for (int i = 0; i < n + 2 * pad; ++i) hb[i] = (float)(i - pad);      // x[i] = i; the pads hold -64..-1 and 1000..1063 (must never show up)
  hipMemcpy(buf, hb, (n + 2 * pad) * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, buf + pad, n, out);
  hipMemcpy(h, out, 160, hipMemcpyDeviceToHost);
  const char* nm[10] = {"last 4 elements", "straddling the end by 2", "starting at the end", "far past the end", "offset -8", "offset -16", "voffset 32 + soffset -48", "voffset 64 + soffset -48 (= 16)", "voffset 16 + soffset 48 (= 64)", "voffset 16 + soffset (n-6)*4"};
  for (int c = 0; c < 10; ++c) printf("%-34s %g %g %g %g\n", nm[c], h[c * 4], h[c * 4 + 1], h[c * 4 + 2], h[c * 4 + 3]);
  return 0;
}
