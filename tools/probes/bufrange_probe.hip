// Probe: raw buffer load range-check semantics on gfx950 (is soffset part of the bounds check?)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* src, int nrec, int voff, int soff, float* out) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nrec, 0x00020000);
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  out[0] = __builtin_bit_cast(float, v[0]);
}
int main() {
  float *src, *out; (void)hipMalloc(&src, 1 << 20); (void)hipMalloc(&out, 16);
  float h[1 << 18]; for (int i = 0; i < (1 << 18); ++i) h[i] = (float)(i + 1);
  (void)hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
  struct { int nrec, voff, soff; } cases[] = {
      {4096, 0, 0}, {4096, 4080, 0}, {4096, 4096, 0}, {4096, 0, 4080}, {4096, 0, 4096}, {4096, 2048, 2048},
      {4096, 2048, 2032}, {4096, 1024, 8192}, {0, 0, 0}, {0, 0, 4096}, {0, 16, 4096}, {4096, -16, 0},
      {4096, -16, 4096}, {4096, 0x7fffff00, 0}, {4096, (int)0x80000000u, 0}, {4096, (int)0x80000000u, 1024}};
  for (auto c : cases) {
    float o = -1.f;
    (void)hipMemset(out, 0xff, 16);
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, src, c.nrec, c.voff, c.soff, out);
    (void)hipMemcpy(&o, out, 4, hipMemcpyDeviceToHost);
    printf("num_records %6d voffset %11d soffset %6d -> %g  (in-range value would be %g)\n", c.nrec, c.voff, c.soff, o,
           (double)(((long long)c.voff + c.soff) / 4 + 1));
  }
  return 0;
}
