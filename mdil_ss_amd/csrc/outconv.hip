// Decoder.output_conv forward: ConvTranspose2d(16, nc, 2, stride 2) on NHWC fp32 (gfx950).
//
//   out[n, 2h+a, 2w+b, c] = bias[c] + sum_ci x[n, h, w, ci] * W[ci][c][a][b]
//
// Stride == kernel size: the four output parity classes are four independent 16 -> nc pointwise
// maps of the same input pixel, so ONE pass reads x once (50 MB at config 3) and writes the
// logits once (252 MB) -- the generic tap-conv path needed one launch per parity class and read x
// four times.  1,280 FMAs per input pixel against 1.3 KB of traffic: HBM-bound (write-bound), so
// the contraction runs on the VALU with the weights broadcast from LDS; a thread owns one (input
// pixel, b) pair, lane pairs write the two adjacent output pixels 2w, 2w+1, a wave 5 KB
// contiguous per output row.
#include "common.h"

namespace {

template <int NC, int P>
__global__ __launch_bounds__(MDIL_WG) void outconv_fwd_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ w,
                                                              const float* __restrict__ bias,
                                                              long long npix, int H, int W,
                                                              float* __restrict__ out) {
  MDIL_HBM_KERNEL_PRIO();
  static_assert(P % 4 == 0 && P >= NC && P - NC < 4, "row pitch");
  __shared__ __attribute__((aligned(16))) float Wl[2][2][16][P];   // [a][b][ci][c], pad = 0
  __shared__ __attribute__((aligned(16))) float Bl[P];
  for (int i = threadIdx.x; i < 2 * 2 * 16 * P; i += MDIL_WG) {
    const int c = i % P, ci = (i / P) % 16, b = (i / (P * 16)) % 2, a = i / (P * 32);
    Wl[a][b][ci][c] = c < NC ? w[((ci * NC + c) * 2 + a) * 2 + b] : 0.f;
  }
  for (int i = threadIdx.x; i < P; i += MDIL_WG) Bl[i] = i < NC ? bias[i] : 0.f;
  __syncthreads();

  // thread = (input pixel, output row a): it writes the two adjacent output pixels 2w, 2w+1 of
  // that row (160 contiguous bytes; consecutive lanes continue the row).  The weight reads are
  // wave-uniform LDS broadcasts; the asm statement keeps them inside the loop (hoisted, the 640
  // loop-invariant registers spill).
  const long long total = npix * 2;
  for (long long i = (long long)blockIdx.x * MDIL_WG + threadIdx.x; i < total;
       i += (long long)gridDim.x * MDIL_WG) {
    asm volatile("" ::: "memory");
    const int a = (int)((i / W) & 1);          // rows alternate per run of W threads
    const long long q = (i / (2 * (long long)W)) * W + i % W;
    f32x4 xv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) xv[k] = *reinterpret_cast<const f32x4*>(x + q * 16 + k * 4);
    const int wi = (int)(q % W);
    const long long r = q / W;                 // n * H + h
    float* o = out + ((2 * r + a) * 2 * W + 2 * wi) * P;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      f32x4 acc[P / 4];
#pragma unroll
      for (int j = 0; j < P / 4; ++j) acc[j] = *reinterpret_cast<const f32x4*>(&Bl[j * 4]);
#pragma unroll
      for (int ci = 0; ci < 16; ++ci) {
        const float xs = xv[ci >> 2][ci & 3];
#pragma unroll
        for (int j = 0; j < P / 4; ++j)
          acc[j] += xs * *reinterpret_cast<const f32x4*>(&Wl[a][b][ci][j * 4]);
      }
#pragma unroll
      for (int j = 0; j < P / 4; ++j) *reinterpret_cast<f32x4*>(o + b * P + j * 4) = acc[j];
    }
  }
}

}  // namespace

extern "C" int mdil_outconv_fwd(const float* x, const float* w, const float* bias, int N, int H, int W,
                                int nc, int pitch, float* out, void* stream) {
  MDIL_CHECK_ARG(x && w && bias && out && N > 0 && H > 0 && W > 0, "outconv_fwd: bad argument");
  const long long npix = (long long)N * H * W;
  long long blocks = (npix * 2 + MDIL_WG - 1) / MDIL_WG;
  const int grid = (int)(blocks > 8192 ? 8192 : blocks);
  if (nc == 20 && pitch == 20)
    hipLaunchKernelGGL((outconv_fwd_kernel<20, 20>), dim3(grid), dim3(MDIL_WG), 0, (hipStream_t)stream,
                       x, w, bias, npix, H, W, out);
  else if (nc == 27 && pitch == 28)
    hipLaunchKernelGGL((outconv_fwd_kernel<27, 28>), dim3(grid), dim3(MDIL_WG), 0, (hipStream_t)stream,
                       x, w, bias, npix, H, W, out);
  else {
    mdil_set_error("outconv_fwd: unsupported nc=%d pitch=%d", nc, pitch);
    return MDIL_ERR_UNSUPPORTED;
  }
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}
