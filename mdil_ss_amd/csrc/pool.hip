// MaxPool2d(2, stride 2) half of the DownsamplerBlock, NHWC fp32, written straight into the
// channel slice of the concatenated tensor (no torch.cat copy).  HBM-bound, channel-fastest.
#include "common.h"

namespace {

__global__ __launch_bounds__(MDIL_WG) void maxpool_fwd_kernel(const float* __restrict__ x, int N,
                                                              int H, int W, int C,
                                                              float* __restrict__ z, int z_pitch,
                                                              int coff) {
  MDIL_HBM_KERNEL_PRIO();

  const int HO = H >> 1, WO = W >> 1;
  const long long total = (long long)N * HO * WO * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long p = i / C;
    const int wo = (int)(p % WO);
    const int ho = (int)((p / WO) % HO);
    const int n = (int)(p / ((long long)WO * HO));
    const float* b = x + (((long long)n * H + 2 * ho) * W + 2 * wo) * C + c;
    float m = b[0];
    float v = b[C];
    if (v > m) m = v;
    v = b[(long long)W * C];
    if (v > m) m = v;
    v = b[(long long)W * C + C];
    if (v > m) m = v;
    z[p * z_pitch + coff + c] = m;
  }
}

// first maximum in scan order (dh,dw) = (0,0),(0,1),(1,0),(1,1) receives the gradient, like
// ATen's max_pool2d_with_indices (strict '>' update).
__global__ __launch_bounds__(MDIL_WG) void maxpool_bwd_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ gz, int N,
                                                              int H, int W, int C, int z_pitch,
                                                              int coff, float* __restrict__ gx) {
  MDIL_HBM_KERNEL_PRIO();

  const int HO = H >> 1, WO = W >> 1;
  const long long total = (long long)N * HO * WO * C;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long long p = i / C;
    const int wo = (int)(p % WO);
    const int ho = (int)((p / WO) % HO);
    const int n = (int)(p / ((long long)WO * HO));
    const long long base = (((long long)n * H + 2 * ho) * W + 2 * wo) * C + c;
    const long long o1 = C, o2 = (long long)W * C, o3 = (long long)W * C + C;
    float m = x[base];
    int arg = 0;
    float v = x[base + o1];
    if (v > m) { m = v; arg = 1; }
    v = x[base + o2];
    if (v > m) { m = v; arg = 2; }
    v = x[base + o3];
    if (v > m) { m = v; arg = 3; }
    const float g = gz[p * z_pitch + coff + c];
    gx[base] = arg == 0 ? g : 0.f;
    gx[base + o1] = arg == 1 ? g : 0.f;
    gx[base + o2] = arg == 2 ? g : 0.f;
    gx[base + o3] = arg == 3 ? g : 0.f;
  }
}

inline int grid_for(long long total) {
  long long b = (total + MDIL_WG - 1) / MDIL_WG;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int mdil_maxpool_concat_fwd(const float* x, int N, int H, int W, int C, float* z,
                                       int z_pitch, int coff, void* stream) {
  MDIL_CHECK_ARG(x && z && (H % 2 == 0) && (W % 2 == 0), "maxpool_fwd: bad argument");
  const long long total = (long long)N * (H / 2) * (W / 2) * C;
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total)), dim3(MDIL_WG), 0,
                     (hipStream_t)stream, x, N, H, W, C, z, z_pitch, coff);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}

extern "C" int mdil_maxpool_concat_bwd(const float* x, const float* gz, int N, int H, int W, int C,
                                       int z_pitch, int coff, float* gx, void* stream) {
  MDIL_CHECK_ARG(x && gz && gx && (H % 2 == 0) && (W % 2 == 0), "maxpool_bwd: bad argument");
  const long long total = (long long)N * (H / 2) * (W / 2) * C;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for(total)), dim3(MDIL_WG), 0,
                     (hipStream_t)stream, x, gz, N, H, W, C, z_pitch, coff, gx);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}
