// Device half of the input pipeline (MyCoTransform, train_new_task_step2.py:48-81): the host
// decodes and resizes with PIL (as the reference does) and ships uint8 pixels; this kernel applies,
// per sample, the horizontal flip, the +-2 px translation with the reference's fill rules,
// ToTensor (uint8 -> float / 255) and ToLabel + Relabel(255 -> C-1), writing the NHWC fp32 image
// and the int64 label map the training step consumes.  Pure streaming: 4 B in, 20 B out per pixel.
#include "common.h"

namespace {

__global__ __launch_bounds__(MDIL_WG) void augment_kernel(
    const unsigned char* __restrict__ img, const unsigned char* __restrict__ lab,
    const int* __restrict__ params, int H, int W, int relabel_from, int relabel_to,
    float* __restrict__ out_img, long long* __restrict__ out_lab) {
  MDIL_HBM_KERNEL_PRIO();

  const int n = blockIdx.y;
  const int flip = params[3 * n], tx = params[3 * n + 1], ty = params[3 * n + 2];
  const long long hw = (long long)H * W;
  const unsigned char* im = img + (long long)n * hw * 3;
  const unsigned char* lb = lab + (long long)n * hw;
  float* oi = out_img + (long long)n * hw * 3;
  long long* ol = out_lab + (long long)n * hw;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < hw;
       p += (long long)gridDim.x * blockDim.x) {
    const int y = (int)(p / W), x = (int)(p - (long long)y * W);
    float r, g, b;
    int l;
    // ImageOps.expand(border=(tx, ty, 0, 0)) makes a (W+tx) x (H+ty) canvas; the following
    // crop((0, 0, W, H)) pads what lies beyond that canvas with 0 -- for the label too
    if ((tx < 0 && x >= W + tx) || (ty < 0 && y >= H + ty)) {
      r = g = b = 0.f;
      l = 0;
    } else if ((tx > 0 && x < tx) || (ty > 0 && y < ty)) {   // the expand border: fill 0 / 255
      r = g = b = 0.f;
      l = 255;
    } else {
      int sx = x - tx;
      const int sy = y - ty;
      if (flip) sx = W - 1 - sx;
      const long long s = (long long)sy * W + sx;
      r = (float)im[3 * s] / 255.f;                           // ToTensor: byte -> float, div(255)
      g = (float)im[3 * s + 1] / 255.f;
      b = (float)im[3 * s + 2] / 255.f;
      l = lb[s];
    }
    if (l == relabel_from) l = relabel_to;
    oi[3 * p] = r;
    oi[3 * p + 1] = g;
    oi[3 * p + 2] = b;
    ol[p] = l;
  }
}

// NCHW -> NHWC for a small channel count (the RGB input of the step: 3): one pixel per thread, the C
// plane reads are coalesced over the pixels, a wave writes 64 * C contiguous floats
__global__ __launch_bounds__(MDIL_WG) void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                               long long hw, int C) {
  MDIL_HBM_KERNEL_PRIO();
  const int n = blockIdx.y;
  const float* src = in + (long long)n * C * hw;
  float* dst = out + (long long)n * C * hw;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < hw; p += (long long)gridDim.x * blockDim.x) {
    if (C == 3) {
      const float r = src[p], g = src[hw + p], b = src[2 * hw + p];
      dst[3 * p] = r;
      dst[3 * p + 1] = g;
      dst[3 * p + 2] = b;
    } else {
      for (int c = 0; c < C; ++c) dst[p * C + c] = src[(long long)c * hw + p];
    }
  }
}

// Dropout2d factors from ONE uniform draw for all blocks: element e is kept (factor 1 / keep[e]) where
// u[e] < keep[e], else 0 (nn.Dropout2d: one Bernoulli(1 - p) per (image, channel), scaled by 1 / (1 - p))
__global__ __launch_bounds__(MDIL_WG) void dropout_factors_kernel(const float* __restrict__ u, const float* __restrict__ keep,
                                                                  const float* __restrict__ inv, float* __restrict__ out,
                                                                  int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = u[i] < keep[i] ? inv[i] : 0.f;
}

}  // namespace

extern "C" int mdil_nchw_to_nhwc(const float* in, int N, int C, int H, int W, float* out, void* stream) {
  MDIL_CHECK_ARG(in && out, "nchw_to_nhwc: null pointer");
  MDIL_CHECK_ARG(N > 0 && C > 0 && C <= 32 && H > 0 && W > 0, "nchw_to_nhwc: bad shape %d x %d x %d x %d", N, C, H, W);
  const long long hw = (long long)H * W;
  int bx = cdiv(hw, MDIL_WG * 4);
  if (bx > 2048) bx = 2048;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(bx, N), dim3(MDIL_WG), 0, (hipStream_t)stream, in, out, hw, C);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}

extern "C" int mdil_dropout_factors(const float* uniform, const float* keep, const float* inv_keep, float* out,
                                    int n, void* stream) {
  MDIL_CHECK_ARG(uniform && keep && inv_keep && out && n > 0, "dropout_factors: bad argument");
  hipLaunchKernelGGL(dropout_factors_kernel, dim3(cdiv(n, MDIL_WG)), dim3(MDIL_WG), 0, (hipStream_t)stream, uniform,
                     keep, inv_keep, out, n);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}

extern "C" int mdil_augment_batch(const unsigned char* img_u8, const unsigned char* lab_u8,
                                  const int* params, int N, int H, int W, int relabel_from,
                                  int relabel_to, float* out_img, long long* out_lab, void* stream) {
  MDIL_CHECK_ARG(img_u8 && lab_u8 && params && out_img && out_lab, "augment: null pointer");
  MDIL_CHECK_ARG(N > 0 && H > 0 && W > 0, "augment: bad shape %d x %d x %d", N, H, W);
  const long long hw = (long long)H * W;
  int bx = cdiv(hw, MDIL_WG * 4);
  if (bx > 1024) bx = 1024;
  hipLaunchKernelGGL(augment_kernel, dim3(bx, N), dim3(MDIL_WG), 0, (hipStream_t)stream, img_u8,
                     lab_u8, params, H, W, relabel_from, relabel_to, out_img, out_lab);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}
