// Fused Adam (L2 weight decay added to the gradient -- torch.optim.Adam, not AdamW) over a flat
// fp32 segment: one launch per learning-rate group instead of 278 per-tensor updates.
#include "common.h"

namespace {

__global__ __launch_bounds__(MDIL_WG) void adam_kernel(float* __restrict__ p,
                                                       const float* __restrict__ g,
                                                       float* __restrict__ m,
                                                       float* __restrict__ v, long long n, float lr,
                                                       float b1, float b2, float eps, float wd,
                                                       float bc1, float sqrt_bc2, float gscale) {
  const float step = lr / bc1;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float pi = p[i];
    const float gi = g[i] * gscale + wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / sqrt_bc2 + eps;
    p[i] = pi - step * (mi / denom);
  }
}

}  // namespace

extern "C" int mdil_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                              long long n, float lr, float beta1, float beta2, float eps,
                              float weight_decay, float bias_correction1, float bias_correction2,
                              float grad_scale, void* stream) {
  MDIL_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n >= 0, "adam: bad argument");
  if (n == 0) return MDIL_OK;
  long long b = (n + MDIL_WG - 1) / MDIL_WG;
  if (b > 2048) b = 2048;
  hipLaunchKernelGGL(adam_kernel, dim3((int)b), dim3(MDIL_WG), 0, (hipStream_t)stream, param, grad,
                     exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, bias_correction1,
                     sqrtf(bias_correction2), grad_scale);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}
