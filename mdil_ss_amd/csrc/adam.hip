// Fused Adam (L2 weight decay added to the gradient -- torch.optim.Adam, not AdamW) over a flat
// fp32 segment: one launch per learning-rate group instead of 278 per-tensor updates.
#include <math.h>

#include "common.h"

namespace {

__global__ __launch_bounds__(MDIL_WG) void adam_kernel(float* __restrict__ p,
                                                       const float* __restrict__ g,
                                                       float* __restrict__ m,
                                                       float* __restrict__ v, long long n,
                                                       float step, float b1, float omb1, float b2,
                                                       float omb2, float eps, float wd,
                                                       float sqrt_bc2, float gscale) {
  MDIL_HBM_KERNEL_PRIO();

  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float pi = p[i];
    const float gi = g[i] * gscale + wd * pi;
    const float mi = b1 * m[i] + omb1 * gi;
    const float vi = b2 * v[i] + omb2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / sqrt_bc2 + eps;
    p[i] = pi - step * (mi / denom);
  }
}

}  // namespace

extern "C" int mdil_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                              long long n, double lr, double beta1, double beta2, double eps,
                              double weight_decay, double bias_correction1,
                              double bias_correction2, double grad_scale, void* stream) {
  MDIL_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n >= 0, "adam: bad argument");
  if (n == 0) return MDIL_OK;
  long long b = (n + MDIL_WG - 1) / MDIL_WG;
  if (b > 2048) b = 2048;
  hipLaunchKernelGGL(adam_kernel, dim3((int)b), dim3(MDIL_WG), 0, (hipStream_t)stream, param, grad,
                     exp_avg, exp_avg_sq, n, (float)(lr / bias_correction1), (float)beta1,
                     (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                     (float)weight_decay, (float)sqrt(bias_correction2), (float)grad_scale);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}
