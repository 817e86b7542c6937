// Per-pixel losses on NHWC logits [npix][C] (fp32), one thread per pixel, one pass each:
//   - weighted cross-entropy (CrossEntropyLoss2d) with its gradient,
//   - the reference's KLDivLoss-on-probabilities distillation term with its gradient,
//   - argmax + confusion counts for mIoU.
// HBM-bound: logits are read once, gradients written once; reductions are block partials added
// in a fixed order by a 1-block finalize kernel (deterministic, no float atomics).
#include "common.h"

namespace {

constexpr int LOSS_MAX_BLOCKS = 2048;
constexpr int MAXC = 32;

__device__ __forceinline__ float block_sum(float v, float* sh) {
  // 256 threads -> one value in thread 0 (fixed order)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// A pixel's C logits live in a row of P >= C floats, P a multiple of 4 (P = 28 for the 27-class
// head: 16-byte vectors stay aligned; the pad channel is ignored on load and written as zero).
template <int C, int P>
__device__ __forceinline__ void load_row(const float* __restrict__ p, float (&x)[C]) {
  static_assert(P % 4 == 0 && P >= C && P - C < 4, "row pitch");
#pragma unroll
  for (int q = 0; q < P / 4; ++q) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p + q * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (q * 4 + k < C) x[q * 4 + k] = v[k];
  }
}

template <int C, int P>
__device__ __forceinline__ void store_row(float* __restrict__ p, const float (&x)[C]) {
#pragma unroll
  for (int q = 0; q < P / 4; ++q) {
    f32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (q * 4 + k < C) ? x[(q * 4 + k < C) ? q * 4 + k : 0] : 0.f;
    *reinterpret_cast<f32x4*>(p + q * 4) = v;
  }
}

__global__ __launch_bounds__(MDIL_WG) void ce_wsum_kernel(const long long* __restrict__ target,
                                                          const float* __restrict__ weight,
                                                          long long npix, int C,
                                                          float* __restrict__ part,
                                                          int* __restrict__ label_errors) {
  MDIL_HBM_KERNEL_PRIO();

  __shared__ float sh[4];
  float s = 0.f;
  int bad = 0;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix;
       p += (long long)gridDim.x * blockDim.x) {
    // a label outside [0, C) (an un-relabelled 255, another dataset's ids) would index past the
    // class-weight table: it is dropped like a zero-weight pixel and counted; the host raises
    // (torch's nll_loss raises a device assert here, train_new_task_step2.py:92)
    const long long y = target[p];
    const bool ok = y >= 0 && y < C;
    s += ok ? weight[ok ? y : 0] : 0.f;
    bad += ok ? 0 : 1;
  }
  if (bad && label_errors) atomicAdd(label_errors, bad);
  s = block_sum(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// out[0] = sum(part[0..n))   (double accumulation, fixed order)
__global__ void sum_partials_kernel(const float* __restrict__ part, int n, float* out) {
  MDIL_HBM_KERNEL_PRIO();

  __shared__ double sh[MDIL_WG];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += MDIL_WG) s += (double)part[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = MDIL_WG / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = (float)sh[0];
}

template <int C, int P>
__global__ __launch_bounds__(MDIL_WG) void ce_main_kernel(const float* __restrict__ logits,
                                                          const long long* __restrict__ target,
                                                          const float* __restrict__ weight,
                                                          long long npix,
                                                          const float* __restrict__ wsum,
                                                          const float* __restrict__ gscale,
                                                          float* __restrict__ part,
                                                          float* __restrict__ dlogits) {
  MDIL_HBM_KERNEL_PRIO();

  __shared__ float sh[4];
  const float inv_w = (gscale ? gscale[0] : 1.0f) / wsum[0];
  float acc = 0.f;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix;
       p += (long long)gridDim.x * blockDim.x) {
    float x[C];
    load_row<C, P>(logits + p * P, x);
    const long long yl = target[p];
    const bool yok = yl >= 0 && yl < C;
    const int y = yok ? (int)yl : 0;
    const float wy = yok ? weight[y] : 0.f;        // out-of-range label: dropped (counted above)
    float m = x[0];
#pragma unroll
    for (int k = 1; k < C; ++k) m = fmaxf(m, x[k]);
    float se = 0.f, xy = 0.f;
#pragma unroll
    for (int k = 0; k < C; ++k) {
      x[k] = x[k] - m;
      if (k == y) xy = x[k];
      x[k] = expf(x[k]);
      se += x[k];
    }
    const float lse = logf(se);
    acc += wy * (lse - xy);
    if (dlogits) {
      const float f = wy * inv_w, rs = 1.0f / se;
#pragma unroll
      for (int k = 0; k < C; ++k) x[k] = f * (x[k] * rs - (k == y ? 1.f : 0.f));
      store_row<C, P>(dlogits + p * P, x);
    }
  }
  acc = block_sum(acc, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

__global__ void ce_finalize_kernel(const float* __restrict__ part, int n,
                                   const float* __restrict__ wsum, float* loss) {
  MDIL_HBM_KERNEL_PRIO();

  __shared__ double sh[MDIL_WG];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += MDIL_WG) s += (double)part[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = MDIL_WG / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (float)(sh[0] / (double)wsum[0]);
}

template <int C, int P>
__global__ __launch_bounds__(MDIL_WG) void kld_main_kernel(const float* __restrict__ s_logits,
                                                           const float* __restrict__ t_logits,
                                                           long long npix, float inv_numel,
                                                           const float* __restrict__ gscale_ptr,
                                                           float* __restrict__ part,
                                                           float* __restrict__ ds) {
  MDIL_HBM_KERNEL_PRIO();

  __shared__ float sh[4];
  const float gscale = (gscale_ptr ? gscale_ptr[0] : 1.0f) * inv_numel;
  float acc = 0.f;
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix;
       p += (long long)gridDim.x * blockDim.x) {
    float s[C], t[C];
    load_row<C, P>(s_logits + p * P, s);
    load_row<C, P>(t_logits + p * P, t);
    float ms = s[0], mt = t[0];
#pragma unroll
    for (int k = 1; k < C; ++k) {
      ms = fmaxf(ms, s[k]);
      mt = fmaxf(mt, t[k]);
    }
    float ses = 0.f, set = 0.f;
    float lt[C];
#pragma unroll
    for (int k = 0; k < C; ++k) {
      s[k] = expf(s[k] - ms);
      ses += s[k];
      lt[k] = t[k] - mt;
      t[k] = expf(lt[k]);
      set += t[k];
    }
    const float rs = 1.0f / ses, rt = 1.0f / set, lset = logf(set);
    float term = 0.f, dot = 0.f;
#pragma unroll
    for (int k = 0; k < C; ++k) {
      s[k] *= rs;               // student probability
      t[k] *= rt;               // teacher probability
      term += t[k] * ((lt[k] - lset) - s[k]);
      dot += t[k] * s[k];
    }
    acc += term;
    if (ds) {
#pragma unroll
      for (int k = 0; k < C; ++k) s[k] = -gscale * s[k] * (t[k] - dot);
      store_row<C, P>(ds + p * P, s);
    }
  }
  acc = block_sum(acc, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

__global__ void kld_finalize_kernel(const float* __restrict__ part, int n, double inv_numel,
                                    float* loss) {
  MDIL_HBM_KERNEL_PRIO();

  __shared__ double sh[MDIL_WG];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += MDIL_WG) s += (double)part[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = MDIL_WG / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (float)(sh[0] * inv_numel);
}

template <int C, int P>
__global__ __launch_bounds__(MDIL_WG) void argmax_confusion_kernel(
    const float* __restrict__ logits, const long long* __restrict__ target, long long npix,
    int ignore, unsigned long long* __restrict__ counts, int* __restrict__ label_errors) {
  MDIL_HBM_KERNEL_PRIO();

  __shared__ unsigned int h[3 * MAXC];
  for (int i = threadIdx.x; i < 3 * MAXC; i += MDIL_WG) h[i] = 0;
  __syncthreads();
  for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix;
       p += (long long)gridDim.x * blockDim.x) {
    float x[C];
    load_row<C, P>(logits + p * P, x);
    int arg = 0;
    float m = x[0];
#pragma unroll
    for (int k = 1; k < C; ++k)
      if (x[k] > m) { m = x[k]; arg = k; }
    const long long yl = target[p];
    if (yl < 0 || yl >= C) {                        // iouEval's one-hot scatter_ would raise here
      if (label_errors) atomicAdd(label_errors, 1);
      continue;
    }
    const int y = (int)yl;
    if (y == ignore) continue;
    if (arg == y) {
      atomicAdd(&h[y], 1u);
    } else {
      if (arg != ignore) atomicAdd(&h[MAXC + arg], 1u);
      atomicAdd(&h[2 * MAXC + y], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * C; i += MDIL_WG) {
    const unsigned int v = h[(i / C) * MAXC + i % C];
    if (v) atomicAdd(&counts[i], (unsigned long long)v);
  }
}

inline int loss_grid(long long npix) {
  long long b = (npix + MDIL_WG - 1) / MDIL_WG;
  return (int)(b > LOSS_MAX_BLOCKS ? LOSS_MAX_BLOCKS : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" size_t mdil_loss_workspace(long long npix) {
  (void)npix;
  return (size_t)(LOSS_MAX_BLOCKS + 8) * sizeof(float);
}

extern "C" int mdil_ce_loss(const float* logits, const long long* target, const float* weight,
                            long long npix, int C, int pitch, const float* grad_scale,
                            float* loss, float* dlogits, int* label_errors, void* workspace,
                            size_t workspace_bytes, void* stream) {
  MDIL_CHECK_ARG(logits && target && weight && loss, "ce_loss: null argument");
  MDIL_CHECK_ARG(workspace && workspace_bytes >= mdil_loss_workspace(npix), "ce_loss: workspace");
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  float* wsum = part + LOSS_MAX_BLOCKS;
  const int grid = loss_grid(npix);
  hipLaunchKernelGGL(ce_wsum_kernel, dim3(grid), dim3(MDIL_WG), 0, st, target, weight, npix, C, part,
                     dlogits ? nullptr : label_errors);   // counted once: by the forward call
  MDIL_CHECK_LAUNCH();
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(MDIL_WG), 0, st, part, grid, wsum);
  MDIL_CHECK_LAUNCH();
  if (C == 20 && pitch == 20)
    hipLaunchKernelGGL((ce_main_kernel<20, 20>), dim3(grid), dim3(MDIL_WG), 0, st, logits, target,
                       weight, npix, wsum, grad_scale, part, dlogits);
  else if (C == 27 && pitch == 28)
    hipLaunchKernelGGL((ce_main_kernel<27, 28>), dim3(grid), dim3(MDIL_WG), 0, st, logits, target,
                       weight, npix, wsum, grad_scale, part, dlogits);
  else {
    mdil_set_error("ce_loss: unsupported C=%d pitch=%d", C, pitch);
    return MDIL_ERR_UNSUPPORTED;
  }
  MDIL_CHECK_LAUNCH();
  hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(MDIL_WG), 0, st, part, grid, wsum, loss);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}

extern "C" int mdil_kld_loss(const float* s_logits, const float* t_logits, long long npix, int C,
                             int pitch, const float* grad_scale, float* loss, float* ds, void* workspace,
                             size_t workspace_bytes, void* stream) {
  MDIL_CHECK_ARG(s_logits && t_logits && loss, "kld_loss: null argument");
  MDIL_CHECK_ARG(workspace && workspace_bytes >= mdil_loss_workspace(npix), "kld_loss: workspace");
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  const int grid = loss_grid(npix);
  const double inv_numel = 1.0 / ((double)npix * (double)C);
  if (C == 20 && pitch == 20)
    hipLaunchKernelGGL((kld_main_kernel<20, 20>), dim3(grid), dim3(MDIL_WG), 0, st, s_logits,
                       t_logits, npix, (float)inv_numel, grad_scale, part, ds);
  else if (C == 27 && pitch == 28)
    hipLaunchKernelGGL((kld_main_kernel<27, 28>), dim3(grid), dim3(MDIL_WG), 0, st, s_logits,
                       t_logits, npix, (float)inv_numel, grad_scale, part, ds);
  else {
    mdil_set_error("kld_loss: unsupported C=%d pitch=%d", C, pitch);
    return MDIL_ERR_UNSUPPORTED;
  }
  MDIL_CHECK_LAUNCH();
  hipLaunchKernelGGL(kld_finalize_kernel, dim3(1), dim3(MDIL_WG), 0, st, part, grid, inv_numel, loss);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}

extern "C" int mdil_argmax_confusion(const float* logits, const long long* target, long long npix,
                                     int C, int pitch, int ignore, long long* counts,
                                     int* label_errors, void* stream) {
  MDIL_CHECK_ARG(logits && target && counts && C <= MAXC, "argmax_confusion: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int grid = loss_grid(npix);
  if (C == 20 && pitch == 20)
    hipLaunchKernelGGL((argmax_confusion_kernel<20, 20>), dim3(grid), dim3(MDIL_WG), 0, st, logits,
                       target, npix, ignore, (unsigned long long*)counts, label_errors);
  else if (C == 27 && pitch == 28)
    hipLaunchKernelGGL((argmax_confusion_kernel<27, 28>), dim3(grid), dim3(MDIL_WG), 0, st, logits,
                       target, npix, ignore, (unsigned long long*)counts, label_errors);
  else {
    mdil_set_error("argmax_confusion: unsupported C=%d pitch=%d", C, pitch);
    return MDIL_ERR_UNSUPPORTED;
  }
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}
