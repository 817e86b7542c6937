// BatchNorm2d (eps 1e-3, momentum 0.1) for NHWC fp32 tensors on gfx950: HBM-bound kernels.
//
// Statistics use Welford updates per thread and Chan merges up a fixed tree (thread -> LDS tree
// -> per-block partial -> finalize), so the result is run-to-run deterministic and free of the
// E[x^2]-E[x]^2 cancellation.  All tensor traffic is 16-byte vectors with the channel dimension
// fastest (a wave reads 1 KiB contiguous).
#include "common.h"
#include "bnfin.h"

namespace {

constexpr int BN_MAX_BLOCKS = MDIL_BN_MAX_BLOCKS;  // partial blocks (one per CU); finalize merges them
#ifndef MDIL_BN_T
#define MDIL_BN_T 512
#endif
constexpr int BN_T = MDIL_BN_T;     // threads per stats / reduce block (8 waves)
static_assert(BN_T >= 512, "bnfin.h needs J * C = 512 threads");

struct BnPlan {
  int nblk;
  int pix_per_block;
};

inline BnPlan bn_plan(long long npix, int C) {
  const int tpp = C / 4;
  const int ppi = BN_T / tpp;  // pixels per block iteration
  BnPlan p;
  long long iters = (npix + ppi - 1) / ppi;
  int nblk = (int)(iters < BN_MAX_BLOCKS ? iters : BN_MAX_BLOCKS);
  if (nblk < 1) nblk = 1;
  long long ipb = (iters + nblk - 1) / nblk;  // iterations per block
  p.pix_per_block = (int)(ipb * ppi);
  p.nblk = (int)((npix + p.pix_per_block - 1) / p.pix_per_block);
  return p;
}

// partial layout: [nblk][2][C] (mean, M2) then [nblk] counts
__global__ __launch_bounds__(BN_T) void bn_stats_kernel(const float* __restrict__ z, int npix,
                                                        int C, int pix_per_block,
                                                        float* __restrict__ partial,
                                                        float* __restrict__ pcount, const BnFinFwd fin) {
  MDIL_HBM_KERNEL_PRIO();

  __shared__ float s_mean[BN_T * 4];
  __shared__ float s_m2[BN_T * 4];
  __shared__ float s_n[BN_T];
  const int tid = threadIdx.x;
  const int tpp = C >> 2;
  const int ppi = BN_T / tpp;
  const int cq = tid % tpp, pl = tid / tpp;
  const int p_begin = blockIdx.x * pix_per_block;
  int p_end = p_begin + pix_per_block;
  if (p_end > npix) p_end = npix;

  float n = 0.f;
  f32x4 mean = {0.f, 0.f, 0.f, 0.f}, m2 = {0.f, 0.f, 0.f, 0.f};
  auto push = [&](const f32x4& x) {
    n += 1.f;
    const float rn = 1.f / n;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float d = x[k] - mean[k];
      mean[k] += d * rn;
      m2[k] += d * (x[k] - mean[k]);
    }
  };
  // U rows in flight per thread; each batch is summarised on its own (mean of the U values, then
  // squared deviations from it: two passes over registers, no long dependent chain and no
  // cancellation) and Chan-merged into the running summary -- the per-element Welford chain
  // with one row in flight left this kernel latency-bound at 20-35 % of the HBM rate.
  constexpr int U = 8;
  int p = p_begin + pl;
  for (; p + (U - 1) * ppi < p_end; p += U * ppi) {
    f32x4 x[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      x[u] = *reinterpret_cast<const f32x4*>(z + (long long)(p + u * ppi) * C + cq * 4);
    f32x4 bm = x[0];
#pragma unroll
    for (int u = 1; u < U; ++u) bm += x[u];
    bm *= (1.0f / U);
    f32x4 bq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < U; ++u) bq += (x[u] - bm) * (x[u] - bm);
    const float nn = n + (float)U;
    const float f = (float)U / nn;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float d = bm[k] - mean[k];
      mean[k] += d * f;
      m2[k] += bq[k] + d * d * n * f;
    }
    n = nn;
  }
  for (; p < p_end; p += ppi) push(*reinterpret_cast<const f32x4*>(z + (long long)p * C + cq * 4));
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    s_mean[tid * 4 + k] = mean[k];
    s_m2[tid * 4 + k] = m2[k];
  }
  s_n[tid] = n;
  __syncthreads();
  for (int s = BN_T / 2; s >= tpp; s >>= 1) {
    if (tid < s) {
      const float nb = s_n[tid + s];
      float na = s_n[tid];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float nn = na, mm = s_mean[tid * 4 + k], qq = s_m2[tid * 4 + k];
        welford_merge(nn, mm, qq, nb, s_mean[(tid + s) * 4 + k], s_m2[(tid + s) * 4 + k]);
        s_mean[tid * 4 + k] = mm;
        s_m2[tid * 4 + k] = qq;
        if (k == 3) s_n[tid] = nn;
      }
    }
    __syncthreads();
  }
  if (tid < tpp) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float* r0 = partial + ((long long)blockIdx.x * 2 + 0) * C + tid * 4 + k;
      float* r1 = partial + ((long long)blockIdx.x * 2 + 1) * C + tid * 4 + k;
      if (fin.ticket) {
        bnfin_st(r0, s_mean[tid * 4 + k]);
        bnfin_st(r1, s_m2[tid * 4 + k]);
      } else {
        *r0 = s_mean[tid * 4 + k];
        *r1 = s_m2[tid * 4 + k];
      }
    }
    if (tid == 0) {
      if (fin.ticket)
        bnfin_st(pcount + blockIdx.x, s_n[0]);
      else
        pcount[blockIdx.x] = s_n[0];
    }
  }
  if (fin.ticket) {      // the last block to arrive merges the rows into the coefficients (bnfin.h)
    if (bnfin_arrive(fin.ticket, gridDim.x, reinterpret_cast<int*>(s_n)))
      bnfin_forward(fin, partial, pcount, gridDim.x, C, s_mean);
  }
}

// Stand-alone finalize kernels: ONE work-group running the device functions of bnfin.h -- the very
// code a producer's last-arriving work-group runs when the finalize rides inside the producing
// launch (the shipped path), so fused and unfused forms are bit-identical.  (Round 3 ran one wave per
// channel here: 4.7 us alone, 10-12 us between the convs of the other streams, 156 launches per
// step; profiles/r03_experiments.txt #6.)
constexpr int FIN_T = 512;

__global__ __launch_bounds__(FIN_T) void bn_finalize_kernel(const float* partial, const float* pcount,
                                                            int nblk, int C, const BnFinFwd f) {
  MDIL_HBM_KERNEL_PRIO();
  __shared__ float scratch[3 * 8 * 64];          // 3 * J * C floats, J * C = 512
  bnfin_forward(f, partial, pcount, nblk, C, scratch);
}

__global__ void bn_eval_coeffs_kernel(int C, const float* __restrict__ gamma,
                                      const float* __restrict__ beta,
                                      const float* __restrict__ rm, const float* __restrict__ rv,
                                      float eps, float* scale, float* shift) {
  MDIL_HBM_KERNEL_PRIO();

  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const float invstd = 1.0f / sqrtf(rv[c] + eps);
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
  }
}

__global__ __launch_bounds__(MDIL_WG) void bn_apply_kernel(
    const float* __restrict__ z, long long nvec, int pix_per_image, int C,
    const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ drop, const float* __restrict__ res, int relu,
    float* __restrict__ y) {
  MDIL_HBM_KERNEL_PRIO();

  const int cq_n = C >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (long long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % cq_n);
    f32x4 v = *reinterpret_cast<const f32x4*>(z + i * 4);
    const f32x4 s = *reinterpret_cast<const f32x4*>(scale + cq * 4);
    const f32x4 t = *reinterpret_cast<const f32x4*>(shift + cq * 4);
    v = v * s + t;
    if (drop) {
      const long long n = (i / cq_n) / pix_per_image;
      v *= *reinterpret_cast<const f32x4*>(drop + n * C + cq * 4);
    }
    if (res) v += *reinterpret_cast<const f32x4*>(res + i * 4);
    if (relu) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
    }
    *reinterpret_cast<f32x4*>(y + i * 4) = v;
  }
}

__device__ __forceinline__ f32x4 bn_bwd_g(const float* __restrict__ gy,
                                          const float* __restrict__ relu_src,
                                          const float* __restrict__ drop, long long off,
                                          long long dropoff) {
  f32x4 g = *reinterpret_cast<const f32x4*>(gy + off);
  if (relu_src) {
    const f32x4 r = *reinterpret_cast<const f32x4*>(relu_src + off);
#pragma unroll
    for (int k = 0; k < 4; ++k) g[k] = r[k] > 0.f ? g[k] : 0.f;
  }
  if (drop) g *= *reinterpret_cast<const f32x4*>(drop + dropoff);
  return g;
}

// partial layout: [nblk][2][C] (sum g, sum g*xhat)
__global__ __launch_bounds__(BN_T) void bn_bwd_reduce_kernel(
    const float* __restrict__ gy, const float* __restrict__ relu_src,
    const float* __restrict__ drop, const float* __restrict__ z, int npix, int pix_per_image,
    int C, int pix_per_block, const float* __restrict__ save_mean,
    const float* __restrict__ save_invstd, float* __restrict__ partial, const BnFinBwd fin) {
  MDIL_HBM_KERNEL_PRIO();

  __shared__ float s_a[BN_T * 4];
  __shared__ float s_b[BN_T * 4];
  const int tid = threadIdx.x;
  const int tpp = C >> 2, ppi = BN_T / tpp;
  const int cq = tid % tpp, pl = tid / tpp;
  const int p_begin = blockIdx.x * pix_per_block;
  int p_end = p_begin + pix_per_block;
  if (p_end > npix) p_end = npix;
  const f32x4 mean = *reinterpret_cast<const f32x4*>(save_mean + cq * 4);
  const f32x4 istd = *reinterpret_cast<const f32x4*>(save_invstd + cq * 4);
  f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
  // U rows in flight per thread (3-4 loads each): one row at a time left the kernel latency-bound
  // (28 us for 75-150 MB).  The summation order per thread is unchanged.
  constexpr int U = 4;
  int p = p_begin + pl;
  for (; p + (U - 1) * ppi < p_end; p += U * ppi) {
    f32x4 g[U], x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int pp = p + u * ppi;
      const long long off = (long long)pp * C + cq * 4;
      g[u] = bn_bwd_g(gy, relu_src, drop, off, (long long)(pp / pix_per_image) * C + cq * 4);
      x[u] = *reinterpret_cast<const f32x4*>(z + off);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      sa += g[u];
      sb += g[u] * ((x[u] - mean) * istd);
    }
  }
  for (; p < p_end; p += ppi) {
    const long long off = (long long)p * C + cq * 4;
    const f32x4 g = bn_bwd_g(gy, relu_src, drop, off, (long long)(p / pix_per_image) * C + cq * 4);
    const f32x4 x = *reinterpret_cast<const f32x4*>(z + off);
    sa += g;
    sb += g * ((x - mean) * istd);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    s_a[tid * 4 + k] = sa[k];
    s_b[tid * 4 + k] = sb[k];
  }
  __syncthreads();
  for (int s = BN_T / 2; s >= tpp; s >>= 1) {
    if (tid < s) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        s_a[tid * 4 + k] += s_a[(tid + s) * 4 + k];
        s_b[tid * 4 + k] += s_b[(tid + s) * 4 + k];
      }
    }
    __syncthreads();
  }
  if (tid < tpp) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float* r0 = partial + ((long long)blockIdx.x * 2 + 0) * C + tid * 4 + k;
      float* r1 = partial + ((long long)blockIdx.x * 2 + 1) * C + tid * 4 + k;
      if (fin.ticket) {
        bnfin_st(r0, s_a[tid * 4 + k]);
        bnfin_st(r1, s_b[tid * 4 + k]);
      } else {
        *r0 = s_a[tid * 4 + k];
        *r1 = s_b[tid * 4 + k];
      }
    }
  }
  if (fin.ticket) {
    // (s_b: untouched by the scratch writes; s_a doubles as the 2 * J * C doubles of scratch = 8 KB)
    if (bnfin_arrive(fin.ticket, gridDim.x, reinterpret_cast<int*>(s_b)))
      bnfin_backward(fin, partial, gridDim.x, C, reinterpret_cast<double*>(s_a));
  }
}

// coef layout: [3][C] = gamma*invstd, sum(g)/n, sum(g*xhat)/n
__global__ __launch_bounds__(FIN_T) void bn_bwd_finalize_kernel(const float* partial, int nblk, int C,
                                                                const BnFinBwd f) {
  MDIL_HBM_KERNEL_PRIO();
  __shared__ double scratch[2 * 8 * 64];
  bnfin_backward(f, partial, nblk, C, scratch);
}

__global__ __launch_bounds__(MDIL_WG) void bn_bwd_apply_kernel(
    const float* __restrict__ gy, const float* __restrict__ relu_src,
    const float* __restrict__ drop, const float* __restrict__ z, long long nvec,
    int pix_per_image, int C, const float* __restrict__ save_mean,
    const float* __restrict__ save_invstd, const float* __restrict__ coef,
    float* __restrict__ gz) {
  MDIL_HBM_KERNEL_PRIO();

  const int cq_n = C >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (long long)gridDim.x * blockDim.x) {
    const int cq = (int)(i % cq_n);
    const long long p = i / cq_n;
    const f32x4 g = bn_bwd_g(gy, relu_src, drop, i * 4, (p / pix_per_image) * C + cq * 4);
    const f32x4 x = *reinterpret_cast<const f32x4*>(z + i * 4);
    const f32x4 mean = *reinterpret_cast<const f32x4*>(save_mean + cq * 4);
    const f32x4 istd = *reinterpret_cast<const f32x4*>(save_invstd + cq * 4);
    const f32x4 k0 = *reinterpret_cast<const f32x4*>(coef + cq * 4);
    const f32x4 k1 = *reinterpret_cast<const f32x4*>(coef + C + cq * 4);
    const f32x4 k2 = *reinterpret_cast<const f32x4*>(coef + 2 * C + cq * 4);
    const f32x4 xhat = (x - mean) * istd;
    *reinterpret_cast<f32x4*>(gz + i * 4) = k0 * (g - k1 - xhat * k2);
  }
}

inline int ew_grid(long long nvec) {
  long long b = (nvec + MDIL_WG - 1) / MDIL_WG;
  return (int)(b > 256 * 16 ? 256 * 16 : (b < 1 ? 1 : b));
}

inline bool bn_c_ok(int C) { return C == 16 || C == 64 || C == 128; }

}  // namespace

extern "C" size_t mdil_bn_workspace(long long npix, int C) {
  (void)npix;
  return ((size_t)BN_MAX_BLOCKS * 2 * C + BN_MAX_BLOCKS + 3 * (size_t)C) * sizeof(float);
}

// library-internal: stand-alone finalize launches on the shared device functions (bnfin.h)
int mdil_bn_finalize_fwd(const float* partial, const float* pcount, int nblk, int C, const BnFinFwd& f,
                         hipStream_t st) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(FIN_T), 0, st, partial, pcount, nblk, C, f);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}
int mdil_bn_finalize_bwd(const float* partial, int nblk, int C, const BnFinBwd& f, hipStream_t st) {
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(1), dim3(FIN_T), 0, st, partial, nblk, C, f);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}

static BnFinFwd fin_fwd(const float* gamma, const float* beta, float* running_mean, float* running_var,
                        long long* nbt, float eps, float momentum, float* save_mean, float* save_invstd,
                        float* scale, float* shift, unsigned* ticket) {
  BnFinFwd f;
  f.ticket = ticket;
  f.gamma = gamma, f.beta = beta;
  f.running_mean = running_mean, f.running_var = running_var, f.nbt = nbt;
  f.eps = eps, f.momentum = momentum;
  f.save_mean = save_mean, f.save_invstd = save_invstd, f.scale = scale, f.shift = shift;
  return f;
}

extern "C" int mdil_bn_train_stats(const float* z, long long npix, int C, const float* gamma,
                                   const float* beta, float* running_mean, float* running_var,
                                   long long* num_batches_tracked, float eps, float momentum,
                                   float* save_mean, float* save_invstd, float* scale,
                                   float* shift, void* workspace, size_t workspace_bytes,
                                   unsigned int* ticket, void* stream) {
  MDIL_CHECK_ARG(bn_c_ok(C), "bn: unsupported C=%d", C);
  MDIL_CHECK_ARG(z && gamma && beta && save_mean && save_invstd && scale && shift, "bn: null");
  MDIL_CHECK_ARG(npix > 0 && npix < (1ll << 31), "bn: npix=%lld", npix);
  MDIL_CHECK_ARG(workspace && workspace_bytes >= mdil_bn_workspace(npix, C), "bn: workspace");
  hipStream_t st = (hipStream_t)stream;
  const BnPlan p = bn_plan(npix, C);
  float* partial = (float*)workspace;
  float* pcount = partial + (size_t)BN_MAX_BLOCKS * 2 * C;
  const BnFinFwd f = fin_fwd(gamma, beta, running_mean, running_var, num_batches_tracked, eps, momentum,
                             save_mean, save_invstd, scale, shift, ticket);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(p.nblk), dim3(BN_T), 0, st, z, (int)npix, C,
                     p.pix_per_block, partial, pcount, f);
  MDIL_CHECK_LAUNCH();
  if (ticket) return MDIL_OK;           // finalized by the launch's last-arriving block
  return mdil_bn_finalize_fwd(partial, pcount, p.nblk, C, f, st);
}

extern "C" int mdil_bn_train_finalize(const float* partial, const float* pcount, int nblk, int C,
                                      const float* gamma, const float* beta, float* running_mean,
                                      float* running_var, long long* num_batches_tracked, float eps,
                                      float momentum, float* save_mean, float* save_invstd,
                                      float* scale, float* shift, void* stream) {
  MDIL_CHECK_ARG(bn_c_ok(C), "bn: unsupported C=%d", C);
  MDIL_CHECK_ARG(partial && pcount && nblk > 0, "bn_train_finalize: partials");
  MDIL_CHECK_ARG(gamma && beta && save_mean && save_invstd && scale && shift, "bn: null");
  return mdil_bn_finalize_fwd(partial, pcount, nblk, C,
                              fin_fwd(gamma, beta, running_mean, running_var, num_batches_tracked, eps,
                                      momentum, save_mean, save_invstd, scale, shift, nullptr),
                              (hipStream_t)stream);
}

extern "C" int mdil_bn_eval_coeffs(int C, const float* gamma, const float* beta,
                                   const float* running_mean, const float* running_var, float eps,
                                   float* scale, float* shift, void* stream) {
  MDIL_CHECK_ARG(gamma && beta && running_mean && running_var && scale && shift, "bn: null");
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(cdiv(C, 128)), dim3(128), 0, (hipStream_t)stream,
                     C, gamma, beta, running_mean, running_var, eps, scale, shift);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}

extern "C" int mdil_bn_apply(const float* z, long long npix, int pix_per_image, int C,
                             const float* scale, const float* shift, const float* drop,
                             const float* res, int relu, float* y, void* stream) {
  MDIL_CHECK_ARG(C % 4 == 0 && z && scale && shift && y, "bn_apply: bad argument");
  const long long nvec = npix * (C / 4);
  hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(nvec)), dim3(MDIL_WG), 0, (hipStream_t)stream, z,
                     nvec, pix_per_image, C, scale, shift, drop, res, relu, y);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}

// last pass of the BatchNorm backward alone: the reductions were produced AND finalized elsewhere
// (coef [3][C]: mdil_tapconv_bnred / mdil_tapconv_tail with a ticket); g is the already gated
// gradient, `drop` the Dropout2d factor the reductions were taken with
extern "C" int mdil_bn_backward_apply(const float* g, const float* drop, const float* z, long long npix,
                                      int pix_per_image, int C, const float* save_mean,
                                      const float* save_invstd, const float* coef, float* gz,
                                      void* stream) {
  MDIL_CHECK_ARG(bn_c_ok(C), "bn_backward_apply: unsupported C=%d", C);
  MDIL_CHECK_ARG(g && z && save_mean && save_invstd && coef && gz, "bn_backward_apply: null");
  const long long nvec = npix * (C / 4);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(nvec)), dim3(MDIL_WG), 0, (hipStream_t)stream, g,
                     (const float*)nullptr, drop, z, nvec, pix_per_image, C, save_mean, save_invstd,
                     coef, gz);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}

static BnFinBwd fin_bwd(const float* gamma, const float* save_invstd, float* dgamma, float* dbeta,
                        int accumulate, long long npix, float* coef, unsigned* ticket) {
  BnFinBwd f;
  f.ticket = ticket;
  f.gamma = gamma, f.save_invstd = save_invstd;
  f.dgamma = dgamma, f.dbeta = dbeta, f.accumulate = accumulate;
  f.n = (float)npix;
  f.coef = coef;
  return f;
}

// second half of mdil_bn_backward alone: the reductions were produced elsewhere
// (mdil_tapconv_bnred / mdil_tapconv_tail without a ticket); g is the already gated gradient (no
// relu_src here; `drop` = the Dropout2d factor the reductions were taken with)
extern "C" int mdil_bn_backward_partials(const float* g, const float* drop, const float* z,
                                         long long npix, int pix_per_image, int C, const float* gamma,
                                         const float* save_mean, const float* save_invstd,
                                         const float* partial, int nblk, float* dgamma, float* dbeta,
                                         int accumulate, float* gz, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  MDIL_CHECK_ARG(bn_c_ok(C), "bn_backward_partials: unsupported C=%d", C);
  MDIL_CHECK_ARG(g && z && gamma && save_mean && save_invstd && gz && partial && nblk > 0,
                 "bn_backward_partials: null");
  MDIL_CHECK_ARG(workspace && workspace_bytes >= 3 * (size_t)C * sizeof(float), "bn_backward_partials: ws");
  hipStream_t st = (hipStream_t)stream;
  float* coef = (float*)workspace;
  int rc = mdil_bn_finalize_bwd(partial, nblk, C,
                                fin_bwd(gamma, save_invstd, dgamma, dbeta, accumulate, npix, coef, nullptr), st);
  if (rc) return rc;
  return mdil_bn_backward_apply(g, drop, z, npix, pix_per_image, C, save_mean, save_invstd, coef, gz, stream);
}

extern "C" int mdil_bn_backward(const float* gy, const float* relu_src, const float* drop,
                                const float* z, long long npix, int pix_per_image, int C,
                                const float* gamma, const float* save_mean,
                                const float* save_invstd, float* dgamma, float* dbeta,
                                int accumulate, float* gz, void* workspace,
                                size_t workspace_bytes, unsigned int* ticket, void* stream) {
  MDIL_CHECK_ARG(bn_c_ok(C), "bn_backward: unsupported C=%d", C);
  MDIL_CHECK_ARG(gy && z && gamma && save_mean && save_invstd && gz, "bn_backward: null");
  MDIL_CHECK_ARG(npix > 0 && npix < (1ll << 31), "bn_backward: npix=%lld", npix);
  MDIL_CHECK_ARG(workspace && workspace_bytes >= mdil_bn_workspace(npix, C), "bn_backward: ws");
  hipStream_t st = (hipStream_t)stream;
  const BnPlan p = bn_plan(npix, C);
  float* partial = (float*)workspace;
  float* coef = partial + (size_t)BN_MAX_BLOCKS * 2 * C + BN_MAX_BLOCKS;
  const BnFinBwd f = fin_bwd(gamma, save_invstd, dgamma, dbeta, accumulate, npix, coef, ticket);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(p.nblk), dim3(BN_T), 0, st, gy, relu_src, drop,
                     z, (int)npix, pix_per_image, C, p.pix_per_block, save_mean, save_invstd,
                     partial, f);
  MDIL_CHECK_LAUNCH();
  if (!ticket) {
    const int rc = mdil_bn_finalize_bwd(partial, p.nblk, C, f, st);
    if (rc) return rc;
  }
  const long long nvec = npix * (C / 4);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(nvec)), dim3(MDIL_WG), 0, st, gy, relu_src,
                     drop, z, nvec, pix_per_image, C, save_mean, save_invstd, coef, gz);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}
