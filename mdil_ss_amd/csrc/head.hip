// Decoder.output_conv fused with the per-pixel loss that consumes its logits (gfx950, NHWC fp32).
//
//   logits[n, 2h+a, 2w+b, c] = bias[c] + sum_ci x[n, h, w, ci] * W[ci][c][a][b]      (ConvTranspose2d(16, nc, 2, 2))
//   CE  = -sum_p w[y_p] log_softmax(logits_p)[y_p] / sum_p w[y_p]                    (CrossEntropyLoss2d)
//   KLD = mean over all elements of  t (log t - p_s),  t = softmax(teacher), p_s = softmax(student)
//
// The unfused path writes the logits (252 MB per forward at config 3), reads them back for the
// loss, again for the loss gradient, writes the logit gradient (252 MB) and reads it twice more for
// the transposed conv's dgrad and wgrad: ~1.7 GB of HBM traffic per student graph for a tensor
// that is a 16 -> 4*nc pointwise map of a 50 MB activation.  Here the logits are never
// materialised: the forward reads x (and the labels) and emits the loss; the backward reads x
// again, RECOMPUTES the logits (1,280 FMAs per input pixel), forms the logit gradient in
// registers and contracts it on the spot --
//   gx[ci]            = sum_{a,b,c} dl[a][b][c] W[ci][c][a][b]        VALU, weights broadcast from LDS
//   dW[ci][c][a][b]   = sum_pixels x[ci] dl[a][b][c]                  v_mfma_f32_16x16x4_f32, K = pixels:
//                       a wave stages its 64 pixels' x and dl through LDS into MFMA operand order
//   db[c]             = sum dl                                         per-thread sums, one reduction at the end
// -- per-block partials are added in a fixed order by a small reduction kernel (deterministic, no
// float atomics).  One thread owns one INPUT pixel (its four output pixels in turn).
#include <math.h>

#include "common.h"

namespace {

constexpr int HD_T = 256;            // 4 waves
constexpr int HD_WAVES = HD_T / 64;
constexpr int HD_MAX_BLOCKS = 1024;
constexpr int HD_XLD = 20;           // LDS row strides (floats): 16-byte aligned rows, b128 writes of
constexpr int HD_DLD = 36;           // 8 consecutive lanes hit 32 distinct banks
constexpr int HD_COLS = 4 * 32;      // dW partial: [16 ci][(a*2+b)*32 + c]
// LDS scratch of the backward kernels: per-wave MFMA operand staging (64 pixels x (x row + dl row)),
// reused by the final cross-wave reduction ([wave][16][HD_COLS])
constexpr int HD_STAGE = HD_WAVES * 64 * (HD_XLD + HD_DLD);
static_assert(HD_STAGE >= HD_WAVES * 16 * HD_COLS, "the reduction fits the staging buffer");

// LDS accesses of one wave execute in program order; this only stops the COMPILER from moving
// them across the hand-over between the lanes that write a staging row and the lanes that read it
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int P>
struct HeadLds {
  float W[2][2][16][P];   // [a][b][ci][c], pad columns 0
  float B[P];
};

template <int NC, int P>
__device__ __forceinline__ void head_load(HeadLds<P>& L, const float* __restrict__ w,
                                          const float* __restrict__ bias) {
  float* Wf = &L.W[0][0][0][0];
  for (int i = threadIdx.x; i < 2 * 2 * 16 * P; i += HD_T) {
    const int c = i % P, ci = (i / P) % 16, b = (i / (P * 16)) % 2, a = i / (P * 32);
    Wf[i] = c < NC ? w[((ci * NC + c) * 2 + a) * 2 + b] : 0.f;
  }
  for (int i = threadIdx.x; i < P; i += HD_T) L.B[i] = i < NC ? bias[i] : 0.f;
}

// LDS reads through an address the compiler cannot see through: `L` is a non-escaping shared
// variable, so a plain "memory" clobber does not stop hipcc from hoisting all 1,280 (loop-invariant)
// weight reads of the four output pixels out of the pixel loop -- and spilling them.  An opaque
// 32-bit LDS address per (a, b) keeps the reads where they are used.
typedef const f32x4 __attribute__((address_space(3))) * hd_lds_f4;
__device__ __forceinline__ f32x4 hd_lds(unsigned addr) { return *(hd_lds_f4)(__SIZE_TYPE__)addr; }
template <typename T>
__device__ __forceinline__ unsigned hd_opaque(const T* p) {
  unsigned a = (unsigned)(__SIZE_TYPE__)((const __attribute__((address_space(3))) T*)p);
  asm volatile("" : "+v"(a) : : "memory");
  return a;
}

// logits of one output pixel (a, b) of the input pixel whose 16 channels are xv; wab / bl: opaque
// LDS addresses of W[a][b][0][0] and B[0]
template <int P>
__device__ __forceinline__ void head_logits(const f32x4 (&xv)[4], unsigned wab, unsigned bl,
                                            f32x4 (&acc)[P / 4]) {
#pragma unroll
  for (int j = 0; j < P / 4; ++j) acc[j] = hd_lds(bl + j * 16);
#pragma unroll
  for (int ci = 0; ci < 16; ++ci) {
    // left alone, the scheduler issues all 16 x P/4 LDS reads before the first FMA (the FMAs wait
    // for x from global memory) and spills them: two weight rows in flight at a time
    if ((ci & 1) == 0) asm volatile("" ::: "memory");
    const float xs = xv[ci >> 2][ci & 3];
#pragma unroll
    for (int j = 0; j < P / 4; ++j) acc[j] += xs * hd_lds(wab + (ci * P + j * 4) * 4);
  }
}

// softmax pieces of NC logits held in acc: e[k] = exp(l_k - max), returns (max, sum e)
template <int NC, int P>
__device__ __forceinline__ void head_softmax(f32x4 (&acc)[P / 4], float& m, float& se) {
  m = acc[0][0];
#pragma unroll
  for (int k = 1; k < NC; ++k) m = fmaxf(m, acc[k >> 2][k & 3]);
  se = 0.f;
#pragma unroll
  for (int k = 0; k < P; ++k) {
    const float e = k < NC ? expf(acc[k >> 2][k & 3] - m) : 0.f;
    acc[k >> 2][k & 3] = e;
    se += e;
  }
}

__device__ __forceinline__ float hd_block_sum(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// ------------------------------------------------------------------------------------ forward
template <int NC, int P, bool STORE>
__global__ __launch_bounds__(HD_T) void head_ce_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    const long long* __restrict__ target, const float* __restrict__ cw, long long npix, int W,
    float* __restrict__ part, float* __restrict__ logits_out, int* __restrict__ label_errors) {
  MDIL_HBM_KERNEL_PRIO();
  __shared__ __attribute__((aligned(16))) HeadLds<P> L;
  __shared__ float sh[4];
  head_load<NC, P>(L, w, bias);
  __syncthreads();
  float accl = 0.f, accw = 0.f;
  int bad = 0;
#pragma unroll 1
  for (long long q = (long long)blockIdx.x * HD_T + threadIdx.x; q < npix; q += (long long)gridDim.x * HD_T) {
    asm volatile("" ::: "memory");          // keep the LDS weight reads inside the loop (no 1,280-register hoist)
    f32x4 xv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) xv[k] = *reinterpret_cast<const f32x4*>(x + q * 16 + k * 4);
    const int wi = (int)(q % W);
    const long long r = q / W;               // n * H + h
    // the four output pixels in turn (NOT unrolled: four copies of the 16 x P/4 weight reads in one
    // scheduling region is what the register allocator cannot hold)
#pragma unroll 1
    for (int ab = 0; ab < 4; ++ab) {
      const int a = ab >> 1, b = ab & 1;
      const long long op = (2 * r + a) * (2 * (long long)W) + 2 * wi + b;
      f32x4 acc[P / 4];
      head_logits<P>(xv, hd_opaque(&L.W[0][0][0][0]) + ab * 16 * P * 4, hd_opaque(&L.B[0]), acc);
      if (STORE) {
#pragma unroll
        for (int j = 0; j < P / 4; ++j) *reinterpret_cast<f32x4*>(logits_out + op * P + j * 4) = acc[j];
      }
      const long long yl = target[op];
      const bool yok = yl >= 0 && yl < NC;   // out-of-range label: dropped and counted (loss.hip)
      const int y = yok ? (int)yl : 0;
      const float wy = yok ? cw[y] : 0.f;
      bad += yok ? 0 : 1;
      float ly = 0.f;
#pragma unroll
      for (int k = 0; k < NC; ++k) ly = k == y ? acc[k >> 2][k & 3] : ly;
      float m, se;
      head_softmax<NC, P>(acc, m, se);
      accl += wy * (logf(se) - (ly - m));
      accw += wy;
    }
  }
  if (bad && label_errors) atomicAdd(label_errors, bad);
  accl = hd_block_sum(accl, sh);
  accw = hd_block_sum(accw, sh);
  if (threadIdx.x == 0) {
    part[blockIdx.x] = accl;
    part[HD_MAX_BLOCKS + blockIdx.x] = accw;
  }
}

// loss[0] = sum(part0) / sum(part1); wsum[0] = sum(part1)   (double accumulation, fixed order)
__global__ void head_ce_finalize_kernel(const float* __restrict__ part, int n, float* loss, float* wsum) {
  __shared__ double s0[HD_T], s1[HD_T];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < n; i += HD_T) {
    a += (double)part[i];
    b += (double)part[HD_MAX_BLOCKS + i];
  }
  s0[threadIdx.x] = a;
  s1[threadIdx.x] = b;
  __syncthreads();
  for (int o = HD_T / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      s0[threadIdx.x] += s0[threadIdx.x + o];
      s1[threadIdx.x] += s1[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    loss[0] = (float)(s0[0] / s1[0]);
    wsum[0] = (float)s1[0];
  }
}

template <int NC, int P>
__global__ __launch_bounds__(HD_T) void head_kld_fwd_kernel(
    const float* __restrict__ xs, const float* __restrict__ ws, const float* __restrict__ bs,
    const float* __restrict__ xt, const float* __restrict__ wt, const float* __restrict__ bt,
    long long npix, float* __restrict__ part) {
  MDIL_HBM_KERNEL_PRIO();
  __shared__ __attribute__((aligned(16))) HeadLds<P> Ls;
  __shared__ __attribute__((aligned(16))) HeadLds<P> Lt;
  __shared__ float sh[4];
  head_load<NC, P>(Ls, ws, bs);
  head_load<NC, P>(Lt, wt, bt);
  __syncthreads();
  float acct = 0.f;
#pragma unroll 1
  for (long long q = (long long)blockIdx.x * HD_T + threadIdx.x; q < npix; q += (long long)gridDim.x * HD_T) {
    asm volatile("" ::: "memory");
    f32x4 xsv[4], xtv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      xsv[k] = *reinterpret_cast<const f32x4*>(xs + q * 16 + k * 4);
      xtv[k] = *reinterpret_cast<const f32x4*>(xt + q * 16 + k * 4);
    }
#pragma unroll 1
    for (int ab = 0; ab < 4; ++ab) {
      f32x4 s[P / 4], t[P / 4];
      head_logits<P>(xsv, hd_opaque(&Ls.W[0][0][0][0]) + ab * 16 * P * 4, hd_opaque(&Ls.B[0]), s);
      head_logits<P>(xtv, hd_opaque(&Lt.W[0][0][0][0]) + ab * 16 * P * 4, hd_opaque(&Lt.B[0]), t);
      float mt = t[0][0];
#pragma unroll
      for (int k = 1; k < NC; ++k) mt = fmaxf(mt, t[k >> 2][k & 3]);
      float ms, ses;
      head_softmax<NC, P>(s, ms, ses);
      // teacher: log-probabilities are needed too
      float set = 0.f;
      float lt[NC];
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        lt[k] = t[k >> 2][k & 3] - mt;
        const float e = expf(lt[k]);
        t[k >> 2][k & 3] = e;
        set += e;
      }
      const float rs = 1.0f / ses, rt = 1.0f / set, lset = logf(set);
      float term = 0.f;
#pragma unroll
      for (int k = 0; k < NC; ++k) {
        const float ps = s[k >> 2][k & 3] * rs, pt = t[k >> 2][k & 3] * rt;
        term += pt * ((lt[k] - lset) - ps);
      }
      acct += term;
    }
  }
  acct = hd_block_sum(acct, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = acct;
}

__global__ void head_kld_finalize_kernel(const float* __restrict__ part, int n, double inv_numel, float* loss) {
  __shared__ double s0[HD_T];
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += HD_T) a += (double)part[i];
  s0[threadIdx.x] = a;
  __syncthreads();
  for (int o = HD_T / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) s0[threadIdx.x] += s0[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (float)(s0[0] * inv_numel);
}

// ----------------------------------------------------------------------------------- backward
// Shared tail of both backward kernels, per output pixel (a, b) of the thread's input pixel:
// dl (P/4 vectors, pad = 0) -> gx accumulation, bias-gradient sums, and (WGRAD) the weight-
// gradient MFMAs over the wave's 64 pixels.
template <int P, bool WGRAD>
__device__ __forceinline__ void head_contract(const f32x4 (&dl)[P / 4], unsigned wab,
                                              f32x4 (&gxv)[4], float (&dbacc)[P], float* xs_w,
                                              float* dls_w, f32x4 (&dacc)[2], int lane) {   // dacc: the CURRENT class's pair
#pragma unroll
  for (int k = 0; k < P; ++k) dbacc[k] += dl[k >> 2][k & 3];
  asm volatile("" ::: "memory");
#pragma unroll
  for (int ci = 0; ci < 16; ++ci) {
    if ((ci & 1) == 0) asm volatile("" ::: "memory");
    f32x4 s = dl[0] * hd_lds(wab + (ci * P) * 4);
#pragma unroll
    for (int j = 1; j < P / 4; ++j) s += dl[j] * hd_lds(wab + (ci * P + j * 4) * 4);
    gxv[ci >> 2][ci & 3] += (s[0] + s[1]) + (s[2] + s[3]);
  }
  if constexpr (WGRAD) {
    wave_sync();                               // the previous (a, b)'s MFMA operand reads are done
#pragma unroll
    for (int j = 0; j < P / 4; ++j) *reinterpret_cast<f32x4*>(dls_w + lane * HD_DLD + j * 4) = dl[j];
    wave_sync();
    const int i = lane & 15, kk = lane >> 4;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float av = xs_w[(4 * s + kk) * HD_XLD + i];
      const float b0 = dls_w[(4 * s + kk) * HD_DLD + i];
      const float b1 = dls_w[(4 * s + kk) * HD_DLD + 16 + i];
      dacc[0] = mfma16(av, b0, dacc[0]);
      dacc[1] = mfma16(av, b1, dacc[1]);
    }
  }
}

// class accumulators one place down: [0] <- [1] <- [2] <- [3] <- [0]
__device__ __forceinline__ void head_rotate(f32x4 (&dacc)[4][2]) {
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const f32x4 first = dacc[0][t];
    dacc[0][t] = dacc[1][t];
    dacc[1][t] = dacc[2][t];
    dacc[2][t] = dacc[3][t];
    dacc[3][t] = first;
  }
}

// per-block partials: wpart[blk][16][HD_COLS] (dW) and bpart[blk][32] (db)
template <int P, bool WGRAD>
__device__ __forceinline__ void head_emit_partials(f32x4 (&dacc)[4][2], float (&dbacc)[P], float* red,
                                                   float* __restrict__ wpart, float* __restrict__ bpart) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if constexpr (WGRAD) {
    // red: [HD_WAVES][16][HD_COLS] floats
    __syncthreads();
#pragma unroll
    for (int ab = 0; ab < 4; ++ab)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          red[(wave * 16 + 4 * (lane >> 4) + r) * HD_COLS + ab * 32 + t * 16 + (lane & 15)] = dacc[ab][t][r];
    __syncthreads();
    for (int i = threadIdx.x; i < 16 * HD_COLS; i += HD_T) {
      float s = red[i];
#pragma unroll
      for (int wv = 1; wv < HD_WAVES; ++wv) s += red[wv * 16 * HD_COLS + i];
      wpart[(long long)blockIdx.x * 16 * HD_COLS + i] = s;
    }
    __syncthreads();
    // db: wave sums (fixed butterfly), then the four waves in order
#pragma unroll
    for (int k = 0; k < P; ++k) {
      float v = dbacc[k];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
      if (lane == 0) red[wave * 32 + k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 32)
      bpart[(long long)blockIdx.x * 32 + threadIdx.x] =
          threadIdx.x < P ? (red[threadIdx.x] + red[32 + threadIdx.x]) + (red[64 + threadIdx.x] + red[96 + threadIdx.x])
                          : 0.f;
  }
}

template <int NC, int P, bool WGRAD>
__global__ __launch_bounds__(HD_T) void head_ce_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    const long long* __restrict__ target, const float* __restrict__ cw, long long npix, int W,
    const float* __restrict__ wsum, const float* __restrict__ gscale, float* __restrict__ gx,
    float* __restrict__ wpart, float* __restrict__ bpart) {
  MDIL_HBM_KERNEL_PRIO();
  __shared__ __attribute__((aligned(16))) HeadLds<P> L;
  __shared__ __attribute__((aligned(16))) float stage[WGRAD ? HD_STAGE : 4];
  head_load<NC, P>(L, w, bias);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // per-wave operand staging (inside `stage`, which the final reduction reuses): x rows, dl rows
  float* xs_w = stage + wave * (64 * HD_XLD + 64 * HD_DLD);
  float* dls_w = xs_w + 64 * HD_XLD;
  if constexpr (WGRAD) {
    for (int i = threadIdx.x; i < HD_STAGE; i += HD_T) stage[i] = 0.f;   // pad columns stay 0
  }
  __syncthreads();
  const float inv_w = (gscale ? gscale[0] : 1.0f) / wsum[0];
  f32x4 dacc[4][2];
  float dbacc[P];
#pragma unroll
  for (int ab = 0; ab < 4; ++ab) dacc[ab][0] = dacc[ab][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < P; ++k) dbacc[k] = 0.f;
  // wave-uniform trip count: a wave walks batches of 64 pixels
#pragma unroll 1
  for (long long base = ((long long)blockIdx.x * HD_WAVES + wave) * 64; base < npix;
       base += (long long)gridDim.x * HD_WAVES * 64) {
    asm volatile("" ::: "memory");
    const long long q = base + lane;
    const bool valid = q < npix;
    const long long qc = valid ? q : npix - 1;
    f32x4 xv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) xv[k] = *reinterpret_cast<const f32x4*>(x + qc * 16 + k * 4);
    if constexpr (WGRAD) {
      wave_sync();
#pragma unroll
      for (int k = 0; k < 4; ++k)
        *reinterpret_cast<f32x4*>(xs_w + lane * HD_XLD + k * 4) = valid ? xv[k] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int wi = (int)(qc % W);
    const long long r = qc / W;
    f32x4 gxv[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // the four output pixels in turn, not unrolled (see the forward).  The weight-gradient
    // accumulators of the four classes ROTATE through dacc[0] (the class being worked on), so the
    // loop body indexes registers statically; after four turns every set is back in its place.
#pragma unroll 1
    for (int ab = 0; ab < 4; ++ab) {
      const int a = ab >> 1, b = ab & 1;
      const long long op = (2 * r + a) * (2 * (long long)W) + 2 * wi + b;
      f32x4 acc[P / 4];
      const unsigned wab = hd_opaque(&L.W[0][0][0][0]) + ab * 16 * P * 4;
      head_logits<P>(xv, wab, hd_opaque(&L.B[0]), acc);
      const long long yl = target[op];
      const bool yok = valid && yl >= 0 && yl < NC;
      const int y = yok ? (int)yl : 0;
      const float f = yok ? cw[y] * inv_w : 0.f;
      float m, se;
      head_softmax<NC, P>(acc, m, se);
      const float rs = 1.0f / se;
#pragma unroll
      for (int k = 0; k < P; ++k)
        acc[k >> 2][k & 3] = k < NC ? f * (acc[k >> 2][k & 3] * rs - (k == y ? 1.f : 0.f)) : 0.f;
      head_contract<P, WGRAD>(acc, wab, gxv, dbacc, xs_w, dls_w, dacc[0], lane);
      head_rotate(dacc);
    }
    if (valid) {
#pragma unroll
      for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(gx + q * 16 + k * 4) = gxv[k];
    }
  }
  head_emit_partials<P, WGRAD>(dacc, dbacc, stage, wpart, bpart);
}

template <int NC, int P, bool WGRAD>
__global__ __launch_bounds__(HD_T) void head_kld_bwd_kernel(
    const float* __restrict__ xs, const float* __restrict__ ws, const float* __restrict__ bs,
    const float* __restrict__ xt, const float* __restrict__ wt, const float* __restrict__ bt,
    long long npix, float inv_numel, const float* __restrict__ gscale_ptr, float* __restrict__ gx,
    float* __restrict__ wpart, float* __restrict__ bpart) {
  MDIL_HBM_KERNEL_PRIO();
  __shared__ __attribute__((aligned(16))) HeadLds<P> Ls;
  __shared__ __attribute__((aligned(16))) HeadLds<P> Lt;
  __shared__ __attribute__((aligned(16))) float stage[WGRAD ? HD_STAGE : 4];
  head_load<NC, P>(Ls, ws, bs);
  head_load<NC, P>(Lt, wt, bt);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* xs_w = stage + wave * (64 * HD_XLD + 64 * HD_DLD);
  float* dls_w = xs_w + 64 * HD_XLD;
  if constexpr (WGRAD) {
    for (int i = threadIdx.x; i < HD_STAGE; i += HD_T) stage[i] = 0.f;
  }
  __syncthreads();
  const float gscale = (gscale_ptr ? gscale_ptr[0] : 1.0f) * inv_numel;
  f32x4 dacc[4][2];
  float dbacc[P];
#pragma unroll
  for (int ab = 0; ab < 4; ++ab) dacc[ab][0] = dacc[ab][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < P; ++k) dbacc[k] = 0.f;
#pragma unroll 1
  for (long long base = ((long long)blockIdx.x * HD_WAVES + wave) * 64; base < npix;
       base += (long long)gridDim.x * HD_WAVES * 64) {
    asm volatile("" ::: "memory");
    const long long q = base + lane;
    const bool valid = q < npix;
    const long long qc = valid ? q : npix - 1;
    f32x4 xsv[4], xtv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      xsv[k] = *reinterpret_cast<const f32x4*>(xs + qc * 16 + k * 4);
      xtv[k] = *reinterpret_cast<const f32x4*>(xt + qc * 16 + k * 4);
    }
    if constexpr (WGRAD) {
      wave_sync();
#pragma unroll
      for (int k = 0; k < 4; ++k)
        *reinterpret_cast<f32x4*>(xs_w + lane * HD_XLD + k * 4) = valid ? xsv[k] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 gxv[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll 1
    for (int ab = 0; ab < 4; ++ab) {
      f32x4 s[P / 4], t[P / 4];
      const unsigned wab = hd_opaque(&Ls.W[0][0][0][0]) + ab * 16 * P * 4;
      head_logits<P>(xsv, wab, hd_opaque(&Ls.B[0]), s);
      head_logits<P>(xtv, hd_opaque(&Lt.W[0][0][0][0]) + ab * 16 * P * 4, hd_opaque(&Lt.B[0]), t);
      float ms, ses, mt, set;
      head_softmax<NC, P>(s, ms, ses);
      head_softmax<NC, P>(t, mt, set);
      const float rs = 1.0f / ses, rt = 1.0f / set;
      float dot = 0.f;
#pragma unroll
      for (int k = 0; k < NC; ++k) dot += (t[k >> 2][k & 3] * rt) * (s[k >> 2][k & 3] * rs);
      const float g = valid ? gscale : 0.f;
      // d/ds_k of  sum_j t_j (log t_j - p_j)  =  -p_k (t_k - sum_j t_j p_j)
#pragma unroll
      for (int k = 0; k < P; ++k)
        s[k >> 2][k & 3] = k < NC ? -g * (s[k >> 2][k & 3] * rs) * (t[k >> 2][k & 3] * rt - dot) : 0.f;
      head_contract<P, WGRAD>(s, wab, gxv, dbacc, xs_w, dls_w, dacc[0], lane);
      head_rotate(dacc);
    }
    if (valid) {
#pragma unroll
      for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(gx + q * 16 + k * 4) = gxv[k];
    }
  }
  head_emit_partials<P, WGRAD>(dacc, dbacc, stage, wpart, bpart);
}

// dw[ci][c][a][b] (+)= sum_blk wpart[blk][ci][(a*2+b)*32 + c];  db[c] (+)= sum_blk bpart[blk][c]
__global__ __launch_bounds__(HD_T) void head_wgrad_reduce_kernel(const float* __restrict__ wpart,
                                                                 const float* __restrict__ bpart, int nblk,
                                                                 int NC, float* dw, float* db, int accumulate) {
  const int i = blockIdx.x * HD_T + threadIdx.x;
  const int n_w = 16 * 4 * NC;
  if (i < n_w) {
    const int c = i % NC, ab = (i / NC) % 4, ci = i / (4 * NC);
    double s = 0.0;
    for (int k = 0; k < nblk; ++k) s += (double)wpart[((long long)k * 16 + ci) * HD_COLS + ab * 32 + c];
    float* d = dw + ((ci * NC + c) * 2 + (ab >> 1)) * 2 + (ab & 1);
    *d = accumulate ? *d + (float)s : (float)s;
  } else if (i < n_w + NC && db) {
    const int c = i - n_w;
    double s = 0.0;
    for (int k = 0; k < nblk; ++k) s += (double)bpart[(long long)k * 32 + c];
    db[c] = accumulate ? db[c] + (float)s : (float)s;
  }
}

inline int head_grid(long long npix, int per_block) {
  long long b = (npix + per_block - 1) / per_block;
  return (int)(b > HD_MAX_BLOCKS ? HD_MAX_BLOCKS : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" size_t mdil_head_workspace(void) {
  return ((size_t)2 * HD_MAX_BLOCKS + 8 + (size_t)HD_MAX_BLOCKS * (16 * HD_COLS + 32)) * sizeof(float);
}

#define HEAD_DISPATCH(NCv, Pv, ...)                \
  if (nc == NCv && pitch == Pv) {                  \
    constexpr int NC = NCv, P = Pv;                \
    __VA_ARGS__;                                   \
  } else

extern "C" int mdil_head_ce(const float* x, const float* w, const float* bias, int N, int H, int W, int nc,
                            const long long* target, const float* class_weight, const float* grad_scale,
                            float* loss, float* wsum, float* gx, float* dw, float* db, int accumulate,
                            float* logits_out, int* label_errors, void* workspace, size_t workspace_bytes,
                            void* stream) {
  MDIL_CHECK_ARG(x && w && bias && target && class_weight && wsum && N > 0 && H > 0 && W > 0,
                 "head_ce: bad argument");
  MDIL_CHECK_ARG(workspace && workspace_bytes >= mdil_head_workspace(), "head_ce: workspace");
  MDIL_CHECK_ARG((gx == nullptr) == (grad_scale == nullptr) && (gx != nullptr || loss != nullptr),
                 "head_ce: forward (loss, no gx) or backward (grad_scale + gx)");
  MDIL_CHECK_ARG((dw == nullptr) == (db == nullptr), "head_ce: dw and db go together");
  hipStream_t st = (hipStream_t)stream;
  const long long npix = (long long)N * H * W;
  const int pitch = (nc + 3) / 4 * 4;
  float* part = (float*)workspace;
  float* wpart = part + 2 * HD_MAX_BLOCKS + 8;
  float* bpart = wpart + (size_t)HD_MAX_BLOCKS * 16 * HD_COLS;
  if (gx == nullptr) {
    const int grid = head_grid(npix, HD_T);
    HEAD_DISPATCH(20, 20, if (logits_out) hipLaunchKernelGGL((head_ce_fwd_kernel<NC, P, true>), dim3(grid), dim3(HD_T), 0, st, x, w, bias, target, class_weight, npix, W, part, logits_out, label_errors);
                          else hipLaunchKernelGGL((head_ce_fwd_kernel<NC, P, false>), dim3(grid), dim3(HD_T), 0, st, x, w, bias, target, class_weight, npix, W, part, logits_out, label_errors))
    HEAD_DISPATCH(27, 28, if (logits_out) hipLaunchKernelGGL((head_ce_fwd_kernel<NC, P, true>), dim3(grid), dim3(HD_T), 0, st, x, w, bias, target, class_weight, npix, W, part, logits_out, label_errors);
                          else hipLaunchKernelGGL((head_ce_fwd_kernel<NC, P, false>), dim3(grid), dim3(HD_T), 0, st, x, w, bias, target, class_weight, npix, W, part, logits_out, label_errors)) {
      mdil_set_error("head_ce: unsupported nc=%d", nc);
      return MDIL_ERR_UNSUPPORTED;
    }
    MDIL_CHECK_LAUNCH();
    hipLaunchKernelGGL(head_ce_finalize_kernel, dim3(1), dim3(HD_T), 0, st, part, grid, loss, wsum);
    MDIL_CHECK_LAUNCH();
    return MDIL_OK;
  }
  const int grid = head_grid(npix, 64 * HD_WAVES);
  HEAD_DISPATCH(20, 20, if (dw) hipLaunchKernelGGL((head_ce_bwd_kernel<NC, P, true>), dim3(grid), dim3(HD_T), 0, st, x, w, bias, target, class_weight, npix, W, wsum, grad_scale, gx, wpart, bpart);
                        else hipLaunchKernelGGL((head_ce_bwd_kernel<NC, P, false>), dim3(grid), dim3(HD_T), 0, st, x, w, bias, target, class_weight, npix, W, wsum, grad_scale, gx, wpart, bpart))
  HEAD_DISPATCH(27, 28, if (dw) hipLaunchKernelGGL((head_ce_bwd_kernel<NC, P, true>), dim3(grid), dim3(HD_T), 0, st, x, w, bias, target, class_weight, npix, W, wsum, grad_scale, gx, wpart, bpart);
                        else hipLaunchKernelGGL((head_ce_bwd_kernel<NC, P, false>), dim3(grid), dim3(HD_T), 0, st, x, w, bias, target, class_weight, npix, W, wsum, grad_scale, gx, wpart, bpart)) {
    mdil_set_error("head_ce: unsupported nc=%d", nc);
    return MDIL_ERR_UNSUPPORTED;
  }
  MDIL_CHECK_LAUNCH();
  if (dw) {
    hipLaunchKernelGGL(head_wgrad_reduce_kernel, dim3(cdiv(16 * 4 * nc + nc, HD_T)), dim3(HD_T), 0, st, wpart,
                       bpart, grid, nc, dw, db, accumulate);
    MDIL_CHECK_LAUNCH();
  }
  return MDIL_OK;
}

extern "C" int mdil_head_kld(const float* xs, const float* ws, const float* bs, const float* xt,
                             const float* wt, const float* bt, int N, int H, int W, int nc,
                             const float* grad_scale, float* loss, float* gx, float* dw, float* db,
                             int accumulate, void* workspace, size_t workspace_bytes, void* stream) {
  MDIL_CHECK_ARG(xs && ws && bs && xt && wt && bt && N > 0 && H > 0 && W > 0, "head_kld: bad argument");
  MDIL_CHECK_ARG(workspace && workspace_bytes >= mdil_head_workspace(), "head_kld: workspace");
  MDIL_CHECK_ARG((gx == nullptr) == (grad_scale == nullptr) && (gx != nullptr || loss != nullptr),
                 "head_kld: forward (loss, no gx) or backward (grad_scale + gx)");
  MDIL_CHECK_ARG((dw == nullptr) == (db == nullptr), "head_kld: dw and db go together");
  hipStream_t st = (hipStream_t)stream;
  const long long npix = (long long)N * H * W;
  const int pitch = (nc + 3) / 4 * 4;
  const double inv_numel = 1.0 / ((double)npix * 4.0 * (double)nc);
  float* part = (float*)workspace;
  float* wpart = part + 2 * HD_MAX_BLOCKS + 8;
  float* bpart = wpart + (size_t)HD_MAX_BLOCKS * 16 * HD_COLS;
  if (gx == nullptr) {
    const int grid = head_grid(npix, HD_T);
    HEAD_DISPATCH(20, 20, hipLaunchKernelGGL((head_kld_fwd_kernel<NC, P>), dim3(grid), dim3(HD_T), 0, st, xs, ws, bs, xt, wt, bt, npix, part))
    HEAD_DISPATCH(27, 28, hipLaunchKernelGGL((head_kld_fwd_kernel<NC, P>), dim3(grid), dim3(HD_T), 0, st, xs, ws, bs, xt, wt, bt, npix, part)) {
      mdil_set_error("head_kld: unsupported nc=%d", nc);
      return MDIL_ERR_UNSUPPORTED;
    }
    MDIL_CHECK_LAUNCH();
    hipLaunchKernelGGL(head_kld_finalize_kernel, dim3(1), dim3(HD_T), 0, st, part, grid, inv_numel, loss);
    MDIL_CHECK_LAUNCH();
    return MDIL_OK;
  }
  const int grid = head_grid(npix, 64 * HD_WAVES);
  HEAD_DISPATCH(20, 20, if (dw) hipLaunchKernelGGL((head_kld_bwd_kernel<NC, P, true>), dim3(grid), dim3(HD_T), 0, st, xs, ws, bs, xt, wt, bt, npix, (float)inv_numel, grad_scale, gx, wpart, bpart);
                        else hipLaunchKernelGGL((head_kld_bwd_kernel<NC, P, false>), dim3(grid), dim3(HD_T), 0, st, xs, ws, bs, xt, wt, bt, npix, (float)inv_numel, grad_scale, gx, wpart, bpart))
  HEAD_DISPATCH(27, 28, if (dw) hipLaunchKernelGGL((head_kld_bwd_kernel<NC, P, true>), dim3(grid), dim3(HD_T), 0, st, xs, ws, bs, xt, wt, bt, npix, (float)inv_numel, grad_scale, gx, wpart, bpart);
                        else hipLaunchKernelGGL((head_kld_bwd_kernel<NC, P, false>), dim3(grid), dim3(HD_T), 0, st, xs, ws, bs, xt, wt, bt, npix, (float)inv_numel, grad_scale, gx, wpart, bpart)) {
    mdil_set_error("head_kld: unsupported nc=%d", nc);
    return MDIL_ERR_UNSUPPORTED;
  }
  MDIL_CHECK_LAUNCH();
  if (dw) {
    hipLaunchKernelGGL(head_wgrad_reduce_kernel, dim3(cdiv(16 * 4 * nc + nc, HD_T)), dim3(HD_T), 0, st, wpart,
                       bpart, grid, nc, dw, db, accumulate);
    MDIL_CHECK_LAUNCH();
  }
  return MDIL_OK;
}
