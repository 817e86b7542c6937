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
// again, RECOMPUTES the logits, forms the logit gradient in registers and contracts it on the spot.
//
// Everything is v_mfma_f32_16x16x4_f32 on 16-pixel tiles, the head's weights live in registers in
// MFMA operand order (a first version did the 16 -> 4*nc map on the VALU with the weights
// broadcast from LDS: 640 ds_read_b128 per pixel made it LDS-issue bound, 560 us per backward):
//   * logits^T[class][pixel] = W^T x^T: A = weights (lane = (class-in-tile, ci group)), B = x as
//     loaded -- one 16-byte load per lane, lane = (pixel, 4 consecutive channels); the K order is
//     permuted (k-step s of lane group g <-> channel 4g + s) so that load IS the operand.  One
//     output pixel's classes (a group (a, b), padded to 32 rows = 2 tiles) end up spread over the 4
//     lane groups x 4 registers x 2 tiles of the pixel's column: softmax = in-lane work + two
//     cross-group shuffles.
//   * gx^T[ci][pixel] = W dl^T: the D registers of the logit gradient are, in the same permuted K
//     order, exactly the B operand -- no data movement.
//   * dW[ci][class] = sum_pixels x dl needs the pixels as K: the wave stages x and dl of its 16
//     pixels through LDS (10 KB) into operand order.  db[class] = in-lane sums, reduced at the end.
// Per-block partials are added in a fixed order by a small reduction kernel (deterministic, no
// float atomics).
#include <math.h>

#include "common.h"

namespace {

constexpr int HD_T = 256;            // 4 waves
constexpr int HD_WAVES = HD_T / 64;
constexpr int HD_MAX_BLOCKS = 1024;
constexpr int HD_XLD = 20;           // LDS row strides (floats) of the dW staging: 16-byte aligned rows
constexpr int HD_DLD = 132;
constexpr int HD_COLS = 4 * 32;      // dW partial: [16 ci][(a*2+b)*32 + c]
constexpr int HD_STAGE_W = 16 * (HD_XLD + HD_DLD);        // floats per wave
constexpr int HD_RED = HD_WAVES * 16 * HD_COLS;           // cross-wave reduction of the dW accumulators
constexpr int HD_LDS = HD_RED > HD_WAVES * HD_STAGE_W ? HD_RED : HD_WAVES * HD_STAGE_W;

// LDS accesses of one wave execute in program order; this only stops the COMPILER from moving
// them across the hand-over between the lanes that write a staging row and the lanes that read it
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// sum / max over the four lane groups of a pixel column (lanes j, j+16, j+32, j+48)
__device__ __forceinline__ float grp_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float grp_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

// Class placement.  A group's logits come out of the MFMAs as 2 tiles x 16 rows; row 4g + r of tile
// t sits in register r of lane group g ("slot" 4t + r of that group).  The NC classes are dealt to
// the four lane groups evenly -- group g holds classes CPG*g .. CPG*g + CPG-1 in slots 0 .. CPG-1
// (CPG = ceil(NC / 4): 5 for 20 classes, 7 for 27) -- so every lane does the same amount of
// softmax work and slots >= CPG are dead at compile time (dealing them tile by tile would leave
// three of four lane groups idle in the second tile: the softmax VALU work, not the MFMAs, bounds
// these kernels).  Which class a row holds is purely a matter of how the weights are arranged.
template <int NC>
struct HeadMap {
  static constexpr int CPG = (NC + 3) / 4;
  static_assert(CPG <= 8, "at most 32 classes");
  // class of (lane group g, slot) or -1
  __device__ static __forceinline__ int cls(int g, int slot) {
    const int c = CPG * g + slot;
    return (slot < CPG && c < NC) ? c : -1;
  }
  // class held by row `row` (0..15) of tile t
  __device__ static __forceinline__ int row_cls(int t, int row) { return cls(row >> 2, 4 * t + (row & 3)); }
};

// The head's parameters in registers.  lane = (j = lane & 15, g = lane >> 4).
//   wa[ab][t][s] = W[ci = 4g + s][class of row j of tile t][a][b]          A operand of the logits MFMAs
//   wg[ab][t][s] = W[ci = j][class of row 4g + s of tile t][a][b]          A operand of the gx MFMAs
//   bi[t][r]     = bias[class of row 4g + r of tile t]                      initial value of the accumulators
// (rows without a class read as 0; ab = 2a + b)
template <int NC, bool NEED_G>
struct HeadRegs {
  using M = HeadMap<NC>;
  float wa[4][2][4];
  float wg[NEED_G ? 4 : 1][2][4];
  f32x4 bi[2];

  __device__ __forceinline__ void load(const float* __restrict__ w, const float* __restrict__ bias, int lane) {
    const int j = lane & 15, g = lane >> 4;
#pragma unroll
    for (int ab = 0; ab < 4; ++ab)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int c1 = M::row_cls(t, j), ci1 = 4 * g + s;
          wa[ab][t][s] = c1 >= 0 ? w[(ci1 * NC + c1) * 4 + ab] : 0.f;
          if constexpr (NEED_G) {
            const int c2 = M::cls(g, 4 * t + s);
            wg[ab][t][s] = c2 >= 0 ? w[(j * NC + c2) * 4 + ab] : 0.f;
          }
        }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = M::cls(g, 4 * t + r);
        bi[t][r] = c >= 0 ? bias[c] : 0.f;
      }
  }

  // logits of output pixel group ab for the tile's 16 pixels: d[t][r], lane (g, j) = class cls(g, 4t + r)
  __device__ __forceinline__ void logits(int ab, const f32x4& xb, f32x4 (&d)[2]) const {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      d[t] = bi[t];
#pragma unroll
      for (int s = 0; s < 4; ++s) d[t] = mfma16(wa[ab][t][s], xb[s], d[t]);
    }
  }

  // gx^T += W[ab] dl^T, dl in the logits layout
  __device__ __forceinline__ void gx_acc(int ab, const f32x4 (&dl)[2], f32x4& gx) const {
    static_assert(NEED_G, "gx weights not loaded");
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s = 0; s < 4; ++s) gx = mfma16(wg[ab][t][s], dl[t][s], gx);
  }
};

// exp / log / 1/x on the transcendental unit (v_exp_f32, v_log_f32, v_rcp_f32: 1 ulp) -- the
// library versions cost 15-20 VALU instructions each and these kernels are VALU-bound
__device__ __forceinline__ float hd_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float hd_log(float x) { return __builtin_amdgcn_logf(x) * 0.693147180559945309f; }
__device__ __forceinline__ float hd_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// in-place: d <- exp(d - max) over the pixel's NC classes (dead slots -> 0); returns max, sum
template <int NC>
__device__ __forceinline__ void head_softmax(f32x4 (&d)[2], int g, float& m, float& se) {
  using M = HeadMap<NC>;
  float ml = -INFINITY;
#pragma unroll
  for (int sl = 0; sl < M::CPG; ++sl)
    if (M::CPG * 3 + sl < NC || M::cls(g, sl) >= 0) ml = fmaxf(ml, d[sl >> 2][sl & 3]);
  m = grp_max(ml);
  float sum = 0.f;
#pragma unroll
  for (int sl = 0; sl < 8; ++sl) {
    float e = 0.f;
    if (sl < M::CPG) {
      e = hd_exp(d[sl >> 2][sl & 3] - m);
      if (!(M::CPG * 3 + sl < NC)) e = M::cls(g, sl) >= 0 ? e : 0.f;     // only the last group can run out of classes
      sum += e;
    }
    d[sl >> 2][sl & 3] = e;
  }
  se = grp_sum(sum);
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

// tile walk shared by all kernels: a wave takes 16-pixel tiles widx, widx + nwaves, ...
struct TileCtx {
  long long q;       // the lane's input pixel (clamped)
  bool valid;
  int wi;
  long long r;       // n * H + h
};
__device__ __forceinline__ TileCtx tile_ctx(long long tile, int lane, long long npix, int W) {
  TileCtx c;
  const long long q = tile * 16 + (lane & 15);
  c.valid = q < npix;
  c.q = c.valid ? q : npix - 1;
  c.wi = (int)(c.q % W);
  c.r = c.q / W;
  return c;
}
__device__ __forceinline__ long long out_pixel(const TileCtx& c, int ab, int W) {
  return (2 * c.r + (ab >> 1)) * (2 * (long long)W) + 2 * c.wi + (ab & 1);
}

// ------------------------------------------------------------------------------------ forward
template <int NC, int P, bool STORE>
__global__ __launch_bounds__(HD_T) void head_ce_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    const long long* __restrict__ target, const float* __restrict__ cw, long long npix, int W,
    float* __restrict__ part, float* __restrict__ logits_out, int* __restrict__ label_errors) {
  MDIL_HBM_KERNEL_PRIO();
  using M = HeadMap<NC>;
  __shared__ float sh[4];
  __shared__ float cwl[32];               // class weights: a dependent global load per label otherwise
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  if (threadIdx.x < 32) cwl[threadIdx.x] = threadIdx.x < NC ? cw[threadIdx.x] : 0.f;
  HeadRegs<NC, false> R;
  R.load(w, bias, lane);
  __syncthreads();
  float accl = 0.f, accw = 0.f;
  int bad = 0;
  const long long ntiles = (npix + 15) / 16;
  for (long long tile = (long long)blockIdx.x * HD_WAVES + wave; tile < ntiles; tile += (long long)gridDim.x * HD_WAVES) {
    const TileCtx c = tile_ctx(tile, lane, npix, W);
    const f32x4 xb = *reinterpret_cast<const f32x4*>(x + c.q * 16 + 4 * g);
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) {
      const long long op = out_pixel(c, ab, W);
      f32x4 d[2];
      R.logits(ab, xb, d);
      if (STORE && c.valid) {
#pragma unroll
        for (int sl = 0; sl < M::CPG; ++sl) {
          const int cl = M::cls(g, sl);
          if (cl >= 0) logits_out[op * P + cl] = d[sl >> 2][sl & 3];
        }
        if (P > NC && g == 3) logits_out[op * P + NC] = 0.f;      // the pad entry reads as 0
      }
      const long long yl = target[op];
      const bool yok = yl >= 0 && yl < NC;      // out-of-range label: dropped and counted (loss.hip)
      const int y = yok ? (int)yl : -1;
      const float wy = (yok && c.valid) ? cwl[yok ? y : 0] : 0.f;
      bad += (c.valid && !yok && g == 0) ? 1 : 0;
      float ly = 0.f;
      const int ys = y - M::CPG * g;            // the slot of class y in this lane group (if any)
#pragma unroll
      for (int sl = 0; sl < M::CPG; ++sl) ly = (sl == ys) ? d[sl >> 2][sl & 3] : ly;
      float m, se;
      head_softmax<NC>(d, g, m, se);
      // -log_softmax[y] = log(se) + m - l_y: the lane that holds class y brings l_y, group 0 the rest
      accl += wy * ((g == 0 ? hd_log(se) + m : 0.f) - ly);
      accw += g == 0 ? wy : 0.f;
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

// student / teacher logits of one output pixel group in, probabilities out (s = student, t =
// teacher); returns the lane's share of sum_k t_k (log t_k - p_k) and dot = sum_k t_k p_k
template <int NC>
__device__ __forceinline__ float kld_terms(f32x4 (&s)[2], f32x4 (&t)[2], int g, float& dot) {
  float ms, ses, mt, set;
  const f32x4 lt[2] = {t[0], t[1]};
  head_softmax<NC>(s, g, ms, ses);
  head_softmax<NC>(t, g, mt, set);
  const float rs = hd_rcp(ses), rt = hd_rcp(set), lset = hd_log(set);
  float term = 0.f, dl_ = 0.f;
#pragma unroll
  for (int sl = 0; sl < HeadMap<NC>::CPG; ++sl) {
    const int tt = sl >> 2, r = sl & 3;
    const float ps = s[tt][r] * rs, pt = t[tt][r] * rt;     // dead slots: e = 0 -> p = 0 -> no contribution
    s[tt][r] = ps;
    t[tt][r] = pt;
    term += pt * (((lt[tt][r] - mt) - lset) - ps);
    dl_ += pt * ps;
  }
  dot = grp_sum(dl_);
  return term;
}

template <int NC>
__global__ __launch_bounds__(HD_T) void head_kld_fwd_kernel(
    const float* __restrict__ xs, const float* __restrict__ ws, const float* __restrict__ bs,
    const float* __restrict__ xt, const float* __restrict__ wt, const float* __restrict__ bt,
    long long npix, float* __restrict__ part) {
  MDIL_HBM_KERNEL_PRIO();
  __shared__ float sh[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  HeadRegs<NC, false> Rs, Rt;
  Rs.load(ws, bs, lane);
  Rt.load(wt, bt, lane);
  float acct = 0.f;
  const long long ntiles = (npix + 15) / 16;
  for (long long tile = (long long)blockIdx.x * HD_WAVES + wave; tile < ntiles; tile += (long long)gridDim.x * HD_WAVES) {
    const long long q0 = tile * 16 + (lane & 15);
    const bool valid = q0 < npix;
    const long long q = valid ? q0 : npix - 1;
    const f32x4 xsb = *reinterpret_cast<const f32x4*>(xs + q * 16 + 4 * g);
    const f32x4 xtb = *reinterpret_cast<const f32x4*>(xt + q * 16 + 4 * g);
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) {
      f32x4 s[2], t[2];
      Rs.logits(ab, xsb, s);
      Rt.logits(ab, xtb, t);
      float dot;
      const float term = kld_terms<NC>(s, t, g, dot);
      acct += valid ? term : 0.f;
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
// weight / bias gradient state of a wave
struct HeadWgrad {
  f32x4 dw[8];     // [T = 2ab + t]: lane (g, j) reg r = dW[ci = 4g + r][ab][class = 16t + j]
  f32x4 db[2];     // [t] reg r: sum over the lane's pixels and ab of dl[class 16t + 4g + r]
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < 8; ++i) dw[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    db[0] = db[1] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
};

// the tile's x and all four groups' dl are staged in LDS (rows = pixels) -> 32 MFMAs, K = pixels
__device__ __forceinline__ void head_wgrad_tile(HeadWgrad& G, const float* xs_w, const float* dls_w, int lane) {
  const int j = lane & 15, g = lane >> 4;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const float a = xs_w[(4 * s + g) * HD_XLD + j];          // A[i = ci = j][k]: pixel 4s + g
#pragma unroll
    for (int T = 0; T < 8; ++T) G.dw[T] = mfma16(a, dls_w[(4 * s + g) * HD_DLD + 16 * T + j], G.dw[T]);
  }
}

// per-block partials: wpart[blk][16 ci][(a*2+b)*32 + 16t + row] (dW, rows as the MFMAs hold them) and
// bpart[blk][8 * lane group + slot] (db); the reduction kernel maps rows / slots back to classes
__device__ __forceinline__ void head_emit_partials(HeadWgrad& G, float* red, float* __restrict__ wpart,
                                                   float* __restrict__ bpart) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  __syncthreads();
#pragma unroll
  for (int T = 0; T < 8; ++T)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(wave * 16 + 4 * g + r) * HD_COLS + 16 * T + j] = G.dw[T][r];
  __syncthreads();
  for (int i = threadIdx.x; i < 16 * HD_COLS; i += HD_T) {
    float s = red[i];
#pragma unroll
    for (int wv = 1; wv < HD_WAVES; ++wv) s += red[wv * 16 * HD_COLS + i];
    wpart[(long long)blockIdx.x * 16 * HD_COLS + i] = s;
  }
  __syncthreads();
  // db: sum over the 16 pixel lanes of each group (fixed butterfly), then the four waves in order
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = G.db[t][r];
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
      if (j == 0) red[wave * 32 + 8 * g + 4 * t + r] = v;          // (lane group, slot)
    }
  __syncthreads();
  if (threadIdx.x < 32)
    bpart[(long long)blockIdx.x * 32 + threadIdx.x] =
        (red[threadIdx.x] + red[32 + threadIdx.x]) + (red[64 + threadIdx.x] + red[96 + threadIdx.x]);
}

template <int NC, bool WGRAD>
__global__ __launch_bounds__(HD_T) void head_ce_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    const long long* __restrict__ target, const float* __restrict__ cw, long long npix, int W,
    const float* __restrict__ wsum, const float* __restrict__ gscale, float* __restrict__ gx,
    float* __restrict__ wpart, float* __restrict__ bpart) {
  MDIL_HBM_KERNEL_PRIO();
  __shared__ __attribute__((aligned(16))) float stage[WGRAD ? HD_LDS : 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
  float* xs_w = stage + wave * HD_STAGE_W;
  float* dls_w = xs_w + 16 * HD_XLD;
  __shared__ float cwl[32];
  if (threadIdx.x < 32) cwl[threadIdx.x] = threadIdx.x < NC ? cw[threadIdx.x] : 0.f;
  HeadRegs<NC, true> R;
  R.load(w, bias, lane);
  __syncthreads();
  const float inv_w = (gscale ? gscale[0] : 1.0f) / wsum[0];
  HeadWgrad G;
  G.zero();
  const long long ntiles = (npix + 15) / 16;
  for (long long tile = (long long)blockIdx.x * HD_WAVES + wave; tile < ntiles; tile += (long long)gridDim.x * HD_WAVES) {
    const TileCtx c = tile_ctx(tile, lane, npix, W);
    const f32x4 xb = *reinterpret_cast<const f32x4*>(x + c.q * 16 + 4 * g);
    if constexpr (WGRAD) {
      wave_sync();                              // the previous tile's operand reads are done
      *reinterpret_cast<f32x4*>(xs_w + j * HD_XLD + 4 * g) = c.valid ? xb : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 gxv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) {
      const long long op = out_pixel(c, ab, W);
      f32x4 d[2];
      R.logits(ab, xb, d);
      const long long yl = target[op];
      const bool yok = c.valid && yl >= 0 && yl < NC;
      const int y = yok ? (int)yl : -1;
      const float f = yok ? cwl[yok ? y : 0] * inv_w : 0.f;
      float m, se;
      head_softmax<NC>(d, g, m, se);
      const float frs = f * hd_rcp(se);
      const int ys = y - HeadMap<NC>::CPG * g;
#pragma unroll
      for (int sl = 0; sl < HeadMap<NC>::CPG; ++sl)
        d[sl >> 2][sl & 3] = d[sl >> 2][sl & 3] * frs - ((sl == ys) ? f : 0.f);    // (dead slots stay 0)
      R.gx_acc(ab, d, gxv);
      if constexpr (WGRAD) {
        G.db[0] += d[0];
        G.db[1] += d[1];
#pragma unroll
        for (int t = 0; t < 2; ++t) *reinterpret_cast<f32x4*>(dls_w + j * HD_DLD + 32 * ab + 16 * t + 4 * g) = d[t];
      }
    }
    if (c.valid) *reinterpret_cast<f32x4*>(gx + c.q * 16 + 4 * g) = gxv;
    if constexpr (WGRAD) {
      wave_sync();
      head_wgrad_tile(G, xs_w, dls_w, lane);
    }
  }
  if constexpr (WGRAD) head_emit_partials(G, stage, wpart, bpart);
}

template <int NC, bool WGRAD>
__global__ __launch_bounds__(HD_T) void head_kld_bwd_kernel(
    const float* __restrict__ xs, const float* __restrict__ ws, const float* __restrict__ bs,
    const float* __restrict__ xt, const float* __restrict__ wt, const float* __restrict__ bt,
    long long npix, float inv_numel, const float* __restrict__ gscale_ptr, float* __restrict__ gx,
    float* __restrict__ wpart, float* __restrict__ bpart) {
  MDIL_HBM_KERNEL_PRIO();
  __shared__ __attribute__((aligned(16))) float stage[WGRAD ? HD_LDS : 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
  float* xs_w = stage + wave * HD_STAGE_W;
  float* dls_w = xs_w + 16 * HD_XLD;
  HeadRegs<NC, true> Rs;
  HeadRegs<NC, false> Rt;
  Rs.load(ws, bs, lane);
  Rt.load(wt, bt, lane);
  const float gscale = (gscale_ptr ? gscale_ptr[0] : 1.0f) * inv_numel;
  HeadWgrad G;
  G.zero();
  const long long ntiles = (npix + 15) / 16;
  for (long long tile = (long long)blockIdx.x * HD_WAVES + wave; tile < ntiles; tile += (long long)gridDim.x * HD_WAVES) {
    const long long q0 = tile * 16 + j;
    const bool valid = q0 < npix;
    const long long q = valid ? q0 : npix - 1;
    const f32x4 xsb = *reinterpret_cast<const f32x4*>(xs + q * 16 + 4 * g);
    const f32x4 xtb = *reinterpret_cast<const f32x4*>(xt + q * 16 + 4 * g);
    if constexpr (WGRAD) {
      wave_sync();
      *reinterpret_cast<f32x4*>(xs_w + j * HD_XLD + 4 * g) = valid ? xsb : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float gs = valid ? gscale : 0.f;
    f32x4 gxv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) {
      f32x4 s[2], t[2];
      Rs.logits(ab, xsb, s);
      Rt.logits(ab, xtb, t);
      float dot;
      kld_terms<NC>(s, t, g, dot);
      // d/ds_k of  sum_j t_j (log t_j - p_j)  =  -p_k (t_k - sum_j t_j p_j)
#pragma unroll
      for (int sl = 0; sl < HeadMap<NC>::CPG; ++sl)
        s[sl >> 2][sl & 3] = -gs * s[sl >> 2][sl & 3] * (t[sl >> 2][sl & 3] - dot);
      Rs.gx_acc(ab, s, gxv);
      if constexpr (WGRAD) {
        G.db[0] += s[0];
        G.db[1] += s[1];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) *reinterpret_cast<f32x4*>(dls_w + j * HD_DLD + 32 * ab + 16 * tt + 4 * g) = s[tt];
      }
    }
    if (valid) *reinterpret_cast<f32x4*>(gx + q * 16 + 4 * g) = gxv;
    if constexpr (WGRAD) {
      wave_sync();
      head_wgrad_tile(G, xs_w, dls_w, lane);
    }
  }
  if constexpr (WGRAD) head_emit_partials(G, stage, wpart, bpart);
}

// dw[ci][c][a][b] (+)= sum_blk wpart[blk][ci][(a*2+b)*32 + c];  db[c] (+)= sum_blk bpart[blk][c]
// one work-group per 64 outputs: 4 slices of the block list per output, 16 loads in flight, fixed order
__global__ __launch_bounds__(HD_T) void head_wgrad_reduce_kernel(const float* __restrict__ wpart,
                                                                 const float* __restrict__ bpart, int nblk,
                                                                 int NC, float* dw, float* db, int accumulate) {
  __shared__ double sh[4][64];
  const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + o;
  const int n_w = 16 * 4 * NC;
  const float* src = nullptr;
  long long stride = 0;
  float* dst = nullptr;
  const int CPG = (NC + 3) / 4;                 // HeadMap: class c = (lane group c / CPG, slot c % CPG)
  if (i < n_w) {
    const int c = i % NC, ab = (i / NC) % 4, ci = i / (4 * NC);
    const int grp = c / CPG, slot = c % CPG;
    src = wpart + ci * HD_COLS + ab * 32 + 16 * (slot >> 2) + 4 * grp + (slot & 3);
    stride = 16 * HD_COLS;
    dst = dw + ((ci * NC + c) * 2 + (ab >> 1)) * 2 + (ab & 1);
  } else if (i < n_w + NC && db) {
    const int c = i - n_w;
    src = bpart + 8 * (c / CPG) + c % CPG;
    stride = 32;
    dst = db + c;
  }
  double s = 0.0;
  if (src) {
    for (int k0 = sl; k0 < nblk; k0 += 4 * 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int k = k0 + 4 * u;
        const float xv = src[(long long)(k < nblk ? k : nblk - 1) * stride];
        v[u] = k < nblk ? xv : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) s += (double)v[u];
    }
  }
  sh[sl][o] = s;
  __syncthreads();
  if (sl == 0 && dst) {
    const float r = (float)((sh[0][o] + sh[1][o]) + (sh[2][o] + sh[3][o]));
    *dst = accumulate ? *dst + r : r;
  }
}

inline int head_grid(long long npix) {
  long long b = ((npix + 15) / 16 + HD_WAVES - 1) / HD_WAVES;
  return (int)(b > HD_MAX_BLOCKS ? HD_MAX_BLOCKS : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" size_t mdil_head_workspace(void) {
  return ((size_t)2 * HD_MAX_BLOCKS + 8 + (size_t)HD_MAX_BLOCKS * (16 * HD_COLS + 32)) * sizeof(float);
}

#define HEAD_NC(NCv, Pv, ...)   \
  if (nc == NCv) {              \
    constexpr int NC = NCv;     \
    constexpr int P = Pv;       \
    (void)P;                    \
    __VA_ARGS__;                \
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
  float* part = (float*)workspace;
  float* wpart = part + 2 * HD_MAX_BLOCKS + 8;
  float* bpart = wpart + (size_t)HD_MAX_BLOCKS * 16 * HD_COLS;
  const int grid = head_grid(npix);
  if (gx == nullptr) {
    HEAD_NC(20, 20, if (logits_out) hipLaunchKernelGGL((head_ce_fwd_kernel<NC, P, true>), dim3(grid), dim3(HD_T), 0, st, x, w, bias, target, class_weight, npix, W, part, logits_out, label_errors);
                    else hipLaunchKernelGGL((head_ce_fwd_kernel<NC, P, false>), dim3(grid), dim3(HD_T), 0, st, x, w, bias, target, class_weight, npix, W, part, logits_out, label_errors))
    HEAD_NC(27, 28, if (logits_out) hipLaunchKernelGGL((head_ce_fwd_kernel<NC, P, true>), dim3(grid), dim3(HD_T), 0, st, x, w, bias, target, class_weight, npix, W, part, logits_out, label_errors);
                    else hipLaunchKernelGGL((head_ce_fwd_kernel<NC, P, false>), dim3(grid), dim3(HD_T), 0, st, x, w, bias, target, class_weight, npix, W, part, logits_out, label_errors)) {
      mdil_set_error("head_ce: unsupported nc=%d", nc);
      return MDIL_ERR_UNSUPPORTED;
    }
    MDIL_CHECK_LAUNCH();
    hipLaunchKernelGGL(head_ce_finalize_kernel, dim3(1), dim3(HD_T), 0, st, part, grid, loss, wsum);
    MDIL_CHECK_LAUNCH();
    return MDIL_OK;
  }
  HEAD_NC(20, 20, if (dw) hipLaunchKernelGGL((head_ce_bwd_kernel<NC, true>), dim3(grid), dim3(HD_T), 0, st, x, w, bias, target, class_weight, npix, W, wsum, grad_scale, gx, wpart, bpart);
                  else hipLaunchKernelGGL((head_ce_bwd_kernel<NC, false>), dim3(grid), dim3(HD_T), 0, st, x, w, bias, target, class_weight, npix, W, wsum, grad_scale, gx, wpart, bpart))
  HEAD_NC(27, 28, if (dw) hipLaunchKernelGGL((head_ce_bwd_kernel<NC, true>), dim3(grid), dim3(HD_T), 0, st, x, w, bias, target, class_weight, npix, W, wsum, grad_scale, gx, wpart, bpart);
                  else hipLaunchKernelGGL((head_ce_bwd_kernel<NC, false>), dim3(grid), dim3(HD_T), 0, st, x, w, bias, target, class_weight, npix, W, wsum, grad_scale, gx, wpart, bpart)) {
    mdil_set_error("head_ce: unsupported nc=%d", nc);
    return MDIL_ERR_UNSUPPORTED;
  }
  MDIL_CHECK_LAUNCH();
  if (dw) {
    hipLaunchKernelGGL(head_wgrad_reduce_kernel, dim3(cdiv(16 * 4 * nc + nc, 64)), dim3(HD_T), 0, st, wpart,
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
  const double inv_numel = 1.0 / ((double)npix * 4.0 * (double)nc);
  float* part = (float*)workspace;
  float* wpart = part + 2 * HD_MAX_BLOCKS + 8;
  float* bpart = wpart + (size_t)HD_MAX_BLOCKS * 16 * HD_COLS;
  const int grid = head_grid(npix);
  if (gx == nullptr) {
    HEAD_NC(20, 20, hipLaunchKernelGGL((head_kld_fwd_kernel<NC>), dim3(grid), dim3(HD_T), 0, st, xs, ws, bs, xt, wt, bt, npix, part))
    HEAD_NC(27, 28, hipLaunchKernelGGL((head_kld_fwd_kernel<NC>), dim3(grid), dim3(HD_T), 0, st, xs, ws, bs, xt, wt, bt, npix, part)) {
      mdil_set_error("head_kld: unsupported nc=%d", nc);
      return MDIL_ERR_UNSUPPORTED;
    }
    MDIL_CHECK_LAUNCH();
    hipLaunchKernelGGL(head_kld_finalize_kernel, dim3(1), dim3(HD_T), 0, st, part, grid, inv_numel, loss);
    MDIL_CHECK_LAUNCH();
    return MDIL_OK;
  }
  HEAD_NC(20, 20, if (dw) hipLaunchKernelGGL((head_kld_bwd_kernel<NC, true>), dim3(grid), dim3(HD_T), 0, st, xs, ws, bs, xt, wt, bt, npix, (float)inv_numel, grad_scale, gx, wpart, bpart);
                  else hipLaunchKernelGGL((head_kld_bwd_kernel<NC, false>), dim3(grid), dim3(HD_T), 0, st, xs, ws, bs, xt, wt, bt, npix, (float)inv_numel, grad_scale, gx, wpart, bpart))
  HEAD_NC(27, 28, if (dw) hipLaunchKernelGGL((head_kld_bwd_kernel<NC, true>), dim3(grid), dim3(HD_T), 0, st, xs, ws, bs, xt, wt, bt, npix, (float)inv_numel, grad_scale, gx, wpart, bpart);
                  else hipLaunchKernelGGL((head_kld_bwd_kernel<NC, false>), dim3(grid), dim3(HD_T), 0, st, xs, ws, bs, xt, wt, bt, npix, (float)inv_numel, grad_scale, gx, wpart, bpart)) {
    mdil_set_error("head_kld: unsupported nc=%d", nc);
    return MDIL_ERR_UNSUPPORTED;
  }
  MDIL_CHECK_LAUNCH();
  if (dw) {
    hipLaunchKernelGGL(head_wgrad_reduce_kernel, dim3(cdiv(16 * 4 * nc + nc, 64)), dim3(HD_T), 0, st, wpart,
                       bpart, grid, nc, dw, db, accumulate);
    MDIL_CHECK_LAUNCH();
  }
  return MDIL_OK;
}
