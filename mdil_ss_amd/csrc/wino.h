// Helpers shared by the Winograd streaming convolutions (wconv.hip: F(2,3), w4conv.hip: F(4,3)):
// LDS / buffer loads in MFMA operand order, DPP lane exchanges, the row-level reduce-scatter and
// Welford / Chan merge of the fused BatchNorm statistics / BatchNorm-backward reductions.
#pragma once
#include "common.h"
#include "bnfin.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

typedef const f32x4 __attribute__((address_space(3))) * wlds_f4_ptr;
__device__ __forceinline__ f32x4 wlds_ld(unsigned addr) { return *(wlds_f4_ptr)(__SIZE_TYPE__)addr; }
constexpr unsigned WC_WIN = 61440;

__device__ __forceinline__ f32x4 wbuf_load(const __amdgpu_buffer_rsrc_t r, unsigned voff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
  return __builtin_bit_cast(f32x4, v);
}

// value of the lane the DPP control CTRL names (row_ror:n = 0x120 + n, row_half_mirror = 0x141,
// quad_perm = its 8-bit pattern)
template <int CTRL>
__device__ __forceinline__ float wc_ror(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf,
                                                               0xf, false));
}

// Reduce-scatter over the 16 lanes of a row (which hold the same TM * 4 channels of 16 different
// pixel pairs): four exchange steps (row_ror:8, row_half_mirror, quad_perm xor 2 / xor 1), each
// halving the channels a lane is still responsible for.  Returns, in lane li, the sum over the row
// of channel j = 4m + k = li (TM = 4); for TM = 2 the first step is skipped: j = li & 7 and the two
// half rows hold the sums over their own 8 lanes.  15 add + 30 select instead of the 64 shuffles of
// an all-reduce, and the caller keeps ONE running register per quantity.
template <int TM>
__device__ __forceinline__ float wc_reduce_scatter(const f32x4 (&v)[TM], int li) {
  float w8[8];
  if constexpr (TM == 4) {
    const bool hi = (li & 8) != 0;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float lo_v = v[t >> 2][t & 3], hi_v = v[2 + (t >> 2)][t & 3];
      w8[t] = (hi ? hi_v : lo_v) + wc_ror<0x128>(hi ? lo_v : hi_v);
    }
  } else {
#pragma unroll
    for (int t = 0; t < 8; ++t) w8[t] = v[t >> 2][t & 3];
  }
  float w4[4], w2[2];
  {
    const bool hi = (li & 4) != 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) w4[t] = (hi ? w8[4 + t] : w8[t]) + wc_ror<0x141>(hi ? w8[t] : w8[4 + t]);
  }
  {
    const bool hi = (li & 2) != 0;
#pragma unroll
    for (int t = 0; t < 2; ++t) w2[t] = (hi ? w4[2 + t] : w4[t]) + wc_ror<0x4E>(hi ? w4[t] : w4[2 + t]);
  }
  const bool hi = (li & 1) != 0;
  return (hi ? w2[1] : w2[0]) + wc_ror<0xB1>(hi ? w2[0] : w2[1]);
}

// one level of the 16-lane Welford / Chan merge: every lane combines its (n, mean, M2) per channel
// with those of the lane CTRL points at (the count is shared by all of a lane's channels)
template <int CTRL, int TM>
__device__ __forceinline__ void wc_stat_level(float& n, f32x4 (&mean)[TM], f32x4 (&m2)[TM]) {
  const float on = wc_ror<CTRL>(n);
  const float nn = n + on;
  const float f = nn > 0.f ? on * __builtin_amdgcn_rcpf(nn) : 0.f;
  const float nf = n * f;
#pragma unroll
  for (int m = 0; m < TM; ++m)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float om = wc_ror<CTRL>(mean[m][k]), oq = wc_ror<CTRL>(m2[m][k]);
      const float d = om - mean[m][k];
      mean[m][k] += d * f;
      m2[m][k] += oq + d * d * nf;
    }
  n = nn;
}

struct wconv_args {
  const float* in0;
  const float* in1;
  const float* wpk;      // [tap][C][C] fp32 in the geometry's tap order
  float* out;
  mdil_epilogue e;
  int N, H, W;
  int axis;              // 0: taps along H, 1: along W
  int delta;             // dilation
  int tap[3];            // geometry tap index of the offsets -delta, 0, +delta
  int tap_ad, src_ad;    // adapter tap (wpk index) and its source tensor
  int src3;              // source tensor of the three conv taps
  float* stats;
  float* stats_count;
  const float* bn_z;
  const float* bn_mean;
  const float* bn_invstd;
  const float* t_gate;   // MODE 3 (tail): the stored value is gated by t_gate > 0 ...
  const float* t_drop;   // ... and the reductions are taken of (stored value) * t_drop[image][channel]
  int sh_delta, sh_nb, sh_W, sh_H;   // log2 of delta / pair blocks per axis / W / H when ALL are powers of two, else -1
  BnFinFwd ff;           // MODE 1: finalize by the last-arriving work-group (ticket != nullptr)
  BnFinBwd fb;           // MODE 2 / 3: the same for the BatchNorm-backward reductions
};

inline int wc_num_cu() {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    if (n > MDIL_BN_MAX_BLOCKS) n = MDIL_BN_MAX_BLOCKS;   // one statistics partial per queue
    n_cu = n;
  }
  return n_cu;
}

// geometry -> (axis, dilation, tap order); false when the call is not a 3-tap conv along one axis
// (+ optional centre tap from the second source) with complete output groups (mult = 2: pairs, F(2,3); 4: quads, F(4,3))
inline bool wconv_plan(const mdil_geom* g, int cin, wconv_args* a, int mult = 2) {
  int conv[3], nconv = 0, ad = -1;
  if (g->ntaps == 4) {          // the adapter: the one tap whose source differs from the others'
    int n1 = 0, t1 = -1, n0 = 0, t0 = -1;
    for (int t = 0; t < 4; ++t) {
      if (g->src[t]) {
        ++n1;
        t1 = t;
      } else {
        ++n0;
        t0 = t;
      }
    }
    ad = n1 == 1 ? t1 : (n0 == 1 ? t0 : -1);
    if (ad < 0 || g->dh[ad] || g->dw[ad]) return false;
  } else if (g->ntaps != 3) {
    return false;
  }
  for (int t = 0; t < g->ntaps; ++t)
    if (t != ad) conv[nconv++] = t;
  if (nconv != 3 || (g->ntaps == 4) != (ad >= 0)) return false;
  const int src3 = g->src[conv[0]];
  int axis = -1, delta = 0, ord[3] = {-1, -1, -1};
  for (int k = 0; k < 3; ++k) {
    const int t = conv[k];
    if (g->src[t] != src3) return false;
    const int dh = g->dh[t], dw = g->dw[t];
    if (dh && dw) return false;
    if (!dh && !dw) {
      ord[1] = t;
      continue;
    }
    const int ax = dw ? 1 : 0, off = dw ? dw : dh;
    if (axis >= 0 && ax != axis) return false;
    axis = ax;
    const int ad_ = off < 0 ? -off : off;
    if (delta && ad_ != delta) return false;
    delta = ad_;
    ord[off < 0 ? 0 : 2] = t;
  }
  if (axis < 0 || delta <= 0 || ord[0] < 0 || ord[1] < 0 || ord[2] < 0) return false;
  const int L = axis ? g->WO : g->HO;
  if (L % (mult * delta)) return false;
  if (a) {
    a->axis = axis;
    a->delta = delta;
    for (int k = 0; k < 3; ++k) a->tap[k] = ord[k];
    a->tap_ad = ad >= 0 ? ad : 0;
    a->src_ad = ad >= 0 ? g->src[ad] : 0;
    a->src3 = src3;
  }
  return true;
}

}  // namespace
