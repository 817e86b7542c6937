// Shared device/host helpers for libmdil_hip.so (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mdil_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MDIL_WG 256  // 4 wavefronts of 64 lanes

// BatchNorm partial-summary blocks a workspace holds (layout: partial[MDIL_BN_MAX_BLOCKS][2][C],
// counts, coefficients; bn.hip, blocks.cpp, ops.py).  The persistent conv kernels emit one partial
// per work-group queue, so their queue count is capped at this (a part with more than 256 CUs
// leaves the surplus idle rather than overrunning the layout).
#define MDIL_BN_MAX_BLOCKS 256

// thread-local last-error text (mdil_last_error)
void mdil_set_error(const char* fmt, ...);

#define MDIL_CHECK_ARG(cond, ...)            \
  do {                                       \
    if (!(cond)) {                           \
      mdil_set_error(__VA_ARGS__);           \
      return MDIL_ERR_INVALID;               \
    }                                        \
  } while (0)

#define MDIL_CHECK_LAUNCH()                                              \
  do {                                                                   \
    hipError_t e__ = hipGetLastError();                                  \
    if (e__ != hipSuccess) {                                             \
      mdil_set_error("%s:%d launch failed: %s", __FILE__, __LINE__,      \
                     hipGetErrorString(e__));                            \
      return MDIL_ERR_LAUNCH;                                            \
    }                                                                    \
  } while (0)

// HBM-bound helper kernels (BN, pooling, losses, reductions) raise their wave priority: under the
// multi-stream step they share CUs with MFMA-bound conv waves (priority 0, 1 while in their MFMA
// phase) and only need a few issue slots to keep their loads in flight.  Measured +1 % on the
// step (173.4 -> 175.2 img/s) for the BN family alone.
#ifndef MDIL_HBM_PRIO
#define MDIL_HBM_PRIO 2
#endif
#define MDIL_HBM_KERNEL_PRIO() __builtin_amdgcn_s_setprio(MDIL_HBM_PRIO)

// prof.cpp: brackets one launch with events when a profile is running (mdil_profile_begin)
struct MdilProfScope {
  MdilProfScope(hipStream_t st, int kind, const mdil_geom* g, int cin, int cout);
  ~MdilProfScope();
  int idx;
  int path;
  hipStream_t st;
};

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// f32-input MFMA: D[16x16] += A[16x4] * B[4x16]: exact fp32 products, fp32 accumulation.
//   A operand: lane l holds A[i = l & 15][k = l >> 4]
//   B operand: lane l holds B[k = l >> 4][j = l & 15]
//   C/D      : lane l, reg r holds D[row = 4 * (l >> 4) + r][col = l & 15]
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Chan/Welford merge of two (count, mean, M2) summaries.
__device__ __forceinline__ void welford_merge(float& n, float& mean, float& m2, float nb,
                                              float meanb, float m2b) {
  if (nb == 0.f) return;
  float nn = n + nb;
  float d = meanb - mean;
  float f = nb / nn;
  mean = mean + d * f;
  m2 = m2 + m2b + d * d * n * f;
  n = nn;
}

// sconv.hip: streaming (barrier-free, weights resident in LDS) C -> C stride-1 tap convolution;
// MDIL_ERR_UNSUPPORTED when the call is outside its coverage (mdil_sconv_covers).  `stats` /
// `stats_count` (optional): one (mean, M2) / count partial per work-group queue
// (mdil_sconv_stat_blocks of them) of the STORED values, for the BatchNorm that follows.
// With `bn_z` / `bn_mean` / `bn_invstd` the partials are instead the BatchNorm-BACKWARD reductions
// (sum g, sum g * xhat) of the stored gradient g (xhat from the BN input z).
bool mdil_sconv_covers(const mdil_geom* g, int cin, int cout);
int mdil_sconv_stat_blocks(const mdil_geom* g, int cin);
// *fused (optional, out): 1 when the launch also finalized its partial rows (ff / fb given and the
// Winograd kernel took the call), 0 when the caller still has to run the stand-alone finalize.
int mdil_sconv(const mdil_geom* g, int cin, int cout, const float* in0, const float* in1,
               const float* wpk, const mdil_epilogue* epi, float* out, float* stats,
               float* stats_count, const float* bn_z, const float* bn_mean, const float* bn_invstd,
               hipStream_t st, const struct BnFinFwd* ff = nullptr, const struct BnFinBwd* fb = nullptr,
               int* fused = nullptr);

// c16conv.hip: streaming (no LDS, weights in registers) 16 -> 16 channel stride-1 3-tap convolution
bool mdil_c16conv_covers(const mdil_geom* g, int cin, int cout, const mdil_epilogue* e);
int mdil_c16conv(const mdil_geom* g, const float* in0, const float* in1, const float* wpk,
                 const mdil_epilogue* epi, float* out, hipStream_t st);

// wconv.hip: the 3-tap convs (+ optional adapter tap) in Winograd F(2,3) form along the conv axis:
// 4 instead of 6 MFMA contractions per output pair; same contract and statistics layout as mdil_sconv
bool mdil_wconv_covers(const mdil_geom* g, int cin, int cout);
int mdil_wconv_stat_blocks(const mdil_geom* g, int cin);
bool mdil_wconv_tail_covers(const mdil_geom* g);   // tail form: batch size the LDS staging holds
// tail_gate (+ optional tail_drop [N][C]): the "tail" form -- the stored value is gated by
// tail_gate > 0 and the partials are the BatchNorm-backward reductions of stored * tail_drop against
// bn_z; the epilogue's residual (+ res_gate) is applied first.
// ff / fb (bnfin.h; optional): the launch's last-arriving work-group also finalizes the partial rows
// (statistics -> coefficients / reductions -> dgamma, dbeta and the apply coefficients).
struct BnFinFwd;
struct BnFinBwd;
int mdil_wconv(const mdil_geom* g, int cin, const float* in0, const float* in1, const float* wpk,
               const mdil_epilogue* epi, float* out, float* stats, float* stats_count,
               const float* bn_z, const float* bn_mean, const float* bn_invstd, hipStream_t st,
               const float* tail_gate = nullptr, const float* tail_drop = nullptr,
               const BnFinFwd* ff = nullptr, const BnFinBwd* fb = nullptr);

// w4conv.hip: the same in Winograd F(4,3) form (6 contractions per output QUAD; axis length a
// multiple of 4 x dilation).  mdil_wconv / mdil_wconv_stat_blocks dispatch to it where it covers;
// mdil_wconv_form tells which form a call takes (4, 2, or 0: neither).
bool mdil_w4conv_covers(const mdil_geom* g, int cin, int cout);
int mdil_w4conv_stat_blocks(const mdil_geom* g, int cin);
int mdil_w4conv(const mdil_geom* g, int cin, const float* in0, const float* in1, const float* wpk,
                const mdil_epilogue* epi, float* out, float* stats, float* stats_count,
                const float* bn_z, const float* bn_mean, const float* bn_invstd, hipStream_t st,
                const float* tail_gate = nullptr, const float* tail_drop = nullptr,
                const BnFinFwd* ff = nullptr, const BnFinBwd* fb = nullptr);
static inline int mdil_wconv_form(const mdil_geom* g, int cin, int cout) {
  return mdil_w4conv_covers(g, cin, cout) ? 4 : mdil_wconv_covers(g, cin, cout) ? 2 : 0;
}

// bn.hip: stand-alone finalize launches (ONE work-group on the device functions of bnfin.h) for
// producers that could not finalize their own partial rows
int mdil_bn_finalize_fwd(const float* partial, const float* pcount, int nblk, int C, const BnFinFwd& f,
                         hipStream_t st);
int mdil_bn_finalize_bwd(const float* partial, int nblk, int C, const BnFinBwd& f, hipStream_t st);
