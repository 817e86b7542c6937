// Weight (and bias) gradients of a tap convolution on gfx950, NHWC fp32.
//
//   dW[t][co][ci] = sum_p gout[out(p)][co] * in_t[in(p,t)][ci]          (K = pixels)
//
// Split-K over pixel chunks: workgroup (chunk, tap) streams its pixels in stages of 64, stages
// the gout tile [64][CO] and the (tap-shifted) input tile [64][CI] in LDS, and contracts them on
// v_mfma_f32_16x16x4_f32 with M = co, N = ci, K = pixel (operands are single ds_read_b32 per
// lane, rows padded so the two 16-lane halves of a 32-lane read group fall on different bank
// halves).  Partials [chunk][t][CO_P][CI_P] go to the caller's workspace; a second kernel adds
// them in chunk order (fixed order => run-to-run deterministic, no float atomics) and scatters
// into the PyTorch weight layout.  The bias gradient (column sums of gout) rides along in the
// tap-0 workgroups.  Output tiles are at most 64x64 channels per workgroup (blockIdx.z walks the
// tiles of a 128-channel conv): small accumulators -> 3 workgroups per CU, so one workgroup's
// staging/barriers hide under another's MFMAs.  The trailing taps of a launch may be routed to a
// second weight tensor (the 1x1 adapter rides as 4th tap of the 1x3 conv it is summed with).
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {


__host__ __device__ constexpr int pad32(int c) { return (c % 32 == 0) ? c + 16 : c; }

template <int CO, int CI, bool STEM_>
struct WgCfg {
  static constexpr bool STEM = STEM_;
  static constexpr int CO_P = (CO + 15) / 16 * 16;
  static constexpr int CI_P = STEM ? 32 : (CI + 15) / 16 * 16;
  static constexpr int MT = CO_P / 16, NT = CI_P / 16;
  // pixels per stage: small channel tiles take longer stages so a stage still carries enough
  // MFMAs to amortise its two barriers
  static constexpr int PS = (CO_P * CI_P <= 256) ? 256 : ((CO_P * CI_P <= 1024) ? 128 : 64);
  static constexpr int LDG = pad32(CO_P), LDX = pad32(CI_P);
  static constexpr bool TILE_SPLIT = (MT % 2 == 0) && (NT % 2 == 0) && (MT * NT >= 16);
  static constexpr int TM = TILE_SPLIT ? MT / 2 : MT;
  static constexpr int TN = TILE_SPLIT ? NT / 2 : NT;
  static constexpr int SMEM_TILES = PS * (LDG + LDX);
  static constexpr int SMEM_RED = TILE_SPLIT ? 0 : 4 * MT * NT * 256;
  static constexpr int SMEM = SMEM_TILES > SMEM_RED ? SMEM_TILES : SMEM_RED;
};

struct WgPlan {
  int stages_total, stages_per_chunk, nchunks;
};

inline size_t max2(size_t a, size_t b) { return a > b ? a : b; }

inline WgPlan make_plan(long long npix, int ntaps, int nz, int PS) {
  WgPlan p;
  p.stages_total = cdiv(npix, PS);
  const int target = 768;  // workgroups per launch (3 per CU); partial bytes = WGs x tile bytes
  int nch = target / (ntaps * nz);
  if (nch < 1) nch = 1;
  if (nch > p.stages_total) nch = p.stages_total;
  p.stages_per_chunk = cdiv(p.stages_total, nch);
  p.nchunks = cdiv(p.stages_total, p.stages_per_chunk);
  return p;
}

template <int CO, int CI, bool STEM>
__global__ __launch_bounds__(MDIL_WG) void wgrad_kernel(const mdil_geom g,
                                                        const float* __restrict__ in0,
                                                        const float* __restrict__ in1,
                                                        const float* __restrict__ gout,
                                                        int stages_per_chunk, int want_bias,
                                                        int co_total, int ci_total, int nz_ci,
                                                        float* __restrict__ partial,
                                                        float* __restrict__ partial_bias) {
  // CO / CI are the TILE dims; the conv has co_total x ci_total channels, tile z = blockIdx.z
  using C = WgCfg<CO, CI, STEM>;
  constexpr int PS = C::PS;
  const int co_base = (blockIdx.z / nz_ci) * CO, ci_base = (blockIdx.z % nz_ci) * CI;
  constexpr bool PIPE = !STEM && (CO % 4 == 0);       // register-prefetched staging
  constexpr int QG = C::CO_P / 4, QX = C::CI_P / 4;
  constexpr int GI = PIPE ? PS * QG / MDIL_WG : 1, XI = PIPE ? PS * QX / MDIL_WG : 1;
  __shared__ __attribute__((aligned(16))) float smem[C::SMEM + 2 * PS * 4];
  float* Gs = smem;
  float* Xs = smem + PS * C::LDG;
  int* pcb = reinterpret_cast<int*>(smem + C::SMEM);  // [2][PS][4]: n, ho, wo, valid

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int chunk = blockIdx.x, t = blockIdx.y;
  const int npix = g.N * g.HO * g.WO, hw = g.HO * g.WO;
  const int s_src = STEM ? 0 : g.src[t];
  const float* __restrict__ xin = s_src ? in1 : in0;
  const int xpitch = STEM ? 3 : g.in_pitch[s_src];
  const int dh = STEM ? 0 : g.dh[t], dw = STEM ? 0 : g.dw[t];

  f32x4 acc[C::TM][C::TN];
#pragma unroll
  for (int a = 0; a < C::TM; ++a)
#pragma unroll
    for (int b = 0; b < C::TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  constexpr int BCOLS = CO <= 16 ? 16 : (CO <= 32 ? 32 : 64);   // threads across a tile's columns
  constexpr int BPARTS = MDIL_WG / BCOLS;                        // row slices per column
  const int bcol = tid % BCOLS, bpart = tid / BCOLS;

  const int m0 = C::TILE_SPLIT ? (wave >> 1) * C::TM : 0;
  const int n0 = C::TILE_SPLIT ? (wave & 1) * C::TN : 0;

  auto fill_pc = [&](int* pc, int P0) {   // threads < PS: pixel coordinates of one stage
    if (tid < PS) {
      const int P = P0 + tid;
      int n = 0, ho = 0, wo = 0, ok = 0;
      if (P < npix) {
        n = P / hw;
        const int r = P - n * hw;
        ho = r / g.WO;
        wo = r - ho * g.WO;
        ok = 1;
      }
      pc[tid * 4 + 0] = n;
      pc[tid * 4 + 1] = ho;
      pc[tid * 4 + 2] = wo;
      pc[tid * 4 + 3] = ok;
    }
  };

#ifndef WG_PF2
#define WG_PF2 0      // 1: global loads run two stages ahead of their LDS write (two register sets);
                      // measured -1.5 % per launch (142 VGPRs, no latency left to hide), so off
#endif
  constexpr int NSET = (WG_PF2 && PIPE) ? 2 : 1;
  f32x4 regGs[NSET][GI], regXs[NSET][XI];
  unsigned okGs[NSET] = {}, okXs[NSET] = {};
  // all loads are unconditional (clamped address); zero-fill happens at LDS-write time so that
  // nothing waits on a load before the MFMAs of the current stage.  The pixel coordinates are
  // pulled from LDS into registers first so the global loads issue back to back.
  auto issue_loads = [&](auto SET, const int* pc) __attribute__((always_inline)) {
    f32x4 (&regG)[GI] = regGs[decltype(SET)::value];
    f32x4 (&regX)[XI] = regXs[decltype(SET)::value];
    unsigned& okG = okGs[decltype(SET)::value];
    unsigned& okX = okXs[decltype(SET)::value];
    if constexpr (PIPE) {
      typedef int i32x4 __attribute__((ext_vector_type(4)));
      i32x4 cg[GI], cx[XI];
#pragma unroll
      for (int i = 0; i < GI; ++i) cg[i] = *reinterpret_cast<const i32x4*>(&pc[((tid + MDIL_WG * i) / QG) * 4]);
#pragma unroll
      for (int i = 0; i < XI; ++i) cx[i] = *reinterpret_cast<const i32x4*>(&pc[((tid + MDIL_WG * i) / QX) * 4]);
#pragma unroll
      for (int i = 0; i < GI; ++i) {
        const int q = (tid + MDIL_WG * i) % QG;
        const bool ok = cg[i][3] && q * 4 < CO && co_base + q * 4 < co_total;
        long long off = ((long long)(cg[i][0] * g.OH + cg[i][1] * g.ohs + g.oho) * g.OW +
                         (cg[i][2] * g.ows + g.owo)) * g.out_pitch + g.out_coff + co_base + q * 4;
        off = ok ? off : (long long)g.out_coff;   // computed unconditionally: no divergent region
        regG[i] = *reinterpret_cast<const f32x4*>(gout + off);
        okG = ok ? (okG | (1u << i)) : (okG & ~(1u << i));
      }
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        const int q = (tid + MDIL_WG * i) % QX;
        const int hi = cx[i][1] * g.ihs + dh, wi = cx[i][2] * g.iws + dw;
        const bool ok = cx[i][3] && hi >= 0 && hi < g.HI && wi >= 0 && wi < g.WI && q * 4 < CI &&
                        ci_base + q * 4 < ci_total;
        long long off = ((long long)(cx[i][0] * g.HI + hi) * g.WI + wi) * xpitch + ci_base + q * 4;
        off = ok ? off : 0ll;
        regX[i] = *reinterpret_cast<const f32x4*>(xin + off);
        okX = ok ? (okX | (1u << i)) : (okX & ~(1u << i));
      }
    }
  };
  auto write_lds = [&](auto SET) __attribute__((always_inline)) {
    f32x4 (&regG)[GI] = regGs[decltype(SET)::value];
    f32x4 (&regX)[XI] = regXs[decltype(SET)::value];
    const unsigned okG = okGs[decltype(SET)::value], okX = okXs[decltype(SET)::value];
    if constexpr (PIPE) {
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < GI; ++i) {
        const int idx = tid + MDIL_WG * i;
        *reinterpret_cast<f32x4*>(&Gs[(idx / QG) * C::LDG + (idx % QG) * 4]) =
            ((okG >> i) & 1u) ? regG[i] : z;
      }
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        const int idx = tid + MDIL_WG * i;
        *reinterpret_cast<f32x4*>(&Xs[(idx / QX) * C::LDX + (idx % QX) * 4]) =
            ((okX >> i) & 1u) ? regX[i] : z;
      }
    }
  };

  const int st_begin = chunk * stages_per_chunk;
  int st_end = st_begin + stages_per_chunk;
  {
    const int total = (npix + PS - 1) / PS;
    if (st_end > total) st_end = total;
  }
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, NSET - 1>;
  constexpr int DIST = NSET;
  auto compute_stage = [&]() __attribute__((always_inline)) {
    if (want_bias && t == 0 && ci_base == 0 && bcol < CO) {
      // column sums of the gout tile, spread over all 256 threads: thread (bcol, bpart) adds its
      // PS/BPARTS rows; the BPARTS slices are combined once at the end of the kernel
      float s = 0.f;
#pragma unroll
      for (int p = 0; p < PS / BPARTS; ++p) s += Gs[(bpart * (PS / BPARTS) + p) * C::LDG + bcol];
      bsum += s;
    }
    // ---- MFMA over the 64 pixels of the stage; operand reads run one k-step ahead ----
    if constexpr (C::TILE_SPLIT) {
      float a[2][C::TM], b[2][C::TN];
#pragma unroll
      for (int m = 0; m < C::TM; ++m) a[0][m] = Gs[lg * C::LDG + (m0 + m) * 16 + li];
#pragma unroll
      for (int n = 0; n < C::TN; ++n) b[0][n] = Xs[lg * C::LDX + (n0 + n) * 16 + li];
#pragma unroll
      for (int s = 0; s < PS / 4; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        if (s + 1 < PS / 4) {
          const int row = 4 * (s + 1) + lg;
#pragma unroll
          for (int m = 0; m < C::TM; ++m) a[nxt][m] = Gs[row * C::LDG + (m0 + m) * 16 + li];
#pragma unroll
          for (int n = 0; n < C::TN; ++n) b[nxt][n] = Xs[row * C::LDX + (n0 + n) * 16 + li];
        }
#pragma unroll
        for (int m = 0; m < C::TM; ++m)
#pragma unroll
          for (int n = 0; n < C::TN; ++n) acc[m][n] = mfma16(a[cur][m], b[cur][n], acc[m][n]);
      }
    } else {
      float a[PS / 16][C::TM], b[PS / 16][C::TN];   // wave w takes k-steps s = 4*ss + w
#pragma unroll
      for (int ss = 0; ss < PS / 16; ++ss) {
        const int row = 4 * (4 * ss + wave) + lg;
#pragma unroll
        for (int m = 0; m < C::TM; ++m) a[ss][m] = Gs[row * C::LDG + m * 16 + li];
#pragma unroll
        for (int n = 0; n < C::TN; ++n) b[ss][n] = Xs[row * C::LDX + n * 16 + li];
      }
#pragma unroll
      for (int ss = 0; ss < PS / 16; ++ss)
#pragma unroll
        for (int m = 0; m < C::TM; ++m)
#pragma unroll
          for (int n = 0; n < C::TN; ++n) acc[m][n] = mfma16(a[ss][m], b[ss][n], acc[m][n]);
    }
  };
  // one stage: hand register set SET to LDS, refill it for stage st + DIST (in flight under this
  // and, with two sets, the next stage's MFMAs), contract the stage
  auto stage = [&](auto SET, int st) __attribute__((always_inline)) {
    const int P0 = st * PS;
    int* pc = pcb + ((st - st_begin) & 1) * PS * 4;
    __syncthreads();  // previous stage fully consumed
    if constexpr (PIPE) {
      write_lds(SET);
      if (st + DIST < st_end) fill_pc(pc, (st + DIST) * PS);
      __syncthreads();
      if (st + DIST < st_end) issue_loads(SET, pc);
    } else {
      fill_pc(pc, P0);
      __syncthreads();
      // ---- gout tile (scalar path: CO not a multiple of 4 -> the 13-channel stem slice) ----
      for (int idx = tid; idx < PS * C::CO_P; idx += MDIL_WG) {
        const int p = idx / C::CO_P, c = idx % C::CO_P;
        const bool ok = pc[p * 4 + 3] && c < CO;
        const long long off =
            ok ? ((long long)(pc[p * 4] * g.OH + pc[p * 4 + 1] * g.ohs + g.oho) * g.OW +
                  (pc[p * 4 + 2] * g.ows + g.owo)) * g.out_pitch + g.out_coff + c
               : (long long)g.out_coff;
        const float v = gout[off];
        Gs[p * C::LDG + c] = ok ? v : 0.f;
      }
      // ---- im2col-on-load of the 3x3 stride-2 RGB stem ----
      for (int idx = tid; idx < PS * 9; idx += MDIL_WG) {
        const int p = idx / 9, tap = idx - p * 9;
        const int hi = 2 * pc[p * 4 + 1] + tap / 3 - 1, wi = 2 * pc[p * 4 + 2] + tap % 3 - 1;
        const bool ok = pc[p * 4 + 3] && hi >= 0 && hi < g.HI && wi >= 0 && wi < g.WI;
        const float* sp = in0 + (ok ? ((long long)(pc[p * 4] * g.HI + hi) * g.WI + wi) * 3 : 0ll);
        const float v0 = sp[0], v1 = sp[1], v2 = sp[2];
        float* d = &Xs[p * C::LDX + 3 * tap];
        d[0] = ok ? v0 : 0.f;
        d[1] = ok ? v1 : 0.f;
        d[2] = ok ? v2 : 0.f;
      }
      for (int idx = tid; idx < PS * 5; idx += MDIL_WG) Xs[(idx / 5) * C::LDX + 27 + idx % 5] = 0.f;
      __syncthreads();
    }
    compute_stage();
  };
  if constexpr (PIPE) {
    if (st_begin < st_end) {
      fill_pc(pcb, st_begin * PS);
      if (NSET == 2 && st_begin + 1 < st_end) fill_pc(pcb + PS * 4, (st_begin + 1) * PS);
      __syncthreads();
      issue_loads(Set0{}, pcb);
      if (NSET == 2 && st_begin + 1 < st_end) issue_loads(Set1{}, pcb + PS * 4);
    }
  }
  if constexpr (NSET == 2) {
    for (int st = st_begin; st < st_end; st += 2) {
      stage(Set0{}, st);
      if (st + 1 < st_end) stage(Set1{}, st + 1);
    }
  } else {
    for (int st = st_begin; st < st_end; ++st) stage(Set0{}, st);
  }

  float* pout = partial + ((long long)((chunk * gridDim.y + t) * gridDim.z + blockIdx.z)) * C::CO_P * C::CI_P;
  if constexpr (C::TILE_SPLIT) {
#pragma unroll
    for (int m = 0; m < C::TM; ++m)
#pragma unroll
      for (int n = 0; n < C::TN; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          pout[((m0 + m) * 16 + 4 * lg + r) * C::CI_P + (n0 + n) * 16 + li] = acc[m][n][r];
  } else {
    __syncthreads();
    float* red = smem;  // [wave][tile][lane][4]
#pragma unroll
    for (int m = 0; m < C::TM; ++m)
#pragma unroll
      for (int n = 0; n < C::TN; ++n)
        *reinterpret_cast<f32x4*>(&red[((wave * C::MT * C::NT + m * C::NT + n) * 64 + lane) * 4]) =
            acc[m][n];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int m = 0; m < C::TM; ++m)
#pragma unroll
        for (int n = 0; n < C::TN; ++n) {
          f32x4 v = acc[m][n];
#pragma unroll
          for (int w = 1; w < 4; ++w)
            v += *reinterpret_cast<const f32x4*>(
                &red[((w * C::MT * C::NT + m * C::NT + n) * 64 + lane) * 4]);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            pout[(m * 16 + 4 * lg + r) * C::CI_P + n * 16 + li] = v[r];
        }
    }
  }
  if (want_bias && t == 0 && ci_base == 0) {
    __syncthreads();  // tiles / reduction scratch no longer needed
    float* bs = smem;  // [BPARTS][BCOLS]
    bs[bpart * BCOLS + bcol] = bsum;
    __syncthreads();
    if (tid < CO) {
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < BPARTS; ++k) s += bs[k * BCOLS + tid];
      partial_bias[(long long)chunk * (gridDim.z / nz_ci) * C::CO_P + (blockIdx.z / nz_ci) * C::CO_P + tid] = s;
    }
  }
}

// 256 threads = 32 consecutive outputs x 8 chunk slices; slice j adds chunks j, j+8, ... in
// order, the 8 slice sums are then added in a fixed tree (deterministic).
constexpr int RED_OUT = 32, RED_SL = 8, RED_MLP = 16;

struct RedArgs {
  int nchunks, ntaps, nz, nz_ci;
  int CO, CI;        // channel counts of the conv
  int CO_T, CI_T;    // tile dims, padded tile dims
  int CO_P, CI_P;
  int ktap[MDIL_MAX_TAPS];
  int ntaps1;        // taps [0, ntaps1) -> dw / dbias, taps [ntaps1, ntaps) -> dw2 / dbias2
  int s_co, s_ci, s_co2, s_ci2;
  int stem, accumulate, nblk_w;
};

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial,
                                                           const float* __restrict__ partial_bias,
                                                           const RedArgs a, float* __restrict__ dw,
                                                           float* __restrict__ dbias,
                                                           float* __restrict__ dw2,
                                                           float* __restrict__ dbias2) {
  MDIL_HBM_KERNEL_PRIO();

  __shared__ float sh[RED_SL][RED_OUT];
  const int ox = threadIdx.x % RED_OUT, sl = threadIdx.x / RED_OUT;
  const bool bias_blk = (int)blockIdx.x >= a.nblk_w;
  const int total = bias_blk ? a.CO : a.ntaps * a.CO * a.CI;
  const int gid = (bias_blk ? (blockIdx.x - a.nblk_w) : blockIdx.x) * RED_OUT + ox;
  float s = 0.f;
  int t = 0, co = 0, ci = 0;
  if (gid < total) {
    if (bias_blk) {
      const int cob = (a.nz / a.nz_ci) * a.CO_P;   // bias partial row length
      const int idx = (gid / a.CO_T) * a.CO_P + gid % a.CO_T;
      for (int c0 = sl; c0 < a.nchunks; c0 += RED_SL * RED_MLP) {
        float v[RED_MLP];
#pragma unroll
        for (int u = 0; u < RED_MLP; ++u) {      // clamped, unconditional: RED_MLP loads in flight
          const int c = c0 + u * RED_SL;
          const float x = partial_bias[(long long)(c < a.nchunks ? c : a.nchunks - 1) * cob + idx];
          v[u] = c < a.nchunks ? x : 0.f;
        }
#pragma unroll
        for (int u = 0; u < RED_MLP; ++u) s += v[u];
      }
    } else {
      ci = gid % a.CI;
      co = (gid / a.CI) % a.CO;
      t = gid / (a.CI * a.CO);
      const int z = (co / a.CO_T) * a.nz_ci + ci / a.CI_T;
      const float* p = partial + (((long long)t * a.nz + z) * a.CO_P + co % a.CO_T) * a.CI_P + ci % a.CI_T;
      const long long stride = (long long)a.ntaps * a.nz * a.CO_P * a.CI_P;
      for (int c0 = sl; c0 < a.nchunks; c0 += RED_SL * RED_MLP) {
        float v[RED_MLP];
#pragma unroll
        for (int u = 0; u < RED_MLP; ++u) {
          const int c = c0 + u * RED_SL;
          const float x = p[(long long)(c < a.nchunks ? c : a.nchunks - 1) * stride];
          v[u] = c < a.nchunks ? x : 0.f;
        }
#pragma unroll
        for (int u = 0; u < RED_MLP; ++u) s += v[u];
      }
    }
  }
  sh[sl][ox] = s;
  __syncthreads();
  if (sl == 0 && gid < total) {
    const float r = ((sh[0][ox] + sh[1][ox]) + (sh[2][ox] + sh[3][ox])) +
                    ((sh[4][ox] + sh[5][ox]) + (sh[6][ox] + sh[7][ox]));
    if (bias_blk) {
      if (dbias) dbias[gid] = a.accumulate ? dbias[gid] + r : r;
      if (dbias2) dbias2[gid] = a.accumulate ? dbias2[gid] + r : r;
    } else if (a.stem) {
      const long long dst = (long long)co * 27 + (ci % 3) * 9 + ci / 3;  // [13][3][3][3] <- 3*tap+c
      dw[dst] = a.accumulate ? dw[dst] + r : r;
    } else if (t < a.ntaps1) {
      const long long dst = (long long)co * a.s_co + (long long)ci * a.s_ci + a.ktap[t];
      dw[dst] = a.accumulate ? dw[dst] + r : r;
    } else {
      const long long dst = (long long)co * a.s_co2 + (long long)ci * a.s_ci2 + a.ktap[t];
      dw2[dst] = a.accumulate ? dw2[dst] + r : r;
    }
  }
}

// The same reduction, four consecutive input channels per thread (16-byte partial loads): used
// whenever the channel counts are multiples of 4 (every conv but the RGB stem).  32 float4 groups
// x 8 chunk slices per block, 16 loads in flight per thread.
__device__ __forceinline__ void reduce4_block(const int block, const float* __restrict__ partial,
                                              const float* __restrict__ partial_bias, const RedArgs& a,
                                              float* __restrict__ dw, float* __restrict__ dbias,
                                              float* __restrict__ dw2, float* __restrict__ dbias2) {
  __shared__ f32x4 sh[RED_SL][RED_OUT];
  const int ox = threadIdx.x % RED_OUT, sl = threadIdx.x / RED_OUT;
  const bool bias_blk = block >= a.nblk_w;
  const int total4 = bias_blk ? a.CO / 4 : a.ntaps * a.CO * (a.CI / 4);
  const int gid = (bias_blk ? (block - a.nblk_w) : block) * RED_OUT + ox;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int t = 0, co = 0, ci = 0;
  if (gid < total4) {
    const float* p;
    long long stride;
    if (bias_blk) {
      const int c0 = gid * 4;
      stride = (long long)(a.nz / a.nz_ci) * a.CO_P;
      p = partial_bias + (c0 / a.CO_T) * a.CO_P + c0 % a.CO_T;
    } else {
      const int q = a.CI / 4;
      ci = (gid % q) * 4;
      co = (gid / q) % a.CO;
      t = gid / (q * a.CO);
      const int z = (co / a.CO_T) * a.nz_ci + ci / a.CI_T;
      p = partial + (((long long)t * a.nz + z) * a.CO_P + co % a.CO_T) * a.CI_P + ci % a.CI_T;
      stride = (long long)a.ntaps * a.nz * a.CO_P * a.CI_P;
    }
    for (int c0 = sl; c0 < a.nchunks; c0 += RED_SL * RED_MLP) {
      f32x4 v[RED_MLP];
#pragma unroll
      for (int u = 0; u < RED_MLP; ++u) {      // clamped, unconditional: RED_MLP loads in flight
        const int c = c0 + u * RED_SL;
        const f32x4 x = *reinterpret_cast<const f32x4*>(p + (long long)(c < a.nchunks ? c : a.nchunks - 1) * stride);
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        v[u] = c < a.nchunks ? x : zero;
      }
#pragma unroll
      for (int u = 0; u < RED_MLP; ++u) s += v[u];
    }
  }
  sh[sl][ox] = s;
  __syncthreads();
  if (sl == 0 && gid < total4) {
    const f32x4 r = ((sh[0][ox] + sh[1][ox]) + (sh[2][ox] + sh[3][ox])) +
                    ((sh[4][ox] + sh[5][ox]) + (sh[6][ox] + sh[7][ox]));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (bias_blk) {
        const int c = gid * 4 + k;
        if (dbias) dbias[c] = a.accumulate ? dbias[c] + r[k] : r[k];
        if (dbias2) dbias2[c] = a.accumulate ? dbias2[c] + r[k] : r[k];
      } else if (t < a.ntaps1) {
        const long long dst = (long long)co * a.s_co + (long long)(ci + k) * a.s_ci + a.ktap[t];
        dw[dst] = a.accumulate ? dw[dst] + r[k] : r[k];
      } else {
        const long long dst = (long long)co * a.s_co2 + (long long)(ci + k) * a.s_ci2 + a.ktap[t];
        dw2[dst] = a.accumulate ? dw2[dst] + r[k] : r[k];
      }
    }
  }
}

__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(const float* __restrict__ partial,
                                                            const float* __restrict__ partial_bias,
                                                            const RedArgs a, float* __restrict__ dw,
                                                            float* __restrict__ dbias,
                                                            float* __restrict__ dw2,
                                                            float* __restrict__ dbias2) {
  MDIL_HBM_KERNEL_PRIO();
  reduce4_block((int)blockIdx.x, partial, partial_bias, a, dw, dbias, dw2, dbias2);
}

// Deferred reductions: a weight-gradient launch may leave its partial sums in a caller-owned arena
// and hand back a job record instead of launching its own reduction; up to RED_BATCH jobs are then
// reduced by ONE launch (the record table travels as the kernel argument).  A training step has
// ~140 weight-gradient launches on the factorised blocks; their 8 us reductions are latency-bound
// launches between chip-filling MFMA kernels and cost the step 4 % (measured by skipping them).
struct RedJob {
  RedArgs a;
  const float* partial;
  const float* pbias;
  float *dw, *dbias, *dw2, *dbias2;
  int nblk;          // blocks of this job (weights + bias)
  int pad_;
};
static_assert(sizeof(RedJob) <= sizeof(mdil_wgrad_job), "mdil_wgrad_job is too small");
constexpr int RED_BATCH = 16;
struct RedBatch {
  int njobs;
  int start[RED_BATCH + 1];   // first block of every job
  RedJob jobs[RED_BATCH];
};
static_assert(sizeof(RedBatch) <= 4000, "the job table must fit the kernel argument segment");

__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(const RedBatch b) {
  MDIL_HBM_KERNEL_PRIO();
  int j = 0;
  for (int k = 1; k < b.njobs; ++k) j = (int)blockIdx.x >= b.start[k] ? k : j;
  const RedJob& jb = b.jobs[j];
  reduce4_block((int)blockIdx.x - b.start[j], jb.partial, jb.pbias, jb.a, jb.dw, jb.dbias, jb.dw2, jb.dbias2);
}

// launches the reduction that fits the channel counts
// `defer` != nullptr: no launch; *defer receives the job (the caller batches it, see RedJob)
inline void launch_reduce(const RedArgs& a0, int want_bias, const float* partial, const float* pbias,
                          float* dw, float* dbias, float* dw2, float* dbias2, hipStream_t st,
                          mdil_wgrad_job* defer = nullptr) {
  RedArgs a = a0;
#ifdef WG_SKIP_REDUCE   // tuning builds only (results wrong): upper bound of what fusing the reduction could gain
  return;
#endif
  const bool vec = !a.stem && a.CI % 4 == 0 && a.CO % 4 == 0 && a.CI_T % 4 == 0 && a.CO_T % 4 == 0 &&
                   a.CI_P % 4 == 0 && a.CO_P % 4 == 0;
  if (vec) {
    a.nblk_w = cdiv(a.ntaps * a.CO * (a.CI / 4), RED_OUT);
    const int nblk_b = want_bias ? cdiv(a.CO / 4, RED_OUT) : 0;
    if (defer) {
      RedJob j;
      memset(&j, 0, sizeof(j));
      j.a = a;
      j.partial = partial;
      j.pbias = pbias;
      j.dw = dw;
      j.dbias = dbias;
      j.dw2 = dw2;
      j.dbias2 = dbias2;
      j.nblk = a.nblk_w + nblk_b;
      memset(defer, 0, sizeof(*defer));
      memcpy(defer, &j, sizeof(j));
      return;
    }
    hipLaunchKernelGGL(wgrad_reduce4_kernel, dim3(a.nblk_w + nblk_b), dim3(256), 0, st, partial, pbias,
                       a, dw, dbias, dw2, dbias2);
  } else {
    a.nblk_w = cdiv(a.ntaps * a.CO * a.CI, RED_OUT);
    const int nblk_b = want_bias ? cdiv(a.CO, RED_OUT) : 0;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(a.nblk_w + nblk_b), dim3(256), 0, st, partial, pbias,
                       a, dw, dbias, dw2, dbias2);
  }
}

struct WgCall {
  const mdil_geom* g;
  const float *in0, *in1, *gout;
  const int* ktap;
  int s_co, s_ci;
  float *dw, *dbias;
  int ntaps2, s_co2, s_ci2;
  float *dw2, *dbias2;
  int accumulate;
  void* ws;
  size_t ws_bytes;
  hipStream_t st;
  int co_total, ci_total;
  mdil_wgrad_job* defer;   // non-NULL: leave the partial sums in ws and describe the reduction here
};

template <int CO_T, int CI_T, bool STEM>
size_t ws_need(const mdil_geom* g, int co_total, int ci_total) {
  using C = WgCfg<CO_T, CI_T, STEM>;
  const long long npix = (long long)g->N * g->HO * g->WO;
  const int ntaps = STEM ? 1 : g->ntaps;
  const int nz_co = cdiv(co_total, CO_T), nz_ci = STEM ? 1 : cdiv(ci_total, CI_T);
  const WgPlan p = make_plan(npix, ntaps, nz_co * nz_ci, C::PS);
  return ((size_t)p.nchunks * ntaps * nz_co * nz_ci * C::CO_P * C::CI_P +
          (size_t)p.nchunks * nz_co * C::CO_P) * sizeof(float);
}

template <int CO_T, int CI_T, bool STEM>
int launch_wgrad(const WgCall& c) {
  using C = WgCfg<CO_T, CI_T, STEM>;
  const mdil_geom* g = c.g;
  const long long npix = (long long)g->N * g->HO * g->WO;
  const int ntaps = STEM ? 1 : g->ntaps;
  const int nz_co = cdiv(c.co_total, CO_T), nz_ci = STEM ? 1 : cdiv(c.ci_total, CI_T);
  const int nz = nz_co * nz_ci;
  const WgPlan p = make_plan(npix, ntaps, nz, C::PS);
  const size_t need = ws_need<CO_T, CI_T, STEM>(g, c.co_total, c.ci_total);
  MDIL_CHECK_ARG(c.ws && c.ws_bytes >= need, "wgrad: workspace %zu < %zu", c.ws_bytes, need);
  float* partial = (float*)c.ws;
  float* pbias = partial + (size_t)p.nchunks * ntaps * nz * C::CO_P * C::CI_P;
  const int want_bias = (c.dbias || c.dbias2) ? 1 : 0;
  hipLaunchKernelGGL((wgrad_kernel<CO_T, CI_T, STEM>), dim3(p.nchunks, ntaps, nz), dim3(MDIL_WG), 0,
                     c.st, *g, c.in0, c.in1, c.gout, p.stages_per_chunk, want_bias, c.co_total,
                     STEM ? 27 : c.ci_total, nz_ci, partial, pbias);
  MDIL_CHECK_LAUNCH();
  RedArgs a;
  memset(&a, 0, sizeof(a));
  a.nchunks = p.nchunks;
  a.ntaps = ntaps;
  a.nz = nz;
  a.nz_ci = nz_ci;
  a.CO = c.co_total;
  a.CI = STEM ? 27 : c.ci_total;
  a.CO_T = CO_T;
  a.CI_T = STEM ? 32 : CI_T;
  a.CO_P = C::CO_P;
  a.CI_P = C::CI_P;
  for (int t = 0; t < ntaps && !STEM; ++t) a.ktap[t] = c.ktap[t];
  a.ntaps1 = ntaps - c.ntaps2;
  a.s_co = c.s_co;
  a.s_ci = c.s_ci;
  a.s_co2 = c.s_co2;
  a.s_ci2 = c.s_ci2;
  a.stem = STEM ? 1 : 0;
  a.accumulate = c.accumulate;
  launch_reduce(a, want_bias, partial, pbias, c.dw, c.dbias, c.dw2, c.dbias2, c.st, c.defer);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}


// ------------------------------------------------------------------------------------------------
// Streaming weight gradient for the C -> C stride-1 convs (C = 64 / 128), W % 16 == 0.
//
// fp32 MFMA is slow enough (32 cycles per instruction per SIMD) that neither operand needs LDS:
// a lane's ONE 16-byte load g[p = P0 + lg][co = 4 li .. 4 li + 3] is the A operand of four MFMAs
// (output-channel tile e = {4 i + e}), one load x[p'][ci = 4 li ..] the B operand of four (input-
// channel tile f), so two buffer loads feed the 16 MFMAs of a 64 x 64 channel block for 4 pixels.
// Each wave owns one (tap, 64 x 64 block) for a contiguous run of 16-pixel quads and streams it
// with loads one quad ahead: no barrier, no LDS, no VALU in the main loop.  Addressing is scalar:
// a buffer descriptor per image row whose num_records ends at the row end (after the tap's column
// shift), the position inside the row in soffset -- out-of-image taps read 0 through the range
// check.  Negative shifts are applied to the other operand (dW[t] = sum_q g[q - s] x[q]), so only
// the upper bound is ever needed.  The waves of a work-group cover all taps of the same pixels
// (L1 reuse of g), partials of the work-group's chunks are added through LDS in a fixed order,
// and wgrad_reduce_kernel adds the per-work-group partials in order (deterministic).
// ------------------------------------------------------------------------------------------------
typedef unsigned int u32x4w __attribute__((ext_vector_type(4)));

struct wg2_args {
  const float* in0;
  const float* in1;
  const float* gout;
  float* partial;
  float* partial_bias;
  int N, H, W;
  int dh[4], dw[4], src[4];
  int quads_per_chunk;
  int bias_tap;   // tap without shift (its g stream is the plain pixel sequence), -1: no bias
};

#ifndef WG2_PIN
#define WG2_PIN 1
#endif

template <int C, int NTAPS, int CPW>
__global__ __launch_bounds__(NTAPS * CPW * 64) void wgrad2_kernel(const wg2_args a) {
  constexpr int NB = C / 64, NZ = NB * NB;
  constexpr int STR = C * 4;                     // bytes per pixel
  constexpr int NRED = (CPW / 2) * NTAPS;        // 16 KB slots of the in-work-group reduction
  __shared__ __attribute__((aligned(16))) float red[NRED * 4096 + CPW * 64];
  float* bred = red + NRED * 4096;

  const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tap = wave % NTAPS, cw = wave / NTAPS;
  int z = 0, cg = blockIdx.x;
  if constexpr (NZ > 1) {       // the blocks of one chunk group sit on one XCD (blockIdx % 8)
    z = (blockIdx.x >> 3) & (NZ - 1);
    cg = (blockIdx.x & 7) | ((blockIdx.x >> 5) << 3);
  }
  const int cob = z / NB, cib = z % NB;
  const int H = a.H, W = a.W;
  const int nquads = a.N * H * (W >> 4);
  const int qpc = a.quads_per_chunk;
  const int q0 = (cg * CPW + cw) * qpc;
  const int q1 = min(q0 + qpc, nquads);

  const int dh = a.dh[tap], dw = a.dw[tap];
  const int axr = dh > 0 ? dh : 0, agr = dh < 0 ? -dh : 0;   // row shift of x / of g
  const int sx = dw > 0 ? dw : 0, sg = dw < 0 ? -dw : 0;     // column shift of x / of g
  const float* xin = a.src[tap] ? a.in1 : a.in0;
  const bool do_bias = a.bias_tap == tap && cib == 0;

  // prefetch pointer (scalar): next quad to load and its (image row, column)
  int pq = q0;
  int w0 = (pq % (W >> 4)) << 4;
  int row = pq / (W >> 4);            // img * H + h
  int h = row % H;
  __amdgpu_buffer_rsrc_t rsx, rsg;
  auto rows = [&]() {
    const bool live = pq < q1;
    const bool okx = live && h + axr < H && sx < W, okg = live && h + agr < H && sg < W;
    const long long ox = ((long long)(row + axr) * W + sx) * C + cib * 64;
    const long long og = ((long long)(row + agr) * W + sg) * C + cob * 64;
    rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin + (okx ? ox : 0)), 0,
                                            okx ? (W - sx - 1) * STR + 256 : 0, 0x00020000);
    rsg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gout + (okg ? og : 0)), 0,
                                            okg ? (W - sg - 1) * STR + 256 : 0, 0x00020000);
  };
  rows();
  const int voff = lg * STR + li * 16;

  f32x4 gq[2][4], xq[2][4];
  f32x4 acc[4][4];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[e][f] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

  constexpr int GI = 4096 / (4 * STR) > 0 ? 4096 / (4 * STR) : 1;   // groups reachable by the immediate
  auto load_g = [&](int s, int j) __attribute__((always_inline)) {
    const u32x4w v = __builtin_amdgcn_raw_buffer_load_b128(
        rsg, voff + (j % GI) * 4 * STR, w0 * STR + (j / GI) * GI * 4 * STR, 0);
    gq[s][j] = __builtin_bit_cast(f32x4, v);
  };
  auto load_x = [&](int s, int j) __attribute__((always_inline)) {
    const u32x4w v = __builtin_amdgcn_raw_buffer_load_b128(
        rsx, voff + (j % GI) * 4 * STR, w0 * STR + (j / GI) * GI * 4 * STR, 0);
    xq[s][j] = __builtin_bit_cast(f32x4, v);
  };
  auto advance = [&]() __attribute__((always_inline)) {
    ++pq;
    w0 += 16;
    if (w0 == W || pq >= q1) {
      if (w0 == W) {
        w0 = 0;
        ++row;
        h = (h + 1 == H) ? 0 : h + 1;
      }
      rows();
    }
  };
  auto mf = [&](int s, int j, int k) __attribute__((always_inline)) {
    const int e = k >> 2, f = k & 3;
    acc[e][f] = mfma16(gq[s][j][e], xq[s][j][f], acc[e][f]);
  };
  // one quad: MFMAs of ring slot s, loads of the NEXT quad into slot s ^ 1 spread between them
  auto quad = [&](int s) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      load_g(s ^ 1, j);
#if WG2_PIN
      __builtin_amdgcn_sched_barrier(0);
#endif
      mf(s, j, 0);
      mf(s, j, 1);
#if WG2_PIN
      __builtin_amdgcn_sched_barrier(0);
#endif
      load_x(s ^ 1, j);
#if WG2_PIN
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int k = 2; k < 16; ++k) mf(s, j, k);
      if (do_bias) bsum += gq[s][j];
#if WG2_PIN
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    advance();
  };

  if (q0 < q1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      load_g(0, j);
      load_x(0, j);
    }
    advance();
    for (int q = q0; q < q1; q += 2) {
      quad(0);
      quad(1);   // an odd tail multiplies zeros: every descriptor is empty past q1
    }
  }

  // ---- in-work-group reduction over the chunks (fixed order), then one partial per work-group ----
  auto put = [&](int slot) {
    float* d = red + slot * 4096;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int f = 0; f < 4; ++f)
        *reinterpret_cast<f32x4*>(&d[((e * 4 + f) * 64 + lane) * 4]) = acc[e][f];
  };
  auto add = [&](int slot) {
    const float* d = red + slot * 4096;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int f = 0; f < 4; ++f)
        acc[e][f] += *reinterpret_cast<const f32x4*>(&d[((e * 4 + f) * 64 + lane) * 4]);
  };
  if (do_bias) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      bsum[k] += __shfl_xor(bsum[k], 16, 64);
      bsum[k] += __shfl_xor(bsum[k], 32, 64);
    }
    if (lg == 0) *reinterpret_cast<f32x4*>(&bred[cw * 64 + li * 4]) = bsum;
  }
  if constexpr (CPW == 4) {
    if (cw >= 2) put(tap * 2 + (cw - 2));
    __syncthreads();
    if (cw < 2) add(tap * 2 + cw);
    __syncthreads();
  }
  if constexpr (CPW >= 2) {
    if (cw == 1) put(tap * (CPW / 2));
    __syncthreads();
    if (cw == 0) add(tap * (CPW / 2));
  } else {
    __syncthreads();
  }
  if (cw == 0) {
    // acc[e][f][r] = dW[co = 4 (4 lg + r) + e][ci = 4 li + f]: a lane writes 4 consecutive ci
    float* pout = a.partial + ((long long)(cg * NTAPS + tap) * NZ + z) * 4096;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x4 v = {acc[e][0][r], acc[e][1][r], acc[e][2][r], acc[e][3][r]};
        *reinterpret_cast<f32x4*>(&pout[(4 * (4 * lg + r) + e) * 64 + 4 * li]) = v;
      }
    if (do_bias) {
      float sum = 0.f;
#pragma unroll
      for (int c = 0; c < CPW; ++c) sum += bred[c * 64 + lane];
      a.partial_bias[((long long)cg * NB + cob) * 64 + lane] = sum;
    }
  }
}

template <int C, int NTAPS, int CPW>
int launch_wgrad2(const WgCall& c, int bias_tap) {
  constexpr int NB = C / 64, NZ = NB * NB;
  const mdil_geom* g = c.g;
  const int nquads = g->N * g->HO * (g->WO >> 4);
  static const int wg2_cus = getenv("MDIL_WGRAD2_CUS") ? atoi(getenv("MDIL_WGRAD2_CUS")) : 256;   // tuning
  int ngroups = wg2_cus / NZ;                      // one work-group per CU
  const int need = cdiv(nquads, CPW);
  if (ngroups > need) ngroups = need;
  if (NZ > 1) ngroups = (ngroups + 7) / 8 * 8;     // chunk-group bits of blockIdx around the block bits
  const int qpc = cdiv(nquads, ngroups * CPW);
  const size_t need_ws = ((size_t)ngroups * NTAPS * NZ * 4096 + (size_t)ngroups * NB * 64) * sizeof(float);
  MDIL_CHECK_ARG(c.ws && c.ws_bytes >= need_ws, "wgrad: workspace %zu < %zu", c.ws_bytes, need_ws);
  wg2_args a;
  memset(&a, 0, sizeof(a));
  a.in0 = c.in0;
  a.in1 = c.in1;
  a.gout = c.gout;
  a.partial = (float*)c.ws;
  a.partial_bias = a.partial + (size_t)ngroups * NTAPS * NZ * 4096;
  a.N = g->N;
  a.H = g->HO;
  a.W = g->WO;
  for (int t = 0; t < NTAPS; ++t) {
    a.dh[t] = g->dh[t];
    a.dw[t] = g->dw[t];
    a.src[t] = g->src[t];
  }
  a.quads_per_chunk = qpc;
  const int want_bias = (c.dbias || c.dbias2) ? 1 : 0;
  a.bias_tap = want_bias ? bias_tap : -1;
  hipLaunchKernelGGL((wgrad2_kernel<C, NTAPS, CPW>), dim3(ngroups * NZ), dim3(NTAPS * CPW * 64), 0,
                     c.st, a);
  MDIL_CHECK_LAUNCH();
  RedArgs r;
  memset(&r, 0, sizeof(r));
  r.nchunks = ngroups;
  r.ntaps = NTAPS;
  r.nz = NZ;
  r.nz_ci = NB;
  r.CO = r.CI = C;
  r.CO_T = r.CI_T = 64;
  r.CO_P = r.CI_P = 64;
  for (int t = 0; t < NTAPS; ++t) r.ktap[t] = c.ktap[t];
  r.ntaps1 = NTAPS - c.ntaps2;
  r.s_co = c.s_co;
  r.s_ci = c.s_ci;
  r.s_co2 = c.s_co2;
  r.s_ci2 = c.s_ci2;
  r.accumulate = c.accumulate;
  launch_reduce(r, want_bias, a.partial, a.partial_bias, c.dw, c.dbias, c.dw2, c.dbias2, c.st, c.defer);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}

// -> tap index without a shift when the call is one the streaming kernel covers, else -1
int wgrad2_eligible(const mdil_geom* g, int cin, int cout, bool want_bias) {
  if (getenv("MDIL_NO_WGRAD2")) return -1;
  if (cin != cout || (cin != 64 && cin != 128)) return -1;
  if (g->ntaps != 3 && g->ntaps != 4) return -1;
  if (g->ihs != 1 || g->iws != 1 || g->ohs != 1 || g->ows != 1 || g->oho || g->owo ||
      g->HI != g->HO || g->WI != g->WO || g->OH != g->HO || g->OW != g->WO || g->out_coff ||
      g->out_pitch != cin || g->in_pitch[0] != cin || (g->WO & 15))
    return -1;
  if ((long long)g->N * g->HO * g->WO * cin * 4 >= (1ll << 31)) return -1;
  int center = -1;
  for (int t = 0; t < g->ntaps; ++t) {
    if (g->src[t] && g->in_pitch[1] != cin) return -1;
    if (g->dh[t] && g->dw[t]) return -1;            // one shift direction per tap
    if (!g->dh[t] && !g->dw[t] && center < 0) center = t;
  }
  if (want_bias && center < 0) return -1;
  return center < 0 ? 0 : center;
}

size_t wgrad2_ws(const mdil_geom* g, int cin) {
  const int NB = cin / 64, NZ = NB * NB;
  const int ngroups = (256 / NZ + 7) / 8 * 8;     // upper bound for every MDIL_WGRAD2_CUS <= 256
  return ((size_t)ngroups * g->ntaps * NZ * 4096 + (size_t)ngroups * NB * 64) * sizeof(float);
}

// ------------------------------------------------------------------------------------------------
// Streaming weight gradient of the 16 -> 16 channel 3-tap convs (the decoder's last two blocks,
// 786,432 pixels at config 3): HBM-bound -- 100 MB of x and g against 1.2 GFLOP.
// dW[t][co][ci] = sum_p g[p][co] x[p + off_t][ci] is one 16x16 MFMA tile per tap with K = pixels.
// A wave walks 16-pixel tiles: every lane loads 16 bytes (lane = (pixel, 4 channels): 1 KB
// contiguous per wave and tensor, the three taps' x tiles overlap and hit L1), the wave stages the
// four 16x16 tiles through 5 KB of its own LDS into operand order (rows = pixels) and issues 12
// MFMAs; no work-group barrier in the loop.  Taps that leave the image load from a clamped address
// and are zeroed by a select.  Partials go out in the layout of the LDS-tiled kernel,
// [chunk][t][16][16] + [chunk][16], so the reduction (and its deferral) is shared.  (The generic
// kernel ran these launches at 16 % of the MFMA and 26 % of the HBM rate.)
constexpr int WG16_T = 256;        // 4 waves per work-group
constexpr int WG16_LD = 20;        // LDS row stride (floats): 16-byte aligned rows

struct wg16_args {
  const float* x;
  const float* g;
  float* partial;        // [nchunks][3][16][16]
  float* partial_bias;   // [nchunks][16]
  int H, W;
  long long npix;
  int dh[3], dw[3];
  int tiles_per_wave;    // 16-pixel tiles each wave walks (consecutive)
};

__global__ __launch_bounds__(WG16_T) void wgrad16_kernel(const wg16_args a) {
  MDIL_HBM_KERNEL_PRIO();
  __shared__ __attribute__((aligned(16))) float stage[4][4 * 16 * WG16_LD];   // per wave: g, x(tap 0..2)
  __shared__ float red[4][3 * 256 + 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int px = lane >> 2, cq = lane & 3;          // load / staging role: pixel, channel quad
  const int ch = lane & 15, kk = lane >> 4;         // MFMA operand role: channel, pixel group
  const int H = a.H, W = a.W;
  float* st = stage[wave];
  f32x4 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  long long off[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) off[t] = ((long long)a.dh[t] * W + a.dw[t]) * 16;
  const long long tile0 = ((long long)blockIdx.x * 4 + wave) * a.tiles_per_wave;
  // operands of a tile: g and the three taps' x, 16 bytes per lane each; out-of-image taps and the
  // pixels past the end read as 0
  auto load_tile = [&](long long tile, f32x4& gv, f32x4 (&xv)[3]) __attribute__((always_inline)) {
    const long long p = tile * 16 + px;
    const bool valid = p < a.npix;
    const long long pc = valid ? p : 0;
    const int w = (int)(pc % W), h = (int)((pc / W) % H);
    gv = *reinterpret_cast<const f32x4*>(a.g + pc * 16 + cq * 4);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const bool in = valid && (unsigned)(h + a.dh[t]) < (unsigned)H && (unsigned)(w + a.dw[t]) < (unsigned)W;
      xv[t] = *reinterpret_cast<const f32x4*>(a.x + (in ? pc * 16 + off[t] : pc * 16) + cq * 4);
      if (!in) xv[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (!valid) gv = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  f32x4 gv, xv[3];
  load_tile(tile0, gv, xv);
  for (int k = 0; k < a.tiles_per_wave; ++k) {
    // the next tile's loads are in flight under this tile's staging and MFMAs
    f32x4 gn, xn[3];
    load_tile(tile0 + k + 1 < tile0 + a.tiles_per_wave ? tile0 + k + 1 : tile0 + k, gn, xn);
    bsum += gv;
    // LDS accesses of a wave execute in order; the fences only pin the compiler
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    *reinterpret_cast<f32x4*>(st + px * WG16_LD + cq * 4) = gv;
#pragma unroll
    for (int t = 0; t < 3; ++t) *reinterpret_cast<f32x4*>(st + (t + 1) * 16 * WG16_LD + px * WG16_LD + cq * 4) = xv[t];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float av = st[(4 * s + kk) * WG16_LD + ch];                       // A[co = ch][pixel 4s + kk]
#pragma unroll
      for (int t = 0; t < 3; ++t)
        acc[t] = mfma16(av, st[(t + 1) * 16 * WG16_LD + (4 * s + kk) * WG16_LD + ch], acc[t]);   // B[pixel][ci = ch]
    }
    gv = gn;
#pragma unroll
    for (int t = 0; t < 3; ++t) xv[t] = xn[t];
  }
  // bias: channel quad cq of the lane, summed over its 16 pixel lanes (lane bits 2..5)
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) bsum[c] += __shfl_xor(bsum[c], o, 64);
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][(t * 16 + 4 * kk + r) * 16 + ch] = acc[t][r];   // D[co = 4kk + r][ci = ch]
  if (lane < 4) {
#pragma unroll
    for (int c = 0; c < 4; ++c) red[wave][3 * 256 + lane * 4 + c] = bsum[c];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * 256 + 16; i += WG16_T) {
    const float v = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
    if (i < 3 * 256)
      a.partial[(long long)blockIdx.x * 3 * 256 + i] = v;
    else
      a.partial_bias[(long long)blockIdx.x * 16 + (i - 3 * 256)] = v;
  }
}

constexpr int WG16_CHUNKS = 1024;

bool wgrad16_eligible(const mdil_geom* g, int cin, int cout) {
  static const bool off = getenv("MDIL_NO_WGRAD16") != nullptr;
  if (off || cin != 16 || cout != 16 || g->ntaps != 3) return false;
  if (g->ihs != 1 || g->iws != 1 || g->ohs != 1 || g->ows != 1 || g->oho || g->owo || g->HI != g->HO ||
      g->WI != g->WO || g->OH != g->HO || g->OW != g->WO || g->out_coff || g->out_pitch != 16 ||
      g->in_pitch[0] != 16)
    return false;
  for (int t = 0; t < 3; ++t)
    if (g->src[t]) return false;
  return (long long)g->N * g->HO * g->WO * 16 * 4 < (1ll << 31);
}

size_t wgrad16_ws() { return (size_t)WG16_CHUNKS * (3 * 256 + 16) * sizeof(float); }

int launch_wgrad16(const WgCall& c) {
  const mdil_geom* g = c.g;
  const long long npix = (long long)g->N * g->HO * g->WO;
  const long long tiles = (npix + 15) / 16;
  int nchunks = (int)((tiles + 3) / 4);                               // >= one tile per wave
  if (nchunks > WG16_CHUNKS) nchunks = WG16_CHUNKS;
  if (nchunks < 1) nchunks = 1;
  MDIL_CHECK_ARG(c.ws && c.ws_bytes >= (size_t)nchunks * (3 * 256 + 16) * sizeof(float), "wgrad16: workspace");
  wg16_args a;
  memset(&a, 0, sizeof(a));
  a.x = c.in0;
  a.g = c.gout;
  a.partial = (float*)c.ws;
  a.partial_bias = a.partial + (size_t)nchunks * 3 * 256;
  a.H = g->HO;
  a.W = g->WO;
  a.npix = npix;
  for (int t = 0; t < 3; ++t) {
    a.dh[t] = g->dh[t];
    a.dw[t] = g->dw[t];
  }
  a.tiles_per_wave = (int)((tiles + (long long)nchunks * 4 - 1) / ((long long)nchunks * 4));
  hipLaunchKernelGGL(wgrad16_kernel, dim3(nchunks), dim3(WG16_T), 0, c.st, a);
  MDIL_CHECK_LAUNCH();
  RedArgs r;
  memset(&r, 0, sizeof(r));
  r.nchunks = nchunks;
  r.ntaps = 3;
  r.nz = r.nz_ci = 1;
  r.CO = r.CI = r.CO_T = r.CI_T = r.CO_P = r.CI_P = 16;
  for (int t = 0; t < 3; ++t) r.ktap[t] = c.ktap[t];
  r.ntaps1 = 3 - c.ntaps2;
  r.s_co = c.s_co;
  r.s_ci = c.s_ci;
  r.s_co2 = c.s_co2;
  r.s_ci2 = c.s_ci2;
  r.accumulate = c.accumulate;
  const int want_bias = (c.dbias || c.dbias2) ? 1 : 0;
  launch_reduce(r, want_bias, a.partial, a.partial_bias, c.dw, c.dbias, c.dw2, c.dbias2, c.st, c.defer);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}

// ------------------------------------------------------------------------------------------------
// Winograd F(2,3) weight gradient for the 3x1 (taps along H) convs: the transposed form of wconv.hip.
// For the output pair (p, p + d) along H with dy0 = g(p), dy1 = g(p + d) and d0..d3 = x(p - d),
// x(p), x(p + d), x(p + 2d):
//     M0 += dy0 (d0 - d2)     M1 += (dy0 + dy1)(d1 + d2)     M2 += (dy0 - dy1)(d2 - d1)
//     M3' += dy1 (d1 - d3)
//     dW[-d] = M0 + (M1 + M2)/2     dW[0] = (M1 - M2)/2     dW[+d] = (M1 + M2)/2 - M3'
// four 64 x 64 contractions over PAIRS of pixels instead of three over pixels: a third fewer MFMAs.
// Same machinery as wgrad2_kernel<C, 4, 2> (8 waves: 4 positions x 2 pixel chunks, scalar row
// descriptors, loads one quad ahead, no LDS in the loop); each operand is now the sum or difference
// of two image rows (2 x 4 VALU instructions per 16 MFMAs), rows outside the image are null
// descriptors, and the four M blocks are combined through LDS before the per-work-group partial
// is written in the usual [tap][64][64] layout -- the reduction kernel does not change.
// ------------------------------------------------------------------------------------------------
struct wgw_args {
  const float* x;
  const float* gout;
  float* partial;
  float* partial_bias;
  int N, H, W, delta;
  int tapidx[3];          // partial slot (geometry tap index) of the offsets -d, 0, +d
  int quads_per_chunk;    // pair-quads (16 pairs = 2 x 16 pixels of two rows) per wave
  int want_bias;
};

template <int C>
__global__ __launch_bounds__(512) void wgradw_kernel(const wgw_args a) {
  constexpr int NB = C / 64, NZ = NB * NB;
  constexpr int STR = C * 4;
  constexpr int CPW = 2, NPOS = 4;
  __shared__ __attribute__((aligned(16))) float red[NPOS * 4096 + 2 * CPW * 64];
  float* bred = red + NPOS * 4096;

  const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pos = wave % NPOS, cw = wave / NPOS;
  int z = 0, cg = blockIdx.x;
  if constexpr (NZ > 1) {
    z = (blockIdx.x >> 3) & (NZ - 1);
    cg = (blockIdx.x & 7) | ((blockIdx.x >> 5) << 3);
  }
  const int cob = z / NB, cib = z % NB;
  const int H = a.H, W = a.W, dl = a.delta;
  const int W16 = W >> 4, HP = H >> 1;
  const int npq = a.N * HP * W16;
  const int qpc = a.quads_per_chunk;
  const int q0 = (cg * CPW + cw) * qpc;
  const int q1 = min(q0 + qpc, npq);

  // rows (relative to the pair's first row h) and signs of this position's operands:
  //   A = g(h + ga) + sg * g(h + gb),   B = x(h + xa) + sx * x(h + xb);   a row offset of NONE = absent
  constexpr int NONE = 1 << 20;
  const int ga = pos == 3 ? NONE : 0, gb = pos == 0 ? NONE : dl;
  const float sg = pos == 2 ? -1.f : 1.f;
  const int xa = pos == 0 ? -dl : (pos == 2 ? dl : 0);
  const int xb = pos == 0 ? dl : (pos == 1 ? dl : (pos == 2 ? 0 : 2 * dl));
  const float sx = pos == 1 ? 1.f : -1.f;
  const bool do_bias = a.want_bias && cib == 0 && (pos == 0 || pos == 3);

  // prefetch pointer (scalar): pair-quad -> (image, pair row, column)
  int pq = q0;
  int w0 = (pq % W16) << 4;
  int pr = pq / W16;                       // img * HP + pair row
  int img = pr / HP;
  int rr = pr % HP;
  int hb = rr / dl, qq = rr % dl;
  __amdgpu_buffer_rsrc_t rga, rgb, rxa, rxb;
  auto rows = [&]() {
    const bool live = pq < q1;
    const int h = 2 * dl * hb + qq;
    auto desc = [&](const float* base, int off, int chan) {
      const int hh = h + off;
      const bool ok = live && off != NONE && hh >= 0 && hh < H;
      const long long o = ((long long)(img * H + (ok ? hh : 0)) * W) * C + chan * 64;
      return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base + (ok ? o : 0)), 0,
                                               ok ? (W - 1) * STR + 256 : 0, 0x00020000);
    };
    rga = desc(a.gout, ga, cob);
    rgb = desc(a.gout, gb, cob);
    rxa = desc(a.x, xa, cib);
    rxb = desc(a.x, xb, cib);
  };
  rows();
  const int voff = lg * STR + li * 16;

  f32x4 gqa[2][4], gqb[2][4], xqa[2][4], xqb[2][4];
  f32x4 acc[4][4];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[e][f] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

  constexpr int GI = 4096 / (4 * STR) > 0 ? 4096 / (4 * STR) : 1;
  auto ld = [&](const __amdgpu_buffer_rsrc_t r, int j) __attribute__((always_inline)) {
    const u32x4w v = __builtin_amdgcn_raw_buffer_load_b128(r, voff + (j % GI) * 4 * STR,
                                                           w0 * STR + (j / GI) * GI * 4 * STR, 0);
    return __builtin_bit_cast(f32x4, v);
  };
  auto advance = [&]() __attribute__((always_inline)) {
    ++pq;
    w0 += 16;
    if (w0 == W || pq >= q1) {
      if (w0 == W) {
        w0 = 0;
        if (++qq == dl) {
          qq = 0;
          if (++hb == H / (2 * dl)) {
            hb = 0;
            ++img;
          }
        }
      }
      rows();
    }
  };
  auto quad = [&](int s) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      gqa[s ^ 1][j] = ld(rga, j);
      gqb[s ^ 1][j] = ld(rgb, j);
      const f32x4 A = gqa[s][j] + gqb[s][j] * sg;
      const f32x4 B = xqa[s][j] + xqb[s][j] * sx;
#if WG2_PIN
      __builtin_amdgcn_sched_barrier(0);
#endif
      acc[0][0] = mfma16(A[0], B[0], acc[0][0]);
      acc[0][1] = mfma16(A[0], B[1], acc[0][1]);
#if WG2_PIN
      __builtin_amdgcn_sched_barrier(0);
#endif
      xqa[s ^ 1][j] = ld(rxa, j);
      xqb[s ^ 1][j] = ld(rxb, j);
#if WG2_PIN
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int k = 2; k < 16; ++k) acc[k >> 2][k & 3] = mfma16(A[k >> 2], B[k & 3], acc[k >> 2][k & 3]);
      if (do_bias) bsum += A;
#if WG2_PIN
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    advance();
  };

  if (q0 < q1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      gqa[0][j] = ld(rga, j);
      gqb[0][j] = ld(rgb, j);
      xqa[0][j] = ld(rxa, j);
      xqb[0][j] = ld(rxb, j);
    }
    advance();
    for (int q = q0; q < q1; q += 2) {
      quad(0);
      quad(1);   // an odd tail multiplies zeros: every descriptor is empty past q1
    }
  }

  // ---- chunk reduction, Winograd output transform (both through LDS, fixed order), partial ----
  auto put = [&](int slot) {
    float* d = red + slot * 4096;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int f = 0; f < 4; ++f)
        *reinterpret_cast<f32x4*>(&d[((e * 4 + f) * 64 + lane) * 4]) = acc[e][f];
  };
  auto get = [&](int slot, int e, int f) {
    return *reinterpret_cast<const f32x4*>(&red[slot * 4096 + ((e * 4 + f) * 64 + lane) * 4]);
  };
  if (do_bias) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      bsum[k] += __shfl_xor(bsum[k], 16, 64);
      bsum[k] += __shfl_xor(bsum[k], 32, 64);
    }
    if (lg == 0) *reinterpret_cast<f32x4*>(&bred[(cw * 2 + (pos == 3)) * 64 + li * 4]) = bsum;
  }
  if (cw == 1) put(pos);
  __syncthreads();
  if (cw == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int f = 0; f < 4; ++f) acc[e][f] += get(pos, e, f);
  }
  __syncthreads();
  if (cw == 0) put(pos);
  __syncthreads();
  if (cw == 0 && pos < 3) {
    // tap t = pos:  t = 0: M0 + (M1 + M2)/2,   t = 1: (M1 - M2)/2,   t = 2: (M1 + M2)/2 - M3'
    float* pout = a.partial + ((long long)(cg * 3 + a.tapidx[pos]) * NZ + z) * 4096;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f32x4 v[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const f32x4 m1 = get(1, e, f), m2 = get(2, e, f);
        if (pos == 0)
          v[f] = get(0, e, f) + (m1 + m2) * 0.5f;
        else if (pos == 1)
          v[f] = (m1 - m2) * 0.5f;
        else
          v[f] = (m1 + m2) * 0.5f - get(3, e, f);
      }
      // v[f][r] = dW[co = 4 (4 lg + r) + e][ci = 4 li + f]: a lane writes 4 consecutive ci
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x4 o = {v[0][r], v[1][r], v[2][r], v[3][r]};
        *reinterpret_cast<f32x4*>(&pout[(4 * (4 * lg + r) + e) * 64 + 4 * li]) = o;
      }
    }
  }
  if (a.want_bias && cib == 0 && wave == 0) {
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 2 * CPW; ++c) sum += bred[c * 64 + lane];
    a.partial_bias[((long long)cg * NB + cob) * 64 + lane] = sum;
  }
}

// The same for the 1x3 convs (taps along W).  A work item is now 16 pairs of ONE image row: the 32
// pixels [w0, w0 + 32); pair t of the quad starts at pixel px(t) = 2d (t / d) + t % d (d in
// {1, 2, 4, 8, 16}), its partner d further.  With t = 4 j + lg (MFMA k index lg, load group j) px
// separates into a scalar part G(j) and a lane part vpix(lg), so every operand is addressed as
//     row descriptor + soffset (w0 + G(j) + shift) * STR + voffset (vpix * STR + 16 li)
// with shift in {-d, 0, d, 2d}: the right image edge falls out of the descriptor's range check,
// the left edge (x(p - d) of a row's first pairs) gets per-lane offsets that are out of range.
template <int C>
__global__ __launch_bounds__(512) void wgradx_kernel(const wgw_args a) {
  constexpr int NB = C / 64, NZ = NB * NB;
  constexpr int STR = C * 4;
  constexpr int CPW = 2, NPOS = 4;
  __shared__ __attribute__((aligned(16))) float red[NPOS * 4096 + 2 * CPW * 64];
  float* bred = red + NPOS * 4096;

  const int lane = threadIdx.x & 63, li = lane & 15, lg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pos = wave % NPOS, cw = wave / NPOS;
  int z = 0, cg = blockIdx.x;
  if constexpr (NZ > 1) {
    z = (blockIdx.x >> 3) & (NZ - 1);
    cg = (blockIdx.x & 7) | ((blockIdx.x >> 5) << 3);
  }
  const int cob = z / NB, cib = z % NB;
  const int H = a.H, W = a.W, dl = a.delta;
  const int W32 = W >> 5;
  const int npq = a.N * H * W32;
  const int qpc = a.quads_per_chunk;
  const int q0 = (cg * CPW + cw) * qpc;
  const int q1 = min(q0 + qpc, npq);

  // shifts (pixels) and signs of this position's operands; NONE = operand absent
  constexpr int NONE = 1 << 20;
  const int ga = pos == 3 ? NONE : 0, gb = pos == 0 ? NONE : dl;
  const float sg = pos == 2 ? -1.f : 1.f;
  const int xa = pos == 0 ? -dl : (pos == 2 ? dl : 0);
  const int xb = pos == 0 ? dl : (pos == 1 ? dl : (pos == 2 ? 0 : 2 * dl));
  const float sx = pos == 1 ? 1.f : -1.f;
  const bool do_bias = a.want_bias && cib == 0 && (pos == 0 || pos == 3);

  auto px_of = [&](int t) { return 2 * dl * (t / dl) + t % dl; };
  int G[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) G[j] = __builtin_amdgcn_readfirstlane(px_of(4 * j));
  const int vpix = px_of(lg);
  const unsigned vbase = (unsigned)(vpix * STR + li * 16);
  // x(p - d) at the start of a row: per-lane offsets, out of range where the pixel does not exist
  unsigned vfirst[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int t = G[j] + vpix - dl;
    vfirst[j] = t < 0 ? 0x80000000u : (unsigned)(t * STR + li * 16);
  }

  int pq = q0;
  int w0 = (pq % W32) << 5;
  int row = pq / W32;                      // img * H + h
  __amdgpu_buffer_rsrc_t rg, rx;
  const __amdgpu_buffer_rsrc_t rnull =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, 0, 0x00020000);
  auto rows = [&]() {
    const bool live = pq < q1;
    const long long o = (long long)(live ? row : 0) * W * C;
    rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gout + o + cob * 64), 0,
                                           live ? (W - 1) * STR + 256 : 0, 0x00020000);
    rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + o + cib * 64), 0,
                                           live ? (W - 1) * STR + 256 : 0, 0x00020000);
  };
  rows();

  f32x4 gqa[2][4], gqb[2][4], xqa[2][4], xqb[2][4];
  f32x4 acc[4][4];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int f = 0; f < 4; ++f) acc[e][f] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

  auto ld = [&](const __amdgpu_buffer_rsrc_t r, int sh, int j) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t rr = sh == NONE ? rnull : r;
    const int s = w0 + G[j] + sh;                       // scalar pixel position (may be < 0 only for sh = -d)
    const bool first = sh < 0 && s < 0;                 // row start: per-lane validity
    const unsigned vo = first ? vfirst[j] : vbase;
    const int so = (first || sh == NONE) ? 0 : s * STR;
    const u32x4w v = __builtin_amdgcn_raw_buffer_load_b128(rr, (int)vo, so, 0);
    return __builtin_bit_cast(f32x4, v);
  };
  auto advance = [&]() __attribute__((always_inline)) {
    ++pq;
    w0 += 32;
    if (w0 == W || pq >= q1) {
      if (w0 == W) {
        w0 = 0;
        ++row;
      }
      rows();
    }
  };
  auto quad = [&](int s) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      gqa[s ^ 1][j] = ld(rg, ga, j);
      gqb[s ^ 1][j] = ld(rg, gb, j);
      const f32x4 A = gqa[s][j] + gqb[s][j] * sg;
      const f32x4 B = xqa[s][j] + xqb[s][j] * sx;
#if WG2_PIN
      __builtin_amdgcn_sched_barrier(0);
#endif
      acc[0][0] = mfma16(A[0], B[0], acc[0][0]);
      acc[0][1] = mfma16(A[0], B[1], acc[0][1]);
#if WG2_PIN
      __builtin_amdgcn_sched_barrier(0);
#endif
      xqa[s ^ 1][j] = ld(rx, xa, j);
      xqb[s ^ 1][j] = ld(rx, xb, j);
#if WG2_PIN
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int k = 2; k < 16; ++k) acc[k >> 2][k & 3] = mfma16(A[k >> 2], B[k & 3], acc[k >> 2][k & 3]);
      if (do_bias) bsum += A;
#if WG2_PIN
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    advance();
  };

  if (q0 < q1) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      gqa[0][j] = ld(rg, ga, j);
      gqb[0][j] = ld(rg, gb, j);
      xqa[0][j] = ld(rx, xa, j);
      xqb[0][j] = ld(rx, xb, j);
    }
    advance();
    for (int q = q0; q < q1; q += 2) {
      quad(0);
      quad(1);
    }
  }

  auto put = [&](int slot) {
    float* d = red + slot * 4096;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int f = 0; f < 4; ++f)
        *reinterpret_cast<f32x4*>(&d[((e * 4 + f) * 64 + lane) * 4]) = acc[e][f];
  };
  auto get = [&](int slot, int e, int f) {
    return *reinterpret_cast<const f32x4*>(&red[slot * 4096 + ((e * 4 + f) * 64 + lane) * 4]);
  };
  if (do_bias) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      bsum[k] += __shfl_xor(bsum[k], 16, 64);
      bsum[k] += __shfl_xor(bsum[k], 32, 64);
    }
    if (lg == 0) *reinterpret_cast<f32x4*>(&bred[(cw * 2 + (pos == 3)) * 64 + li * 4]) = bsum;
  }
  if (cw == 1) put(pos);
  __syncthreads();
  if (cw == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int f = 0; f < 4; ++f) acc[e][f] += get(pos, e, f);
  }
  __syncthreads();
  if (cw == 0) put(pos);
  __syncthreads();
  if (cw == 0 && pos < 3) {
    float* pout = a.partial + ((long long)(cg * 3 + a.tapidx[pos]) * NZ + z) * 4096;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f32x4 v[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const f32x4 m1 = get(1, e, f), m2 = get(2, e, f);
        if (pos == 0)
          v[f] = get(0, e, f) + (m1 + m2) * 0.5f;
        else if (pos == 1)
          v[f] = (m1 - m2) * 0.5f;
        else
          v[f] = (m1 + m2) * 0.5f - get(3, e, f);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x4 o = {v[0][r], v[1][r], v[2][r], v[3][r]};
        *reinterpret_cast<f32x4*>(&pout[(4 * (4 * lg + r) + e) * 64 + 4 * li]) = o;
      }
    }
  }
  if (a.want_bias && cib == 0 && wave == 0) {
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 2 * CPW; ++c) sum += bred[c * 64 + lane];
    a.partial_bias[((long long)cg * NB + cob) * 64 + lane] = sum;
  }
}

// 3 taps along one axis with dilation d from one source, complete pairs: -> d (else 0); *axis = 1
// for taps along W (those need W % 32 == 0 and d in {1, 2, 4, 8, 16})
int wgradw_eligible(const mdil_geom* g, int cin, int cout, int* tapidx, int* axis) {
  static const bool off = getenv("MDIL_NO_WGRADW") != nullptr || getenv("MDIL_NO_WGRAD2") != nullptr;
  static const bool offx = getenv("MDIL_NO_WGRADX") != nullptr;
  if (off || g->ntaps != 3 || wgrad2_eligible(g, cin, cout, false) < 0) return 0;
  int d = 0, ax = -1, idx[3] = {-1, -1, -1};
  for (int t = 0; t < 3; ++t) {
    if (g->src[t] != g->src[0] || (g->dh[t] && g->dw[t])) return 0;
    const int o = g->dh[t] ? g->dh[t] : g->dw[t];
    if (o == 0) {
      idx[1] = t;
      continue;
    }
    const int a_ = g->dw[t] ? 1 : 0;
    if (ax >= 0 && a_ != ax) return 0;
    ax = a_;
    const int ad = o < 0 ? -o : o;
    if (d && ad != d) return 0;
    d = ad;
    idx[o < 0 ? 0 : 2] = t;
  }
  if (d <= 0 || idx[0] < 0 || idx[1] < 0 || idx[2] < 0) return 0;
  if (ax == 0) {
    if (g->HO % (2 * d)) return 0;
  } else {
    if (offx || (g->WO & 31) || !(d == 1 || d == 2 || d == 4 || d == 8 || d == 16)) return 0;
  }
  if (tapidx)
    for (int t = 0; t < 3; ++t) tapidx[t] = idx[t];
  if (axis) *axis = ax;
  return d;
}

template <int C>
int launch_wgradw(const WgCall& c, int delta, const int* tapidx, int axis) {
  constexpr int NB = C / 64, NZ = NB * NB, CPW = 2;
  const mdil_geom* g = c.g;
  const int npq = axis ? g->N * g->HO * (g->WO >> 5) : g->N * (g->HO >> 1) * (g->WO >> 4);
  int ngroups = 256 / NZ;
  const int need = cdiv(npq, CPW);
  if (ngroups > need) ngroups = need;
  if (NZ > 1) ngroups = (ngroups + 7) / 8 * 8;
  const int qpc = cdiv(npq, ngroups * CPW);
  const size_t need_ws = ((size_t)ngroups * 3 * NZ * 4096 + (size_t)ngroups * NB * 64) * sizeof(float);
  MDIL_CHECK_ARG(c.ws && c.ws_bytes >= need_ws, "wgrad: workspace %zu < %zu", c.ws_bytes, need_ws);
  wgw_args a;
  memset(&a, 0, sizeof(a));
  a.x = g->src[0] ? c.in1 : c.in0;
  a.gout = c.gout;
  a.partial = (float*)c.ws;
  a.partial_bias = a.partial + (size_t)ngroups * 3 * NZ * 4096;
  a.N = g->N;
  a.H = g->HO;
  a.W = g->WO;
  a.delta = delta;
  for (int t = 0; t < 3; ++t) a.tapidx[t] = tapidx[t];
  a.quads_per_chunk = qpc;
  const int want_bias = (c.dbias || c.dbias2) ? 1 : 0;
  a.want_bias = want_bias;
  if (axis)
    hipLaunchKernelGGL((wgradx_kernel<C>), dim3(ngroups * NZ), dim3(512), 0, c.st, a);
  else
    hipLaunchKernelGGL((wgradw_kernel<C>), dim3(ngroups * NZ), dim3(512), 0, c.st, a);
  MDIL_CHECK_LAUNCH();
  RedArgs r;
  memset(&r, 0, sizeof(r));
  r.nchunks = ngroups;
  r.ntaps = 3;
  r.nz = NZ;
  r.nz_ci = NB;
  r.CO = r.CI = C;
  r.CO_T = r.CI_T = 64;
  r.CO_P = r.CI_P = 64;
  for (int t = 0; t < 3; ++t) r.ktap[t] = c.ktap[t];
  r.ntaps1 = 3 - c.ntaps2;
  r.s_co = c.s_co;
  r.s_ci = c.s_ci;
  r.s_co2 = c.s_co2;
  r.s_ci2 = c.s_ci2;
  r.accumulate = c.accumulate;
  launch_reduce(r, want_bias, a.partial, a.partial_bias, c.dw, c.dbias, c.dw2, c.dbias2, c.st, c.defer);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}

}  // namespace

// (cout, cin) of the conv -> compiled tile configuration
#define WG_CONFIGS(X)      \
  X(64, 64, 64, 64, false)     \
  X(128, 128, 64, 64, false)   \
  X(64, 128, 64, 64, false)    \
  X(16, 16, 16, 16, false)     \
  X(48, 16, 48, 16, false)     \
  X(16, 64, 16, 64, false)     \
  X(20, 16, 20, 16, false)     \
  X(27, 16, 28, 16, false)     \
  X(13, 27, 13, 27, true)

extern "C" size_t mdil_wgrad_workspace(const mdil_geom* g, int cin, int cout) {
  size_t w2 = wgrad2_eligible(g, cin, cout, false) >= 0 ? wgrad2_ws(g, cin) : 0;
  if (wgrad16_eligible(g, cin, cout)) w2 = max2(w2, wgrad16_ws());
#define X(co, ci, cot, cit, stem) \
  if (cout == co && cin == ci) return max2(w2, ws_need<cot, cit, stem>(g, co, ci));
  WG_CONFIGS(X)
#undef X
  return 0;
}

static int wgrad_impl(const mdil_geom* g, int cin, int cout, const float* in0, const float* in1,
                      const float* gout, const int* ktap, int s_co, int s_ci, float* dw,
                      float* dbias, int ntaps2, int s_co2, int s_ci2, float* dw2, float* dbias2,
                      int accumulate, void* workspace, size_t workspace_bytes, mdil_wgrad_job* defer,
                      void* stream);

extern "C" int mdil_wgrad(const mdil_geom* g, int cin, int cout, const float* in0,
                          const float* in1, const float* gout, const int* ktap, int s_co, int s_ci,
                          float* dw, float* dbias, int ntaps2, int s_co2, int s_ci2, float* dw2,
                          float* dbias2, int accumulate, void* workspace, size_t workspace_bytes,
                          void* stream) {
  return wgrad_impl(g, cin, cout, in0, in1, gout, ktap, s_co, s_ci, dw, dbias, ntaps2, s_co2, s_ci2,
                    dw2, dbias2, accumulate, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int mdil_wgrad_deferred(const mdil_geom* g, int cin, int cout, const float* in0,
                                   const float* in1, const float* gout, const int* ktap, int s_co,
                                   int s_ci, float* dw, float* dbias, int ntaps2, int s_co2,
                                   int s_ci2, float* dw2, float* dbias2, void* workspace,
                                   size_t workspace_bytes, mdil_wgrad_job* job, void* stream) {
  MDIL_CHECK_ARG(job != nullptr, "wgrad_deferred: job record missing");
  MDIL_CHECK_ARG(cin % 4 == 0 && cout % 4 == 0, "wgrad_deferred: channel counts must be multiples of 4");
  return wgrad_impl(g, cin, cout, in0, in1, gout, ktap, s_co, s_ci, dw, dbias, ntaps2, s_co2, s_ci2,
                    dw2, dbias2, 1, workspace, workspace_bytes, job, stream);
}

extern "C" int mdil_wgrad_reduce_batch(const mdil_wgrad_job* jobs, int njobs, void* stream) {
  MDIL_CHECK_ARG(njobs >= 0 && (njobs == 0 || jobs), "wgrad_reduce_batch: bad argument");
  for (int j0 = 0; j0 < njobs; j0 += RED_BATCH) {
    RedBatch b;
    memset(&b, 0, sizeof(b));
    b.njobs = njobs - j0 < RED_BATCH ? njobs - j0 : RED_BATCH;
    int total = 0;
    for (int k = 0; k < b.njobs; ++k) {
      memcpy(&b.jobs[k], &jobs[j0 + k], sizeof(RedJob));
      MDIL_CHECK_ARG(b.jobs[k].nblk > 0 && b.jobs[k].partial && b.jobs[k].dw,
                     "wgrad_reduce_batch: job %d is not a record written by mdil_wgrad_deferred", j0 + k);
      b.start[k] = total;
      total += b.jobs[k].nblk;
    }
    b.start[b.njobs] = total;
    hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, b);
    MDIL_CHECK_LAUNCH();
  }
  return MDIL_OK;
}

static int wgrad_impl(const mdil_geom* g, int cin, int cout, const float* in0, const float* in1,
                      const float* gout, const int* ktap, int s_co, int s_ci, float* dw,
                      float* dbias, int ntaps2, int s_co2, int s_ci2, float* dw2, float* dbias2,
                      int accumulate, void* workspace, size_t workspace_bytes, mdil_wgrad_job* defer,
                      void* stream) {
  MDIL_CHECK_ARG(g && in0 && gout && dw, "wgrad: null argument");
  MDIL_CHECK_ARG(g->ntaps >= 1 && g->ntaps <= MDIL_MAX_TAPS, "wgrad: ntaps=%d", g->ntaps);
  MDIL_CHECK_ARG(cin == 27 || ktap, "wgrad: ktap missing");
  MDIL_CHECK_ARG(ntaps2 >= 0 && ntaps2 < g->ntaps && (ntaps2 == 0 || dw2), "wgrad: second target");
  for (int t = 0; t < g->ntaps; ++t)
    MDIL_CHECK_ARG(g->src[t] == 0 || (g->src[t] == 1 && in1), "wgrad: tap %d source", t);
  WgCall c{g, in0, in1, gout, ktap, s_co, s_ci, dw, dbias, ntaps2, s_co2, s_ci2, dw2, dbias2,
           accumulate, workspace, workspace_bytes, (hipStream_t)stream, cout, cin, defer};
  MdilProfScope ps((hipStream_t)stream, 1, g, cin, cout);
  if (wgrad16_eligible(g, cin, cout)) {
    ps.path = 1;
    return launch_wgrad16(c);
  }
  {
    const int bt = wgrad2_eligible(g, cin, cout, dbias || dbias2);
    if (bt >= 0) {
      int tapidx[3], axis = 0;
      ps.path = 1;
      if (const int d = wgradw_eligible(g, cin, cout, tapidx, &axis)) {    // 3-tap convs: Winograd form
        ps.path = 2;
        return cin == 64 ? launch_wgradw<64>(c, d, tapidx, axis) : launch_wgradw<128>(c, d, tapidx, axis);
      }
      if (cin == 64) return g->ntaps == 3 ? launch_wgrad2<64, 3, 2>(c, bt) : launch_wgrad2<64, 4, 2>(c, bt);
      return g->ntaps == 3 ? launch_wgrad2<128, 3, 4>(c, bt) : launch_wgrad2<128, 4, 2>(c, bt);
    }
  }
#define X(co, ci, cot, cit, stem) \
  if (cout == co && cin == ci) return launch_wgrad<cot, cit, stem>(c);
  WG_CONFIGS(X)
#undef X
  mdil_set_error("wgrad: no tile configuration for cin=%d cout=%d", cin, cout);
  return MDIL_ERR_UNSUPPORTED;
}
