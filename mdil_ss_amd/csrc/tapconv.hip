// Generic MFMA "tap convolution" for NHWC fp32 tensors on gfx950.
//
//   out[p][co] = epi( sum_t sum_ci in_t[p (+) tap t][ci] * W[t][co][ci] )
//
// One workgroup (4 waves) owns BM consecutive pixels of the flattened (n,ho,wo) iteration grid
// and ALL output channels.  Per (tap, K-chunk) stage the input tile [BM][KC] and the weight
// tile [COUT_P][KC] are staged through LDS (registers -> ds_write_b128, rows padded by 16 B so
// the 8 lanes that write one row and the 16-lane ds_read_b128 groups spread over the banks);
// the next stage's global loads are issued before the current stage's MFMAs (issue-early /
// write-late).  The contraction runs on v_mfma_f32_16x16x4_f32 with M = output channel,
// N = pixel, so every lane ends up with 4 consecutive output channels of one pixel -> one
// 16-byte NHWC store, and the epilogue tensors (residual, gates) are read the same way.
//
// K ordering trick: one ds_read_b128 gives a lane 4 consecutive input channels; lane group
// g = lane>>4 owns channels [16r+4g, 16r+4g+4) of round r and feeds element s to MFMA s, i.e.
// MFMA s contracts channels {16r+4g+s : g=0..3}.  A and B use the same map, so the sum over
// K is complete and each operand needs a single LDS read per 4 MFMAs.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "bnfin.h"

#ifndef TC_TIMING
#define TC_TIMING 0   // tuning builds only: per-workgroup wall-clock stamps into a debug buffer
#endif
#if TC_TIMING
__device__ unsigned long long* tc_stamps = nullptr;
extern "C" int mdil_debug_set_stamps(void* p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(tc_stamps), &p, sizeof(p));
}
#define TC_STAMP(k)                                                             \
  do {                                                                          \
    if (threadIdx.x == 0 && tc_stamps) tc_stamps[blockIdx.x * 4 + (k)] = wall_clock64(); \
  } while (0)
#else
#define TC_STAMP(k)
#endif

namespace {

template <int CIN, int COUT, int BM_, bool STEM_>
struct TapCfg {
  static constexpr bool STEM = STEM_;
  static constexpr int BM = BM_;
  static constexpr int CIN_P = STEM ? 32 : ((CIN + 15) / 16 * 16);
#ifndef TC_KC_BIG
#define TC_KC_BIG 32
#endif
  static constexpr int KC = (CIN_P <= 48) ? CIN_P : TC_KC_BIG;   // K-chunk staged per barrier pair
  static constexpr int NCHUNK = CIN_P / KC;
  static constexpr int QPR = KC / 4;  // float4 per LDS row
  static constexpr int LD = KC + 4;   // LDS row stride (floats)
  static constexpr int MT = (COUT + 15) / 16;
  static constexpr int COUT_P = MT * 16;
  static constexpr int NT = BM / 16;
  static constexpr bool SPLIT_CO = (MT % 4 == 0);
  static constexpr int TM = SPLIT_CO ? MT / 4 : MT;
  static constexpr int TN = SPLIT_CO ? NT : NT / 4;
  static constexpr int IN_ITEMS = BM * QPR / MDIL_WG;
  static constexpr int W_ITEMS = (COUT_P * QPR + MDIL_WG - 1) / MDIL_WG;
  static_assert(CIN_P % KC == 0, "chunking");
  static_assert((BM * QPR) % MDIL_WG == 0, "in-tile items");
  static_assert(SPLIT_CO || (NT % 4 == 0), "pixel tiles per wave");
};

template <int CIN, int COUT, int BM, bool STEM>
__global__ __launch_bounds__(MDIL_WG) void tapconv_kernel(const mdil_geom g,
                                                          const float* __restrict__ in0,
                                                          const float* __restrict__ in1,
                                                          const float* __restrict__ wpk,
                                                          const mdil_epilogue e,
                                                          float* __restrict__ out) {
  using C = TapCfg<CIN, COUT, BM, STEM>;
#ifndef TC_LDS_MIN
#define TC_LDS_MIN 0
#endif
  constexpr int SMEM_STAGE = (BM + C::COUT_P) * C::LD;
  // output tile [BM][COUT_P + 4] staged through LDS so the epilogue's global traffic is fully
  // coalesced (plus BM 8-byte output offsets)
  constexpr int LDO = C::COUT_P + 4;
  constexpr int SMEM_OUT = (COUT % 4 == 0) ? BM * LDO + 2 * BM : 0;
  constexpr int SMEM_F = SMEM_STAGE > SMEM_OUT ? SMEM_STAGE : SMEM_OUT;
  __shared__ __attribute__((aligned(16))) float smem[SMEM_F > TC_LDS_MIN ? SMEM_F : TC_LDS_MIN];
  float* Is = smem;
  float* Ws = smem + BM * C::LD;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int npix = g.N * g.HO * g.WO;
  const int hw = g.HO * g.WO;
  const int tile0 = blockIdx.x * BM;

  // ---- per-thread staging items (fixed across stages) ----
  int it_h[C::IN_ITEMS], it_w[C::IN_ITEMS];
  long long it_base[C::IN_ITEMS];   // (n*HI + h)*WI + w of the tile pixel in the input plane
  if constexpr (!STEM) {
#pragma unroll
    for (int i = 0; i < C::IN_ITEMS; ++i) {
      const int idx = tid + MDIL_WG * i;
      const int p = idx / C::QPR;
      const int P = tile0 + p;
      if (P < npix) {
        const int n = P / hw;
        const int r = P - n * hw;
        const int ho = r / g.WO;
        const int wo = r - ho * g.WO;
        it_h[i] = ho * g.ihs;
        it_w[i] = wo * g.iws;
        it_base[i] = ((long long)n * g.HI + it_h[i]) * g.WI + it_w[i];
      } else {
        it_h[i] = -(1 << 28);  // forces the bounds test to fail
        it_w[i] = 0;
        it_base[i] = 0;
      }
    }
  }

  // Prefetch distance.  With two register sets the global loads run TWO stages ahead of their
  // LDS write.  Measured (MI355X, full-size shapes): +6 % on the 64-channel launches (6-8 short
  // stages: one stage of MFMAs does not cover the L2/HBM latency), -3 % on the 128-channel ones
  // (12-16 stages, already covered; the extra 16 VGPRs only cost occupancy) -> on for CIN_P == 64.
#ifndef TC_PF2
#define TC_PF2 -1     // -1 auto, 0 off, 1 on for every config
#endif
  constexpr bool PF2 = TC_PF2 < 0 ? (C::CIN_P == 64) : (TC_PF2 != 0);
  constexpr int NSET = (PF2 && !STEM) ? 2 : 1;
  f32x4 regIs[NSET][C::IN_ITEMS];
  f32x4 regWs[NSET][C::W_ITEMS];
  unsigned okIs[NSET] = {};

  // NOTE: every load below is UNCONDITIONAL (clamped address, select afterwards).  A load under
  // `if (ok)` makes hipcc branch around it and drain vmcnt per element, which serialises the
  // stage's global loads behind each other's latency.
  auto issue_loads = [&](auto SET, int t, int kc) {
    f32x4 (&regI)[C::IN_ITEMS] = regIs[decltype(SET)::value];
    f32x4 (&regW)[C::W_ITEMS] = regWs[decltype(SET)::value];
    unsigned& okI = okIs[decltype(SET)::value];
    if constexpr (!STEM) {
      const int s = g.src[t];
      const float* __restrict__ src = s ? in1 : in0;
      const int pitch = g.in_pitch[s];
      const int dh = g.dh[t], dw = g.dw[t];
      // wave-uniform part of the address: tap shift + K-chunk (scalar registers); the per-pixel
      // part is fixed for the tile.  One 64-bit add per load, no divergent region around it.
      const long long tap_off = ((long long)dh * g.WI + dw) * pitch + kc * C::KC;
#pragma unroll
      for (int i = 0; i < C::IN_ITEMS; ++i) {
        const int idx = tid + MDIL_WG * i;
        const int q = idx % C::QPR;
        const int hi = it_h[i] + dh, wi = it_w[i] + dw;
        const int k = kc * C::KC + q * 4;
        const bool ok = (hi >= 0) && (hi < g.HI) && (wi >= 0) && (wi < g.WI) && (k < CIN);
        long long off = it_base[i] * pitch + (q * 4 + tap_off);
        off = ok ? off : 0ll;
        regI[i] = *reinterpret_cast<const f32x4*>(src + off);
        okI = ok ? (okI | (1u << i)) : (okI & ~(1u << i));   // zero-fill is applied at write time,
      }                                                      // so nothing waits on the load here
    }
#pragma unroll
    for (int i = 0; i < C::W_ITEMS; ++i) {
      int idx = tid + MDIL_WG * i;
      if constexpr ((C::COUT_P * C::QPR) % MDIL_WG != 0) idx = idx < C::COUT_P * C::QPR ? idx : 0;
      const int co = idx / C::QPR, q = idx % C::QPR;
      regW[i] = *reinterpret_cast<const f32x4*>(wpk + ((long long)(t * C::COUT_P + co)) * C::CIN_P +
                                                kc * C::KC + q * 4);
    }
  };

  auto write_lds = [&](auto SET) {
    f32x4 (&regI)[C::IN_ITEMS] = regIs[decltype(SET)::value];
    f32x4 (&regW)[C::W_ITEMS] = regWs[decltype(SET)::value];
    const unsigned okI = okIs[decltype(SET)::value];
    if constexpr (!STEM) {
#pragma unroll
      for (int i = 0; i < C::IN_ITEMS; ++i) {
        const int idx = tid + MDIL_WG * i;
        const int p = idx / C::QPR, q = idx % C::QPR;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(&Is[p * C::LD + q * 4]) = ((okI >> i) & 1u) ? regI[i] : z;
      }
    }
#pragma unroll
    for (int i = 0; i < C::W_ITEMS; ++i) {
      const int idx = tid + MDIL_WG * i;
      if (((C::COUT_P * C::QPR) % MDIL_WG == 0) || idx < C::COUT_P * C::QPR) {
        const int co = idx / C::QPR, q = idx % C::QPR;
        *reinterpret_cast<f32x4*>(&Ws[co * C::LD + q * 4]) = regW[i];
      }
    }
  };

  f32x4 acc[C::TM][C::TN];
#pragma unroll
  for (int a = 0; a < C::TM; ++a)
#pragma unroll
    for (int b = 0; b < C::TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int co_tile0 = C::SPLIT_CO ? wave * C::TM : 0;
  const int px_tile0 = C::SPLIT_CO ? 0 : wave * C::TN;

  if constexpr (STEM) {
    // im2col-on-load of the 3x3 stride-2 RGB stem: row = [9 taps][3 ch] + 5 zero columns.
    for (int idx = tid; idx < BM * 9; idx += MDIL_WG) {
      const int p = idx / 9, tap = idx - p * 9;
      const int P = tile0 + p;
      float v0 = 0.f, v1 = 0.f, v2 = 0.f;
      if (P < npix) {
        const int n = P / hw;
        const int r = P - n * hw;
        const int ho = r / g.WO, wo = r - (r / g.WO) * g.WO;
        const int hi = 2 * ho + tap / 3 - 1, wi = 2 * wo + tap % 3 - 1;
        if (hi >= 0 && hi < g.HI && wi >= 0 && wi < g.WI) {
          const float* s = in0 + ((long long)(n * g.HI + hi) * g.WI + wi) * 3;
          v0 = s[0];
          v1 = s[1];
          v2 = s[2];
        }
      }
      float* d = &Is[p * C::LD + 3 * tap];
      d[0] = v0;
      d[1] = v1;
      d[2] = v2;
    }
    for (int idx = tid; idx < BM * 5; idx += MDIL_WG) Is[(idx / 5) * C::LD + 27 + idx % 5] = 0.f;
  }

// non-temporal output stores: the tile is not re-read by this kernel and leaving 25-50 MB dirty in
// L2 lengthens the end-of-kernel write-back (measured -3..5 % per launch)
#ifndef TC_NT_STORE
#define TC_NT_STORE 1
#endif
#ifndef TC_ABLATE
#define TC_ABLATE 0   // tuning builds only: 1 = no global loads after stage 0, 2 = also no LDS
#endif                // writes / barriers after stage 0 (results are then wrong by construction)
  const int nstage = g.ntaps * C::NCHUNK;
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, NSET - 1>;
  TC_STAMP(0);
#if defined(TC_STAGGER) && TC_STAGGER > 0
  // tuning builds only: de-phase the work-groups that share a CU (measured: strictly slower)
  {
    const int slot = (blockIdx.x >> 8) % 3;
    for (int i = 0; i < slot * TC_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
  }
#endif
  auto mfma_stage = [&]() {
#ifndef TC_ROUND_UNROLL
#define TC_ROUND_UNROLL 1
#endif
#pragma unroll TC_ROUND_UNROLL
    for (int r = 0; r < C::KC / 16; ++r) {
      f32x4 a[C::TM], b[C::TN];
#pragma unroll
      for (int m = 0; m < C::TM; ++m)
        a[m] = *reinterpret_cast<const f32x4*>(&Ws[((co_tile0 + m) * 16 + li) * C::LD + r * 16 + lg * 4]);
#pragma unroll
      for (int n = 0; n < C::TN; ++n)
        b[n] = *reinterpret_cast<const f32x4*>(&Is[((px_tile0 + n) * 16 + li) * C::LD + r * 16 + lg * 4]);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int m = 0; m < C::TM; ++m)
#pragma unroll
          for (int n = 0; n < C::TN; ++n) acc[m][n] = mfma16(a[m][s], b[n][s], acc[m][n]);
    }
  };
  constexpr int DIST = NSET;            // how many stages ahead the global loads run
  // one stage: LDS hand-over of register set SET, refill of the same set for stage st + DIST
  // (in flight under this and, with two sets, the next stage's MFMAs), then the MFMAs
  auto stage = [&](auto SET, int st) {
    if (st == 1) TC_STAMP(1);
    if (TC_ABLATE < 2 || st == 0) {
      __syncthreads();  // everyone finished reading the previous stage
      write_lds(SET);
      __syncthreads();
    }
    if (st + DIST < nstage && (TC_ABLATE == 0)) {
      const int nx = st + DIST;
      issue_loads(SET, nx / C::NCHUNK, nx % C::NCHUNK);
    }
    // waves in their MFMA phase win issue arbitration over waves (of the co-resident work-groups)
    // that are staging: +3-4 % per launch in isolation (C=128 3 taps 54.4 -> 52.9 us, C=64
    // 62.8 -> 61.1 us), neutral on the 3-stream step
#ifndef TC_SETPRIO
#define TC_SETPRIO 1
#endif
#if TC_SETPRIO
    __builtin_amdgcn_s_setprio(1);
#endif
    mfma_stage();
#if TC_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
  };
  issue_loads(Set0{}, 0, 0);
  if constexpr (NSET == 2) {
    if (nstage > 1) issue_loads(Set1{}, 1 / C::NCHUNK, 1 % C::NCHUNK);
    for (int st = 0; st < nstage; st += 2) {
      stage(Set0{}, st);
      if (st + 1 < nstage) stage(Set1{}, st + 1);
    }
  } else {
    for (int st = 0; st < nstage; ++st) stage(Set0{}, st);
  }

  TC_STAMP(2);
  // ---- epilogue ----
  // lane holds acc = out[pixel = tile pixel li][co = 16*mt + 4*lg .. +3]: written straight to
  // global that is 16 pixels x 64 B per instruction (half lines, store-issue bound: measured
  // 14 us of a 58 us launch).  Instead the tile is transposed through LDS and written out with
  // consecutive lanes on consecutive 16-byte pieces of a pixel's channel row (1 KiB contiguous per
  // wave-instruction at C=128); residual / gate tensors are read with the same pattern.
  if constexpr (COUT % 4 == 0) {
    float* Os = smem;
    long long* Ob = reinterpret_cast<long long*>(smem + BM * LDO);
    // a thread writes the same 4 output channels in every iteration of the store loop when the
    // work-group size is a multiple of the pieces per pixel: fetch their bias / scale / shift
    // once, now, so the latency hides under the transpose (it used to be one dependent global
    // round trip per iteration).  v*scale+bias form: bias-only -> scale 1; folded BN -> scale,
    // shift (+ bias*scale folded in).
    constexpr bool QINV = (MDIL_WG % (COUT / 4)) == 0;
    f32x4 vscale = {1.f, 1.f, 1.f, 1.f}, vbias = {0.f, 0.f, 0.f, 0.f};
    if constexpr (QINV) {
      const int co0 = (tid % (COUT / 4)) * 4;
      if (e.bias) vbias = *reinterpret_cast<const f32x4*>(e.bias + co0);
      if (e.bias2) vbias += *reinterpret_cast<const f32x4*>(e.bias2 + co0);
      if (e.scale) {
        vscale = *reinterpret_cast<const f32x4*>(e.scale + co0);
        vbias = vbias * vscale + *reinterpret_cast<const f32x4*>(e.shift + co0);
      }
    }
    __syncthreads();  // all MFMA operand reads of the staging tiles are done
#pragma unroll
    for (int n = 0; n < C::TN; ++n)
#pragma unroll
      for (int m = 0; m < C::TM; ++m)
        *reinterpret_cast<f32x4*>(&Os[((px_tile0 + n) * 16 + li) * LDO + (co_tile0 + m) * 16 + lg * 4]) =
            acc[m][n];
    if (tid < BM) {
      const int P = tile0 + tid;
      long long ob = -1;
      if (P < npix) {
        const int ni = P / hw;
        const int r = P - ni * hw;
        const int ho = r / g.WO;
        const int wo = r - ho * g.WO;
        ob = ((long long)(ni * g.OH + ho * g.ohs + g.oho) * g.OW + (wo * g.ows + g.owo)) * g.out_pitch +
             g.out_coff;
      }
      Ob[tid] = ob;
    }
    __syncthreads();
    constexpr int QO = COUT / 4;  // 16-byte pieces per pixel
    constexpr int NIT = (BM * QO + MDIL_WG - 1) / MDIL_WG;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      // no early-out and only clamped, unconditional loads: the residual / gate reads of all
      // iterations can be in flight together instead of one round trip per iteration
      const int idx = tid + it * MDIL_WG;
      const bool in_tile = ((BM * QO) % MDIL_WG == 0) || idx < BM * QO;
      const int p = in_tile ? idx / QO : 0, q = idx % QO;
      const long long ob = Ob[p];
      const bool ok = in_tile && ob >= 0;
      const long long obase = ok ? ob : 0;
      const int co = q * 4;
      f32x4 v = *reinterpret_cast<const f32x4*>(&Os[p * LDO + co]);
      if constexpr (QINV) {
        v = v * vscale + vbias;          // bias / folded-BN vectors fetched before the transpose
      } else {
        if (e.bias) v += *reinterpret_cast<const f32x4*>(e.bias + co);
        if (e.bias2) v += *reinterpret_cast<const f32x4*>(e.bias2 + co);
        if (e.scale)
          v = v * *reinterpret_cast<const f32x4*>(e.scale + co) +
              *reinterpret_cast<const f32x4*>(e.shift + co);
      }
      if (e.res) {
        f32x4 rr = *reinterpret_cast<const f32x4*>(e.res + obase + co);
        if (e.res_gate) {
          const f32x4 gg = *reinterpret_cast<const f32x4*>(e.res_gate + obase + co);
#pragma unroll
          for (int k = 0; k < 4; ++k) rr[k] = gg[k] > 0.f ? rr[k] : 0.f;
        }
        v += rr;
      }
      if (e.relu) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
      }
      if (e.gate) {
        const f32x4 gg = *reinterpret_cast<const f32x4*>(e.gate + obase + co);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = gg[k] > 0.f ? v[k] : 0.f;
      }
      if (ok) {
#if TC_NT_STORE
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out + obase + co));
#else
        *reinterpret_cast<f32x4*>(out + obase + co) = v;
#endif
      }
    }
  } else {
    // scalar path (13-channel stem slice)
#pragma unroll
    for (int n = 0; n < C::TN; ++n) {
      const int P = tile0 + (px_tile0 + n) * 16 + li;
      if (P >= npix) continue;
      const int ni = P / hw;
      const int r = P - ni * hw;
      const int ho = r / g.WO;
      const int wo = r - ho * g.WO;
      const long long obase =
          ((long long)(ni * g.OH + ho * g.ohs + g.oho) * g.OW + (wo * g.ows + g.owo)) * g.out_pitch +
          g.out_coff;
#pragma unroll
      for (int m = 0; m < C::TM; ++m) {
        const int co = (co_tile0 + m) * 16 + lg * 4;
        const f32x4 v = acc[m][n];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int c = co + k;
          if (c >= COUT) continue;
          float x = v[k];
          if (e.bias) x += e.bias[c];
          if (e.bias2) x += e.bias2[c];
          if (e.scale) x = x * e.scale[c] + e.shift[c];
          if (e.res) {
            float rr = e.res[obase + c];
            if (e.res_gate) rr = e.res_gate[obase + c] > 0.f ? rr : 0.f;
            x += rr;
          }
          if (e.relu) x = fmaxf(x, 0.f);
          if (e.gate) x = e.gate[obase + c] > 0.f ? x : 0.f;
          out[obase + c] = x;
        }
      }
    }
  }
#if TC_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stores of this wave have been acknowledged
  TC_STAMP(3);
#endif
}

__device__ __forceinline__ void pack_one(const mdil_pack_job& j, int first, int step) {
  const int total = j.ntaps * j.M_P * j.K_P;
  for (int idx = first; idx < total; idx += step) {
    const int k = idx % j.K_P;
    const int m = (idx / j.K_P) % j.M_P;
    const int t = idx / (j.K_P * j.M_P);
    float v = 0.f;
    if (m < j.M && k < j.K) {
      const long long off = j.stem ? (long long)m * 27 + (k % 3) * 9 + k / 3   // [co][c][tap] -> 3*tap+c
                                   : (long long)m * j.s_m + (long long)k * j.s_k + j.ktap[t];
      v = j.src[off];
    }
    j.dst[idx] = v;
  }
}

__global__ void pack_weights_kernel(const mdil_pack_job j) {
  pack_one(j, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// one launch refreshes every packed image of the model: blockIdx.x = job, blockIdx.y strides it
__global__ void pack_weights_batch_kernel(const mdil_pack_job* __restrict__ jobs) {
  const mdil_pack_job j = jobs[blockIdx.x];
  pack_one(j, blockIdx.y * blockDim.x + threadIdx.x, gridDim.y * blockDim.x);
}

template <int CIN, int COUT, int BM, bool STEM>
int launch_tapconv(const mdil_geom* g, const float* in0, const float* in1, const float* wpk,
                   const mdil_epilogue* epi, float* out, hipStream_t st) {
  const long long npix = (long long)g->N * g->HO * g->WO;
  const int grid = cdiv(npix, BM);
  hipLaunchKernelGGL((tapconv_kernel<CIN, COUT, BM, STEM>), dim3(grid), dim3(MDIL_WG), 0, st, *g,
                     in0, in1, wpk, *epi, out);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}

}  // namespace

extern "C" int mdil_pack_weights(const float* src, float* dst, int ntaps, const int* ktap, int M,
                                 int K, int M_P, int K_P, int s_m, int s_k, int stem, void* stream) {
  MDIL_CHECK_ARG(ntaps >= 1 && ntaps <= MDIL_MAX_TAPS, "pack_weights: ntaps=%d", ntaps);
  MDIL_CHECK_ARG(M_P >= M && K_P >= K, "pack_weights: padded dims smaller than dims");
  mdil_pack_job j;
  memset(&j, 0, sizeof(j));
  j.src = src;
  j.dst = dst;
  j.ntaps = ntaps;
  j.M = M;
  j.K = K;
  j.M_P = M_P;
  j.K_P = K_P;
  j.s_m = s_m;
  j.s_k = s_k;
  j.stem = stem;
  for (int t = 0; t < ntaps; ++t) j.ktap[t] = ktap ? ktap[t] : 0;
  const int total = ntaps * M_P * K_P;
  hipLaunchKernelGGL(pack_weights_kernel, dim3(cdiv(total, 256) > 1024 ? 1024 : cdiv(total, 256)),
                     dim3(256), 0, (hipStream_t)stream, j);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}

extern "C" int mdil_pack_weights_batch(const mdil_pack_job* jobs_device, int njobs, void* stream) {
  MDIL_CHECK_ARG(jobs_device && njobs >= 0, "pack_weights_batch: bad argument");
  if (njobs == 0) return MDIL_OK;
  hipLaunchKernelGGL(pack_weights_batch_kernel, dim3(njobs, 16), dim3(256), 0, (hipStream_t)stream,
                     jobs_device);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}

// profile path of a streaming C -> C conv launch (ops._PROF_CONV): 4 = Winograd F(4,3), 2 = F(2,3), 1 = direct
static int conv_path(const mdil_geom* g, int cin, int cout) {
  const int form = mdil_wconv_form(g, cin, cout);
  return form == 4 ? 4 : form == 2 ? 2 : 1;
}

// Blocks of BatchNorm partial statistics a conv launch with fused statistics would produce; 0 when
// this call cannot emit them (the caller then runs mdil_bn_train_stats on the output).
extern "C" int mdil_tapconv_stat_blocks(const mdil_geom* g, int cin, int cout) {
  static const bool use_sconv = getenv("MDIL_NO_SCONV") == nullptr && getenv("MDIL_NO_BNFUSE") == nullptr;
  if (!g || !use_sconv || !mdil_sconv_covers(g, cin, cout)) return 0;
  return mdil_sconv_stat_blocks(g, cin);
}

extern "C" int mdil_tapconv_stats(const mdil_geom* g, int cin, int cout, const float* in0,
                                  const float* in1, const float* wpk, const mdil_epilogue* epi,
                                  float* out, float* partial, float* pcount, void* stream) {
  MDIL_CHECK_ARG(g && epi && in0 && wpk && out && partial && pcount, "tapconv_stats: null argument");
  MDIL_CHECK_ARG(mdil_tapconv_stat_blocks(g, cin, cout) > 0, "tapconv_stats: call cannot emit statistics");
  MDIL_CHECK_ARG((epi->scale == nullptr) == (epi->shift == nullptr), "tapconv_stats: scale/shift");
  MdilProfScope ps((hipStream_t)stream, 0, g, cin, cout);
  ps.path = conv_path(g, cin, cout);
  return mdil_sconv(g, cin, cout, in0, in1, wpk, epi, out, partial, pcount, nullptr, nullptr, nullptr,
                    (hipStream_t)stream);
}

// dgrad launch whose stored (gated) gradient g feeds a BatchNorm backward: the reductions sum(g)
// and sum(g * xhat) ride in the epilogue -> partial[nblk][2][C] for mdil_bn_backward_partials.
static BnFinBwd make_fin_bwd(const mdil_bn_grad* fin, const float* save_invstd, long long npix) {
  BnFinBwd f;
  memset(&f, 0, sizeof(f));
  if (fin && fin->coef) {
    f.ticket = fin->ticket;
    f.gamma = fin->gamma, f.save_invstd = save_invstd;
    f.dgamma = fin->dgamma, f.dbeta = fin->dbeta, f.accumulate = fin->accumulate;
    f.n = (float)npix;
    f.coef = fin->coef;
  }
  return f;
}

extern "C" int mdil_tapconv_bnred(const mdil_geom* g, int cin, int cout, const float* in0,
                                  const float* in1, const float* wpk, const mdil_epilogue* epi,
                                  float* out, const float* bn_z, const float* save_mean,
                                  const float* save_invstd, float* partial, const mdil_bn_grad* fin,
                                  void* stream) {
  MDIL_CHECK_ARG(g && epi && in0 && wpk && out && bn_z && save_mean && save_invstd && partial,
                 "tapconv_bnred: null argument");
  const int nblk = mdil_tapconv_stat_blocks(g, cin, cout);
  MDIL_CHECK_ARG(nblk > 0, "tapconv_bnred: call is not covered");
  MDIL_CHECK_ARG(!epi->res_gate && !(epi->res && epi->gate), "tapconv_bnred: epilogue combination");
  MDIL_CHECK_ARG(!fin || (fin->gamma && fin->coef), "tapconv_bnred: fin needs gamma and coef");
  const BnFinBwd fb = make_fin_bwd(fin, save_invstd, (long long)g->N * g->HO * g->WO);
  int fused = 0;
  {
    MdilProfScope ps((hipStream_t)stream, 0, g, cin, cout);
    ps.path = conv_path(g, cin, cout);
    const int rc = mdil_sconv(g, cin, cout, in0, in1, wpk, epi, out, partial, nullptr, bn_z, save_mean,
                              save_invstd, (hipStream_t)stream, nullptr, fin ? &fb : nullptr, &fused);
    if (rc) return rc;
  }
  if (fin && !fused) {      // no ticket, or the direct-form kernel took the call: finalize here
    BnFinBwd f = fb;
    f.ticket = nullptr;
    return mdil_bn_finalize_bwd(partial, nblk, cout, f, (hipStream_t)stream);
  }
  return MDIL_OK;
}

// conv -> train-mode BatchNorm statistics -> coefficients (include/mdil_hip.h)
extern "C" int mdil_tapconv_bn_train(const mdil_geom* g, int cin, int cout, const float* in0,
                                     const float* in1, const float* wpk, const mdil_epilogue* epi,
                                     float* out, const mdil_bn_train* bn, void* workspace,
                                     size_t workspace_bytes, unsigned int* ticket, void* stream) {
  MDIL_CHECK_ARG(g && epi && in0 && wpk && out && bn, "tapconv_bn_train: null argument");
  MDIL_CHECK_ARG(bn->gamma && bn->beta && bn->coef, "tapconv_bn_train: BatchNorm fields");
  const long long npix = (long long)g->N * g->HO * g->WO;
  MDIL_CHECK_ARG(workspace && workspace_bytes >= mdil_bn_workspace(npix, cout), "tapconv_bn_train: workspace");
  float* c = bn->coef;
  const int nblk = mdil_tapconv_stat_blocks(g, cin, cout);
  if (nblk == 0 || epi->scale) {
    const int rc = mdil_tapconv(g, cin, cout, in0, in1, wpk, epi, out, stream);
    if (rc) return rc;
    return mdil_bn_train_stats(out, npix, cout, bn->gamma, bn->beta, bn->running_mean, bn->running_var,
                               bn->num_batches_tracked, bn->eps, bn->momentum, c, c + cout, c + 2 * cout,
                               c + 3 * cout, workspace, workspace_bytes, ticket, stream);
  }
  float* partial = (float*)workspace;
  float* pcount = partial + (size_t)MDIL_BN_MAX_BLOCKS * 2 * cout;
  BnFinFwd ff;
  ff.ticket = ticket;
  ff.gamma = bn->gamma, ff.beta = bn->beta;
  ff.running_mean = bn->running_mean, ff.running_var = bn->running_var, ff.nbt = bn->num_batches_tracked;
  ff.eps = bn->eps, ff.momentum = bn->momentum;
  ff.save_mean = c, ff.save_invstd = c + cout, ff.scale = c + 2 * cout, ff.shift = c + 3 * cout;
  int fused = 0;
  {
    MdilProfScope ps((hipStream_t)stream, 0, g, cin, cout);
    ps.path = conv_path(g, cin, cout);
    const int rc = mdil_sconv(g, cin, cout, in0, in1, wpk, epi, out, partial, pcount, nullptr, nullptr,
                              nullptr, (hipStream_t)stream, &ff, nullptr, &fused);
    if (rc) return rc;
  }
  if (fused) return MDIL_OK;
  ff.ticket = nullptr;
  return mdil_bn_finalize_fwd(partial, pcount, nblk, cout, ff, (hipStream_t)stream);
}

// Block-boundary fusion: the dgrad launch that produces a block's INPUT gradient
// gx also (a) gates it with that input (= the previous block's output; every consumer of gx applies
// this ReLU gate, so storing the gated value changes no result) and (b) emits the reductions
// sum(g), sum(g * xhat) of the previous block's OUTER BatchNorm backward, g = gated gx * drop --
// that block then runs mdil_bn_backward_partials instead of a reduction pass over three tensors.
// Covered by the Winograd streaming kernel only; mdil_tapconv_tail_blocks returns 0 otherwise.
extern "C" int mdil_tapconv_tail_blocks(const mdil_geom* g, int cin, int cout) {
  static const bool on = getenv("MDIL_NO_SCONV") == nullptr && getenv("MDIL_NO_BNFUSE") == nullptr &&
                         getenv("MDIL_NO_BNTAIL") == nullptr;
  if (!g || !on || !mdil_sconv_covers(g, cin, cout) || !mdil_wconv_covers(g, cin, cout) ||
      !mdil_wconv_tail_covers(g))
    return 0;
  return mdil_wconv_stat_blocks(g, cin);
}

extern "C" int mdil_tapconv_tail(const mdil_geom* g, int cin, int cout, const float* in0,
                                 const float* in1, const float* wpk, const mdil_epilogue* epi,
                                 float* out, const mdil_bn_tail* t, void* stream) {
  MDIL_CHECK_ARG(g && epi && in0 && wpk && out && t, "tapconv_tail: null argument");
  MDIL_CHECK_ARG(t->gate && t->z && t->save_mean && t->save_invstd && t->partial, "tapconv_tail: tail fields");
  MDIL_CHECK_ARG(mdil_tapconv_tail_blocks(g, cin, cout) > 0, "tapconv_tail: call is not covered");
  MDIL_CHECK_ARG(!epi->gate && !epi->relu && !epi->scale && (epi->res || !epi->res_gate),
                 "tapconv_tail: epilogue combination");
  for (int k = 0; k < g->ntaps; ++k)
    MDIL_CHECK_ARG(g->src[k] == 0 || (g->src[k] == 1 && in1), "tapconv_tail: tap %d source", k);
  MDIL_CHECK_ARG(!t->fin.coef || t->fin.gamma, "tapconv_tail: fin needs gamma");
  const BnFinBwd fb = make_fin_bwd(&t->fin, t->save_invstd, (long long)g->N * g->HO * g->WO);
  {
    MdilProfScope ps((hipStream_t)stream, 0, g, cin, cout);
    ps.path = conv_path(g, cin, cout);
    const int rc = mdil_wconv(g, cin, in0, in1, wpk, epi, out, t->partial, nullptr, t->z, t->save_mean,
                              t->save_invstd, (hipStream_t)stream, t->gate, t->drop, nullptr,
                              t->fin.coef ? &fb : nullptr);
    if (rc) return rc;
  }
  if (t->fin.coef && !t->fin.ticket)      // finalize requested without a ticket: stand-alone launch
    return mdil_bn_finalize_bwd(t->partial, mdil_tapconv_tail_blocks(g, cin, cout), cout, fb, (hipStream_t)stream);
  return MDIL_OK;
}

extern "C" int mdil_tapconv(const mdil_geom* g, int cin, int cout, const float* in0,
                            const float* in1, const float* wpk, const mdil_epilogue* epi,
                            float* out, void* stream) {
  MDIL_CHECK_ARG(g && epi && in0 && wpk && out, "tapconv: null argument");
  MDIL_CHECK_ARG(g->ntaps >= 1 && g->ntaps <= MDIL_MAX_TAPS, "tapconv: ntaps=%d", g->ntaps);
  MDIL_CHECK_ARG((long long)g->N * g->HO * g->WO < (1ll << 31), "tapconv: grid too large");
  MDIL_CHECK_ARG((epi->scale == nullptr) == (epi->shift == nullptr), "tapconv: scale/shift");
  for (int t = 0; t < g->ntaps; ++t) {
    MDIL_CHECK_ARG(g->src[t] == 0 || (g->src[t] == 1 && in1), "tapconv: tap %d source", t);
    MDIL_CHECK_ARG((cin == 27 && cout == 13) || g->in_pitch[g->src[t]] % 4 == 0, "tapconv: pitch %% 4");
  }
  hipStream_t st = (hipStream_t)stream;
  MdilProfScope ps(st, 0, g, cin, cout);
  // C -> C stride-1 convs of the factorised blocks (C = 64 / 128, 3 or 4 taps) take the
  // barrier-free streaming kernel (sconv.hip); MDIL_NO_SCONV=1 keeps them on the LDS-tiled
  // kernel below for A/B measurements (both give bit-identical results).
  static const bool use_sconv = getenv("MDIL_NO_SCONV") == nullptr;   // read once
  if (use_sconv) {
    const int rc = mdil_sconv(g, cin, cout, in0, in1, wpk, epi, out, nullptr, nullptr, nullptr, nullptr,
                              nullptr, st);
    if (rc != MDIL_ERR_UNSUPPORTED) {
      ps.path = conv_path(g, cin, cout);
      return rc;
    }
  }
  // 16 -> 16 channel convs of the decoder's last blocks: HBM-bound, no staging at all (c16conv.hip)
  static const bool use_c16 = getenv("MDIL_NO_C16CONV") == nullptr;
  if (use_c16 && mdil_c16conv_covers(g, cin, cout, epi)) {
    ps.path = 3;
    return mdil_c16conv(g, in0, in1, wpk, epi, out, st);
  }
#define TC(ci, co, bm, stem) \
  if (cin == ci && cout == co) return launch_tapconv<ci, co, bm, stem>(g, in0, in1, wpk, epi, out, st)
  TC(64, 64, 128, false);
  TC(128, 128, 64, false);
  TC(16, 16, 128, false);
  TC(16, 48, 128, false);
  TC(48, 16, 128, false);
  TC(128, 64, 128, false);
  TC(64, 128, 64, false);
  TC(64, 16, 128, false);
  TC(16, 64, 128, false);
  TC(16, 20, 128, false);
  TC(20, 16, 128, false);
  TC(16, 27, 128, false);  // 27-class head (IDD): logits rows are 28 floats
  TC(27, 16, 128, false);
  TC(27, 13, 128, true);   // RGB stem: cin==27 & cout==13
#undef TC
  mdil_set_error("tapconv: no tile configuration for cin=%d cout=%d", cin, cout);
  return MDIL_ERR_UNSUPPORTED;
}
