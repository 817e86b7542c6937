// Large-tile variant of the MFMA tap convolution for the C=64 / C=128 stride-1 layers that carry
// >80 % of the step's FLOPs (3x1 / 1x3 dilated convs, adapters as 4th tap, and their dgrads).
//
// Same math, geometry descriptor, weight image and epilogue as tapconv.hip; what changes is the
// schedule, chosen from measurements on MI355X (tools/bench_kernels.py, rocprofv3 PMC):
//   * the small-tile kernel sat at ~60 % MFMA-busy although its inner loop alone sustains
//     155 TFLOP/s (tools/probes/mfma_probe.hip): two barriers per 64 MFMAs, and every workgroup
//     of every CU re-fetching the same weight tile in lock-step (L2 bursts) were the cost;
//   * here one workgroup of 8 waves owns BM = 192 (C=128) / 256 (C=64) pixels x all channels:
//     a weight tile is fetched once per 192-256 pixels (3-4x less L2 traffic per MAC), each wave
//     issues 96-128 MFMAs per stage, the LDS tiles are double buffered so a stage costs ONE
//     barrier, and the next stage's global loads fly under the current stage's MFMAs.
//   * 49,152 pixels / 192 = 256 workgroups = one per CU for the C=128 layers at batch 6.
#include "common.h"

namespace {

constexpr int NT_BIG = 512;  // threads per workgroup (8 waves, 2 per SIMD)

template <int CC, int BM_, int WCO_>
struct BigCfg {
  static constexpr int BM = BM_;
  static constexpr int KC = 32, QPR = 8, LD = KC + 4;
  static constexpr int NCHUNK = CC / KC;
  static constexpr int MT = CC / 16, NT = BM / 16;
  static constexpr int WCO = WCO_, WPX = 8 / WCO_;
  static constexpr int TM = MT / WCO, TN = NT / WPX;
  static constexpr int IN_ITEMS = BM * QPR / NT_BIG;
  static constexpr int W_ITEMS = CC * QPR / NT_BIG;
  static constexpr int BUF = (BM + CC) * LD;  // floats per LDS buffer
  static_assert(MT % WCO == 0 && NT % WPX == 0, "wave tiling");
  static_assert((BM * QPR) % NT_BIG == 0 && (CC * QPR) % NT_BIG == 0, "staging items");
};

template <int CC, int BM, int WCO>
__global__ __launch_bounds__(NT_BIG) void tapconv_big_kernel(const mdil_geom g,
                                                             const float* __restrict__ in0,
                                                             const float* __restrict__ in1,
                                                             const float* __restrict__ wpk,
                                                             const mdil_epilogue e,
                                                             float* __restrict__ out) {
  using C = BigCfg<CC, BM, WCO>;
  extern __shared__ __attribute__((aligned(16))) float smem[];  // 2 x [BM + CC][LD]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int npix = g.N * g.HO * g.WO;
  const int hw = g.HO * g.WO;
  const int tile0 = blockIdx.x * BM;
  const int wco = wave % C::WCO, wpx = wave / C::WCO;
  const int co_tile0 = wco * C::TM, px_tile0 = wpx * C::TN;

  int it_nb[C::IN_ITEMS], it_h[C::IN_ITEMS], it_w[C::IN_ITEMS];
#pragma unroll
  for (int i = 0; i < C::IN_ITEMS; ++i) {
    const int P = tile0 + (tid + NT_BIG * i) / C::QPR;
    if (P < npix) {
      const int n = P / hw;
      const int r = P - n * hw;
      const int ho = r / g.WO;
      it_nb[i] = n * g.HI;
      it_h[i] = ho * g.ihs;
      it_w[i] = (r - ho * g.WO) * g.iws;
    } else {
      it_nb[i] = 0;
      it_h[i] = -(1 << 28);
      it_w[i] = 0;
    }
  }

  f32x4 regI[C::IN_ITEMS], regW[C::W_ITEMS];
  unsigned okI = 0;
  // unconditional loads (clamped address); zero fill is applied when the registers go to LDS
  auto issue_loads = [&](int t, int kc) {
    const int s = g.src[t];
    const float* __restrict__ src = s ? in1 : in0;
    const int pitch = g.in_pitch[s];
    const int dh = g.dh[t], dw = g.dw[t];
#pragma unroll
    for (int i = 0; i < C::IN_ITEMS; ++i) {
      const int q = (tid + NT_BIG * i) % C::QPR;
      const int hi = it_h[i] + dh, wi = it_w[i] + dw;
      const bool ok = (hi >= 0) && (hi < g.HI) && (wi >= 0) && (wi < g.WI);
      const long long off = ok ? ((long long)(it_nb[i] + hi) * g.WI + wi) * pitch + kc * C::KC + q * 4 : 0ll;
      regI[i] = *reinterpret_cast<const f32x4*>(src + off);
      okI = ok ? (okI | (1u << i)) : (okI & ~(1u << i));
    }
#pragma unroll
    for (int i = 0; i < C::W_ITEMS; ++i) {
      const int idx = tid + NT_BIG * i;
      regW[i] = *reinterpret_cast<const f32x4*>(wpk + ((long long)(t * CC + idx / C::QPR)) * CC +
                                                kc * C::KC + (idx % C::QPR) * 4);
    }
  };
  auto write_lds = [&](float* buf) {
    float* Is = buf;
    float* Ws = buf + BM * C::LD;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < C::IN_ITEMS; ++i) {
      const int idx = tid + NT_BIG * i;
      *reinterpret_cast<f32x4*>(&Is[(idx / C::QPR) * C::LD + (idx % C::QPR) * 4]) =
          ((okI >> i) & 1u) ? regI[i] : z;
    }
#pragma unroll
    for (int i = 0; i < C::W_ITEMS; ++i) {
      const int idx = tid + NT_BIG * i;
      *reinterpret_cast<f32x4*>(&Ws[(idx / C::QPR) * C::LD + (idx % C::QPR) * 4]) = regW[i];
    }
  };

  f32x4 acc[C::TM][C::TN];
#pragma unroll
  for (int a = 0; a < C::TM; ++a)
#pragma unroll
    for (int b = 0; b < C::TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nstage = g.ntaps * C::NCHUNK;
  issue_loads(0, 0);
  write_lds(smem);
  __syncthreads();
  for (int st = 0; st < nstage; ++st) {
    float* cur = smem + (st & 1) * C::BUF;
    float* nxt = smem + ((st & 1) ^ 1) * C::BUF;
    const bool more = st + 1 < nstage;
    if (more) issue_loads((st + 1) / C::NCHUNK, (st + 1) % C::NCHUNK);  // under the MFMAs below
    const float* Is = cur;
    const float* Ws = cur + BM * C::LD;
#pragma unroll
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
    if (more) write_lds(nxt);  // the other buffer: last read one barrier ago
    __syncthreads();           // one barrier per stage
  }

  // ---- epilogue (identical semantics to tapconv.hip) ----
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
      f32x4 v = acc[m][n];
      if (e.bias) v += *reinterpret_cast<const f32x4*>(e.bias + co);
      if (e.scale)
        v = v * *reinterpret_cast<const f32x4*>(e.scale + co) +
            *reinterpret_cast<const f32x4*>(e.shift + co);
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
      *reinterpret_cast<f32x4*>(out + obase + co) = v;
    }
  }
}

template <int CC, int BM, int WCO>
int launch_big(const mdil_geom* g, const float* in0, const float* in1, const float* wpk,
               const mdil_epilogue* epi, float* out, hipStream_t st) {
  using C = BigCfg<CC, BM, WCO>;
  const long long npix = (long long)g->N * g->HO * g->WO;
  const size_t lds = (size_t)2 * C::BUF * sizeof(float);
  static bool configured = false;   // one-time attribute set (idempotent, benign if raced)
  if (!configured) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&tapconv_big_kernel<CC, BM, WCO>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    configured = true;
  }
  hipLaunchKernelGGL((tapconv_big_kernel<CC, BM, WCO>), dim3(cdiv(npix, BM)), dim3(NT_BIG), lds, st,
                     *g, in0, in1, wpk, *epi, out);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}

}  // namespace

// called by mdil_tapconv for the C=64 / C=128 layers; returns MDIL_ERR_UNSUPPORTED when the
// problem is too small for the large tile to pay (the small-tile kernel then runs).
int mdil_tapconv_big(const mdil_geom* g, int cin, int cout, const float* in0, const float* in1,
                     const float* wpk, const mdil_epilogue* epi, float* out, hipStream_t st) {
  const long long npix = (long long)g->N * g->HO * g->WO;
  if (cin != cout) return MDIL_ERR_UNSUPPORTED;
  for (int t = 0; t < g->ntaps; ++t)
    if (g->in_pitch[g->src[t]] < cin) return MDIL_ERR_UNSUPPORTED;
  if (cin == 128 && npix >= 192 * 128) return launch_big<128, 192, 4>(g, in0, in1, wpk, epi, out, st);
  if (cin == 64 && npix >= 256 * 128) return launch_big<64, 256, 2>(g, in0, in1, wpk, epi, out, st);
  return MDIL_ERR_UNSUPPORTED;
}
