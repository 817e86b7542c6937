// Streaming tap convolution for the 16 -> 16 channel stride-1 convs of the decoder's last two
// factorised blocks (3x1 / 1x3 at 256x512 per image for a 512x1024 input, and their dgrads), NHWC
// fp32, gfx950.
//
// These launches are HBM-bound: 1.2 GFLOP against 100 MB of compulsory traffic (read T + write T,
// T = 50 MB at batch 6).  The generic LDS-tiled kernel (tapconv.hip) spends its time in the two
// barriers per tap and reaches 3.4 TB/s (29 us); this one 4.5 TB/s (22.4 us).  Here nothing is staged and nothing is shared:
//   * the whole weight set (3 taps x 16 x 16) lives in 12 registers per lane, already in MFMA
//     A-fragment order (lane = output channel li, input channels 4 lg ..);
//   * a wave owns one tile of 32 pixels, issues ALL its loads up front (6 x 16 bytes per lane:
//     one per tap and 16-pixel group, in B-fragment order; out-of-image taps are buffer loads with
//     an out-of-range offset that return 0) plus the epilogue operands, then runs its 24 MFMAs and
//     stores 16 bytes per lane and pixel group;
//   * work-group g -> tiles is XCD-major (blockIdx % 8 = XCD): every XCD walks a contiguous eighth
//     of the tensor, so the +-1 row / column halo of a tile is in that XCD's L2.
// Accumulation order (tap, 4 k-steps) is tapconv.hip's, so results are bit-identical to it.
#include "common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int C16 = 16;
#ifndef C16_TN_
#define C16_TN_ 2   // measured (100 MB launches): 1 -> 25.2, 2 -> 22.4, 4 -> 24.0, 8 -> 28.0 us
#endif
constexpr int C16_TN = C16_TN_;            // 16-pixel groups per wave tile
constexpr int C16_PXT = 16 * C16_TN;       // pixels per wave tile
constexpr int C16_WAVES = 4;

struct c16_args {
  const float* in0;
  const float* in1;
  const float* wpk;      // [tap][16][16]
  float* out;
  mdil_epilogue e;
  int N, H, W;
  int ntaps;
  int dh[4], dw[4], src[4];
  int nwg;               // work-groups that own tiles (the grid is padded to a multiple of 8)
};

__device__ __forceinline__ f32x4 c16_buf_load(const __amdgpu_buffer_rsrc_t r, unsigned voff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
  return __builtin_bit_cast(f32x4, v);
}

template <int NTAPS, bool EOPS>
__global__ __launch_bounds__(C16_WAVES * 64) void c16conv_kernel(const c16_args a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  // XCD-major walk: XCD x = blockIdx % 8 takes work-groups [x * per, (x + 1) * per)
  const int per = gridDim.x >> 3;
  const int g = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (g >= a.nwg) return;
  const int H = a.H, W = a.W, hw = H * W;
  const int npix = a.N * hw;
  const int tile = g * C16_WAVES + wave;
  const int P0 = tile * C16_PXT;
  if (P0 >= npix) return;

  const int in_bytes = npix * C16 * 4;
  const __amdgpu_buffer_rsrc_t rs0 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in0), 0, in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in1 ? a.in1 : a.in0), 0, in_bytes, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;

  // B fragments: lane (li, lg) of group n holds x[pixel 16 n + li, shifted by the tap][4 lg .. +3]
  f32x4 xq[NTAPS][C16_TN];
  bool okp[C16_TN];
  long long pb[C16_TN];
#pragma unroll
  for (int n = 0; n < C16_TN; ++n) {
    const int P = P0 + 16 * n + li;
    okp[n] = P < npix;
    const int Pc = okp[n] ? P : 0;
    pb[n] = (long long)Pc * C16 + lg * 4;
    const int img = Pc / hw;
    const int rem = Pc - img * hw;
    const int h = rem / W;
    const int w = rem - h * W;
    const unsigned base = (unsigned)Pc * (unsigned)(C16 * 4) + (unsigned)lg * 16u;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
      const int hh = h + a.dh[t], ww = w + a.dw[t];
      const bool v = okp[n] && hh >= 0 && hh < H && ww >= 0 && ww < W;
      const unsigned voff = v ? base + (unsigned)((a.dh[t] * W + a.dw[t]) * C16 * 4) : OOB;
      xq[t][n] = c16_buf_load(a.src[t] ? rs1 : rs0, voff);
    }
  }
  // epilogue operands, addressed like the output
  const mdil_epilogue& e = a.e;
  f32x4 ra[C16_TN], rb[C16_TN];
  if constexpr (EOPS) {
    const float* opa = e.res ? e.res : e.gate;
    const float* opb = e.res_gate;
#pragma unroll
    for (int n = 0; n < C16_TN; ++n) {
      if (opa) ra[n] = *reinterpret_cast<const f32x4*>(opa + pb[n]);
      if (opb) rb[n] = *reinterpret_cast<const f32x4*>(opb + pb[n]);
    }
  }
  // A fragments: lane (li, lg) holds W[t][co = li][ci = 4 lg .. +3]
  f32x4 wa[NTAPS];
#pragma unroll
  for (int t = 0; t < NTAPS; ++t) wa[t] = *reinterpret_cast<const f32x4*>(a.wpk + (t * C16 + li) * C16 + lg * 4);
  // epilogue vectors of the lane's 4 output channels (v * scale + bias form)
  f32x4 vscale = {1.f, 1.f, 1.f, 1.f}, vbias = {0.f, 0.f, 0.f, 0.f};
  if (e.bias) vbias = *reinterpret_cast<const f32x4*>(e.bias + lg * 4);
  if (e.bias2) vbias += *reinterpret_cast<const f32x4*>(e.bias2 + lg * 4);
  if (e.scale) {
    vscale = *reinterpret_cast<const f32x4*>(e.scale + lg * 4);
    vbias = vbias * vscale + *reinterpret_cast<const f32x4*>(e.shift + lg * 4);
  }

  f32x4 acc[C16_TN];
#pragma unroll
  for (int n = 0; n < C16_TN; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NTAPS; ++t)
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int n = 0; n < C16_TN; ++n) acc[n] = mfma16(wa[t][s], xq[t][n][s], acc[n]);

  // lane holds out[pixel 16 n + li][co = 4 lg .. +3]
#pragma unroll
  for (int n = 0; n < C16_TN; ++n) {
    f32x4 v = acc[n] * vscale + vbias;
    if (EOPS && e.res) {
      f32x4 x = ra[n];
      if (e.res_gate) {
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = rb[n][k] > 0.f ? x[k] : 0.f;
      }
      v += x;
    }
    if (e.relu) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
    }
    if (EOPS && e.gate && !e.res) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = ra[n][k] > 0.f ? v[k] : 0.f;
    }
    if (okp[n]) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(a.out + pb[n]));
  }
}

template <int NTAPS>
int launch_c16(const c16_args& a, hipStream_t st) {
  const int grid = (a.nwg + 7) / 8 * 8;
  const bool eops = a.e.res || a.e.gate || a.e.res_gate;
  if (eops)
    hipLaunchKernelGGL((c16conv_kernel<NTAPS, true>), dim3(grid), dim3(C16_WAVES * 64), 0, st, a);
  else
    hipLaunchKernelGGL((c16conv_kernel<NTAPS, false>), dim3(grid), dim3(C16_WAVES * 64), 0, st, a);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}

}  // namespace

bool mdil_c16conv_covers(const mdil_geom* g, int cin, int cout, const mdil_epilogue* e) {
  if (cin != 16 || cout != 16 || g->ntaps != 3) return false;
  if (g->ihs != 1 || g->iws != 1 || g->ohs != 1 || g->ows != 1 || g->oho || g->owo ||
      g->HI != g->HO || g->WI != g->WO || g->OH != g->HO || g->OW != g->WO || g->out_coff ||
      g->out_pitch != 16 || g->in_pitch[0] != 16)
    return false;
  for (int t = 0; t < g->ntaps; ++t)
    if (g->src[t] && g->in_pitch[1] != 16) return false;
  if (e->res && e->gate) return false;      // one register set for "residual or gate"
  return (long long)g->N * g->HO * g->WO * 16 * 4 < (1ll << 31);
}

int mdil_c16conv(const mdil_geom* g, const float* in0, const float* in1, const float* wpk,
                 const mdil_epilogue* epi, float* out, hipStream_t st) {
  c16_args a;
  memset(&a, 0, sizeof(a));
  a.in0 = in0;
  a.in1 = in1;
  a.wpk = wpk;
  a.out = out;
  a.e = *epi;
  a.N = g->N;
  a.H = g->HO;
  a.W = g->WO;
  a.ntaps = g->ntaps;
  for (int t = 0; t < g->ntaps; ++t) {
    a.dh[t] = g->dh[t];
    a.dw[t] = g->dw[t];
    a.src[t] = g->src[t];
  }
  const long long npix = (long long)g->N * g->HO * g->WO;
  const long long ntiles = (npix + C16_PXT - 1) / C16_PXT;
  a.nwg = (int)((ntiles + C16_WAVES - 1) / C16_WAVES);
  return launch_c16<3>(a, st);
}
