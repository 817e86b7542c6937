// EXPERIMENT (not built, not shipped): persistent wave-specialised C=64 schedule.  Measured
// 78 us vs 68 us for the generic kernel (3-tap, 196,608 px) on MI355X, and one epilogue
// combination failed parity -- kept only as a record of what was tried.
// Persistent, wave-specialised tap convolution for the C=64 stride-1 layers (3x1 / 1x3 convs,
// adapter as 4th tap, and their dgrads at 128x256 resolution: the largest single share of the
// step).  Same math / geometry / weight image / epilogue semantics as tapconv.hip.
//
// Why a second schedule (all numbers measured on MI355X):
//   * the generic kernel's launch = one resident generation of workgroups in lock step: every
//     workgroup first-touches its input at the same time, all of them store at the same time
//     (25-50 MB burst + end-of-kernel write-back = 10-14 us of a 58-70 us launch with no MFMA
//     running), and every workgroup re-fetches every weight tile from L2;
//   * with C=64 all weights of a launch (3-4 taps x 64 x 64 fp32 = 48-64 KB) fit in LDS.
// So: ONE workgroup per CU stays resident for the whole launch and walks over pixel tiles.
//   - weights are loaded into LDS once per workgroup;
//   - waves 0-3 ("compute", one per SIMD) issue nothing but ds_read_b128 + MFMA, 128 MFMAs per
//     stage; waves 4-7 ("loaders", one per SIMD) do the address arithmetic, the global loads and
//     the LDS writes of the NEXT stage into the other half of a double buffer.  The hardware
//     interleaves a SIMD's two waves, so staging never takes MFMA issue slots (in the generic
//     kernel all waves of a workgroup stage at the same moment);
//   - one barrier per stage (a stage = one tap of one 128-pixel tile, K = 64);
//   - a tile's output is stored (non-temporal) by the compute waves while the loaders already
//     fetch the next tile: stores and first-touch reads are spread over the whole launch.
#include "common.h"

namespace {

constexpr int P_NT = 512;             // threads: 4 compute waves + 4 loader waves
constexpr int P_BM = 128;             // pixels per tile
constexpr int P_C = 64;               // channels (in = out)
constexpr int P_LD = P_C + 4;         // LDS row stride (floats), 16 B pad
constexpr int P_IN = P_BM * P_LD;     // floats per input buffer
constexpr int P_W = P_C * P_LD;       // floats per tap of weights

__global__ __launch_bounds__(P_NT) void tapconv_p64_kernel(const mdil_geom g,
                                                           const float* __restrict__ in0,
                                                           const float* __restrict__ in1,
                                                           const float* __restrict__ wpk,
                                                           const mdil_epilogue e,
                                                           float* __restrict__ out, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wl = smem;                               // [ntaps][64][P_LD]
  float* In = smem + g.ntaps * P_W;               // [2][P_BM][P_LD]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int npix = g.N * g.HO * g.WO, hw = g.HO * g.WO;
  const int ntaps = g.ntaps;
  const bool loader = wave >= 4;

  // ---- weights -> LDS, once (all 512 threads) ----
  for (int idx = tid; idx < ntaps * P_C * 16; idx += P_NT) {
    const int q = idx & 15, row = idx >> 4;   // row = t*64 + co
    *reinterpret_cast<f32x4*>(&Wl[row * P_LD + q * 4]) =
        *reinterpret_cast<const f32x4*>(wpk + (long long)row * P_C + q * 4);
  }

  const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  const int nstage = my_tiles * ntaps;

  if (loader) {
    // =============================== loader waves ===============================
    const int ltid = tid - 256;
    const int q = ltid & 15;                 // 16-byte piece of the 64-channel row
    int c_nb[8], c_h[8], c_w[8];             // coordinates of this thread's 8 pixels (current tile)
    auto tile_coords = [&](int tile) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int P = tile * P_BM + (ltid >> 4) + 16 * i;
        if (P < npix) {
          const int n = P / hw;
          const int r = P - n * hw;
          const int ho = r / g.WO;
          c_nb[i] = n * g.HI;
          c_h[i] = ho * g.ihs;
          c_w[i] = (r - ho * g.WO) * g.iws;
        } else {
          c_nb[i] = 0;
          c_h[i] = -(1 << 28);
          c_w[i] = 0;
        }
      }
    };
    f32x4 reg[8];
    unsigned okm = 0;
    auto issue = [&](int t) {                // unconditional loads, clamped address
      const int s = g.src[t];
      const float* __restrict__ src = s ? in1 : in0;
      const int pitch = g.in_pitch[s];
      const int dh = g.dh[t], dw = g.dw[t];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int hi = c_h[i] + dh, wi = c_w[i] + dw;
        const bool ok = (hi >= 0) && (hi < g.HI) && (wi >= 0) && (wi < g.WI);
        const long long off = ok ? ((long long)(c_nb[i] + hi) * g.WI + wi) * pitch + q * 4 : 0ll;
        reg[i] = *reinterpret_cast<const f32x4*>(src + off);
        okm = ok ? (okm | (1u << i)) : (okm & ~(1u << i));
      }
    };
    auto commit = [&](float* buf) {          // registers -> LDS (zero fill for padding pixels)
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 8; ++i)
        *reinterpret_cast<f32x4*>(&buf[((ltid >> 4) + 16 * i) * P_LD + q * 4]) =
            ((okm >> i) & 1u) ? reg[i] : z;
    };
    if (nstage > 0) {
      tile_coords(blockIdx.x);
      issue(0);
      commit(In);
    }
    __syncthreads();                          // weights + stage 0 visible
    for (int st = 0; st < nstage; ++st) {
      const int nx = st + 1;
      if (nx < nstage) {
        const int t = nx % ntaps;
        if (t == 0) tile_coords(blockIdx.x + (nx / ntaps) * gridDim.x);
        issue(t);
        commit(In + (nx & 1) * P_IN);         // the buffer the compute waves finished one barrier ago
      }
      __syncthreads();
    }
  } else {
    // =============================== compute waves ===============================
    // wave w owns output channels [0,64) x pixels [32w, 32w+32) of the tile: TM = 4, TN = 2
    f32x4 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();                          // weights + stage 0 visible
    for (int st = 0; st < nstage; ++st) {
      const int t = st % ntaps;
      const float* Is = In + (st & 1) * P_IN;
      const float* Ws = Wl + t * P_W;
#pragma unroll
      for (int r = 0; r < 4; ++r) {           // K = 64 = 4 rounds of 16 channels
        f32x4 a[4], b[2];
#pragma unroll
        for (int m = 0; m < 4; ++m)
          a[m] = *reinterpret_cast<const f32x4*>(&Ws[(m * 16 + li) * P_LD + r * 16 + lg * 4]);
#pragma unroll
        for (int n = 0; n < 2; ++n)
          b[n] = *reinterpret_cast<const f32x4*>(&Is[((wave * 2 + n) * 16 + li) * P_LD + r * 16 + lg * 4]);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[m][n] = mfma16(a[m][s], b[n][s], acc[m][n]);
      }
      if (t == ntaps - 1) {
        // ---- tile finished: epilogue straight from the accumulators (stores are asynchronous
        //      and overlap the next tile's MFMAs) ----
        const int tile = blockIdx.x + (st / ntaps) * gridDim.x;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const int P = tile * P_BM + (wave * 2 + n) * 16 + li;
          if (P < npix) {
            const int ni = P / hw;
            const int r = P - ni * hw;
            const int ho = r / g.WO;
            const int wo = r - ho * g.WO;
            const long long obase =
                ((long long)(ni * g.OH + ho * g.ohs + g.oho) * g.OW + (wo * g.ows + g.owo)) * g.out_pitch +
                g.out_coff;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              const int co = m * 16 + lg * 4;
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
              __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(out + obase + co));
            }
          }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      __syncthreads();                        // one barrier per stage
    }
  }
}

}  // namespace

int mdil_tapconv_p64(const mdil_geom* g, int cin, int cout, const float* in0, const float* in1,
                     const float* wpk, const mdil_epilogue* epi, float* out, hipStream_t st) {
  if (cin != 64 || cout != 64 || g->ntaps > 4) return MDIL_ERR_UNSUPPORTED;
  for (int t = 0; t < g->ntaps; ++t)
    if (g->in_pitch[g->src[t]] != 64) return MDIL_ERR_UNSUPPORTED;
  const long long npix = (long long)g->N * g->HO * g->WO;
  const int ntiles = cdiv(npix, P_BM);
  if (ntiles < 512) return MDIL_ERR_UNSUPPORTED;   // small problems: the generic kernel
  const size_t lds = (size_t)(g->ntaps * P_W + 2 * P_IN) * sizeof(float);
  static bool configured = false;
  if (!configured) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&tapconv_p64_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    configured = true;
  }
  const int grid = ntiles < 256 ? ntiles : 256;
  hipLaunchKernelGGL(tapconv_p64_kernel, dim3(grid), dim3(P_NT), lds, st, *g, in0, in1, wpk, *epi, out,
                     ntiles);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}
