// Streaming tap convolution for the C -> C (C = 64 / 128) stride-1 convs of the factorised blocks
// (3x1 / 1x3 dilated convs, +1x1 adapter as 4th tap, and their dgrads) on gfx950, NHWC fp32.
//
// fp32 MFMA (v_mfma_f32_16x16x4_f32) runs at the fp32 VECTOR rate: one instruction occupies a
// SIMD's matrix pipe for 32 cycles while needing 2 x 4 bytes of operand per lane.  Every other
// pipe of the CU is therefore nearly idle next to it, and what limits an LDS-tiled kernel is not
// bandwidth but the work-group barriers around every staged K-chunk: all waves of a CU run in
// lock step, enter their staging phases together and leave the matrix pipe idle (measured:
// 50-60 % of the fp32 MFMA peak for tapconv.hip, 70 % / 56 % even with a free memory system).
//
// This kernel has NO barrier in its main loop:
//   * one persistent work-group of 8 waves per CU; the weights of ALL taps for 64 output channels
//     are loaded into LDS once ([tap][64][C+4] floats: 68 KB for C = 64, 132 KB for a 64-channel
//     half of C = 128) and only read afterwards;
//   * each WAVE owns whole output tiles (64 output channels x TN*16 pixels) and streams its own
//     B operands straight from global memory into registers in MFMA fragment order (lane = pixel
//     li, channels [16r + 4lg, +4): one 16-byte buffer load per 16 pixels x 16 channels, PD rounds
//     ahead of their use).  Out-of-image taps are buffer loads with an out-of-range offset, which
//     return 0 without a select;
//   * the two waves that share a SIMD drift apart (one gets the odd tile), so one wave's epilogue
//     (bias / folded BN / residual / gates / ReLU, 16-byte stores) runs under the other's MFMAs.
// The accumulation order (tap, 16-channel round, 4 MFMAs) is the one tapconv.hip uses, so both
// kernels give bit-identical results.
#include <stdlib.h>

#include "common.h"
#include "bnfin.h"

#ifndef SC_PIN
#define SC_PIN 1   // pin each round's loads / LDS reads in front of its MFMAs (scheduler barrier)
#endif

#ifndef SC_ABLATE
#define SC_ABLATE 0   // tuning builds only (results wrong by construction): 1 = no refill loads in the
#endif                // main loop, 2 = also no LDS reads, 3 = MFMAs only (no epilogue stores either)
#ifndef SC_TIMING
#define SC_TIMING 0   // tuning builds only: per-wave wall-clock stamps into a debug buffer
#endif
#if SC_TIMING
__device__ unsigned long long* sc_stamps = nullptr;
extern "C" int mdil_debug_set_sconv_stamps(void* p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(sc_stamps), &p, sizeof(p));
}
#define SC_STAMP(k)                                                                         \
  do {                                                                                      \
    if (lane == 0 && sc_stamps && (k) < 16) {                                               \
      unsigned long long* d_ = sc_stamps + ((long long)blockIdx.x * SC_WAVES + wave) * 32;  \
      d_[(k)] = wall_clock64();                                                             \
      d_[16 + (k)] = clock64();                                                             \
    }                                                                                       \
  } while (0)
#else
#define SC_STAMP(k)
#endif

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int SC_WAVES = 8;
constexpr int SC_THREADS = SC_WAVES * 64;
constexpr int SC_COW = 64;   // output channels per work-group (4 MFMA tiles per wave)
constexpr int SC_TM = 4;

template <int C, int NTAPS, int TN, int PD>
struct SCfg {
  static constexpr int NH = C / SC_COW;    // work-groups that share a pixel tile (channel halves)
  static constexpr int LD = C + 4;         // LDS row stride (floats)
  static constexpr int RPT = C / 16;       // 16-channel rounds per tap
  static constexpr int R = NTAPS * RPT;    // rounds per output tile
  static constexpr int PXT = 16 * TN;      // pixels per wave tile
  static constexpr int LDS_FLOATS = NTAPS * SC_COW * LD;
  static constexpr int NS = PD + 1;        // ring slots: PD rounds in flight + the one in use
  static_assert(R % 2 == 0 && PD >= 1 && PD <= RPT && R % NS == 0, "the ring must divide a tap's rounds");
};

struct sconv_args {
  const float* in0;
  const float* in1;
  const float* wpk;
  float* out;
  mdil_epilogue e;
  int N, H, W;
  int dh[4], dw[4], src[4];
  float* stats;          // optional [nq][2][C] per-work-group (mean, M2) partials of the stored values
  float* stats_count;    // [nq] pixel counts of the partials
  // BN-backward mode: stats = [nq][2][C] (sum g, sum g*xhat) of the stored gradient
  const float* bn_z;
  const float* bn_mean;
  const float* bn_invstd;
};

// LDS read at an explicit byte address.  The A-fragment reads of a tile use ~100 distinct constant
// offsets over up to 135 KB; left to itself hipcc materialises one base VGPR per offset group
// (49 of them for C = 128).  Three opaque window bases + immediates (< 64 KB) do the same job.
typedef const f32x4 __attribute__((address_space(3))) * lds_f4_ptr;
__device__ __forceinline__ f32x4 lds_ld(unsigned addr) {
  return *(lds_f4_ptr)(__SIZE_TYPE__)addr;
}
constexpr unsigned SC_WIN = 61440;   // window stride in bytes (immediates stay below 65536)

__device__ __forceinline__ f32x4 buf_load(const __amdgpu_buffer_rsrc_t r, unsigned voff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
  return __builtin_bit_cast(f32x4, v);
}

constexpr int SC_STAT_LD = 2 * SC_COW + 4;   // per-wave statistics scratch: mean[64], M2[64], count

template <int C, int NTAPS, int TN, int PD, int MODE, bool EOPS>
__global__ __launch_bounds__(SC_THREADS) void sconv_kernel(const sconv_args a) {
  using K = SCfg<C, NTAPS, TN, PD>;
  __shared__ __attribute__((aligned(16)))
  float Ws[K::LDS_FLOATS + 2 * SC_COW + (MODE ? SC_WAVES * SC_STAT_LD + 2 * SC_COW : 0)];
  constexpr bool STATS = MODE == 1;     // train-mode BN statistics of the stored values
  constexpr bool BNRED = MODE == 2;     // BN-backward reductions (sum g, sum g*xhat) of the stored g
  float* Ep = Ws + K::LDS_FLOATS;        // epilogue vectors of this work-group's 64 channels

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int H = a.H, W = a.W;
  const int hw = H * W;
  const int npix = a.N * hw;
  const int ntiles = (npix + K::PXT - 1) / K::PXT;
  SC_STAMP(0);
  int stamp_k = 2;
  (void)stamp_k;

  // work-group -> (channel half, pixel-tile queue).  The work-groups that share a pixel tile are
  // blockIdx b and b ^ 8: the same XCD (b % 8), so the second reader of a tile hits that XCD's L2.
  int half = 0, gq = blockIdx.x, nq = gridDim.x;
  if constexpr (K::NH == 2) {
    half = (blockIdx.x >> 3) & 1;
    gq = (blockIdx.x & 7) | ((blockIdx.x >> 4) << 3);
    nq = gridDim.x >> 1;
  }

  // ---- weights of all taps for this work-group's 64 output channels -> LDS (once) ----
  // All loads of a batch are issued before the first LDS write (one latency, not one per
  // element), and every work-group starts at a different row so the 256 work-groups that read the
  // same image at the same time do not walk the L2 channels in step.
  {
    constexpr int QPR = C / 4;
    constexpr int TOTAL = NTAPS * SC_COW * QPR;
#ifndef SC_WB
#define SC_WB 8
#endif
    constexpr int WB = SC_WB;                        // 16-byte loads in flight per thread
    static_assert(TOTAL % SC_THREADS == 0, "weight image divides over the work-group");
    constexpr int PER = TOTAL / SC_THREADS;
    const int rot = (blockIdx.x * 37) % (NTAPS * SC_COW);
#pragma unroll
    for (int b0 = 0; b0 < PER; b0 += WB) {
      f32x4 v[WB];
      int dsto[WB];
#pragma unroll
      for (int u = 0; u < WB; ++u) {
        if (b0 + u < PER) {
          const int idx = tid + (b0 + u) * SC_THREADS;
          const int q = idx % QPR;
          int row = idx / QPR + rot;                 // row = t * 64 + co
          row = row >= NTAPS * SC_COW ? row - NTAPS * SC_COW : row;
          const int t = row / SC_COW, co = row % SC_COW;
          v[u] = *reinterpret_cast<const f32x4*>(
              a.wpk + ((long long)(t * C + half * SC_COW + co)) * C + q * 4);
          dsto[u] = row * K::LD + q * 4;
        }
      }
#pragma unroll
      for (int u = 0; u < WB; ++u)
        if (b0 + u < PER) *reinterpret_cast<f32x4*>(&Ws[dsto[u]]) = v[u];
    }
  }

  // epilogue vectors (v * scale + bias form: bias only -> scale 1; folded BN -> scale, shift with
  // bias * scale folded in), staged once: an epilogue then needs no global round trip for them
  if (tid < SC_COW) {
    const int co = half * SC_COW + tid;
    float sc = 1.f, bi = a.e.bias ? a.e.bias[co] : 0.f;
    if (a.e.bias2) bi += a.e.bias2[co];
    if (a.e.scale) {
      sc = a.e.scale[co];
      bi = bi * sc + a.e.shift[co];
    }
    Ep[tid] = sc;
    Ep[SC_COW + tid] = bi;
  }

  // buffer descriptors (wave-uniform: built from kernel arguments only)
  const int in_bytes = npix * C * 4;
  const __amdgpu_buffer_rsrc_t rs0 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in0), 0, in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in1 ? a.in1 : a.in0), 0, in_bytes, 0x00020000);
  constexpr unsigned OOB = 0x80000000u;   // >= num_records for every tensor this kernel accepts

  // per-tile lane state: for every pixel tile n and tap t the byte offset of the lane's 16 bytes
  // (pixel li shifted by the tap, channel group lg), or an out-of-range offset where the tap
  // falls outside the image -- a refill load is then ONE instruction with no address arithmetic
  // (the 16-channel round is the instruction's immediate offset).  VALU instructions between
  // MFMAs are expensive for a wave that owns its SIMD (measured: +8 cycles per MFMA with a
  // handful of address selects per round), so none are left in the main loop.
  int tap_off[NTAPS];
#pragma unroll
  for (int t = 0; t < NTAPS; ++t) tap_off[t] = (a.dh[t] * W + a.dw[t]) * C * 4;
  auto setup = [&](int tile, unsigned (&vb)[TN][NTAPS]) {
#pragma unroll
    for (int n = 0; n < TN; ++n) {
      const int P = tile * K::PXT + 16 * n + li;
      const bool ok = tile < ntiles && P < npix;
      const int Pc = ok ? P : 0;
      const int img = Pc / hw;
      const int rem = Pc - img * hw;
      const int h = rem / W;
      const int w = rem - h * W;
      const unsigned base = (unsigned)Pc * (unsigned)(C * 4) + (unsigned)lg * 16u;
#pragma unroll
      for (int t = 0; t < NTAPS; ++t) {
        const int hh = h + a.dh[t], ww = w + a.dw[t];
        const bool v = ok && hh >= 0 && hh < H && ww >= 0 && ww < W;
        vb[n][t] = v ? base + (unsigned)tap_off[t] : OOB;
      }
    }
  };

  unsigned vbA[TN][NTAPS], vbB[TN][NTAPS];
  f32x4 bq[K::NS][TN];
  f32x4 acc[SC_TM][TN];

  // running BatchNorm statistics of this wave's tiles: (mean, M2) per channel in a wave-private
  // LDS strip (registers are needed for the pipeline), the pixel count in a register
  float st_n = 0.f;
  float* Sw = Ws + K::LDS_FLOATS + 2 * SC_COW + wave * SC_STAT_LD;
  float* Bv = Ws + K::LDS_FLOATS + 2 * SC_COW + SC_WAVES * SC_STAT_LD;   // BNRED: mean[64], invstd[64]
  if constexpr (MODE != 0) {
    Sw[lane] = 0.f;
    Sw[64 + lane] = 0.f;
  }
  if constexpr (BNRED) {
    if (wave == 0) {
      Bv[lane] = a.bn_mean[half * SC_COW + lane];
      Bv[SC_COW + lane] = a.bn_invstd[half * SC_COW + lane];
    }
  }

  // LDS byte address of this lane's A fragment origin (row li, channel group lg) + window bases
  unsigned wbase[3];
  {
    const unsigned b = (unsigned)(__SIZE_TYPE__)((__attribute__((address_space(3))) float*)Ws) +
                       (unsigned)(li * K::LD + lg * 4) * 4u;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
      wbase[w] = b + w * SC_WIN;
      asm volatile("" : "+v"(wbase[w]));     // keep the three bases, do not re-derive per offset
    }
  }
  auto a_frag = [&](int t, int m, int rr) __attribute__((always_inline)) {
    const unsigned off = (unsigned)(((t * SC_COW + m * 16) * K::LD + rr * 16) * 4);
    return lds_ld(wbase[off / SC_WIN] + off % SC_WIN);
  };

  int slot = wave;
  int tile = slot * nq + gq;
  setup(tile, vbA);
#pragma unroll
  for (int r = 0; r < PD; ++r)
#pragma unroll
    for (int n = 0; n < TN; ++n) bq[r][n] = buf_load(a.src[0] ? rs1 : rs0, vbA[n][0] + r * 64);

  __syncthreads();   // the only barrier: weights are resident from here on
  SC_STAMP(1);

  while (tile < ntiles) {
    const int ntile = (slot + SC_WAVES) * nq + gq;
    setup(ntile, vbB);
#pragma unroll
    for (int m = 0; m < SC_TM; ++m)
#pragma unroll
      for (int n = 0; n < TN; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    // operands of the epilogue (residual / gates / BN input, addressed like the output): their
    // loads are issued two rounds before the tile's last MFMA, so the epilogue finds them in
    // registers instead of paying a memory round trip per tile with the matrix pipe idle
    const mdil_epilogue& e = a.e;
    long long pb[TN];
    bool okp[TN];
    f32x4 ra[TN][SC_TM], rb[TN][SC_TM];   // ra: residual or gate, rb: residual gate or BN input
#pragma unroll
    for (int n = 0; n < TN; ++n) {
      const int P = tile * K::PXT + 16 * n + li;
      okp[n] = P < npix;
      pb[n] = (long long)(okp[n] ? P : 0) * C + half * SC_COW + lg * 4;
    }
    // EOPS = false: the launch has no such operand (plain forward convs): no load code at all
    const float* opa = EOPS ? (e.res ? e.res : e.gate) : nullptr;
    const float* opb = EOPS ? (BNRED ? a.bn_z : e.res_gate) : nullptr;
    auto epilogue_loads = [&]() __attribute__((always_inline)) {
      if (opa) {
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
          for (int m = 0; m < SC_TM; ++m) ra[n][m] = *reinterpret_cast<const f32x4*>(opa + pb[n] + m * 16);
      }
      if (opb) {
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
          for (int m = 0; m < SC_TM; ++m) rb[n][m] = *reinterpret_cast<const f32x4*>(opb + pb[n] + m * 16);
      }
    };

    // A fragments (weights, LDS) are read one round ahead of their MFMAs, B fragments (pixels,
    // global) PD rounds ahead: a wave never waits on either inside a round, so it keeps the
    // matrix pipe busy on its own.
    f32x4 av[2][SC_TM];
#pragma unroll
    for (int m = 0; m < SC_TM; ++m)
      av[0][m] = a_frag(0, m, 0);
#if SC_PIN
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int r = 0; r < K::R; ++r) {
      // Issue order of a round, pinned with scheduler fences: the TN refill loads and the SC_TM
      // LDS reads of the NEXT round are spread between the first MFMAs, two MFMAs (64 pipe
      // cycles) per slot, instead of bunched in front of them.
      constexpr int NM = 4 * SC_TM * TN;      // MFMAs per round, order (s, m, n)
      constexpr int PER_SLOT = 2;
      auto mf = [&](int k) __attribute__((always_inline)) {
        const int sx = k / (SC_TM * TN), m = (k / TN) % SC_TM, n = k % TN;
        acc[m][n] = mfma16(av[r & 1][m][sx], bq[r % K::NS][n][sx], acc[m][n]);
      };
      if constexpr (EOPS) {
        if (r == K::R - 2) epilogue_loads();
      }
      const int rn = (r + 1) % K::R;                          // round whose A fragments are read
      const int rl = (r + PD) % K::R;                         // round whose B fragments are loaded
      const int tl = rl / K::RPT;
      const __amdgpu_buffer_rsrc_t rs = a.src[tl] ? rs1 : rs0;
      int k = 0;
#pragma unroll
      for (int j = 0; j < TN + SC_TM; ++j) {
        if (j < TN) {
          // refill the ring PD rounds ahead (this tile, or tap 0 of the wave's next tile); the
          // slot written is not the one being consumed
          const unsigned voff = (r + PD < K::R) ? vbA[j][tl] : vbB[j][tl];
          if (SC_ABLATE < 1) bq[(r + PD) % K::NS][j] = buf_load(rs, voff + (rl % K::RPT) * 64);
          else asm volatile("" : "+v"(bq[(r + PD) % K::NS][j]) : "v"(voff));
        } else {
          const int m = j - TN;
          if (SC_ABLATE < 2)
            av[(r + 1) & 1][m] = a_frag(rn / K::RPT, m, rn % K::RPT);
          else
            asm volatile("" : "+v"(av[(r + 1) & 1][m]));
        }
#if SC_PIN
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int u = 0; u < PER_SLOT; ++u) mf(k++);
#if SC_PIN
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
#pragma unroll
      for (; k < NM; ++k) mf(k);
#if SC_PIN
      __builtin_amdgcn_sched_barrier(0);
#endif
    }

    SC_STAMP(stamp_k);
    // ---- epilogue: lane holds out[pixel 16n + li][co = 64*half + 16m + 4lg .. +3] ----
    f32x4 vscale[SC_TM], vbias[SC_TM];
#pragma unroll
    for (int m = 0; m < SC_TM; ++m) {
      vscale[m] = *reinterpret_cast<const f32x4*>(&Ep[m * 16 + lg * 4]);
      vbias[m] = *reinterpret_cast<const f32x4*>(&Ep[SC_COW + m * 16 + lg * 4]);
    }
#pragma unroll
    for (int n = 0; n < TN; ++n) {
#pragma unroll
      for (int m = 0; m < SC_TM; ++m) {
        f32x4 v = acc[m][n] * vscale[m] + vbias[m];
        if (EOPS && e.res) {
          f32x4 x = ra[n][m];
          if (e.res_gate) {
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = rb[n][m][k] > 0.f ? x[k] : 0.f;
          }
          v += x;
        }
        if (e.relu) {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        if (EOPS && e.gate && !e.res) {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = ra[n][m][k] > 0.f ? v[k] : 0.f;
        }
        if (okp[n]) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(a.out + pb[n] + m * 16));
        acc[m][n] = v;   // kept for the statistics pass below
      }
    }

    if constexpr (BNRED) {
      // BatchNorm backward needs sum(g) and sum(g * xhat) over all pixels before it can form its
      // input gradient: they ride along with the dgrad that produces g (acc holds the stored,
      // gated g; rb the BN input z) -- per-tile sums -> wave strip in LDS -> one partial per
      // work-group, merged in a fixed order by bn_bwd_finalize_kernel.
#pragma unroll
      for (int m = 0; m < SC_TM; ++m) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(&Bv[m * 16 + lg * 4]);
        const f32x4 is = *reinterpret_cast<const f32x4*>(&Bv[SC_COW + m * 16 + lg * 4]);
        f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < TN; ++n) {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float gk = okp[n] ? acc[m][n][k] : 0.f;
            sa[k] += gk;
            sb[k] += gk * ((rb[n][m][k] - mu[k]) * is[k]);
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
          for (int d = 1; d < 16; d <<= 1) {
            sa[k] += __shfl_xor(sa[k], d, 64);
            sb[k] += __shfl_xor(sb[k], d, 64);
          }
        }
        if (li == 0) {
          *reinterpret_cast<f32x4*>(&Sw[m * 16 + lg * 4]) =
              *reinterpret_cast<const f32x4*>(&Sw[m * 16 + lg * 4]) + sa;
          *reinterpret_cast<f32x4*>(&Sw[64 + m * 16 + lg * 4]) =
              *reinterpret_cast<const f32x4*>(&Sw[64 + m * 16 + lg * 4]) + sb;
        }
      }
    }

    if constexpr (STATS) {
      // BatchNorm statistics of the stored values ride along: two passes over the tile in
      // registers (mean, then squared deviations: no E[x^2]-E[x]^2 cancellation), Chan-merged
      // into the wave's running (count, mean, M2); merged per work-group at the end.
      const int nvalid = min(K::PXT, npix - tile * K::PXT);
      const float inv = 1.f / (float)nvalid;
#pragma unroll
      for (int m = 0; m < SC_TM; ++m) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < TN; ++n) {
          const bool ok = tile * K::PXT + 16 * n + li < npix;
#pragma unroll
          for (int k = 0; k < 4; ++k) s[k] += ok ? acc[m][n][k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
          for (int d = 1; d < 16; d <<= 1) s[k] += __shfl_xor(s[k], d, 64);
        }
        const f32x4 mean = s * inv;
        f32x4 q = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < TN; ++n) {
          const bool ok = tile * K::PXT + 16 * n + li < npix;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float d = acc[m][n][k] - mean[k];
            q[k] += ok ? d * d : 0.f;
          }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
          for (int d = 1; d < 16; d <<= 1) q[k] += __shfl_xor(q[k], d, 64);
        }
        if (li == 0) {
          f32x4 om = *reinterpret_cast<const f32x4*>(&Sw[m * 16 + lg * 4]);
          f32x4 oq = *reinterpret_cast<const f32x4*>(&Sw[64 + m * 16 + lg * 4]);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float nn = st_n, mm = om[k], qq = oq[k];
            welford_merge(nn, mm, qq, (float)nvalid, mean[k], q[k]);
            om[k] = mm;
            oq[k] = qq;
          }
          *reinterpret_cast<f32x4*>(&Sw[m * 16 + lg * 4]) = om;
          *reinterpret_cast<f32x4*>(&Sw[64 + m * 16 + lg * 4]) = oq;
        }
      }
      st_n += (float)nvalid;
    }

#if SC_TIMING
    SC_STAMP(stamp_k + 1);
    stamp_k += 2;
#endif
    slot += SC_WAVES;
    tile = ntile;
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
      for (int t = 0; t < NTAPS; ++t) vbA[n][t] = vbB[n][t];
  }

  if constexpr (BNRED) {
    __syncthreads();
    if (wave == 0) {
      const float* S0 = Ws + K::LDS_FLOATS + 2 * SC_COW;
      float sa = 0.f, sb = 0.f;
#pragma unroll
      for (int w = 0; w < SC_WAVES; ++w) {      // wave order: fixed => deterministic
        sa += S0[w * SC_STAT_LD + lane];
        sb += S0[w * SC_STAT_LD + SC_COW + lane];
      }
      a.stats[((long long)gq * 2 + 0) * C + half * SC_COW + lane] = sa;
      a.stats[((long long)gq * 2 + 1) * C + half * SC_COW + lane] = sb;
    }
  }
  if constexpr (STATS) {
    // one (count, mean, M2) partial per work-group: the waves' summaries are merged in wave order
    // (fixed order => deterministic)
    if (lane == 0) Sw[2 * SC_COW] = st_n;
    __syncthreads();
    if (wave == 0) {
      const float* S0 = Ws + K::LDS_FLOATS + 2 * SC_COW;
      float n = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
      for (int w = 0; w < SC_WAVES; ++w)
        welford_merge(n, mean, m2, S0[w * SC_STAT_LD + 2 * SC_COW], S0[w * SC_STAT_LD + lane],
                      S0[w * SC_STAT_LD + SC_COW + lane]);
      a.stats[((long long)gq * 2 + 0) * C + half * SC_COW + lane] = mean;
      a.stats[((long long)gq * 2 + 1) * C + half * SC_COW + lane] = m2;
      if (lane == 0 && half == 0) a.stats_count[gq] = n;
    }
  }
}

int g_num_cu = 0;

int num_cu() {
  if (g_num_cu == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    if (const char* e = getenv("MDIL_SCONV_CUS")) {     // tuning: use only this many CUs per launch
      const int v = atoi(e);
      if (v >= 16 && v <= n) n = v;
    }
    if (n > MDIL_BN_MAX_BLOCKS) n = MDIL_BN_MAX_BLOCKS;   // one statistics partial per queue
    g_num_cu = n;
  }
  return g_num_cu;
}

constexpr int SC_TN = 2;   // pixel tiles (of 16) per wave tile, every configuration

// pixel-tile queues (= work-groups per channel half): one persistent work-group per CU; fewer
// when there are not enough tiles to give every wave one
int sconv_queues(long long npix, int C) {
  const int NH = C / SC_COW;
  const int ntiles = (int)((npix + 16 * SC_TN - 1) / (16 * SC_TN));
  int nq = num_cu() / NH;
  const int need = (ntiles + SC_WAVES - 1) / SC_WAVES;
  if (nq > need) nq = need;
  if (NH == 2) nq = (nq + 7) / 8 * 8;        // the half bit sits above the XCD bits of blockIdx
  return nq;
}

template <int C, int NTAPS, int TN, int PD, int MODE, bool EOPS>
int launch_sconv_(const sconv_args& a, hipStream_t st);

template <int C, int NTAPS, int TN, int PD>
int launch_sconv(const sconv_args& a, hipStream_t st) {
  const bool eops = a.e.res || a.e.gate || a.e.res_gate;
  if (a.stats && a.bn_z) return launch_sconv_<C, NTAPS, TN, PD, 2, true>(a, st);
  if (a.stats)
    return eops ? launch_sconv_<C, NTAPS, TN, PD, 1, true>(a, st) : launch_sconv_<C, NTAPS, TN, PD, 1, false>(a, st);
  return eops ? launch_sconv_<C, NTAPS, TN, PD, 0, true>(a, st) : launch_sconv_<C, NTAPS, TN, PD, 0, false>(a, st);
}

template <int C, int NTAPS, int TN, int PD, int MODE, bool EOPS>
int launch_sconv_(const sconv_args& a, hipStream_t st) {
  using K = SCfg<C, NTAPS, TN, PD>;
  static_assert(TN == SC_TN, "sconv_queues assumes this tile");
  const int nq = sconv_queues((long long)a.N * a.H * a.W, C);
  hipLaunchKernelGGL((sconv_kernel<C, NTAPS, TN, PD, MODE, EOPS>), dim3(nq * K::NH), dim3(SC_THREADS), 0, st, a);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}

}  // namespace

bool mdil_sconv_covers(const mdil_geom* g, int cin, int cout) {
  if (cin != cout || (cin != 64 && cin != 128)) return false;
  if (g->ntaps != 3 && g->ntaps != 4) return false;
  if (g->ihs != 1 || g->iws != 1 || g->ohs != 1 || g->ows != 1 || g->oho || g->owo ||
      g->HI != g->HO || g->WI != g->WO || g->OH != g->HO || g->OW != g->WO || g->out_coff ||
      g->out_pitch != cin || g->in_pitch[0] != cin)
    return false;
  for (int t = 0; t < g->ntaps; ++t)
    if (g->src[t] && g->in_pitch[1] != cin) return false;
  return (long long)g->N * g->HO * g->WO * cin * 4 < (1ll << 31);
}

// the epilogue keeps one register set for "residual or gate": both at once stay on tapconv.hip
static bool sconv_epilogue_ok(const mdil_epilogue* e) { return !(e->res && e->gate); }

int mdil_sconv_stat_blocks(const mdil_geom* g, int cin) {
  if (mdil_wconv_covers(g, cin, cin)) return mdil_wconv_stat_blocks(g, cin);
  return sconv_queues((long long)g->N * g->HO * g->WO, cin);
}

// -> MDIL_ERR_UNSUPPORTED when the call is not a stride-1 C->C conv this kernel covers (the
// caller then uses the generic LDS-tiled kernel)
int mdil_sconv(const mdil_geom* g, int cin, int cout, const float* in0, const float* in1,
               const float* wpk, const mdil_epilogue* epi, float* out, float* stats,
               float* stats_count, const float* bn_z, const float* bn_mean, const float* bn_invstd,
               hipStream_t st, const BnFinFwd* ff, const BnFinBwd* fb, int* fused) {
  if (fused) *fused = 0;
  if (!mdil_sconv_covers(g, cin, cout) || !sconv_epilogue_ok(epi)) return MDIL_ERR_UNSUPPORTED;
  if (bn_z && (epi->res_gate || !stats || !bn_mean || !bn_invstd)) return MDIL_ERR_INVALID;
  // 3-tap convs with complete output pairs: Winograd F(2,3) form, a third fewer MFMAs (wconv.hip)
  // (it declines statistics + epilogue operands at 64 output channels per work-group: that one stays here)
  if (mdil_wconv_covers(g, cin, cout)) {
    const int rc = mdil_wconv(g, cin, in0, in1, wpk, epi, out, stats, stats_count, bn_z, bn_mean, bn_invstd, st,
                              nullptr, nullptr, ff, fb);
    if (rc != MDIL_ERR_UNSUPPORTED) {
      if (fused && rc == MDIL_OK) *fused = (ff && ff->ticket && stats && !bn_z) || (fb && fb->ticket && stats && bn_z);
      return rc;
    }
  }
  sconv_args a;
  memset(&a, 0, sizeof(a));
  a.in0 = in0;
  a.in1 = in1;
  a.wpk = wpk;
  a.out = out;
  a.e = *epi;
  a.N = g->N;
  a.H = g->HO;
  a.W = g->WO;
  a.stats = stats;
  a.stats_count = stats_count;
  a.bn_z = bn_z;
  a.bn_mean = bn_mean;
  a.bn_invstd = bn_invstd;
  if (bn_z && (epi->res_gate || !stats || !bn_mean || !bn_invstd)) return MDIL_ERR_INVALID;
  for (int t = 0; t < g->ntaps; ++t) {
    a.dh[t] = g->dh[t];
    a.dw[t] = g->dw[t];
    a.src[t] = g->src[t];
  }
#ifndef SC_PD64
#define SC_PD64 3
#endif
#ifndef SC_PD128
#define SC_PD128 3
#endif
  if (cin == 64) {
    if (g->ntaps == 3) return launch_sconv<64, 3, SC_TN, SC_PD64>(a, st);
    return launch_sconv<64, 4, SC_TN, SC_PD64>(a, st);
  }
  if (g->ntaps == 3) return launch_sconv<128, 3, SC_TN, SC_PD128>(a, st);
  return launch_sconv<128, 4, SC_TN, SC_PD128>(a, st);
}
