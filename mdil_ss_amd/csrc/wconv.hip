// Winograd F(2,3) streaming convolution for the 3-tap (3x1 / 1x3, dilated) C -> C convs of the
// factorised blocks and their dgrads, C = 64 / 128, NHWC fp32, gfx950.  Same structure as sconv.hip
// (one persistent work-group of 8 waves per CU, weights resident in LDS, no barrier in the main
// loop, B operands streamed from global memory into registers); what changes is the arithmetic.
//
// A 3-tap conv along one axis with dilation d computes, for the output pair (p, p + d),
//     y(p)     = g0 x(p-d) + g1 x(p)   + g2 x(p+d)
//     y(p + d) = g0 x(p)   + g1 x(p+d) + g2 x(p+2d)          (6 multiplications per pair and ci)
// Winograd's minimal form needs 4:   with d0..d3 = x(p-d), x(p), x(p+d), x(p+2d)
//     m1 = (d0 - d2) g0          m2 = (d1 + d2) (g0 + g1 + g2)/2
//     m3 = (d2 - d1) (g0 - g1 + g2)/2          m4 = (d1 - d3) g2
//     y(p) = m1 + m2 + m3        y(p + d) = m2 - m3 - m4
// Over channels each m_i is a C x C contraction, i.e. MFMA work: 4 instead of 6 per output pair --
// one third fewer fp32 MFMAs for the kernels that are bound by them.  The input transform costs
// 16 VALU instructions per 64 MFMAs, the weight transform is done once while the weights are
// loaded into LDS, the output transform in the epilogue.  The 1x1 adapter that rides as 4th tap
// (a different input tensor) joins in M space: A x2(p) is added into m1's accumulator and
// A x2(p+d) into the accumulator that holds -m4, so it costs its usual MFMAs and no extra weight copy.
//
// Numerics: exact fp32 products and accumulation as before, but of transformed operands: results
// differ from the direct form by a few ulp (measured: DESIGN.md 3.0).  Coverage: the axis length
// must be a multiple of 2d (every pair complete); otherwise the caller takes sconv.hip.
#include <stdlib.h>

#include "common.h"
#include "bnfin.h"
#include "wino.h"

namespace {


constexpr int WC_WAVES = 8;
constexpr int WC_THREADS = WC_WAVES * 64;
constexpr int WC_PAIRS = 16;  // output pairs per wave tile (32 pixels)

// COW = output channels per work-group: 64, or 32 where five weight images of 64 rows do not fit
// the LDS (C = 128 with the adapter: 5 x 64 x 132 floats = 165 KB)
template <int C, bool ADAPT, int PD>
struct WCfg {
  static constexpr int COW = (C == 128 && ADAPT) ? 32 : 64;
  static constexpr int TM = COW / 16;
  static constexpr int NH = C / COW;
  static constexpr int LD = C + 4;
  static constexpr int RPT = C / 16;            // 16-channel blocks
  static constexpr int NPOS = ADAPT ? 5 : 4;    // weight images in LDS: U0..U3 (+ adapter)
  static constexpr int NSUB = ADAPT ? 5 : 4;    // sub-rounds per channel block
  static constexpr int R = RPT * NSUB;          // sub-rounds per tile
  static constexpr int LDS_FLOATS = NPOS * COW * LD;
  static constexpr int NS = PD + 1;             // raw-operand ring (channel blocks)
  static_assert(RPT % NS == 0 && PD >= 1, "the ring must divide a tile's channel blocks");
};


constexpr int WC_STAT_LD = 2 * 64 + 4;     // per-wave statistics strip (COW <= 64)
constexpr int WC_TAIL_MAXN = 16;           // tail form: Dropout2d factors [N][COW] staged in LDS, N <= 16

// ---- stores.  In the accumulator layout a store instruction would write, per pixel, the 64 bytes
// of ONE 16-channel tile (lanes li, li+16, li+32, li+48): half a 128-byte line, which the
// non-temporal path hands to the fabric as partial writes (WRITE_SIZE 1.45x the tensor,
// profiles/r02_pmc_WRITE_SIZE.csv).  Two channel tiles of 8 pixels are exchanged between the lane
// halves of each 16-lane row (DPP row_ror:8) so that an instruction writes whole 128-byte lines of
// 8 pixels: WRITE_SIZE = the tensor, 1.00x (profiles/r03_wconv_store_forms.txt: the plain-store and
// half-line forms measured 241.8 / 240.7 img/s against 242.8 for this one).
template <int C, int COW, int TM>
__device__ __forceinline__ void wc_store(float* out, const f32x4 (&ay)[TM][2], int P0, bool ok, int S,
                                         int half, int li, int lg) {
  const bool hi = li >= 8;
  const int P0x = __builtin_amdgcn_update_dpp(0, P0, 0x128, 0xf, 0xf, false);
  const int okx = __builtin_amdgcn_update_dpp(0, (int)ok, 0x128, 0xf, 0xf, false);
  const int pix1 = hi ? P0x : P0, pix2 = hi ? P0 : P0x;
  const bool ok1 = hi ? okx != 0 : ok, ok2 = hi ? ok : okx != 0;
  const int choff = half * COW + (hi ? 16 : 0) + lg * 4;
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    const long long a1 = (long long)(ok1 ? pix1 + n * S : 0) * C + choff;
    const long long a2 = (long long)(ok2 ? pix2 + n * S : 0) * C + choff;
#pragma unroll
    for (int mp = 0; mp < TM / 2; ++mp) {
      const f32x4 A = ay[2 * mp][n], B = ay[2 * mp + 1][n];
      f32x4 R1, R2;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float X = hi ? A[k] : B[k];
        const float Y = wc_ror<0x128>(X);
        R1[k] = hi ? Y : A[k];       // pixels 0..7 of the tile: tile 2mp at li < 8, tile 2mp+1 at li >= 8
        R2[k] = hi ? B[k] : Y;       // pixels 8..15
      }
      if (ok1) __builtin_nontemporal_store(R1, reinterpret_cast<f32x4*>(out + a1 + mp * 32));
      if (ok2) __builtin_nontemporal_store(R2, reinterpret_cast<f32x4*>(out + a2 + mp * 32));
    }
  }
}

template <int C, bool ADAPT, int PD, int MODE, bool EOPS>
__global__ __launch_bounds__(WC_THREADS) void wconv_kernel(const wconv_args a) {
  using K = WCfg<C, ADAPT, PD>;
  constexpr int WC_COW = K::COW, WC_TM = K::TM;
  __shared__ __attribute__((aligned(16)))
  float Ws[K::LDS_FLOATS + 2 * WC_COW + (MODE ? WC_WAVES * WC_STAT_LD + 2 * WC_COW : 0) +
           (MODE == 3 ? WC_TAIL_MAXN * WC_COW : 0)];
  // MODE 1: BatchNorm statistics of the stored values; MODE 2: the stored gradient is gated and the
  // BatchNorm-backward reductions of it against bn_z ride along (the block's inner BN); MODE 3
  // ("tail"): the launch that produces a block's INPUT gradient gates it with that input (= the
  // previous block's output: every consumer of the gradient applies this ReLU gate anyway) and
  // emits the reductions of the previous block's outer BatchNorm backward (Dropout2d mask applied)
  constexpr bool STATS = MODE == 1;
  constexpr bool TAIL = MODE == 3;
  constexpr bool BNRED = MODE == 2 || TAIL;
  constexpr bool AFFINE = MODE < 2;      // bias / folded BN: forward convs only (the dgrad forms carry none)
  float* Ep = Ws + K::LDS_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int H = a.H, W = a.W;
  const int npix = a.N * H * W;
  const int npairs = npix >> 1;
  const int ntiles = (npairs + WC_PAIRS - 1) / WC_PAIRS;
  const int delta = a.delta;

  // work-group -> (channel part, pixel-tile queue): the NH work-groups that share a pixel tile
  // differ only in blockIdx bits 3.. (same XCD = blockIdx % 8, the later readers hit its L2)
  int half = 0, gq = blockIdx.x, nq = gridDim.x;      // `half` = channel part index (0 .. NH-1)
  if constexpr (K::NH > 1) {
    half = (blockIdx.x >> 3) % K::NH;
    gq = (blockIdx.x & 7) | ((blockIdx.x / (8 * K::NH)) << 3);
    nq = gridDim.x / K::NH;
  }

  // ---- weights -> Winograd domain -> LDS (once): row (pos, co), LD floats ----
  {
    constexpr int QPR = C / 4;
    constexpr int ITEMS = WC_COW * QPR;
    static_assert(ITEMS % WC_THREADS == 0, "weight rows divide over the work-group");
    constexpr int PER = ITEMS / WC_THREADS;
    f32x4 g0[PER], g1[PER], g2[PER], ga[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int idx = tid + u * WC_THREADS;
      const int q = idx % QPR;
      int co = (idx / QPR + blockIdx.x * 5) % WC_COW;     // stagger the rows between work-groups
      const long long row = (long long)(half * WC_COW + co) * C + q * 4;
      g0[u] = *reinterpret_cast<const f32x4*>(a.wpk + (long long)a.tap[0] * C * C + row);
      g1[u] = *reinterpret_cast<const f32x4*>(a.wpk + (long long)a.tap[1] * C * C + row);
      g2[u] = *reinterpret_cast<const f32x4*>(a.wpk + (long long)a.tap[2] * C * C + row);
      if constexpr (ADAPT) ga[u] = *reinterpret_cast<const f32x4*>(a.wpk + (long long)a.tap_ad * C * C + row);
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int idx = tid + u * WC_THREADS;
      const int q = idx % QPR;
      const int co = (idx / QPR + blockIdx.x * 5) % WC_COW;
      float* dst = &Ws[co * K::LD + q * 4];
      const f32x4 s02 = g0[u] + g2[u];
      *reinterpret_cast<f32x4*>(dst + 0 * WC_COW * K::LD) = g0[u];
      *reinterpret_cast<f32x4*>(dst + 1 * WC_COW * K::LD) = (s02 + g1[u]) * 0.5f;
      *reinterpret_cast<f32x4*>(dst + 2 * WC_COW * K::LD) = (s02 - g1[u]) * 0.5f;
      *reinterpret_cast<f32x4*>(dst + 3 * WC_COW * K::LD) = g2[u];
      if constexpr (ADAPT) *reinterpret_cast<f32x4*>(dst + 4 * WC_COW * K::LD) = ga[u];
    }
  }

  if (tid < WC_COW) {
    const int co = half * WC_COW + tid;
    float sc = 1.f, bi = a.e.bias ? a.e.bias[co] : 0.f;
    if (a.e.bias2) bi += a.e.bias2[co];
    if (a.e.scale) {
      sc = a.e.scale[co];
      bi = bi * sc + a.e.shift[co];
    }
    Ep[tid] = sc;
    Ep[WC_COW + tid] = bi;
  }

  const int in_bytes = npix * C * 4;
  const __amdgpu_buffer_rsrc_t rs0 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in0), 0, in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in1 ? a.in1 : a.in0), 0, in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs3 = a.src3 ? rs1 : rs0;
  const __amdgpu_buffer_rsrc_t rsa = a.src_ad ? rs1 : rs0;
  constexpr unsigned OOB = 0x80000000u;

  // pair -> pixels.  Pairs are numbered so that 16 consecutive pairs are as contiguous in memory
  // as the dilation allows: along W   pid = ((n H + h) (W / 2d) + wb) d + q,   w = 2d wb + q;
  // along H   pid = ((n (H / 2d) + hb) d + q) W + w,   h = 2d hb + q.   The partner is d further.
  const int L = a.axis ? W : H;                       // axis length
  const int S = a.axis ? delta : delta * W;           // pixel distance of the partner
  const int nb = L / (2 * delta);                     // pair blocks along the axis
  // vb[0..3]: byte offsets of d0..d3 (lane's 16 bytes: channels 4 lg ..); P0: first output pixel
  // (all sizes powers of two -- every layer of the 512 x 1024 network: shifts; else ~25 VALU
  // instructions per integer division, five of them per tile)
  const bool p2 = a.sh_delta >= 0;
  auto setup = [&](int tile, unsigned (&vb)[4], int& P0, int& img, bool& ok) {
    const int pid = tile * WC_PAIRS + li;
    ok = tile < ntiles && pid < npairs;
    const int pc = ok ? pid : 0;
    int x0;
    if (p2) {
      if (a.axis) {
        const int q = pc & (delta - 1), t1 = pc >> a.sh_delta;
        const int wb = t1 & (nb - 1), row = t1 >> a.sh_nb;
        x0 = 2 * delta * wb + q;
        P0 = row * W + x0;
        img = row >> a.sh_H;
      } else {
        const int w = pc & (W - 1), t1 = pc >> a.sh_W;
        const int q = t1 & (delta - 1), t2 = t1 >> a.sh_delta;
        const int hb = t2 & (nb - 1);
        img = t2 >> a.sh_nb;
        x0 = 2 * delta * hb + q;
        P0 = (img * H + x0) * W + w;
      }
    } else if (a.axis) {
      const int q = pc % delta, t1 = pc / delta;
      const int wb = t1 % nb, row = t1 / nb;
      x0 = 2 * delta * wb + q;
      P0 = row * W + x0;
      img = TAIL ? row / H : 0;
    } else {
      const int w = pc % W, t1 = pc / W;
      const int q = t1 % delta, t2 = t1 / delta;
      const int hb = t2 % nb;
      img = t2 / nb;
      x0 = 2 * delta * hb + q;
      P0 = (img * H + x0) * W + w;
    }
    const unsigned base = (unsigned)P0 * (unsigned)(C * 4) + (unsigned)lg * 16u;
    const unsigned sb = (unsigned)S * (unsigned)(C * 4);
    vb[0] = (ok && x0 - delta >= 0) ? base - sb : OOB;
    vb[1] = ok ? base : OOB;
    vb[2] = ok ? base + sb : OOB;
    vb[3] = (ok && x0 + 2 * delta < L) ? base + 2u * sb : OOB;
  };

  unsigned vbA[4], vbB[4];
  int P0A = 0, P0B = 0, imgA = 0, imgB = 0;
  bool okA = false, okB = false;
  f32x4 raw[K::NS][ADAPT ? 6 : 4];
  f32x4 acc[4][WC_TM];          // [Winograd position][16-channel tile], columns = the tile's 16 pairs

  // Running summaries of everything this wave has stored (MODE != 0); nothing is reduced across the
  // whole row per tile any more.  (Round 3 all-reduced every tile over its 16 lanes on the spot:
  // 128 ds_bpermute + ~600 VALU instructions per tile, a third of a C = 64 tile's main loop, and the
  // younger wave of each SIMD pays its epilogues with the matrix pipe idle.)
  //   STATS: per LANE, sn values per channel so far, sA = their mean, sB = their M2 (Welford / Chan);
  //          the 16 lanes of a row are merged once, after the wave's last tile
  //   BNRED: per tile a reduce-scatter over the row; rA = sum g, rB = sum g (z - mean) of ONE channel
  //          per lane (x invstd at the end)
  float sn = 0.f, rA = 0.f, rB = 0.f;
  f32x4 sA[STATS ? WC_TM : 1], sB[STATS ? WC_TM : 1];
  if constexpr (STATS) {
#pragma unroll
    for (int m = 0; m < WC_TM; ++m) sA[m] = sB[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float* Sw = Ws + K::LDS_FLOATS + 2 * WC_COW + wave * WC_STAT_LD;
  float* Bv = Ws + K::LDS_FLOATS + 2 * WC_COW + WC_WAVES * WC_STAT_LD;
  if constexpr (BNRED) {
    if (tid < WC_COW) {
      Bv[tid] = a.bn_mean[half * WC_COW + tid];
      Bv[WC_COW + tid] = a.bn_invstd[half * WC_COW + tid];
    }
  }
  // tail form: the previous block's Dropout2d factors [image][channel] (1 without dropout)
  float* Dt = Bv + 2 * WC_COW;
  if constexpr (TAIL) {
    for (int i = tid; i < a.N * WC_COW; i += WC_THREADS)
      Dt[i] = a.t_drop ? a.t_drop[(long long)(i / WC_COW) * C + half * WC_COW + i % WC_COW] : 1.f;
  }

  unsigned wbase[3];
  {
    const unsigned b = (unsigned)(__SIZE_TYPE__)((__attribute__((address_space(3))) float*)Ws) +
                       (unsigned)(li * K::LD + lg * 4) * 4u;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
      wbase[w] = b + w * WC_WIN;
      asm volatile("" : "+v"(wbase[w]));
    }
  }
  auto a_frag = [&](int pos, int m, int rr) __attribute__((always_inline)) {
    const unsigned off = (unsigned)(((pos * WC_COW + m * 16) * K::LD + rr * 16) * 4);
    return wlds_ld(wbase[off / WC_WIN] + off % WC_WIN);
  };
  auto load_block = [&](int slot, int rr, const unsigned (&vb)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 4; ++k) raw[slot][k] = wbuf_load(rs3, vb[k] + rr * 64);
    if constexpr (ADAPT) {
      raw[slot][4] = wbuf_load(rsa, vb[1] + rr * 64);
      raw[slot][5] = wbuf_load(rsa, vb[2] + rr * 64);
    }
  };

  int slot = wave;
  int tile = slot * nq + gq;
  setup(tile, vbA, P0A, imgA, okA);
#pragma unroll
  for (int r = 0; r < PD; ++r) load_block(r, r, vbA);

  __syncthreads();   // the only barrier: weights are resident from here on

  while (tile < ntiles) {
    const int ntile = (slot + WC_WAVES) * nq + gq;
    setup(ntile, vbB, P0B, imgB, okB);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int m = 0; m < WC_TM; ++m) acc[i][m] = f32x4{0.f, 0.f, 0.f, 0.f};

    // A fragments one sub-round ahead of their MFMAs
    f32x4 av[2][WC_TM];
#pragma unroll
    for (int m = 0; m < WC_TM; ++m) av[0][m] = a_frag(0, m, 0);
    // Input transform: B operand of Winograd position j from the raw block in ring slot sl.  Position 3
    // carries -m4 (d3 - d1 instead of d1 - d3; the output transform adds it), so that the adapter's
    // second contribution, +A x2(p + d), joins it without a negation: bit-identical (every product and
    // partial sum is the exact negative), four VALU instructions fewer per adapter block.
    auto xform = [&](int sl, int j, int c) __attribute__((always_inline)) -> float {
      return j == 0 ? raw[sl][0][c] - raw[sl][2][c]
           : j == 1 ? raw[sl][1][c] + raw[sl][2][c]
           : j == 2 ? raw[sl][2][c] - raw[sl][1][c]
                    : raw[sl][3][c] - raw[sl][1][c];
    };
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll
    for (int rr = 0; rr < K::RPT; ++rr) {
      // refill the ring PD channel blocks ahead (this tile, or block 0.. of the wave's next tile)
      if (rr + PD < K::RPT)
        load_block((rr + PD) % K::NS, rr + PD, vbA);
      else
        load_block((rr + PD) % K::NS, rr + PD - K::RPT, vbB);
      // input transform of this block (B operands of the four positions)
      f32x4 V[4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) V[j][c] = xform(rr % K::NS, j, c);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < K::NSUB; ++i) {
        const int sr = rr * K::NSUB + i;               // sub-round: weight image i of channel block rr
        const int nsr = (sr + 1) % K::R;
        const int npos = nsr % K::NSUB, nrr = nsr / K::NSUB;
        // MFMAs of the sub-round, order (k-step, channel tile); the next sub-round's four LDS reads
        // are spread behind the first ones.  The adapter sub-round feeds two accumulators from the
        // same weights: m1 (x2 at the pair's first pixel) and -m4 (x2 at the second).
        const bool ad = ADAPT && i == 4;
        const int reps = ad ? 2 : 1;
        int k = 0;
#pragma unroll
        for (int rep = 0; rep < reps; ++rep) {
          const int pos = ad ? (rep ? 3 : 0) : i;
          const f32x4 b = ad ? raw[rr % K::NS][4 + rep] : V[i];
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int m = 0; m < WC_TM; ++m) {
              acc[pos][m] = mfma16(av[sr & 1][m][s], b[s], acc[pos][m]);
              if (rep == 0 && (k & 1) && k < 2 * WC_TM) {
                av[(sr + 1) & 1][k / 2] = a_frag(npos, k / 2, nrr);
                __builtin_amdgcn_sched_barrier(0);
              }
              ++k;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // ---- output transform: y(p) = m1 + m2 + m3,  y(p + d) = m2 - m3 - m4 ----
    constexpr int TN = 2;
    f32x4 ay[WC_TM][TN];
#pragma unroll
    for (int m = 0; m < WC_TM; ++m) {
      ay[m][0] = (acc[0][m] + acc[1][m]) + acc[2][m];
      ay[m][1] = (acc[1][m] - acc[2][m]) + acc[3][m];      // acc[3] holds -m4
    }

    // ---- epilogue: lane holds out[pixel n of pair li][co = COW * half + 16m + 4lg .. +3] ----
    const mdil_epilogue& e = a.e;
    long long pb[TN];
#pragma unroll
    for (int n = 0; n < TN; ++n) pb[n] = (long long)(okA ? P0A + n * S : 0) * C + half * WC_COW + lg * 4;
    f32x4 r1[TN][WC_TM], r2[TN][WC_TM], r3[TN][WC_TM];     // operand tiles (r3: tail form only)
    auto ld_tile = [&](f32x4 (&r)[TN][WC_TM], const float* p) __attribute__((always_inline)) {
#pragma unroll
      for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int m = 0; m < WC_TM; ++m) r[n][m] = *reinterpret_cast<const f32x4*>(p + pb[n] + m * 16);
    };

    if constexpr (TAIL) {
      // stored value = (acc + res [where res_gate > 0]) where t_gate > 0; no affine part, no ReLU
      // (mdil_wconv rejects them).  Every operand whose registers are free is requested at once: a
      // head-gated residual (no res_gate: the usual case inside a chain of blocks) leaves room for
      // all three operands in ONE round trip; with res_gate the gate and the BN input follow in a second.
      const bool two = e.res_gate != nullptr;
      if (e.res) ld_tile(r1, e.res);
      if (two) {
        ld_tile(r2, e.res_gate);
      } else {
        ld_tile(r3, a.t_gate);
        ld_tile(r2, a.bn_z);
      }
      if (e.res) {
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
          for (int m = 0; m < WC_TM; ++m) {
            f32x4 x = r1[n][m];
            if (two) {
#pragma unroll
              for (int k = 0; k < 4; ++k) x[k] = r2[n][m][k] > 0.f ? x[k] : 0.f;
            }
            ay[m][n] += x;
          }
      }
      if (two) {
        ld_tile(r3, a.t_gate);
        ld_tile(r2, a.bn_z);
      }
#pragma unroll
      for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int m = 0; m < WC_TM; ++m)
#pragma unroll
          for (int k = 0; k < 4; ++k) ay[m][n][k] = r3[n][m][k] > 0.f ? ay[m][n][k] : 0.f;
      wc_store<C, WC_COW, WC_TM>(a.out, ay, P0A, okA, S, half, li, lg);
    } else {
      const float* opa = EOPS ? (e.res ? e.res : e.gate) : nullptr;
      const float* opb = EOPS ? (BNRED ? a.bn_z : e.res_gate) : nullptr;
      if (opa) ld_tile(r1, opa);
      if (opb) ld_tile(r2, opb);
      {
#pragma unroll
        for (int m = 0; m < WC_TM; ++m) {
          f32x4 vscale, vbias;
          if constexpr (AFFINE) {
            vscale = *reinterpret_cast<const f32x4*>(&Ep[m * 16 + lg * 4]);
            vbias = *reinterpret_cast<const f32x4*>(&Ep[WC_COW + m * 16 + lg * 4]);
          }
#pragma unroll
          for (int n = 0; n < TN; ++n) {
            f32x4 v = ay[m][n];
            if constexpr (AFFINE) v = v * vscale + vbias;
            if (EOPS && e.res) {
              f32x4 x = r1[n][m];
              if (e.res_gate) {
#pragma unroll
                for (int k = 0; k < 4; ++k) x[k] = r2[n][m][k] > 0.f ? x[k] : 0.f;
              }
              v += x;
            }
            if (e.relu) {
#pragma unroll
              for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
            }
            if (EOPS && e.gate && !e.res) {
#pragma unroll
              for (int k = 0; k < 4; ++k) v[k] = r1[n][m][k] > 0.f ? v[k] : 0.f;
            }
            ay[m][n] = v;
          }
        }
      }
      wc_store<C, WC_COW, WC_TM>(a.out, ay, P0A, okA, S, half, li, lg);
    }

    if constexpr (BNRED) {
      // sum g and sum g (z - mean) over this tile's 32 pixels (z sits in r2): the lane's two pixels
      // first, then the reduce-scatter over the row
      f32x4 p[WC_TM], q[WC_TM];
      const bool full = (tile + 1) * WC_PAIRS <= npairs;       // uniform; false on a ragged last tile only
#pragma unroll
      for (int m = 0; m < WC_TM; ++m) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(&Bv[m * 16 + lg * 4]);
        p[m] = ay[m][0] + ay[m][1];
        q[m] = ay[m][0] * (r2[0][m] - mu) + ay[m][1] * (r2[1][m] - mu);
        if constexpr (TAIL) {     // the BN branch carries the Dropout2d factor (both pixels: same image)
          const f32x4 dr = *reinterpret_cast<const f32x4*>(&Dt[imgA * WC_COW + m * 16 + lg * 4]);
          p[m] *= dr;
          q[m] *= dr;
        }
        if (!full && !okA) p[m] = q[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      rA += wc_reduce_scatter<WC_TM>(p, li);
      rB += wc_reduce_scatter<WC_TM>(q, li);
    }
    if constexpr (STATS) {
      // Welford / Chan: merge the pair (x0, x1) -- count 2, mean (x0 + x1)/2, M2 (x0 - x1)^2 / 2 --
      // into the lane's running (sn, mean, M2).  f = 2 / (sn + 2) from v_rcp_f32 (1 ulp: it scales a
      // DEVIATION from the running mean, an error of 1e-7 of a deviation).
      if (okA) {
        const float nn = sn + 2.f;
        const float f = 2.f * __builtin_amdgcn_rcpf(nn);
        const float nf = sn * f;
#pragma unroll
        for (int m = 0; m < WC_TM; ++m) {
          const f32x4 pm = (ay[m][0] + ay[m][1]) * 0.5f;
          const f32x4 dd = ay[m][0] - ay[m][1];
          const f32x4 d = pm - sA[m];
          sA[m] += d * f;
          sB[m] += (dd * dd) * 0.5f + (d * d) * nf;
        }
        sn = nn;
      }
    }

    slot += WC_WAVES;
    tile = ntile;
#pragma unroll
    for (int k = 0; k < 4; ++k) vbA[k] = vbB[k];
    P0A = P0B;
    imgA = imgB;
    okA = okB;
  }

  // ---- after the wave's last tile: the wave's summary goes to its strip, the strips are merged in
  // wave order.  STATS: the 16 lanes of each row (same channels) are merged first, by
  // rotate-and-combine in a fixed order (row_ror 8, 4, 2, 1); lane li == 0 of each row writes.
  if constexpr (BNRED) {
    if constexpr (WC_TM == 2) {
      rA += wc_ror<0x128>(rA);
      rB += wc_ror<0x128>(rB);
    }
    if (li < WC_TM * 4) {      // lane li holds channel 4m + k = li of its row's 16-channel groups
      const int c = (li >> 2) * 16 + lg * 4 + (li & 3);
      Sw[c] = rA;
      Sw[64 + c] = rB * Bv[WC_COW + c];
    }
    __syncthreads();
    if (tid < WC_COW) {
      const float* S0 = Ws + K::LDS_FLOATS + 2 * WC_COW;
      float sa = 0.f, sb = 0.f;
#pragma unroll
      for (int w = 0; w < WC_WAVES; ++w) {
        sa += S0[w * WC_STAT_LD + tid];
        sb += S0[w * WC_STAT_LD + 64 + tid];
      }
      float* r0 = a.stats + ((long long)gq * 2 + 0) * C + half * WC_COW + tid;
      float* r1_ = a.stats + ((long long)gq * 2 + 1) * C + half * WC_COW + tid;
      if (a.fb.ticket) {
        bnfin_st(r0, sa);
        bnfin_st(r1_, sb);
      } else {
        *r0 = sa;
        *r1_ = sb;
      }
    }
    if (a.fb.ticket) {       // the last work-group to arrive turns the rows into coefficients (bnfin.h)
      if (bnfin_arrive(a.fb.ticket, gridDim.x, reinterpret_cast<int*>(Ep)))
        bnfin_backward(a.fb, a.stats, nq, C, reinterpret_cast<double*>(Ws));
    }
  }
  if constexpr (STATS) {
    wc_stat_level<0x128, WC_TM>(sn, sA, sB);
    wc_stat_level<0x124, WC_TM>(sn, sA, sB);
    wc_stat_level<0x122, WC_TM>(sn, sA, sB);
    wc_stat_level<0x121, WC_TM>(sn, sA, sB);
    if (li == 0) {
#pragma unroll
      for (int m = 0; m < WC_TM; ++m) {
        *reinterpret_cast<f32x4*>(&Sw[m * 16 + lg * 4]) = sA[m];
        *reinterpret_cast<f32x4*>(&Sw[64 + m * 16 + lg * 4]) = sB[m];
      }
    }
    if (lane == 0) Sw[2 * 64] = sn;
    __syncthreads();
    if (tid < WC_COW) {
      const float* S0 = Ws + K::LDS_FLOATS + 2 * WC_COW;
      float n = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
      for (int w = 0; w < WC_WAVES; ++w)
        welford_merge(n, mean, m2, S0[w * WC_STAT_LD + 2 * 64], S0[w * WC_STAT_LD + tid],
                      S0[w * WC_STAT_LD + 64 + tid]);
      float* r0 = a.stats + ((long long)gq * 2 + 0) * C + half * WC_COW + tid;
      float* r1_ = a.stats + ((long long)gq * 2 + 1) * C + half * WC_COW + tid;
      if (a.ff.ticket) {
        bnfin_st(r0, mean);
        bnfin_st(r1_, m2);
        if (tid == 0 && half == 0) bnfin_st(a.stats_count + gq, n);
      } else {
        *r0 = mean;
        *r1_ = m2;
        if (tid == 0 && half == 0) a.stats_count[gq] = n;
      }
    }
    if (a.ff.ticket) {
      if (bnfin_arrive(a.ff.ticket, gridDim.x, reinterpret_cast<int*>(Ep)))
        bnfin_forward(a.ff, a.stats, a.stats_count, nq, C, Ws);
    }
  }
}


int wconv_queues(long long npix, int NH) {
  const int ntiles = (int)((npix / 2 + WC_PAIRS - 1) / WC_PAIRS);
  int nq = wc_num_cu() / NH;
  const int need = (ntiles + WC_WAVES - 1) / WC_WAVES;
  if (nq > need) nq = need;
  if (NH > 1) nq = (nq + 7) / 8 * 8;
  return nq;
}

template <int C, bool ADAPT, int PD, int MODE, bool EOPS>
int launch_wconv_(const wconv_args& a, hipStream_t st) {
  using K = WCfg<C, ADAPT, PD>;
  const int nq = wconv_queues((long long)a.N * a.H * a.W, K::NH);
  hipLaunchKernelGGL((wconv_kernel<C, ADAPT, PD, MODE, EOPS>), dim3(nq * K::NH), dim3(WC_THREADS), 0, st, a);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}

// Prefetch distance (channel blocks of B operands in flight per wave) by register budget: the
// variants without epilogue operands and the 32-channel work-groups have room for a deeper ring.
#ifndef WC_PD_EOPS
#define WC_PD_EOPS 1
#endif
#ifndef WC_PD_NOEOPS
#define WC_PD_NOEOPS 1
#endif
#ifndef WC_PD_COW32
#define WC_PD_COW32 1
#endif

template <int C, bool ADAPT>
int launch_wconv(const wconv_args& a, hipStream_t st) {
  constexpr bool COW32 = WCfg<C, ADAPT, 1>::COW == 32;
  constexpr int PDE = COW32 ? WC_PD_COW32 : WC_PD_EOPS, PDN = COW32 ? WC_PD_COW32 : WC_PD_NOEOPS;
  const bool eops = a.e.res || a.e.gate || a.e.res_gate;
  if (a.t_gate) return launch_wconv_<C, ADAPT, PDE, 3, true>(a, st);
  if (a.stats && a.bn_z) return launch_wconv_<C, ADAPT, PDE, 2, true>(a, st);
  if (a.stats) {
    // statistics AND epilogue operands with 64 output channels per work-group: the per-lane
    // summaries leave no room for the operand tiles (register spills) -- the direct-form kernel
    // keeps that combination (sconv.hip; same number of partials: same NH).  No launch of the
    // training step has it: a conv that feeds a train-mode BatchNorm carries biases only.
    if constexpr (!COW32) {
      if (eops) return MDIL_ERR_UNSUPPORTED;
      return launch_wconv_<C, ADAPT, PDN, 1, false>(a, st);
    } else {
      return eops ? launch_wconv_<C, ADAPT, PDE, 1, true>(a, st) : launch_wconv_<C, ADAPT, PDN, 1, false>(a, st);
    }
  }
  return eops ? launch_wconv_<C, ADAPT, PDE, 0, true>(a, st) : launch_wconv_<C, ADAPT, PDN, 0, false>(a, st);
}


}  // namespace

bool mdil_wconv_covers(const mdil_geom* g, int cin, int cout) {
  static const bool off = getenv("MDIL_NO_WCONV") != nullptr;
  if (off) return false;
  return wconv_plan(g, cin, nullptr);
}

// same contract as mdil_sconv; the caller has checked mdil_sconv_covers + the epilogue combination
// and mdil_wconv_covers.  The number of statistics partials equals sconv's (same tile count).
int mdil_wconv(const mdil_geom* g, int cin, const float* in0, const float* in1, const float* wpk,
               const mdil_epilogue* epi, float* out, float* stats, float* stats_count,
               const float* bn_z, const float* bn_mean, const float* bn_invstd, hipStream_t st,
               const float* tail_gate, const float* tail_drop, const BnFinFwd* ff, const BnFinBwd* fb) {
  // complete output quads: F(4,3), a quarter fewer MFMAs again (w4conv.hip)
  if (mdil_w4conv_covers(g, cin, cin))
    return mdil_w4conv(g, cin, in0, in1, wpk, epi, out, stats, stats_count, bn_z, bn_mean, bn_invstd, st,
                       tail_gate, tail_drop, ff, fb);
  wconv_args a;
  memset(&a, 0, sizeof(a));
  if (ff && stats && !bn_z) a.ff = *ff;
  if (fb && stats && bn_z) a.fb = *fb;
  if (!wconv_plan(g, cin, &a)) return MDIL_ERR_UNSUPPORTED;
  if (tail_gate && (!stats || !bn_z || !bn_mean || !bn_invstd || epi->gate || epi->relu)) return MDIL_ERR_INVALID;
  if (tail_gate && g->N > WC_TAIL_MAXN) return MDIL_ERR_UNSUPPORTED;
  // the reduction forms are dgrads: no bias / folded BN (the kernel compiles that stage out)
  if (bn_z && (epi->bias || epi->bias2 || epi->scale || epi->shift)) return MDIL_ERR_INVALID;
  a.t_gate = tail_gate;
  a.t_drop = tail_drop;
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
  {
    // power-of-two fast path of the pair -> pixel map (shifts instead of integer divisions)
    auto lg2 = [](int v) {
      int s = 0;
      while ((1 << s) < v) ++s;
      return (v > 0 && (1 << s) == v) ? s : -1;
    };
    const int L = a.axis ? a.W : a.H;
    const int nb = L / (2 * a.delta);
    a.sh_delta = lg2(a.delta), a.sh_nb = lg2(nb), a.sh_W = lg2(a.W), a.sh_H = lg2(a.H);
    if (a.sh_delta < 0 || a.sh_nb < 0 || a.sh_W < 0 || a.sh_H < 0) a.sh_delta = a.sh_nb = a.sh_W = a.sh_H = -1;
  }
  if (cin == 64) return g->ntaps == 3 ? launch_wconv<64, false>(a, st) : launch_wconv<64, true>(a, st);
  return g->ntaps == 3 ? launch_wconv<128, false>(a, st) : launch_wconv<128, true>(a, st);
}

// the tail form stages the Dropout2d factors of the whole batch in LDS
bool mdil_wconv_tail_covers(const mdil_geom* g) { return g->N <= WC_TAIL_MAXN; }   // w4conv.hip: the same bound

// partial statistics summaries a launch emits (= pixel-tile queues of its configuration)
int mdil_wconv_stat_blocks(const mdil_geom* g, int cin) {
  if (mdil_w4conv_covers(g, cin, cin)) return mdil_w4conv_stat_blocks(g, cin);
  const int NH = cin == 128 ? (g->ntaps == 4 ? 4 : 2) : 1;
  return wconv_queues((long long)g->N * g->HO * g->WO, NH);
}
