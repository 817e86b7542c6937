// Winograd F(4,3) streaming convolution for the 3-tap (3x1 / 1x3, dilated) C -> C convs of the
// factorised blocks and their dgrads, C = 64 / 128, NHWC fp32, gfx950.  Same structure as wconv.hip
// (F(2,3): one persistent work-group of 8 waves per CU, weights resident in LDS in the transform
// domain, no barrier in the main loop, B operands streamed from global memory into registers) with
// one more level of the minimal-filtering recursion.
//
// A 3-tap conv along one axis with dilation d computes the output QUAD (p, p+d, p+2d, p+3d) from
// the six inputs d0..d5 = x(p-d) .. x(p+4d): 12 multiplications per quad and input channel in the
// direct form, 8 with F(2,3), SIX with F(4,3):
//     t = B^T d (6 values)     m_j = t_j * U_j,  U = G g (6 values)     y = A^T m (4 values)
// Over channels every m_j is a C x C contraction, i.e. MFMA work: 6 contractions per 4 outputs where
// wconv.hip needs 8 -- a quarter fewer fp32 MFMAs for kernels that run at the matrix pipe's issue
// rate (profiles/r04_launch_cost_fit.txt).  Interpolation points (0, +-3/4, +-3/2, inf) instead of
// the textbook (0, +-1, +-2, inf): same operation count, every constant exact in fp32, 30 % less
// rounding error (tools/winograd_points.py; rms error / sum|a||b| 4.5e-8 vs 6.4e-8, direct form 1.7e-8).
//
// The 1x1 adapter that rides as 4th tap (a different input tensor) joins in M space where it can:
// A x2(p) is added into m0's accumulator (y0 = m0 + ...) and A x2(p+3d) into m5's (y3 = ... + m5);
// the two middle pixels get accumulators of their own: 10 contractions per quad (F(2,3): 12).
//
// Coverage: the axis length must be a multiple of 4d (every quad complete); otherwise the caller
// takes wconv.hip (multiple of 2d) or sconv.hip.
#include <stdlib.h>

#include "common.h"
#include "bnfin.h"
#include "wino.h"

#ifndef W4_ABLATE
#define W4_ABLATE 0   // tuning builds only (results wrong by construction; bit 0: no input transform, bit 1: no refill
#endif                // loads in the main loop, bit 2: no LDS reads of the weights after the first)
#ifndef W4_TIMING
#define W4_TIMING 0   // tuning builds only: per-wave wall-clock stamps into a debug buffer
#endif
#if W4_TIMING
__device__ unsigned long long* w4_stamps = nullptr;
extern "C" int mdil_debug_set_w4conv_stamps(void* p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(w4_stamps), &p, sizeof(p));
}
#define W4_STAMP(k)                                                                         \
  do {                                                                                      \
    if (lane == 0 && w4_stamps && (k) < 16) {                                               \
      unsigned long long* d_ = w4_stamps + ((long long)blockIdx.x * W4_WAVES + wave) * 32;  \
      d_[(k)] = wall_clock64();                                                             \
      d_[16 + (k)] = clock64();                                                             \
    }                                                                                       \
  } while (0)
#else
#define W4_STAMP(k)
#endif

namespace {

constexpr int W4_WAVES = 8;
constexpr int W4_THREADS = W4_WAVES * 64;
constexpr int W4_QUADS = 16;   // output quads per wave tile (64 pixels)
constexpr int W4_TN = 4;
// epilogue chunk (pixels of the quad whose operand tiles are in flight together)
#ifndef W4_NE_EOPS
#define W4_NE_EOPS 2      // adapter / statistics / 64-channel forms with epilogue operands: 4 would spill
#endif
#ifndef W4_NE_PLAIN
#define W4_NE_PLAIN 4     // 3-tap forms with epilogue operands: one round trip, 212-236 VGPRs (2: 187-195 VGPRs, but
#endif                    // 0.2 % slower in every schedule, profiles/r05_experiments.txt #8)

// interpolation points 0, +-PA, +-PB, inf
constexpr float PA = 0.75f, PB = 1.5f;
constexpr float PA2 = PA * PA, PB2 = PB * PB, PA3 = PA2 * PA, PB3 = PB2 * PB;
constexpr float PA2B2 = PA2 * PB2, PSUM = PA2 + PB2;
constexpr double GN0 = 1.0 / ((double)PA2 * PB2);
constexpr double GN1 = 1.0 / (2.0 * PA2 * ((double)PA2 - PB2));
constexpr double GN3 = 1.0 / (2.0 * PB2 * ((double)PB2 - PA2));

// COW = output channels per work-group (w4_cow below)
template <int C, int COW_, bool ADAPT, int PD>
struct W4Cfg {
  static constexpr int COW = COW_;
  static constexpr int TM = COW / 16;
  static constexpr int NH = C / COW;
  static constexpr int LD = C + 4;
  static constexpr int RPT = C / 16;            // 16-channel blocks
  static constexpr int NPOS = ADAPT ? 7 : 6;    // weight images in LDS: U0..U5 (+ adapter)
  static constexpr int NSUB = NPOS;             // sub-rounds per channel block
  static constexpr int R = RPT * NSUB;          // sub-rounds per tile
  static constexpr int LDS_FLOATS = NPOS * COW * LD;
  static constexpr int NS = PD + 1;             // raw-operand ring (channel blocks)
  static constexpr int NRAW = ADAPT ? 10 : 6;
  static_assert(RPT % NS == 0 && PD >= 1, "the ring must divide a tile's channel blocks");
};

__device__ __forceinline__ f32x4 w4_load(const __amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0);
  return __builtin_bit_cast(f32x4, v);
}

constexpr int W4_STAT_LD = 2 * 64 + 4;     // per-wave statistics strip (COW <= 64)
constexpr int W4_TAIL_MAXN = 16;           // tail form: Dropout2d factors [N][COW] staged in LDS, N <= 16

// stores: whole 128-byte lines of 8 pixels per instruction (two channel tiles exchanged between the
// lane halves of each 16-lane row; see wconv.hip)
template <int C, int COW, int TM, int NE>
__device__ __forceinline__ void w4_store(float* out, const f32x4 (&ay)[TM][W4_TN], int n0, int P0, bool ok, int S,
                                         int half, int li, int lg) {
  const bool hi = li >= 8;
  const int P0x = __builtin_amdgcn_update_dpp(0, P0, 0x128, 0xf, 0xf, false);
  const int okx = __builtin_amdgcn_update_dpp(0, (int)ok, 0x128, 0xf, 0xf, false);
  const int pix1 = hi ? P0x : P0, pix2 = hi ? P0 : P0x;
  const bool ok1 = hi ? okx != 0 : ok, ok2 = hi ? ok : okx != 0;
  const int choff = half * COW + (hi ? 16 : 0) + lg * 4;
#pragma unroll
  for (int nn = 0; nn < NE; ++nn) {
    const int n = n0 + nn;
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

template <int C, int COW, bool ADAPT, int PD, int MODE, bool EOPS>
__global__ __launch_bounds__(W4_THREADS) void w4conv_kernel(const wconv_args a) {
  using K = W4Cfg<C, COW, ADAPT, PD>;
  constexpr int TM = K::TM, TN = W4_TN;
  __shared__ __attribute__((aligned(16)))
  float Ws[K::LDS_FLOATS + 2 * COW + (MODE ? W4_WAVES * W4_STAT_LD + 2 * COW : 0) +
           (MODE == 3 ? W4_TAIL_MAXN * COW : 0)];
  // MODE 1: BatchNorm statistics of the stored values; MODE 2: the stored gradient is gated and the
  // BatchNorm-backward reductions of it against bn_z ride along; MODE 3 ("tail"): the block-boundary
  // form (see wconv.hip)
  constexpr bool STATS = MODE == 1;
  constexpr bool TAIL = MODE == 3;
  constexpr bool BNRED = MODE == 2 || TAIL;
  constexpr bool AFFINE = MODE < 2;
  float* Ep = Ws + K::LDS_FLOATS;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int H = a.H, W = a.W;
  const int npix = a.N * H * W;
  const int nquads = npix >> 2;
  const int ntiles = (nquads + W4_QUADS - 1) / W4_QUADS;
  const int delta = a.delta;
  W4_STAMP(0);
  [[maybe_unused]] int stamp_k = 2;

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
    constexpr int ITEMS = COW * QPR;
    static_assert(ITEMS % W4_THREADS == 0, "weight rows divide over the work-group");
    constexpr int PER = ITEMS / W4_THREADS;
    f32x4 g0[PER], g1[PER], g2[PER], ga[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int idx = tid + u * W4_THREADS;
      const int q = idx % QPR;
      int co = (idx / QPR + blockIdx.x * 5) % COW;     // stagger the rows between work-groups
      const long long row = (long long)(half * COW + co) * C + q * 4;
      g0[u] = *reinterpret_cast<const f32x4*>(a.wpk + (long long)a.tap[0] * C * C + row);
      g1[u] = *reinterpret_cast<const f32x4*>(a.wpk + (long long)a.tap[1] * C * C + row);
      g2[u] = *reinterpret_cast<const f32x4*>(a.wpk + (long long)a.tap[2] * C * C + row);
      if constexpr (ADAPT) ga[u] = *reinterpret_cast<const f32x4*>(a.wpk + (long long)a.tap_ad * C * C + row);
    }
#if W4_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // stamp 10: the raw taps have arrived from L2
    W4_STAMP(10);
#endif
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int idx = tid + u * W4_THREADS;
      const int q = idx % QPR;
      const int co = (idx / QPR + blockIdx.x * 5) % COW;
      float* dst = &Ws[co * K::LD + q * 4];
      const f32x4 ea = g0[u] + g2[u] * PA2, oa = g1[u] * PA;
      const f32x4 eb = g0[u] + g2[u] * PB2, ob = g1[u] * PB;
      *reinterpret_cast<f32x4*>(dst + 0 * COW * K::LD) = g0[u] * (float)GN0;
      *reinterpret_cast<f32x4*>(dst + 1 * COW * K::LD) = (ea + oa) * (float)GN1;
      *reinterpret_cast<f32x4*>(dst + 2 * COW * K::LD) = (ea - oa) * (float)GN1;
      *reinterpret_cast<f32x4*>(dst + 3 * COW * K::LD) = (eb + ob) * (float)GN3;
      *reinterpret_cast<f32x4*>(dst + 4 * COW * K::LD) = (eb - ob) * (float)GN3;
      *reinterpret_cast<f32x4*>(dst + 5 * COW * K::LD) = g2[u];
      if constexpr (ADAPT) *reinterpret_cast<f32x4*>(dst + 6 * COW * K::LD) = ga[u];
    }
#if W4_TIMING
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // stamp 11: transformed and written to LDS
    W4_STAMP(11);
#endif
  }

  if (tid < COW) {
    const int co = half * COW + tid;
    float sc = 1.f, bi = a.e.bias ? a.e.bias[co] : 0.f;
    if (a.e.bias2) bi += a.e.bias2[co];
    if (a.e.scale) {
      sc = a.e.scale[co];
      bi = bi * sc + a.e.shift[co];
    }
    Ep[tid] = sc;
    Ep[COW + tid] = bi;
  }

  const int in_bytes = npix * C * 4;
  const __amdgpu_buffer_rsrc_t rs0 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in0), 0, in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in1 ? a.in1 : a.in0), 0, in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs3 = a.src3 ? rs1 : rs0;
  const __amdgpu_buffer_rsrc_t rsa = a.src_ad ? rs1 : rs0;
  constexpr unsigned OOB = 0x80000000u;

  // quad -> pixels.  Quads are numbered so that 16 consecutive quads are as contiguous in memory
  // as the dilation allows: along W   qid = ((n H + h) (W / 4d) + wb) d + q,   w = 4d wb + q;
  // along H   qid = ((n (H / 4d) + hb) d + q) W + w,   h = 4d hb + q.   The quad's pixels are d apart.
  const int L = a.axis ? W : H;                       // axis length
  const int S = a.axis ? delta : delta * W;           // pixel distance inside a quad
  const int nb = L / (4 * delta);                     // quad blocks along the axis
  const int sb = S * (C * 4);                         // the same in bytes (scalar: rides in soffset)
  // vb[0]: byte offset of d0 = x(p - d) or OOB; vb[1]: of d1 = x(p) (d2..d4 = vb[1] + k sb through
  // the scalar offset); vb[2]: of d5 = x(p + 4d) or OOB.  P0: first output pixel.
  const bool p2 = a.sh_delta >= 0;
  auto setup = [&](int tile, unsigned (&vb)[3], int& P0, int& img, bool& ok) {
    const int qid = tile * W4_QUADS + li;
    ok = tile < ntiles && qid < nquads;
    const int pc = ok ? qid : 0;
    int x0;
    if (p2) {
      if (a.axis) {
        const int q = pc & (delta - 1), t1 = pc >> a.sh_delta;
        const int wb = t1 & (nb - 1), row = t1 >> a.sh_nb;
        x0 = 4 * delta * wb + q;
        P0 = row * W + x0;
        img = row >> a.sh_H;
      } else {
        const int w = pc & (W - 1), t1 = pc >> a.sh_W;
        const int q = t1 & (delta - 1), t2 = t1 >> a.sh_delta;
        const int hb = t2 & (nb - 1);
        img = t2 >> a.sh_nb;
        x0 = 4 * delta * hb + q;
        P0 = (img * H + x0) * W + w;
      }
    } else if (a.axis) {
      const int q = pc % delta, t1 = pc / delta;
      const int wb = t1 % nb, row = t1 / nb;
      x0 = 4 * delta * wb + q;
      P0 = row * W + x0;
      img = TAIL ? row / H : 0;
    } else {
      const int w = pc % W, t1 = pc / W;
      const int q = t1 % delta, t2 = t1 / delta;
      const int hb = t2 % nb;
      img = t2 / nb;
      x0 = 4 * delta * hb + q;
      P0 = (img * H + x0) * W + w;
    }
    const unsigned base = (unsigned)P0 * (unsigned)(C * 4) + (unsigned)lg * 16u;
    vb[0] = (ok && x0 - delta >= 0) ? base - (unsigned)sb : OOB;
    vb[1] = ok ? base : OOB;
    vb[2] = (ok && x0 + 4 * delta < L) ? base + 4u * (unsigned)sb : OOB;
  };

  unsigned vbA[3], vbB[3];
  int P0A = 0, P0B = 0, imgA = 0, imgB = 0;
  bool okA = false, okB = false;
  f32x4 raw[K::NS][K::NRAW];
  f32x4 acc[6][TM];             // [Winograd position][16-channel tile], columns = the tile's 16 quads
  f32x4 accx[ADAPT ? 2 : 1][TM];   // the adapter's two middle pixels

  // running summaries of everything this wave has stored (MODE != 0), see wconv.hip
  float sn = 0.f, rA = 0.f, rB = 0.f;
  f32x4 sA[STATS ? TM : 1], sB[STATS ? TM : 1];
  if constexpr (STATS) {
#pragma unroll
    for (int m = 0; m < TM; ++m) sA[m] = sB[m] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float* Sw = Ws + K::LDS_FLOATS + 2 * COW + wave * W4_STAT_LD;
  float* Bv = Ws + K::LDS_FLOATS + 2 * COW + W4_WAVES * W4_STAT_LD;
  if constexpr (BNRED) {
    if (tid < COW) {
      Bv[tid] = a.bn_mean[half * COW + tid];
      Bv[COW + tid] = a.bn_invstd[half * COW + tid];
    }
  }
  float* Dt = Bv + 2 * COW;
  if constexpr (TAIL) {
    for (int i = tid; i < a.N * COW; i += W4_THREADS)
      Dt[i] = a.t_drop ? a.t_drop[(long long)(i / COW) * C + half * COW + i % COW] : 1.f;
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
    const unsigned off = (unsigned)(((pos * COW + m * 16) * K::LD + rr * 16) * 4);
    return wlds_ld(wbase[off / WC_WIN] + off % WC_WIN);
  };
  auto load_block = [&](int slot, int rr, const unsigned (&vb)[3]) __attribute__((always_inline)) {
    raw[slot][0] = w4_load(rs3, vb[0] + rr * 64, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) raw[slot][1 + k] = w4_load(rs3, vb[1] + rr * 64, k * sb);
    raw[slot][5] = w4_load(rs3, vb[2] + rr * 64, 0);
    if constexpr (ADAPT) {
#pragma unroll
      for (int k = 0; k < 4; ++k) raw[slot][6 + k] = w4_load(rsa, vb[1] + rr * 64, k * sb);
    }
  };

  float kNB2 = -PB2, kNA2 = -PA2, kA2B2 = PA2B2, kNSUM = -PSUM, kPA = PA, kNA = -PA, kPB = PB, kNB = -PB;
  asm volatile("" : "+s"(kNB2), "+s"(kNA2), "+s"(kA2B2), "+s"(kNSUM), "+s"(kPA), "+s"(kNA), "+s"(kPB), "+s"(kNB));

  int slot = wave;
  int tile = slot * nq + gq;
  setup(tile, vbA, P0A, imgA, okA);
#pragma unroll
  for (int r = 0; r < PD; ++r) load_block(r, r, vbA);
#if W4_TIMING
  W4_STAMP(12);                                          // stamp 12: set-up done, first operands requested
#endif

  __syncthreads();   // the only barrier: weights are resident from here on
  W4_STAMP(1);

  while (tile < ntiles) {
    const int ntile = (slot + W4_WAVES) * nq + gq;
    setup(ntile, vbB, P0B, imgB, okB);
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int m = 0; m < TM; ++m) acc[i][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (ADAPT) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int m = 0; m < TM; ++m) accx[i][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // A fragments one sub-round ahead of their MFMAs
    f32x4 av[2][TM];
#pragma unroll
    for (int m = 0; m < TM; ++m) av[0][m] = a_frag(0, m, 0);
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll
    for (int rr = 0; rr < K::RPT; ++rr) {
      // refill the ring PD channel blocks ahead (this tile, or block 0.. of the wave's next tile)
#if !(W4_ABLATE & 2)
      if (rr + PD < K::RPT)
        load_block((rr + PD) % K::NS, rr + PD, vbA);
      else
        load_block((rr + PD) % K::NS, rr + PD - K::RPT, vbB);
#endif
      // input transform of this block: t = B^T d, the B operands of the six positions
      f32x4 V[6];
      {
        const f32x4(&d)[K::NRAW] = raw[rr % K::NS];
#if W4_ABLATE & 1
#pragma unroll
        for (int j = 0; j < 6; ++j) V[j] = d[j];
#else
        // 12 multiply-adds per element = 24 v_pk_fma_f32 per block; the signed constants are opaque
        // scalars (left to fold the signs, hipcc negates operands with a v_xor per register)
        const f32x4 e1 = d[2] * kNB2 + d[4], p1 = d[1] * kNB2 + d[3];
        const f32x4 e2 = d[2] * kNA2 + d[4], p2_ = d[1] * kNA2 + d[3];
        V[0] = d[0] * kA2B2 + (d[2] * kNSUM + d[4]);
        V[1] = p1 * kPA + e1;
        V[2] = p1 * kNA + e1;
        V[3] = p2_ * kPB + e2;
        V[4] = p2_ * kNB + e2;
        V[5] = d[1] * kA2B2 + (d[3] * kNSUM + d[5]);
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < K::NSUB; ++i) {
        const int sr = rr * K::NSUB + i;               // sub-round: weight image i of channel block rr
        const int nsr = (sr + 1) % K::R;
        const int npos = nsr % K::NSUB, nrr = nsr / K::NSUB;
        // MFMAs of the sub-round, order (k-step, channel tile); the next sub-round's LDS reads are
        // spread behind the first ones.  The adapter sub-round feeds four accumulators from the same
        // weights: m0 (x2 at the quad's first pixel), the two middle pixels' own, m5 (the last pixel).
        const bool ad = ADAPT && i == 6;
        const int reps = ad ? 4 : 1;
        int k = 0;
#pragma unroll
        for (int rep = 0; rep < reps; ++rep) {
          const f32x4 b = ad ? raw[rr % K::NS][6 + rep] : V[i];
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int m = 0; m < TM; ++m) {
              f32x4& dst = !ad ? acc[i][m] : rep == 0 ? acc[0][m] : rep == 3 ? acc[5][m] : accx[rep - 1][m];
              dst = mfma16(av[sr & 1][m][s], b[s], dst);
              if (rep == 0 && (k & 1) && k < 2 * TM) {
#if W4_ABLATE & 4
                av[(sr + 1) & 1][k / 2] = av[sr & 1][k / 2];
#else
                av[(sr + 1) & 1][k / 2] = a_frag(npos, k / 2, nrr);
#endif
                __builtin_amdgcn_sched_barrier(0);
              }
              ++k;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    W4_STAMP(stamp_k);
    // ---- output transform y = A^T m:  y_i = m0 [i = 0] + sum_j p_j^i m_j + m5 [i = 3] ----
    f32x4 ay[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m) {
      const f32x4 s12 = acc[1][m] + acc[2][m], d12 = acc[1][m] - acc[2][m];
      const f32x4 s34 = acc[3][m] + acc[4][m], d34 = acc[3][m] - acc[4][m];
      ay[m][0] = (acc[0][m] + s12) + s34;
      ay[m][1] = d12 * PA + d34 * PB;
      ay[m][2] = s12 * PA2 + s34 * PB2;
      ay[m][3] = (d12 * PA3 + d34 * PB3) + acc[5][m];
      if constexpr (ADAPT) {
        ay[m][1] += accx[0][m];
        ay[m][2] += accx[1][m];
      }
    }
    // the accumulators are dead from here: keep the epilogue's operand loads behind this point
    // (hoisted above the transform they cost 64 registers more than the wave has)
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue: lane holds out[pixel n of quad li][co = COW * half + 16m + 4lg .. +3], in chunks of
    // NE pixels of the quad: 4 (one round trip for every operand) where the operand tiles fit beside
    // the next tile's B operands already in flight, 2 for the adapter forms and the 64-channel tiles ----
    const mdil_epilogue& e = a.e;
    constexpr int NE = !EOPS ? 4 : (ADAPT || TM == 4 || STATS) ? W4_NE_EOPS : W4_NE_PLAIN;
    f32x4 bp[BNRED ? TM : 1], bq[BNRED ? TM : 1];       // BNRED: the lane's sums over its quad
#pragma unroll
    for (int n0 = 0; n0 < TN; n0 += NE) {
      long long pb[NE];
#pragma unroll
      for (int n = 0; n < NE; ++n)
        pb[n] = (long long)(okA ? P0A + (n0 + n) * S : 0) * C + half * COW + lg * 4;
      f32x4 r1[NE][TM], r2[NE][TM], r3[NE][TM];     // operand tiles (r3: tail form only)
      auto ld_tile = [&](f32x4 (&r)[NE][TM], const float* p) __attribute__((always_inline)) {
#pragma unroll
        for (int n = 0; n < NE; ++n)
#pragma unroll
          for (int m = 0; m < TM; ++m) r[n][m] = *reinterpret_cast<const f32x4*>(p + pb[n] + m * 16);
      };

      if constexpr (TAIL) {
        // stored value = (acc + res [where res_gate > 0]) where t_gate > 0; no affine part, no ReLU
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
          for (int n = 0; n < NE; ++n)
#pragma unroll
            for (int m = 0; m < TM; ++m) {
              f32x4 x = r1[n][m];
              if (two) {
#pragma unroll
                for (int k = 0; k < 4; ++k) x[k] = r2[n][m][k] > 0.f ? x[k] : 0.f;
              }
              ay[m][n0 + n] += x;
            }
        }
        if (two) {
          ld_tile(r3, a.t_gate);
          ld_tile(r2, a.bn_z);
        }
#pragma unroll
        for (int n = 0; n < NE; ++n)
#pragma unroll
          for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int k = 0; k < 4; ++k) ay[m][n0 + n][k] = r3[n][m][k] > 0.f ? ay[m][n0 + n][k] : 0.f;
      } else {
        const float* opa = EOPS ? (e.res ? e.res : e.gate) : nullptr;
        const float* opb = EOPS ? (BNRED ? a.bn_z : e.res_gate) : nullptr;
        if (opa) ld_tile(r1, opa);
        if (opb) ld_tile(r2, opb);
#pragma unroll
        for (int m = 0; m < TM; ++m) {
          f32x4 vscale, vbias;
          if constexpr (AFFINE) {
            vscale = *reinterpret_cast<const f32x4*>(&Ep[m * 16 + lg * 4]);
            vbias = *reinterpret_cast<const f32x4*>(&Ep[COW + m * 16 + lg * 4]);
          }
#pragma unroll
          for (int n = 0; n < NE; ++n) {
            f32x4 v = ay[m][n0 + n];
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
            ay[m][n0 + n] = v;
          }
        }
      }
      w4_store<C, COW, TM, NE>(a.out, ay, n0, P0A, okA, S, half, li, lg);

      if constexpr (BNRED) {
        // sum g and sum g (z - mean) over the lane's pixels of this chunk (z sits in r2)
#pragma unroll
        for (int m = 0; m < TM; ++m) {
          const f32x4 mu = *reinterpret_cast<const f32x4*>(&Bv[m * 16 + lg * 4]);
          f32x4 ps = ay[m][n0], qs = ay[m][n0] * (r2[0][m] - mu);
#pragma unroll
          for (int n = 1; n < NE; ++n) {
            ps += ay[m][n0 + n];
            qs += ay[m][n0 + n] * (r2[n][m] - mu);
          }
          if (n0 == 0) {
            bp[m] = ps;
            bq[m] = qs;
          } else {
            bp[m] += ps;
            bq[m] += qs;
          }
        }
      }
      if (n0 + NE < TN) __builtin_amdgcn_sched_barrier(0);    // the next chunk's loads stay behind this chunk
    }

    if constexpr (BNRED) {
      // ... over this tile's 64 pixels: the reduce-scatter over the row
      const bool full = (tile + 1) * W4_QUADS <= nquads;       // uniform; false on a ragged last tile only
#pragma unroll
      for (int m = 0; m < TM; ++m) {
        if constexpr (TAIL) {     // the BN branch carries the Dropout2d factor (all four pixels: same image)
          const f32x4 dr = *reinterpret_cast<const f32x4*>(&Dt[imgA * COW + m * 16 + lg * 4]);
          bp[m] *= dr;
          bq[m] *= dr;
        }
        if (!full && !okA) bp[m] = bq[m] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      rA += wc_reduce_scatter<TM>(bp, li);
      rB += wc_reduce_scatter<TM>(bq, li);
    }
    if constexpr (STATS) {
      // Welford / Chan: merge the quad -- count 4, its mean and M2 -- into the lane's running
      // (sn, mean, M2); f = 4 / (sn + 4) from v_rcp_f32 (it scales a DEVIATION from the running mean)
      if (okA) {
        const float nn = sn + 4.f;
        const float f = 4.f * __builtin_amdgcn_rcpf(nn);
        const float nf = sn * f;
#pragma unroll
        for (int m = 0; m < TM; ++m) {
          const f32x4 pm = ((ay[m][0] + ay[m][1]) + (ay[m][2] + ay[m][3])) * 0.25f;
          const f32x4 e0 = ay[m][0] - pm, e1 = ay[m][1] - pm, e2 = ay[m][2] - pm, e3 = ay[m][3] - pm;
          const f32x4 qq = (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
          const f32x4 d = pm - sA[m];
          sA[m] += d * f;
          sB[m] += qq + (d * d) * nf;
        }
        sn = nn;
      }
    }

#if W4_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    W4_STAMP(stamp_k + 1);
    stamp_k += 2;
#endif
    slot += W4_WAVES;
    tile = ntile;
#pragma unroll
    for (int k = 0; k < 3; ++k) vbA[k] = vbB[k];
    P0A = P0B;
    imgA = imgB;
    okA = okB;
  }

  // ---- after the wave's last tile: the wave's summary goes to its strip, the strips are merged in
  // wave order (same layout and finalize hand-off as wconv.hip)
  if constexpr (BNRED) {
    if constexpr (TM == 2) {
      rA += wc_ror<0x128>(rA);
      rB += wc_ror<0x128>(rB);
    }
    if (li < TM * 4) {      // lane li holds channel 4m + k = li of its row's 16-channel groups
      const int c = (li >> 2) * 16 + lg * 4 + (li & 3);
      Sw[c] = rA;
      Sw[64 + c] = rB * Bv[COW + c];
    }
    __syncthreads();
    if (tid < COW) {
      const float* S0 = Ws + K::LDS_FLOATS + 2 * COW;
      float sa = 0.f, sb_ = 0.f;
#pragma unroll
      for (int w = 0; w < W4_WAVES; ++w) {
        sa += S0[w * W4_STAT_LD + tid];
        sb_ += S0[w * W4_STAT_LD + 64 + tid];
      }
      float* r0 = a.stats + ((long long)gq * 2 + 0) * C + half * COW + tid;
      float* r1_ = a.stats + ((long long)gq * 2 + 1) * C + half * COW + tid;
      if (a.fb.ticket) {
        bnfin_st(r0, sa);
        bnfin_st(r1_, sb_);
      } else {
        *r0 = sa;
        *r1_ = sb_;
      }
    }
    if (a.fb.ticket) {       // the last work-group to arrive turns the rows into coefficients (bnfin.h)
      if (bnfin_arrive(a.fb.ticket, gridDim.x, reinterpret_cast<int*>(Ep)))
        bnfin_backward(a.fb, a.stats, nq, C, reinterpret_cast<double*>(Ws));
    }
  }
  if constexpr (STATS) {
    wc_stat_level<0x128, TM>(sn, sA, sB);
    wc_stat_level<0x124, TM>(sn, sA, sB);
    wc_stat_level<0x122, TM>(sn, sA, sB);
    wc_stat_level<0x121, TM>(sn, sA, sB);
    if (li == 0) {
#pragma unroll
      for (int m = 0; m < TM; ++m) {
        *reinterpret_cast<f32x4*>(&Sw[m * 16 + lg * 4]) = sA[m];
        *reinterpret_cast<f32x4*>(&Sw[64 + m * 16 + lg * 4]) = sB[m];
      }
    }
    if (lane == 0) Sw[2 * 64] = sn;
    __syncthreads();
    if (tid < COW) {
      const float* S0 = Ws + K::LDS_FLOATS + 2 * COW;
      float n = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
      for (int w = 0; w < W4_WAVES; ++w)
        welford_merge(n, mean, m2, S0[w * W4_STAT_LD + 2 * 64], S0[w * W4_STAT_LD + tid],
                      S0[w * W4_STAT_LD + 64 + tid]);
      float* r0 = a.stats + ((long long)gq * 2 + 0) * C + half * COW + tid;
      float* r1_ = a.stats + ((long long)gq * 2 + 1) * C + half * COW + tid;
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

int w4conv_queues(long long npix, int NH) {
  const int ntiles = (int)((npix / 4 + W4_QUADS - 1) / W4_QUADS);
  int nq = wc_num_cu() / NH;
  const int need = (ntiles + W4_WAVES - 1) / W4_WAVES;
  if (nq > need) nq = need;
  if (NH > 1) nq = (nq + 7) / 8 * 8;
  return nq;
}

template <int C, int COW, bool ADAPT, int PD, int MODE, bool EOPS>
int launch_w4conv_(const wconv_args& a, hipStream_t st) {
  using K = W4Cfg<C, COW, ADAPT, PD>;
  const int nq = w4conv_queues((long long)a.N * a.H * a.W, K::NH);
  hipLaunchKernelGGL((w4conv_kernel<C, COW, ADAPT, PD, MODE, EOPS>), dim3(nq * K::NH), dim3(W4_THREADS), 0, st, a);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}

#ifndef W4_COW64
#define W4_COW64 64     // output channels per work-group for C = 64 without the adapter (104 KB of weight images)
#endif
#ifndef W4_PD
#define W4_PD 1
#endif

// Output channels per work-group: 32 -- six / seven images of 64 rows do not fit the LDS for C = 128,
// and at C = 64 the adapter's extra accumulators / the epilogue operand tiles do not fit the registers
// beside 96 accumulators.  The plain C = 64 form (bias / folded BN / ReLU only: the most frequent
// C = 64 launch of the step) takes 64: half the input-transform VALU per MFMA.  Every form that
// emits statistics / reduction rows keeps 32, so the row count is one number per geometry.
template <int C, bool ADAPT, int MODE, bool EOPS>
constexpr int w4_cow() { return (C == 64 && !ADAPT && MODE == 0 && !EOPS) ? W4_COW64 : 32; }

template <int C, bool ADAPT, int MODE, bool EOPS>
int launch_w4conv__(const wconv_args& a, hipStream_t st) {
  return launch_w4conv_<C, w4_cow<C, ADAPT, MODE, EOPS>(), ADAPT, W4_PD, MODE, EOPS>(a, st);
}

template <int C, bool ADAPT>
int launch_w4conv(const wconv_args& a, hipStream_t st) {
  const bool eops = a.e.res || a.e.gate || a.e.res_gate;
  if (a.t_gate) return launch_w4conv__<C, ADAPT, 3, true>(a, st);
  if (a.stats && a.bn_z) return launch_w4conv__<C, ADAPT, 2, true>(a, st);
  if (a.stats) return eops ? launch_w4conv__<C, ADAPT, 1, true>(a, st) : launch_w4conv__<C, ADAPT, 1, false>(a, st);
  return eops ? launch_w4conv__<C, ADAPT, 0, true>(a, st) : launch_w4conv__<C, ADAPT, 0, false>(a, st);
}

}  // namespace

bool mdil_w4conv_covers(const mdil_geom* g, int cin, int cout) {
  static const bool off = getenv("MDIL_NO_W4CONV") != nullptr || getenv("MDIL_NO_WCONV") != nullptr;
  if (off) return false;
  // C = 64 with the adapter tap: measured slower than F(2,3) at 64 channels per work-group in every
  // form of the step (32-channel work-groups read each pixel tile twice, and the tail / reduction
  // forms are HBM-bound at this size: profiles/r05_experiments.txt #2)
  if (cin == 64 && g->ntaps == 4) return false;
  return wconv_plan(g, cin, nullptr, 4);
}

// same contract as mdil_wconv (common.h)
int mdil_w4conv(const mdil_geom* g, int cin, const float* in0, const float* in1, const float* wpk,
                const mdil_epilogue* epi, float* out, float* stats, float* stats_count,
                const float* bn_z, const float* bn_mean, const float* bn_invstd, hipStream_t st,
                const float* tail_gate, const float* tail_drop, const BnFinFwd* ff, const BnFinBwd* fb) {
  wconv_args a;
  memset(&a, 0, sizeof(a));
  if (ff && stats && !bn_z) a.ff = *ff;
  if (fb && stats && bn_z) a.fb = *fb;
  if (!wconv_plan(g, cin, &a, 4)) return MDIL_ERR_UNSUPPORTED;
  if (tail_gate && (!stats || !bn_z || !bn_mean || !bn_invstd || epi->gate || epi->relu)) return MDIL_ERR_INVALID;
  if (tail_gate && g->N > W4_TAIL_MAXN) return MDIL_ERR_UNSUPPORTED;
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
    auto lg2 = [](int v) {
      int s = 0;
      while ((1 << s) < v) ++s;
      return (v > 0 && (1 << s) == v) ? s : -1;
    };
    const int L = a.axis ? a.W : a.H;
    const int nb = L / (4 * a.delta);
    a.sh_delta = lg2(a.delta), a.sh_nb = lg2(nb), a.sh_W = lg2(a.W), a.sh_H = lg2(a.H);
    if (a.sh_delta < 0 || a.sh_nb < 0 || a.sh_W < 0 || a.sh_H < 0) a.sh_delta = a.sh_nb = a.sh_W = a.sh_H = -1;
  }
  // C = 64 + adapter is not instantiated: mdil_w4conv_covers sends it to wconv.hip (F(2,3), 64 output channels
  // per work-group), and the six 32-channel forms it would need were dead code -- one of them the only
  // kernel of the library that spilled
  if (cin == 64) return g->ntaps == 3 ? launch_w4conv<64, false>(a, st) : MDIL_ERR_UNSUPPORTED;
  return g->ntaps == 3 ? launch_w4conv<128, false>(a, st) : launch_w4conv<128, true>(a, st);
}

bool mdil_w4conv_tail_covers(const mdil_geom* g) { return g->N <= W4_TAIL_MAXN; }

int mdil_w4conv_stat_blocks(const mdil_geom* g, int cin) {
  const int NH = cin / 32;      // every statistics / reduction form runs 32 output channels per work-group
  return w4conv_queues((long long)g->N * g->HO * g->WO, NH);
}
