// EXPERIMENT -- NOT BUILT, NOT PART OF THE PRODUCT PATH (profiles/r02_bf16_split_probe.txt):
// no faster than sconv.hip on gfx950 and not deterministic with two waves per SIMD.  Kept as the
// record of what was measured.  To build it: copy next to sconv.hip, add it to the Makefile and
// route mdil_sconv / mdil_sconv_stat_blocks to mdil_xconv / mdil_xconv_stat_blocks.
//
// Streaming tap convolution on the bf16 matrix pipe with EXACT fp32 products ("split" kernel) for
// the C -> C (C = 64 / 128) stride-1 convs of the factorised blocks, NHWC fp32, gfx950.
//
// v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 MFMA rate.  An fp32 number is the exact sum of
// three bf16 numbers (24 significand bits = 3 x 8; bf16 has fp32's exponent range, so no scaling is
// needed):  x = hi + mid + lo  with  hi = trunc_bf16(x), mid = trunc_bf16(x - hi),
// lo = x - hi - mid (both subtractions are exact).  The product of two fp32 numbers is then the
// sum of nine bf16 x bf16 products, each of which the matrix pipe forms exactly (8 x 8 bits) and
// adds into an fp32 accumulator.  Nine v_mfma_f32_32x32x16_bf16 per 16-channel block do the work
// of 64 fp32 MFMAs in 9/16 of their time; tensors in memory, accumulators and the epilogue stay
// fp32.  Measured error against fp64 equals the fp32 MFMA's (profiles/r02_bf16_split_probe.txt:
// mean 1.7e-8 vs 2.0e-8 of sum |a||b| at K = 1536).
//
// Structure = sconv.hip: one persistent work-group of 8 waves per CU, weights of all taps resident
// in LDS (already split, in MFMA fragment order: one conflict-free ds_read_b128 per fragment),
// no barrier in the main loop, each wave owns 32-pixel tiles and streams its B operands from
// global memory into registers (out-of-image taps = buffer loads with an out-of-range offset).
// What is new in the loop: the 8 fp32 values a lane loads per 16-channel block are split into
// three packed bf16x8 operands by ~44 VALU instructions, issued one block ahead of their use and
// spread between the MFMAs of the current block (<= 3 per MFMA: they hide in the 32-cycle shadow
// of each MFMA).
#include <stdlib.h>

#include "common.h"

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef XC_NWAVES
#define XC_NWAVES 8
#endif
constexpr int XC_WAVES = XC_NWAVES;
constexpr int XC_THREADS = XC_WAVES * 64;
constexpr int XC_PXT = 32;            // pixels per wave tile (one MFMA column block)
// Hardware note (measured, gfx950 / ROCm 7.2): v_mfma_f32_32x32x16_bf16 can read its A/B source
// registers AFTER the instructions that follow it in program order have executed -- an MFMA that
// waits for its accumulator (same destination as an MFMA a few slots earlier) sits in the matrix
// pipe's queue while the wave runs ahead, and the compiler's hazard tables only cover the
// accumulator.  A source register handed to a new value (a split result, an LDS read) right after
// its last MFMA gave wrong columns 16..31 in ~1 tile of 200, run-to-run different.  Two rules
// keep the kernel safe AND fast:
//   * no MFMA depends on an MFMA fewer than XC_DIST slots earlier (two accumulators per channel
//     block: products with a `lo` piece / the rest, summed in the epilogue), so MFMAs do not queue;
//   * the operands of round r - 1 stay alive (fake use) until XC_KEEP_SLOT MFMAs into round r, and
//     the last round's until a drain pad after the tile's last MFMA.
#ifndef XC_KEEP_SLOT
#define XC_KEEP_SLOT 4
#endif

template <int C, int NTAPS, int COW, int PD>
struct XCfg {
  static constexpr int NH = C / COW;       // work-groups that share a pixel tile (channel parts)
  static constexpr int CT = COW / 32;      // 32-channel MFMA row blocks per wave
  static constexpr int KB = C / 16;        // 16-channel k blocks per tap
  static constexpr int R = NTAPS * KB;     // rounds (k blocks) per output tile
  static constexpr int NS = PD + 1;        // raw-operand ring: PD rounds in flight + the one being split
  static constexpr int LDS_W = NTAPS * CT * KB * 3 * 1024;   // bytes of split weights
  static_assert(R % 4 == 0 && R % NS == 0 && PD >= 1, "ring / operand buffers must divide a tile");
};

struct xconv_args {
  const float* in0;
  const float* in1;
  const float* wpk;      // [tap][C][C] fp32 (the image sconv.hip / tapconv.hip use)
  float* out;
  mdil_epilogue e;
  int N, H, W;
  int dh[4], dw[4], src[4];
  float* stats;
  float* stats_count;
  const float* bn_z;
  const float* bn_mean;
  const float* bn_invstd;
};

typedef const u32x4 __attribute__((address_space(3))) * lds_u4_ptr;
__device__ __forceinline__ u32x4 lds_ld4(unsigned addr) { return *(lds_u4_ptr)(__SIZE_TYPE__)addr; }
constexpr unsigned XC_WIN = 61440;   // LDS window stride (ds_read immediates stay below 65536)

__device__ __forceinline__ f32x4 xbuf_load(const __amdgpu_buffer_rsrc_t r, unsigned voff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, 0, 0);
  return __builtin_bit_cast(f32x4, v);
}

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ unsigned hi16_pair(float x1, float x0) {   // {bf16 trunc(x1), bf16 trunc(x0)}
  return __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u);
}
__device__ __forceinline__ float rem16(float x) {                     // x - trunc_bf16(x), exact
  return x - __uint_as_float(__float_as_uint(x) & 0xffff0000u);
}

// whole split of 8 values (used outside the main loop)
__device__ __forceinline__ void split8(const f32x4 x0, const f32x4 x1, u32x4& hi, u32x4& mid, u32x4& lo) {
  const float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    hi[p] = hi16_pair(x[2 * p + 1], x[2 * p]);
    const float r0 = rem16(x[2 * p]), r1 = rem16(x[2 * p + 1]);
    mid[p] = hi16_pair(r1, r0);
    lo[p] = hi16_pair(rem16(r1), rem16(r0));
  }
}

constexpr int XC_STAT_LD = 2 * 64 + 4;   // per-wave statistics strip: a[COW], b[COW], count (COW <= 64)

template <int C, int NTAPS, int COW, int PD, int MODE, bool EOPS>
__global__ __launch_bounds__(XC_THREADS) void xconv_kernel(const xconv_args a) {
  using K = XCfg<C, NTAPS, COW, PD>;
  constexpr int CT = K::CT;
  constexpr int LDS_W_F = K::LDS_W / 4;
  __shared__ __attribute__((aligned(16)))
  float Ws[LDS_W_F + 2 * COW + (MODE ? XC_WAVES * XC_STAT_LD + 2 * COW : 0)];
  constexpr bool STATS = MODE == 1;
  constexpr bool BNRED = MODE == 2;
  float* Ep = Ws + LDS_W_F;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int px = lane & 31, kh = lane >> 5;
  const int H = a.H, W = a.W;
  const int hw = H * W;
  const int npix = a.N * hw;
  const int ntiles = (npix + XC_PXT - 1) / XC_PXT;

  // work-group -> (channel part, pixel-tile queue); the NH work-groups that share a pixel tile
  // differ only in blockIdx bits 3.. (same XCD = blockIdx % 8: the later readers hit its L2)
  int part = 0, gq = blockIdx.x, nq = gridDim.x;
  if constexpr (K::NH > 1) {
    part = (blockIdx.x >> 3) % K::NH;
    gq = (blockIdx.x & 7) | ((blockIdx.x / (8 * K::NH)) << 3);
    nq = gridDim.x / K::NH;
  }
  const int co0 = part * COW;            // first output channel of this work-group

  // ---- weights -> split -> LDS in fragment order (once) ----
  // block (t, ct, kb, piece) = 64 lanes x 16 bytes; lane (row = lane & 31, kh = lane >> 5) holds the
  // 8 bf16 pieces of W[t][co0 + 32 ct + row][16 kb + 4 kh + {0..3}, 16 kb + 8 + 4 kh + {0..3}]
  // (the k order inside a block is the one the B loads use: two 16-byte loads per lane)
  {
    constexpr int ITEMS = NTAPS * CT * K::KB * 64;
    static_assert(ITEMS % XC_THREADS == 0, "weight image divides over the work-group");
    constexpr int PER = ITEMS / XC_THREADS;
    f32x4 w0[PER], w1[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int idx = tid + u * XC_THREADS;
      const int l = idx & 63, blk = idx >> 6;
      const int kb = blk % K::KB, ct = (blk / K::KB) % CT, t = blk / (K::KB * CT);
      const float* src = a.wpk + ((long long)(t * C + co0 + 32 * ct + (l & 31))) * C + 16 * kb + 4 * (l >> 5);
      w0[u] = *reinterpret_cast<const f32x4*>(src);
      w1[u] = *reinterpret_cast<const f32x4*>(src + 8);
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int idx = tid + u * XC_THREADS;
      const int l = idx & 63, blk = idx >> 6;
      u32x4 hi, mid, lo;
      split8(w0[u], w1[u], hi, mid, lo);
      u32x4* dst = reinterpret_cast<u32x4*>(Ws) + (blk * 3) * 64 + l;
      dst[0] = hi;
      dst[64] = mid;
      dst[128] = lo;
    }
  }

  if (tid < COW) {
    const int co = co0 + tid;
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
  constexpr unsigned OOB = 0x80000000u;

  int tap_off[NTAPS];
#pragma unroll
  for (int t = 0; t < NTAPS; ++t) tap_off[t] = (a.dh[t] * W + a.dw[t]) * C * 4;
  // per (tile, tap): byte offset of the lane's first 16 bytes (pixel px shifted by the tap,
  // channels 4 kh ..), or an out-of-range offset: a refill is one instruction, no address VALU
  auto setup = [&](int tile, unsigned (&vb)[NTAPS]) {
    const int P = tile * XC_PXT + px;
    const bool ok = tile < ntiles && P < npix;
    const int Pc = ok ? P : 0;
    const int img = Pc / hw;
    const int rem = Pc - img * hw;
    const int h = rem / W;
    const int w = rem - h * W;
    const unsigned base = (unsigned)Pc * (unsigned)(C * 4) + (unsigned)kh * 16u;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) {
      const int hh = h + a.dh[t], ww = w + a.dw[t];
      const bool v = ok && hh >= 0 && hh < H && ww >= 0 && ww < W;
      vb[t] = v ? base + (unsigned)tap_off[t] : OOB;
    }
  };

  unsigned vbA[NTAPS], vbB[NTAPS];
  f32x4 raw[K::NS][2];       // fp32 B operands as loaded (ring)
  // split operands: [round % 4][hi, mid, lo].  Four name slots, up to three live (see XC_KEEP_SLOT)
  u32x4 bs[4][3];            // B operands (pixels)
  u32x4 av[4][CT][3];        // A operands (weights) from LDS
  u32x4 atmp[CT][3];         // XC_ACOPY staging
  f32x16 accS[CT], accB[CT];  // products with a lo piece / hi-mid products (summed in the epilogue)
  f32x16 acc[CT];

  float st_n = 0.f;
  float* Sw = Ws + LDS_W_F + 2 * COW + wave * XC_STAT_LD;
  float* Bv = Ws + LDS_W_F + 2 * COW + XC_WAVES * XC_STAT_LD;
  if constexpr (MODE != 0) {
    Sw[lane] = 0.f;
    Sw[64 + lane] = 0.f;
  }
  if constexpr (BNRED) {
    if (tid < COW) {
      Bv[tid] = a.bn_mean[co0 + tid];
      Bv[COW + tid] = a.bn_invstd[co0 + tid];
    }
  }

  unsigned wbase[3];
  {
    const unsigned b = (unsigned)(__SIZE_TYPE__)((__attribute__((address_space(3))) float*)Ws) + (unsigned)lane * 16u;
#pragma unroll
    for (int w = 0; w < 3; ++w) {
      wbase[w] = b + w * XC_WIN;
      asm volatile("" : "+v"(wbase[w]));
    }
  }
  auto a_frag = [&](int t, int ct, int kb, int piece) __attribute__((always_inline)) {
    const unsigned off = (unsigned)(((((t * CT + ct) * K::KB + kb) * 3) + piece) * 1024);
    return lds_ld4(wbase[off / XC_WIN] + off % XC_WIN);
  };
  auto b_load = [&](int rr, int which, const unsigned (&vb)[NTAPS]) __attribute__((always_inline)) {
    const int t = rr / K::KB, kb = rr % K::KB;
    return xbuf_load(a.src[t] ? rs1 : rs0, vb[t] + kb * 64 + which * 32);
  };

  int slot = wave;
  int tile = slot * nq + gq;
  setup(tile, vbA);
#pragma unroll
  for (int r = 0; r <= PD; ++r) {
    raw[r][0] = b_load(r, 0, vbA);
    raw[r][1] = b_load(r, 1, vbA);
  }
  __syncthreads();   // the only barrier: weights are resident from here on
  split8(raw[0][0], raw[0][1], bs[0][0], bs[0][1], bs[0][2]);
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#ifdef XC_ACOPY
      atmp[ct][p] = a_frag(0, ct, 0, p);
#else
      av[0][ct][p] = a_frag(0, ct, 0, p);
#endif
    }

  while (tile < ntiles) {
    const int ntile = (slot + XC_WAVES) * nq + gq;
    setup(ntile, vbB);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int g = 0; g < 16; ++g) accS[ct][g] = accB[ct][g] = 0.f;

    // epilogue operands (residual / gates / BN input), addressed like the output: lane holds
    // out[pixel px][co0 + 32 ct + 8 j + 4 kh .. +3], j = 0..3
    const mdil_epilogue& e = a.e;
    const int P = tile * XC_PXT + px;
    const bool okp = P < npix;
    const long long pb = (long long)(okp ? P : 0) * C + co0 + 4 * kh;
    f32x4 ra[CT][4], rb[CT][4];
    const float* opa = EOPS ? (e.res ? e.res : e.gate) : nullptr;
    const float* opb = EOPS ? (BNRED ? a.bn_z : e.res_gate) : nullptr;
    // 32-channel tiles: ra is requested two rounds before the tile's last MFMA (in registers when
    // the epilogue starts).  Everything else at the epilogue itself: the register file has no room
    // to carry 32-64 more values through the main loop (the partner wave's MFMAs cover the wait)
    auto epilogue_loads_a = [&]() __attribute__((always_inline)) {
      if (opa) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int j = 0; j < 4; ++j) ra[ct][j] = *reinterpret_cast<const f32x4*>(opa + pb + 32 * ct + 8 * j);
      }
    };
    auto epilogue_loads_b = [&]() __attribute__((always_inline)) {
      if (opb) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int j = 0; j < 4; ++j) rb[ct][j] = *reinterpret_cast<const f32x4*>(opb + pb + 32 * ct + 8 * j);
      }
    };

    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < K::R; ++r) {
      if constexpr (EOPS) {
        if (CT == 1 && r == K::R - 2) epilogue_loads_a();
      }
      const int cur = r & 3, nxt = (r + 1) & 3, prv = (r + 3) & 3;
#ifdef XC_ACOPY
      if (r > 0 || true) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
          for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(av[cur][ct][p][i]) : "v"(atmp[ct][p][i]));
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
      const int rn = (r + 1) % K::R;                 // round prepared during this one (next tile's 0 at the end)
      const int rl = r + 1 + PD;                     // round whose raw operands are loaded now
      float rr[8];                                   // remainders carried between split steps
      const f32x4 x0 = raw[(r + 1) % K::NS][0], x1 = raw[(r + 1) % K::NS][1];
      const float xs[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
      // side work of the round, in issue order: 8 split steps, 3 CT fragment reads, 2 refill loads
      constexpr int NSIDE = 8 + 3 * CT + 2;
      auto side = [&](int s) __attribute__((always_inline)) {
        if (s < 8) {
          const int p = s >> 1;
          if ((s & 1) == 0) {
            bs[nxt][0][p] = hi16_pair(xs[2 * p + 1], xs[2 * p]);
            rr[2 * p] = rem16(xs[2 * p]);
            rr[2 * p + 1] = rem16(xs[2 * p + 1]);
          } else {
            bs[nxt][1][p] = hi16_pair(rr[2 * p + 1], rr[2 * p]);
            bs[nxt][2][p] = hi16_pair(rem16(rr[2 * p + 1]), rem16(rr[2 * p]));
          }
        } else if (s < 8 + 3 * CT) {
          const int q = s - 8;
#ifdef XC_ACOPY
          // A fragments land in staging registers (previous round's slot); the MFMA operand
          // registers are only ever written by VALU moves
          atmp[q / 3][q % 3] = a_frag(rn / K::KB, q / 3, rn % K::KB, q % 3);
#else
          av[nxt][q / 3][q % 3] = a_frag(rn / K::KB, q / 3, rn % K::KB, q % 3);
#endif
        } else {
          const int which = s - 8 - 3 * CT;
          raw[rl % K::NS][which] = rl < K::R ? b_load(rl, which, vbA) : b_load(rl - K::R, which, vbB);
        }
      };
      // MFMAs of the round.  Products: 0 ll, 1 ml, 2 lm, 3 hl, 4 lh (-> accS), 5 mm, 6 hm, 7 mh,
      // 8 hh (-> accB); issue order S, B, S, B, ... with the CT channel blocks interleaved, so an
      // accumulator is touched every 2 CT MFMAs at most
      constexpr int PA[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0};   // piece of A: 0 hi, 1 mid, 2 lo
      constexpr int PB[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0};   // piece of B
      constexpr int ORD[9] = {0, 5, 1, 6, 2, 7, 3, 8, 4};
      constexpr int NM = 9 * CT;
      constexpr int PER = (NSIDE + NM - 1) / NM;            // side items per MFMA slot
      int s = 0;
#pragma unroll
      for (int k = 0; k < NM; ++k) {
        const int pr = ORD[k / CT], ct = k % CT;
#ifdef XC_PRENOP
        asm volatile("s_nop 7");
#endif
#ifdef XC_CHAIN
        if (false)
#else
        if (pr < 5)
#endif
          accS[ct] = mfma_bf16(av[cur][ct][PA[pr]], bs[cur][PB[pr]], accS[ct]);
        else
          accB[ct] = mfma_bf16(av[cur][ct][PA[pr]], bs[cur][PB[pr]], accB[ct]);
#pragma unroll
        for (int u = 0; u < PER; ++u)
          if (s < NSIDE) side(s++);
        if (r > 0 && k == XC_KEEP_SLOT) {
          // the previous round's operands have been left alone for XC_KEEP_SLOT MFMAs: release
#pragma unroll
          for (int p = 0; p < 3; ++p) {
            asm volatile("" ::"v"(bs[prv][p]));
#pragma unroll
            for (int c2 = 0; c2 < CT; ++c2) asm volatile("" ::"v"(av[prv][c2][p]));
          }
        }
#ifndef XC_NOFENCE
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
    }
    // drain pad: the last round's operands stay untouched while its MFMAs leave the queue
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15");
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      asm volatile("" ::"v"(bs[(K::R - 1) & 3][p]));
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) asm volatile("" ::"v"(av[(K::R - 1) & 3][ct][p]));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[ct] = accS[ct] + accB[ct];
    if constexpr (EOPS) {
      if (CT > 1) epilogue_loads_a();
      epilogue_loads_b();
    }

    // ---- epilogue ----
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const f32x4 vscale = *reinterpret_cast<const f32x4*>(&Ep[32 * ct + 8 * j + 4 * kh]);
        const f32x4 vbias = *reinterpret_cast<const f32x4*>(&Ep[COW + 32 * ct + 8 * j + 4 * kh]);
        f32x4 v = {acc[ct][4 * j], acc[ct][4 * j + 1], acc[ct][4 * j + 2], acc[ct][4 * j + 3]};
        v = v * vscale + vbias;
        if (EOPS && e.res) {
          f32x4 x = ra[ct][j];
          if (e.res_gate) {
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = rb[ct][j][k] > 0.f ? x[k] : 0.f;
          }
          v += x;
        }
        if (e.relu) {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        if (EOPS && e.gate && !e.res) {
#pragma unroll
          for (int k = 0; k < 4; ++k) v[k] = ra[ct][j][k] > 0.f ? v[k] : 0.f;
        }
#ifdef XC_NONT
        if (okp) *reinterpret_cast<f32x4*>(a.out + pb + 32 * ct + 8 * j) = v;
#else
        if (okp) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(a.out + pb + 32 * ct + 8 * j));
#endif
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[ct][4 * j + k] = v[k];   // kept for the statistics below
        __builtin_amdgcn_sched_barrier(0);                        // one 4-channel group at a time (registers)
      }
    }

    if constexpr (BNRED) {
      // sum(g), sum(g * xhat) of the stored gradient per channel: 32 pixels of a channel sit in
      // the 32 lanes with the same kh -> butterfly over lane bits 0..4, then the wave's LDS strip
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = 32 * ct + 8 * j + 4 * kh;
          const f32x4 mu = *reinterpret_cast<const f32x4*>(&Bv[c]);
          const f32x4 is = *reinterpret_cast<const f32x4*>(&Bv[COW + c]);
          f32x4 sa, sb;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float gk = okp ? acc[ct][4 * j + k] : 0.f;
            sa[k] = gk;
            sb[k] = gk * ((rb[ct][j][k] - mu[k]) * is[k]);
#pragma unroll
            for (int d = 1; d < 32; d <<= 1) {
              sa[k] += __shfl_xor(sa[k], d, 64);
              sb[k] += __shfl_xor(sb[k], d, 64);
            }
          }
          if (px == 0) {
            *reinterpret_cast<f32x4*>(&Sw[c]) = *reinterpret_cast<const f32x4*>(&Sw[c]) + sa;
            *reinterpret_cast<f32x4*>(&Sw[64 + c]) = *reinterpret_cast<const f32x4*>(&Sw[64 + c]) + sb;
          }
        }
    }

    if constexpr (STATS) {
      // train-mode BatchNorm statistics of the stored values: tile mean, then squared deviations
      // (two passes over registers), Chan-merged into the wave's running (count, mean, M2)
      const int nvalid = min(XC_PXT, npix - tile * XC_PXT);
      const float inv = 1.f / (float)nvalid;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = 32 * ct + 8 * j + 4 * kh;
          f32x4 mean, q;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float sv = okp ? acc[ct][4 * j + k] : 0.f;
#pragma unroll
            for (int d = 1; d < 32; d <<= 1) sv += __shfl_xor(sv, d, 64);
            mean[k] = sv * inv;
            const float dv = acc[ct][4 * j + k] - mean[k];
            float qv = okp ? dv * dv : 0.f;
#pragma unroll
            for (int d = 1; d < 32; d <<= 1) qv += __shfl_xor(qv, d, 64);
            q[k] = qv;
          }
          if (px == 0) {
            f32x4 om = *reinterpret_cast<const f32x4*>(&Sw[c]);
            f32x4 oq = *reinterpret_cast<const f32x4*>(&Sw[64 + c]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              float nn = st_n, mm = om[k], qq = oq[k];
              welford_merge(nn, mm, qq, (float)nvalid, mean[k], q[k]);
              om[k] = mm;
              oq[k] = qq;
            }
            *reinterpret_cast<f32x4*>(&Sw[c]) = om;
            *reinterpret_cast<f32x4*>(&Sw[64 + c]) = oq;
          }
        }
      st_n += (float)nvalid;
    }

    slot += XC_WAVES;
    tile = ntile;
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) vbA[t] = vbB[t];
  }

  if constexpr (BNRED) {
    __syncthreads();
    if (tid < COW) {
      const float* S0 = Ws + LDS_W_F + 2 * COW;
      float sa = 0.f, sb = 0.f;
#pragma unroll
      for (int w = 0; w < XC_WAVES; ++w) {      // wave order: fixed => deterministic
        sa += S0[w * XC_STAT_LD + tid];
        sb += S0[w * XC_STAT_LD + 64 + tid];
      }
      a.stats[((long long)gq * 2 + 0) * C + co0 + tid] = sa;
      a.stats[((long long)gq * 2 + 1) * C + co0 + tid] = sb;
    }
  }
  if constexpr (STATS) {
    if (lane == 0) Sw[2 * 64] = st_n;
    __syncthreads();
    if (tid < COW) {
      const float* S0 = Ws + LDS_W_F + 2 * COW;
      float n = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
      for (int w = 0; w < XC_WAVES; ++w)
        welford_merge(n, mean, m2, S0[w * XC_STAT_LD + 2 * 64], S0[w * XC_STAT_LD + tid],
                      S0[w * XC_STAT_LD + 64 + tid]);
      a.stats[((long long)gq * 2 + 0) * C + co0 + tid] = mean;
      a.stats[((long long)gq * 2 + 1) * C + co0 + tid] = m2;
      if (tid == 0 && part == 0) a.stats_count[gq] = n;
    }
  }
}

int xc_num_cu() {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    n_cu = n;
  }
  return n_cu;
}

// channel parts per pixel tile of a configuration (LDS capacity decides: split weights take 6 bytes
// per element; 64 channels x 128 x 4 taps would need 192 KB)
constexpr int xc_cow(int C, int ntaps) { return (C == 128 && ntaps == 4) ? 32 : 64; }

int xconv_queues(long long npix, int C, int ntaps) {
  const int NH = C / xc_cow(C, ntaps);
  const int ntiles = (int)((npix + XC_PXT - 1) / XC_PXT);
  int nq = xc_num_cu() / NH;
  if (getenv("MDIL_XC_ONE_TILE")) nq = 1 << 20;       // debugging: one tile per wave at most
  const int need = (ntiles + XC_WAVES - 1) / XC_WAVES;
  if (nq > need) nq = need;
  if (NH > 1) nq = (nq + 7) / 8 * 8;        // the part bits sit above the XCD bits of blockIdx
  return nq;
}

template <int C, int NTAPS, int PD, int MODE, bool EOPS>
int launch_xconv_(const xconv_args& a, hipStream_t st) {
  constexpr int COW = xc_cow(C, NTAPS);
  using K = XCfg<C, NTAPS, COW, PD>;
  const int nq = xconv_queues((long long)a.N * a.H * a.W, C, NTAPS);
  hipLaunchKernelGGL((xconv_kernel<C, NTAPS, COW, PD, MODE, EOPS>), dim3(nq * K::NH), dim3(XC_THREADS), 0, st, a);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}

template <int C, int NTAPS, int PD>
int launch_xconv(const xconv_args& a, hipStream_t st) {
  const bool eops = a.e.res || a.e.gate || a.e.res_gate;
  if (a.stats && a.bn_z) return launch_xconv_<C, NTAPS, PD, 2, true>(a, st);
  if (a.stats)
    return eops ? launch_xconv_<C, NTAPS, PD, 1, true>(a, st) : launch_xconv_<C, NTAPS, PD, 1, false>(a, st);
  return eops ? launch_xconv_<C, NTAPS, PD, 0, true>(a, st) : launch_xconv_<C, NTAPS, PD, 0, false>(a, st);
}

}  // namespace

int mdil_xconv_stat_blocks(const mdil_geom* g, int cin) {
  return xconv_queues((long long)g->N * g->HO * g->WO, cin, g->ntaps);
}

// same contract as mdil_sconv (the caller has checked coverage and the epilogue combination)
int mdil_xconv(const mdil_geom* g, int cin, const float* in0, const float* in1, const float* wpk,
               const mdil_epilogue* epi, float* out, float* stats, float* stats_count,
               const float* bn_z, const float* bn_mean, const float* bn_invstd, hipStream_t st) {
  xconv_args a;
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
  for (int t = 0; t < g->ntaps; ++t) {
    a.dh[t] = g->dh[t];
    a.dw[t] = g->dw[t];
    a.src[t] = g->src[t];
  }
#ifndef XC_PD
#define XC_PD 3
#endif
  if (cin == 64) {
    if (g->ntaps == 3) return launch_xconv<64, 3, XC_PD>(a, st);
    return launch_xconv<64, 4, XC_PD>(a, st);
  }
  if (g->ntaps == 3) return launch_xconv<128, 3, XC_PD>(a, st);
  return launch_xconv<128, 4, XC_PD>(a, st);
}
