// Probe (round 6): can the C = 128 F(4,3) conv keep its WEIGHTS IN REGISTERS and share the
// transformed pixel operands of a tile between the four waves of a work-group through LDS?
//
// The register file of a CU (512 KB) is the only on-CU storage that holds the whole transformed weight set
// of a C = 128 3-tap conv (6 images x 128 x 128 x 4 B = 393 KB; the LDS holds 160 KB, which is why w4conv.hip
// runs 32 output channels per work-group and every wave loads and transforms its own pixel operands: 6
// buffer loads + 24 v_pk_fma_f32 per 48 MFMAs = 8 of the main loop's 40 cycles per MFMA).  Shape probed here:
//   * work-group = 4 waves, one per SIMD, 512 VGPRs each; wave w keeps the A fragments of output channels
//     [32w, 32w+32) for all six Winograd positions and all eight 16-channel blocks: 6 x 2 x 8 x 4 = 384 registers;
//   * a tile's (16 quads) pixel operands are loaded and transformed ONCE per work-group -- wave w does channel
//     blocks 2w, 2w+1: 12 loads + 48 v_pk_fma_f32 + 12 ds_write_b128 per tile instead of 48 + 192 per wave
//     tile -- into a double-buffered LDS image [2][8 blocks][6 positions][64 lanes] x 16 B = 96 KB that all four
//     waves read back in fragment order (48 ds_read_b128 per tile, lane-private slots: conflict free);
//   * one work-group barrier per tile.
// What the probe answers: does hipcc allocate it without scratch / accvgpr copies, and how many cycles per
// MFMA does a wave that owns its SIMD need for the loop (s_memtime per tile)?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/regweights_probe.hip -o /tmp/rwp && /tmp/rwp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifndef RW_LDS_BLOCKS
#define RW_LDS_BLOCKS 0   // A fragments of the LAST n channel blocks from LDS instead of registers (48 registers each)
#endif
#ifndef RW_AGPR_BLOCKS
#define RW_AGPR_BLOCKS 5  // A fragments of the FIRST n channel blocks are pinned in AccVGPRs ("a" operands of hand-written
#endif                    // MFMA statements: left to itself hipcc copies them back with one v_accvgpr_read per use)
constexpr int NB = 8;                         // 16-channel blocks (C = 128)
constexpr int NREG = NB - RW_LDS_BLOCKS;      // ... whose A fragments sit in registers
constexpr int NAG = RW_AGPR_BLOCKS;           // ... of which in AccVGPRs (48 each)

// D = A * B + D with the A operand in an AccVGPR / a VGPR; the accumulator stays in VGPRs
#define MFMA_A(acc, a, b) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b))
#define MFMA_V(acc, a, b) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))

__device__ __forceinline__ f32x4 ldg(const __amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0));
}

// MODE bit 0: no refill (loads / transform / LDS writes of the next tile), bit 1: no epilogue
template <int MODE>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ wpk, const float* __restrict__ x,
                                             float* __restrict__ out, long long* __restrict__ cyc, int ntiles,
                                             int tiles_total) {
  __shared__ __attribute__((aligned(16))) f32x4 Vs[2][NB][6][64];
  __shared__ __attribute__((aligned(16))) f32x4 As[RW_LDS_BLOCKS ? RW_LDS_BLOCKS : 1][4][6][2][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wpk), 0, 6 * 128 * 128 * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, tiles_total * NB * 6 * 64 * 16, 0x00020000);
  // ---- weights -> registers (fragment order, lane-contiguous 16-byte loads), block by block: the MFMAs of
  // block rr only wait for the loads of blocks <= rr (loads return in order)
  float Aa[NAG ? NAG : 1][6][2][4];               // AccVGPR-resident
  f32x4 A[NREG - NAG ? NREG - NAG : 1][6][2];     // VGPR-resident
#pragma unroll
  for (int rr = 0; rr < NREG; ++rr)
#pragma unroll
    for (int p = 0; p < 6; ++p)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const f32x4 t = ldg(rw, (unsigned)lane * 16u, ((((wave * NB + rr) * 6 + p) * 2 + m) * 64) * 16);
        if (rr < NAG) {
#pragma unroll
          for (int s = 0; s < 4; ++s) asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(Aa[rr < NAG ? rr : 0][p][m][s]) : "v"(t[s]));
        } else {
          A[rr >= NAG ? rr - NAG : 0][p][m] = t;
        }
      }
  if constexpr (RW_LDS_BLOCKS > 0) {
    for (int i = tid; i < RW_LDS_BLOCKS * 4 * 6 * 2 * 64; i += 256) {
      const int l = i & 63, m = (i >> 6) & 1, p = (i >> 7) % 6, w = (i / (64 * 2 * 6)) & 3, b = i / (64 * 2 * 6 * 4);
      (&As[0][0][0][0][0])[i] = ldg(rw, (unsigned)l * 16u, ((((w * NB + NREG + b) * 6 + p) * 2 + m) * 64) * 16);
    }
  }
  const int tile0 = blockIdx.x * ntiles;
  // raw operands of one channel block of a tile: 6 lane-contiguous vectors (the real kernel addresses pixels
  // through the same buffer loads with scalar offsets)
  auto load_raw = [&](f32x4(&d)[6], int tile, int rr) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < 6; ++k) d[k] = ldg(rx, (unsigned)lane * 16u, (((tile * NB + rr) * 6 + k) * 64) * 16);
  };
  float kNB2 = -2.25f, kNA2 = -0.5625f, kA2B2 = 1.265625f, kNSUM = -2.8125f, kPA = 0.75f, kNA = -0.75f, kPB = 1.5f, kNB = -1.5f;
  asm volatile("" : "+s"(kNB2), "+s"(kNA2), "+s"(kA2B2), "+s"(kNSUM), "+s"(kPA), "+s"(kNA), "+s"(kPB), "+s"(kNB));
  auto transform_store = [&](const f32x4(&d)[6], int buf, int rr) __attribute__((always_inline)) {
    const f32x4 e1 = d[2] * kNB2 + d[4], p1 = d[1] * kNB2 + d[3];
    const f32x4 e2 = d[2] * kNA2 + d[4], p2 = d[1] * kNA2 + d[3];
    Vs[buf][rr][0][lane] = d[0] * kA2B2 + (d[2] * kNSUM + d[4]);
    Vs[buf][rr][1][lane] = p1 * kPA + e1;
    Vs[buf][rr][2][lane] = p1 * kNA + e1;
    Vs[buf][rr][3][lane] = p2 * kPB + e2;
    Vs[buf][rr][4][lane] = p2 * kNB + e2;
    Vs[buf][rr][5][lane] = d[1] * kA2B2 + (d[3] * kNSUM + d[5]);
  };
  {   // tile 0 of this work-group
    f32x4 d[6];
    load_raw(d, tile0, 2 * wave);
    transform_store(d, 0, 2 * wave);
    load_raw(d, tile0, 2 * wave + 1);
    transform_store(d, 0, 2 * wave + 1);
  }
  __syncthreads();
  long long t_prev = clock64();
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    f32x4 acc[6][2];
#pragma unroll
    for (int p = 0; p < 6; ++p) acc[p][0] = acc[p][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 raw[6];
    const bool more = t + 1 < ntiles;
    if (!(MODE & 1) && more) load_raw(raw, tile0 + t + 1, 2 * wave);
    f32x4 v[6];
#pragma unroll
    for (int p = 0; p < 6; ++p) v[p] = Vs[buf][0][p][lane];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rr = 0; rr < NB; ++rr) {
#pragma unroll
      for (int p = 0; p < 6; ++p) {
        const f32x4 b = v[p];
        if (rr < NAG) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            MFMA_A(acc[p][0], Aa[rr < NAG ? rr : 0][p][0][s], b[s]);
            MFMA_A(acc[p][1], Aa[rr < NAG ? rr : 0][p][1][s], b[s]);
          }
        } else {
          f32x4 a0, a1;
          if (rr < NREG) {
            a0 = A[(rr >= NAG && rr < NREG) ? rr - NAG : 0][p][0];
            a1 = A[(rr >= NAG && rr < NREG) ? rr - NAG : 0][p][1];
          } else {
            a0 = As[rr - NREG][wave][p][0][lane];
            a1 = As[rr - NREG][wave][p][1][lane];
          }
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            MFMA_V(acc[p][0], a0[s], b[s]);
            MFMA_V(acc[p][1], a1[s], b[s]);
          }
        }
        // the operand of this position for the NEXT block goes into the registers just consumed
        if (rr + 1 < NB) v[p] = Vs[buf][rr + 1][p][lane];
        __builtin_amdgcn_sched_barrier(0);
      }
      // this wave's share of the next tile's operands: block 2w behind channel block 2, block 2w+1 behind block 5
      if (!(MODE & 1) && more && rr == 2) {
        transform_store(raw, buf ^ 1, 2 * wave);
        load_raw(raw, tile0 + t + 1, 2 * wave + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!(MODE & 1) && more && rr == 5) {
        transform_store(raw, buf ^ 1, 2 * wave + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");   // hand-written MFMAs: their results are read below
    if (!(MODE & 2)) {
      // output transform y = A^T m + bias / ReLU + stores: the epilogue of the plain 3-tap form
      float* o = out + ((long long)(tile0 + t) * 4 + wave) * (64 * 32);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const f32x4 s12 = acc[1][m] + acc[2][m], d12 = acc[1][m] - acc[2][m];
        const f32x4 s34 = acc[3][m] + acc[4][m], d34 = acc[3][m] - acc[4][m];
        f32x4 y[4];
        y[0] = (acc[0][m] + s12) + s34;
        y[1] = d12 * 0.75f + d34 * 1.5f;
        y[2] = s12 * 0.5625f + s34 * 2.25f;
        y[3] = (d12 * 0.421875f + d34 * 3.375f) + acc[5][m];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          f32x4 r = y[n] + 0.125f;
#pragma unroll
          for (int k = 0; k < 4; ++k) r[k] = fmaxf(r[k], 0.f);
          __builtin_nontemporal_store(r, reinterpret_cast<f32x4*>(o + ((n * 2 + m) * 64 + lane) * 4));
        }
      }
    } else {
      float sink = 0.f;
#pragma unroll
      for (int p = 0; p < 6; ++p) sink += acc[p][0][0] + acc[p][1][3];
      if (sink == 12345.678f) out[tid] = sink;
    }
    __syncthreads();
    if (lane == 0) {
      const long long now = clock64();
      cyc[((long long)blockIdx.x * 4 + wave) * ntiles + t] = now - t_prev;
      t_prev = now;
    }
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
void run(const char* name, const float* w, const float* x, float* out, long long* cyc, int ntiles, int nwg) {
  const int total = nwg * ntiles;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe<MODE>, dim3(nwg), dim3(256), 0, 0, w, x, out, cyc, ntiles, total);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(probe<MODE>, dim3(nwg), dim3(256), 0, 0, w, x, out, cyc, ntiles, total);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> h((size_t)nwg * 4 * ntiles);
  CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> per;
  for (size_t i = 0; i < h.size(); ++i)
    if ((int)(i % ntiles) > 0) per.push_back((double)h[i] / 384.0);     // steady-state tiles
  std::sort(per.begin(), per.end());
  const double us = ms * 1e3 / reps;
  const double flop = 2.0 * 16 * 16 * 4 * 384.0 * 4 * nwg * ntiles;       // executed MFMA flop
  printf("%-44s %7.2f us / launch  %6.1f TFLOP/s executed (%4.1f %% of 157.3)  cycles per MFMA of a tile: p10 %.1f median %.1f p90 %.1f\n",
         name, us, flop / us / 1e6, flop / us / 1e6 / 157.3 * 100, per[per.size() / 10], per[per.size() / 2],
         per[per.size() * 9 / 10]);
}

int main(int argc, char** argv) {
  const int nwg = 256, ntiles = argc > 1 ? atoi(argv[1]) : 3;          // 3 tiles of 64 pixels per work-group = N = 6
  const size_t wn = 6 * 128 * 128, xn = (size_t)nwg * ntiles * NB * 6 * 64 * 4, on = (size_t)nwg * ntiles * 4 * 64 * 32;
  std::vector<float> hw(wn), hx(xn);
  srand(1);
  for (auto& v : hw) v = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
  for (auto& v : hx) v = rand() / (float)RAND_MAX - 0.5f;
  float *w, *x, *out;
  long long* cyc;
  CK(hipMalloc(&w, wn * 4));
  CK(hipMalloc(&x, xn * 4));
  CK(hipMalloc(&out, on * 4));
  CK(hipMalloc(&cyc, (size_t)nwg * 4 * ntiles * 8));
  CK(hipMemcpy(w, hw.data(), wn * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(x, hx.data(), xn * 4, hipMemcpyHostToDevice));
  printf("register-resident weights probe: %d work-groups x 4 waves, %d tiles each (RW_LDS_BLOCKS=%d)\n", nwg, ntiles, RW_LDS_BLOCKS);
  run<0>("full (refill + epilogue)", w, x, out, cyc, ntiles, nwg);
  run<1>("no refill of the next tile", w, x, out, cyc, ntiles, nwg);
  run<2>("no epilogue", w, x, out, cyc, ntiles, nwg);
  run<3>("MFMAs + LDS operand reads only", w, x, out, cyc, ntiles, nwg);
  return 0;
}
