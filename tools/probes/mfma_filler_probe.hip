// Probe: what does an instruction placed between two v_mfma_f32_16x16x4_f32 cost on gfx950?  One or two
// waves per SIMD run a loop of 16 MFMAs (two accumulators alternating, like a 32-channel wave tile) with
// a filler pattern; reported: shader cycles per MFMA (s_memtime), median over waves.  The matrix pipe
// takes one such MFMA every 32 cycles.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_filler_probe.hip -o /tmp/mfp && /tmp/mfp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define SB() __builtin_amdgcn_sched_barrier(0)

template <int F, int NACC>
__global__ void probe(float* out, long long* cyc, int iters, const float* seed, const float* gbuf) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += blockDim.x) lds[i] = seed[i % 64] * 1e-3f;
  __syncthreads();
  f32x4 acc[NACC];
  for (int m = 0; m < NACC; ++m) acc[m] = f32x4{0, 0, 0, 0};
  float a = seed[lane], b = seed[lane + 1];
  float x0 = seed[lane + 2], x1 = seed[lane + 3], x2 = seed[lane + 4], x3 = seed[lane + 5];
  float y = seed[lane + 6] * 1e-3f, z = seed[lane + 7] * 1e-3f;
  f32x2 p0 = {x0, x1}, p1 = {x2, x3}, py = {y, z};
  int i0 = lane, i1 = lane + 1;
  f32x4 l0 = {0, 0, 0, 0};
  const f32x4* lp = reinterpret_cast<const f32x4*>(lds) + lane;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gbuf), 0, 64 << 20, 0x00020000);
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 g0 = {0, 0, 0, 0}, g1 = g0, g2 = g0, g3 = g0, g4 = g0, g5 = g0;
  unsigned goff = (blockIdx.x * 4096u + tid * 16u) & ((32u << 20) - 1);
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      acc[k % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k % NACC], 0, 0, 0);
      if (F == 1 || F == 2 || F == 10) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x0) : "v"(y), "v"(z));
      if (F == 2 || F == 10) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x1) : "v"(y), "v"(z));
      if (F == 10) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x2) : "v"(y), "v"(z));
      if (F == 10) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x3) : "v"(y), "v"(z));
      if (F == 3) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p0) : "v"(py));
      if (F == 4) asm volatile("v_add_u32 %0, %1, %0" : "+v"(i0) : "v"(i1));
      if (F == 5) asm volatile("v_mov_b32 %0, %1" : "=v"(i0) : "v"(i1));
      if (F == 6) asm volatile("ds_read_b128 %0, %1" : "=v"(l0) : "v"((unsigned)(lane * 16)));
      if (F == 7) asm volatile("s_nop 0");
      if (F == 14 && (k & 7) == 1) { g0 = __builtin_amdgcn_raw_buffer_load_b128(rs, goff, k * 1024, 0); }      // 1 load per 8 MFMAs
      if (F == 15 && (k & 1) == 1) { g0 = __builtin_amdgcn_raw_buffer_load_b128(rs, goff, k * 1024, 0); }      // 1 load per 2 MFMAs
      if (F == 11) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(i0) : "v"(i1));
      if (F == 12) asm volatile("v_max_f32 %0, %1, %0" : "+v"(x0) : "v"(y));
      if (F == 13) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(i0) : "v"(x0));
      SB();
    }
    if (F == 8) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x0) : "v"(y), "v"(z));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x1) : "v"(y), "v"(z));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x2) : "v"(y), "v"(z));
        asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x3) : "v"(y), "v"(z));
      }
      SB();
    }
    if (F == 9) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p0) : "v"(py));
        asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p1) : "v"(py));
      }
      SB();
    }
    if (F == 16) {      // 6 loads clustered per 16 MFMAs (the conv kernels' refill block)
      g0 = __builtin_amdgcn_raw_buffer_load_b128(rs, goff, 0, 0);
      g1 = __builtin_amdgcn_raw_buffer_load_b128(rs, goff, 4096, 0);
      g2 = __builtin_amdgcn_raw_buffer_load_b128(rs, goff, 8192, 0);
      g3 = __builtin_amdgcn_raw_buffer_load_b128(rs, goff, 12288, 0);
      g4 = __builtin_amdgcn_raw_buffer_load_b128(rs, goff, 16384, 0);
      g5 = __builtin_amdgcn_raw_buffer_load_b128(rs, goff, 20480, 0);
      SB();
    }
    if (F == 14 || F == 15 || F == 16) { goff = (goff + 65536u) & ((32u << 20) - 1); asm volatile("" :: "v"(g0), "v"(g1), "v"(g2), "v"(g3), "v"(g4), "v"(g5)); }
    if (F == 6) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  const long long t1 = clock64();
  f32x4 t = {0, 0, 0, 0};
  for (int m = 0; m < NACC; ++m) t += acc[m];
  out[blockIdx.x * blockDim.x + tid] = t[0] + t[1] + t[2] + t[3] + (float)(g0[0] + g1[1] + g2[2] + g3[3] + g4[0] + g5[1]) + x0 + x1 + x2 + x3 + p0[0] + p0[1] + p1[0] + p1[1] +
                                      (float)i0 + l0[0] + lp[0][0];
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + tid / 64] = t1 - t0;
}

template <int F, int NACC>
void run(const char* name, float* out, long long* cyc, const float* seed, const float* gbuf) {
  const int iters = 2000;
  for (int wps = 1; wps <= 2; ++wps) {
    const int threads = 256 * wps, grid = 256;
    probe<F, NACC><<<grid, threads>>>(out, cyc, iters, seed, gbuf);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    probe<F, NACC><<<grid, threads>>>(out, cyc, iters, seed, gbuf);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid * threads / 64);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double per_wave = (double)h[h.size() / 2] / (iters * 16.0);
    // per SIMD: wps waves share the pipe -> cycles per MFMA the SIMD's pipe sees
    printf("%-44s acc %d  %d wave/SIMD: %6.1f cycles per MFMA per wave = %6.1f per SIMD   (%.0f us, clock %.2f GHz)\n", name, NACC,
           wps, per_wave, per_wave / wps, ms * 1e3, (double)h[h.size() / 2] / (ms * 1e-3) / 1e9);
  }
}

int main(int argc, char** argv) {
  const bool soak = argc > 1;        // "soak [seconds] [filler]": bare / filled MFMA loop repeated for a power / clock trace
  float *out, *seed;
  long long* cyc;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  hipMalloc(&cyc, 256 * 8 * sizeof(long long));
  hipMalloc(&seed, 128 * sizeof(float));
  float hs[128];
  for (int i = 0; i < 128; ++i) hs[i] = (float)rand() / RAND_MAX - 0.5f;
  hipMemcpy(seed, hs, sizeof(hs), hipMemcpyHostToDevice);
  float* gbuf;
  hipMalloc(&gbuf, 64 << 20);
  hipMemset(gbuf, 0, 64 << 20);
  if (soak) {
    const double secs = argc > 2 ? atof(argv[2]) : 20.0;
    const int filler = argc > 3 ? atoi(argv[3]) : 0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    double el = 0;
    long long n = 0;
    while (el < secs * 1e3) {
      for (int k = 0; k < 20; ++k) {
        if (filler == 0) probe<0, 2><<<256, 512>>>(out, cyc, 2000, seed, gbuf);
        else if (filler == 9) probe<9, 2><<<256, 512>>>(out, cyc, 2000, seed, gbuf);
        else probe<1, 2><<<256, 512>>>(out, cyc, 2000, seed, gbuf);
      }
      n += 20;
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      el = ms;
    }
    const double flops = (double)n * 256 * 8 * 2000.0 * 16 * 2.0 * 16 * 16 * 4 * 64 / 64;
    printf("soak filler %d: %lld launches in %.1f s: %.1f TFLOP/s\n", filler, n, el * 1e-3, flops / (el * 1e-3) / 1e12);
    return 0;
  }
  run<0, 2>("bare MFMAs", out, cyc, seed, gbuf);
  run<0, 8>("bare MFMAs", out, cyc, seed, gbuf);
  run<1, 2>("+1 v_fma_f32 per MFMA", out, cyc, seed, gbuf);
  run<1, 8>("+1 v_fma_f32 per MFMA", out, cyc, seed, gbuf);
  run<2, 2>("+2 v_fma_f32 per MFMA", out, cyc, seed, gbuf);
  run<10, 2>("+4 v_fma_f32 per MFMA", out, cyc, seed, gbuf);
  run<3, 2>("+1 v_pk_fma_f32 per MFMA", out, cyc, seed, gbuf);
  run<4, 2>("+1 v_add_u32 per MFMA", out, cyc, seed, gbuf);
  run<11, 2>("+1 v_xor_b32 per MFMA", out, cyc, seed, gbuf);
  run<5, 2>("+1 v_mov_b32 per MFMA", out, cyc, seed, gbuf);
  run<12, 2>("+1 v_max_f32 per MFMA", out, cyc, seed, gbuf);
  run<13, 2>("+1 v_cvt_f16_f32 per MFMA", out, cyc, seed, gbuf);
  run<6, 2>("+1 ds_read_b128 per MFMA", out, cyc, seed, gbuf);
  run<7, 2>("+1 s_nop per MFMA", out, cyc, seed, gbuf);
  run<14, 2>("+1 buffer_load_dwordx4 per 8 MFMAs", out, cyc, seed, gbuf);
  run<15, 2>("+1 buffer_load_dwordx4 per 2 MFMAs", out, cyc, seed, gbuf);
  run<16, 2>("6 buffer_load_dwordx4 clustered per 16 MFMAs", out, cyc, seed, gbuf);
  run<8, 2>("16 v_fma_f32 clustered per 16 MFMAs", out, cyc, seed, gbuf);
  run<9, 2>("8 v_pk_fma_f32 clustered per 16 MFMAs", out, cyc, seed, gbuf);
  return 0;
}
