// Calibration probe: v_mfma_f32_16x16x4_f32 rate on REAL (random, non-zero) operands vs zeros, at
// 1..3 waves per SIMD, for the accumulator shapes of tapconv (2x4) and sconv (4x2): the chip
// clocks to its power budget, so the zero-data rate is not the practical ceiling.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int TM, int TN>
__global__ __launch_bounds__(256) void probe(float* out, int iters, const float* seed, unsigned long long* cyc) {
  const int tid = threadIdx.x, lane = tid & 63;
  f32x4 acc[TM][TN];
  for (int m = 0; m < TM; ++m) for (int n = 0; n < TN; ++n) acc[m][n] = f32x4{0, 0, 0, 0};
  f32x4 a[TM], b[TN];
  for (int m = 0; m < TM; ++m) a[m] = *reinterpret_cast<const f32x4*>(seed + 4 * ((lane + 7 * m) & 255));
  for (int n = 0; n < TN; ++n) b[n] = *reinterpret_cast<const f32x4*>(seed + 1024 + 4 * ((lane + 5 * n) & 255));
  const unsigned long long c0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
          for (int n = 0; n < TN; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][s], b[n][s], acc[m][n], 0, 0, 0);
    }
    asm volatile("" : "+v"(a[0]), "+v"(b[0]));
  }
  const unsigned long long c1 = clock64();
  f32x4 t = {0, 0, 0, 0};
  for (int m = 0; m < TM; ++m) for (int n = 0; n < TN; ++n) t += acc[m][n];
  out[blockIdx.x * 256 + tid] = t[0] + t[1] + t[2] + t[3];
  if (tid == 0 && blockIdx.x == 0) *cyc = c1 - c0;
}

template <int TM, int TN>
void run(const char* name, int wgs_per_cu, float* out, const float* seed, unsigned long long* cyc) {
  const int iters = 600, grid = 256 * wgs_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<TM, TN><<<grid, 256>>>(out, iters, seed, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int k = 0; k < 5; ++k) probe<TM, TN><<<grid, 256>>>(out, iters, seed, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double nm = (double)iters * 2 * 4 * TM * TN;
  const double flops = 5.0 * grid * 4.0 * nm * 2.0 * 16 * 16 * 4;
  printf("%-22s acc %dx%d  %d waves/SIMD: %7.1f TFLOP/s  %5.1f shader cycles per MFMA per wave (x%d waves)  clock %.2f GHz\n", name, TM, TN,
         wgs_per_cu, flops / (ms * 1e-3) / 1e12, c / nm, wgs_per_cu, c / (ms / 5 * 1e-3) / 1e9);
}

int main() {
  float *out, *seed; unsigned long long* cyc;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  hipMalloc(&seed, 4096 * sizeof(float));
  hipMalloc(&cyc, 8);
  float h[4096];
  for (int z = 0; z < 2; ++z) {
    srand(1);
    for (int i = 0; i < 4096; ++i) h[i] = z ? 0.f : (rand() / (float)RAND_MAX - 0.5f) * 1e-2f;
    hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
    const char* nm = z ? "zeros" : "random";
    for (int w = 1; w <= 3; ++w) run<2, 4>(nm, w, out, seed, cyc);
    for (int w = 1; w <= 3; ++w) run<4, 2>(nm, w, out, seed, cyc);
    for (int w = 1; w <= 2; ++w) run<4, 4>(nm, w, out, seed, cyc);
  }
  return 0;
}
