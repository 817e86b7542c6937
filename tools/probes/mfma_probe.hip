// Calibration probe: achievable v_mfma_f32_16x16x4_f32 rate on this chip for loop shapes like
// tapconv's inner loop (no memory / + ds_read_b128 operand reads), at 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void probe(float* out, int iters, const float* seed) {
  __shared__ __attribute__((aligned(16))) float lds[192 * 36];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, lg = lane >> 4;
  for (int i = tid; i < 192 * 36; i += 256) lds[i] = seed[i % 64] * 1e-3f;
  __syncthreads();
  f32x4 acc[2][4];
  for (int m = 0; m < 2; ++m) for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0, 0, 0, 0};
  f32x4 a[2], b[4];
  for (int m = 0; m < 2; ++m) a[m] = f32x4{seed[lane], seed[lane + 1], 0.5f, 0.25f};
  for (int n = 0; n < 4; ++n) b[n] = f32x4{seed[lane + n], 1.f, 0.5f, 0.25f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      if (MODE == 1) {
#pragma unroll
        for (int m = 0; m < 2; ++m) a[m] = *reinterpret_cast<const f32x4*>(&lds[((64 + m * 16 + li) * 36) + r * 16 + lg * 4]);
#pragma unroll
        for (int n = 0; n < 4; ++n) b[n] = *reinterpret_cast<const f32x4*>(&lds[((n * 16 + li) * 36) + r * 16 + lg * 4]);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n)
            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m][s], b[n][s], acc[m][n], 0, 0, 0);
    }
    if (MODE == 0) { asm volatile("" : "+v"(a[0]), "+v"(b[0])); }
  }
  f32x4 t = {0, 0, 0, 0};
  for (int m = 0; m < 2; ++m) for (int n = 0; n < 4; ++n) t += acc[m][n];
  out[blockIdx.x * 256 + tid] = t[0] + t[1] + t[2] + t[3];
}

template <int MODE>
void run(const char* name, int wgs_per_cu, float* out, const float* seed) {
  const int iters = 400, grid = 256 * wgs_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE><<<grid, 256>>>(out, iters, seed);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int k = 0; k < 5; ++k) probe<MODE><<<grid, 256>>>(out, iters, seed);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 5.0 * grid * 4.0 * iters * 64.0 * 2.0 * 16 * 16 * 4;
  printf("%-28s %d WG/CU (=waves/SIMD): %8.1f TFLOP/s\n", name, wgs_per_cu, flops / (ms * 1e-3) / 1e12);
}

int main() {
  float *out, *seed;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  hipMalloc(&seed, 4096 * sizeof(float));
  hipMemset(seed, 0, 4096 * sizeof(float));
  for (int w = 1; w <= 4; ++w) run<0>("mfma only", w, out, seed);
  for (int w = 1; w <= 4; ++w) run<1>("mfma + 12 ds_read_b128/64", w, out, seed);
  return 0;
}
