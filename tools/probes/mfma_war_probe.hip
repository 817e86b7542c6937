// Probe: is v_mfma_f32_32x32x16_bf16 safe against later writes to its A/B source registers when
// two waves share a SIMD?  Every wave of a 512-thread group computes the same 64x32xK product via
// 3-way bf16 splits done in registers right before use (register reuse right after the MFMAs);
// all waves must produce bit-identical results.  Variants: PAD s_nops after each MFMA group.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void split8(const float* x, u32x4& hi, u32x4& mid, u32x4& lo) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    unsigned a0 = __float_as_uint(x[2 * p]), a1 = __float_as_uint(x[2 * p + 1]);
    hi[p] = __builtin_amdgcn_perm(a1, a0, 0x07060302u);
    float r0 = x[2 * p] - __uint_as_float(a0 & 0xffff0000u);
    float r1 = x[2 * p + 1] - __uint_as_float(a1 & 0xffff0000u);
    unsigned b0 = __float_as_uint(r0), b1 = __float_as_uint(r1);
    mid[p] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
    float s0 = r0 - __uint_as_float(b0 & 0xffff0000u);
    float s1 = r1 - __uint_as_float(b1 & 0xffff0000u);
    lo[p] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
  }
}
__device__ __forceinline__ f32x16 mm(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// MODE 0: one accumulator per row block, 9 chained MFMAs; MODE 1: two accumulators (lo / rest)
template <int MODE, int PAD>
__global__ __launch_bounds__(512) void gemm_split(const float* A, const float* B, float* C, int K) {
  const int l = threadIdx.x & 63, r = l & 31, kh = l >> 5;
  const int gw = blockIdx.x * 8 + (threadIdx.x >> 6);
  f32x16 acc[2] = {}, acc2[2] = {};
  for (int k0 = 0; k0 < K; k0 += 16) {
    u32x4 bh, bm, bl;
    float xb[8];
    for (int j = 0; j < 8; ++j) xb[j] = B[(size_t)r * K + k0 + 8 * kh + j];
    split8(xb, bh, bm, bl);
    for (int t = 0; t < 2; ++t) {
      u32x4 ah, am, al;
      float xa[8];
      for (int j = 0; j < 8; ++j) xa[j] = A[(size_t)(32 * t + r) * K + k0 + 8 * kh + j];
      split8(xa, ah, am, al);
      if (MODE == 0) {
        acc[t] = mm(al, bl, acc[t]); acc[t] = mm(am, bl, acc[t]); acc[t] = mm(al, bm, acc[t]);
        acc[t] = mm(ah, bl, acc[t]); acc[t] = mm(al, bh, acc[t]); acc[t] = mm(am, bm, acc[t]);
        acc[t] = mm(ah, bm, acc[t]); acc[t] = mm(am, bh, acc[t]); acc[t] = mm(ah, bh, acc[t]);
      } else {
        acc2[t] = mm(al, bl, acc2[t]); acc[t] = mm(am, bm, acc[t]); acc2[t] = mm(am, bl, acc2[t]);
        acc[t] = mm(ah, bm, acc[t]);   acc2[t] = mm(al, bm, acc2[t]); acc[t] = mm(am, bh, acc[t]);
        acc2[t] = mm(ah, bl, acc2[t]); acc[t] = mm(ah, bh, acc[t]);   acc2[t] = mm(al, bh, acc2[t]);
      }
      if (PAD) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15");
    }
  }
  for (int t = 0; t < 2; ++t)
    for (int g = 0; g < 16; ++g) {
      int row = (g & 3) + 8 * (g >> 2) + 4 * kh;
      C[(size_t)gw * 2048 + (size_t)(32 * t + row) * 32 + r] = acc[t][g] + (MODE ? acc2[t][g] : 0.f);
    }
}

static double nrm() { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return sqrt(-2 * log(u)) * cos(6.283185307179586 * v); }

template <int MODE, int PAD>
void run(const float* dA, const float* dB, float* dC, int K, int nwg, int nthr, const char* name) {
  std::vector<float> C((size_t)nwg * 8 * 2048), ref(2048);
  hipLaunchKernelGGL((gemm_split<MODE, PAD>), dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(ref.data(), dC, 2048 * 4, hipMemcpyDeviceToHost));
  long bad_w = 0, bad_e = 0; double mx = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((gemm_split<MODE, PAD>), dim3(nwg), dim3(nthr), 0, 0, dA, dB, dC, K);
    CK(hipDeviceSynchronize());
    int nw = nwg * nthr / 64;
    CK(hipMemcpy(C.data(), dC, (size_t)nw * 2048 * 4, hipMemcpyDeviceToHost));
    for (int w = 0; w < nw; ++w) {
      // waves of a group are at slots blockIdx*8 + wave; with nthr < 512 only the first slots are written
      int slot = (w / (nthr / 64)) * 8 + w % (nthr / 64);
      long be = 0;
      for (int i = 0; i < 2048; ++i) if (C[(size_t)slot * 2048 + i] != ref[i]) { ++be; mx = fmax(mx, fabs(C[(size_t)slot * 2048 + i] - ref[i])); }
      bad_e += be; bad_w += be != 0;
    }
  }
  printf("%-44s %d WGs x %d threads: waves differing from the single-wave result %ld, elements %ld, max diff %.3e\n", name, nwg, nthr, bad_w, bad_e, mx);
}

int main() {
  const int K = 1536;
  std::vector<float> A(64 * K), B(32 * K);
  srand(5);
  for (auto& v : A) v = (float)(nrm() * 0.05);
  for (auto& v : B) v = (float)nrm();
  float *dA, *dB, *dC;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, (size_t)256 * 8 * 2048 * 4));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  run<0, 0>(dA, dB, dC, K, 256, 256, "chained acc, 1 wave/SIMD");
  run<0, 0>(dA, dB, dC, K, 256, 512, "chained acc, 2 waves/SIMD");
  run<1, 0>(dA, dB, dC, K, 256, 256, "two accs, 1 wave/SIMD");
  run<1, 0>(dA, dB, dC, K, 256, 512, "two accs, 2 waves/SIMD");
  run<0, 1>(dA, dB, dC, K, 256, 512, "chained acc, 2 waves/SIMD, pad after group");
  run<1, 1>(dA, dB, dC, K, 256, 512, "two accs, 2 waves/SIMD, pad after group");
  return 0;
}
