// BatchNorm "finalize" steps -- merge the <= 256 per-work-group partial rows a producer kernel has
// written into per-channel coefficients -- as DEVICE functions, so that the producer's LAST-ARRIVING
// work-group runs them itself instead of a separate one-wave-per-channel launch between every conv
// and the apply pass that waits for it (round 3: 156 such launches per step-2 iteration, 4.7 us each
// alone, 10-12 us beside the convs of the other streams).  The stand-alone finalize kernels of bn.hip
// call the same functions with the same thread layout: fused and unfused results are bit-identical.
//
// Cross-work-group hand-off (MI355X_MICROARCH.md, "Workgroup dispatch ... inter-workgroup
// visibility"; per-XCD L2s are not coherent, a CU's L1 is never refreshed):
//   producer (every work-group): partial rows with agent-scope relaxed atomic stores (global_store
//     sc1: write-through, the line does not stay in the XCD's L2) -> s_waitcnt vmcnt(0) in the
//     storing lanes -> work-group barrier -> ONE lane: agent-scope relaxed fetch_add on the ticket;
//   consumer (the work-group that drew the last ticket): reads the rows with agent-scope relaxed
//     atomic loads (global_load sc1: not served from its L1) -- the "sc1 stores AND sc1 loads" form,
//     no fences, no L2 write-back of the conv's freshly stored output behind the hand-off.
// The ticket is a caller-owned device word, zero before the first launch; the last arriver puts it
// back to zero, so launches on one stream can share it.
//
// Fixed merge order (deterministic, independent of which work-group arrives last): channel c is
// handled by J = (C == 128 ? 4 : 8) slices; slice j merges rows j, j + J, j + 2J, ... in that
// order; the J slices are merged 0, 1, ..., J-1.  Needs blockDim.x >= J * C threads (512 for every
// user) and 3 * J * C floats (forward) / 2 * J * C doubles (backward) of LDS scratch.
#pragma once
#include "common.h"

// The hand-off below is written against gfx942 / gfx950 code generation (sc1 write-through stores and
// sc1 loads for agent-scope relaxed atomics, one s_waitcnt vmcnt(0) between the row stores and the
// ticket): under the HIP memory model alone it would need a release on the ticket and an acquire in the
// last arriver.  Refuse any other target rather than run it there; the bit-identity tests against the
// stand-alone finalize (tests/test_bn_finalize_gpu.py, MDIL_NO_BNFIN) guard the assumption on this one.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "bnfin.h: the cross-work-group hand-off is validated for gfx950 (MI355X) only"
#endif

#ifndef BNFIN_ACQREL
#define BNFIN_ACQREL 0     // 1: release / acquire on the ticket instead of the sc1-stores + sc1-loads form (A/B switch)
#endif

struct BnFinFwd {          // train-mode statistics -> coefficients (+ running statistics)
  unsigned* ticket;        // nullptr: no fused finalize
  const float* gamma;
  const float* beta;
  float* running_mean;     // may be null
  float* running_var;
  long long* nbt;          // may be null
  float eps, momentum;
  float* save_mean;        // [C] each
  float* save_invstd;
  float* scale;
  float* shift;
};

struct BnFinBwd {          // BN-backward reductions -> dgamma / dbeta (+=) and the apply coefficients
  unsigned* ticket;        // nullptr: no fused finalize
  const float* gamma;
  const float* save_invstd;
  float* dgamma;           // may be null
  float* dbeta;
  int accumulate;
  float n;                 // pixels per channel
  float* coef;             // [3][C]: gamma * invstd, sum(g) / n, sum(g * xhat) / n
};

__device__ __forceinline__ int bnfin_slices(int C) { return C == 128 ? 4 : 8; }

__device__ __forceinline__ float bnfin_ld(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void bnfin_st(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Chan / Welford merge with v_rcp_f32 for the weight (1 ulp on a weight in [0, 1])
__device__ __forceinline__ void bnfin_merge(float& n, float& mean, float& m2, float nb, float meanb,
                                            float m2b) {
  const bool has = nb > 0.f;          // padded rows carry a clamped row's values with count 0
  const float nn = n + nb;
  const float f = has ? nb * __builtin_amdgcn_rcpf(nn) : 0.f;
  const float d = meanb - mean;
  mean = mean + d * f;
  m2 = m2 + (has ? m2b : 0.f) + d * d * n * f;
  n = nn;
}

// Called by ALL threads of a work-group after its partial-row stores (bnfin_st) are issued.
// -> true in every thread of the work-group that arrived last (total = work-groups of the launch).
__device__ __forceinline__ bool bnfin_arrive(unsigned* ticket, unsigned total, int* lds_flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this lane's row stores have left the CU
  __syncthreads();
  if (threadIdx.x == 0) {
#if BNFIN_ACQREL
    // HIP-memory-model form: RELEASE on the arrival (buffer_wbl2 sc1 + s_waitcnt: the XCD L2's
    // dirty lines are written back first -- including whatever of the conv's freshly stored output tile is
    // still there), ACQUIRE in the last arriver (buffer_inv sc1).  Measured against the relaxed form:
    // profiles/r06_experiments.txt #B1.
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = t == total - 1;
    if (last) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#else
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = t == total - 1;
    if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    *lds_flag = last ? 1 : 0;
  }
  __syncthreads();
  return *lds_flag != 0;
}

// partial: [nblk][2][C] (mean, M2), pcount: [nblk].  scratch: 3 * J * C floats of LDS.
__device__ __forceinline__ void bnfin_forward(const BnFinFwd& f, const float* partial,
                                              const float* pcount, int nblk, int C, float* scratch) {
  const int J = bnfin_slices(C);
  const int t = threadIdx.x, c = t % C, j = t / C;
  if (j < J) {
    float n = 0.f, mean = 0.f, m2 = 0.f;
    constexpr int U = 16;              // rows in flight per thread
    for (int b0 = j; b0 < nblk; b0 += U * J) {
      float pn[U], pm[U], pq[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int b = b0 + u * J;
        const int bc = b < nblk ? b : nblk - 1;      // clamped, unconditional loads; select afterwards
        const float vn = bnfin_ld(pcount + bc);
        pm[u] = bnfin_ld(partial + ((long long)bc * 2 + 0) * C + c);
        pq[u] = bnfin_ld(partial + ((long long)bc * 2 + 1) * C + c);
        pn[u] = b < nblk ? vn : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) bnfin_merge(n, mean, m2, pn[u], pm[u], pq[u]);
    }
    scratch[(j * 3 + 0) * C + c] = n;
    scratch[(j * 3 + 1) * C + c] = mean;
    scratch[(j * 3 + 2) * C + c] = m2;
  }
  __syncthreads();
  if (t < C) {
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int s = 0; s < J; ++s)
      bnfin_merge(n, mean, m2, scratch[(s * 3 + 0) * C + t], scratch[(s * 3 + 1) * C + t],
                  scratch[(s * 3 + 2) * C + t]);
    const float var = m2 / n;
    const float invstd = 1.0f / sqrtf(var + f.eps);
    f.save_mean[t] = mean;
    f.save_invstd[t] = invstd;
    const float sc = f.gamma[t] * invstd;
    f.scale[t] = sc;
    f.shift[t] = f.beta[t] - mean * sc;
    if (f.running_mean) {
      const float unbiased = n > 1.f ? m2 / (n - 1.f) : var;
      f.running_mean[t] = (1.f - f.momentum) * f.running_mean[t] + f.momentum * mean;
      f.running_var[t] = (1.f - f.momentum) * f.running_var[t] + f.momentum * unbiased;
    }
    if (t == 0 && f.nbt) *f.nbt += 1;
  }
}

// partial: [nblk][2][C] (sum g, sum g * xhat).  scratch: 2 * J * C doubles of LDS.
__device__ __forceinline__ void bnfin_backward(const BnFinBwd& f, const float* partial, int nblk, int C,
                                               double* scratch) {
  const int J = bnfin_slices(C);
  const int t = threadIdx.x, c = t % C, j = t / C;
  if (j < J) {
    double a = 0.0, b = 0.0;
    constexpr int U = 16;
    for (int b0 = j; b0 < nblk; b0 += U * J) {
      float pa[U], pb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int blk = b0 + u * J;
        const int bc = blk < nblk ? blk : nblk - 1;
        const float va = bnfin_ld(partial + ((long long)bc * 2 + 0) * C + c);
        const float vb = bnfin_ld(partial + ((long long)bc * 2 + 1) * C + c);
        pa[u] = blk < nblk ? va : 0.f;
        pb[u] = blk < nblk ? vb : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        a += (double)pa[u];
        b += (double)pb[u];
      }
    }
    scratch[(j * 2 + 0) * C + c] = a;
    scratch[(j * 2 + 1) * C + c] = b;
  }
  __syncthreads();
  if (t < C) {
    double a = 0.0, b = 0.0;
    for (int s = 0; s < J; ++s) {
      a += scratch[(s * 2 + 0) * C + t];
      b += scratch[(s * 2 + 1) * C + t];
    }
    float db0 = 0.f, dg0 = 0.f;
    if (f.accumulate) {
      if (f.dbeta) db0 = f.dbeta[t];
      if (f.dgamma) dg0 = f.dgamma[t];
    }
    if (f.dbeta) f.dbeta[t] = db0 + (float)a;
    if (f.dgamma) f.dgamma[t] = dg0 + (float)b;
    f.coef[0 * C + t] = f.gamma[t] * f.save_invstd[t];
    f.coef[1 * C + t] = (float)(a / (double)f.n);
    f.coef[2 * C + t] = (float)(b / (double)f.n);
  }
}
