// EXPERIMENT RECORD (round 6, NOT part of libmdil_hip.so; profiles/r06_experiments.txt #R1): the C = 128 3-tap
// F(4,3) conv with the weights resident in REGISTERS.  This is the kernel block as it was built and measured inside
// mdil_ss_amd/csrc/w4conv.hip (it uses that file's W4Cfg / w4_load / w4_store / PA.. / GN.. definitions and was
// dispatched from launch_w4conv<128, false> for MODE 0): correct on every epilogue (tests/test_r4conv_gpu.py of
// the same commit: <= 1 ulp from w4conv, 1e-4 of scale from F.conv2d), 10 % SLOWER per tile than w4conv.
//   launch us at N = 6 / 12:  3x1 bias+relu 31.6 / 53.5 (w4conv 26.0 / 45.7),  dgrad 1x3 + gate 36.7 / 62.1 (27.7 / 50.4)
//   step: 264.7 img/s against 271.8 (same box, three A/B pairs)
// Why: one 512-register wave per SIMD has nobody to cover its epilogue (output transform, stores), its share of
// the next tile's operand staging or a late load -- 40 cycles per MFMA all-in, where w4conv's two waves per SIMD
// reach 36 although each of them loads and transforms four times as much (tools/probes/regweights_probe.hip has
// the ablation: 34.1 cycles per MFMA for MFMAs + LDS operand reads, +1.6 epilogue, +2.9 .. 4.5 staging).
// Kept for the record of HOW (AccVGPR-pinned operands through hand-written MFMA statements, the hazards hipcc
// does not pad around them, the register budget); not compiled by the Makefile.
// ================================================================================================
// r4conv: the C = 128 3-tap forms with the WEIGHTS RESIDENT IN REGISTERS (round 6).
//
// The register file of a CU (4 x 128 KB) is the only on-CU storage that holds the whole transformed weight
// set of a C = 128 conv (6 images x 128 x 128 x 4 B = 393 KB; the LDS holds 160 KB, which is why the kernel
// above runs 32 output channels per work-group and every wave loads and transforms its own pixel operands:
// 6 buffer loads + 24 v_pk_fma_f32 per 48 MFMAs = 8 of its main loop's 40 cycles per MFMA, and a fill of
// 3.2 us per launch before the first MFMA).  Here:
//   * work-group = 4 waves, one per SIMD, up to 512 registers each; wave w owns output channels
//     [32w, 32w + 32) of EVERY pixel tile of the work-group and keeps their A fragments for the six Winograd
//     positions: channel blocks 0-4 in AccVGPRs (240; hand-written MFMA statements take them as "a" operands:
//     left to itself hipcc copies each one back with a v_accvgpr_read per use), blocks 5-6 in VGPRs (96),
//     block 7 in LDS (48 KB for the four slices);
//   * a tile's (16 quads) pixel operands are loaded and transformed ONCE per work-group -- wave w does
//     channel blocks 2w and 2w + 1: 12 buffer loads + 48 v_pk_fma_f32 + 12 ds_write_b128 per tile instead of
//     48 + 192 per wave tile -- into a double-buffered LDS image [2][8 blocks][6 positions][64 lanes] x 16 B
//     (96 KB) that all four waves read back in fragment order (lane-private 16-byte slots: conflict free);
//   * ONE work-group barrier per tile; each pixel tile is read from memory once (above: by four work-groups).
// Accumulation order per accumulator (channel block, k-step) and every transform expression equal w4conv's:
// results are bit-identical to it (tests/test_hip_parity.py::test_r4conv_is_bit_identical_to_w4conv).
// Covers MODE 0 (bias / folded BN / ReLU / residual / gates); statistics / reduction / tail forms and the
// adapter forms (a seventh image: no room) stay on the kernel above.  MDIL_NO_R4CONV=1: A/B switch.
constexpr int R4_WAVES = 4, R4_THREADS = 256;
constexpr int R4_NB = 8;          // 16-channel blocks of C = 128
constexpr int R4_NAG = 5;         // blocks whose A fragments sit in AccVGPRs
constexpr int R4_NLD = 1;         // ... in LDS (the last ones)
constexpr int R4_NVG = R4_NB - R4_NAG - R4_NLD;
constexpr int R4_XPOS = 2;        // ... and the first R4_XPOS positions of block R4_NAG (the 16 AccVGPRs five blocks leave)
// where the A fragments of (block rr, position p) live: 0 AccVGPR, 1 VGPR, 2 LDS
constexpr int r4_home(int rr, int p) {
  return (rr < R4_NAG || (rr == R4_NAG && p < R4_XPOS)) ? 0 : rr < R4_NAG + R4_NVG ? 1 : 2;
}

// D = A B + D on v_mfma_f32_16x16x4_f32, A operand in an AccVGPR ("a") or a VGPR ("v"); the *0 forms start an
// accumulator (C = 0 literal: no zero-fill moves, no VALU-write -> MFMA-read hazard in front of a hand-written
// statement).  hipcc pads nothing around inline asm: see the s_nop before the epilogue.
#define R4_MFMA_A0(acc, a, b) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(acc) : "a"(a), "v"(b))
#define R4_MFMA_A(acc, a, b) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b))
#define R4_MFMA_V(acc, a, b) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
// (tools/kernel_resources.py --r4 checks the built kernel: no v_accvgpr_read, no VALU write of an A / B operand
// between its first and last MFMA)

template <bool EOPS>
__global__ __launch_bounds__(R4_THREADS) void r4conv_kernel(const wconv_args a) {
  constexpr int C = 128, COW = 32, TM = 2, TN = W4_TN;
  __shared__ __attribute__((aligned(16))) f32x4 Vs[2][R4_NB][6][64];          // transformed pixel operands
  __shared__ __attribute__((aligned(16))) f32x4 As[R4_NLD][R4_WAVES][6][TM][64];   // A fragments of the last block(s)
  __shared__ __attribute__((aligned(16))) float Ep[2 * C];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int H = a.H, W = a.W;
  const int npix = a.N * H * W;
  const int nquads = npix >> 2;
  const int ntiles = (nquads + W4_QUADS - 1) / W4_QUADS;
  const int delta = a.delta;

  const int in_bytes = npix * C * 4;
  const __amdgpu_buffer_rsrc_t rs0 =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in0), 0, in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.in1 ? a.in1 : a.in0), 0, in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs3 = a.src3 ? rs1 : rs0;
  constexpr unsigned OOB = 0x80000000u;

  // quad -> pixels: as in w4conv_kernel (power-of-two fast path, else integer divisions)
  const int L = a.axis ? W : H;
  const int S = a.axis ? delta : delta * W;
  const int nb = L / (4 * delta);
  const int sb = S * (C * 4);
  const bool p2 = a.sh_delta >= 0;
  auto setup = [&](int tile, unsigned (&vb)[3], int& P0, bool& ok) {
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
      } else {
        const int w = pc & (W - 1), t1 = pc >> a.sh_W;
        const int q = t1 & (delta - 1), t2 = t1 >> a.sh_delta;
        const int hb = t2 & (nb - 1);
        const int img = t2 >> a.sh_nb;
        x0 = 4 * delta * hb + q;
        P0 = (img * H + x0) * W + w;
      }
    } else if (a.axis) {
      const int q = pc % delta, t1 = pc / delta;
      const int wb = t1 % nb, row = t1 / nb;
      x0 = 4 * delta * wb + q;
      P0 = row * W + x0;
    } else {
      const int w = pc % W, t1 = pc / W;
      const int q = t1 % delta, t2 = t1 / delta;
      const int hb = t2 % nb;
      const int img = t2 / nb;
      x0 = 4 * delta * hb + q;
      P0 = (img * H + x0) * W + w;
    }
    const unsigned base = (unsigned)P0 * (unsigned)(C * 4) + (unsigned)lg * 16u;
    vb[0] = (ok && x0 - delta >= 0) ? base - (unsigned)sb : OOB;
    vb[1] = ok ? base : OOB;
    vb[2] = (ok && x0 + 4 * delta < L) ? base + 4u * (unsigned)sb : OOB;
  };
  // raw operands d0..d5 of channel block rr of a tile (lane = quad li, channels 16 rr + 4 lg .. + 3)
  auto load_raw = [&](f32x4 (&d)[6], int rr, const unsigned (&vb)[3]) __attribute__((always_inline)) {
    d[0] = w4_load(rs3, vb[0] + rr * 64, 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) d[1 + k] = w4_load(rs3, vb[1] + rr * 64, k * sb);
    d[5] = w4_load(rs3, vb[2] + rr * 64, 0);
  };
  float kNB2 = -PB2, kNA2 = -PA2, kA2B2 = PA2B2, kNSUM = -PSUM, kPA = PA, kNA = -PA, kPB = PB, kNB = -PB;
  asm volatile("" : "+s"(kNB2), "+s"(kNA2), "+s"(kA2B2), "+s"(kNSUM), "+s"(kPA), "+s"(kNA), "+s"(kPB), "+s"(kNB));
  // t = B^T d (the expressions of w4conv_kernel) -> this block's six B operands, into the LDS image
  auto transform_store = [&](const f32x4 (&d)[6], int buf, int rr) __attribute__((always_inline)) {
    const f32x4 e1 = d[2] * kNB2 + d[4], p1 = d[1] * kNB2 + d[3];
    const f32x4 e2 = d[2] * kNA2 + d[4], p2_ = d[1] * kNA2 + d[3];
    Vs[buf][rr][0][lane] = d[0] * kA2B2 + (d[2] * kNSUM + d[4]);
    Vs[buf][rr][1][lane] = p1 * kPA + e1;
    Vs[buf][rr][2][lane] = p1 * kNA + e1;
    Vs[buf][rr][3][lane] = p2_ * kPB + e2;
    Vs[buf][rr][4][lane] = p2_ * kNB + e2;
    Vs[buf][rr][5][lane] = d[1] * kA2B2 + (d[3] * kNSUM + d[5]);
  };

  // ---- first tile's operands are requested before anything else
  int tile = blockIdx.x;
  const int step = gridDim.x;
  unsigned vbA[3], vbB[3];
  int P0A = 0, P0B = 0;
  bool okA = false, okB = false;
  setup(tile, vbA, P0A, okA);
  f32x4 raw[6];
  load_raw(raw, 2 * wave, vbA);

  // ---- weights -> Winograd domain -> registers / LDS, in MFMA fragment order.  Lane (li, lg) of fragment
  // (block rr, channel tile m) holds U_pos[co = 32 wave + 16 m + li][ci = 16 rr + 4 lg .. + 3]: three 16-byte
  // loads of the raw taps, the transform expressions of w4conv_kernel's fill, one register per k-step.
  // Two channel blocks (12 loads) per round: more in flight would push the VGPR-resident fragments into
  // AccVGPR spill slots, and a v_accvgpr_read in front of a hand-written MFMA is an unpadded hazard.
  float Aa[R4_NAG + 1][6][TM][4];      // "a": AccVGPRs (r4_home == 0)
  f32x4 Av[R4_NVG][6][TM];             // VGPRs (r4_home == 1)
  auto wload = [&](int rr, int m, int t) __attribute__((always_inline)) {
    const long long row = (long long)(wave * COW + m * 16 + li) * C + rr * 16 + lg * 4;
    return *reinterpret_cast<const f32x4*>(a.wpk + (long long)a.tap[t] * C * C + row);
  };
  auto wtransform = [&](const f32x4 g0, const f32x4 g1, const f32x4 g2, f32x4 (&U)[6]) __attribute__((always_inline)) {
    const f32x4 ea = g0 + g2 * PA2, oa = g1 * PA;
    const f32x4 eb = g0 + g2 * PB2, ob = g1 * PB;
    U[0] = g0 * (float)GN0;
    U[1] = (ea + oa) * (float)GN1;
    U[2] = (ea - oa) * (float)GN1;
    U[3] = (eb + ob) * (float)GN3;
    U[4] = (eb - ob) * (float)GN3;
    U[5] = g2;
  };
  constexpr int RB = 2;                 // channel blocks per round
#pragma unroll
  for (int q = 0; q < R4_NB / RB; ++q) {
    f32x4 g[RB][TM][3];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int t = 0; t < 3; ++t) g[r][m][t] = wload(q * RB + r, m, t);
    __builtin_amdgcn_sched_barrier(0);      // the round's 12 loads are in flight before the first use
    if (q == 1) {
      // the first tile's operands have arrived meanwhile: block 2w into LDS image 0, block 2w + 1 requested
      transform_store(raw, 0, 2 * wave);
      load_raw(raw, 2 * wave + 1, vbA);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (q == 3) {
      transform_store(raw, 0, 2 * wave + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int rr = q * RB + r;
#pragma unroll
      for (int m = 0; m < TM; ++m) {
        f32x4 U[6];
        wtransform(g[r][m][0], g[r][m][1], g[r][m][2], U);
#pragma unroll
        for (int p = 0; p < 6; ++p) {
          if (r4_home(rr, p) == 0) {
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_)
              asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(Aa[rr <= R4_NAG ? rr : 0][p][m][s_]) : "v"(U[p][s_]));
          } else if (r4_home(rr, p) == 1) {
            Av[(rr >= R4_NAG && rr < R4_NAG + R4_NVG) ? rr - R4_NAG : 0][p][m] = U[p];
          } else {
            As[rr >= R4_NAG + R4_NVG ? rr - R4_NAG - R4_NVG : 0][wave][p][m][lane] = U[p];
          }
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int i = tid; i < C; i += R4_THREADS) {
    float sc = 1.f, bi = a.e.bias ? a.e.bias[i] : 0.f;
    if (a.e.bias2) bi += a.e.bias2[i];
    if (a.e.scale) {
      sc = a.e.scale[i];
      bi = bi * sc + a.e.shift[i];
    }
    Ep[i] = sc;
    Ep[C + i] = bi;
  }
  __syncthreads();

  int buf = 0;
  while (tile < ntiles) {
    const int ntile = tile + step;
    setup(ntile, vbB, P0B, okB);
    const bool more = ntile < ntiles;            // uniform
    if (more) load_raw(raw, 2 * wave, vbB);      // next tile, this wave's first block
    f32x4 acc[6][TM];
    f32x4 v[6];
#pragma unroll
    for (int p = 0; p < 6; ++p) v[p] = Vs[buf][0][p][lane];
    f32x4 al[TM];                                // A fragments of the LDS-resident block (rolling, like v)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rr = 0; rr < R4_NB; ++rr) {
      if (rr == R4_NAG + R4_NVG) {
#pragma unroll
        for (int m = 0; m < TM; ++m) al[m] = As[0][wave][0][m][lane];
      }
#pragma unroll
      for (int p = 0; p < 6; ++p) {
        const f32x4 b = v[p];
        if (r4_home(rr, p) == 0) {
#pragma unroll
          for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
            for (int m = 0; m < TM; ++m) {
              if (rr == 0 && s_ == 0) {
                R4_MFMA_A0(acc[p][m], Aa[rr <= R4_NAG ? rr : 0][p][m][s_], b[s_]);
              } else {
                R4_MFMA_A(acc[p][m], Aa[rr <= R4_NAG ? rr : 0][p][m][s_], b[s_]);
              }
            }
        } else if (r4_home(rr, p) == 1) {
#pragma unroll
          for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
            for (int m = 0; m < TM; ++m)
              R4_MFMA_V(acc[p][m], Av[(rr >= R4_NAG && rr < R4_NAG + R4_NVG) ? rr - R4_NAG : 0][p][m][s_], b[s_]);
        } else {
          f32x4 cur[TM];
#pragma unroll
          for (int m = 0; m < TM; ++m) cur[m] = al[m];
#pragma unroll
          for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
            for (int m = 0; m < TM; ++m) R4_MFMA_V(acc[p][m], cur[m][s_], b[s_]);
          // the next position's fragments go into the registers just consumed
          if (p + 1 < 6) {
#pragma unroll
            for (int m = 0; m < TM; ++m) al[m] = As[rr - R4_NAG - R4_NVG][wave][p + 1][m][lane];
          }
        }
        // this position's operand of the NEXT block goes into the registers just consumed
        if (rr + 1 < R4_NB) v[p] = Vs[buf][rr + 1][p][lane];
        __builtin_amdgcn_sched_barrier(0);
      }
      // this wave's share of the next tile's operands (blocks 2w, 2w + 1), each in ONE VALU block behind an
      // MFMA block, three channel blocks (~5,000 cycles) behind its loads
      if (rr == 3 && more) {
        transform_store(raw, buf ^ 1, 2 * wave);
        load_raw(raw, 2 * wave + 1, vbB);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (rr == 6 && more) {
        transform_store(raw, buf ^ 1, 2 * wave + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // hand-written MFMAs: their results are read by compiler code below (8-pass XDL: >= 11 wait states)
    asm volatile("s_nop 15\n\ts_nop 1" ::: "memory");

    // ---- output transform y = A^T m (w4conv_kernel's expressions)
    f32x4 ay[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m) {
      const f32x4 s12 = acc[1][m] + acc[2][m], d12 = acc[1][m] - acc[2][m];
      const f32x4 s34 = acc[3][m] + acc[4][m], d34 = acc[3][m] - acc[4][m];
      ay[m][0] = (acc[0][m] + s12) + s34;
      ay[m][1] = d12 * PA + d34 * PB;
      ay[m][2] = s12 * PA2 + s34 * PB2;
      ay[m][3] = (d12 * PA3 + d34 * PB3) + acc[5][m];
      // pin the transform HERE (opaque values): left alone hipcc sinks it below the epilogue's operand loads, the
      // 48 accumulator registers stay live beside the operand tiles, and VGPR-resident A fragments get spilled
#pragma unroll
      for (int n = 0; n < TN; ++n) asm volatile("" : "+v"(ay[m][n]));
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- epilogue (MODE 0 of w4conv_kernel): lane holds out[pixel n of quad li][co = 32 wave + 16 m + 4 lg .. + 3]
    const mdil_epilogue& e = a.e;
    constexpr int NE = EOPS ? W4_NE_EOPS : 4;
#pragma unroll
    for (int n0 = 0; n0 < TN; n0 += NE) {
      long long pb[NE];
#pragma unroll
      for (int n = 0; n < NE; ++n) pb[n] = (long long)(okA ? P0A + (n0 + n) * S : 0) * C + wave * COW + lg * 4;
      f32x4 r1[NE][TM], r2[NE][TM];
      auto ld_tile = [&](f32x4 (&r)[NE][TM], const float* p) __attribute__((always_inline)) {
#pragma unroll
        for (int n = 0; n < NE; ++n)
#pragma unroll
          for (int m = 0; m < TM; ++m) r[n][m] = *reinterpret_cast<const f32x4*>(p + pb[n] + m * 16);
      };
      const float* opa = EOPS ? (e.res ? e.res : e.gate) : nullptr;
      const float* opb = EOPS ? e.res_gate : nullptr;
      if (opa) ld_tile(r1, opa);
      if (opb) ld_tile(r2, opb);
#pragma unroll
      for (int m = 0; m < TM; ++m) {
        const f32x4 vscale = *reinterpret_cast<const f32x4*>(&Ep[wave * COW + m * 16 + lg * 4]);
        const f32x4 vbias = *reinterpret_cast<const f32x4*>(&Ep[C + wave * COW + m * 16 + lg * 4]);
#pragma unroll
        for (int n = 0; n < NE; ++n) {
          f32x4 x_ = ay[m][n0 + n];
          x_ = x_ * vscale + vbias;
          if (EOPS && e.res) {
            f32x4 x = r1[n][m];
            if (e.res_gate) {
#pragma unroll
              for (int k = 0; k < 4; ++k) x[k] = r2[n][m][k] > 0.f ? x[k] : 0.f;
            }
            x_ += x;
          }
          if (e.relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) x_[k] = fmaxf(x_[k], 0.f);
          }
          if (EOPS && e.gate && !e.res) {
#pragma unroll
            for (int k = 0; k < 4; ++k) x_[k] = r1[n][m][k] > 0.f ? x_[k] : 0.f;
          }
          ay[m][n0 + n] = x_;
        }
      }
      w4_store<C, COW, TM, NE>(a.out, ay, n0, P0A, okA, S, wave, li, lg);
      if (n0 + NE < TN) __builtin_amdgcn_sched_barrier(0);
    }

    __syncthreads();      // image buf^1 complete, image buf free for the tile after next
    tile = ntile;
#pragma unroll
    for (int k = 0; k < 3; ++k) vbA[k] = vbB[k];
    P0A = P0B;
    okA = okB;
    buf ^= 1;
  }
}

template <bool EOPS>
int launch_r4conv(const wconv_args& a, hipStream_t st) {
  const int ntiles = (int)(((long long)a.N * a.H * a.W / 4 + W4_QUADS - 1) / W4_QUADS);
  int nwg = wc_num_cu();
  if (nwg > ntiles) nwg = ntiles;
  hipLaunchKernelGGL((r4conv_kernel<EOPS>), dim3(nwg), dim3(R4_THREADS), 0, st, a);
  MDIL_CHECK_LAUNCH();
  return MDIL_OK;
}

bool r4conv_enabled() {
  static const bool on = getenv("MDIL_NO_R4CONV") == nullptr;
  return on;
}

