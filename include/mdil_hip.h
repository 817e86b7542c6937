/*
 * libmdil_hip.so -- C ABI of the MI355X (gfx950) ERFNet-RAP step-2 training path.
 *
 * The reference (prachigarg23/MDIL-SS) is pure Python on torch; it has no FFI layer.  Its seam
 * is the set of ATen operators its nn.Modules dispatch.  Each entry point below
 * replaces one of those operator families, for NHWC fp32 tensors, and cites the reference call
 * site it stands in for.  Conventions:
 *
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 *     (PyTorch's caching allocator in our host side); the library allocates no device memory and
 *     every call is re-entrant (forward on the main thread, backward on autograd worker threads).
 *     Process-wide state it does keep, none of which a result depends on: the A/B switches read
 *     ONCE from the environment (MDIL_NO_SCONV / _WCONV / _W4CONV / _WGRADW / _WGRAD16 / _BNFUSE / _BNTAIL /
 *     _C16CONV, function-local statics), the cached compute-unit count of the device, the
 *     thread-local error text, and the launch profiler's record buffer between
 *     mdil_profile_begin and mdil_profile_end (csrc/prof.cpp; measurement only, mutex-guarded).
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued there, no implicit sync.
 *   - return 0 on success, negative on error; mdil_last_error() gives thread-local text.
 *   - activations are NHWC ("channels_last" storage); weights are passed in the reference's
 *     PyTorch layout and re-packed on device by mdil_pack_weights.
 *   - arithmetic is fp32 throughout: contractions run on v_mfma_f32_16x16x4_f32 (fp32 products,
 *     fp32 accumulation), everything else on the fp32 VALU.  The 3-tap convs, their dgrads and
 *     weight gradients take the Winograd F(2,3) form along the conv axis where the axis length
 *     allows (sums / differences of two inputs and of the taps are formed in fp32 first): results
 *     differ from the direct form by a few ulp, measured against fp64 in DESIGN.md 3.0.
 */
#ifndef MDIL_HIP_H
#define MDIL_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDIL_OK 0
#define MDIL_ERR_INVALID (-1)
#define MDIL_ERR_LAUNCH (-2)
#define MDIL_ERR_UNSUPPORTED (-3)

#define MDIL_MAX_TAPS 9

const char* mdil_last_error(void);
int mdil_version(void);

/* ------------------------------------------------------------------------------------------
 * Geometry of one "tap convolution":   for every pixel p=(n,ho,wo) of an iteration grid
 *      out[n, ho*ohs+oho, wo*ows+owo, co] = epi( sum_t sum_ci  in_{src[t]}[n, ho*ihs+dh[t],
 *                                                   wo*iws+dw[t], ci] * Wpk[t][co][ci] )
 * (out-of-range input pixels read as zero).  Stride-1 (dilated) convs, stride-2 convs,
 * transposed convs (one launch per output parity class) and all of their dgrads are instances.
 * ---------------------------------------------------------------------------------------- */
typedef struct mdil_geom {
  int N, HO, WO;           /* iteration grid */
  int HI, WI;              /* spatial size of the input tensor(s) */
  int ihs, iws;            /* input coordinate scale */
  int ntaps;
  int dh[MDIL_MAX_TAPS], dw[MDIL_MAX_TAPS];
  int src[MDIL_MAX_TAPS];  /* which input tensor (0/1) each tap reads */
  int in_pitch[2];         /* floats per pixel of each input tensor */
  int OH, OW;              /* spatial size of the output tensor */
  int ohs, oho, ows, owo;  /* output coordinate map */
  int out_pitch, out_coff; /* floats per output pixel, first output channel */
} mdil_geom;

/* epilogue of mdil_tapconv:  v = acc + bias[co] (+ bias2[co]);  v = v*scale[co] + shift[co];
 *   v += res[...] (optionally only where res_gate[...] > 0);  v = relu(v);
 *   v = gate[...] > 0 ? v : 0;   out = v.      NULL pointers skip a stage.
 *   res / res_gate / gate are addressed exactly like `out`. */
typedef struct mdil_epilogue {
  const float* bias;
  const float* scale;
  const float* shift;
  const float* res;
  const float* res_gate;
  const float* gate;
  int relu;
  const float* bias2;   /* second bias vector, added to `bias` (the 1x1 adapter's bias riding with
                           the 1x3 conv's: conv1x3 + parallel_conv, erfnet_RA_parallel.py:95-98) */
} mdil_epilogue;

/* Re-pack a conv / transposed-conv weight into the [tap][M_P][K_P] image the MFMA kernels read
 * (zero padded):  dst[t][m][k] = src[m*s_m + k*s_k + ktap[t]]   (stem != 0: the 3x3 RGB stem's
 * im2col image dst[0][co][3*tap+c] = src[co][c][tap]).
 * replaces: implicit cuDNN filter transforms of nn.Conv2d / nn.ConvTranspose2d
 * (models/erfnet_RA_parallel.py:17,72-76,93-98,155,179). */
typedef struct mdil_pack_job {
  const float* src;
  float* dst;
  int ntaps, M, K, M_P, K_P, s_m, s_k, stem;
  int ktap[MDIL_MAX_TAPS];
} mdil_pack_job;
int mdil_pack_weights(const float* src, float* dst, int ntaps, const int* ktap, int M, int K,
                      int M_P, int K_P, int s_m, int s_k, int stem, void* stream);
/* the same for a table of jobs resident in DEVICE memory: one launch refreshes every packed
 * image of a model after an optimizer step. */
int mdil_pack_weights_batch(const mdil_pack_job* jobs_device, int njobs, void* stream);

/* Generic MFMA tap convolution (forward convs, adapters, dgrads).  cin/cout select a compiled
 * tile configuration; cin == 27 selects the 3x3-stride-2 RGB stem (im2col-on-load).
 * replaces: F.conv2d / F.conv_transpose2d forward and backward-data on the hot path
 * (models/erfnet_RA_parallel.py:23,54-58,92-107,159,188). */
int mdil_tapconv(const mdil_geom* g, int cin, int cout, const float* in0, const float* in1,
                 const float* wpk, const mdil_epilogue* epi, float* out, void* stream);

/* The same with the train-mode BatchNorm statistics of the STORED output riding along (the
 * reference's conv -> bn pairs, models/erfnet_RA_parallel.py:95-100,105-109): the launch also
 * writes nblk = mdil_tapconv_stat_blocks(...) partial summaries, partial[nblk][2][C] (mean, M2) and
 * pcount[nblk] (pixels), which mdil_bn_train_finalize merges in a fixed order -- the separate
 * statistics pass over the tensor (mdil_bn_train_stats) disappears.  stat_blocks returns 0 when
 * the call is not covered (then use mdil_tapconv + mdil_bn_train_stats). */
int mdil_tapconv_stat_blocks(const mdil_geom* g, int cin, int cout);
int mdil_tapconv_stats(const mdil_geom* g, int cin, int cout, const float* in0, const float* in1,
                       const float* wpk, const mdil_epilogue* epi, float* out, float* partial,
                       float* pcount, void* stream);

/* TICKETS: finalize inside the producing launch.  Turning a producer's <= 256 partial rows into
 * per-channel coefficients used to be a launch of its own between every conv and the pass that
 * waits for it (156 per step-2 iteration).  Entry points that take a `ticket` let the producer's
 * LAST-ARRIVING work-group do it instead (agent-scope hand-off, csrc/bnfin.h): `ticket` is a DEVICE
 * unsigned int owned by the caller, zero before its first use; every launch leaves it zero again,
 * so the launches of one stream can share one word (launches that may run concurrently -- other
 * streams -- need their own).  NULL = a separate finalize launch follows inside the call; results
 * are bit-identical either way (same device code, same fixed merge order). */

/* conv -> train-mode BatchNorm statistics -> coefficients in ONE call (conv1x3 [+ adapter] -> bn of
 * models/erfnet_RA_parallel.py:95-100,105-109): `out` is written, bn->coef receives [4][C]
 * (save_mean, save_invstd, scale, shift), running statistics / num_batches_tracked are updated.
 * Where the streaming conv covers the call the statistics ride in its epilogue (and, with a ticket,
 * so does the finalize); otherwise mdil_tapconv + mdil_bn_train_stats run inside.
 * workspace: mdil_bn_workspace(N*HO*WO, cout) bytes. */
typedef struct mdil_bn_train {
  const float *gamma, *beta;
  float *running_mean, *running_var;         /* NULL: not tracked */
  long long* num_batches_tracked;            /* NULL: not tracked */
  float eps, momentum;
  float* coef;                               /* out [4][C] */
} mdil_bn_train;
int mdil_tapconv_bn_train(const mdil_geom* g, int cin, int cout, const float* in0, const float* in1,
                          const float* wpk, const mdil_epilogue* epi, float* out,
                          const mdil_bn_train* bn, void* workspace, size_t workspace_bytes,
                          unsigned int* ticket, void* stream);

/* What the finalize of a BatchNorm BACKWARD's reductions needs and produces: dgamma / dbeta
 * (accumulate: +=; NULL = not wanted) and coef [3][C] = gamma * invstd, sum(g) / n, sum(g * xhat) / n
 * for mdil_bn_backward_apply. */
typedef struct mdil_bn_grad {
  const float* gamma;
  float *dgamma, *dbeta;
  int accumulate;
  float* coef;                               /* out [3][C] */
  unsigned int* ticket;                      /* NULL: stand-alone finalize launch inside the call */
} mdil_bn_grad;

/* A dgrad launch whose stored, gated gradient g is the input of a BatchNorm backward (the inner
 * BN of a factorised block: g = conv3x1^T(...) * (u > 0), models/erfnet_RA_parallel.py:99-103 in
 * reverse): the BN-backward reductions sum(g), sum(g * xhat) ride in the epilogue ->
 * partial[nblk][2][C] (nblk = mdil_tapconv_stat_blocks).  fin == NULL: the caller continues with
 * mdil_bn_backward_partials (finalize + apply); fin != NULL: the reductions are finalized by the
 * call (inside the launch with fin->ticket) and the caller continues with mdil_bn_backward_apply. */
int mdil_tapconv_bnred(const mdil_geom* g, int cin, int cout, const float* in0, const float* in1,
                       const float* wpk, const mdil_epilogue* epi, float* out, const float* bn_z,
                       const float* save_mean, const float* save_invstd, float* partial,
                       const mdil_bn_grad* fin, void* stream);

/* Block-boundary fusion of the OUTER BatchNorm backward of a factorised block
 * (out = relu(bn2(z2) * drop + x), models/erfnet_RA_parallel.py:105-113 in reverse).  The gradient
 * dL/dout arrives from the NEXT block, whose last backward launch -- the dgrad that produces its
 * input gradient gx -- already holds it.  In the tail form that launch
 *   (a) stores gx * (gate > 0), gate = its own input = the previous block's `out` (every consumer of
 *       gx applies exactly this ReLU gate, so storing the gated value changes no result), and
 *   (b) emits partial[nblk][2][C] = sum(g), sum(g * xhat) with g = gated gx * drop[image][c] and
 *       xhat = (z - save_mean) * save_invstd of the previous block's bn2,
 * after which that block runs mdil_bn_backward_partials (finalize + apply) instead of
 * mdil_bn_backward's reduction pass over gy, out and z2.  nblk = mdil_tapconv_tail_blocks (0 = not
 * covered: plain mdil_tapconv, and mdil_bn_backward in the previous block).  The epilogue may carry
 * the residual (res, res_gate); not bias / scale / relu / gate. */
typedef struct mdil_bn_tail {
  const float* gate;         /* [N,H,W,C]: the previous block's output (this block's input) */
  const float* z;            /* [N,H,W,C]: the previous block's bn2 input z2 */
  const float* save_mean;    /* [C] */
  const float* save_invstd;  /* [C] */
  const float* drop;         /* [N][C] Dropout2d factors of the previous block, NULL = none */
  float* partial;            /* out: [nblk][2][C] */
  /* optional: finalize the reductions in the tail launch itself (the previous block's bn2: its
   * gamma, gradient sinks, and the [3][C] table its mdil_bn_backward_apply will read).
   * fin.coef == NULL: partial rows only (the previous block runs mdil_bn_backward_partials). */
  mdil_bn_grad fin;
} mdil_bn_tail;
int mdil_tapconv_tail_blocks(const mdil_geom* g, int cin, int cout);
int mdil_tapconv_tail(const mdil_geom* g, int cin, int cout, const float* in0, const float* in1,
                      const float* wpk, const mdil_epilogue* epi, float* out, const mdil_bn_tail* tail,
                      void* stream);

/* Weight gradient of a tap convolution: partial[chunk][t][co][ci] over pixel chunks (MFMA,
 * split-K), then a fixed-order reduction into the PyTorch-layout gradient
 * (dst[co*s_co + ci*s_ci + ktap[t]]) and, optionally, the bias gradient (column sums of g).
 * `g` is addressed through the geometry's OUTPUT map, `in0/in1` through its input map.
 * replaces: cuDNN backward-filter + bias reduction (autograd of the convs above). */
size_t mdil_wgrad_workspace(const mdil_geom* g, int cin, int cout);
int mdil_wgrad(const mdil_geom* g, int cin, int cout, const float* in0, const float* in1,
               const float* gout, const int* ktap, int s_co, int s_ci, float* dw, float* dbias,
               /* the trailing ntaps2 taps go to a second weight (dst2[co*s_co2+ci*s_ci2+ktap[t]],
                * same bias gradient): the 1x1 adapter summed with a 1x3 conv is its 4th tap */
               int ntaps2, int s_co2, int s_ci2, float* dw2, float* dbias2,
               int accumulate /* 0: dw = ..., 1: dw += ... (only the taps in ktap are touched) */,
               void* workspace, size_t workspace_bytes, void* stream);

/* Deferred reduction: the same launch, but the partial sums STAY in `workspace` (caller-owned; it
 * must not be reused before the batch reduction below has been enqueued on the same stream) and
 * *job receives the description of the pending reduction (dw += ..., always accumulating).
 * mdil_wgrad_reduce_batch then performs any number of pending reductions with one launch per 16
 * jobs.  Jobs of one batch must not target the same gradient elements.  Exists because the ~140
 * 8-microsecond reductions of a step, sitting between chip-filling MFMA launches, cost 4 % of it. */
typedef struct mdil_wgrad_job { unsigned char opaque[192]; } mdil_wgrad_job;
int mdil_wgrad_deferred(const mdil_geom* g, int cin, int cout, const float* in0, const float* in1,
                        const float* gout, const int* ktap, int s_co, int s_ci, float* dw,
                        float* dbias, int ntaps2, int s_co2, int s_ci2, float* dw2, float* dbias2,
                        void* workspace, size_t workspace_bytes, mdil_wgrad_job* job, void* stream);
int mdil_wgrad_reduce_batch(const mdil_wgrad_job* jobs /* host memory */, int njobs, void* stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm2d(eps=1e-3, momentum=0.1) on NHWC tensors.
 * replaces: F.batch_norm train/eval + backward (models/erfnet_RA_parallel.py:24,58,63,100,109,160)
 * ---------------------------------------------------------------------------------------- */
size_t mdil_bn_workspace(long long npix, int C);
/* train-mode statistics: Welford/Chan tree over pixels (fixed order) -> save_mean, save_invstd,
 * scale = gamma*invstd, shift = beta - mean*scale; running stats updated in place (unbiased
 * variance), *num_batches_tracked += 1 when non-NULL. */
int mdil_bn_train_stats(const float* z, long long npix, int C, const float* gamma,
                        const float* beta, float* running_mean, float* running_var,
                        long long* num_batches_tracked, float eps, float momentum,
                        float* save_mean, float* save_invstd, float* scale, float* shift,
                        void* workspace, size_t workspace_bytes, unsigned int* ticket /* see TICKETS */,
                        void* stream);
/* second half of mdil_bn_train_stats alone: merge nblk partial (mean, M2, count) summaries
 * produced elsewhere (mdil_tapconv_stats) -> coefficients + running statistics. */
int mdil_bn_train_finalize(const float* partial, const float* pcount, int nblk, int C,
                           const float* gamma, const float* beta, float* running_mean,
                           float* running_var, long long* num_batches_tracked, float eps,
                           float momentum, float* save_mean, float* save_invstd, float* scale,
                           float* shift, void* stream);
/* eval-mode coefficients from running statistics */
int mdil_bn_eval_coeffs(int C, const float* gamma, const float* beta, const float* running_mean,
                        const float* running_var, float eps, float* scale, float* shift,
                        void* stream);
/* y = relu?( (z*scale+shift) * drop[n][c]  +  res )        (drop / res may be NULL) */
int mdil_bn_apply(const float* z, long long npix, int pix_per_image, int C, const float* scale,
                  const float* shift, const float* drop, const float* res, int relu, float* y,
                  void* stream);
/* backward:  g = gy * (relu_src > 0) * drop[n][c];   dbeta = sum g;  dgamma = sum g*xhat;
 *            gz = gamma*invstd * (g - dbeta/n - xhat*dgamma/n).   (dgamma/dbeta may be NULL) */
int mdil_bn_backward(const float* gy, const float* relu_src, const float* drop, const float* z,
                     long long npix, int pix_per_image, int C, const float* gamma,
                     const float* save_mean, const float* save_invstd, float* dgamma,
                     float* dbeta, int accumulate /* dgamma/dbeta += */, float* gz,
                     void* workspace, size_t workspace_bytes, unsigned int* ticket /* see TICKETS */,
                     void* stream);

/* the same given the reductions (mdil_tapconv_bnred / mdil_tapconv_tail) and the already gated
 * gradient g: finalize + apply only; `drop` ([N][C], NULL = none) is the Dropout2d factor the
 * reductions were taken with (gz uses g * drop).  workspace: 3*C floats. */
int mdil_bn_backward_partials(const float* g, const float* drop, const float* z, long long npix,
                              int pix_per_image, int C, const float* gamma, const float* save_mean,
                              const float* save_invstd, const float* partial, int nblk,
                              float* dgamma, float* dbeta, int accumulate, float* gz,
                              void* workspace, size_t workspace_bytes, void* stream);
/* the last pass alone: the reductions were produced AND finalized elsewhere (coef [3][C] of an
 * mdil_bn_grad: mdil_tapconv_bnred / mdil_tapconv_tail with fin):
 *            gz = coef[0] * (g * drop - coef[1] - xhat * coef[2]) */
int mdil_bn_backward_apply(const float* g, const float* drop, const float* z, long long npix,
                           int pix_per_image, int C, const float* save_mean, const float* save_invstd,
                           const float* coef, float* gz, void* stream);

/* ------------------------------------------------------------------------------------------
 * MaxPool2d(2, stride 2) half of DownsamplerBlock, written into / read from the channel slice
 * [coff, coff+C) of the concatenated tensor (models/erfnet_RA_parallel.py:18,23).
 * ---------------------------------------------------------------------------------------- */
int mdil_maxpool_concat_fwd(const float* x, int N, int H, int W, int C, float* z, int z_pitch,
                            int coff, void* stream);
/* gx (fully written) = scatter of gz[..., coff:coff+C] to the first-max position of each window */
int mdil_maxpool_concat_bwd(const float* x, const float* gz, int N, int H, int W, int C,
                            int z_pitch, int coff, float* gx, void* stream);

/* ------------------------------------------------------------------------------------------
 * Decoder.output_conv forward: ConvTranspose2d(16, nc, 2, stride 2) (models/erfnet_RA_parallel.py:
 * 179-180,188) in ONE pass: x [N,H,W,16] -> out [N,2H,2W,pitch] (a pixel's nc logits in a row of
 * `pitch` floats, pad written 0); w is the PyTorch weight [16][nc][2][2] as it is (no packing).
 * (backward: mdil_tapconv / mdil_wgrad on the same tensors.)
 * ---------------------------------------------------------------------------------------- */
int mdil_outconv_fwd(const float* x, const float* w, const float* bias, int N, int H, int W, int nc,
                     int pitch, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Block level: one call = one non_bottleneck_1d / non_bottleneck_1d_RAP block
 * (models/erfnet_RA_parallel.py:68-116 RAP encoder block, :31-66 plain decoder block), forward or
 * backward.  The call enqueues, on `stream`, exactly the launches listed in DESIGN.md 3 for the
 * block (6 forward in train mode, 3 in eval mode, 8-10 backward) through the entry points above;
 * it exists so that a host pays one foreign call per block and direction instead of one per
 * launch.  All tensors are NHWC [N,H,W,C] fp32, C in {16, 64, 128}.
 *
 *   a1 = relu(conv3x1_1(x));  z1 = conv1x3_1(a1) [+ parallel_conv_1(x)];  u  = relu(bn1(z1))
 *   a2 = relu(conv3x1_2(u));  z2 = conv1x3_2(a2) [+ parallel_conv_2(u)];  out = relu(bn2(z2)*drop + x)
 * ---------------------------------------------------------------------------------------- */
typedef struct mdil_nb_half {   /* conv3x1 -> relu -> conv1x3 (+ 1x1 adapter) -> bn  */
  /* forward: packed images (mdil_pack_weights) "fwd" of the 3x1 [3][C][C] and of the 1x3 with the
   * adapter as 4th tap [3 or 4][C][C].  backward: the 1x3 "dgrad" image in wp13 and the 3x1 "dgrad"
   * image with the adapter^T as 4th tap in wp31. */
  const float *wp31, *wp13;
  const float *b31, *b13, *pb;           /* biases (pb NULL without adapter) */
  const float *gamma, *beta;             /* this domain's BatchNorm */
  float *running_mean, *running_var;
  long long* num_batches_tracked;
  float* coef;                           /* [4][C] save_mean, save_invstd, scale, shift: written by the
                                            train forward, read by the backward; eval: scratch [2][C] */
  /* backward: ACCUMULATION targets in PyTorch layout; NULL = gradient not wanted (frozen) */
  float *dw31, *db31, *dw13, *db13, *dpw, *dpb, *dgamma, *dbeta;
} mdil_nb_half;

typedef struct mdil_nb_block {
  int N, H, W, C;
  int dilation;                          /* of the second half (the first is always 1) */
  int rap;                               /* 1: parallel 1x1 adapters present */
  int train;                             /* forward only: batch statistics (1) or running statistics (0) */
  float bn_eps, bn_momentum;             /* 1e-3, 0.1 in the reference */
  mdil_nb_half half[2];
  const float* x;
  const float* drop;                     /* Dropout2d factors [N][C] (0 or 1/(1-p)); NULL = none */
  /* forward outputs, saved for the backward (train).  eval: a1, u, out only (a2 may alias a1) */
  float *a1, *z1, *u, *a2, *z2, *out;
  /* backward */
  const float* gy;                       /* dL/dout */
  float *gz2, *ga, *gu, *gx;             /* scratch [N,H,W,C] x3 and the result dL/dx */
  void* bn_workspace;   size_t bn_workspace_bytes;     /* >= mdil_bn_workspace(N*H*W, C) */
  void* wgrad_workspace; size_t wgrad_workspace_bytes; /* >= mdil_nb_block_wgrad_workspace(...) */
  /* backward, block-boundary fusion (mdil_tapconv_tail; both optional, NULL / 0 = off):
   * head: gy arrives ALREADY gated by out > 0 together with the reductions of this block's bn2
   *       backward (written by the next block's tail launch) -> no reduction pass here;
   * tail: this block's last launch gates gx by x > 0 and emits the reductions of the PREVIOUS
   *       block's bn2 backward into tail.partial (tail.gate is ignored: it is b->x). */
  const float* head_partial; int head_nblk;
  mdil_bn_tail tail;
  /* head, finalized form: the next block's tail launch has already turned the reductions into the
   * [3][C] table (tail.fin of THAT call pointed here) and added dgamma / dbeta of this block's bn2:
   * only the apply pass runs.  Takes precedence over head_partial. */
  const float* head_coef;
  /* TICKETS (above): finalize every BatchNorm step of the block inside its producing launch */
  unsigned int* ticket;
  /* eval-mode forward: != 0 = half[h].coef already holds the folded coefficients [2][C] (scale,
   * shift = mdil_bn_eval_coeffs of the four BatchNorm tensors) -- a frozen model's never change, so
   * its caller computes them once instead of two launches per block and forward; 0 = the call
   * computes them into coef itself. */
  int eval_coef_ready;
} mdil_nb_block;
/* number of partials a tail launch of this block shape emits (0: the shape is not covered) */
int mdil_nb_block_tail_blocks(int N, int H, int W, int C, int rap);

size_t mdil_nb_block_wgrad_workspace(int N, int H, int W, int C, int dilation, int rap);
int mdil_nb_block_forward(const mdil_nb_block* b, void* stream);
int mdil_nb_block_backward(const mdil_nb_block* b, void* stream);
/* the same with deferred weight-gradient reductions: every weight-gradient launch of the block takes
 * its own slice of b->wgrad_workspace (consecutive, *workspace_used bytes in total) and appends its
 * job to jobs[0 .. *njobs) (room for 4 needed); the caller later runs mdil_wgrad_reduce_batch */
int mdil_nb_block_backward_deferred(const mdil_nb_block* b, mdil_wgrad_job* jobs, int* njobs,
                                    size_t* workspace_used, void* stream);

/* ------------------------------------------------------------------------------------------
 * Decoder.output_conv FUSED with the loss that consumes its logits: the logits are never
 * materialised (csrc/head.hip).  Replaces, for the training step,
 *   outputs = decoder[t].output_conv(y)            models/erfnet_RA_parallel.py:179-180,188
 *   loss = criterion(outputs, targets[:, 0])       train_new_task_step2.py:293      (mdil_head_ce)
 *   loss_kld = KLDivLoss()(softmax(outputs_prev_task), softmax(outputs_prev_model))  :296-297
 *                                                                                   (mdil_head_kld)
 * and their backward passes.  x: [N,H,W,16] decoder features; w: ConvTranspose2d weight
 * [16][nc][2][2], bias [nc]; target: int64 [N,2H,2W]; nc in {20, 27}.
 * FORWARD call: grad_scale = gx = NULL -> loss[0] (and, for CE, wsum[0] = sum of the target pixels'
 * class weights, which the backward call needs; logits_out, optional, receives the logits in rows
 * of r4(nc) floats for callers that want them, e.g. --iouTrain).
 * BACKWARD call: grad_scale = DEVICE scalar dL/dloss, gx [N,H,W,16] = dL/dx (written); dw / db (both
 * or neither; NULL = frozen head) receive the weight / bias gradient (accumulate: +=).  The logits
 * are recomputed from x.  workspace: mdil_head_workspace() bytes.
 * ---------------------------------------------------------------------------------------- */
size_t mdil_head_workspace(void);
int mdil_head_ce(const float* x, const float* w, const float* bias, int N, int H, int W, int nc,
                 const long long* target, const float* class_weight, const float* grad_scale,
                 float* loss, float* wsum, float* gx, float* dw, float* db, int accumulate,
                 float* logits_out, int* label_errors, void* workspace, size_t workspace_bytes,
                 void* stream);
int mdil_head_kld(const float* xs, const float* ws, const float* bs, const float* xt, const float* wt,
                  const float* bt, int N, int H, int W, int nc, const float* grad_scale, float* loss,
                  float* gx, float* dw, float* db, int accumulate, void* workspace,
                  size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Losses on NHWC logits: a pixel's C classes sit in a row of `pitch` floats (pitch = C = 20, or
 * 28 for the 27-class head so rows stay 16-byte aligned; pad entries are ignored / written 0).
 * ---------------------------------------------------------------------------------------- */
size_t mdil_loss_workspace(long long npix);
/* CrossEntropyLoss2d = NLLLoss2d(weight)(log_softmax(x,1), y)  (train_new_task_step2.py:84-92):
 * loss[0] = -sum w[y]*logp[y] / sum w[y];
 * dlogits (may be NULL) = grad_scale[0] * d loss / d logits  (grad_scale: DEVICE scalar, NULL = 1,
 * so the upstream autograd gradient never has to visit the host).
 * label_errors (may be NULL): DEVICE counter incremented for every target outside [0, C) -- torch
 * raises a device assert there; here such a pixel is dropped (weight 0) and the caller raises
 * when it next reads the counter (no host sync inside the library). */
int mdil_ce_loss(const float* logits, const long long* target, const float* weight,
                 long long npix, int C, int pitch, const float* grad_scale, float* loss, float* dlogits,
                 int* label_errors, void* workspace, size_t workspace_bytes, void* stream);
/* KLDivLoss()(softmax(s), softmax(t)) with the reference's quirk (probabilities as input,
 * 'mean' over all elements; train_new_task_step2.py:241,296-297):
 * loss[0] = mean( t*(log t - p_s) );  ds (may be NULL) = grad_scale[0] * d loss / d s. */
int mdil_kld_loss(const float* s_logits, const float* t_logits, long long npix, int C,
                  int pitch, const float* grad_scale, float* loss, float* ds, void* workspace,
                  size_t workspace_bytes, void* stream);
/* eval: argmax over C + confusion counts (iouEval.addBatch, iouEval.py:21-70) accumulated into
 * counts[3][C] (tp, fp, fn as int64); pixels with target == ignore are dropped; targets outside
 * [0, C) are dropped and counted in label_errors (may be NULL). */
int mdil_argmax_confusion(const float* logits, const long long* target, long long npix, int C,
                          int pitch, int ignore, long long* counts, int* label_errors, void* stream);

/* ------------------------------------------------------------------------------------------
 * Adam with L2 weight decay on a flat fp32 segment (torch.optim.Adam semantics,
 * train_new_task_step2.py:237-239,306).  Hyper-parameters and bias corrections arrive as doubles
 * (1-beta, lr/bc1, sqrt(bc2) are formed in double like torch does, then rounded to fp32).
 * grad_scale multiplies the gradient first (1/world_size after an all-reduce SUM).
 * ---------------------------------------------------------------------------------------- */
int mdil_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                   long long n, double lr, double beta1, double beta2, double eps,
                   double weight_decay, double bias_correction1, double bias_correction2,
                   double grad_scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Input pipeline, device half (MyCoTransform, train_new_task_step2.py:48-81; transform.py:62-80):
 * img_u8 [N,H,W,3] / lab_u8 [N,H,W] are the PIL-resized bytes; params[n] = {hflip, transX,
 * transY} drawn on the host in the reference's order (random.random() < 0.5, randint(-2,2) x2).
 * Per pixel: flip, translate (expand border filled with 0 / 255, crop overhang filled with 0 for
 * image AND label, as PIL does), ToTensor (/255), ToLabel, Relabel(relabel_from -> relabel_to).
 * out_img: NHWC fp32 [N,H,W,3]; out_lab: int64 [N,H,W].
 * ---------------------------------------------------------------------------------------- */
int mdil_augment_batch(const unsigned char* img_u8, const unsigned char* lab_u8, const int* params,
                       int N, int H, int W, int relabel_from, int relabel_to, float* out_img,
                       long long* out_lab, void* stream);

/* The reference hands the model NCHW float images (ToTensor, train_new_task_step2.py:76; Net.forward
 * models/erfnet_RA_parallel.py:207); the path's kernels read NHWC.  in [N,C,H,W] -> out [N,H,W,C], C <= 32
 * (replaces the ATen permute + contiguous copy in front of the stem). */
int mdil_nchw_to_nhwc(const float* in, int N, int C, int H, int W, float* out, void* stream);

/* nn.Dropout2d (models/erfnet_RA_parallel.py:88,110-111) for all encoder blocks of a forward pass at once:
 * out[e] = uniform[e] < keep[e] ? inv_keep[e] : 0 over the n = sum_blocks N * C_block (image, channel)
 * elements; `uniform` is one draw of the caller's generator (torch's device RNG, as the reference uses). */
int mdil_dropout_factors(const float* uniform, const float* keep, const float* inv_keep, float* out, int n,
                         void* stream);

/* ---- per-launch timing (measurement only: bench.py's roofline leg) --------------------------------
 * The reference times its iteration with time.time() around the loop (train_new_task_step2.py:276,
 * 309-313); there is no per-operator timing to replace.  Between _begin and _end every conv /
 * weight-gradient entry point above (called directly or from inside mdil_nb_block_*) brackets its
 * launch with HIP events on the launch stream, so what is timed is the shipped call sequence.
 * mdil_profile_end synchronises the events and returns the number of records written.
 * kind: 0 = conv (mdil_tapconv*), 1 = weight gradient (mdil_wgrad*).
 * path: conv 0 = LDS-tiled tapconv, 1 = streaming sconv, 2 = Winograd F(2,3) wconv, 3 = c16conv,
 *   4 = Winograd F(4,3) w4conv;
 *       weight gradient 0 = LDS-tiled, 1 = streaming wgrad2, 2 = Winograd wgradw / wgradx. */
typedef struct mdil_profile_record {
  int kind, path, cin, cout, ntaps;
  long long npix;      /* N * HO * WO of the launch geometry */
  float ms;            /* launch duration on its stream; < 0 when the events could not be read */
} mdil_profile_record;
int mdil_profile_begin(int capacity);
int mdil_profile_end(mdil_profile_record* out, int max_records);

#ifdef __cplusplus
}
#endif
#endif /* MDIL_HIP_H */
