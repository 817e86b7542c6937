// Block-level entry points: one foreign call enqueues every launch of a factorised residual block
// (non_bottleneck_1d / non_bottleneck_1d_RAP, models/erfnet_RA_parallel.py:31-116 of the
// reference), forward or backward.  Host code only: the launches go through the per-kernel entry
// points of this library, in the order and with the fusions DESIGN.md 3 describes.
#include "common.h"

namespace {

mdil_geom geom(const mdil_nb_block* b, int dil, bool along_w, bool flip, bool adapter) {
  mdil_geom g;
  memset(&g, 0, sizeof(g));
  g.N = b->N;
  g.HO = g.HI = g.OH = b->H;
  g.WO = g.WI = g.OW = b->W;
  g.ihs = g.iws = g.ohs = g.ows = 1;
  g.ntaps = adapter ? 4 : 3;
  const int s = flip ? -1 : 1;
  for (int k = 0; k < 3; ++k) {
    (along_w ? g.dw : g.dh)[k] = s * (k - 1) * dil;
  }
  if (adapter) g.src[3] = 1;  // centre tap of the second source
  g.in_pitch[0] = g.in_pitch[1] = b->C;
  g.out_pitch = b->C;
  return g;
}

#define TRY(call)          \
  do {                     \
    int rc__ = (call);     \
    if (rc__) return rc__; \
  } while (0)

int check_common(const mdil_nb_block* b) {
  MDIL_CHECK_ARG(b != nullptr, "nb_block: NULL descriptor");
  MDIL_CHECK_ARG(b->N > 0 && b->H > 0 && b->W > 0, "nb_block: empty tensor");
  MDIL_CHECK_ARG(b->C == 16 || b->C == 64 || b->C == 128, "nb_block: C=%d not in {16,64,128}", b->C);
  MDIL_CHECK_ARG(b->dilation >= 1, "nb_block: dilation %d", b->dilation);
  MDIL_CHECK_ARG(b->x != nullptr, "nb_block: x is NULL");
  for (int h = 0; h < 2; ++h) {
    const mdil_nb_half& p = b->half[h];
    MDIL_CHECK_ARG(p.wp31 && p.wp13, "nb_block: half %d packed weights missing", h);
    MDIL_CHECK_ARG(p.gamma != nullptr && p.coef != nullptr, "nb_block: half %d BatchNorm missing", h);
  }
  return MDIL_OK;
}

// statistics / reductions partial layout inside the BatchNorm workspace (bn.hip): partial
// [256][2][C], pcount [256], then the backward's finalize scratch
inline float* ws_partial(const mdil_nb_block* b) { return (float*)b->bn_workspace; }
inline float* ws_pcount(const mdil_nb_block* b) { return ws_partial(b) + 256 * 2 * b->C; }
inline float* ws_finalize(const mdil_nb_block* b) { return ws_pcount(b) + 256; }

// z = conv1x3(a) [+ adapter(inp)] + biases, with the train-mode statistics -> coef
int conv_bn_train(const mdil_nb_block* b, const mdil_geom& g, const mdil_nb_half& p, const float* a,
                  const float* inp, float* z, void* st) {
  const int C = b->C;
  mdil_epilogue e;
  memset(&e, 0, sizeof(e));
  e.bias = p.b13;
  e.bias2 = b->rap ? p.pb : nullptr;
  mdil_bn_train bn;
  bn.gamma = p.gamma, bn.beta = p.beta;
  bn.running_mean = p.running_mean, bn.running_var = p.running_var;
  bn.num_batches_tracked = p.num_batches_tracked;
  bn.eps = b->bn_eps, bn.momentum = b->bn_momentum;
  bn.coef = p.coef;
  return mdil_tapconv_bn_train(&g, C, C, a, inp, p.wp13, &e, z, &bn, b->bn_workspace, b->bn_workspace_bytes,
                               b->ticket, st);
}

}  // namespace

extern "C" size_t mdil_nb_block_wgrad_workspace(int N, int H, int W, int C, int dilation, int rap) {
  mdil_nb_block b;
  memset(&b, 0, sizeof(b));
  b.N = N, b.H = H, b.W = W, b.C = C;
  size_t need = 0;
  for (int dil : {1, dilation}) {
    const mdil_geom gs[3] = {geom(&b, dil, true, false, rap != 0), geom(&b, dil, true, false, false),
                             geom(&b, dil, false, false, false)};
    for (const mdil_geom& g : gs) {
      size_t n = mdil_wgrad_workspace(&g, C, C);
      if (n > need) need = n;
    }
  }
  return need;
}

extern "C" int mdil_nb_block_forward(const mdil_nb_block* b, void* st) {
  TRY(check_common(b));
  MDIL_CHECK_ARG(b->a1 && b->u && b->out, "nb_block_forward: output tensors missing");
  const int C = b->C;
  const long long npix = (long long)b->N * b->H * b->W;
  const int ppi = b->H * b->W;
  const bool rap = b->rap != 0;
  const mdil_geom G31a = geom(b, 1, false, false, false), G13a = geom(b, 1, true, false, rap);
  const mdil_geom G31b = geom(b, b->dilation, false, false, false);
  const mdil_geom G13b = geom(b, b->dilation, true, false, rap);
  const mdil_nb_half &p1 = b->half[0], &p2 = b->half[1];
  mdil_epilogue e;
  memset(&e, 0, sizeof(e));
  e.bias = p1.b31;
  e.relu = 1;
  TRY(mdil_tapconv(&G31a, C, C, b->x, nullptr, p1.wp31, &e, b->a1, st));
  if (b->train) {
    MDIL_CHECK_ARG(b->z1 && b->a2 && b->z2, "nb_block_forward: train-mode tensors missing");
    MDIL_CHECK_ARG(b->bn_workspace != nullptr && b->bn_workspace_bytes >= mdil_bn_workspace(npix, C),
                   "nb_block_forward: BatchNorm workspace too small");
    TRY(conv_bn_train(b, G13a, p1, b->a1, b->x, b->z1, st));
    TRY(mdil_bn_apply(b->z1, npix, ppi, C, p1.coef + 2 * C, p1.coef + 3 * C, nullptr, nullptr, 1, b->u,
                      st));
    e.bias = p2.b31;
    TRY(mdil_tapconv(&G31b, C, C, b->u, nullptr, p2.wp31, &e, b->a2, st));
    TRY(conv_bn_train(b, G13b, p2, b->a2, b->u, b->z2, st));
    return mdil_bn_apply(b->z2, npix, ppi, C, p2.coef + 2 * C, p2.coef + 3 * C, b->drop, b->x, 1,
                         b->out, st);
  }
  // eval: BatchNorm folded into the conv epilogues, Dropout2d is the identity
  float* a2 = b->a2 ? b->a2 : b->a1;
  if (!b->eval_coef_ready) {
    TRY(mdil_bn_eval_coeffs(C, p1.gamma, p1.beta, p1.running_mean, p1.running_var, b->bn_eps, p1.coef,
                            p1.coef + C, st));
    TRY(mdil_bn_eval_coeffs(C, p2.gamma, p2.beta, p2.running_mean, p2.running_var, b->bn_eps, p2.coef,
                            p2.coef + C, st));
  }
  mdil_epilogue f;
  memset(&f, 0, sizeof(f));
  f.bias = p1.b13, f.bias2 = rap ? p1.pb : nullptr, f.scale = p1.coef, f.shift = p1.coef + C, f.relu = 1;
  TRY(mdil_tapconv(&G13a, C, C, b->a1, b->x, p1.wp13, &f, b->u, st));
  e.bias = p2.b31;
  TRY(mdil_tapconv(&G31b, C, C, b->u, nullptr, p2.wp31, &e, a2, st));
  f.bias = p2.b13, f.bias2 = rap ? p2.pb : nullptr, f.scale = p2.coef, f.shift = p2.coef + C;
  f.res = b->x;
  return mdil_tapconv(&G13b, C, C, a2, b->u, p2.wp13, &f, b->out, st);
}

namespace {

// weight-gradient launches of a block: immediate reduction (d == nullptr: all share the workspace)
// or deferred (each takes the next slice of the workspace and appends a job)
struct Defer {
  mdil_wgrad_job* jobs;
  int njobs;
  size_t cursor;
};

int block_wgrad(const mdil_nb_block* b, Defer* d, const mdil_geom& g, const float* in0,
                const float* in1, const float* gout, const int* ktap, int s_co, int s_ci, float* dw,
                float* db, int n2, int s_co2, int s_ci2, float* dw2, float* db2, void* st) {
  const int C = b->C;
  if (d == nullptr)
    return mdil_wgrad(&g, C, C, in0, in1, gout, ktap, s_co, s_ci, dw, db, n2, s_co2, s_ci2, dw2, db2, 1,
                      b->wgrad_workspace, b->wgrad_workspace_bytes, st);
  const size_t need = (mdil_wgrad_workspace(&g, C, C) + 255) / 256 * 256;
  MDIL_CHECK_ARG(d->cursor + need <= b->wgrad_workspace_bytes,
                 "nb_block_backward_deferred: weight-gradient arena exhausted (%zu + %zu > %zu)",
                 d->cursor, need, b->wgrad_workspace_bytes);
  const int rc = mdil_wgrad_deferred(&g, C, C, in0, in1, gout, ktap, s_co, s_ci, dw, db, n2, s_co2, s_ci2,
                                     dw2, db2, (char*)b->wgrad_workspace + d->cursor, need,
                                     &d->jobs[d->njobs], st);
  if (rc == MDIL_OK) {
    d->cursor += need;
    d->njobs += 1;
  }
  return rc;
}

// Backward of  z = conv1x3(relu(conv3x1(inp))) [+ adapter(inp)]  given gz = dL/dz:
// weight gradients, ga = conv1x3^T(gz) * (a > 0) and the input gradient into `ginp`
// (+ res_in where res_gate > 0).  With `bn_z` the input gradient is instead gated by
// relu_src and the reductions of the BatchNorm backward that consumes it ride in the same launch
// (*fused_blocks = number of partials; 0 = not covered, plain store done).
int half_backward(const mdil_nb_block* b, const mdil_nb_half& p, int dil, const float* gz,
                  const float* a, const float* inp, float* ga, float* ginp, const float* res_in,
                  const float* res_gate, const float* relu_src, const float* bn_z,
                  const float* bn_coef, int* fused_blocks, Defer* d, void* st,
                  const mdil_bn_tail* tail = nullptr, const mdil_bn_grad* bn_fin = nullptr) {
  const int C = b->C;
  const bool rap = b->rap != 0;
  static const int kt4[4] = {0, 1, 2, 0}, kt1[1] = {0};
  if (p.dw13 && rap && p.dpw) {
    // one launch: 3 taps of the 1x3 (source a) + the adapter as 4th tap (source inp)
    const mdil_geom G4 = geom(b, dil, true, false, true);
    TRY(block_wgrad(b, d, G4, a, inp, gz, kt4, C * 3, 3, p.dw13, p.db13, 1, C, 1, p.dpw, p.dpb, st));
  } else {
    if (p.dw13) {
      const mdil_geom G3 = geom(b, dil, true, false, false);
      TRY(block_wgrad(b, d, G3, a, nullptr, gz, kt4, C * 3, 3, p.dw13, p.db13, 0, 0, 0, nullptr, nullptr,
                      st));
    }
    if (rap && p.dpw) {
      mdil_geom G1 = geom(b, 1, true, false, false);
      G1.ntaps = 1;
      G1.dw[0] = 0;
      TRY(block_wgrad(b, d, G1, inp, nullptr, gz, kt1, C, 1, p.dpw, p.dpb, 0, 0, 0, nullptr, nullptr, st));
    }
  }
  const mdil_geom G13t = geom(b, dil, true, true, false);
  mdil_epilogue e;
  memset(&e, 0, sizeof(e));
  e.gate = a;
  TRY(mdil_tapconv(&G13t, C, C, gz, nullptr, p.wp13, &e, ga, st));
  if (p.dw31) {
    const mdil_geom G3 = geom(b, dil, false, false, false);
    TRY(block_wgrad(b, d, G3, inp, nullptr, ga, kt4, C * 3, 3, p.dw31, p.db31, 0, 0, 0, nullptr, nullptr,
                    st));
  }
  const mdil_geom G31t = geom(b, dil, false, true, rap);
  *fused_blocks = 0;
  if (bn_z != nullptr) {
    const int nblk = mdil_tapconv_stat_blocks(&G31t, C, C);
    if (nblk > 0) {
      memset(&e, 0, sizeof(e));
      e.gate = relu_src;
      *fused_blocks = nblk;
      return mdil_tapconv_bnred(&G31t, C, C, ga, gz, p.wp31, &e, ginp, bn_z, bn_coef, bn_coef + C,
                                ws_partial(b), bn_fin, st);
    }
  }
  memset(&e, 0, sizeof(e));
  e.res = res_in;
  e.res_gate = res_gate;
  if (tail != nullptr && tail->partial != nullptr) {
    // block-boundary fusion: gate ginp by this block's input and emit the reductions of the
    // previous block's outer BatchNorm backward
    mdil_bn_tail t = *tail;
    t.gate = inp;
    return mdil_tapconv_tail(&G31t, C, C, ga, gz, p.wp31, &e, ginp, &t, st);
  }
  return mdil_tapconv(&G31t, C, C, ga, gz, p.wp31, &e, ginp, st);
}

}  // namespace

static int nb_block_backward_impl(const mdil_nb_block* b, Defer* d, void* st) {
  TRY(check_common(b));
  MDIL_CHECK_ARG(b->gy && b->a1 && b->z1 && b->u && b->a2 && b->z2 && b->out,
                 "nb_block_backward: saved tensors missing");
  MDIL_CHECK_ARG(b->gz2 && b->ga && b->gu && b->gx, "nb_block_backward: scratch / result missing");
  const int C = b->C;
  const long long npix = (long long)b->N * b->H * b->W;
  const int ppi = b->H * b->W;
  MDIL_CHECK_ARG(b->bn_workspace != nullptr && b->bn_workspace_bytes >= mdil_bn_workspace(npix, C),
                 "nb_block_backward: BatchNorm workspace too small");
  MDIL_CHECK_ARG(d != nullptr || b->wgrad_workspace_bytes >= mdil_nb_block_wgrad_workspace(
                                                                  b->N, b->H, b->W, C, b->dilation, b->rap),
                 "nb_block_backward: weight-gradient workspace too small");
  const mdil_nb_half &p1 = b->half[0], &p2 = b->half[1];
  // second half:  out = relu(bn2(z2) * drop + x)
  const bool head_fin = b->head_coef != nullptr;
  const bool head = head_fin || (b->head_partial != nullptr && b->head_nblk > 0);
  if (head_fin) {
    // gy is already gated by out > 0; the next block's tail launch has finalized the reductions
    // (and added this block's dgamma / dbeta): the apply pass is all that is left
    TRY(mdil_bn_backward_apply(b->gy, b->drop, b->z2, npix, ppi, C, p2.coef, p2.coef + C, b->head_coef,
                               b->gz2, st));
  } else if (head) {
    // gy is already gated by out > 0 and the reductions came with it (the next block's tail launch)
    TRY(mdil_bn_backward_partials(b->gy, b->drop, b->z2, npix, ppi, C, p2.gamma, p2.coef, p2.coef + C,
                                  b->head_partial, b->head_nblk, p2.dgamma, p2.dbeta, 1, b->gz2,
                                  ws_finalize(b), (size_t)3 * C * sizeof(float), st));
  } else {
    TRY(mdil_bn_backward(b->gy, b->out, b->drop, b->z2, npix, ppi, C, p2.gamma, p2.coef, p2.coef + C,
                         p2.dgamma, p2.dbeta, 1, b->gz2, b->bn_workspace, b->bn_workspace_bytes,
                         b->ticket, st));
  }
  int fused = 0;
  // the inner BatchNorm's reductions ride in the dgrad that produces gu; with a ticket that launch
  // finalizes them too (dgamma / dbeta of bn1, the [3][C] table behind the partial rows)
  mdil_bn_grad fin1;
  fin1.gamma = p1.gamma, fin1.dgamma = p1.dgamma, fin1.dbeta = p1.dbeta, fin1.accumulate = 1;
  fin1.coef = ws_finalize(b), fin1.ticket = b->ticket;
  TRY(half_backward(b, p2, b->dilation, b->gz2, b->a2, b->u, b->ga, b->gu, nullptr, nullptr, b->u,
                    b->z1, p1.coef, &fused, d, st, nullptr, b->ticket ? &fin1 : nullptr));
  // first half:  u = relu(bn1(z1)); gz1 overwrites gu
  if (fused > 0 && b->ticket) {
    TRY(mdil_bn_backward_apply(b->gu, nullptr, b->z1, npix, ppi, C, p1.coef, p1.coef + C, ws_finalize(b),
                               b->gu, st));
  } else if (fused > 0) {
    TRY(mdil_bn_backward_partials(b->gu, nullptr, b->z1, npix, ppi, C, p1.gamma, p1.coef, p1.coef + C,
                                  ws_partial(b), fused, p1.dgamma, p1.dbeta, 1, b->gu, ws_finalize(b),
                                  (size_t)3 * C * sizeof(float), st));
  } else {
    TRY(mdil_bn_backward(b->gu, b->u, nullptr, b->z1, npix, ppi, C, p1.gamma, p1.coef, p1.coef + C,
                         p1.dgamma, p1.dbeta, 1, b->gu, b->bn_workspace, b->bn_workspace_bytes, b->ticket,
                         st));
  }
  // the block input also receives gy * (out > 0) through the residual connection
  // (a head-gated gy needs no gate here)
  if (b->tail.partial != nullptr) {
    MDIL_CHECK_ARG(b->tail.z && b->tail.save_mean && b->tail.save_invstd, "nb_block_backward: tail fields");
    MDIL_CHECK_ARG(mdil_nb_block_tail_blocks(b->N, b->H, b->W, C, b->rap) > 0,
                   "nb_block_backward: tail requested for a shape the tail launch does not cover");
  }
  return half_backward(b, p1, 1, b->gu, b->a1, b->x, b->ga, b->gx, b->gy, head ? nullptr : b->out, nullptr,
                       nullptr, nullptr, &fused, d, st, &b->tail);
}

extern "C" int mdil_nb_block_tail_blocks(int N, int H, int W, int C, int rap) {
  if (C != 64 && C != 128) return 0;
  mdil_nb_block b;
  memset(&b, 0, sizeof(b));
  b.N = N, b.H = H, b.W = W, b.C = C, b.rap = rap;
  const mdil_geom G31t = geom(&b, 1, false, true, rap != 0);
  return mdil_tapconv_tail_blocks(&G31t, C, C);
}

extern "C" int mdil_nb_block_backward(const mdil_nb_block* b, void* st) {
  return nb_block_backward_impl(b, nullptr, st);
}

extern "C" int mdil_nb_block_backward_deferred(const mdil_nb_block* b, mdil_wgrad_job* jobs, int* njobs,
                                               size_t* workspace_used, void* st) {
  MDIL_CHECK_ARG(jobs && njobs && workspace_used, "nb_block_backward_deferred: null argument");
  Defer d{jobs, 0, 0};
  const int rc = nb_block_backward_impl(b, &d, st);
  *njobs = d.njobs;
  *workspace_used = d.cursor;
  return rc;
}
