"""GPU: mIoU parity of a two-stage training run (north_star: "matching the reference's mIoU
within +-0.1 on identical inputs/seeds").  The golden (tests/golden/miou_run.npz) is
the imported REFERENCE model trained on CPU by tools/gen_miou_golden.py: step 1 on the first
domain (train_RAPFT_step1.py semantics, 7,680 iterations), then step 2 on the second domain with
KD from the step-1 model (train_new_task_step2.py, 4,096 iterations) -- the reference's batch
size, optimizer, LR schedules and loss, on a seeded procedural dataset that is learnable (all 19
evaluated classes present, well separated colours: the new-domain head reaches 85 % mIoU) and a
validation set of 512 images per domain.  This test repeats the identical protocol
(tests/miou_protocol.py: same init, batches, dropout masks) on the HIP path -- Step1Engine then
Step2Engine (3-stream schedule) -- and compares the final mIoU of both validation sets
(iouEval.py:72-77) and the loss curves.

A training run of this length is a chaotic function of fp32 rounding: the reference does not
reproduce ITSELF to 0.1 point (another oneDNN thread count or initial weights perturbed by 1e-7
relative move the new-domain mIoU by up to a point).  What "within +-0.1" can mean,
and what is asserted, is therefore a statement about the MEANS of the two implementations'
run-to-run distributions, resolved with enough samples:

  * the golden holds every independent reference run made in the build container
    (``ref_miou_new``: tools/miou_ref_sample.py, initial weights x (1 + 1e-7 N(0,1)), one seed per
    run) and every HIP run recorded on an MI355X with the same kind of perturbation
    (``hip_miou_new``: tools/miou_hip_sample.py); this test adds its own live runs to the HIP
    sample, computes both standard deviations and the standard error of the difference of the
    means FROM THE DATA (nothing hard-coded) and asserts |mean_hip - mean_ref| <= 0.1 point with
    a standard error <= 0.1 point;
  * only recorded HIP runs of the BUILD UNDER TEST count (every sample carries the id of the
    device sources that produced it, tests/helpers.kernel_build_id; at least 32 are required);
    each live run must also be a plausible member of that distribution (within 4 sigma, sigma = the
    larger of the two implementations' estimates);
  * the metric path itself is exact: the HIP eval forward + fused argmax/confusion kernel and the
    oracle's eval forward + iouEval restatement give the same mIoU (< 0.02 point) on the same
    trained weights;
  * ONE-STEP PARITY FROM TRAINED STATES: at stage-A iteration 7,679 (the last step-1 iteration)
    and at stage-B iterations 0, 1024 and 4095 the training iteration itself is run gate-forced
    on the HIP path (the shipped schedule: block-level C ABI, deferred weight-gradient
    reductions, three streams) and replayed by the oracle FROM THE SAME STATE (weights, BN
    buffers, Adam moments, step counts, learning rates): every gradient element-wise at 1e-3,
    every BN buffer, the post-Adam parameters.  Random-init checks say nothing about dead ReLUs,
    large running statistics and saturated softmaxes; this is the regime a run lives in;
  * THE SAME AT A COVERING SIZE: the protocol's 32x64 images leave a 4x8 map in front of the deepest
    blocks, where the dilations 8 and 16 (and 4 along H) cannot form complete Winograd pairs and the
    direct-form kernels run instead.  The trained states of the end of stage A and of stage-B
    iterations 1024 and 4095 are therefore ALSO stepped once on an N=2, 256x512 batch (deepest map
    32x64: every dilation pairs up on both axes) -- same comparison against the oracle from the same
    state, plus the in-library launch profile asserting that no 3-tap conv left the Winograd path;
  * MULTI-STEP PARITY AT THE COVERING SIZE (round 6): from the same three trained states, K = 8 and 16
    FREE-GATE training steps on N=2, 256x512 batches on the HIP path and on the oracle (identical batches
    and masks, each side's own ReLU gates and Adam), then both states scored on 8 held-out covering
    batches: logits rel-L2, argmax agreement, confusion-matrix mIoU of both heads pairwise <= 0.1 point;
    the launch profile of every step must show w4conv carrying more than half of the 3-tap conv launches
    (tests/covering_trajectory.py);
  * PAIRED TRAJECTORIES (round 5): from the trained student states at stage-B iterations 0, 1024 and 4064 the
    HIP run and the oracle both make the next 32 training steps -- identical batches and dropout masks, each
    implementation's OWN ReLU gates, the oracle's own Adam from the same moments -- and after 8 / 16 / 32 steps
    both states are scored on BOTH validation sets: |d mIoU| <= 0.1 point on each head, pairwise.  This is the
    +-0.1 statement for the old-domain head, whose run-to-run sigma of 2 points no number of independent runs
    resolves (measured: <= 0.012 point on the new head, <= 0.0004 on the old one);
  * first-iteration loss to 1e-5, loss curves within twice the run-to-run drift.
"""
import os

import numpy as np
import pytest
import torch

from oracle import fixtures as fx
from oracle import rap_oracle as O
from tests import covering_trajectory as CT
from tests import miou_protocol as MP
from tests import helpers as Hh

pytestmark = pytest.mark.gpu

CHECK_AT_B = (0, 1024, 4095)          # stage-B iterations checked one step ahead against the oracle
COVER_AT_B = (1024, 4095)             # ... and, from the same states, at the covering size (256x512)
# PAIRED trajectories: from the trained states at these stage-B iterations the HIP run and
# the oracle both make the next 32 training steps on identical batches / masks, each with its OWN ReLU
# gates, and both heads are scored on the validation sets after 8 / 16 / 32 steps
PAIR_AT_B = (0, 1024, 4064)
PAIR_K = (8, 16, 32)


def _smooth(x, k=200):
    return np.convolve(x, np.ones(k) / k, mode="valid")


def _cpu(sd):
    return {k: v.detach().cpu().clone() for k, v in sd.items()}


def _adam_snapshot(opt):
    """-> per-parameter (exp_avg, exp_avg_sq) CPU clones in group order + (step, lr) per group."""
    m, v = opt.exp_avg.detach().cpu().clone(), opt.exp_avg_sq.detach().cpu().clone()
    return m, v, [(g["step"], g["lr"], g["offset"]) for g in opt.param_groups]


def _compare(a, b, rtol, atol_rel, what):
    """element-wise |a-b| <= rtol*|b| + atol_rel*max|b|"""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    tol = rtol * b.abs() + atol_rel * float(b.abs().max()) + 1e-30
    bad = (a - b).abs() > tol
    assert not bool(bad.any()), (what, int(bad.sum()), float((a - b).abs().max()), float(b.abs().max()))
    return float((a - b).norm() / (b.norm() + 1e-30))


def _one_step_check(tag, where, names, trainable, pre_sd, teacher_sd, adam, opt, model, images, labels,
                    masks_new, masks_old, gates_new, gates_old, weight, task, lambdac, ce_hip, kld_hip,
                    grad_atol_rel=1e-4):
    """The iteration the HIP path has just made, replayed by the oracle from the same state."""
    S = {k: v.clone() for k, v in pre_sd.items()}
    for n in names:
        S[n].requires_grad_(trainable(n))
    if teacher_sd is None:                       # step 1: one train-mode forward + CE
        out = O.net_forward(S, images, task, True, masks_new, gates=[g.cpu() for g in gates_new])
        ce_o = O.ce2d(out, labels[:, 0], weight)
        ce_o.backward()
        kld_o = None
    else:
        ce_o, kld_o, *_ = O.step2_iteration(S, {k: v.clone() for k, v in teacher_sd.items()}, images,
                                            labels, weight, task, lambdac, masks_new, masks_old,
                                            [g.cpu() for g in gates_new], [g.cpu() for g in gates_old])
    np.testing.assert_allclose(float(ce_hip), float(ce_o), rtol=5e-5, err_msg=f"{where}: CE")
    if kld_o is not None:
        np.testing.assert_allclose(float(kld_hip), float(kld_o), rtol=2e-4, atol=1e-7, err_msg=f"{where}: KLD")
    params = dict(model.named_parameters())
    worst, n_grad = ("", 0.0), 0
    for n in names:
        gd, gc = params[n].grad if trainable(n) else None, S[n].grad
        assert (gd is None) == (gc is None), f"{where}: gradient presence of {n}"
        if gc is None:
            continue
        n_grad += 1
        if Hh.zero_grad_bias(n):
            scale = float(S[n.replace(".bias", ".weight")].grad.abs().max())
            assert float(gd.abs().max()) <= 1e-3 * scale + 1e-7, (where, n)
            continue
        rel = _compare(gd, gc, 1e-3, grad_atol_rel, f"{where}: grad {n}")
        if rel > worst[1]:
            worst = (n, rel)
    # BN buffers after the iteration (the KD forward moves the frozen domain's running statistics too)
    post = _cpu(model.state_dict())
    for k in post:
        if O.is_buffer(k) and post[k].is_floating_point():
            _compare(post[k], S[k].detach(), 2e-4, 2e-5, f"{where}: buffer {k}")
        elif O.is_buffer(k):
            assert int(post[k]) == int(S[k]), (where, k)
    # Adam: the update each parameter received, from the same moments / step counts / learning rates
    m_all, v_all, groups = adam
    worst_u = ("", 0.0)
    for gi, g in enumerate(opt.param_groups):
        step, lr, off = groups[gi]
        for p, n in zip(g["params"], g["names"]):
            k = p.numel()
            pp = pre_sd[n].clone()
            m, v = m_all[off:off + k].view(pp.shape).clone(), v_all[off:off + k].view(pp.shape).clone()
            O.adam_l2_step(pp, S[n].grad, m, v, step + 1, lr)
            upd_o, upd_h = (pp - pre_sd[n]).double(), (post[n] - pre_sd[n]).double()
            if step == 0:
                # first Adam step: update = lr * gt / (|gt| + eps'), gt = g + weight_decay * p -- where
                # the two terms cancel to noise level the sign of gt is not determined; compare the rest
                gt = S[n].grad + 1e-4 * pre_sd[n]
                sel = gt.abs() > 1e-2 * gt.abs().max()
                upd_o, upd_h = upd_o[sel], upd_h[sel]
            if upd_o.numel() and not Hh.zero_grad_bias(n):
                # parameters are ~1e-1, updates ~1e-5..1e-7: the fp32 subtraction p - step leaves
                # |p| * 2^-24 of rounding in either implementation
                rt = 1e-2 if step == 0 else 2e-3       # (first step: d update / d gt is eps / (|gt| + eps)^2)
                tol = rt * upd_o.abs() + 2e-4 * float(upd_o.abs().max()) + 1.2e-7 * pre_sd[n].abs().max().double()
                bad = (upd_h - upd_o).abs() > tol
                assert not bool(bad.any()), (where, "Adam update", n, int(bad.sum()),
                                             float((upd_h - upd_o).abs().max()), float(upd_o.abs().max()))
                rel = float((upd_h - upd_o).norm() / (upd_o.norm() + 1e-30))
                if rel > worst_u[1]:
                    worst_u = (n, rel)
            off += k
    print(f"[{tag}] one-step parity from the trained state at {where}: {n_grad} gradients, worst rel-L2 "
          f"{worst[1]:.2e} ({worst[0]}); Adam updates worst rel-L2 {worst_u[1]:.2e} ({worst_u[0]})", flush=True)
    return worst[1]


def _covering_step_check(dev, tag, where, pre_sd, teacher_sd, adam, seed):
    """ONE gate-forced training iteration from the trained state (``pre_sd``, Adam moments / step counts
    / learning rates ``adam``; step 2 when ``teacher_sd`` is given, else step 1) on an N=2, 256x512
    batch, HIP (three-stream schedule for step 2) against the oracle: every gradient, BN buffer and
    Adam update as in ``_one_step_check`` -- and the in-library launch profile must show that EVERY
    3-tap conv / dgrad of the C = 64 / 128 blocks took the Winograd kernel the full-size step ships
    (no launch on the direct ``sconv`` / LDS-tiled ``tapconv`` path) and that the Winograd weight-
    gradient kernels ran: at the protocol's own 32x64 size the dilations 8 and 16 cannot pair up and
    fall back to the direct forms (tests/miou_protocol.py)."""
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import ops
    from mdil_ss_amd import train_new_task_step2 as T
    from mdil_ss_amd.engine import Step1Engine, Step2Engine
    from mdil_ss_amd.models.erfnet_RA_parallel import Net
    ops.invalidate_packs()
    cfg = MP.CONFIG
    weight_cpu = torch.tensor(fx.WEIGHT_BDD)
    weight = weight_cpu.to(dev)
    images, labels = MP.covering_batch(seed, old_domain=teacher_sd is None)
    N = images.shape[0]
    m_new, m_old = MP.masks_for(900000 + seed, N)
    if teacher_sd is None:
        model = Net([20], 1, 0)
        model.load_state_dict(pre_sd)
        model.to(dev)
        eng = Step1Engine(model, weight, current_task=0)
        names = [n for n, _ in model.named_parameters()]
        eng.optimizer.param_groups[0]["names"] = names
        trainable = lambda n: True
        task, lam = 0, 0.0
    else:
        model = Net([20, 20], 2, 1)
        model.load_state_dict(pre_sd)
        model.to(dev)
        frozen = Net([20], 1, 0)
        frozen.load_state_dict(teacher_sd)
        frozen.to(dev)
        ops.invalidate_packs()
        T.current_task = 1
        T.apply_step2_freeze(model, frozen, 1)
        eng = Step2Engine(model, frozen, weight, current_task=1, lambdac=cfg["lambdac"],
                          is_shared=T.is_shared, is_ds_curr=T.is_DS_curr)
        names = [n for n, _ in model.named_parameters()]
        by_id = {id(p): n for n, p in model.named_parameters()}
        for g in eng.optimizer.param_groups:
            g["names"] = [by_id[id(p)] for p in g["params"]]
        trainable = lambda n: O.step2_trainable("module." + n, 1)
        task, lam = 1, cfg["lambdac"]
    m_all, v_all, groups = adam

    def restore():
        model.load_state_dict(pre_sd)                   # in place: weights and BN buffers
        ops._stale_refresh()                            # packed weight images follow, on this stream
        eng.optimizer.exp_avg.copy_(m_all.to(dev))
        eng.optimizer.exp_avg_sq.copy_(v_all.to(dev))
        for g, (step, lr, off) in zip(eng.optimizer.param_groups, groups):
            assert g["offset"] == off
            g["step"], g["lr"] = step, lr
        torch.cuda.synchronize()

    x, y = images.to(dev), labels.to(dev)
    restore()
    if teacher_sd is not None:
        # the engine's first iteration runs on one stream (it creates the packed images and scratch
        # every stream reads afterwards): spend it, put the state back, check the three-stream one
        q = [m_new, m_old]
        model.mask_provider = lambda n: q.pop(0)
        eng.iteration(x, y)
        restore()
    q = [m_new, m_old] if teacher_sd is not None else [m_new]
    model.mask_provider = lambda n: q.pop(0)
    multi = teacher_sd is not None
    ops.GATE_LOG = {0: [], 1: []} if multi else []
    ops.profile_begin()
    out = eng.iteration(x, y)
    prof = ops.profile_end()
    log, ops.GATE_LOG = ops.GATE_LOG, None
    if teacher_sd is not None:
        assert getattr(eng, "multi_stream", False)
        _, ce, kld = out
        g_new, g_old = log[0], log[1]
        assert len(g_new) == 73 and len(g_old) == 73, (len(g_new), len(g_old))
    else:
        ce, kld, g_new, g_old = out, None, log, None
    # ---- which kernels ran
    conv = [(k, cin, nt) for k, cin, cout, nt, _, _ in prof if k in ops._PROF_CONV and cin == cout and cin in (64, 128)
            and nt in (3, 4)]
    wg = [(k, cin, nt) for k, cin, cout, nt, _, _ in prof if k in ops._PROF_WGRAD and cin == cout and cin in (64, 128)
          and nt in (3, 4)]         # (the 9-tap 64 -> 64 conv of the second down-sampler is not a block conv)
    by_path = {}
    for k, cin, nt in conv:
        by_path[k] = by_path.get(k, 0) + 1
    n_fwd = 3 if teacher_sd is not None else 1          # forwards; backward graphs: n_fwd - 1 or 1
    n_bwd = 2 if teacher_sd is not None else 1
    # 15 factorised C = 64 / 128 blocks x 4 convs per forward, x 4 dgrads per backward graph: all
    # Winograd; nothing on the direct streaming kernel; the only other 4-tap C -> C launch is one
    # parity class of the stride-2 dgrad of the 64 -> 128 down-sampler (LDS-tiled, once per graph)
    assert by_path.get("sconv", 0) == 0, f"{where}: 3-tap convs on the direct kernel: {by_path}"
    # (F(4,3) where the axis is a multiple of 4 x dilation, F(2,3) for the rest: d = 16 on 32 rows)
    assert by_path.get("wconv", 0) + by_path.get("w4conv", 0) == 15 * 4 * (n_fwd + n_bwd), (where, by_path)
    assert by_path.get("w4conv", 0) > by_path.get("wconv", 0) > 0, (where, by_path)
    assert by_path.get("tapconv", 0) <= n_bwd, (where, by_path)
    wpath = {}
    for k, cin, nt in wg:
        wpath[k] = wpath.get(k, 0) + 1
    assert wpath.get("wgradw", 0) > 0 and wpath.get("wgrad", 0) == 0, (where, wpath)
    print(f"[{tag}] covering-size launches at {where}: convs {by_path}, weight gradients {wpath}", flush=True)
    return _one_step_check(tag, where + f" @ N={N} {MP.COVER['height']}x{MP.COVER['width']}", names, trainable,
                           pre_sd, teacher_sd, adam, eng.optimizer, model, images, labels, m_new,
                           m_old if teacher_sd is not None else None, g_new, g_old, weight_cpu, task, lam, ce, kld,
                           # every weight-gradient element sums 64x the pixels of the protocol's own
                           # size, in fp32 on BOTH sides (the oracle is oneDNN fp32): the noise floor of
                           # elements near zero rises with it (measured: 1.9e-4 of the tensor's largest
                           # element on the stem conv, 1 element of 351; per-tensor rel-L2 stays <= 3e-4)
                           grad_atol_rel=5e-4)


def _score_both_heads(dev, sd):
    """mIoU (both validation sets) of a step-2 student state through the HIP eval path."""
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import ops
    from mdil_ss_amd.iouEval import iouEval
    from mdil_ss_amd.models.erfnet_RA_parallel import Net
    ops.invalidate_packs()
    model = Net([20, 20], 2, 1)
    model.load_state_dict(sd)
    model.to(dev).eval()
    res = {}
    with torch.no_grad():
        for task, name in ((1, "new"), (0, "old")):
            ev = iouEval(20, 19)
            for images, labels in MP.val_batches(task):
                ev.addBatch(model(images.to(dev), task), labels.to(dev))
            res[name] = float(ev.getIoU()[0])
    return res


def _paired_trajectory(dev, tag, win, teacher_sd):
    """The oracle makes the same max(PAIR_K) training steps as the HIP run did from ``win['pre']`` (weights,
    BN buffers, Adam moments / step counts / learning rates), on the recorded batches and dropout masks,
    with its OWN ReLU gates and its own Adam restatement; after each K of PAIR_K both implementations'
    states are scored on both validation sets through the same (HIP) eval path.
    -> {K: (d_new, d_old)} in mIoU POINTS (HIP - oracle)."""
    cfg = MP.CONFIG
    weight_cpu = torch.tensor(fx.WEIGHT_BDD)
    trainable = lambda n: O.step2_trainable("module." + n, 1)
    S = {k: v.clone() for k, v in win["pre"].items()}
    for n in S:
        if S[n].is_floating_point() and not O.is_buffer(n):
            S[n].requires_grad_(trainable(n))
    T_sd = {k: v.clone() for k, v in teacher_sd.items()}
    m_all, v_all, groups = win["adam"]
    mom = {}
    for gi, names in enumerate(win["groups"]):
        step, lr, off = groups[gi]
        for n in names:
            k = S[n].numel()
            mom[n] = (m_all[off:off + k].view(S[n].shape).clone(), v_all[off:off + k].view(S[n].shape).clone(), gi)
            off += k
    out = {}
    for j, (images, labels, m_new, m_old) in enumerate(win["batches"]):
        for n in mom:
            S[n].grad = None
        O.step2_iteration(S, T_sd, images, labels, weight_cpu, 1, cfg["lambdac"], m_new, m_old)
        with torch.no_grad():
            for n, (m, v, gi) in mom.items():
                if S[n].grad is not None:
                    step, lr, _ = groups[gi]
                    O.adam_l2_step(S[n], S[n].grad, m, v, step + j + 1, lr)
        k = j + 1
        if k in PAIR_K:
            o = _score_both_heads(dev, {n: t.detach().clone() for n, t in S.items()})
            h = _score_both_heads(dev, win["hip"][k])
            out[k] = ((h["new"] - o["new"]) * 100.0, (h["old"] - o["old"]) * 100.0, h, o)
    line = "; ".join(f"K={k}: new {out[k][0]:+.4f} old {out[k][1]:+.4f}" for k in PAIR_K)
    hk, ok = out[max(PAIR_K)][2], out[max(PAIR_K)][3]
    print(f"[{tag}] paired trajectories from stage-B iteration {win['it0']}: HIP - oracle mIoU (points) {line}  "
          f"(after {max(PAIR_K)} steps: HIP new {hk['new'] * 100:.3f} old {hk['old'] * 100:.3f}, "
          f"oracle new {ok['new'] * 100:.3f} old {ok['old'] * 100:.3f})", flush=True)
    return out


def _run_protocol(dev, tag, perturb_seed=None, checks=False, oracle_eval=True):
    """One full two-stage run on the HIP path -> dict(lossesA, losses, miou_new, miou_old).
    ``perturb_seed``: initial weights x (1 + 1e-7 N(0,1)) exactly like tools/gen_miou_golden.py
    --perturb (an independent sample of the run-to-run noise).  ``checks``: one-step parity from
    the trained states (module docstring).  ``oracle_eval``: also score the trained weights through
    the oracle's eval forward + iouEval restatement (CPU: the slow part; sampling runs skip it)."""
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import ops
    from mdil_ss_amd import train_new_task_step2 as T
    from mdil_ss_amd.engine import Step1Engine, Step2Engine
    from mdil_ss_amd.iouEval import iouEval
    from mdil_ss_amd.models.erfnet_RA_parallel import Net
    ops.invalidate_packs()
    cfg = MP.CONFIG
    weight_cpu = torch.tensor(fx.WEIGHT_BDD)
    weight = weight_cpu.to(dev)
    # ---- stage A: step 1 on the first domain -> the teacher
    teacher = Net([20], 1, 0)
    init = MP.step1_initial_state()
    if perturb_seed is not None:
        gp = torch.Generator().manual_seed(int(perturb_seed))
        for k, v in init.items():
            if v.is_floating_point():
                v.mul_(1.0 + 1e-7 * torch.randn(v.shape, generator=gp))
    teacher.load_state_dict(init)
    teacher.to(dev)
    engA = Step1Engine(teacher, weight, current_task=0)
    namesA = [n for n, _ in teacher.named_parameters()]
    engA.optimizer.param_groups[0]["names"] = namesA
    n_itA = cfg["epochs_step1"] * (cfg["n_train"] // cfg["batch"])
    lossesA, it, worst, cover_states = [], 0, [], []
    for epoch in range(1, cfg["epochs_step1"] + 1):
        engA.optimizer.set_epoch(epoch, cfg["epochs_step1"])
        for images, labels in MP.train_batches(epoch, old_domain=True):
            masks = MP.masks_for(it, images.shape[0])[0]
            q = [masks]
            teacher.mask_provider = lambda n: q.pop(0)
            chk = checks and it == n_itA - 1
            if chk:
                pre, adam = _cpu(teacher.state_dict()), _adam_snapshot(engA.optimizer)
                ops.GATE_LOG = []
            lossesA.append(engA.iteration(images.to(dev), labels.to(dev)))     # device scalar: no sync
            if chk:
                gates, ops.GATE_LOG = ops.GATE_LOG, None
                worst.append(_one_step_check(tag, f"stage A iteration {it}", namesA, lambda n: True, pre, None,
                                             adam, engA.optimizer, teacher, images, labels, masks, None,
                                             gates, None, weight_cpu, 0, 0.0, lossesA[-1], None))
                cover_states.append((f"stage A iteration {it}", pre, None, adam))
            it += 1
    lossesA = torch.stack(lossesA).double().cpu().numpy()
    # ---- stage B: step 2 with KD from the step-1 model
    teacher.eval()
    teacher.mask_provider = None
    teacher_sd = _cpu(teacher.state_dict())
    student = Net([20, 20], 2, 1)
    student.load_state_dict(MP.step2_student_state(teacher_sd))
    student.to(dev)
    frozen = Net([20], 1, 0)            # a fresh module for the frozen teacher (own parameter storage)
    frozen.load_state_dict(teacher_sd)
    frozen.to(dev)
    ops.invalidate_packs()
    T.current_task = 1
    T.apply_step2_freeze(student, frozen, 1)
    eng = Step2Engine(student, frozen, weight, current_task=1, lambdac=cfg["lambdac"],
                      is_shared=T.is_shared, is_ds_curr=T.is_DS_curr)
    names = [n for n, _ in student.named_parameters()]
    by_id = {id(p): n for n, p in student.named_parameters()}
    for g in eng.optimizer.param_groups:
        g["names"] = [by_id[id(p)] for p in g["params"]]
    trainable = lambda n: O.step2_trainable("module." + n, 1)
    losses, it, win, windows = [], 0, None, []
    for epoch in range(1, cfg["epochs"] + 1):
        eng.optimizer.set_epoch(epoch, cfg["epochs"])
        for images, labels in MP.train_batches(epoch):
            m_new, m_old = MP.masks_for(100000 + it, images.shape[0])
            q = [m_new, m_old]
            student.mask_provider = lambda n: q.pop(0)
            chk = checks and it in CHECK_AT_B
            if checks and it in PAIR_AT_B:
                win = {"it0": it, "pre": _cpu(student.state_dict()), "adam": _adam_snapshot(eng.optimizer),
                       "groups": [list(g["names"]) for g in eng.optimizer.param_groups], "batches": [], "hip": {}}
            if win is not None:
                win["batches"].append((images, labels, m_new, m_old))
            if chk:
                pre, adam = _cpu(student.state_dict()), _adam_snapshot(eng.optimizer)
                ops.GATE_LOG = {0: [], 1: []}       # slot 0 = new-task graph, 1 = old-task graph
                multi = getattr(eng, "multi_stream", False)
                if not multi:
                    ops.GATE_LOG = []
            total, ce, kld = eng.iteration(images.to(dev), labels.to(dev))
            if chk:
                log, ops.GATE_LOG = ops.GATE_LOG, None
                g_new, g_old = (log[0], log[1]) if isinstance(log, dict) else (log[:len(log) // 2], log[len(log) // 2:])
                assert len(g_new) == 73 and len(g_old) == 73, (len(g_new), len(g_old))
                worst.append(_one_step_check(tag, f"stage B iteration {it} ({'3 streams' if multi else '1 stream'})",
                                             names, trainable, pre, teacher_sd, adam, eng.optimizer, student,
                                             images, labels, m_new, m_old, g_new, g_old, weight_cpu, 1,
                                             cfg["lambdac"], ce, kld))
                if it in COVER_AT_B:
                    cover_states.append((f"stage B iteration {it}", pre, teacher_sd, adam))
            losses.append(torch.stack([ce, kld]))
            if win is not None:
                k = it - win["it0"] + 1
                if k in PAIR_K:
                    win["hip"][k] = _cpu(student.state_dict())
                if k == max(PAIR_K):
                    windows.append(win)
                    win = None
            it += 1
    losses = torch.stack(losses).double().cpu().numpy()
    out = {"lossesA": lossesA, "losses": losses, "one_step_worst": worst, "cover_states": cover_states,
           "pair_windows": windows, "teacher_sd": teacher_sd}
    student.eval()
    S = _cpu(student.state_dict())
    for task, name in ((1, "new"), (0, "old")):
        ev = iouEval(20, 19)
        tp = torch.zeros(19, dtype=torch.float64)
        fp_, fn = torch.zeros(19, dtype=torch.float64), torch.zeros(19, dtype=torch.float64)
        with torch.no_grad():
            for images, labels in MP.val_batches(task):
                ev.addBatch(student(images.to(dev), task), labels.to(dev))
                if oracle_eval:
                    # the same trained weights through the oracle's eval forward + iouEval restatement
                    a, b, c = O.iou_counts(O.net_forward(S, images, task, False).max(1)[1], labels[:, 0], 20, 19)
                    tp += a
                    fp_ += b
                    fn += c
        m = float(ev.getIoU()[0])
        if oracle_eval:
            m_oracle = float(O.miou(tp, fp_, fn)[0])
            print(f"[{tag}] mIoU {name}: HIP eval path {m * 100:.4f} vs oracle eval of the same weights "
                  f"{m_oracle * 100:.4f}")
            # a difference beyond a few boundary pixels would be a bias of the METRIC path
            assert abs(m - m_oracle) < 2e-4, (name, m, m_oracle)
        out["miou_" + name] = m
    return out


def _stats(x):
    x = np.asarray(x, dtype=np.float64)
    return float(x.mean()), float(x.std(ddof=1)), len(x)


def mean_difference(hip, ref):
    """-> (delta, standard error of delta) in mIoU POINTS, both from the data (Welch)."""
    mh, sh, nh = _stats(hip)
    mr, sr, nr = _stats(ref)
    return (mh - mr) * 100.0, 100.0 * (sh * sh / nh + sr * sr / nr) ** 0.5


def test_training_run_matches_reference_miou():
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "miou_run.npz"))
    dev = torch.device("cuda:0")
    # live runs: the unperturbed protocol (with the one-step checks from trained states) and one
    # perturbed sample (a seed the recorded HIP sample does not hold)
    torch.set_num_threads(Hh.host_threads(16))     # the oracle legs run on the host cores
    runs = [_run_protocol(dev, "hip", checks=True), _run_protocol(dev, "hip, seed 9001", perturb_seed=9001)]
    assert len(runs[0]["one_step_worst"]) == 1 + len(CHECK_AT_B)
    # ---- the same trained states, one step at the size where every dilation takes the shipped
    # Winograd kernels (end of stage A; stage B iterations 1024 and 4095)
    cover = runs[0].pop("cover_states")
    assert len(cover) == 1 + len(COVER_AT_B)
    for i, (where, pre, t_sd, adam) in enumerate(cover):
        _covering_step_check(dev, "hip", where, pre, t_sd, adam, seed=31 + i)
    # ---- ... and 16 FREE-GATE training steps from each of them at that size, HIP against the oracle, both
    # scored on 8 held-out covering batches after 8 and 16 steps (tests/covering_trajectory.py): multi-step
    # parity on the F(4,3) kernels the bench runs, which the 32x64 protocol itself barely launches
    for i, (where, pre, t_sd, adam) in enumerate(cover):
        CT.covering_trajectory(dev, "hip", where, pre, t_sd, adam, seed=61 + i)
    runs[1].pop("cover_states")
    # ---- paired trajectories: the +-0.1 statement for BOTH heads, pairwise (the old-domain head's
    # run-to-run sigma of 2 points does not enter: both implementations start from the same state and
    # see the same batches; 32 steps in, the one-step difference of ~1e-5 has not decorrelated them)
    windows, t_sd = runs[0].pop("pair_windows"), runs[0].pop("teacher_sd")
    runs[1].pop("pair_windows"), runs[1].pop("teacher_sd")
    assert [w["it0"] for w in windows] == list(PAIR_AT_B)
    for w in windows:
        res = _paired_trajectory(dev, "hip", w, t_sd)
        for k in PAIR_K:
            d_new, d_old = res[k][0], res[k][1]
            assert abs(d_new) <= 0.1 and abs(d_old) <= 0.1, (w["it0"], k, d_new, d_old)
    # ---- loss curves of the first run against the golden run
    r = runs[0]
    refA, altA = G["losses_step1"], G["alt_losses_step1"]
    # only the first iteration is a deterministic function of the inputs; Adam at lr 5e-4 on every
    # parameter (sign-like first steps) makes the second one already differ at the 1e-4 level
    np.testing.assert_allclose(r["lossesA"][0], refA[0], rtol=1e-5)
    np.testing.assert_allclose(r["lossesA"][:5], refA[:5], rtol=1e-2)
    driftA = np.abs(_smooth(altA) - _smooth(refA)).max()
    errA = np.abs(_smooth(r["lossesA"]) - _smooth(refA)).max()
    driftA_hip = np.abs(_smooth(runs[1]["lossesA"]) - _smooth(r["lossesA"])).max()
    print(f"step-1 CE curve: max smoothed |hip-ref| {errA:.4f}, reference-vs-reference drift {driftA:.4f}, "
          f"hip-vs-hip drift {driftA_hip:.4f}")
    assert errA <= 2 * max(driftA, driftA_hip) + 0.02 * _smooth(refA).mean(), (errA, driftA, driftA_hip)
    ref, alt = G["losses"], G["alt_losses"]
    assert r["losses"].shape == ref.shape
    drift = np.abs(_smooth(alt[:, 0]) - _smooth(ref[:, 0])).max()
    err = np.abs(_smooth(r["losses"][:, 0]) - _smooth(ref[:, 0])).max()
    drift_hip = np.abs(_smooth(runs[1]["losses"][:, 0]) - _smooth(r["losses"][:, 0])).max()
    print(f"step-2 CE curve: max smoothed |hip-ref| {err:.4f}, reference-vs-reference drift {drift:.4f}, "
          f"hip-vs-hip drift {drift_hip:.4f}")
    assert err <= 2 * max(drift, drift_hip) + 0.02 * _smooth(ref[:, 0]).mean(), (err, drift, drift_hip)
    # ---- final mIoU: the two implementations' run-to-run distributions, from the data
    # recorded HIP samples: ONLY those of the build under test (tests/helpers.kernel_build_id; samples
    # of superseded kernel sets stay in the golden, tagged, and do not count -- a kernel change that
    # shifted the mean could otherwise hide behind them)
    build = Hh.kernel_build_id()
    tags = G["hip_build"] if "hip_build" in G.files else np.array(["untagged"] * len(G["hip_seeds"]))
    mine = np.array([str(t) == build for t in tags])
    print(f"build under test {build}: {int(mine.sum())} recorded HIP runs of this build "
          f"({len(mine) - int(mine.sum())} of older builds excluded)")
    assert mine.sum() >= 32, (f"only {int(mine.sum())} recorded mIoU samples of build {build}: re-sample with "
                              "tools/miou_hip_sample.py --seeds 4001-4040 and tools/merge_miou_samples.py")
    for name in ("new", "old"):
        refs = [float(v) for v in G[f"ref_miou_{name}"]]
        rec = [float(v) for v in G[f"hip_miou_{name}"][mine]]
        live = [x["miou_" + name] for x in runs]
        mh, sh, nh = _stats(rec)
        mr, sr, nr = _stats(refs)
        d_rec, se_rec = mean_difference(rec, refs)
        d_all, se_all = mean_difference(rec + live, refs)
        print(f"mIoU {name}: reference {nr} independent runs mean {mr * 100:.3f} sigma {sr * 100:.3f}; HIP "
              f"recorded {nh} runs mean {mh * 100:.3f} sigma {sh * 100:.3f}; live {np.round(np.array(live) * 100, 3)}; "
              f"difference of the means {d_all:+.3f} +- {se_all:.3f} points (recorded only {d_rec:+.3f} +- {se_rec:.3f})")
        # every live run is a plausible member of the recorded HIP distribution.  sigma = the larger of the
        # two implementations' estimates: the old-domain head's distribution is left-skewed (reference
        # range 74.4 .. 82.6 around a mean of 80.2), the estimate from 80 HIP runs (1.66) sits below the
        # reference's (2.17), and one of this build's two live runs (73.70) lies 3.9 of the smaller sigmas out
        assert all(abs(v - mh) <= 4 * max(sh, sr) for v in live), (name, live, mh, sh, sr)
        if name == "new":
            # the north_star's statement, on the head the step trains: means within 0.1 point,
            # resolved to 0.1 point
            assert nr >= 24 and nh >= 32, (nr, nh)
            assert se_all <= 0.1, (se_all, sr, sh, nr, nh)
            assert abs(d_all) <= 0.1, (d_all, se_all)
        else:
            # old-domain head: its BN running statistics are overwritten by the KD forward (reference
            # quirk), run-to-run sigma is ~2 points in BOTH implementations: the
            # means must agree within 3 standard errors of their difference
            assert abs(d_all) <= 3 * se_all, (d_all, se_all)
