"""GPU: mIoU parity of a two-stage training run (north_star: "matching the reference's mIoU
within +-0.1 on identical inputs/seeds").  The golden (tests/golden/miou_run.npz) is
the imported REFERENCE model trained on CPU by tools/gen_miou_golden.py: step 1 on the first
domain (train_RAPFT_step1.py semantics, 7,680 iterations), then step 2 on the second domain with
KD from the step-1 model (train_new_task_step2.py, 4,096 iterations) -- the reference's batch
size, optimizer, LR schedules and loss, on a seeded procedural dataset that is learnable (all 19
evaluated classes present, well separated colours) and a validation set of 512 images per domain.
This test repeats the identical protocol (tests/miou_protocol.py: same init, batches, dropout
masks) on the HIP path -- Step1Engine then Step2Engine (3-stream schedule) -- and compares the
final mIoU of both validation sets (iouEval.py:72-77) and the loss curves.

What can be resolved: the golden holds the SAME reference code run at two CPU thread counts
(different fp32 summation orders inside oneDNN); their difference is the protocol's own fp32
noise floor, printed next to the HIP-vs-reference delta.  The HIP path must match the reference
within 0.1 mIoU point or within the reference's own spread, whichever is larger."""
import os

import numpy as np
import pytest
import torch

from oracle import fixtures as fx
from oracle import rap_oracle as O
from tests import miou_protocol as MP

pytestmark = pytest.mark.gpu


def _smooth(x, k=200):
    return np.convolve(x, np.ones(k) / k, mode="valid")


def test_training_run_matches_reference_miou():
    G = np.load(os.path.join(os.path.dirname(__file__), "golden", "miou_run.npz"))
    dev = torch.device("cuda:0")
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import ops
    from mdil_ss_amd import train_new_task_step2 as T
    from mdil_ss_amd.engine import Step1Engine, Step2Engine
    from mdil_ss_amd.iouEval import iouEval
    from mdil_ss_amd.models.erfnet_RA_parallel import Net
    ops.invalidate_packs()
    cfg = MP.CONFIG
    weight = torch.tensor(fx.WEIGHT_BDD, device=dev)
    # ---- stage A: step 1 on the first domain -> the teacher
    teacher = Net([20], 1, 0)
    teacher.load_state_dict(MP.step1_initial_state())
    teacher.to(dev)
    engA = Step1Engine(teacher, weight, current_task=0)
    lossesA, it = [], 0
    for epoch in range(1, cfg["epochs_step1"] + 1):
        engA.optimizer.set_epoch(epoch, cfg["epochs_step1"])
        for images, labels in MP.train_batches(epoch, old_domain=True):
            q = [MP.masks_for(it, images.shape[0])[0]]
            teacher.mask_provider = lambda n: q.pop(0)
            lossesA.append(engA.iteration(images.to(dev), labels.to(dev)))     # device scalar: no sync
            it += 1
    lossesA = torch.stack(lossesA).double().cpu().numpy()
    refA, altA = G["losses_step1"], G["alt_losses_step1"]
    # only the first iteration is a deterministic function of the inputs; Adam at lr 5e-4 on every
    # parameter (sign-like first steps) makes the second one already differ at the 1e-4 level
    np.testing.assert_allclose(lossesA[0], refA[0], rtol=1e-5)
    np.testing.assert_allclose(lossesA[:5], refA[:5], rtol=1e-2)
    driftA = np.abs(_smooth(altA) - _smooth(refA)).max()
    errA = np.abs(_smooth(lossesA) - _smooth(refA)).max()
    print(f"step-1 CE curve: max smoothed |hip-ref| {errA:.4f}, reference thread-count drift {driftA:.4f}")
    assert errA <= 2 * driftA + 0.02 * _smooth(refA).mean(), (errA, driftA)
    # ---- stage B: step 2 with KD from the step-1 model
    teacher.eval()
    teacher.mask_provider = None
    teacher_sd = {k: v.detach().cpu().clone() for k, v in teacher.state_dict().items()}
    student = Net([20, 20], 2, 1)
    student.load_state_dict(MP.step2_student_state(teacher_sd))
    student.to(dev)
    frozen = Net([20], 1, 0)            # a fresh module for the frozen teacher (own parameter storage)
    frozen.load_state_dict(teacher_sd)
    frozen.to(dev)
    ops.invalidate_packs()
    T.current_task = 1
    T.apply_step2_freeze(student, frozen, 1)
    # MDIL_MIOU_SINGLE_STREAM=1: the single-stream schedule (another fp32 summation order of the
    # shared-encoder gradients) -- a second HIP sample for the noise-floor discussion in DESIGN.md
    eng = Step2Engine(student, frozen, weight, current_task=1, lambdac=cfg["lambdac"],
                      is_shared=T.is_shared, is_ds_curr=T.is_DS_curr,
                      streams=os.environ.get("MDIL_MIOU_SINGLE_STREAM") != "1")
    losses, it = [], 0
    for epoch in range(1, cfg["epochs"] + 1):
        eng.optimizer.set_epoch(epoch, cfg["epochs"])
        for images, labels in MP.train_batches(epoch):
            q = list(MP.masks_for(100000 + it, images.shape[0]))
            student.mask_provider = lambda n: q.pop(0)
            total, ce, kld = eng.iteration(images.to(dev), labels.to(dev))
            losses.append(torch.stack([ce, kld]))
            it += 1
    losses = torch.stack(losses).double().cpu().numpy()
    ref, alt = G["losses"], G["alt_losses"]
    assert losses.shape == ref.shape
    drift = np.abs(_smooth(alt[:, 0]) - _smooth(ref[:, 0])).max()
    err = np.abs(_smooth(losses[:, 0]) - _smooth(ref[:, 0])).max()
    print(f"step-2 CE curve: max smoothed |hip-ref| {err:.4f}, reference thread-count drift {drift:.4f}")
    assert err <= 2 * drift + 0.02 * _smooth(ref[:, 0]).mean(), (err, drift)
    student.eval()
    results = {}
    for task, name in ((1, "new"), (0, "old")):
        ev = iouEval(20, 19)
        with torch.no_grad():
            for images, labels in MP.val_batches(task):
                ev.addBatch(student(images.to(dev), task), labels.to(dev))
        m, _ = ev.getIoU()
        # the metric path itself (eval-mode forward with folded BN + fused argmax / confusion
        # kernel) against the oracle's eval forward + iouEval restatement on the SAME trained
        # weights: any difference beyond a few boundary pixels would be a bias of the eval path,
        # not of training
        S = {k: v.detach().cpu().clone() for k, v in student.state_dict().items()}
        tp = torch.zeros(19, dtype=torch.float64)
        fp_, fn = torch.zeros(19, dtype=torch.float64), torch.zeros(19, dtype=torch.float64)
        with torch.no_grad():
            for images, labels in MP.val_batches(task):
                a, b, c = O.iou_counts(O.net_forward(S, images, task, False).max(1)[1], labels[:, 0], 20, 19)
                tp += a
                fp_ += b
                fn += c
        m_oracle = float(O.miou(tp, fp_, fn)[0])
        print(f"mIoU {name}: HIP eval path {float(m) * 100:.4f} vs oracle eval of the same weights "
              f"{m_oracle * 100:.4f}")
        assert abs(float(m) - m_oracle) < 2e-4, (name, float(m), m_oracle)
        ref_runs = G[f"all_miou_{name}"]                 # the reference at several CPU thread counts
        spread = float(ref_runs.max() - ref_runs.min())
        delta = min(abs(float(m) - float(r)) for r in ref_runs)
        tol = max(0.001, spread)                         # mIoU in [0,1]; 0.001 = 0.1 point
        print(f"mIoU {name}: hip {float(m) * 100:.3f}  reference runs {np.round(ref_runs * 100, 3)} "
              f"(reference-vs-reference spread {spread * 100:.3f} points; |hip - nearest reference| "
              f"{delta * 100:.3f} points; tolerance {tol * 100:.3f})")
        results[name] = (float(m), float(ref_runs.min()) - tol, float(ref_runs.max()) + tol, ref_runs)
    for name, (m, lo, hi, ref_runs) in results.items():
        assert lo <= m <= hi, (name, m, ref_runs)
