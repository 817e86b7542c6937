#!/usr/bin/env python3
"""mIoU-parity golden: a short step-2 training run of the IMPORTED REFERENCE model
on a seeded procedural dataset, on CPU, with explicit dropout masks -> tests/golden/miou_run.npz.

    python tools/gen_miou_golden.py        # ~5 min on 8 cores

The GPU test (tests/test_miou_parity.py) repeats the run on the HIP path with the same batches,
masks, init and schedule and compares the loss curve and the final mIoU on both validation sets.
"""
import os
import sys
import types
import importlib

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import fixtures as fx          # noqa: E402
from oracle import rap_oracle as O         # noqa: E402
from tests import miou_protocol as MP      # noqa: E402

def main(threads=8, tag="", perturb=0.0):
    torch.set_num_threads(threads)
    sys.path.insert(0, "/root/reference")
    ref_model = importlib.import_module("models.erfnet_RA_parallel")
    ref_iou = importlib.import_module("iouEval")
    cfg = MP.CONFIG
    state = {"masks": None, "k": 0}

    class Replay(torch.nn.Module):
        """stands in for blk.dropout on a reference instance: multiplies by the protocol's mask"""

        def __init__(self, p):
            super().__init__()
            self.p = p

        def forward(self, x):
            if not self.training:
                return x
            m = state["masks"][state["k"]]
            state["k"] += 1
            return x * m

    def patch_dropout(net):
        for blk in net.encoder.layers:
            if hasattr(blk, "dropout"):
                blk.dropout = Replay(blk.dropout.p)

    weight = torch.tensor(fx.WEIGHT_BDD)
    crit = torch.nn.NLLLoss(weight)
    # ---------------- stage A: step-1 training of the first domain (train_RAPFT_step1.py) ----------
    teacher = ref_model.Net([20], 1, 0)
    init = MP.step1_initial_state()
    if perturb:
        # noise-floor probe: the same reference code from an initial state that differs by a few
        # fp32 ulps (relative `perturb`) -- how far does ANY rounding-level difference carry?
        gp = torch.Generator().manual_seed(int(os.environ.get('MDIL_PERTURB_SEED', '123')))
        for k, v in init.items():
            if v.is_floating_point():
                v.mul_(1.0 + perturb * torch.randn(v.shape, generator=gp))
    teacher.load_state_dict(init)
    patch_dropout(teacher)
    optA = torch.optim.Adam(teacher.parameters(), 5e-4, (0.9, 0.999), eps=1e-8, weight_decay=1e-4)
    lossesA, it = [], 0
    for epoch in range(1, cfg["epochs_step1"] + 1):
        for g_ in optA.param_groups:
            g_["lr"] = O.poly_lr(5e-4, epoch, cfg["epochs_step1"])
        teacher.train()
        for images, labels in MP.train_batches(epoch, old_domain=True):
            state["masks"], state["k"] = MP.masks_for(it, images.shape[0])[0], 0
            ce = crit(torch.log_softmax(teacher(images, 0), 1), labels[:, 0])
            optA.zero_grad()
            ce.backward()
            optA.step()
            lossesA.append(ce.item())
            it += 1
        if epoch % 10 == 0:
            print(tag, "step1 epoch", epoch, np.mean(lossesA[-64:]), flush=True)
    teacher.eval()
    teacher_sd = {k: v.clone() for k, v in teacher.state_dict().items()}
    # ---------------- stage B: step-2 (CS -> BDD style) with KD ---------------------------------
    student = ref_model.Net([20, 20], 2, 1)
    student.load_state_dict(MP.step2_student_state(teacher_sd))
    for p in teacher.parameters():
        p.requires_grad = False
    for n, p in student.named_parameters():
        p.requires_grad = O.step2_trainable("module." + n, 1)
    named = [("module." + n, p) for n, p in student.named_parameters()]
    opt = torch.optim.Adam([{"params": [p for n, p in named if O.is_shared(n)], "lr": 5e-6},
                            {"params": [p for n, p in named if O.is_ds_curr(n, 1)]}],
                           5e-4, (0.9, 0.999), eps=1e-8, weight_decay=1e-4)
    base = [g["lr"] for g in opt.param_groups]
    patch_dropout(student)
    kl = torch.nn.KLDivLoss()
    losses = []
    it = 0
    for epoch in range(1, cfg["epochs"] + 1):
        f = O.poly_lr(1.0, epoch, cfg["epochs"])
        for g, b in zip(opt.param_groups, base):
            g["lr"] = b * f
        student.train()
        teacher.eval()
        for images, labels in MP.train_batches(epoch):
            m_new, m_old = MP.masks_for(100000 + it, images.shape[0])
            state["masks"], state["k"] = m_new + m_old, 0
            out = student(images, 1)
            out_prev = student(images, 0)
            with torch.no_grad():
                out_t = teacher(images, 0)
            ce = crit(torch.log_softmax(out, 1), labels[:, 0])
            kld = kl(torch.softmax(out_prev, 1), torch.softmax(out_t, 1))
            total = ce + cfg["lambdac"] * kld
            opt.zero_grad()
            total.backward()
            opt.step()
            losses.append([ce.item(), kld.item()])
            it += 1
            if it % 500 == 0:
                print(tag, it, losses[-1], flush=True)
    G = {"losses": np.array(losses), "losses_step1": np.array(lossesA)}
    student.eval()
    for task, name in ((1, "new"), (0, "old")):
        ev = ref_iou.iouEval(20, 19)
        with torch.no_grad():
            for images, labels in MP.val_batches(task):
                ev.addBatch(student(images, task).max(1)[1].unsqueeze(1), labels)
        m, per = ev.getIoU()
        G[f"miou_{name}"] = np.array(float(m))
        G[f"tp_{name}"], G[f"fp_{name}"], G[f"fn_{name}"] = ev.tp.numpy(), ev.fp.numpy(), ev.fn.numpy()
        print(tag, "mIoU", name, float(m))
    return G


if __name__ == "__main__":
    import argparse
    import warnings
    warnings.simplefilter("ignore")
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--out", default=None, help="write this single run to a scratch .npz")
    ap.add_argument("--perturb", type=float, default=0.0,
                    help="relative fp32-ulp-level perturbation of the initial weights (noise-floor probe)")
    ap.add_argument("--merge", nargs="*", default=None,
                    help="scratch runs (first = the golden run, the others = the same reference code at "
                         "other CPU thread counts: the protocol's own fp32 noise floor) -> tests/golden")
    a = ap.parse_args()
    if a.merge:
        runs = [dict(np.load(f)) for f in a.merge]
        G = runs[0]
        for k in ("losses", "losses_step1", "miou_new", "miou_old"):
            G["alt_" + k] = runs[1][k]
        G["all_miou_new"] = np.array([float(r["miou_new"]) for r in runs])
        G["all_miou_old"] = np.array([float(r["miou_old"]) for r in runs])
        G["threads"] = np.array([int(r["threads"]) for r in runs])
        G["perturbs"] = np.array([float(r["perturb"]) if "perturb" in r else 0.0 for r in runs])
        np.savez_compressed(os.path.join(REPO, "tests", "golden", "miou_run.npz"), **G)
        print("mIoU new:", G["all_miou_new"], " old:", G["all_miou_old"], "threads", G["threads"])
    else:
        G = main(a.threads, tag=f"[t{a.threads}{'p' if a.perturb else ''}]", perturb=a.perturb)
        G["threads"] = np.array(a.threads)
        G["perturb"] = np.array(a.perturb)
        G["perturb_seed"] = np.array(int(os.environ.get("MDIL_PERTURB_SEED", "123")))
        np.savez_compressed(a.out or f"/tmp/miou_run_t{a.threads}{'p' if a.perturb else ''}.npz", **G)
