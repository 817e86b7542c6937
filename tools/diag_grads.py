#!/usr/bin/env python3
"""GPU diagnostic: run the golden step-2 iteration on the HIP path and dump every gradient next
to the oracle's (CPU) for offline comparison -> gpurun_out/diag_grads.npz"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fixtures as fx          # noqa: E402
from oracle import rap_oracle as O         # noqa: E402
from tests import helpers as Hh            # noqa: E402
from tests.test_model_golden import _build  # noqa: E402


def main():
    golden = np.load("tests/golden/step2_tiny.npz")
    dev = torch.device("cuda:0")
    from mdil_ss_amd import ops
    student, teacher = _build(golden, dev)
    names = [n for n, _ in student.named_parameters()]
    m_new, m_old = Hh.golden_masks(golden, 0)
    q = [m_new, m_old]
    student.mask_provider = lambda n: q.pop(0)
    images = torch.from_numpy(golden["it0_images"])
    labels = torch.from_numpy(golden["it0_labels"])
    weight = torch.tensor(fx.WEIGHT_BDD)
    student.train()
    teacher.eval()
    out_new = student(images.to(dev), 1)
    out_prev = student(images.to(dev), 0)
    with torch.no_grad():
        out_t = teacher(images.to(dev), 0)
    ce = ops.cross_entropy2d(out_new, labels[:, 0].to(dev), weight.to(dev))
    kld = ops.kld_prob(out_prev, out_t)
    (ce + 0.1 * kld).backward()
    # oracle
    t_sd, s_sd = Hh.golden_scenario(golden)
    for n in names:
        s_sd[n].requires_grad_(O.step2_trainable("module." + n, 1))
    O.step2_iteration(s_sd, t_sd, images, labels, weight, 1, 0.1, m_new, m_old)
    out = {}
    rows = []
    for n, p in student.named_parameters():
        if p.grad is None:
            continue
        g, r = p.grad.cpu().numpy(), s_sd[n].grad.numpy()
        out["hip_" + n] = g
        out["ref_" + n] = r
        den = np.linalg.norm(r)
        rows.append((np.linalg.norm(g - r) / max(den, 1e-12), den, n))
    rows.sort(reverse=True)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/diag_grads.txt", "w") as f:
        for e, d, n in rows[:40]:
            f.write(f"{e:.3e}  |ref|={d:.3e}  {n}\n")
    np.savez_compressed("gpurun_out/diag_grads.npz", **out)
    print(open("gpurun_out/diag_grads.txt").read())


if __name__ == "__main__":
    main()
