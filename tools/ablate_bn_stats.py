#!/usr/bin/env python3
"""Upper bound of what statistics in the `tapconv` / `c16conv` epilogues could buy (fourth request).

The BatchNorm statistics of the down- / up-samplers, the stem and the 16-channel decoder blocks are still a pass of
their own (`bn_stats_kernel`, 18 launches per step, ~25 us each at 12-31 % of HBM).  Fusing them into the producing
kernels' epilogues cannot save more than removing the launches altogether does.  This tool measures exactly that:
the step-2 iteration of `bench.py` (same models, batch, schedule) timed twice in one process -- as shipped, and with
`ops.bn_train_stats` replaced by a stub that returns the coefficient table of the previous real call WITHOUT launching
anything (timing only: the statistics are stale by construction, which changes no other launch and no byte moved).
Covers the 10 launches per step the host issues itself (stem, down- and up-samplers of both student graphs); the 8 of the
16-channel decoder blocks sit inside the block-level C ABI call (same kernel, 50 MB tensors: scale by 18 / 10 for a bound
on all of them).  Nothing here is reachable from the product.

    python tools/ablate_bn_stats.py [--steps 60]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=15)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    from mdil_ss_amd import ops
    from mdil_ss_amd.engine import Step2Engine
    student, teacher, T = bench.build_models(dev)
    T.current_task = 1
    eng = Step2Engine(student, teacher, torch.tensor(bench.WEIGHT_BDD, device=dev), current_task=1, lambdac=0.1,
                      is_shared=T.is_shared, is_ds_curr=T.is_DS_curr)
    g = torch.Generator().manual_seed(3)
    img = torch.rand(6, 3, 512, 1024, generator=g).to(dev)
    lab = torch.randint(0, 20, (6, 1, 512, 1024), generator=g).to(dev)
    real = ops.bn_train_stats
    cache, calls = {}, [0, 0]

    def stub(z, gamma, beta, rm, rv, nbt):
        key = (tuple(z.shape), gamma.data_ptr())
        calls[0] += 1
        if key not in cache:
            cache[key] = real(z, gamma, beta, rm, rv, nbt)
            calls[1] += 1
        return cache[key]

    def timed(label):
        for _ in range(a.warmup):
            eng.iteration(img, lab)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(a.steps):
            eng.iteration(img, lab)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / a.steps
        print(f"{label:58s} {dt * 1e3:7.3f} ms / step  {6 / dt:6.1f} img/s", flush=True)
        return dt

    res = []
    for rnd in range(2):
        ops.bn_train_stats = real
        res.append(("shipped", timed("shipped (18 bn_stats launches per step)")))
        ops.bn_train_stats = stub
        res.append(("stub", timed("the 10 host-issued bn_stats launches stubbed out")))
    ops.bn_train_stats = real
    s = sum(t for k, t in res if k == "shipped") / 2
    n = sum(t for k, t in res if k == "stub") / 2
    print(f"stub calls {calls[0]} ({calls[1]} real); upper bound of ANY fusion of these statistics: {(s - n) * 1e3:+.3f} ms / step "
          f"= {(s / n - 1) * 100:+.2f} % of the step")


if __name__ == "__main__":
    main()
