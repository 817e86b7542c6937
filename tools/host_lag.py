#!/usr/bin/env python3
"""How far does the host run ahead of the GPU in the step-2 loop, and where does it wait?
Runs bench.py's step-2 configuration for a few dozen iterations; at every iteration start it records a
device event and looks up the newest event that has completed (lead = iterations in flight), and it times
the host phases of engine.Step2Engine._fwd_bwd_streams (forward enqueue / new-domain backward / old-domain
backward / join + Adam) with perf_counter.

    python tools/host_lag.py [--steps 40]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    a = ap.parse_args()
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import engine
    dev = torch.device("cuda:0")
    from mdil_ss_amd.engine import Step2Engine
    pool = []
    for i in range(8):
        g = torch.Generator().manual_seed(1234 + i)
        img = torch.rand(6, 3, 512, 1024, generator=g)
        lab = torch.randint(0, 20, (6, 1, 32, 64), generator=g).repeat_interleave(16, 2).repeat_interleave(16, 3).contiguous()
        pool.append((img.to(dev), lab.to(dev)))
    student, teacher, T = bench.build_models(dev)
    T.current_task = 1
    eng = Step2Engine(student, teacher, torch.tensor(bench.WEIGHT_BDD, device=dev), current_task=1, lambdac=0.1,
                      is_shared=T.is_shared, is_ds_curr=T.is_DS_curr)
    eng.optimizer.set_epoch(1, 150)
    marks = []
    real_backward = engine._backward

    def timed_backward(loss, streams=()):
        t0 = time.perf_counter()
        real_backward(loss, streams)
        marks.append(("bwd", time.perf_counter() - t0))
    engine._backward = timed_backward
    for i in range(6):
        eng.iteration(*pool[i % len(pool)])
    torch.cuda.synchronize()
    evs, rows = [], []
    t_prev = time.perf_counter()
    for i in range(a.steps):
        ev = torch.cuda.Event()
        ev.record()
        evs.append(ev)
        done = max([j for j, e in enumerate(evs) if e.query()], default=-1)
        marks.clear()
        t0 = time.perf_counter()
        eng.iteration(*pool[i % len(pool)])
        t1 = time.perf_counter()
        rows.append((i - done, (t1 - t0) * 1e3, [round(m[1] * 1e3, 2) for m in marks]))
    torch.cuda.synchronize()
    for i, (lead, ms, b) in enumerate(rows):
        print(f"iteration {i:3d}: host {ms:6.2f} ms, backward calls {b} ms, iterations in flight at its start: {lead}")


if __name__ == "__main__":
    main()
