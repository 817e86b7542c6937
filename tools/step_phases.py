#!/usr/bin/env python3
"""Where each stream of the step-2 schedule is at what time, WITHOUT a profiler in the way (rocprofv3 slows the
host enough to distort the staggered schedule): device events behind every plan step of the three forwards and
around both backward calls, read back after the run.  Prints, for steady-state iterations, the time (ms after the
iteration's first launch) at which every stream passes its marks, for the staggered and the lock-step schedule.

    python tools/step_phases.py [--iters 12] [--stagger 8|off]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def run(stagger, iters):
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import engine
    from mdil_ss_amd.engine import Step2Engine
    from mdil_ss_amd.models.erfnet_RA_parallel import Net
    dev = torch.device("cuda:0")
    pool = []
    for i in range(4):
        g = torch.Generator().manual_seed(1234 + i)
        img = torch.rand(6, 3, 512, 1024, generator=g)
        lab = torch.randint(0, 20, (6, 1, 32, 64), generator=g).repeat_interleave(16, 2).repeat_interleave(16, 3).contiguous()
        pool.append((img.to(dev), lab.to(dev)))
    student, teacher, T = bench.build_models(dev)
    T.current_task = 1
    eng = Step2Engine(student, teacher, torch.tensor(bench.WEIGHT_BDD, device=dev), current_task=1, lambdac=0.1,
                      is_shared=T.is_shared, is_ds_curr=T.is_DS_curr)
    eng.stagger = stagger
    eng.optimizer.set_epoch(1, 150)
    marks = []          # (iteration, tag, index, event)
    state = {"it": -1, "calls": 0}
    orig_plan = Net.plan

    def plan(self, task, masks=None, head=True):
        steps = orig_plan(self, task, masks, head)
        tag = "teacher" if self is teacher else ("new" if task == 1 else "old")

        def wrap(f, i):
            def g(y):
                out = f(y)
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks.append((state["it"], "F " + tag, i, ev))
                return out
            return g
        return [wrap(f, i) for i, f in enumerate(steps)]
    Net.plan = plan
    real_backward = engine._backward

    def backward(loss, streams=()):
        k = state["calls"]
        state["calls"] += 1
        real_backward(loss, streams)
        tag = "B both" if streams else ("B new" if k % 2 == 0 else "B old")
        evs = []
        for st in (streams or (torch.cuda.current_stream(),)):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(st)
            evs.append(ev)
        for j, ev in enumerate(evs):
            marks.append((state["it"], tag + (f" (stream {j})" if len(evs) > 1 else ""), 99, ev))
    engine._backward = backward
    starts, ends = [], []
    try:
        for i in range(iters + 4):
            state["it"] = i
            state["calls"] = 0
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            starts.append(ev)
            eng.iteration(*pool[i % len(pool)])
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            ends.append(ev)
        torch.cuda.synchronize()
    finally:
        Net.plan = orig_plan
        engine._backward = real_backward
    rows = {}
    for it, tag, idx, ev in marks:
        if it < 4:
            continue
        rows.setdefault((tag, idx), []).append(starts[it].elapsed_time(ev))
    step = np.median([starts[i].elapsed_time(ends[i]) for i in range(4, iters + 4)])
    print(f"== stagger {stagger}: median time of an iteration's launches on the GPU (first launch -> Adam done) {step:.2f} ms")
    for tag in ("F new", "F teacher", "F old"):
        pts = [(idx, np.median(v)) for (t, idx), v in rows.items() if t == tag]
        pts.sort()
        if pts:
            print(f"  {tag:10s} plan step done at (ms): " + " ".join(f"{i}:{t:.1f}" for i, t in pts if i in (0, 1, 6, 7, 11, 15, 16, 18, 19, 21)))
    for (t, idx), v in sorted(rows.items()):
        if t.startswith("B"):
            print(f"  {t:18s} backward drained at {np.median(v):.1f} ms")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--stagger", default="8", help="plan steps, or 'off' for the lock-step schedule (one process per schedule)")
    a = ap.parse_args()
    run(None if a.stagger == "off" else int(a.stagger), a.iters)


if __name__ == "__main__":
    main()
