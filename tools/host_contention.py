#!/usr/bin/env python3
"""Host-side proxy for 8 ranks sharing one node's CPU quota ("the risk is host-side
launch overhead and stragglers, not xGMI bandwidth").  An 8-GPU run is not ours to launch; what can
be measured on the 1-GPU box is the HOST cost of enqueuing a step-2 iteration when N processes do
it at the same time under the box's CPU quota (16 CPUs' worth): N concurrent tools/host_cost.py at
64x128 (the launch sequence of the full-size step, ~0 GPU work, so the one GPU is not what the
processes wait for), against the same measurement solo.

    python tools/host_contention.py [--procs 8] [--reps 150]
"""
import argparse
import os
import re
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))


def run(n, reps):
    cmd = [sys.executable, os.path.join(HERE, "host_cost.py"), "--height", "64", "--width", "128",
           "--reps", str(reps)]
    t0 = time.time()
    ps = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(n)]
    outs = [p.communicate()[0] for p in ps]
    med = []
    for o in outs:
        m = re.search(r"median ([0-9.]+) ms, min ([0-9.]+) ms", o)
        if m:
            med.append((float(m.group(1)), float(m.group(2))))
    return med, time.time() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--reps", type=int, default=150)
    a = ap.parse_args()
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        cpus = "unlimited" if quota == "max" else f"{int(quota) / int(period):.0f}"
    except (OSError, ValueError):
        cpus = "?"
    print(f"host CPUs: {os.cpu_count()} logical, cgroup quota {cpus}")
    for n in a.procs:
        med, wall = run(n, a.reps)
        if not med:
            print(f"{n} process(es): no result")
            continue
        ms = sorted(m for m, _ in med)
        print(f"{n} concurrent process(es): host enqueue per step-2 iteration median "
              f"{ms[len(ms) // 2]:.2f} ms (per process: {' '.join(f'{m:.2f}' for m in ms)}; "
              f"best minimum {min(b for _, b in med):.2f} ms), wall {wall:.0f} s", flush=True)


if __name__ == "__main__":
    main()
