#!/usr/bin/env python3
"""Independent reference runs of the mIoU protocol (tests/miou_protocol.py) in a process pool.

Each run is tools/gen_miou_golden.py (the IMPORTED reference model on CPU) from initial weights
perturbed by 1e-7 relative with its own seed -- one independent sample of the protocol's
run-to-run noise.  Runs only in the build container (needs /root/reference).

    nice -n 19 python tools/miou_ref_sample.py --procs 6 --seeds 2001-2048 --out gpurun_tmp/miou_ref

One thread per process is the efficient shape here (32x64 images: 0.113 s / step-1 iteration on 1
thread, 0.094 on 2).  Finished runs are skipped, so the command can be restarted.
tools/merge_miou_samples.py folds the results into tests/golden/miou_run.npz.
"""
import argparse
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse_seeds(s):
    out = []
    for part in s.split(","):
        if "-" in part:
            a, b = part.split("-")
            out += list(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=6)
    ap.add_argument("--seeds", default="2001-2048")
    ap.add_argument("--out", default=os.path.join(REPO, "gpurun_tmp", "miou_ref"))
    ap.add_argument("--perturb", type=float, default=1e-7)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    todo = [s for s in parse_seeds(a.seeds) if not os.path.exists(os.path.join(a.out, f"ref_{s}.npz"))]
    running = {}
    while todo or running:
        # a STOP file ends the pool after the runs in flight (frees the cores for other work)
        if os.path.exists(os.path.join(a.out, "STOP")):
            todo = []
        while todo and len(running) < a.procs:
            s = todo.pop(0)
            env = dict(os.environ, MDIL_PERTURB_SEED=str(s), OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
            log = open(os.path.join(a.out, f"ref_{s}.log"), "w")
            tmp = os.path.join(a.out, f"ref_{s}.part.npz")
            p = subprocess.Popen([sys.executable, os.path.join(REPO, "tools", "gen_miou_golden.py"),
                                  "--threads", "1", "--perturb", str(a.perturb), "--out", tmp],
                                 env=env, stdout=log, stderr=subprocess.STDOUT, cwd=REPO)
            running[s] = (p, tmp, time.time())
        time.sleep(20)
        for s, (p, tmp, t0) in list(running.items()):
            if p.poll() is None:
                continue
            del running[s]
            if p.returncode == 0 and os.path.exists(tmp):
                os.replace(tmp, os.path.join(a.out, f"ref_{s}.npz"))
                print(f"seed {s}: done in {(time.time() - t0) / 60:.1f} min", flush=True)
            else:
                print(f"seed {s}: FAILED rc={p.returncode}", flush=True)


if __name__ == "__main__":
    main()
