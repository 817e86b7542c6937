#!/usr/bin/env python3
"""Independent runs of the mIoU protocol (tests/miou_protocol.py) on the HIP path (needs an MI355X).

    python tools/miou_hip_sample.py --seeds 3001-3032 --procs 4 --out gpurun_out/miou_hip

Each seed is one run from initial weights perturbed by 1e-7 relative -- the same kind of sample
tools/miou_ref_sample.py draws from the reference -- written to <out>/hip_<seed>.npz (seed 0 = the
unperturbed protocol).  The 32x64 protocol is launch-bound, so several processes share the one GPU.
The kernel variant can be chosen by the environment (MDIL_NO_WCONV, MDIL_NO_WGRADW, ...);
tools/merge_miou_samples.py folds the results into tests/golden/miou_run.npz.
"""
import argparse
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def one(seed, out, checks=False):
    import numpy as np
    import torch
    torch.set_num_threads(4)          # the box grants 16 CPUs' worth of time, not the 256 it shows
    from tests.test_miou_parity import _run_protocol
    from tests.helpers import kernel_build_id
    tag = " ".join(k for k in sorted(os.environ) if k.startswith("MDIL_NO_")) or "shipped build"
    t0 = time.time()
    r = _run_protocol(torch.device("cuda:0"), f"{tag}, seed {seed}", perturb_seed=seed or None, checks=checks, oracle_eval=checks)
    np.savez_compressed(out, miou_new=r["miou_new"], miou_old=r["miou_old"], seed=seed,
                        losses=r["losses"], losses_step1=r["lossesA"], variant=tag,
                        device=torch.cuda.get_device_name(0), build=kernel_build_id())
    print(f"SAMPLE [{tag}] seed {seed} new {r['miou_new'] * 100:.3f} old {r['miou_old'] * 100:.3f} "
          f"({time.time() - t0:.0f} s)", flush=True)


def main():
    from tools.miou_ref_sample import parse_seeds
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="3001-3032")
    ap.add_argument("--procs", type=int, default=4)
    ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "miou_hip"))
    ap.add_argument("--stall", type=float, default=900.0,
                    help="give up (exit 3) when no worker has finished for this many seconds")
    ap.add_argument("--one", type=int, default=None, help="(worker) run this single seed")
    ap.add_argument("--checks", action="store_true",
                    help="with --one: also run the one-step parity checks from the trained states")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    if a.one is not None:
        return one(a.one, os.path.join(a.out, f"hip_{a.one}.npz"), a.checks)
    todo = [s for s in parse_seeds(a.seeds) if not os.path.exists(os.path.join(a.out, f"hip_{s}.npz"))]
    running = []
    failed = 0
    last = time.time()
    while todo or running:
        if time.time() - last > a.stall:
            for _, p in running:
                p.kill()
            print(f"no worker finished in {a.stall:.0f} s: giving up", flush=True)
            sys.exit(3)
        while todo and len(running) < a.procs:
            s = todo.pop(0)
            log = open(os.path.join(a.out, f"hip_{s}.log"), "w")
            env = dict(os.environ, OMP_NUM_THREADS="4", MKL_NUM_THREADS="4")
            running.append((s, subprocess.Popen([sys.executable, os.path.abspath(__file__), "--one", str(s),
                                                 "--out", a.out], stdout=log, stderr=subprocess.STDOUT,
                                                cwd=REPO, env=env)))
        time.sleep(2)
        for s, p in list(running):
            if p.poll() is not None:
                running.remove((s, p))
                last = time.time()
                tail = open(os.path.join(a.out, f"hip_{s}.log")).read().strip().splitlines()[-1:]
                print(f"seed {s}: rc {p.returncode} {tail}", flush=True)
                failed += p.returncode != 0
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
