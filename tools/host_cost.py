"""Host cost of enqueuing one step-2 iteration: sync, call iteration(), stop the clock when the
call returns (queues empty at the start, so no back-pressure; the GPU work is still running).
Full-size tensors.  --profile prints the cProfile top of the same loop."""
import argparse
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--ab", action="store_true", help="alternate block-level / per-launch host paths")
    ap.add_argument("--graph", action="store_true", help="hipGraph replay of forward + losses + backward (Step2Engine.enable_graph)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    from mdil_ss_amd.engine import Step2Engine
    B, H, W = 6, args.height, args.width
    g = torch.Generator().manual_seed(1)
    img = torch.rand(B, 3, H, W, generator=g).to(dev)
    lab = torch.randint(0, 20, (B, 1, H, W), generator=g).to(dev)
    student, teacher, T = bench.build_models(dev)
    T.current_task = 1
    eng = Step2Engine(student, teacher, torch.tensor(bench.WEIGHT_BDD, device=dev), current_task=1,
                      lambdac=0.1, is_shared=T.is_shared, is_ds_curr=T.is_DS_curr)
    eng.optimizer.set_epoch(1, 150)
    for _ in range(5):
        eng.iteration(img, lab)
    if args.graph:
        eng.enable_graph(img, lab)
        for _ in range(3):
            eng.iteration(img, lab)
    if args.ab:
        from mdil_ss_amd import ops
        res = {True: [], False: []}
        for r in range(2 * args.reps):
            ops.BLOCK_ABI = r % 2 == 0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.iteration(img, lab)
            res[ops.BLOCK_ABI].append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        for k, v in res.items():
            v.sort()
            print(f"BLOCK_ABI={k}: median {v[len(v) // 2] * 1e3:.2f} ms, min {v[0] * 1e3:.2f} ms")
        return
    ts = []
    pr = cProfile.Profile() if args.profile else None
    for _ in range(args.reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if pr:
            pr.enable()
        eng.iteration(img, lab)
        if pr:
            pr.disable()
        ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    ts.sort()
    print(f"host enqueue per iteration: median {ts[len(ts) // 2] * 1e3:.2f} ms, min {ts[0] * 1e3:.2f} ms "
          f"({args.reps} reps, {H}x{W}, MDIL_PY_BLOCKS={os.environ.get('MDIL_PY_BLOCKS')}, hipGraph replay={bool(args.graph)})")
    if pr:
        pstats.Stats(pr).sort_stats("tottime").print_stats(28)


if __name__ == "__main__":
    main()
