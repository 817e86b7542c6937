#!/usr/bin/env python3
"""cProfile of the host side of N step-2 iterations (where does enqueue time go?)."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    from mdil_ss_amd.engine import Step2Engine
    student, teacher, T = bench.build_models(dev)
    T.current_task = 1
    eng = Step2Engine(student, teacher, torch.tensor(bench.WEIGHT_BDD, device=dev), current_task=1,
                      lambdac=0.1, is_shared=T.is_shared, is_ds_curr=T.is_DS_curr)
    img = torch.rand(6, 3, 64, 128, device=dev)
    lab = torch.randint(0, 20, (6, 1, 64, 128), device=dev)
    for _ in range(4):
        eng.iteration(img, lab)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        eng.iteration(img, lab)
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(35)
    st.sort_stats("cumulative").print_stats(45)


if __name__ == "__main__":
    main()
