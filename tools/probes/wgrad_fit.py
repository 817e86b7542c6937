#!/usr/bin/env python3
"""Weight-gradient launches of the step at batch N (run under `rocprofv3 --kernel-trace --stats`, one
process per N; tools/gpu/r4_call14.sh fits time = a + b N per kernel from the averages).

    python tools/probes/wgrad_fit.py --N 6
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mdil_ss_amd  # noqa: E402,F401
from mdil_ss_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=6)
    ap.add_argument("--iters", type=int, default=12)
    a = ap.parse_args()
    N = a.N
    for C, (H, W), d in ((128, (64, 128), 2), (64, (128, 256), 1)):
        x = torch.randn(N, H, W, C, device=dev).relu_()
        x2 = torch.randn(N, H, W, C, device=dev)
        g = torch.randn(N, H, W, C, device=dev)
        w3 = torch.randn(C, C, 3, 1, device=dev) * 0.05
        w13 = torch.randn(C, C, 1, 3, device=dev) * 0.05
        pw = torch.randn(C, C, 1, 1, device=dev) * 0.05
        b = torch.randn(C, device=dev)
        g3 = ops.make_geom(N, H, W, H, W, ops._taps_3x1(d), C, H, W, C)
        g13 = ops.make_geom(N, H, W, H, W, ops._taps_1x3(d), C, H, W, C)
        g4 = ops.make_geom(N, H, W, H, W, ops._taps_1x3(d) + [(0, 0, 1)], C, H, W, C)
        for _ in range(a.iters):
            ops.wgrad(g3, C, C, x, None, g, (0, 1, 2), C * 3, 3, w3, b)                 # wgradw
            ops.wgrad(g13, C, C, x, None, g, (0, 1, 2), C * 3, 3, w13, b)               # wgradx
            ops.wgrad(g4, C, C, x, x2, g, (0, 1, 2), C * 3, 3, w13, b, second=(1, C, 1, pw, b))   # wgrad2 4 taps
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
