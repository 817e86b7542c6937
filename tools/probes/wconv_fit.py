#!/usr/bin/env python3
"""Fixed cost of a Winograd conv launch: times the forms of the step at batch 2 .. 12 (1 .. 6 wave
tiles per wave at the bench's layer shapes) and fits  t = a + b * N  -- a = what a launch pays
whatever its size (weights -> LDS, first operands, last tile's epilogue, finalize), b * 6 = the
batch-6 body.

    python tools/probes/wconv_fit.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mdil_ss_amd  # noqa: E402,F401
from mdil_ss_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    Ns = (2, 4, 6, 8, 12)
    for C, (H, W), d in ((128, (64, 128), 2), (64, (128, 256), 1)):
        res = {}
        for N in Ns:
            x = torch.randn(N, H, W, C, device=dev).relu_()
            x2 = torch.randn(N, H, W, C, device=dev)
            out = torch.empty_like(x)
            w3 = torch.randn(C, C, 3, 1, device=dev) * 0.05
            w13 = torch.randn(C, C, 1, 3, device=dev) * 0.05
            pw = torch.randn(C, C, 1, 1, device=dev) * 0.05
            b = torch.randn(C, device=dev)
            g3 = ops.make_geom(N, H, W, H, W, ops._taps_3x1(d), C, H, W, C)
            g13 = ops.make_geom(N, H, W, H, W, ops._taps_1x3(d), C, H, W, C)
            g4 = ops.make_geom(N, H, W, H, W, ops._taps_1x3(d) + [(0, 0, 1)], C, H, W, C)
            wp = ops.pack_conv(w3, "fwd")
            wp13 = ops.pack_conv(w13, "fwd")
            wp4 = ops.pack_pair(w13, pw, "fwd")
            gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
            rm_, rv_ = torch.zeros(C, device=dev), torch.ones(C, device=dev)
            nbt_ = torch.zeros((), dtype=torch.int64, device=dev)
            forms = {
                "3x1 bias+relu": lambda: ops.tapconv(g3, C, C, x, None, wp, out, bias=b, relu=True),
                "1x3+adapter": lambda: ops.tapconv(g4, C, C, x, x2, wp4, out, bias=b),
                "1x3+adapter +stats+finalize": lambda: ops.tapconv_bn(g4, C, C, x, x2, wp4, out, gam, bet, rm_, rv_, nbt_, bias=b, bias2=b),
                "dgrad 1x3 gate": lambda: ops.tapconv(g13, C, C, x2, None, wp13, out, gate=x),
            }
            for k, f in forms.items():
                res.setdefault(k, []).append(timeit(f))
            ops.invalidate_packs()
        for k, ts in res.items():
            bfit, afit = np.polyfit(np.array(Ns, float), np.array(ts), 1)
            print(f"C={C:3d} {k:30s} " + " ".join(f"N={n}: {t:6.1f}" for n, t in zip(Ns, ts)) +
                  f"   fit: {afit:5.1f} us fixed + {bfit:5.2f} us/image (batch 6 body {6 * bfit:5.1f} us)"
                  f"   2 x N=6 -> N=12: {2 * ts[2]:5.1f} -> {ts[4]:5.1f} ({100 * (1 - ts[4] / (2 * ts[2])):4.1f} % saved)", flush=True)


if __name__ == "__main__":
    main()
