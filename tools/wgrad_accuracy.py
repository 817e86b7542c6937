"""Accuracy of the Winograd weight-gradient kernels (wgradw / wgradx) vs the direct streaming kernel
(wgrad2) against an fp64 reference.  MDIL_NO_WGRADW=1 python tools/wgrad_accuracy.py = direct form."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mdil_ss_amd  # noqa: E402,F401
from mdil_ss_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    name = "direct (wgrad2)" if os.environ.get("MDIL_NO_WGRADW") else "Winograd (wgradw/x)"
    for C, H, W, d, axis in ((128, 64, 128, 4, "h"), (64, 128, 256, 1, "h"), (128, 64, 128, 16, "w"), (64, 128, 256, 1, "w")):
        N = 6
        x = F.relu(torch.randn(N, C, H, W, generator=g))
        go = torch.randn(N, C, H, W, generator=g) * 0.01
        kk, pad, dil = ((1, 3), (0, d), (1, d)) if axis == "w" else ((3, 1), (d, 0), (d, 1))
        w = torch.zeros(C, C, *kk, dtype=torch.float64, requires_grad=True)
        y = F.conv2d(x.double(), w, None, padding=pad, dilation=dil)
        y.backward(go.double())
        want = w.grad
        mag = None
        wa = torch.zeros(C, C, *kk, dtype=torch.float64, requires_grad=True)
        F.conv2d(x.double().abs(), wa, None, padding=pad, dilation=dil).backward(go.double().abs())
        mag = wa.grad
        taps = ops._taps_1x3(d) if axis == "w" else ops._taps_3x1(d)
        G = ops.make_geom(N, H, W, H, W, taps, C, H, W, C)
        xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
        gd = go.permute(0, 2, 3, 1).contiguous().to(dev)
        wd = torch.zeros(C, C, *kk, device=dev)
        dw, _ = ops.wgrad(G, C, C, xd, None, gd, (0, 1, 2), C * 3, 3, wd, None)
        e = (dw.cpu().double() - want) / mag
        per_tap = [float(e.reshape(C, C, 3)[:, :, k].pow(2).mean().sqrt()) for k in range(3)]
        print(f"{name:20s} C{C} {H}x{W} d{d} axis {axis}: err/sum|g||x| rms {e.pow(2).mean().sqrt():.3e} max {e.abs().max():.3e}"
              f"  rms per tap {per_tap[0]:.2e} {per_tap[1]:.2e} {per_tap[2]:.2e}")


if __name__ == "__main__":
    main()
