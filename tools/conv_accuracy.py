"""Accuracy of the Winograd conv kernels (w4conv: F(4,3), wconv: F(2,3)) vs the direct fp32-MFMA kernel
(sconv) against an fp64 reference, on the factorised convs of the path.  Run once per kernel (the
choice is read from the environment at load time): MDIL_NO_W4CONV=1 python tools/conv_accuracy.py
(F(2,3)), MDIL_NO_W4CONV=1 MDIL_NO_WCONV=1 ... (direct form)"""
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
    name = ("w4conv (Winograd F(4,3))" if not os.environ.get("MDIL_NO_W4CONV") and not os.environ.get("MDIL_NO_WCONV")
            else "wconv (Winograd F(2,3))" if not os.environ.get("MDIL_NO_WCONV") else "sconv (fp32 MFMA)")
    for C, H, W, d, relu_in in ((128, 96, 128, 4, True), (64, 128, 256, 1, True), (128, 64, 128, 16, False)):
        N = 2
        x = torch.randn(N, C, H, W, generator=g)
        if relu_in:
            x = F.relu(x)
        w = torch.randn(C, C, 3, 1, generator=g) * (1.0 / (3 * C)) ** 0.5
        b = torch.randn(C, generator=g) * 0.1
        want = F.conv2d(x.double(), w.double(), b.double(), padding=(d, 0), dilation=(d, 1))
        mag = F.conv2d(x.double().abs(), w.double().abs(), b.double().abs(), padding=(d, 0), dilation=(d, 1))
        xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
        wd = w.to(dev)
        G = ops.make_geom(N, H, W, H, W, ops._taps_3x1(d), C, H, W, C)
        out = ops.tapconv(G, C, C, xd, None, ops.pack_conv(wd, "fwd"), torch.empty_like(xd), bias=b.to(dev))
        got = out.permute(0, 3, 1, 2).cpu().double()
        e = (got - want).abs() / mag
        print(f"{name:24s} C{C} {H}x{W} d{d}: max err/sum|a||b| {e.max():.3e}  rms {e.pow(2).mean().sqrt():.3e}  "
              f"mean signed {((got - want) / mag).mean():+.3e}  max abs err {(got - want).abs().max():.3e}")
        ops.invalidate_packs()


if __name__ == "__main__":
    main()
