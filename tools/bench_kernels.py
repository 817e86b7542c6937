#!/usr/bin/env python3
"""Micro-benchmarks of the hot kernels at the full-size shapes of BASELINE config 3 (N=6,
512x1024): HIP-event timing over repeated launches on the launch stream.

    python tools/bench_kernels.py [--filter tapconv128] [--iters 30]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mdil_ss_amd  # noqa: E402,F401
from mdil_ss_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--filter", default="")
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    N = 6
    shapes = {128: (64, 128), 64: (128, 256), 16: (256, 512)}
    rows = []

    def add(name, fn, flops=None, bytes_=None, executed=None):
        """flops: algorithmic (direct-form) flop of the launch; executed: the fraction of them the MFMA
        pipe executes (Winograd F(4,3): 1/2 for 3 taps, 5/8 for 3 taps + adapter; F(2,3): 2/3, 3/4; None: all)."""
        if a.filter and a.filter not in name:
            return
        t = timeit(fn, a.iters)
        s = f"{name:44s} {t * 1e6:9.1f} us"
        if flops:
            ex = flops * (executed if executed else 1.0)
            s += (f"  executed {ex / t / 1e12:7.2f} TFLOP/s = {ex / t / 1e12 / 157.3 * 100:5.1f}% of the fp32 MFMA peak"
                  f" (algorithmic-equivalent {flops / t / 1e12:7.2f} TFLOP/s)")
        if bytes_:
            s += f"  {bytes_ / t / 1e9:8.1f} GB/s ({bytes_ / t / 1e9 / 8000 * 100:5.1f}% of 8 TB/s)"
        print(s, flush=True)

    for C, (H, W) in shapes.items():
        x = torch.randn(N, H, W, C, device=dev).relu_()
        x2 = torch.randn(N, H, W, C, device=dev)
        out = torch.empty_like(x)
        npix = N * H * W
        T = npix * C * 4
        w3 = torch.randn(C, C, 3, 1, device=dev) * 0.05
        w13 = torch.randn(C, C, 1, 3, device=dev) * 0.05
        pw = torch.randn(C, C, 1, 1, device=dev) * 0.05
        b = torch.randn(C, device=dev)
        for d in ((2, 16) if C == 128 else (1,)):
            g3 = ops.make_geom(N, H, W, H, W, ops._taps_3x1(d), C, H, W, C)
            wp = ops.pack_conv(w3, "fwd")
            f43 = not (os.environ.get("MDIL_NO_W4CONV") or os.environ.get("MDIL_NO_WCONV"))
            wino = ((0.5 if f43 else 2 / 3) if not os.environ.get("MDIL_NO_WCONV") else None) if C != 16 else None
            wino_ad = 0.625 if (f43 and C == 128) else 0.75      # C = 64 + adapter stays on F(2,3)
            add(f"tapconv{C} 3x1 d{d} bias+relu", lambda: ops.tapconv(g3, C, C, x, None, wp, out, bias=b, relu=True),
                2.0 * npix * 3 * C * C, 2 * T, wino)
            g13 = ops.make_geom(N, H, W, H, W, ops._taps_1x3(d), C, H, W, C)
            wp13 = ops.pack_conv(w13, "fwd")
            add(f"tapconv{C} 1x3 d{d} bias", lambda: ops.tapconv(g13, C, C, x, None, wp13, out, bias=b),
                2.0 * npix * 3 * C * C, 2 * T, wino)
            if C != 16:
                g4 = ops.make_geom(N, H, W, H, W, ops._taps_1x3(d) + [(0, 0, 1)], C, H, W, C)
                wp4 = ops.pack_pair(w13, pw, "fwd")
                add(f"tapconv{C} 1x3+adapter d{d}", lambda: ops.tapconv(g4, C, C, x, x2, wp4, out, bias=b),
                    2.0 * npix * 4 * C * C, 3 * T, wino_ad)
            add(f"tapconv{C} dgrad1x3 d{d} gate", lambda: ops.tapconv(g13, C, C, x2, None, wp13, out, gate=x),
                2.0 * npix * 3 * C * C, 3 * T, wino)
            if C != 16:
                # the forms the training step launches: statistics / BN-backward reductions / block-boundary tail
                from mdil_ss_amd import _lib
                import ctypes as CT
                gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
                rm_, rv_ = torch.zeros(C, device=dev), torch.ones(C, device=dev)
                nbt_ = torch.zeros((), dtype=torch.int64, device=dev)
                add(f"conv{C} 1x3+adapter d{d} +stats+finalize", lambda: ops.tapconv_bn(
                    g4, C, C, x, x2, wp4, out, gam, bet, rm_, rv_, nbt_, bias=b, bias2=b),
                    2.0 * npix * 4 * C * C, 3 * T, wino_ad)
                add(f"conv{C} 1x3 d{d} +stats+finalize", lambda: ops.tapconv_bn(
                    g13, C, C, x, None, wp13, out, gam, bet, rm_, rv_, nbt_, bias=b),
                    2.0 * npix * 3 * C * C, 2 * T, wino)
                coef_ = ops.bn_train_stats(x2, gam, bet, rm_, rv_, nbt_)
                g3a = ops.make_geom(N, H, W, H, W, ops._taps_3x1(d, True) + [(0, 0, 1)], C, H, W, C)
                wp3a = ops.pack_pair(w3, pw, "fwd")
                add(f"dgrad{C} 3x1+adapterT d{d} gate +bnred", lambda: ops.tapconv_bnred(
                    g3a, C, C, x, x2, wp3a, out, x, x2, coef_), 2.0 * npix * 4 * C * C, 5 * T, wino_ad)
                lib_ = _lib.load()
                nblk_ = lib_.mdil_tapconv_tail_blocks(CT.byref(g3a), C, C)
                if nblk_ > 0:
                    part_ = torch.empty(256 * 2 * C, device=dev)
                    drop_ = torch.ones(N, C, device=dev)
                    res_ = torch.randn(N, H, W, C, device=dev)
                    zt_ = torch.randn(N, H, W, C, device=dev)
                    tl = _lib.BnTail(x.data_ptr(), zt_.data_ptr(), coef_.data_ptr(), coef_.data_ptr() + 4 * C,
                                     drop_.data_ptr(), part_.data_ptr())
                    ep = _lib.Epilogue(None, None, None, res_.data_ptr(), None, None, 0, None)
                    def tail_fn():
                        _lib.check(lib_.mdil_tapconv_tail(CT.byref(g3a), C, C, x.data_ptr(), x2.data_ptr(),
                                                          wp3a.data_ptr(), CT.byref(ep), out.data_ptr(),
                                                          CT.byref(tl), ops._stream()), "tail")
                    add(f"dgrad{C} 3x1+adapterT d{d} res +tail", tail_fn, 2.0 * npix * 4 * C * C, 7 * T, wino_ad)
            add(f"wgrad{C} 3x1 d{d} (+reduce)", lambda: ops.wgrad(g3, C, C, x, None, x2, (0, 1, 2), C * 3, 3, w3, b),
                2.0 * npix * 3 * C * C, 2 * T)
            add(f"wgrad{C} 1x3 d{d} (+reduce)", lambda: ops.wgrad(g13, C, C, x, None, x2, (0, 1, 2), C * 3, 3, w13, b),
                2.0 * npix * 3 * C * C, 2 * T)
            add(f"wgrad{C} 1x1 adapter (+reduce)", lambda: ops.wgrad(
                ops.make_geom(N, H, W, H, W, [(0, 0, 0)], C, H, W, C), C, C, x, None, x2, (0,), C, 1, pw, b),
                2.0 * npix * C * C, 2 * T)
        gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        nbt = torch.zeros((), dtype=torch.int64, device=dev)
        coef = ops.bn_train_stats(x2, gamma, beta, rm, rv, nbt)
        add(f"bn_train_stats C{C}", lambda: ops.bn_train_stats(x2, gamma, beta, rm, rv, nbt), None, T)
        add(f"bn_apply relu C{C}", lambda: ops.bn_apply(x2, coef[2], coef[3], relu=True, out=out), None, 2 * T)
        add(f"bn_apply res relu C{C}", lambda: ops.bn_apply(x2, coef[2], coef[3], res=x, relu=True, out=out), None, 3 * T)
        add(f"bn_backward C{C}", lambda: ops.bn_backward(x2, x, None, x2, gamma, beta, coef, True, out=out), None, 7 * T)
    # losses at full resolution
    L = torch.randn(6, 512, 1024, 20, device=dev)
    L2 = torch.randn(6, 512, 1024, 20, device=dev)
    tgt = torch.randint(0, 20, (6, 512, 1024), device=dev)
    wgt = torch.rand(20, device=dev)
    lg = L.permute(0, 3, 1, 2).requires_grad_(True)
    def ce_fb():
        lg.grad = None
        ops.cross_entropy2d(lg, tgt, wgt).backward()
    def kld_fb():
        lg.grad = None
        ops.kld_prob(lg, L2.permute(0, 3, 1, 2)).backward()
    add("ce fwd+bwd (3 logit passes)", ce_fb, None, 3 * L.numel() * 4)
    add("kld fwd+bwd (5 logit passes)", kld_fb, None, 5 * L.numel() * 4)
    # the fused head: output_conv + loss without logits in memory (csrc/head.hip); bytes = the
    # algorithmic traffic of the fused form (features read, labels read, feature gradient written)
    xf = torch.randn(6, 256, 512, 16, device=dev).relu_().requires_grad_(True)
    xt = torch.randn(6, 256, 512, 16, device=dev).relu_()
    wh = (torch.randn(16, 20, 2, 2, device=dev) * 0.3).requires_grad_(True)
    bh = torch.zeros(20, device=dev, requires_grad=True)
    wt, bt = torch.randn(16, 20, 2, 2, device=dev) * 0.3, torch.zeros(20, device=dev)
    F_ = xf.numel() * 4
    def head_ce_fb():
        xf.grad = wh.grad = bh.grad = None
        ops.head_ce(xf, wh, bh, tgt, wgt).backward()
    def head_kld_fb():
        xf.grad = None
        ops.head_kld(xf, wh.detach(), bh.detach(), xt, wt, bt).backward()
    def unfused_ce_fb():
        xf.grad = wh.grad = bh.grad = None
        ops.cross_entropy2d(ops.OutFn.apply(xf, wh, bh).permute(0, 3, 1, 2), tgt, wgt).backward()
    add("head_ce fwd", lambda: ops.head_ce(xf.detach(), wh.detach(), bh.detach(), tgt, wgt), None, F_ + tgt.numel() * 8)
    add("head_ce fwd+bwd (gx, dw, db)", head_ce_fb, None, 3 * F_ + 2 * tgt.numel() * 8)
    add("head_kld fwd+bwd (gx; frozen head)", head_kld_fb, None, 5 * F_)
    add("unfused outconv+ce fwd+bwd (reference point)", unfused_ce_fb, None, 3 * F_ + 6 * L.numel() * 4)
    ops.invalidate_packs()


if __name__ == "__main__":
    main()
