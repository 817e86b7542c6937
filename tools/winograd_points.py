#!/usr/bin/env python3
"""Choice of the interpolation points of the 1-D Winograd F(4,3) transform used by csrc/w4conv.hip.

Builds the Toom-Cook matrices A^T (4x6), G (6x3), B^T (6x6) for points (0, +-a, +-b, inf), checks them
against the direct correlation in float64, and measures the fp32 rounding error of a C = 128, 3-tap
channel contraction (post-ReLU inputs, weights ~ N(0, 0.05)) computed through them -- every step rounded to
fp32 as the kernel does -- relative to sum|x||w|, against an fp64 reference.  CPU only (numpy).

    python tools/winograd_points.py            # scan of symmetric point pairs + the direct form
"""
import numpy as np


def toom(points, m=4, r=3):
    n = m + r - 1
    p = np.array(points, float)
    AT, G = np.zeros((m, n)), np.zeros((n, r))
    for j in range(n - 1):
        N = np.prod([p[j] - p[l] for l in range(n - 1) if l != j])
        for i in range(m):
            AT[i, j] = p[j] ** i
        for k in range(r):
            G[j, k] = p[j] ** k / N
    AT[m - 1, n - 1] = 1
    G[n - 1, r - 1] = 1
    rows, rhs = [], []                    # B^T from  sum_j AT[i,j] G[j,k] BT[j,l] = [l == i + k]
    for i in range(m):
        for k in range(r):
            for l in range(n):
                row = np.zeros((n, n))
                row[:, l] = AT[i, :] * G[:, k]
                rows.append(row.ravel())
                rhs.append(1.0 if l == i + k else 0.0)
    BT = np.linalg.lstsq(np.array(rows), np.array(rhs), rcond=None)[0].reshape(n, n)
    return AT, G, BT


def main():
    rng = np.random.default_rng(0)
    C, npx = 128, 2048
    x = np.maximum(rng.standard_normal((npx, 6, C)), 0).astype(np.float32)      # the 6 inputs of a quad
    w = (rng.standard_normal((3, C, C)) * 0.05).astype(np.float32)
    x64, w64 = x.astype(np.float64), w.astype(np.float64)
    ref = np.stack([sum(x64[:, i + k, :] @ w64[k].T for k in range(3)) for i in range(4)], 1)
    mag = np.stack([sum(np.abs(x64[:, i + k, :]) @ np.abs(w64[k]).T for k in range(3)) for i in range(4)], 1)

    def err(y):
        e = (y - ref) / mag
        return np.sqrt((e ** 2).mean()), np.abs(e).max()

    y = np.stack([sum((x[:, i + k, :] @ w[k].T) for k in range(3)) for i in range(4)], 1)
    print("direct form (fp32)                 rms %.2e  max %.2e" % err(y))
    res = []
    for a in (0.5, 0.625, 0.75, 0.875, 1.0):
        for b in (1.25, 1.5, 1.75, 2.0, 2.5):
            AT, G, BT = toom((0, a, -a, b, -b))
            d, g = rng.standard_normal(6), rng.standard_normal(3)
            chk = AT @ ((G @ g) * (BT @ d)) - np.array([sum(g[k] * d[i + k] for k in range(3)) for i in range(4)])
            assert np.abs(chk).max() < 1e-12
            AT32, G32, BT32 = AT.astype(np.float32), G.astype(np.float32), BT.astype(np.float32)
            U = np.einsum("jk,koi->joi", G32, w).astype(np.float32)
            V = np.einsum("jl,pli->pji", BT32, x).astype(np.float32)
            M = np.stack([V[:, j, :] @ U[j].T for j in range(6)], 1).astype(np.float32)
            res.append(err(np.einsum("ij,pjo->pio", AT32, M).astype(np.float32)) + (a, b))
    for rms, mx, a, b in sorted(res):
        tag = "  <- csrc/w4conv.hip" if (a, b) == (0.75, 1.5) else ("  <- textbook" if (a, b) == (1.0, 2.0) else "")
        print("points (0, +-%.3f, +-%.3f, inf)   rms %.2e  max %.2e%s" % (a, b, rms, mx, tag))


if __name__ == "__main__":
    main()
