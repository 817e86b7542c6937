#!/usr/bin/env python3
"""Print the mIoU statistics tests/test_miou_parity.py asserts, from tests/golden/miou_run.npz:
both samples' sizes, means, standard deviations, the difference of the means and its standard
error (Welch), new-domain and old-domain head."""
import os
import sys

import numpy as np

G = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "miou_run.npz"))
for name in ("new", "old"):
    r, h = G[f"ref_miou_{name}"] * 100, G[f"hip_miou_{name}"] * 100
    se = (h.var(ddof=1) / len(h) + r.var(ddof=1) / len(r)) ** 0.5
    print(f"{name}-domain head: reference {len(r)} runs mean {r.mean():.3f} sigma {r.std(ddof=1):.3f} "
          f"[{r.min():.2f}, {r.max():.2f}] | HIP {len(h)} runs mean {h.mean():.3f} sigma {h.std(ddof=1):.3f} "
          f"[{h.min():.2f}, {h.max():.2f}] | hip - ref = {h.mean() - r.mean():+.3f} +- {se:.3f}")
print("reference seeds", G["ref_seeds"].tolist())
print("HIP seeds", G["hip_seeds"].tolist())
