#!/usr/bin/env python3
"""Print the mIoU statistics tests/test_miou_parity.py asserts, from tests/golden/miou_run.npz:
both samples' sizes, means, standard deviations, the difference of the means and its standard
error (Welch), new-domain and old-domain head -- first for the HIP runs of the BUILD UNDER TEST
(tests/helpers.kernel_build_id: the only ones the test counts), then for the HIP runs of every build
recorded so far (superseded kernel sets included; context, not asserted)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.helpers import kernel_build_id  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "miou_run.npz"))
build = kernel_build_id()
tags = G["hip_build"] if "hip_build" in G.files else np.array(["untagged"] * len(G["hip_seeds"]))
mine = np.array([str(t) == build for t in tags])
for what, sel in ((f"build under test {build}", mine), ("every recorded build", np.ones(len(mine), bool))):
    for name in ("new", "old"):
        r, h = G[f"ref_miou_{name}"] * 100, G[f"hip_miou_{name}"][sel] * 100
        if len(h) < 2:
            print(f"{name}-domain head: reference {len(r)} runs | HIP {len(h)} runs ({what}): not enough samples")
            continue
        se = (h.var(ddof=1) / len(h) + r.var(ddof=1) / len(r)) ** 0.5
        print(f"{name}-domain head: reference {len(r)} runs mean {r.mean():.3f} sigma {r.std(ddof=1):.3f} "
              f"[{r.min():.2f}, {r.max():.2f}] | HIP {len(h)} runs mean {h.mean():.3f} sigma {h.std(ddof=1):.3f} "
              f"[{h.min():.2f}, {h.max():.2f}] ({what}) | hip - ref = {h.mean() - r.mean():+.3f} +- {se:.3f}")
print("reference seeds", G["ref_seeds"].tolist())
print("HIP seeds", G["hip_seeds"].tolist())
