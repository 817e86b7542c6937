#!/usr/bin/env python3
"""Fold mIoU-protocol samples into tests/golden/miou_run.npz (arrays only).

    python tools/merge_miou_samples.py --ref gpurun_tmp/miou_ref/ref_*.npz /tmp/miou_extra_*.npz ... \
                                       --hip gpurun_out/miou_hip/hip_*.npz

Reference samples = runs of the IMPORTED reference (tools/gen_miou_golden.py --perturb 1e-7, one
seed each, made in the build container); HIP samples = runs of the product path on an MI355X
(tools/miou_hip_sample.py).  Only perturbed-seed runs count as independent samples (the
unperturbed reference runs at 2-4 threads are one trajectory).  Existing entries are kept; a reference
run is identified by its seed, a HIP run by (seed, build id) -- tests/helpers.kernel_build_id of the
sources that produced it, ``hip_build`` in the golden; samples recorded before builds were tagged
carry "untagged" and no longer count (tests/test_miou_parity.py uses the samples of the build under
test only).  The test derives every statistic from these arrays."""
import argparse
import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden", "miou_run.npz")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", nargs="*", default=[])
    ap.add_argument("--hip", nargs="*", default=[])
    a = ap.parse_args()
    G = dict(np.load(GOLD))
    # ---- reference side: keyed by seed
    have = {int(s): (float(n), float(o)) for s, n, o in zip(
        G.get("ref_seeds", []), G.get("ref_miou_new", []), G.get("ref_miou_old", []))}
    for f in a.ref:
        r = np.load(f)
        if not float(r["perturb"]):
            continue                                  # unperturbed thread-count variants: not independent
        seed = int(r["perturb_seed"]) if "perturb_seed" in r else None
        if seed is None:                              # runs made before the seed was recorded
            seed = {"miou_run_t3p": 123, "miou_run_t4p": 124}.get(
                os.path.basename(f).split(".")[0], int("".join(c for c in os.path.basename(f) if c.isdigit()) or 0))
        have[seed] = (float(r["miou_new"]), float(r["miou_old"]))
    seeds = sorted(have)
    G["ref_seeds"] = np.array(seeds, dtype=np.int64)
    G["ref_miou_new"] = np.array([have[s][0] for s in seeds])
    G["ref_miou_old"] = np.array([have[s][1] for s in seeds])
    # ---- HIP side: keyed by (seed, build)
    old_build = G.get("hip_build", np.array(["untagged"] * len(G.get("hip_seeds", []))))
    hh = {(int(s), str(b)): (float(n), float(o)) for s, b, n, o in zip(
        G.get("hip_seeds", []), old_build, G.get("hip_miou_new", []), G.get("hip_miou_old", []))}
    for f in a.hip:
        r = np.load(f)
        seed = int(r["seed"])
        if seed == 0:
            continue                                  # the unperturbed protocol: the live test runs it
        if str(r["variant"]) != "shipped build":
            continue                                  # A/B variants are not samples of the product
        hh[(seed, str(r["build"]) if "build" in r else "untagged")] = (float(r["miou_new"]), float(r["miou_old"]))
    keys = sorted(hh)
    G["hip_seeds"] = np.array([k[0] for k in keys], dtype=np.int64)
    G["hip_build"] = np.array([k[1] for k in keys])
    G["hip_miou_new"] = np.array([hh[k][0] for k in keys])
    G["hip_miou_old"] = np.array([hh[k][1] for k in keys])
    for b in sorted(set(G["hip_build"])):
        sel = G["hip_build"] == b
        x = G["hip_miou_new"][sel] * 100
        print(f"hip build {b}: {sel.sum()} runs, new-domain mean {x.mean():.3f}" +
              (f" sigma {x.std(ddof=1):.3f}" if sel.sum() > 1 else ""))
    np.savez_compressed(GOLD, **G)
    for side in ("ref", "hip"):
        x = G[f"{side}_miou_new"] * 100
        if len(x) > 1:
            print(f"{side}: {len(x)} runs, new-domain mIoU mean {x.mean():.3f} sigma {x.std(ddof=1):.3f} "
                  f"(SE {x.std(ddof=1) / len(x) ** 0.5:.3f}); old-domain mean {G[side + '_miou_old'].mean() * 100:.2f} "
                  f"sigma {G[side + '_miou_old'].std(ddof=1) * 100:.2f}")
    if len(G["ref_miou_new"]) > 1 and len(G["hip_miou_new"]) > 1:
        h, r = G["hip_miou_new"] * 100, G["ref_miou_new"] * 100
        se = (h.var(ddof=1) / len(h) + r.var(ddof=1) / len(r)) ** 0.5
        print(f"difference of the means (hip - ref): {h.mean() - r.mean():+.3f} +- {se:.3f} points")


if __name__ == "__main__":
    main()
