#!/usr/bin/env python3
"""Where a training step's wall time goes on the GPU: reads a `rocprofv3 --kernel-trace` CSV
(*_kernel_trace.csv) of `bench.py`, cuts it into steps at the optimizer launches (adam_kernel, two
per step-2 iteration) and reports, for the steady-state steps, the time with at least one
matrix-pipe kernel resident, with only bandwidth-bound kernels resident, and with nothing resident,
plus each kernel family's busy time and its EXCLUSIVE time (nothing else running beside it).

    python tools/timeline.py path/to/kernel_trace.csv [--adam-per-step 2]
"""
import argparse
import csv
import re
from collections import defaultdict

MFMA = ("w4conv_kernel", "wconv_kernel", "sconv_kernel", "tapconv_kernel", "wgradw_kernel", "wgradx_kernel", "wgrad2_kernel",
        "wgrad_kernel", "wgrad16_kernel", "c16conv_kernel", "head_")


def family(name):
    m = re.search(r"(\w+?)(_kernel|<|\()", name.replace("(anonymous namespace)::", "").replace("void ", ""))
    base = name.replace("(anonymous namespace)::", "").replace("void ", "").split("<")[0].split("(")[0]
    if base.startswith("at::native"):
        return "torch elementwise"
    if "copyBuffer" in base or "fillBuffer" in base:
        return "copy/fill"
    return base


def union_len(iv):
    iv.sort()
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--adam-per-step", type=int, default=2)
    ap.add_argument("--skip", type=int, default=2, help="leading steps to drop (start-up, warm-up)")
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
    cuts = [rows[adam[i]][1] for i in range(a.adam_per_step - 1, len(adam), a.adam_per_step)]
    steps = list(zip(cuts[:-1], cuts[1:]))[a.skip:]
    if not steps:
        raise SystemExit("no complete steps in the trace")
    t0, t1 = steps[0][0], steps[-1][1]
    n = len(steps)
    sel = [r for r in rows if r[0] >= t0 and r[1] <= t1]
    mf = [(s, e) for s, e, k in sel if any(x in k for x in MFMA)]
    al = [(s, e) for s, e, k in sel]
    span = t1 - t0
    u_all, u_mf = union_len(al[:]), union_len(mf[:])
    print(f"steps analysed: {n}; step = {span / n / 1e6:.3f} ms; launches per step = {len(sel) / n:.0f}")
    print(f"  matrix-pipe kernel resident  {u_mf / n / 1e6:7.3f} ms")
    print(f"  only other kernels resident  {(u_all - u_mf) / n / 1e6:7.3f} ms")
    print(f"  nothing resident             {(span - u_all) / n / 1e6:7.3f} ms")
    # exclusive time per family: sweep over boundaries
    ev = []
    for s, e, k in sel:
        fam = family(k)
        ev.append((s, 1, fam))
        ev.append((e, -1, fam))
    ev.sort()
    live = defaultdict(int)
    excl = defaultdict(int)
    busy = defaultdict(int)
    cnt = defaultdict(int)
    last = None
    for t, d, fam in ev:
        if last is not None and t > last:
            act = [f for f, c in live.items() if c > 0]
            for f in act:
                busy[f] += t - last
            if len(act) == 1:
                excl[act[0]] += t - last
        live[fam] += d
        if d == 1:
            cnt[fam] += 1
        last = t
    # how deep do the persistent matrix-pipe kernels stack?  (each fills every CU with one work-group whose LDS /
    # register footprint excludes a second one: two "in flight" means the later one's work-groups are waiting for
    # CUs or taking them over one by one as the earlier launch drains -- VERDICT r5 #6)
    evm = sorted([(s_, 1) for s_, e_ in mf] + [(e_, -1) for s_, e_ in mf])
    depth, lastt, by_depth = 0, None, defaultdict(int)
    for t, d in evm:
        if lastt is not None and t > lastt:
            by_depth[min(depth, 3)] += t - lastt
        depth += d
        lastt = t
    tot = sum(by_depth.values()) or 1
    print("  matrix-pipe kernels in flight at once: " + ", ".join(
        f"{k}{'+' if k == 3 else ''}: {by_depth[k] / n / 1e6:.3f} ms ({by_depth[k] / tot * 100:.0f} %)" for k in (0, 1, 2, 3)))
    print(f"  {'family':34s} {'launches':>8s} {'busy ms':>8s} {'alone ms':>9s}   (per step)")
    for fam in sorted(busy, key=lambda f: -busy[f]):
        print(f"  {fam[:34]:34s} {cnt[fam] / n:8.1f} {busy[fam] / n / 1e6:8.3f} {excl[fam] / n / 1e6:9.3f}")


if __name__ == "__main__":
    main()
