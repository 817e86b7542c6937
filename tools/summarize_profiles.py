#!/usr/bin/env python3
"""gpurun_out/$TAG/ (tools/collect_profiles.sh, GPU box; TAG defaults to r03) -> the summaries
committed under profiles/."""
import collections
import csv
import glob
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = os.environ.get("TAG", "r05")
SRC = os.path.join(ROOT, "gpurun_out", TAG)
DST = os.path.join(ROOT, "profiles")


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


def main():
    for name in ("stats_single", "stats_3streams"):
        f = sorted(glob.glob(os.path.join(SRC, name, "*", "*_kernel_stats.csv")), key=os.path.getmtime, reverse=True)
        if f:
            out = os.path.join(DST, f"{TAG}_kernel_{name.replace('stats_single', 'stats_single_stream')}.csv")
            shutil.copy(f[0], out)
    for d in sorted(glob.glob(os.path.join(SRC, "pmc_*"))):
        f = sorted(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime, reverse=True)
        if not f:
            continue
        agg = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f[0])):
            k = (short(r["Kernel_Name"]), r["Counter_Name"])
            agg[k][0] += float(r["Counter_Value"])
            agg[k][1] += 1
        with open(os.path.join(DST, f"{TAG}_{os.path.basename(d)}.csv"), "w") as o:
            o.write("kernel,counter,launches,mean_per_launch\n")
            for (k, c), (v, n) in sorted(agg.items()):
                o.write(f'"{k}",{c},{n},{v / n:.1f}\n')
    for f in glob.glob(os.path.join(SRC, "bench_*.json")) + glob.glob(os.path.join(SRC, "*.txt")):
        shutil.copy(f, os.path.join(DST, f"{TAG}_{os.path.basename(f)}"))
    print(sorted(os.listdir(DST)))


if __name__ == "__main__":
    main()
