#!/usr/bin/env python3
"""Write the current mIoU statistics of tests/golden/miou_run.npz into DESIGN.md / README.md
(between the <!-- miou-stats --> markers)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "miou_stats.py")], capture_output=True, text=True).stdout
lines = out.strip().splitlines()
block = "<!-- miou-stats -->\n```\n" + "\n".join(lines[:4]) + "\n```\n<!-- /miou-stats -->"
new = re.search(r"new-domain head: reference (\d+) runs mean ([0-9.]+) sigma ([0-9.]+).*HIP (\d+) runs mean ([0-9.]+) sigma "
                r"([0-9.]+).*hip - ref = ([+-][0-9.]+) \+- ([0-9.]+)", lines[0]).groups()
allb = re.search(r"HIP (\d+) runs mean ([0-9.]+).*hip - ref = ([+-][0-9.]+) \+- ([0-9.]+)", lines[2]).groups()
short = (f"{new[4]} (HIP, {new[3]} runs of the build under test) vs {new[1]} (reference, {new[0]} runs): {new[6]} +- {new[7]} point; "
         f"over all {allb[0]} recorded HIP runs {allb[1]}: {allb[2]} +- {allb[3]}")
for name in ("DESIGN.md", "README.md"):
    p = os.path.join(ROOT, name)
    s = open(p).read()
    s = s.replace("MIOU_STATS_PLACEHOLDER", block)
    s = re.sub(r"<!-- miou-stats -->.*?<!-- /miou-stats -->", block, s, flags=re.S)
    s = s.replace("SAMPLES_PLACEHOLDER", "<!-- miou-short -->" + short + "<!-- /miou-short -->")
    s = s.replace("**MIOU_PLACEHOLDER**", "<!-- miou-short -->" + short + "<!-- /miou-short -->")
    s = re.sub(r"<!-- miou-short -->.*?<!-- /miou-short -->", "<!-- miou-short -->" + short + "<!-- /miou-short -->", s, flags=re.S)
    open(p, "w").write(s)
print(short)
