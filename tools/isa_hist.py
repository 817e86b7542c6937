#!/usr/bin/env python3
"""Instruction histogram per kernel of a hipcc -S listing (gfx950): total / MFMA / VALU / DPP /
LDS / VMEM / SALU counts, divisions and spills.  Usage: tools/isa_hist.py file.s [name-filter]"""
import collections
import re
import sys


def main():
    txt = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for m in re.finditer(r"\n(_Z\w+):[^\n]*\n(.*?)\n\.Lfunc_end\d+:", txt, re.S):
        name, body = m.group(1), m.group(2)
        if flt not in name:
            continue
        cat = collections.Counter()
        for line in body.split("\n"):
            line = line.split(";")[0].strip()
            if not line or line.startswith(".") or line.endswith(":"):
                continue
            op = line.split()[0]
            cat["total"] += 1
            if op.startswith("v_mfma"):
                cat["mfma"] += 1
            elif op.startswith("ds_"):
                cat[op] += 1
            elif op.startswith(("buffer_", "global_", "scratch_", "flat_")):
                cat["_".join(op.split("_")[:2])] += 1
            elif op.startswith("v_"):
                cat["valu"] += 1
                if "dpp" in line or "row_" in line or "quad_perm" in line:
                    cat["dpp"] += 1
                if op.startswith(("v_div_scale", "v_rcp", "v_exp", "v_log")):
                    cat[op] += 1
            elif op.startswith("s_"):
                cat["salu"] += 1
        short = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", name)
        print(short[:60], dict(cat))


if __name__ == "__main__":
    main()
