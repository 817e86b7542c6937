#!/usr/bin/env python3
"""Where a kernel's register spills sit relative to its MFMAs: compiles one .hip file to ISA and prints,
for every kernel that uses scratch, each scratch instruction with the number of MFMAs emitted before it
(spills inside the MFMA region are the ones that cost).

    python tools/spill_map.py mdil_ss_amd/csrc/w4conv.hip [substring of the demangled kernel name]
"""
import os
import re
import subprocess
import sys
import tempfile


def main():
    src = os.path.abspath(sys.argv[1])
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    extra = sys.argv[3:]
    with tempfile.TemporaryDirectory() as d:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src,
                        "-o", "x.o", "--save-temps"] + extra, cwd=d, stderr=subprocess.DEVNULL)
        asm = [f for f in os.listdir(d) if f.endswith("gfx950.s")][0]
        lines = open(os.path.join(d, asm)).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    starts.append((len(lines), None))
    for (a, name), (b, _) in zip(starts, starts[1:]):
        dem = subprocess.run(["c++filt", name], stdout=subprocess.PIPE, text=True).stdout.strip()
        dem = dem.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        if filt not in dem:
            continue
        body = lines[a:b]
        total = sum(1 for x in body if "v_mfma" in x)
        nm = 0
        ev = []
        for l in body:
            if "v_mfma" in l:
                nm += 1
            if "scratch_" in l:
                ev.append((nm, l.strip().split(";")[0].strip()))
        if not ev:
            continue
        inside = [e for e in ev if 0 < e[0] < total]
        print(f"{dem}: {total} MFMAs, {len(ev)} scratch instructions, {len(inside)} inside the MFMA region")
        for nm, l in inside[:40]:
            print(f"    after MFMA {nm:4d}: {l}")


if __name__ == "__main__":
    main()
