#!/usr/bin/env python3
"""Register / LDS / spill summary of every kernel in one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py mdil_ss_amd/csrc/w4conv.hip [-D...]
"""
import re
import subprocess
import sys


def main():
    src = sys.argv[1]
    extra = sys.argv[2:]
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage"] + extra
    out = subprocess.run(cmd, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True).stderr
    cur = None
    rows = []
    for line in out.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            if "error" in line:
                print(line)
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            name = t.split(":", 1)[1].strip()
            dem = subprocess.run(["c++filt", name], stdout=subprocess.PIPE, text=True).stdout.strip()
            dem = dem.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
            cur = {"name": dem}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    for r in rows:
        print(f"{r['name']:58s} VGPR {r.get('VGPRs', '?'):>4s} AGPR {r.get('AGPRs', '?'):>3s} spill {r.get('VGPRs Spill', '?'):>3s}"
              f" scratch {r.get('ScratchSize [bytes/lane]', '?'):>4s} SGPR {r.get('SGPRs', '?'):>4s} LDS {r.get('LDS Size [bytes/block]', '?'):>7s}"
              f" occ {r.get('Occupancy [waves/SIMD]', '?')}")


if __name__ == "__main__":
    main()
