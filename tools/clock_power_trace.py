#!/usr/bin/env python3
"""Samples socket power, shader clock and temperature of GPU 0 (amdsmi) at ~20 Hz while a command runs.

    python tools/clock_power_trace.py --out profiles/r05_trace_bench.csv -- python bench.py --steps 1200 --no-cpu-baseline

Prints a summary (mean / percentiles over the samples taken while the GPU was busy) and writes the
samples as CSV (t_s, power_w, sclk_mhz, mclk_mhz, temp_c, busy_pct)."""
import argparse
import subprocess
import sys
import threading
import time


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--hz", type=float, default=20.0)
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    import amdsmi as S
    S.amdsmi_init()
    h = S.amdsmi_get_processor_handles()[0]
    rows, stop = [], threading.Event()

    def num(v):
        try:
            return float(v)
        except Exception:
            return float("nan")

    def sample():
        t0 = time.time()
        while not stop.is_set():
            r = [time.time() - t0]
            try:
                p = S.amdsmi_get_power_info(h)
                r.append(num(p.get("current_socket_power", p.get("average_socket_power"))))
            except Exception:
                r.append(float("nan"))
            for ct in (S.AmdSmiClkType.GFX, S.AmdSmiClkType.MEM):
                try:
                    r.append(num(S.amdsmi_get_clock_info(h, ct).get("clk")))
                except Exception:
                    r.append(float("nan"))
            try:
                r.append(num(S.amdsmi_get_temp_metric(h, S.AmdSmiTemperatureType.HOTSPOT, S.AmdSmiTemperatureMetric.CURRENT)))
            except Exception:
                r.append(float("nan"))
            try:
                r.append(num(S.amdsmi_get_gpu_activity(h).get("gfx_activity")))
            except Exception:
                r.append(float("nan"))
            rows.append(r)
            time.sleep(max(0.0, 1.0 / a.hz - 0.002))

    th = threading.Thread(target=sample, daemon=True)
    th.start()
    time.sleep(1.0)                      # one second of idle samples first
    t_start = time.time()
    rc = subprocess.call(cmd)
    t_end = time.time()
    time.sleep(0.5)
    stop.set()
    th.join(timeout=2)
    if a.out:
        with open(a.out, "w") as f:
            f.write("t_s,power_w,sclk_mhz,mclk_mhz,temp_hotspot_c,busy_pct\n")
            for r in rows:
                f.write(",".join(f"{v:.3f}" for v in r) + "\n")
    import statistics as st
    busy = [r for r in rows if r[5] == r[5] and r[5] >= 50]
    idle = [r for r in rows if r[5] == r[5] and r[5] < 5]

    def q(v, p):
        v = sorted(x for x in v if x == x)
        return v[min(len(v) - 1, int(p * len(v)))] if v else float("nan")

    print(f"command {' '.join(cmd)!r}: rc {rc}, {t_end - t_start:.1f} s, {len(rows)} samples ({len(busy)} busy >= 50 %, {len(idle)} idle)")
    for name, col in (("power W", 1), ("sclk MHz", 2), ("mclk MHz", 3), ("hotspot C", 4)):
        b = [r[col] for r in busy]
        i = [r[col] for r in idle]
        if b:
            print(f"  {name:10s} busy: mean {st.fmean(x for x in b if x == x):8.1f}  p10 {q(b, .1):8.1f}  median {q(b, .5):8.1f}  p90 {q(b, .9):8.1f}  max {q(b, 1.0):8.1f}"
                  + (f"   idle mean {st.fmean(x for x in i if x == x):8.1f}" if i else ""))
    try:
        cap = S.amdsmi_get_power_cap_info(h)
        print("  power cap info:", {k: cap[k] for k in cap})
    except Exception as e:
        print("  power cap info: n/a", e)
    S.amdsmi_shut_down()
    return rc


if __name__ == "__main__":
    sys.exit(main())
