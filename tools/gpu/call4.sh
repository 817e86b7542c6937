#!/bin/bash
# round 3, GPU call 4: balanced-class MFMA head, staged wgrad16; then the HIP mIoU sample
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03d; mkdir -p $O
cd $R
export TMPDIR=/tmp
S=$O/summary.txt
timeout 900 python -m pytest tests -m gpu -q --tb=short --deselect tests/test_miou_parity.py > $O/pytest_gpu.log 2>&1; prc=$?; echo "pytest rc $prc $(grep -E 'passed|failed' $O/pytest_gpu.log | tail -1 | cut -c1-200)" >> $S
grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | cut -c1-200 >> $S
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $? $(tail -1 $O/smoke.log | cut -c1-200)" >> $S
b() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name $(python -c "import json,sys; d=json.loads(open('$O/bench_$name.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel'], r['avg_launch_us'], r['frac'], r['alg_equiv_frac'])" 2>&1 | tail -1)" >> $S; }
b default A=1
b nohead MDIL_NO_HEADFUSE=1
b nowgrad16 MDIL_NO_WGRAD16=1
b default2 A=1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_single -- python $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 --profile-steps 0 --single-stream > /dev/null 2>&1)
python - <<PY >> $S 2>&1
import csv, glob
f = sorted(glob.glob("$O/stats_single/*/*_kernel_stats.csv"))
if f:
    rows = sorted(csv.DictReader(open(f[-1])), key=lambda r: -float(r["TotalDurationNs"]))
    print("total kernel ms/step", sum(float(r["TotalDurationNs"]) for r in rows) / 6e6)
    for r in rows:
        n = r["Name"].replace("(anonymous namespace)::", "")
        if "head" in n or "wgrad16" in n or "wgrad_kernel<16" in n:
            print(f'  {float(r["TotalDurationNs"])/6e6:7.3f} ms/step {int(r["Calls"])/6:6.1f} calls avg {float(r["AverageNs"])/1e3:7.1f} us  {n[:70]}')
PY
if [ $prc -eq 0 ]; then
  t0=$(date +%s); timeout 2000 python tools/miou_hip_sample.py --seeds 3006-3036 --procs 2 --stall 500 --out $O/miou_hip > $O/miou_pool.log 2>&1; echo "miou pool(2) rc $? $(( $(date +%s) - t0 )) s" >> $S; grep -c SAMPLE $O/miou_pool.log >> $S
fi
cat $S | cut -c1-220
