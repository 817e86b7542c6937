#!/bin/bash
# round 3, GPU call 7: the complete GPU suite (mIoU parity test included) on the final build + smoke + bench
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03f; mkdir -p $O
cd $R
S=$O/summary.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=short -s > $O/pytest_gpu.log 2>&1; echo "pytest rc $? $(grep -E ' passed| failed' $O/pytest_gpu.log | tail -1 | cut -c1-200)" >> $S
grep -E "^FAILED|^ERROR" $O/pytest_gpu.log | cut -c1-250 >> $S
grep -E "one-step parity|mIoU new:|mIoU old:|CE curve|HIP eval path|block-boundary|shipped engine path|fp64 adjudication|gate-forced" $O/pytest_gpu.log | cut -c1-330 >> $S
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $? $(tail -1 $O/smoke.log | cut -c1-200)" >> $S
timeout 400 python bench.py > $O/bench_step2.json 2> $O/bench.err; echo "bench $(cut -c1-200 $O/bench_step2.json)" >> $S
for w in step3; do timeout 300 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_$w.json 2> /dev/null; echo "bench $w $(cut -c1-160 $O/bench_$w.json)" >> $S; done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_single -- python $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 --profile-steps 0 --single-stream > /dev/null 2>&1)
python - <<PY >> $S 2>&1
import csv, glob
f = sorted(glob.glob("$O/stats_single/*/*_kernel_stats.csv"))
if f:
    rows = sorted(csv.DictReader(open(f[-1])), key=lambda r: -float(r["TotalDurationNs"]))
    print("total kernel ms/step", sum(float(r["TotalDurationNs"]) for r in rows) / 6e6)
    for r in rows:
        n = r["Name"].replace("(anonymous namespace)::", "")
        if "wgrad16" in n: print(f'  {int(r["Calls"])/6:6.1f} calls avg {float(r["AverageNs"])/1e3:7.1f} us  {n[:70]}')
PY
cat $S | cut -c1-330
