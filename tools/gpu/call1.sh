#!/bin/bash
# round 3, GPU call 1: smoke, wconv store-form variants (time + WRITE_SIZE), GPU test suite, HIP mIoU samples
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03a; mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/summary.txt
for v in base s1 s2 s3 s4; do
  if [ $v = base ]; then unset MDIL_HIP_LIB; else export MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_$v.so; fi
  timeout 300 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "nb_block or winograd or three_tap" > $O/parity_$v.log 2>&1; echo "$v parity rc $? $(tail -1 $O/parity_$v.log)" >> $O/summary.txt
  timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err; echo "$v bench $(cut -c1-200 $O/bench_$v.json)" >> $O/summary.txt
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$v -- python $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 --profile-steps 0 --single-stream > /dev/null 2>&1)
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmcw_$v -- python $R/tools/bench_kernels.py --filter "tapconv" --iters 3 > /dev/null 2>&1)
  python - <<PY >> $O/summary.txt 2>&1
import csv, glob, collections
f = sorted(glob.glob("$O/stats_$v/*/*_kernel_stats.csv"))
if f:
    rows = [r for r in csv.DictReader(open(f[-1])) if "wconv" in r["Name"]]
    for r in rows: print("$v", r["Name"][:70], r["Calls"], r["AverageNs"])
    tot = sum(float(r["TotalDurationNs"]) for r in csv.DictReader(open(f[-1])))
    print("$v total kernel ms over 6 steps", tot / 1e6)
f = sorted(glob.glob("$O/pmcw_$v/*/*_counter_collection.csv"))
if f:
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f[-1])):
        if "conv" in r["Kernel_Name"]:
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    for k, (s_, n) in sorted(agg.items()): print("$v WRITE_SIZE", k, n, round(s_ / n, 1))
PY
done
unset MDIL_HIP_LIB
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_miou_parity.py > $O/pytest_gpu.log 2>&1; echo "pytest rc $? $(tail -1 $O/pytest_gpu.log)" >> $O/summary.txt
timeout 900 python tools/miou_hip_sample.py --one 0 --checks --out $O/miou_hip > $O/miou_checks.log 2>&1; echo "miou checks rc $? $(grep -c 'one-step parity' $O/miou_checks.log)" >> $O/summary.txt
timeout 2400 python tools/miou_hip_sample.py --seeds 3001-3036 --procs 4 --out $O/miou_hip > $O/miou_pool.log 2>&1; echo "miou pool rc $?" >> $O/summary.txt
cat $O/summary.txt
