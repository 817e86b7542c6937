#!/bin/bash
# round 3, GPU call 2: new fusions (BN tail, head+loss), staged exchange, loader cache; A/B benches; mIoU checks + sampling probe
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03b; mkdir -p $O
cd $R
export TMPDIR=/tmp
S=$O/summary.txt
timeout 900 python -m pytest tests -m gpu -q --tb=short --deselect tests/test_miou_parity.py > $O/pytest_gpu.log 2>&1; echo "pytest rc $? $(tail -1 $O/pytest_gpu.log | cut -c1-200)" >> $S
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $? $(tail -1 $O/smoke.log | cut -c1-300)" >> $S
b() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err; echo "bench $name $(cut -c1-170 $O/bench_$name.json)" >> $S; }
b default A=1
b nohead MDIL_NO_HEADFUSE=1
b notail MDIL_NO_BNTAIL=1
b neither MDIL_NO_HEADFUSE=1 MDIL_NO_BNTAIL=1
b bnt1024 MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_bnt.so
b default2 A=1
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_single -- python $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 --profile-steps 0 --single-stream > /dev/null 2>&1)
python - <<PY >> $S 2>&1
import csv, glob
f = sorted(glob.glob("$O/stats_single/*/*_kernel_stats.csv"))
if f:
    rows = sorted(csv.DictReader(open(f[-1])), key=lambda r: -float(r["TotalDurationNs"]))
    print("total kernel ms/step", sum(float(r["TotalDurationNs"]) for r in rows) / 6e6)
    for r in rows[:40]:
        print(f'{float(r["TotalDurationNs"])/6e6:7.3f} ms/step {int(r["Calls"])/6:6.1f} calls avg {float(r["AverageNs"])/1e3:7.1f} us  {r["Name"].replace("(anonymous namespace)::","")[:80]}')
PY
timeout 600 python tools/miou_hip_sample.py --one 0 --checks --out $O/miou_hip > $O/miou_checks.log 2>&1; echo "miou checks rc $? $(grep -c 'one-step parity' $O/miou_checks.log) $(tail -1 $O/miou_checks.log | cut -c1-200)" >> $S
t0=$(date +%s); timeout 600 python tools/miou_hip_sample.py --one 3001 --out $O/miou_hip > $O/miou_one.log 2>&1; echo "miou single run rc $? $(( $(date +%s) - t0 )) s $(tail -1 $O/miou_one.log | cut -c1-200)" >> $S
t0=$(date +%s); timeout 900 python tools/miou_hip_sample.py --seeds 3002-3005 --procs 2 --stall 500 --out $O/miou_hip > $O/miou_pool.log 2>&1; echo "miou pool(2) rc $? $(( $(date +%s) - t0 )) s" >> $S; cat $O/miou_pool.log | cut -c1-200 >> $S
timeout 600 python tools/bench_loader.py --workers 4 8 16 --images 48 --cached --device > $O/loader_throughput.txt 2>&1; echo "loader rc $?" >> $S; grep -v "^/tmp" $O/loader_throughput.txt | cut -c1-200 >> $S
timeout 400 python tools/host_contention.py --procs 1 4 8 --reps 100 > $O/host_contention.txt 2>&1; echo "contention rc $?" >> $S; cat $O/host_contention.txt | cut -c1-300 >> $S
cat $S
