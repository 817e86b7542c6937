#!/bin/bash
# round 4, call 10 (re-entry): full GPU suite, default bench line, rocprofv3 stats (single / three streams)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04j; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=short > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
timeout 400 python bench.py > $O/bench_step2.json 2> $O/bench_step2.err; tail -1 $O/bench_step2.json
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_single -- $B --steps 4 --warmup 1 --profile-steps 0 --single-stream > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_3streams -- $B --steps 4 --warmup 1 --profile-steps 0 > /dev/null 2>&1
cd $R
for t in single 3streams; do s=$(find $O/stats_$t -name "*kernel_stats.csv" | head -1); cp $s $O/kernel_stats_$t.csv; rm -rf $O/stats_$t; done
head -30 $O/kernel_stats_single.csv | cut -c1-160
