#!/bin/bash
# round 4, call 1: the rewritten wconv epilogue (per-lane statistics, reduce-scatter BN reductions,
# one-round tail operands, shift-based pair map) against round 3's library in one call
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04a; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
for v in new r3; do
  L=""; [ $v = r3 ] && L="MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_r3.so"
  env $L timeout 600 python tools/bench_kernels.py --filter "d" > $O/microbench_$v.txt 2>&1
  env $L timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 > $O/bench_$v.json 2> $O/bench_$v.err
  env $L timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --profile-steps 0 --single-stream > $O/bench1s_$v.json 2>> $O/bench_$v.err
done
env timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 > $O/bench_new2.json 2>> $O/bench_new.err
grep -h "conv\|dgrad" $O/microbench_new.txt | cut -c1-110 > $O/mb_new.txt
grep -h "conv\|dgrad" $O/microbench_r3.txt | cut -c1-110 > $O/mb_r3.txt
paste -d'|' <(cut -c1-58 $O/mb_new.txt) <(cut -c46-58 $O/mb_r3.txt)
for f in $O/bench_*.json $O/bench1s_*.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
