#!/bin/bash
# round 4, call 11: wconv main-loop variants (transform spread between the MFMAs, no SLP packing, static
# priority for the younger half), same-box A/B; timeline of the shipped 3-stream step
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04k; mkdir -p $O
cd $R
MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_spreadnoslp.so timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q > $O/pytest_spread.log 2>&1; tail -2 $O/pytest_spread.log
VS="base spread noslp spreadnoslp yprio base2"
for v in $VS; do
  L="A=1"; [ ${v:0:4} != base ] && L="MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_$v.so"
  env $L timeout 600 python tools/bench_kernels.py --filter "conv" > $O/microbench_$v.txt 2>&1
  for r in 1 2; do
  env $L timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 > $O/bench_${v}_$r.json 2> $O/bench_$v.err
  echo $v $r $(python -c "import json,sys; d=json.loads(open('$O/bench_${v}_$r.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  done
done
for v in $VS; do grep -h "conv\|dgrad" $O/microbench_$v.txt | grep -v "16 \|unfused" | cut -c1-58 > $O/mb_$v.txt; done
paste -d'|' $O/mb_base.txt <(cut -c46-58 $O/mb_spread.txt) <(cut -c46-58 $O/mb_noslp.txt) <(cut -c46-58 $O/mb_spreadnoslp.txt) <(cut -c46-58 $O/mb_yprio.txt) <(cut -c46-58 $O/mb_base2.txt) | tee $O/microbench_table.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace3 -- python $R/bench.py --no-cpu-baseline --steps 6 --warmup 2 --profile-steps 0 > /dev/null 2>&1
cd $R
f=$(find $O/trace3 -name "*kernel_trace.csv" | head -1)
python tools/timeline.py $f > $O/timeline_trace3.txt 2>&1; cat $O/timeline_trace3.txt | head -70
gzip -c $f > $O/kernel_trace_3streams.csv.gz; rm -rf $O/trace3
