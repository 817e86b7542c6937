#!/bin/bash
# round 4, call 2: kernel trace of the shipped 3-stream step and of the single-stream step -> timeline
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace3 -- $B --steps 6 --warmup 2 --profile-steps 0 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace1 -- $B --steps 6 --warmup 2 --profile-steps 0 --single-stream > /dev/null 2>&1
cd $R
for t in trace3 trace1; do
  f=$(find $O/$t -name "*kernel_trace.csv" | head -1)
  echo "== $t $f"; python tools/timeline.py $f | tee $O/timeline_$t.txt
  s=$(find $O/$t -name "*kernel_stats.csv" | head -1); cp $s $O/stats_$t.csv
  cp $f $O/kernel_trace_$t.csv
done
rm -rf $O/trace3 $O/trace1
ls -la $O
