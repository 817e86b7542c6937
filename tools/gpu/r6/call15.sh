#!/bin/bash
# round 6, call 15: a kernel trace of the three-stream step that is NOT host-bound under the profiler: hipGraph replay
# (host cost ~0), then tools/timeline.py: which kernel families really run alone?
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r06j; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_graph -- python $R/bench.py --no-cpu-baseline --steps 8 --warmup 3 --profile-steps 0 --graph > $O/bench_graph.log 2>&1
tail -1 $O/bench_graph.log | cut -c1-200
cd $R
t=$(find $O/trace_graph -name "*kernel_trace.csv" | head -1); ls -la $t
python tools/timeline.py $t --skip 4 2>&1 | tee $O/timeline_graph_replay.txt | head -40
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete 2>/dev/null
