#!/bin/bash
# round 4, call 17: async_wgrad under the staggered schedule (diagnostic); kernel trace of the staggered step
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04q; mkdir -p $O
cd $R
timeout 600 python tools/gpu/r4_dbg_async.py 2>&1 | grep -v amdgpu.ids | tee $O/dbg_async.txt
cd /tmp && export TMPDIR=/tmp
for k in 8 off; do
MDIL_STAGGER=$k timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$k -- python $R/bench.py --no-cpu-baseline --steps 6 --warmup 2 --profile-steps 0 --pipeline-teacher > /dev/null 2>&1
f=$(find $O/trace_$k -name "*kernel_trace.csv" | head -1)
python $R/tools/timeline.py $f > $O/timeline_$k.txt 2>&1; head -8 $O/timeline_$k.txt
gzip -c $f > $O/kernel_trace_stagger_$k.csv.gz; rm -rf $O/trace_$k
done
