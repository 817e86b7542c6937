#!/bin/bash
# round 6, call 11: the round's rocprofv3 / bench evidence (tools/collect_profiles.sh, TAG=r06) + the three-stream timeline +
# host cost of an iteration, eager and hipGraph replay
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
export TAG=r06
O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
python -c "from tests.helpers import kernel_build_id; print('build', kernel_build_id())" | tee $O/build_id.txt
bash tools/collect_profiles.sh > $O/collect.log 2>&1
t=$(find $O/stats_3streams -name "*kernel_trace.csv" | head -1); python tools/timeline.py $t --skip 2 > $O/timeline_3streams.txt 2>&1; head -8 $O/timeline_3streams.txt
t=$(find $O/stats_single -name "*kernel_trace.csv" | head -1); python tools/timeline.py $t --skip 2 > $O/timeline_single_stream.txt 2>&1
(timeout 300 python tools/host_cost.py; timeout 300 python tools/host_cost.py --graph) 2>&1 | grep -v amdgpu.ids | grep "host enqueue" | tee $O/host_cost.txt
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete 2>/dev/null
python -c "import json; d=json.load(open('$O/bench_step2.json')); print(d['value'], d['ms_per_step'], d['roofline'], d['cpu_baseline']['value'])"
du -sh $O
