#!/bin/bash
# round 5, call 3: per-kernel stats in the step (single stream / three streams) with F(4,3) on and off
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_single_w4 -- $B --steps 4 --warmup 1 --profile-steps 0 --single-stream > /dev/null 2>&1
MDIL_NO_W4CONV=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_single_w2 -- $B --steps 4 --warmup 1 --profile-steps 0 --single-stream > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_3s_w4 -- $B --steps 4 --warmup 1 --profile-steps 0 > /dev/null 2>&1
MDIL_NO_W4CONV=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_3s_w2 -- $B --steps 4 --warmup 1 --profile-steps 0 > /dev/null 2>&1
cd $R
for v in w4 w2; do for s in single 3s; do f=$(ls $O/stats_${s}_$v/*/*_kernel_stats.csv | head -1); cp $f $O/kstats_${s}_$v.csv; done; done
rm -rf $O/stats_*
b() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --profile-steps 0 --single-stream > $O/b_$name.json 2> $O/b_$name.err; echo "$name $(python -c "import json; d=json.loads(open('$O/b_$name.json').read().strip().splitlines()[-1]); print('%.1f img/s  %.3f ms/step' % (d['value'], d['ms_per_step']))" 2>&1 | tail -1)"; }
b single_w4 X=1; b single_w2 MDIL_NO_W4CONV=1
