#!/bin/bash
# round 5, call 7: w4conv (clustered packed transform without operand negations, C = 64 plain form at 64 channels per
# work-group, chunked epilogues): parity subset, step A/B, per-kernel stats in the step
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05g; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_bn_finalize_gpu.py -m gpu -x -q > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
b() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 > $O/b_$name.json 2> $O/b_$name.err; echo "$name $(python -c "import json; d=json.loads(open('$O/b_$name.json').read().strip().splitlines()[-1]); print('%.1f img/s  %.3f ms/step' % (d['value'], d['ms_per_step']))" 2>&1 | tail -1)"; }
for r in 1 2; do b w4_$r X=1; b w2_$r MDIL_NO_W4CONV=1; done
b w4_single X=1 2>/dev/null
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_single_w4 -- $B --steps 4 --warmup 1 --profile-steps 0 --single-stream > /dev/null 2>&1
cd $R
f=$(ls $O/stats_single_w4/*/*_kernel_stats.csv | head -1); cp $f $O/kstats_single_w4.csv; rm -rf $O/stats_single_w4
