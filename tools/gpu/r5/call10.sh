#!/bin/bash
# round 5, call 10: wgradw / wgradx with ONE VALU block per quad: parity, micro-benchmarks, step, in-step stats
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05j; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_dp_gpu.py -m gpu -x -q > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
timeout 300 python tools/wgrad_accuracy.py 2>&1 | grep -v amdgpu.ids | tail -8
timeout 600 python tools/bench_kernels.py --filter wgrad --iters 40 2>&1 | grep -v "amdgpu.ids" > $O/kb_wgrad.txt; cat $O/kb_wgrad.txt
for r in 1 2; do timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f img/s %.3f ms' % (d['value'], d['ms_per_step']))"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_single -- python $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 --profile-steps 0 --single-stream > /dev/null 2>&1
cd $R
f=$(ls $O/stats_single/*/*_kernel_stats.csv | head -1); cp $f $O/kstats_single.csv; rm -rf $O/stats_single
grep -a "wgrad" $O/kstats_single.csv | cut -c1-200
