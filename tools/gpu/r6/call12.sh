#!/bin/bash
# round 6, call 12: final validation of the round's last source state: the whole GPU suite (mIoU statistic on the 80 samples
# of the build included), smoke(), the default bench line
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r06_final; mkdir -p $O
cd $R
python -c "from tests.helpers import kernel_build_id; print('build', kernel_build_id())"
timeout 3000 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log | cut -c1-200
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -a smoke | cut -c1-300 | tee $O/smoke.txt
timeout 900 python bench.py > $O/bench_step2_default.json 2>/dev/null; python -c "import json; d=json.load(open('$O/bench_step2_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])"
