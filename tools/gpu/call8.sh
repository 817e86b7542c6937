#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03g; mkdir -p $O
cd $R
S=$O/summary.txt
timeout 900 python -m pytest tests/test_multi_task_gpu.py tests/test_input_pipeline.py tests/test_step3_gpu.py tests/test_trainer_gpu.py tests/test_ft_baselines_gpu.py -m gpu -q --tb=short > $O/pytest.log 2>&1; echo "pytest rc $? $(grep -E ' passed| failed' $O/pytest.log | tail -1 | cut -c1-200)" >> $S
grep -E "^FAILED|^ERROR" $O/pytest.log | cut -c1-250 >> $S
for w in multitask step1; do timeout 300 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_$w.json 2> /dev/null; echo "bench $w $(python -c "import json; d=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")" >> $S; done
cat $S
