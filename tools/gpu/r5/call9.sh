#!/bin/bash
# round 5, call 9: golden test with gate-flip detection, second-consumer test with sinks, new small kernels, full suite
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05i; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_model_golden.py tests/test_gradient_adjudication.py tests/test_hip_parity.py -m gpu -q -x -s -k "golden or second_consumer or nchw" > $O/pytest_sel.log 2>&1; grep -a "relu gates\|passed\|failed\|Error" $O/pytest_sel.log | tail -12
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_miou_parity.py > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
for r in 1 2; do timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f img/s %.3f ms' % (d['value'], d['ms_per_step']))"; done
