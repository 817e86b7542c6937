#!/bin/bash
# round 6, call 3: r4conv (C = 128 3-tap convs with the weights resident in registers): bit-identity with w4conv, parity
# subset, launch times against batch size, step A/B; + the golden / trajectory tests of call 2 that did not get to run
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r06c; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_r4conv_gpu.py -m gpu -q -x > $O/pytest_r4.log 2>&1; tail -15 $O/pytest_r4.log | cut -c1-300
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_covering_trajectory.py tests/test_model_golden.py -m gpu -q -x -s > $O/pytest_parity.log 2>&1; grep -a "covering-size\|relu gates\|gradients vs\|eval logits\|passed\|failed\|Error" $O/pytest_parity.log | tail -14 | cut -c1-330
echo "== launch time against batch size: r4conv (shipped) / w4conv (MDIL_NO_R4CONV=1)"
timeout 600 python tools/probes/wconv_fit.py 2>&1 | grep -v amdgpu.ids | grep "C=128" | sed 's/^/r4conv  /' | tee $O/wconv_fit.txt
MDIL_NO_R4CONV=1 timeout 600 python tools/probes/wconv_fit.py 2>&1 | grep -v amdgpu.ids | grep "C=128" | sed 's/^/w4conv  /' | tee -a $O/wconv_fit.txt
echo "== step A/B"
for r in 1 2 3; do
  timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r4conv (shipped)        %.1f img/s %.3f ms' % (d['value'], d['ms_per_step']))"
  MDIL_NO_R4CONV=1 timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('w4conv (MDIL_NO_R4CONV) %.1f img/s %.3f ms' % (d['value'], d['ms_per_step']))"
done | tee $O/bench_ab.txt
