#!/bin/bash
# round 6, call 2: new parity tests (covering-size free-gate trajectories, fixed-tolerance golden gradients, eval logits,
# TensorBoard scalars of every trainer), w4conv fill stamps, release/acquire ticket A/B, register-resident-weights probe
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r06b; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_covering_trajectory.py tests/test_model_golden.py -m gpu -q -x -s -k "trajectory or golden" > $O/pytest_new.log 2>&1; grep -a "covering-size\|relu gates\|gradients vs\|eval logits\|passed\|failed\|Error\|assert" $O/pytest_new.log | tail -30
timeout 900 python -m pytest tests/test_trainer_gpu.py tests/test_step3_gpu.py tests/test_multi_task_gpu.py tests/test_ft_baselines_gpu.py tests/test_bn_finalize_gpu.py tests/test_gradient_adjudication.py -m gpu -q -x -k "trainer or chain or finalize or second_consumer" > $O/pytest_trainers.log 2>&1; tail -5 $O/pytest_trainers.log
echo "== w4conv fill stamps"; MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_w4timing.so timeout 300 python tools/probes/w4conv_stamp_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/w4conv_stamps.txt | grep -E "^C=|fill|resident|kernel end"
echo "== ticket: relaxed sc1 form vs release / acquire"
for r in 1 2; do
  timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shipped (sc1 stores + sc1 loads) %.1f img/s %.3f ms' % (d['value'], d['ms_per_step']))"
  MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_acqrel.so timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('release / acquire            %.1f img/s %.3f ms' % (d['value'], d['ms_per_step']))"
done | tee $O/acqrel.txt
timeout 300 python tools/bench_kernels.py --filter "finalize" --iters 40 2>&1 | grep -v amdgpu.ids | sed 's/^/shipped  /' | tee -a $O/acqrel.txt
MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_acqrel.so timeout 300 python tools/bench_kernels.py --filter "finalize" --iters 40 2>&1 | grep -v amdgpu.ids | sed 's/^/acq-rel  /' | tee -a $O/acqrel.txt
MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_acqrel.so timeout 600 python -m pytest tests/test_bn_finalize_gpu.py -m gpu -q 2>&1 | tail -2 | tee -a $O/acqrel.txt
echo "== register-resident weights probe"
for c in 0_0 1_5 2_5 1_4; do echo "-- RW_LDS_BLOCKS / RW_AGPR_BLOCKS = $c"; timeout 120 $R/gpurun_tmp/rwp_$c 3; timeout 120 $R/gpurun_tmp/rwp_$c 6 | tail -4; done 2>&1 | tee $O/regweights_probe.txt
