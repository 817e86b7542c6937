#!/bin/bash
# round 6, call 4: host-side profile of an iteration; the mIoU protocol test with the covering-size trajectories from the
# trained states (the recorded samples are of the previous build: the sample-count assert at its end is expected to fail)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r06d; mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_model_golden.py -m gpu -q -x -s -k "eval_forward" 2>&1 | grep -a "eval logits\|passed\|failed" | cut -c1-250
timeout 300 python tools/host_profile.py > $O/host_profile.txt 2>&1; grep -a -A40 "Ordered by: internal time" $O/host_profile.txt | head -48 | cut -c1-160
timeout 300 python tools/host_cost.py 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/host_cost.txt
timeout 2400 python -m pytest tests/test_miou_parity.py -m gpu -q -x -s > $O/pytest_miou.log 2>&1; grep -a "covering-size\|paired traj\|one-step parity\|mIoU \|build under test\|passed\|failed\|Error" $O/pytest_miou.log | cut -c1-330 | tail -50
