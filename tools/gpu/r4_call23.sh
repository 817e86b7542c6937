#!/bin/bash
# round 4, call 23: mIoU parity test with the re-sampled golden
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04v; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_miou_parity.py -m gpu -q -s > $O/pytest_miou.log 2>&1; echo "pytest rc $?"; grep -E "mIoU (new|old)|build under test|passed|failed|curve" $O/pytest_miou.log | cut -c1-420
