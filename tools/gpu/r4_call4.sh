#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04d; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_miou_parity.py -m gpu -q -s > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
grep -E "passed|failed|error|covering|one-step parity|Error" $O/pytest_gpu.log | tail -30
