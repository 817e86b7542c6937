#!/bin/bash
# round 4, call 21: full GPU suite with the staggered schedule as default
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04u; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --tb=short > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_gpu.log; grep "^FAILED\|^ERROR" $O/pytest_gpu.log | head
