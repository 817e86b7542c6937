#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04h; mkdir -p $O
cd $R
timeout 840 python -m pytest tests/test_dp_gpu.py -m gpu -x -q -s -k two_ranks > $O/pytest_dp2.log 2>&1; echo "pytest exit $?" >> $O/pytest_dp2.log
tail -40 $O/pytest_dp2.log
