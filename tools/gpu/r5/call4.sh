#!/bin/bash
# round 5, call 4: chunked epilogue: parity subset + micro-benchmarks of the epilogue-operand forms
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05d; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_bn_finalize_gpu.py -m gpu -x -q > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
timeout 600 python tools/bench_kernels.py --filter dgrad --iters 40 > $O/kb_w4.txt 2>&1; cat $O/kb_w4.txt
MDIL_NO_W4CONV=1 timeout 600 python tools/bench_kernels.py --filter dgrad --iters 40 > $O/kb_w2.txt 2>&1; cat $O/kb_w2.txt
