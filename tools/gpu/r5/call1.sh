#!/bin/bash
# round 5, call 1: first F(4,3) kernels: accuracy, parity subset, micro-benchmarks A/B
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05a; mkdir -p $O
cd $R
for v in "" "MDIL_NO_W4CONV=1" "MDIL_NO_W4CONV=1 MDIL_NO_WCONV=1"; do env $v timeout 300 python tools/conv_accuracy.py; done > $O/conv_accuracy.txt 2>&1; cat $O/conv_accuracy.txt
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_bn_finalize_gpu.py -m gpu -x -q > $O/pytest_parity.log 2>&1; tail -5 $O/pytest_parity.log
timeout 600 python tools/bench_kernels.py --filter conv --iters 40 > $O/kb_w4.txt 2>&1; cat $O/kb_w4.txt
timeout 600 python tools/bench_kernels.py --filter dgrad --iters 40 >> $O/kb_w4.txt 2>&1; tail -12 $O/kb_w4.txt
MDIL_NO_W4CONV=1 timeout 600 python tools/bench_kernels.py --filter conv --iters 40 > $O/kb_w2.txt 2>&1; cat $O/kb_w2.txt
