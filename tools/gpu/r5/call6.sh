#!/bin/bash
# round 5, call 6: software-pipelined w4conv main loop (scalar fillers behind every MFMA): parity, micro-benchmarks, stamps
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05f; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_bn_finalize_gpu.py -m gpu -x -q > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
timeout 300 python tools/conv_accuracy.py 2>&1 | grep -v amdgpu.ids
timeout 600 python tools/bench_kernels.py --filter conv --iters 40 2>&1 | grep -v "amdgpu.ids\|tapconv16\|unfused" > $O/kb_w4.txt; cat $O/kb_w4.txt
timeout 600 python tools/bench_kernels.py --filter dgrad --iters 40 2>&1 | grep -v "amdgpu.ids\|tapconv16\|unfused" >> $O/kb_w4.txt; tail -9 $O/kb_w4.txt
MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_w4timing.so timeout 300 python tools/probes/w4conv_stamp_probe.py > $O/stamps.txt 2>&1; cat $O/stamps.txt
