#!/bin/bash
# round 5, call 14: epilogue chunk of the 3-tap forms with epilogue operands: 2 pixels (187-195 VGPRs) vs 4 (212-236)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05n; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_bn_finalize_gpu.py -m gpu -x -q 2>&1 | tail -1
b() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 $EXTRA > $O/b_$name.json 2> $O/b_$name.err; echo "$name $(python -c "import json; d=json.loads(open('$O/b_$name.json').read().strip().splitlines()[-1]); print('%.1f img/s  %.3f ms/step' % (d['value'], d['ms_per_step']))" 2>&1 | tail -1)"; }
for r in 1 2 3; do b ne2_$r X=1; b ne4_$r MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_w4ne4.so; done
EXTRA="--single-stream" b ne2_single X=1; EXTRA="--single-stream" b ne4_single MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_w4ne4.so
MDIL_STAGGER=off b ne2_lock X=1; MDIL_STAGGER=off b ne4_lock MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_w4ne4.so
