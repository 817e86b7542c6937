#!/bin/bash
# round 4, call 19d: async_wgrad after the join in engine._backward
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04s; mkdir -p $O
cd $R
for k in off 8; do MDIL_STAGGER=$k timeout 600 python tools/gpu/r4_dbg_async2.py 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-400 | tee $O/dbg3_$k.txt; done
for r in 1 2; do timeout 600 python -m pytest tests/test_model_golden.py tests/test_dp_gpu.py tests/test_gradient_adjudication.py -m gpu -q > $O/pytestc_$r.log 2>&1; echo "run $r: $(tail -1 $O/pytestc_$r.log)"; grep "AssertionError:\|^FAILED" $O/pytestc_$r.log | head -4; done
