#!/bin/bash
# round 6, call 8: first tile's operand requests BEFORE the weight fill (-DW4_EARLY=1 -DWC_EARLY=1): stamps, launch times, step A/B
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r06g; mkdir -p $O
cd $R
V=$R/gpurun_tmp/libmdil_w4early.so
MDIL_HIP_LIB=$V timeout 600 python -m pytest tests/test_hip_parity.py tests/test_bn_finalize_gpu.py -m gpu -q -x 2>&1 | tail -2
echo "== stamps, early build"; MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_w4earlyt.so timeout 300 python tools/probes/w4conv_stamp_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/w4conv_stamps_early.txt | grep -E "^C=|resident|tile 0: mfma|kernel end"
echo "== launch times"
timeout 600 python tools/probes/wconv_fit.py 2>&1 | grep -v amdgpu.ids | sed 's/^/shipped /' | cut -c1-175 | tee $O/wconv_fit.txt
MDIL_HIP_LIB=$V timeout 600 python tools/probes/wconv_fit.py 2>&1 | grep -v amdgpu.ids | sed 's/^/early   /' | cut -c1-175 | tee -a $O/wconv_fit.txt
echo "== step A/B"
for r in 1 2 3; do
  timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shipped %.1f img/s %.3f ms' % (d['value'], d['ms_per_step']))"
  MDIL_HIP_LIB=$V timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('early   %.1f img/s %.3f ms' % (d['value'], d['ms_per_step']))"
done | tee $O/bench_ab.txt
