#!/bin/bash
# round 6, call 13: C = 64 + adapter in F(4,3) at 64 channels per work-group, adapter as a second pass (AD2; MDIL_W4_AD2=1):
# parity subset, launch times, step A/B against the shipped F(2,3) forms (same library, switch off)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r06i; mkdir -p $O
cd $R
export MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_ad2.so
MDIL_W4_AD2=1 timeout 900 python -m pytest tests/test_hip_parity.py tests/test_bn_finalize_gpu.py tests/test_gradient_adjudication.py -m gpu -q -x 2>&1 | tail -3 | cut -c1-250
echo "== launch times (C = 64 rows)"
timeout 600 python tools/probes/wconv_fit.py 2>&1 | grep -v amdgpu.ids | grep "C= 64" | sed 's/^/F(2,3) wconv  /' | cut -c1-150 | tee $O/wconv_fit.txt
MDIL_W4_AD2=1 timeout 600 python tools/probes/wconv_fit.py 2>&1 | grep -v amdgpu.ids | grep "C= 64" | sed 's/^/F(4,3) AD2    /' | cut -c1-150 | tee -a $O/wconv_fit.txt
echo "== step A/B"
for r in 1 2 3; do
  timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('shipped forms %.1f img/s %.3f ms' % (d['value'], d['ms_per_step']))"
  MDIL_W4_AD2=1 timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('AD2           %.1f img/s %.3f ms' % (d['value'], d['ms_per_step']))"
done | tee $O/bench_ab.txt
