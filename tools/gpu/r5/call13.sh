#!/bin/bash
# round 5, call 13: w4conv refill loads spread behind the MFMAs (two blocks of lead): parity, micro-benchmarks, step A/B, stamps
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05m; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_bn_finalize_gpu.py -m gpu -x -q > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
for v in spread nospread; do
  L=""; [ $v = nospread ] && L="MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_w4nospread.so"
  (env $L timeout 600 python tools/bench_kernels.py --filter conv --iters 40; env $L timeout 600 python tools/bench_kernels.py --filter dgrad --iters 40) 2>&1 | grep "128\|conv64 \|tapconv64 3x1\|tapconv64 1x3 d1 bias" | grep -v "tapconv16" | cut -c1-60 > $O/kb_$v.txt; echo "== $v"; cat $O/kb_$v.txt
done
b() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 > $O/b_$name.json 2> $O/b_$name.err; echo "$name $(python -c "import json; d=json.loads(open('$O/b_$name.json').read().strip().splitlines()[-1]); print('%.1f img/s  %.3f ms/step' % (d['value'], d['ms_per_step']))" 2>&1 | tail -1)"; }
for r in 1 2; do b spread_$r X=1; b nospread_$r MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_w4nospread.so; done
MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_w4timing.so timeout 300 python tools/probes/w4conv_stamp_probe.py 2>&1 | grep -v amdgpu.ids > $O/stamps.txt; grep "^C=\|loop: cycles\|kernel end" $O/stamps.txt
