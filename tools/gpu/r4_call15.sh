#!/bin/bash
# round 4, call 15: 32 output channels per work-group for EVERY C = 128 Winograd conv (half the LDS fill)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04o; mkdir -p $O
cd $R
MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_cow32.so timeout 600 python -m pytest tests/test_hip_parity.py tests/test_bn_finalize_gpu.py -m gpu -x -q > $O/pytest_cow32.log 2>&1; tail -2 $O/pytest_cow32.log
VS="base cow32 base2 cow32b"
for v in $VS; do
  L="A=1"; [ ${v:0:4} != base ] && L="MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_cow32.so"
  env $L timeout 600 python tools/bench_kernels.py --filter "128" > $O/microbench_$v.txt 2>&1
  for r in 1 2; do
  env $L timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 > $O/bench_${v}_$r.json 2> $O/bench_$v.err
  echo $v $r $(python -c "import json,sys; d=json.loads(open('$O/bench_${v}_$r.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  done
done
for v in $VS; do echo "== $v"; grep -h "conv\|dgrad" $O/microbench_$v.txt | grep -v "unfused" | cut -c1-58; done
python tools/host_cost.py > $O/host_cost.txt 2>&1; tail -15 $O/host_cost.txt
