#!/bin/bash
# round 4, call 9: matrix-pipe grant between the two waves of a SIMD (WC_GRANT experiment)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04i; mkdir -p $O
cd $R
MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_grant.so timeout 600 python -m pytest tests/test_hip_parity.py tests/test_bn_finalize_gpu.py -m gpu -x -q > $O/pytest_grant.log 2>&1; tail -3 $O/pytest_grant.log
for v in base grant base2 grant2; do
  L="A=1"; [ ${v:0:5} = grant ] && L="MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_grant.so"
  env $L timeout 600 python tools/bench_kernels.py --filter "conv" > $O/microbench_$v.txt 2>&1
  env $L timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 > $O/bench_$v.json 2> $O/bench_$v.err
  echo $v $(python -c "import json,sys; d=json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
done
for v in base grant base2 grant2; do grep -h "conv\|dgrad" $O/microbench_$v.txt | grep -v "16 \|unfused" | cut -c1-58 > $O/mb_$v.txt; done
paste -d'|' $O/mb_base.txt <(cut -c46-58 $O/mb_grant.txt) <(cut -c46-58 $O/mb_base2.txt) <(cut -c46-58 $O/mb_grant2.txt)
