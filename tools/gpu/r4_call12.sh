#!/bin/bash
# round 4, call 12: three waves per SIMD for the wconv variants whose registers allow it; 32 partial rows in
# flight per thread in the fused BatchNorm finalize
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04l; mkdir -p $O
cd $R
MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_nw12.so timeout 600 python -m pytest tests/test_hip_parity.py tests/test_bn_finalize_gpu.py -m gpu -x -q > $O/pytest_nw12.log 2>&1; tail -2 $O/pytest_nw12.log
MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_fin32.so timeout 600 python -m pytest tests/test_hip_parity.py tests/test_bn_finalize_gpu.py -m gpu -x -q > $O/pytest_fin32.log 2>&1; tail -2 $O/pytest_fin32.log
VS="base nw12c nw12p nw12 fin32 base2"
for v in $VS; do
  L="A=1"; [ ${v:0:4} != base ] && L="MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_$v.so"
  env $L timeout 600 python tools/bench_kernels.py --filter "conv" > $O/microbench_$v.txt 2>&1
  for r in 1 2; do
  env $L timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 > $O/bench_${v}_$r.json 2> $O/bench_$v.err
  echo $v $r $(python -c "import json,sys; d=json.loads(open('$O/bench_${v}_$r.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  done
done
for v in $VS; do grep -h "conv\|dgrad" $O/microbench_$v.txt | grep -v "16 \|unfused" | cut -c1-58 > $O/mb_$v.txt; wc -l $O/mb_$v.txt; done
paste -d'|' $O/mb_base.txt <(cut -c46-58 $O/mb_nw12c.txt) <(cut -c46-58 $O/mb_nw12p.txt) <(cut -c46-58 $O/mb_nw12.txt) <(cut -c46-58 $O/mb_fin32.txt) <(cut -c46-58 $O/mb_base2.txt) | tee $O/microbench_table.txt
