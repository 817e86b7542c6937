#!/bin/bash
# round 4, call 33: epilogue operands of the 32-channel work-groups requested behind the first channel block (WC_OPREF)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04ab; mkdir -p $O
cd $R
MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_opref.so timeout 600 python -m pytest tests/test_hip_parity.py tests/test_bn_finalize_gpu.py tests/test_model_golden.py -m gpu -x -q > $O/pytest_opref.log 2>&1; tail -2 $O/pytest_opref.log
for v in base opref base2 opref2; do
  L="A=1"; [ ${v:0:4} != base ] && L="MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_opref.so"
  env $L timeout 600 python tools/bench_kernels.py --filter "128" --iters 60 > $O/microbench_$v.txt 2>&1
  echo "== $v"; grep -h "adapter\|dgrad" $O/microbench_$v.txt | grep -v wgrad | cut -c1-58
  for r in 1 2; do
  env $L timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 > $O/bench_${v}_$r.json 2> $O/bench_$v.err
  echo $v $r $(python -c "import json,sys; d=json.loads(open('$O/bench_${v}_$r.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['final_total_loss'])")
  done
done
