#!/bin/bash
# round 5, call 11: prefetch depth 3 for the plain w4conv forms (variant library), stagger sweep, pipelined frozen model
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05k; mkdir -p $O
cd $R
for v in base pd3; do
  L=""; [ $v = pd3 ] && L="MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_w4pd3.so"
  env $L timeout 600 python tools/bench_kernels.py --filter tapconv --iters 40 2>&1 | grep -v "amdgpu.ids\|tapconv16\|adapter" > $O/kb_$v.txt; echo "== $v"; cat $O/kb_$v.txt
done
b() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 $EXTRA > $O/b_$name.json 2> $O/b_$name.err; echo "$name $(python -c "import json; d=json.loads(open('$O/b_$name.json').read().strip().splitlines()[-1]); print('%.1f img/s  %.3f ms/step' % (d['value'], d['ms_per_step']))" 2>&1 | tail -1)"; }
for r in 1 2; do
  b base_$r X=1
  b pd3_$r MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_w4pd3.so
  for k in 4 6 10 12; do b stagger${k}_$r MDIL_STAGGER=$k; done
  EXTRA="--pipeline-teacher" b pipe_$r X=1
done
