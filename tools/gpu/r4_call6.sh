#!/bin/bash
# round 4, call 6: prefetch distance of wconv per variant (PD 3 where the registers allow)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04f; mkdir -p $O
cd $R
for v in base pd3 pd3n pd3c; do
  L="A=1"; [ $v != base ] && L="MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_$v.so"
  env $L timeout 600 python tools/bench_kernels.py --filter "conv" > $O/microbench_$v.txt 2>&1
  for r in 1 2; do
  env $L timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 > $O/bench_${v}_$r.json 2> $O/bench_$v.err
  echo $v $r $(python -c "import json,sys; d=json.loads(open('$O/bench_${v}_$r.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
  done
done
for v in base pd3 pd3n pd3c; do grep -h "conv\|dgrad" $O/microbench_$v.txt | grep -v "16 \|unfused" | cut -c1-58 > $O/mb_$v.txt; done
paste -d'|' $O/mb_base.txt <(cut -c46-58 $O/mb_pd3.txt) <(cut -c46-58 $O/mb_pd3n.txt) <(cut -c46-58 $O/mb_pd3c.txt)
