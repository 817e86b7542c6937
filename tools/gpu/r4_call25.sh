#!/bin/bash
# round 4, call 25: what the +8 us of a launch with a fused finalize consist of (timing diagnostic builds:
# 1 = no hand-off, 2 = stores drained + barrier, 3 = + ticket atomic, base = + finalize by the last work-group)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04w; mkdir -p $O
cd $R
for v in base fdiag1 fdiag2 fdiag3 base2; do
  L="A=1"; [ ${v:0:4} != base ] && L="MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_$v.so"
  env $L timeout 600 python tools/bench_kernels.py --filter "s" --iters 60 > $O/microbench_$v.txt 2>&1
  echo "== $v"; grep -h "finalize\|bnred\|tail\|1x3+adapter d\|bn_train_stats\|1x3 d.* bias" $O/microbench_$v.txt | cut -c1-58
done
