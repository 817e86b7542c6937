#!/bin/bash
# round 4, call 28: ablation of the epilogue fusions UNDER THE STAGGERED SCHEDULE (HBM-bound passes are now largely
# covered by the other graph's MFMA phase: does moving them into conv epilogues still pay?)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04z; mkdir -p $O
cd $R
S=$O/ablation.txt
b() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 > $O/b_$name.json 2> /dev/null; echo "$(printf '%-62s' "$name") $(python -c "import json; d=json.loads(open('$O/b_$name.json').read().strip().splitlines()[-1]); print('%.1f img/s  %.3f ms/step' % (d['value'], d['ms_per_step']))")" >> $S; }
echo "# bench.py --steps 60 --warmup 15 (one MI355X, one call, same box), staggered schedule unless noted" > $S
b shipped_build A=1
b no_block_boundary_bn_fusion MDIL_NO_BNTAIL=1
b no_bn_fusion_at_all_MDIL_NO_BNFUSE MDIL_NO_BNFUSE=1
b no_fused_finalize MDIL_NO_BNFIN=1
b no_fused_head MDIL_NO_HEADFUSE=1
b no_wgrad16 MDIL_NO_WGRAD16=1
b lock_step MDIL_STAGGER=off
b lock_step_no_block_boundary_bn_fusion MDIL_STAGGER=off MDIL_NO_BNTAIL=1
b shipped_build_again A=1
cat $S
