#!/bin/bash
# round-3 ablation in ONE call: python bench.py --steps 60 --warmup 15 --no-cpu-baseline, each switch off in turn
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03h; mkdir -p $O
cd $R
S=$O/ablation.txt
b() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 > $O/b_$name.json 2> /dev/null; echo "$(printf '%-62s' "$name") $(python -c "import json; d=json.loads(open('$O/b_$name.json').read().strip().splitlines()[-1]); print('%.1f img/s  %.3f ms/step' % (d['value'], d['ms_per_step']))")" >> $S; }
echo "# bench.py --steps 60 --warmup 15 (one MI355X, one call, same box); switches: MDIL_NO_HEADFUSE, MDIL_NO_BNTAIL, MDIL_NO_WGRAD16, -DWC_STORE=0 build" > $S
b shipped_build A=1
b no_fused_head MDIL_NO_HEADFUSE=1
b no_block_boundary_bn_fusion MDIL_NO_BNTAIL=1
b no_wgrad16 MDIL_NO_WGRAD16=1
b half_line_stores_WC_STORE_0 MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_store0.so
b all_four_off_=_round_2_kernels MDIL_NO_HEADFUSE=1 MDIL_NO_BNTAIL=1 MDIL_NO_WGRAD16=1 MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_store0.so
b shipped_build_again A=1
cat $S
