#!/bin/bash
# round 5, call 12: where do the main loop's 40 cycles per MFMA go?  ablation builds (timing only) under the stamp probe
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05l; mkdir -p $O
cd $R
for v in w4timing w4abl1 w4abl2 w4abl4 w4abl7; do
  echo "=== $v"; MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_$v.so timeout 300 python tools/probes/w4conv_stamp_probe.py 2>&1 | grep -v amdgpu.ids | grep "^C=\|loop: cycles\|kernel end\|waves 0-3 tile 1\|weights resident" 
done > $O/ablate.txt 2>&1
cat $O/ablate.txt
