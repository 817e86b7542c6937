#!/bin/bash
# round 5, call 5: per-wave stamps of w4conv launches (tuning build)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05e; mkdir -p $O
cd $R
MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_w4timing.so timeout 300 python tools/probes/w4conv_stamp_probe.py > $O/stamps.txt 2>&1; cat $O/stamps.txt
