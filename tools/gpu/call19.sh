#!/bin/bash
# last GPU minutes of the round: HIP-side mIoU protocol runs, seeds 3137-3146, two processes sharing the GPU
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03t; mkdir -p $O
cd $R
timeout 380 python tools/miou_hip_sample.py --seeds 3137-3146 --procs 2 --stall 300 --out $O/miou_hip > $O/sample.log 2>&1
echo "rc $?" >> $O/sample.log
grep -c SAMPLE $O/sample.log
