#!/bin/bash
# more HIP-side mIoU protocol runs from the final build (seeds 3125-3150), two processes sharing the GPU
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03s; mkdir -p $O
cd $R
timeout 760 python tools/miou_hip_sample.py --seeds 3125-3150 --procs 2 --stall 500 --out $O/miou_hip > $O/sample.log 2>&1
echo "rc $?" >> $O/sample.log
grep -c SAMPLE $O/sample.log
tail -3 $O/sample.log
