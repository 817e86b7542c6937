#!/bin/bash
# round 3, GPU call 6: more HIP mIoU samples + the mIoU parity test itself
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03e; mkdir -p $O
cd $R
t0=$(date +%s); timeout 1700 python tools/miou_hip_sample.py --seeds 3035-3066 --procs 2 --stall 500 --out $O/miou_hip > $O/miou_pool.log 2>&1; echo "miou pool(2) rc $? $(( $(date +%s) - t0 )) s $(grep -c SAMPLE $O/miou_pool.log)" > $O/summary.txt
cat $O/summary.txt
