#!/bin/bash
# round 5, call 17: 40 more mIoU-protocol samples of the frozen build (seeds 5041-5080)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/miou_hip_r05b; mkdir -p $O
cd $R
python -c "from tests.helpers import kernel_build_id; print('build', kernel_build_id())"
timeout 3300 python tools/miou_hip_sample.py --seeds 5041-5080 --procs 4 --out $O 2>&1 | grep -v amdgpu.ids | grep SAMPLE | tail -3
ls $O/*.npz | wc -l
