#!/bin/bash
# round 4, call 34: 40 more mIoU-protocol samples of the build under test (seeds 4081-4120)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/miou_hip_r04c; mkdir -p $O
cd $R
python -c "from tests.helpers import kernel_build_id; print('build', kernel_build_id())"
timeout 3000 python tools/miou_hip_sample.py --seeds 4081-4120 --procs 4 --out $O 2>&1 | grep -v amdgpu.ids | grep SAMPLE | wc -l
ls $O/*.npz | wc -l
