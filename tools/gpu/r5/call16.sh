#!/bin/bash
# round 5, call 16: 40 mIoU-protocol samples of the frozen build (seeds 5001-5040)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/miou_hip_r05; mkdir -p $O
cd $R
python -c "from tests.helpers import kernel_build_id; print('build', kernel_build_id())"
timeout 3300 python tools/miou_hip_sample.py --seeds 5001-5040 --procs 4 --out $O 2>&1 | grep -v amdgpu.ids | grep SAMPLE | tail -5
ls $O/*.npz | wc -l
