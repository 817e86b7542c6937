#!/bin/bash
# round 4, call 22: 40 mIoU-protocol samples of the build under test (tests/test_miou_parity.py needs >= 32)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/miou_hip_r04; mkdir -p $O
cd $R
python -c "from tests.helpers import kernel_build_id; print('build', kernel_build_id())"
timeout 3000 python tools/miou_hip_sample.py --seeds 4001-4040 --procs 4 --out $O 2>&1 | grep -v amdgpu.ids | tail -45
ls $O | wc -l
