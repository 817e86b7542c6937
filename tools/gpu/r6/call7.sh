#!/bin/bash
# round 6, call 7: GPU suite of the frozen build (mIoU test apart) + 40 mIoU-protocol samples of it (seeds 6001-6040)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/miou_hip_r06; mkdir -p $O
cd $R
python -c "from tests.helpers import kernel_build_id; print('build', kernel_build_id())"
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_miou_parity.py > $R/gpurun_out/r06_pytest_gpu.log 2>&1; tail -4 $R/gpurun_out/r06_pytest_gpu.log | cut -c1-200
timeout 3300 python tools/miou_hip_sample.py --seeds 6001-6040 --procs 4 --out $O 2>&1 | grep -v amdgpu.ids | grep SAMPLE | tail -3
ls $O/*.npz | wc -l
