#!/bin/bash
# round 4, call 27: how far the host runs ahead of the GPU in the step-2 loop, and where it waits
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04y; mkdir -p $O
cd $R
for k in 8 off; do echo "== MDIL_STAGGER=$k"; MDIL_STAGGER=$k timeout 300 python tools/host_lag.py --steps 30 2>&1 | grep -v amdgpu.ids | tee $O/host_lag_$k.txt | tail -14; done
