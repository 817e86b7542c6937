#!/bin/bash
# round 4, call 31: half-chip persistent kernels under the staggered schedule (128 / 192 work-groups per conv and
# weight-gradient launch: two streams' kernels side by side, each LDS fill amortised over twice the tiles)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04aa; mkdir -p $O
cd $R
MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_cus128.so MDIL_WGRAD2_CUS=128 MDIL_SCONV_CUS=128 timeout 600 python -m pytest tests/test_hip_parity.py tests/test_bn_finalize_gpu.py -m gpu -x -q > $O/pytest_cus128.log 2>&1; tail -2 $O/pytest_cus128.log
b() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 > $O/b_$name.json 2> $O/b_$name.err; echo "$name $(python -c "import json; d=json.loads(open('$O/b_$name.json').read().strip().splitlines()[-1]); print('%.1f img/s  %.3f ms/step  loss %.5f' % (d['value'], d['ms_per_step'], d['final_total_loss']))" 2>&1 | tail -1)"; }
for r in 1 2; do
b base_$r A=1
b cus128_$r MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_cus128.so MDIL_WGRAD2_CUS=128 MDIL_SCONV_CUS=128
b cus192_$r MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_cus192.so MDIL_WGRAD2_CUS=192 MDIL_SCONV_CUS=192
b cus128_lock_$r MDIL_STAGGER=off MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_cus128.so MDIL_WGRAD2_CUS=128 MDIL_SCONV_CUS=128
done
env MDIL_HIP_LIB=$R/gpurun_tmp/libmdil_cus128.so timeout 300 python tools/bench_kernels.py --filter "conv128 1x3\|wgrad128" 2>&1 | grep -v amdgpu | cut -c1-60 | head
