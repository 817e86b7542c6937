#!/bin/bash
# round 4, call 26: fused finalize only where the producer is a conv (stand-alone statistics / reduction passes and
# the 16-channel blocks back on separate finalize launches) -- host-side switches, bit-identical
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04x; mkdir -p $O
cd $R
MDIL_BNFIN_PASS=0 MDIL_BNFIN_MINC=64 timeout 600 python -m pytest tests/test_hip_parity.py tests/test_bn_finalize_gpu.py tests/test_model_golden.py -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
b() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 > $O/b_$name.json 2> $O/b_$name.err; echo "$name $(python -c "import json; d=json.loads(open('$O/b_$name.json').read().strip().splitlines()[-1]); print('%.1f img/s  %.3f ms/step  loss %.5f' % (d['value'], d['ms_per_step'], d['final_total_loss']))" 2>&1 | tail -1)"; }
for r in 1 2; do
b base_$r A=1
b pass0_$r MDIL_BNFIN_PASS=0
b minc64_$r MDIL_BNFIN_MINC=64
b both_$r MDIL_BNFIN_PASS=0 MDIL_BNFIN_MINC=64
done
