#!/bin/bash
# round 5, call 2: F(4,3) in the step: bench A/B (w4conv on / off), full GPU suite
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05b; mkdir -p $O
cd $R
b() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline > $O/b_$name.json 2> $O/b_$name.err; echo "$name $(python -c "import json; d=json.loads(open('$O/b_$name.json').read().strip().splitlines()[-1]); print('%.1f img/s  %.3f ms/step' % (d['value'], d['ms_per_step']), d.get('roofline'))" 2>&1 | tail -1)"; }
for r in 1 2; do b w4_$r X=1; b w2_$r MDIL_NO_W4CONV=1; done
timeout 1800 python -m pytest tests -m gpu -q -x --deselect tests/test_miou_parity.py > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
