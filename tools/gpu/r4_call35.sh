#!/bin/bash
# round 4, call 35: Step3Engine, staggered phase B: tests + bench A/B
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04ac; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_step3_gpu.py -m gpu -q > $O/pytest_step3.log 2>&1; tail -2 $O/pytest_step3.log; grep "^FAILED\|Error" $O/pytest_step3.log | head -5
b() { name=$1; shift; env "$@" timeout 300 python bench.py --workload step3 --steps 30 --warmup 8 --no-cpu-baseline > $O/b_$name.json 2> $O/b_$name.err; echo "$name $(python -c "import json; d=json.loads(open('$O/b_$name.json').read().strip().splitlines()[-1]); print('%.1f img/s  %.3f ms/step' % (d['value'], d['ms_per_step']))" 2>&1 | tail -1)"; }
for r in 1 2; do for k in off 4 8 12; do b s3_${k}_$r MDIL_STAGGER3=$k; done; done
