#!/bin/bash
# round 5, call 18: final validation from the round's last source state: the whole GPU suite as the driver runs it, smoke, default bench line
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05final; mkdir -p $O
cd $R
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu_final.log 2>&1; tail -3 $O/pytest_gpu_final.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log | cut -c1-200
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['alg_equiv_frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])"
