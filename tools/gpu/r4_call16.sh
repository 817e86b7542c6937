#!/bin/bash
# round 4, call 16: staggered three-stream schedule (old-domain graph starts k plan steps behind the
# new-domain graph, one backward per graph) against lock step; with / without the pipelined frozen model
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04p; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_model_golden.py tests/test_dp_gpu.py tests/test_gradient_adjudication.py -m gpu -x -q > $O/pytest_stagger.log 2>&1; tail -3 $O/pytest_stagger.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
b() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 $PT > $O/b_$name.json 2> $O/b_$name.err; echo "$name $(python -c "import json; d=json.loads(open('$O/b_$name.json').read().strip().splitlines()[-1]); print('%.1f img/s  %.3f ms/step  loss %.5f' % (d['value'], d['ms_per_step'], d['final_total_loss']))" 2>&1 | tail -1)"; }
PT=""
for k in off 0 4 8 12 16 19 22; do b stag_$k MDIL_STAGGER=$k; done
PT="--pipeline-teacher"
for k in off 0 4 8 12 16 19 22; do b stagpt_$k MDIL_STAGGER=$k; done
