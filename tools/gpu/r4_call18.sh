#!/bin/bash
# round 4, call 18: staggered schedule, release point of the pipelined frozen-model forward
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04r; mkdir -p $O
cd $R
b() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 $PT > $O/b_$name.json 2> $O/b_$name.err; echo "$name $(python -c "import json; d=json.loads(open('$O/b_$name.json').read().strip().splitlines()[-1]); print('%.1f img/s  %.3f ms/step  loss %.5f' % (d['value'], d['ms_per_step'], d['final_total_loss']))" 2>&1 | tail -1)"; }
PT=""
b lock MDIL_STAGGER=off
b s8 MDIL_STAGGER=8
PT="--pipeline-teacher"
for k in 6 8 10; do for ta in -1 12 15 18 21; do b s${k}_t$ta MDIL_STAGGER=$k MDIL_TEACHER_AT=$ta; done; done
PT=""
b lock2 MDIL_STAGGER=off
b s8b MDIL_STAGGER=8
