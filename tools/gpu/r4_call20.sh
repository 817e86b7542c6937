#!/bin/bash
# round 4, call 20: staggered schedule -- stream priorities, frozen model serial on the old-domain stream
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04t; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_model_golden.py -m gpu -q -k "staggered or three_stream" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
b() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 $PT > $O/b_$name.json 2> $O/b_$name.err; echo "$name $(python -c "import json; d=json.loads(open('$O/b_$name.json').read().strip().splitlines()[-1]); print('%.1f img/s  %.3f ms/step  loss %.5f' % (d['value'], d['ms_per_step'], d['final_total_loss']))" 2>&1 | tail -1)"; }
PT=""
b s8 MDIL_STAGGER=8
for pr in "-1,0,0" "0,-1,0" "0,0,0" "-1,-1,-1" "0,0,-1"; do b s8_prio_$pr MDIL_STAGGER=8 MDIL_STREAM_PRIO=$pr; done
for k in 0 4 8 12; do b serial_$k MDIL_STAGGER=$k MDIL_TEACHER_SERIAL=1; done
b s8b MDIL_STAGGER=8
PT="--pipeline-teacher"
b s8pt MDIL_STAGGER=8
for pr in "-1,0,0" "0,-1,0" "0,0,0" "-1,-1,-1"; do b s8pt_prio_$pr MDIL_STAGGER=8 MDIL_STREAM_PRIO=$pr; done
