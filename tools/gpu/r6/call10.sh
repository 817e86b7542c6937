#!/bin/bash
# round 6, call 10: where the frozen model's forward sits in the staggered schedule (MDIL_TEACHER_LAG = plan steps behind the
# new-domain graph; 0 = shipped) + stagger depth re-check on the same box
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r06h; mkdir -p $O
cd $R
run() { timeout 300 env "$@" python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f img/s %.3f ms' % (d['value'], d['ms_per_step']))"; }
for r in 1 2; do
  for lag in 0 4 8 12 16 20; do echo "teacher lag $lag (stagger 8):  $(run MDIL_TEACHER_LAG=$lag)"; done
  for sg in 6 10 12; do echo "stagger $sg (teacher lag 0):     $(run MDIL_STAGGER=$sg)"; done
  echo "stagger 12, teacher lag 8:      $(run MDIL_STAGGER=12 MDIL_TEACHER_LAG=8)"
  echo "pipelined teacher (next batch), stagger 8: $(timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 --pipeline-teacher 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f img/s %.3f ms' % (d['value'], d['ms_per_step']))")"
done | tee $O/teacher_lag.txt
