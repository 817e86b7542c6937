#!/bin/bash
# round 6, call 6: what the stream structure of a launch-paired student costs: both student graphs on ONE stream beside the
# frozen model's (MDIL_STUDENT_ONE_STREAM=1), lock step and staggered, against the shipped three-stream schedule
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r06f; mkdir -p $O
cd $R
run() { timeout 300 env "$@" python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f img/s %.3f ms' % (d['value'], d['ms_per_step']))"; }
for r in 1 2; do
  echo "shipped: three streams, staggered            $(run A=1)"
  echo "three streams, lock step                     $(run MDIL_STAGGER=off)"
  echo "students on ONE stream + frozen, lock step   $(run MDIL_STUDENT_ONE_STREAM=1 MDIL_STAGGER=off)"
  echo "students on ONE stream + frozen, staggered   $(run MDIL_STUDENT_ONE_STREAM=1)"
  echo "everything on one stream                     $(timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 --single-stream 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f img/s %.3f ms' % (d['value'], d['ms_per_step']))")"
done | tee $O/student_one_stream.txt
