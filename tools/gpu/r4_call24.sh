#!/bin/bash
# round 4, call 24: the round's evidence from the final build (tools/collect_profiles.sh, TAG=r04)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
cd $R
TAG=r04 timeout 2400 bash tools/collect_profiles.sh 2>&1 | tail -5
O=$R/gpurun_out/r04
for f in $O/bench_step2*.json; do echo "$(basename $f) $(python -c "import json; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")"; done
du -sh $O
