#!/bin/bash
# round 4, call 7: explicit boundary chain (no tensor attributes): GPU suite + bench
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04g; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_miou_parity.py > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -25 $O/pytest_gpu.log
for v in a b; do
  timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 > $O/bench_$v.json 2> $O/bench_$v.err
  echo $v; python -c "import json,sys; d=json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
python - <<'PY'
import torch, sys
sys.path.insert(0, '.')
import bench
PY
