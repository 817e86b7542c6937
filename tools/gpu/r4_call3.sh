#!/bin/bash
# round 4, call 3: full GPU suite (eval-coefficient cache, async_wgrad fake-rank test, covering-size
# trained-state checks inside the mIoU test) + bench
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04c; mkdir -p $O
cd $R
timeout 300 python tools/gpu/r4_dbg_async.py > $O/dbg_async.txt 2>&1; cat $O/dbg_async.txt | tail -5
timeout 1500 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
grep -E "passed|failed|error|covering|one-step parity" $O/pytest_gpu.log | tail -30
timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 > $O/bench.json 2> $O/bench.err


for f in $O/bench*.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
