#!/bin/bash
# re-entry validation of the restored tree: GPU tests, smoke(), the default bench line
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03q; mkdir -p $O
cd $R
timeout 420 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?" >> $O/smoke.log
timeout 200 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?" >> $O/bench_default.err
tail -3 $O/pytest_gpu.log; tail -2 $O/smoke.log; cat $O/bench_default.json
