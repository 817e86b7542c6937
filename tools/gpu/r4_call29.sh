#!/bin/bash
# round 4, call 29: final validation from the round's last source state: GPU suite, smoke, default bench line
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04final; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q --tb=short > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -1; grep "^FAILED\|^ERROR" $O/pytest_gpu.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log | cut -c1-300
timeout 400 python bench.py > $O/bench_step2_default.json 2> $O/bench.err; tail -1 $O/bench_step2_default.json | cut -c1-1800
