#!/bin/bash
# round 5, call 15: frozen build: full GPU suite (without the mIoU statistic), smoke, default bench line
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05o; mkdir -p $O
cd $R
python -c "from tests.helpers import kernel_build_id; print('build', kernel_build_id())"
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_miou_parity.py > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; cat $O/bench_default.json | cut -c1-1500
