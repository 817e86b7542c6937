#!/bin/bash
# round 5, call 8: full GPU suite on the w4conv build; power / clock traces (bench loop, bare MFMA soak, MFMA + clustered VALU soak)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05h; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_miou_parity.py > $O/pytest_gpu.log 2>&1; tail -15 $O/pytest_gpu.log
python tools/clock_power_trace.py --out $O/trace_bench.csv -- python bench.py --steps 1200 --warmup 20 --no-cpu-baseline --profile-steps 0 > $O/trace_bench.txt 2>&1; tail -8 $O/trace_bench.txt
python tools/clock_power_trace.py --out $O/trace_soak_bare.csv -- gpurun_tmp/mfp soak 20 0 > $O/trace_soak_bare.txt 2>&1; tail -8 $O/trace_soak_bare.txt
python tools/clock_power_trace.py --out $O/trace_soak_pk.csv -- gpurun_tmp/mfp soak 20 9 > $O/trace_soak_pk.txt 2>&1; tail -8 $O/trace_soak_pk.txt
