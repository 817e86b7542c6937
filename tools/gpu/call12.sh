#!/bin/bash
# A/B on one box: round-2 finalize kernels (libmdil_finlds.so) vs the single-wave form
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03l; mkdir -p $O
cd $R
S=$O/summary.txt
b() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 > $O/b_$name.json 2> /dev/null; echo "$name $(python -c "import json; d=json.loads(open('$O/b_$name.json').read().strip().splitlines()[-1]); print('%.1f img/s  %.3f ms/step' % (d['value'], d['ms_per_step']))")" >> $S; }
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_gradient_adjudication.py tests/test_model_golden.py -m gpu -q --tb=short > $O/pytest.log 2>&1; echo "pytest rc $? $(grep -E ' passed| failed' $O/pytest.log | tail -1 | cut -c1-200)" >> $S
for i in 1 2 3; do
b old$i MDIL_HIP_LIB=$R/mdil_ss_amd/libmdil_finlds.so
b new$i A=1
done
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_3streams -- python $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 --profile-steps 0 > /dev/null 2>&1)
f=$(find $O/stats_3streams -name "*kernel_stats.csv" | head -1)
grep -E "finalize" $f | cut -c1-60,150-400 >> $S
cat $S
