#!/bin/bash
# LDS-free single-wave BN finalize kernels: tests, A/B bench lines, 3-stream kernel stats
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03k; mkdir -p $O
cd $R
S=$O/summary.txt
b() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 > $O/b_$name.json 2> /dev/null; echo "$name $(python -c "import json; d=json.loads(open('$O/b_$name.json').read().strip().splitlines()[-1]); print('%.1f img/s  %.3f ms/step' % (d['value'], d['ms_per_step']))")" >> $S; }
b shipped A=1
b shipped2 A=1
timeout 1000 python -m pytest tests -m gpu -q --tb=short --deselect tests/test_miou_parity.py > $O/pytest.log 2>&1; echo "pytest rc $? $(grep -E ' passed| failed' $O/pytest.log | tail -1 | cut -c1-200)" >> $S
grep -E "^FAILED|^ERROR" $O/pytest.log | cut -c1-250 >> $S
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $? $(tail -1 $O/smoke.log | cut -c1-160)" >> $S
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_3streams -- python $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 --profile-steps 0 > /dev/null 2>&1)
f=$(find $O/stats_3streams -name "*kernel_stats.csv" | head -1)
grep -E "finalize|bn_stats|bn_bwd_reduce" $f | cut -c1-60,100-400 >> $S
b shipped3 A=1
cat $S
