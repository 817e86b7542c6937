#!/bin/bash
# finalize merged into the BN apply launches: tests, same-box A/B, 3-stream kernel stats
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03m; mkdir -p $O
cd $R
S=$O/summary.txt
b() { name=$1; shift; env "$@" timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 > $O/b_$name.json 2> /dev/null; echo "$name $(python -c "import json; d=json.loads(open('$O/b_$name.json').read().strip().splitlines()[-1]); print('%.1f img/s  %.3f ms/step' % (d['value'], d['ms_per_step']))")" >> $S; }
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q --tb=short -x -k "bn_finalize or nb_block or down_block or up_block" > $O/pytest1.log 2>&1; echo "pytest1 rc $? $(grep -E ' passed| failed' $O/pytest1.log | tail -1 | cut -c1-200)" >> $S
for i in 1 2 3; do
b old$i MDIL_HIP_LIB=$R/mdil_ss_amd/libmdil_finlds.so MDIL_NO_BNFINFUSE=1
b wave$i MDIL_NO_BNFINFUSE=1
b fused$i A=1
done
timeout 1000 python -m pytest tests -m gpu -q --tb=short --deselect tests/test_miou_parity.py > $O/pytest.log 2>&1; echo "pytest rc $? $(grep -E ' passed| failed' $O/pytest.log | tail -1 | cut -c1-200)" >> $S
grep -E "^FAILED|^ERROR" $O/pytest.log | cut -c1-250 >> $S
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_3streams -- python $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 --profile-steps 0 > /dev/null 2>&1)
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_single -- python $R/bench.py --no-cpu-baseline --steps 4 --warmup 1 --profile-steps 0 --single-stream > /dev/null 2>&1)
for k in 3streams single; do f=$(find $O/stats_$k -name "*kernel_stats.csv" | head -1); echo $k >> $S; grep -E "bn_" $f | cut -c1-50,120-400 >> $S; done
cat $S
