#!/bin/bash
# round 4, call 5: BatchNorm finalize inside the producing launches (tickets): bit-identity tests,
# the GPU suite, A/B bench against MDIL_NO_BNFIN=1
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04e; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_bn_finalize_gpu.py -m gpu -x -q > $O/pytest_fin.log 2>&1; tail -30 $O/pytest_fin.log
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_miou_parity.py > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $O/pytest_gpu.log
tail -25 $O/pytest_gpu.log
for v in fin nofin fin2 nofin2; do
  E="A=1"; [ ${v:0:5} = nofin ] && E="MDIL_NO_BNFIN=1"
  env $E timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 > $O/bench_$v.json 2> $O/bench_$v.err
  echo $v; python -c "import json,sys; d=json.loads(open('$O/bench_$v.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
env timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --profile-steps 0 --single-stream > $O/bench1s_fin.json 2>> $O/bench_fin.err
env MDIL_NO_BNFIN=1 timeout 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --profile-steps 0 --single-stream > $O/bench1s_nofin.json 2>> $O/bench_fin.err
for f in $O/bench1s_*.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
