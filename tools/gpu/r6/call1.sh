#!/bin/bash
# round 6, call 1: (a) N = 12 against 2 x N = 6 for the shared-weight conv / weight-gradient launches (VERDICT r5 #1a),
# (b) the reference's own stack on this GPU (stock PyTorch-ROCm / MIOpen), (c) this box's bench line
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r06a; mkdir -p $O
cd $R
for r in 1 2; do timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench %.1f img/s %.3f ms' % (d['value'], d['ms_per_step']))"; done | tee $O/bench.txt
timeout 600 python tools/probes/wconv_fit.py 2>&1 | grep -v amdgpu.ids | tee $O/wconv_fit.txt
cd /tmp && export TMPDIR=/tmp
for n in 2 4 6 8 12; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/w$n -- python $R/tools/probes/wgrad_fit.py --N $n > $O/run_$n.log 2>&1
  f=$(find $O/w$n -name "*kernel_stats.csv" | head -1); cp $f $O/wgrad_stats_N$n.csv; rm -rf $O/w$n
done
cd $R
python - <<'PY' | tee $O/wgrad_fit.txt
import csv, os, numpy as np, collections
O=os.environ.get('GRAFT_REPO_ROOT', os.getcwd())+'/gpurun_out/r06a'
Ns=[2,4,6,8,12]; t=collections.defaultdict(dict)
for n in Ns:
    for r in csv.DictReader(open(f'{O}/wgrad_stats_N{n}.csv')):
        k=r['Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0]
        if 'wgrad' in k: t[k][n]=float(r['AverageNs'])/1e3
for k,v in sorted(t.items()):
    if len(v)<len(Ns): continue
    b,a=np.polyfit(np.array(Ns,float), np.array([v[n] for n in Ns]),1)
    print(f"{k:40s} "+" ".join(f"N={n}: {v[n]:6.1f}" for n in Ns)+f"   fit {a:5.1f} us fixed + {b:5.2f} us/image   2 x N=6 -> N=12: {2*v[6]:6.1f} -> {v[12]:6.1f} ({100*(1-v[12]/(2*v[6])):4.1f} % saved)")
PY
timeout 500 python tools/stock_rocm_baseline.py --iters 10 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/stock_rocm.txt
timeout 500 python tools/stock_rocm_baseline.py --iters 10 --benchmark 2>&1 | grep -v amdgpu.ids | tail -3 | tee -a $O/stock_rocm.txt
