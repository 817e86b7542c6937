#!/bin/bash
# round 4, call 14: fixed / marginal cost of the weight-gradient kernels (rocprofv3 averages against batch size)
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04n; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in 2 4 6 8 12; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/w$n -- python $R/tools/probes/wgrad_fit.py --N $n > $O/run_$n.log 2>&1
  f=$(find $O/w$n -name "*kernel_stats.csv" | head -1); cp $f $O/wgrad_stats_N$n.csv; rm -rf $O/w$n
done
cd $R
python - <<'PY' | tee $O/wgrad_fit.txt
import csv, os, numpy as np, collections
O=os.environ.get('GRAFT_REPO_ROOT', os.getcwd())+'/gpurun_out/r04n'
Ns=[2,4,6,8,12]; t=collections.defaultdict(dict)
for n in Ns:
    for r in csv.DictReader(open(f'{O}/wgrad_stats_N{n}.csv')):
        k=r['Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0]
        if 'wgrad' in k: t[k][n]=float(r['AverageNs'])/1e3
for k,v in sorted(t.items()):
    if len(v)<len(Ns): continue
    b,a=np.polyfit(np.array(Ns,float), np.array([v[n] for n in Ns]),1)
    print(f"{k:40s} "+" ".join(f"N={n}: {v[n]:6.1f}" for n in Ns)+f"   fit {a:5.1f} us fixed + {b:5.2f} us/image (batch 6 body {6*b:5.1f})")
PY
