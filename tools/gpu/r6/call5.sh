#!/bin/bash
# round 6, call 5: hipGraph replay of the step against eager enqueue (host floor); upper bound of fusing the
# separate BatchNorm statistics launches
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r06e; mkdir -p $O
cd $R
for r in 1 2; do
  timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('eager enqueue   %.1f img/s %.3f ms  host enqueue %.2f ms/step' % (d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step', -1)))"
  timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 --graph 2>$O/graph_err_$r.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('hipGraph replay %.1f img/s %.3f ms  host enqueue %.2f ms/step hipgraph=%s' % (d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step', -1), d.get('hipgraph')))"
done | tee $O/graph_ab.txt
tail -3 $O/graph_err_1.txt
timeout 600 python tools/ablate_bn_stats.py 2>&1 | grep -v amdgpu.ids | tee $O/ablate_bn_stats.txt
