#!/bin/bash
# round 4, call 13: fixed cost of a wconv launch (time against batch size); effective shader clock per kernel
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04m; mkdir -p $O
cd $R
timeout 600 python tools/probes/wconv_fit.py 2>&1 | tee $O/wconv_fit.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc_clk -- python $R/tools/bench_kernels.py --filter "conv" --iters 4 > /dev/null 2>&1
cd $R
python - <<'PY' | tee $O/clock.txt
import csv, glob, collections, os
O=os.environ.get('GRAFT_REPO_ROOT', os.getcwd())+'/gpurun_out/r04m'
cs=glob.glob(O+'/pmc_clk/**/*counter_collection.csv', recursive=True)
ks=glob.glob(O+'/pmc_clk/**/*kernel_trace.csv', recursive=True)
print(cs, ks)
dur={}
for r in csv.DictReader(open(ks[0])):
    dur[r['Dispatch_Id']]=(r['Kernel_Name'], int(r['End_Timestamp'])-int(r['Start_Timestamp']))
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cs[0])):
    d=r['Dispatch_Id']
    if d in dur: acc[dur[d][0]][r['Counter_Name']].append((float(r['Counter_Value']), dur[d][1]))
for k,v in acc.items():
    if 'wconv' not in k and 'wgrad' not in k: continue
    g=v.get('GRBM_GUI_ACTIVE',[]); m=v.get('SQ_VALU_MFMA_BUSY_CYCLES',[])
    if not g: continue
    gv=sum(x for x,_ in g)/len(g); ns=sum(t for _,t in g)/len(g); mv=sum(x for x,_ in m)/len(m) if m else 0
    print(f"{k[-58:]:58s} dur {ns/1e3:6.1f} us  GUI_ACTIVE {gv:10.0f}  -> {gv/ns:6.2f} cyc/ns (x1/8: {gv/ns/8:5.2f})  MFMA busy/SIMD {mv/1024:8.0f} cyc = {mv/1024/(ns)*1.0:5.2f} cyc/ns")
PY
rm -rf $O/pmc_clk
