#!/bin/bash
# Collects the round's rocprofv3 / bench evidence on the GPU box into gpurun_out/$TAG/ (run through
# gpurun from the repo root; TAG defaults to r05); tools/summarize_profiles.py then writes the
# summaries that are committed under profiles/.
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
TAG=${TAG:-r05}
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_single -- $B --steps 4 --warmup 1 --profile-steps 0 --single-stream > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_3streams -- $B --steps 4 --warmup 1 --profile-steps 0 > /dev/null 2>&1
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT TCC_MISS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  n=$(echo $c | tr " " "_" | cut -c1-24)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$n -- python $R/tools/bench_kernels.py --filter "d" --iters 4 > /dev/null 2>&1   # "d": every name with a d (conv d*, wgrad, head, kld, ...)
done
cd $R
python tools/bench_kernels.py > $O/kernel_microbench.txt 2>&1
python bench.py > $O/bench_step2.json 2> /dev/null
# kernel-set A/B of round 5 (same box, 60 timed steps each): shipped (F(4,3) convs), F(2,3) only (round 4's conv kernels), direct form
for r in 1 2; do
python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 > $O/bench_step2_f43_$r.json 2> /dev/null
MDIL_NO_W4CONV=1 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 > $O/bench_step2_f23_$r.json 2> /dev/null
MDIL_NO_WCONV=1 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 > $O/bench_step2_direct_$r.json 2> /dev/null
done
MDIL_STAGGER=off python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 > $O/bench_step2_lockstep.json 2> /dev/null
python bench.py --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 0 --single-stream > $O/bench_step2_single_stream.json 2> /dev/null
for w in step1 step3 multitask eval; do python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_$w.json 2> /dev/null; done
python tools/bench_loader.py --workers 4 8 16 --cached --device > $O/loader_throughput.txt 2>&1
python tools/host_contention.py > $O/host_contention.txt 2>&1
ls $O
