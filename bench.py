#!/usr/bin/env python3
"""Headline benchmark: images/sec of the ERFNet-RAP step-2 (CS->BDD, KD on) training iteration at
1024x512, batch 6 per GPU (BASELINE.json configs[2]; train_new_task_step2.py:285-306).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = 2 student forwards (train mode, batch-stat BN, Dropout2d) + frozen teacher forward
(eval) + weighted CE + 0.1*KLD + backward through both student graphs + RCCL all-reduce of the
flat gradient buffer (N>1) + fused Adam, on a fresh synthetic batch already resident in HBM.
Prints ONE JSON line (rank 0) with the throughput, the roofline of the dominant kernel (HIP-event
timed on the launch stream) and a CPU baseline of the same iteration run by the oracle.
"""
import argparse
import json
import os
import sys
import time

# before anything initialises HIP (mdil_ss_amd/__init__.py explains): eight hardware queues instead of four, so
# that the communication stream of a multi-rank run does not share a queue with an engine stream
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# weight_BDD (train_new_task_step2.py:125-127) with [19]=0 (:134)
WEIGHT_BDD = [3.6525147483016243, 8.799815287822142, 4.781908267406055, 10.034828238618045,
              9.5567865464289, 9.645099012085169, 10.315292989325766, 10.163473632969513,
              4.791692009441432, 9.556915153488912, 4.142994047786311, 10.246903827488143,
              10.47145010979545, 6.006704177894196, 9.60620532303246, 9.964959813857726,
              10.478333987902301, 10.468010534454706, 10.440929141422366, 0.0]

ALG_BYTES_PER_IMAGE = 11.26e9     # fused-minimum fp32 HBM bytes, step 2
ALG_FLOPS_PER_IMAGE = 404e9
HBM_PEAK = 8000.0                 # GB/s   (MI355X_MICROARCH.md)
MFMA_F32_PEAK = 157.3             # TFLOP/s dense fp32 MFMA (= fp32 vector peak)
def pmc_traffic(key):
    """Fabric bytes per launch of the dominant kernel, READ from the committed rocprofv3 PMC
    summaries (profiles/rNN_pmc_FETCH_SIZE.csv / _WRITE_SIZE.csv, newest round present; separate
    passes over tools/bench_kernels.py at the bench shapes): 2 x FETCH_SIZE (gfx950 counts a wide
    coalesced read at half its bytes, MI355X_MICROARCH.md HBM section; WRITE_SIZE is 1:1 --
    calibrated on kernels with a known byte count, profiles/README.md) + WRITE_SIZE, KiB -> bytes.
    Counters cannot be collected from inside this process, so this is the committed measurement of
    the same kernel, averaged over its template variants; None when no summary names it."""
    import csv
    import glob
    kind, cin, _, ntaps = key
    name = {"wconv": f"wconv_kernel<{cin}, {'true' if ntaps == 4 else 'false'},",
            "w4conv": f"w4conv_kernel<{cin}, 32, {'true' if ntaps == 4 else 'false'},",
            "sconv": f"sconv_kernel<{cin}, {ntaps},", "wgrad2": f"wgrad2_kernel<{cin}, {ntaps}",
            "wgradw": f"wgradw_kernel<{cin}>", "tapconv": f"tapconv_kernel<{cin},",
            "wgrad": f"wgrad_kernel<{cin},"}.get(kind)
    pdir = os.path.join(ROOT, "profiles")
    rounds = sorted({os.path.basename(f)[:3] for f in glob.glob(os.path.join(pdir, "r*_pmc_FETCH_SIZE.csv"))})
    if name is None or not rounds:
        return None, None
    # weights of the kernel's template variants = their launch counts in the step itself (the newest
    # committed single-stream kernel statistics of bench.py): the plain form the micro-benchmark
    # times most is a minority of what a training step launches (statistics / BN-reduction / tail forms)
    calls = {}
    sf = os.path.join(pdir, f"{rounds[-1]}_kernel_stats_single_stream.csv")
    if os.path.exists(sf):
        for r in csv.DictReader(open(sf)):
            k = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if k.startswith(name):
                calls[k] = calls.get(k, 0.0) + float(r["Calls"])
    out, variants = {}, {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        f = os.path.join(pdir, f"{rounds[-1]}_pmc_{counter}.csv")
        if not os.path.exists(f):
            return None, None
        tot = n = 0.0
        for r in csv.DictReader(open(f)):
            if r["kernel"].startswith(name) and r["counter"] == counter:
                w = calls.get(r["kernel"], 0.0) if calls else float(r["launches"])
                tot += float(r["mean_per_launch"]) * w
                n += w
                variants.setdefault(r["kernel"], {})[counter] = float(r["mean_per_launch"])
        if not n:
            return None, None
        out[counter] = tot / n
    src = f"profiles/{rounds[-1]}_pmc_*.csv" + (", variants weighted by their launch counts in "
                                                f"profiles/{rounds[-1]}_kernel_stats_single_stream.csv" if calls else "")
    return (2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024.0, src


def build_models(dev):
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd.models.erfnet_RA_parallel import Net
    from mdil_ss_amd import train_new_task_step2 as T
    torch.manual_seed(1)
    teacher = Net([20], 1, 0)
    torch.manual_seed(0)
    student = Net([20, 20], 2, 1)
    ckpt = {"module." + k: v for k, v in teacher.state_dict().items()}
    new = T.student_init_dict(ckpt, {"module." + k for k in student.state_dict()}, 1)
    student.load_state_dict({k[len("module."):]: v for k, v in new.items()}, strict=False)
    student.to(dev)
    teacher.to(dev)
    T.apply_step2_freeze(student, teacher, 1)
    return student, teacher, T


def _host_threads():
    """CPU threads this process may really use (cgroup quota; the GPU box shows 256 logical CPUs
    but grants 16 CPUs' worth of time)."""
    n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _cpu_worker(batch, h, w, threads):
    """Child process: the oracle's step-2 iteration (stock torch fp32 ops on the host cores) at
    the bench's own batch size; prints the seconds of every iteration as it finishes (the first
    one is the warm-up: oneDNN primitive creation)."""
    from oracle import fixtures as fx
    from oracle import rap_oracle as O
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd.models.erfnet_RA_parallel import Net
    torch.set_num_threads(threads)
    torch.manual_seed(1)
    t_sd = {k: v.clone() for k, v in Net([20], 1, 0).state_dict().items()}
    torch.manual_seed(0)
    net = Net([20, 20], 2, 1)
    s_sd = {k: v.clone() for k, v in net.state_dict().items()}
    for k, v in O.student_init_from_teacher(t_sd, s_sd, 1).items():
        s_sd[k].copy_(v)
    names = [n for n, _ in net.named_parameters()]
    for n in names:
        s_sd[n].requires_grad_(O.step2_trainable("module." + n, 1))
    weight = torch.tensor(WEIGHT_BDD)
    for it in range(4):
        images, labels = fx.make_batch(batch, h, w, 20, seed=it, block=16)
        t0 = time.time()
        for n in names:
            s_sd[n].grad = None
        O.step2_iteration(s_sd, t_sd, images, labels, weight, 1, 0.1,
                          O.draw_dropout_masks(batch), O.draw_dropout_masks(batch))
        with torch.no_grad():
            for n in names:
                if s_sd[n].grad is not None:
                    s_sd[n].add_(s_sd[n].grad, alpha=-1e-6)   # stand-in for the optimizer's pass
        print("CPU_BASELINE_SECONDS", it, time.time() - t0, flush=True)


def cpu_baseline(batch, h, w, budget=170.0):
    """Bounded CPU leg (rank 0, N=1): the oracle in a child process at the bench's configuration
    (batch 6, 512x1024, all usable cores, 1 warm-up + 3 timed iterations), cut off
    after ``budget`` seconds -- whatever timed iterations finished by then are averaged; if not
    even one did, a batch-1 sample is scaled by the image count.  Reported beside the GPU number,
    never as the thing measured."""
    import subprocess
    threads = _host_threads()
    for (bb, limit) in ((batch, budget), (1, 60.0)):
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(bb), str(h),
                              str(w), str(threads)], stdout=subprocess.PIPE, text=True,
                             env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
        try:
            out, _ = p.communicate(timeout=limit)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        sec = [float(l.split()[2]) for l in (out or "").splitlines() if l.startswith("CPU_BASELINE_SECONDS")]
        timed = sec[1:] if len(sec) > 1 else []
        if timed:
            mean = sum(timed) / len(timed)
            return {"value": round(bb / mean, 4), "unit": "images/sec", "cores": threads, "kind": "port",
                    "sample": f"oracle (stock torch fp32 ops, the reference's nn graph restated) step-2 "
                              f"iteration, batch {bb} at {w}x{h}, 1 warm-up ({sec[0]:.1f} s) + {len(timed)} "
                              f"timed iteration(s) of {mean:.1f} s, {threads} threads (cgroup CPU quota of "
                              f"{os.cpu_count()} logical CPUs)"}
    return {"value": None, "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": "oracle iteration did not finish inside the bench's CPU time limit"}


# secondary configurations: (algorithmic GB / image, GFLOP / image, description)  SURVEY 8(d)
SECONDARY = {
    "step1": (4.983, 181.0, "ERFNet-RAP step-1 (1 fwd + CE + bwd + Adam), num_classes [20]"),
    "step3": (17.59, 628.0, "ERFNet-RAP step-3 (CE step, then KD on 2 old domains with the "
                            "previous model in train mode: 3 student fwd+bwd, 2 previous-model fwd, "
                            "2 Adam steps), num_classes [20,20,27]"),
    "multitask": (4.9, 181.0, "ERFNet multi-task joint training, 3 heads [20,20,27]: one step = one "
                              "round-robin pass (3 sub-steps of 1 fwd + CE + bwd + Adam); images "
                              "counted over the 3 sub-steps"),
    "eval": (1.256 + 0.05, 60.2, "ERFNet-RAP eval: folded-BN forward + fused argmax/confusion"),
}


def build_secondary(wl, dev, pool, streams):
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import ops
    from mdil_ss_amd import engine as E
    from mdil_ss_amd import train_new_task_step2 as T2
    from mdil_ss_amd.models.erfnet_RA_parallel import Net
    w20 = torch.tensor(WEIGHT_BDD, device=dev)
    w27 = torch.cat([torch.full((26,), 8.0), torch.zeros(1)]).to(dev)
    torch.manual_seed(0)
    if wl == "step1":
        model = Net([20], 1, 0).to(dev)
        eng = E.Step1Engine(model, w20, 0)
        return eng, lambda i: (eng.iteration(*pool[i % len(pool)]),)
    if wl == "eval":
        from mdil_ss_amd.iouEval import iouEval
        model = Net([20, 20], 2, 1).to(dev).eval()
        meter = iouEval(20, 19)

        def step(i):
            img, lab = pool[i % len(pool)]
            with torch.no_grad():
                out = model(img, 1)
                loss = ops.cross_entropy2d(out, lab[:, 0], w20)
                meter.addBatch(out, lab)
            return (loss,)
        return None, step
    if wl == "multitask":
        from mdil_ss_amd.models.erfnet_multi_task import Net as NetMT
        model = NetMT([20, 20, 27], 3, 0).to(dev)
        eng = E.MultiTaskEngine(model, [w20, w20, w27])

        def step(i):
            out = None
            for ind in range(3):
                out = eng.sub_step(ind, *pool[(i + ind) % len(pool)])
            return (out,)
        return eng, step
    from mdil_ss_amd import train_new_task_step3 as T3
    torch.manual_seed(1)
    teacher = Net([20, 20], 2, 1)
    torch.manual_seed(0)
    student = Net([20, 20, 27], 3, 2)
    ckpt = {"module." + k: v for k, v in teacher.state_dict().items()}
    new = T2.student_init_dict(ckpt, {"module." + k for k in student.state_dict()}, 2)
    student.load_state_dict({k[len("module."):]: v for k, v in new.items()}, strict=False)
    student.to(dev)
    teacher.to(dev)
    T2.apply_step2_freeze(student, teacher, 2)
    T3.current_task = 2
    eng = E.Step3Engine(student, teacher, w27, current_task=2, lambdac=0.1, is_shared=T3.is_shared,
                        is_ds_curr=T3.is_DS_curr, streams=streams)
    return eng, lambda i: eng.iteration(*pool[i % len(pool)])


def spawn_ranks(n):
    """Re-execute this command line under torch.distributed.run with one process per GPU on this
    node (rendezvous on 127.0.0.1, a free port) and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    # The image's environment contract: the host driver only supports dmabuf IPC; without this RCCL
    # (and any device-tensor sharing across processes) fails with `hipIpcGetMemHandle: invalid
    # argument`.  It is already exported on the pool's boxes -- setdefault only covers a shell that
    # dropped it; a value the caller set is left alone.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def timed_steps(step, steps, warmup, world, sync):
    """The contract's timed region: W untimed steps, then EXACTLY K steps bracketed by a barrier +
    device synchronisation on both sides; -> (seconds = MAX over ranks, host enqueue seconds,
    result of the last step)."""
    out = None
    for i in range(warmup):
        step(i)
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        out = step(i)
    t_enq = time.perf_counter() - t0           # host time to enqueue the steps (GPU still busy)
    if world > 1:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=out[0].device if out is not None else None)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt, t_enq, out


def stub_cpu(args, world, rank):
    """--stub-cpu: the same launch / barrier / max-over-ranks / one-JSON-line harness on the gloo
    backend with a stand-in step (a small matmul + a gradient-sized all-reduce)."""
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        world = dist.get_world_size()
    g = torch.zeros(1 << 16)

    def step(i):
        g.add_(torch.ones(1 << 16) * (rank + 1))
        if world > 1:
            dist.all_reduce(g)
        return (g[:1].clone(),)
    dt, t_enq, out = timed_steps(step, args.steps, args.warmup, world, lambda: None)
    if rank == 0:
        print(json.dumps({"metric": "stub steps/sec (harness self-test, not a measurement)",
                          "value": round(world * args.steps / dt, 3), "unit": "steps/sec", "n_gpus": world,
                          "rccl_ranks": world, "backend": "gloo", "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "stub", "parallelism": f"dp{world}"},
                          "checksum": float(out[0])}))
    if world > 1:
        dist.destroy_process_group()
    return 0


def main():
    if len(sys.argv) >= 6 and sys.argv[1] == "--cpu-worker":
        return _cpu_worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch-size", type=int, default=6)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--profile-steps", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single-stream", action="store_true",
                    help="enqueue the three forwards / two backwards on one stream")
    ap.add_argument("--async-wgrad", action="store_true",
                    help="hand weight-gradient launches to side streams (experiment: neutral at first, 23 %% "
                         "slower under the current stream / wave priorities)")
    ap.add_argument("--workload", default="step2",
                    choices=["step2", "step1", "step3", "multitask", "eval"],
                    help="step2 = the headline metric (default); the others are the secondary "
                         "configurations of SURVEY 8(d) over the same kernels")
    ap.add_argument("--pipeline-teacher", action="store_true",
                    help="enqueue the frozen model's forward for batch i+1 under batch i's backward")
    ap.add_argument("--graph", action="store_true",
                    help="capture fwd+bwd into a hipGraph (replay costs as much host time as eager "
                         "launches on ROCm 7.2, so it is off by default)")
    ap.add_argument("--stub-cpu", action="store_true",
                    help="(tests) run the launch / timing / reporting harness on CPU tensors with the gloo "
                         "backend and a stand-in step: exercises --gpus N -> N ranks without GPUs")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` (no launcher): become N ranks, one process per GPU
        # (train_new_task_step2.py:474-475 wraps the model in nn.DataParallel over all visible GPUs;
        # here every rank is its own process and the gradients meet in an RCCL all-reduce)
        return spawn_ranks(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); reporting the "
              f"{world} that exist", file=sys.stderr)
    if args.stub_cpu:
        return stub_cpu(args, world, rank)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
        world = dist.get_world_size()           # what is reported is what the process group has

    from mdil_ss_amd import ops
    from mdil_ss_amd.engine import Step2Engine
    B, H, W = args.batch_size, args.height, args.width
    wl = args.workload
    n_label_classes = 27 if wl == "step3" else 20
    pool = []
    for i in range(8):                      # pre-generated pool, resident in HBM
        g = torch.Generator().manual_seed(1234 + 97 * rank + i)
        img = torch.rand(B, 3, H, W, generator=g)
        lab = torch.randint(0, n_label_classes, (B, 1, H // 16, W // 16), generator=g)
        lab = lab.repeat_interleave(16, 2).repeat_interleave(16, 3).contiguous()
        pool.append((img.to(dev), lab.to(dev)))

    if wl == "step2":
        student, teacher, T = build_models(dev)
        T.current_task = 1
        eng = Step2Engine(student, teacher, torch.tensor(WEIGHT_BDD, device=dev), current_task=1,
                          lambdac=0.1, is_shared=T.is_shared, is_ds_curr=T.is_DS_curr,
                          async_wgrad=args.async_wgrad, streams=not args.single_stream)
        eng.optimizer.set_epoch(1, 150)

        def step(i):
            img, lab = pool[i % len(pool)]
            nxt = pool[(i + 1) % len(pool)][0] if args.pipeline_teacher else None
            return eng.iteration(img, lab, nxt)
    else:
        eng, step = build_secondary(wl, dev, pool, not args.single_stream)
        args.single_stream = True if wl in ("step1", "multitask", "eval") else args.single_stream

    step(0)                                  # first step runs on one stream (builds weight images)
    if not args.single_stream:
        step(1)                              # first 3-stream step (per-stream scratch buffers)
        if args.graph and wl == "step2":
            eng.enable_graph(*pool[0])
    dt, t_enq, losses = timed_steps(step, args.steps, args.warmup, world, torch.cuda.synchronize)
    total_loss = float(losses[0])

    # ---- roofline leg: per-launch HIP-event timing of the MFMA kernels on their stream ----
    roof = None
    if args.profile_steps > 0:
        # every rank runs the extra steps (they contain the gradient all-reduce); rank 0 reports.
        # The library brackets each conv / weight-gradient launch of the shipped call sequence
        # (block-level C ABI, deferred reductions) with HIP events on its stream (csrc/prof.cpp).
        saved = (getattr(eng, "graph", None), getattr(eng, "multi_stream", False))
        want = getattr(eng, "want_streams", False)
        if eng is not None:
            eng.graph, eng.multi_stream = None, False   # clean per-kernel durations: eager, one stream
            eng.want_streams = False
        torch.cuda.synchronize()
        ops.profile_begin()
        for i in range(args.profile_steps):
            step(i)
        torch.cuda.synchronize()
        records = ops.profile_end()
        if eng is not None:
            eng.graph, eng.multi_stream = saved
            eng.want_streams = want
    if rank == 0 and args.profile_steps > 0 and records:
        agg = {}
        for kind, cin, cout, ntaps, flops, sec_ in records:
            a = agg.setdefault((kind, cin, cout, ntaps), [0.0, 0.0, 0])
            a[0] += flops
            a[1] += sec_
            a[2] += 1
        key = max(agg, key=lambda k: agg[k][1])
        fl, sec, cnt = agg[key]
        alg = fl / sec / 1e12
        # Winograd launches execute fewer MFMA contractions than the direct form counts: F(2,3) 4 (+2 for
        # the adapter tap) per output pair against 6 (+2); F(4,3) 6 (+4) per output quad against 12 (+4):
        # the flops the matrix pipe really executes
        executed = {"wconv": 2.0 / 3.0 if key[3] == 3 else 0.75, "wgradw": 2.0 / 3.0,
                    "w4conv": 0.5 if key[3] == 3 else 0.625}.get(key[0], 1.0)
        ach = alg * executed
        traffic, traffic_src = pmc_traffic(key)
        roof = {"bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_F32_PEAK,
                "unit": "TFLOP/s", "frac": round(ach / MFMA_F32_PEAK, 4),
                # `achieved` / `frac`: flops the MFMA pipe EXECUTES per second; the same launch priced
                # at its algorithmic (direct-form) flops -- SURVEY 8(d)'s per-unit figure:
                "alg_equiv_achieved": round(alg, 2), "alg_equiv_frac": round(alg / MFMA_F32_PEAK, 4),
                "traffic": None if traffic is None else round(traffic),
                "traffic_note": None if traffic is None else
                f"fabric bytes/launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 read from {traffic_src} "
                "(committed rocprofv3 PMC passes of the same kernel, not live)",
                "kernel": f"{key[0]}_kernel<C={key[1]}, taps={key[3]}>",
                "launches": cnt, "avg_launch_us": round(sec / cnt * 1e6, 2),
                "alg_flops_per_launch": round(fl / cnt / 1e9, 4),
                "share_of_mfma_kernel_time": round(sec / sum(v[1] for v in agg.values()), 3)}

    if rank == 0 and wl != "step2":
        ips = world * B * args.steps / dt * (3 if wl == "multitask" else 1)
        gb, gf, what = SECONDARY[wl]
        print(json.dumps({
            "metric": f"images/sec at 1024x512, {what}, batch 6/GPU", "value": round(ips, 3),
            "unit": "images/sec", "n_gpus": world, "rccl_ranks": world, "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": what, "batch_per_gpu": B, "height": H, "width": W,
                       "parallelism": f"dp{world}"},
            "roofline": roof,
            "step_hbm": {"alg_GBps_per_gpu": round(gb * ips / world, 1),
                         "frac_of_8TBps": round(gb * ips / world / HBM_PEAK, 4),
                         "alg_TFLOPps_per_gpu": round(gf * ips / world / 1e3, 2)},
            "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 2)}))
    elif rank == 0:
        ips = world * B * args.steps / dt
        out = {
            "metric": "images/sec at 1024x512 ERFNet-RA step-2 train (CS->BDD, KD on), batch 6/GPU",
            "value": round(ips, 3), "unit": "images/sec", "n_gpus": world, "rccl_ranks": world,
            "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "ERFNet-RAP step-2 (2 student fwd + teacher fwd + CE + 0.1*KLD "
                                   "+ bwd + Adam), random-init weights of the reference "
                                   "architecture, num_classes [20,20]",
                       "batch_per_gpu": B, "global_batch": B * world, "height": H, "width": W,
                       "parallelism": f"dp{world}"},
            "roofline": roof,
            "step_hbm": {"alg_GBps_per_gpu": round(ALG_BYTES_PER_IMAGE * ips / world / 1e9, 1),
                         "frac_of_8TBps": round(ALG_BYTES_PER_IMAGE * ips / world / 1e9 / HBM_PEAK, 4),
                         "alg_TFLOPps_per_gpu": round(ALG_FLOPS_PER_IMAGE * ips / world / 1e12, 2)},
            "final_total_loss": round(total_loss, 5),
            "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 2),
            "streams": 1 if args.single_stream else 3,
            "schedule": ("single stream" if args.single_stream else
                         "lock step" if getattr(eng, "stagger", None) is None else
                         f"staggered: old-domain graph {eng.stagger} plan steps behind, one backward per graph"
                         + (", frozen model pipelined one batch ahead" if args.pipeline_teacher else "")),
            "hipgraph": bool(getattr(eng, "graph", None) is not None),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(B, H, W)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
