"""CPU (gloo, world_size 2): the data-parallel logic of engine.py -- flat-buffer layout, bucket
boundaries and the bucketed gradient all-reduce -- without a GPU.  Per-rank gradients come from
the oracle so the check is 'averaged flat gradient == mean of the per-rank oracle gradients'."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import mdil_ss_amd  # noqa: F401
        from mdil_ss_amd.engine import GradExchange
        from mdil_ss_amd import train_new_task_step2 as T
        from mdil_ss_amd.models.erfnet_RA_parallel import Net
        from oracle import fixtures as fx
        from oracle import rap_oracle as O

        torch.manual_seed(0)
        net = Net([20, 20], 2, 1)
        T.current_task = 1
        named = [("module." + n, p) for n, p in net.named_parameters()]
        T.apply_step2_freeze(net, Net([20], 1, 0), 1)
        shared = [(n, p) for n, p in named if T.is_shared(n)]
        ds = [(n, p) for n, p in named if T.is_DS_curr(n)]
        order = shared + ds
        sizes = [p.numel() for _, p in order]
        n_shared = sum(p.numel() for _, p in shared)
        assert n_shared == 1868252 and sum(sizes) == 2370048            # SURVEY 2.2 [probed]
        from mdil_ss_amd.engine import Step2Engine
        eng = Step2Engine(net, Net([20], 1, 0), torch.ones(20), current_task=1,
                          is_shared=T.is_shared, is_ds_curr=T.is_DS_curr)
        n_dec = sum(p.numel() for n, p in ds if "decoder" in n)
        assert eng.bucket_dec.numel() == n_dec > 0
        assert eng.bucket_ds_enc.numel() + n_dec == eng.bucket_ds.numel() == sum(p.numel() for _, p in ds)
        assert eng.bucket_dec.data_ptr() == eng.bucket_ds.data_ptr() + 4 * eng.bucket_ds_enc.numel()
        last = [p for n, p in ds if "decoder" in n][-1]
        assert last.grad.data_ptr() + 4 * last.numel() == eng.bucket_dec.data_ptr() + 4 * n_dec
        # per-rank oracle gradients on this rank's shard of the batch (rank-local BN, like DP)
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        tsd = {k: v.clone() for k, v in Net([20], 1, 0).state_dict().items()}
        for n, _ in named:
            sd[n[7:]].requires_grad_(O.step2_trainable(n, 1))
        img, lab = fx.make_batch(1, 16, 32, 20, seed=50 + rank)
        O.step2_iteration(sd, tsd, img, lab, torch.tensor(fx.WEIGHT_BDD), 1, 0.1,
                          O.draw_dropout_masks(1, torch.Generator().manual_seed(rank)),
                          O.draw_dropout_masks(1, torch.Generator().manual_seed(10 + rank)))
        flat = torch.cat([sd[n[7:]].grad.reshape(-1) for n, _ in order])
        local = flat.clone()
        ex = GradExchange()
        assert ex.world == world
        ex.start(flat[n_shared:])          # new-domain bucket first (final after the CE backward)
        ex.start(flat[:n_shared])          # shared bucket
        ex.join()
        flat.mul_(1.0 / world)
        gathered = [torch.empty_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        want = sum(gathered) / world
        torch.testing.assert_close(flat, want, rtol=1e-6, atol=1e-9)
        # every rank ends with the same averaged gradient -> identical replicas after Adam
        ck = [torch.empty(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(ck, flat.double().sum().reshape(1))
        assert float(ck[0]) == float(ck[1])
        if rank == 0:
            out.put("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_bucketed_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(280)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) == "ok"


def _worker_step3(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import mdil_ss_amd  # noqa: F401
        from mdil_ss_amd.engine import Step3Engine
        from mdil_ss_amd import train_new_task_step2 as T2
        from mdil_ss_amd import train_new_task_step3 as T
        from mdil_ss_amd.models.erfnet_RA_parallel import Net

        torch.manual_seed(0)
        student, teacher = Net([20, 20, 27], 3, 2), Net([20, 20], 2, 1)
        T.current_task = 2
        T2.apply_step2_freeze(student, teacher, 2)
        eng = Step3Engine(student, teacher, torch.ones(27), current_task=2, is_shared=T.is_shared,
                          is_ds_curr=T.is_DS_curr)
        assert eng.world == world
        g0, g1 = eng.optimizer.param_groups
        assert g0["numel"] == 1868252 and eng.bucket_shared.numel() == g0["numel"]
        assert eng.bucket_ds.numel() == g1["numel"] == eng.optimizer.flat_grad.numel() - g0["numel"]
        # the KD step exchanges the shared bucket only and steps group 0 only
        eng.optimizer.flat_grad.fill_(float(rank + 1))
        eng.exchange.start(eng.bucket_shared)
        eng.exchange.join()
        assert float(eng.bucket_shared[0]) == 3.0 and float(eng.bucket_ds[0]) == float(rank + 1)
        # running statistics of the train-mode previous model: rank mean, identical on all ranks
        for n, b in teacher.named_buffers():
            if b.dtype.is_floating_point:
                b.fill_(float(rank))
        eng.teacher.train()
        eng._sync_teacher_stats()
        for n, b in teacher.named_buffers():
            if b.dtype.is_floating_point:
                assert torch.all(b == 0.5), n
        if rank == 0:
            out.put("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_step3_exchange_and_teacher_stats_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_step3, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(280)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) == "ok"


def _worker_replicas(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import mdil_ss_amd  # noqa: F401
        from mdil_ss_amd.engine import Step1Engine, Step2Engine
        from mdil_ss_amd import train_new_task_step2 as T
        from mdil_ss_amd.models.erfnet_RA_parallel import Net

        # every process starts from its OWN random init (what torchrun gives an unseeded script)
        torch.manual_seed(1000 + 17 * rank)
        student, teacher = Net([20, 20], 2, 1), Net([20], 1, 0)
        for b in student.buffers():
            if b.dtype.is_floating_point:
                b.add_(float(rank))
        T.current_task = 1
        T.apply_step2_freeze(student, teacher, 1)
        before = torch.cat([t.detach().reshape(-1).double() for t in student.parameters()]).sum()
        eng = Step2Engine(student, teacher, torch.ones(20), current_task=1, is_shared=T.is_shared,
                          is_ds_curr=T.is_DS_curr)

        def digest(m):
            parts = [t.detach().reshape(-1).double() for t in list(m.parameters()) + list(m.buffers())]
            v = torch.cat(parts)
            return torch.stack([v.sum(), v.abs().sum(), (v * torch.arange(v.numel()).double()).sum()])

        for m in (student, teacher):
            mine = digest(m)
            got = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(got, mine)
            assert all(torch.equal(g, got[0]) for g in got), "replicas differ after engine init"
        # rank 0's values won (its sum is unchanged), the others were overwritten
        after = torch.cat([t.detach().reshape(-1).double() for t in student.parameters()]).sum()
        assert (rank != 0) or float(after) == float(before)
        # the flat optimizer buffer holds the broadcast values too
        flat = eng.optimizer.flat_param.double().sum().reshape(1)
        got = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(got, flat)
        assert float(got[0]) == float(got[1])
        # dropout masks: each replica draws its own (same global seed on purpose)
        torch.manual_seed(5)
        m = torch.cat([x.reshape(-1) for x in student.draw_masks(4, torch.device("cpu"))])
        got = [torch.empty_like(m) for _ in range(world)]
        dist.all_gather(got, m)
        assert not torch.equal(got[0], got[1]), "replicas drew identical dropout masks"
        # step-1 engine: everything is trainable and nothing comes from a checkpoint
        torch.manual_seed(2000 + rank)
        net = Net([20], 1, 0)
        Step1Engine(net, torch.ones(20), 0)
        mine = digest(net)
        got = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(got, mine)
        assert torch.equal(got[0], got[1])
        # BN running statistics drift apart during training (rank-local batch
        # statistics); before a validation pass every rank takes rank 0's -- the checkpointed ones
        # (engine.broadcast_buffers, called by every trainer's eval()).  Parameters are untouched.
        from mdil_ss_amd.engine import broadcast_buffers
        with torch.no_grad():
            for b in net.buffers():
                b.add_(rank + 1)                      # rank-local drift (also num_batches_tracked)
        pre_params = torch.cat([t.detach().reshape(-1).double() for t in net.parameters()]).clone()
        mine0 = torch.cat([b.detach().reshape(-1).double() for b in net.buffers()]).clone()
        broadcast_buffers(net)
        bufs = torch.cat([b.detach().reshape(-1).double() for b in net.buffers()])
        got = [torch.empty_like(bufs) for _ in range(world)]
        dist.all_gather(got, bufs)
        assert torch.equal(got[0], got[1]), "buffers differ after broadcast_buffers"
        assert (rank != 0) or torch.equal(bufs, mine0), "rank 0's buffers must win"
        assert torch.equal(pre_params, torch.cat([t.detach().reshape(-1).double() for t in net.parameters()]))
        if rank == 0:
            out.put("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_replicas_identical_without_common_seed_world2_gloo():
    """ranks are NOT seeded identically here; the engines must broadcast rank 0's
    parameters and buffers (nn.DataParallel semantics, train_new_task_step2.py:474-475)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_replicas, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(280)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) == "ok"


def _worker_global_ce(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import mdil_ss_amd  # noqa: F401
        from mdil_ss_amd.engine import global_weighted_ce
        from mdil_ss_amd.models.erfnet_RA_parallel import Net
        from oracle import fixtures as fx
        from oracle import rap_oracle as O

        torch.manual_seed(0)
        names = [n for n, _ in Net([20, 20], 2, 1).named_parameters()]
        torch.manual_seed(0)
        base = {k: v.clone() for k, v in Net([20, 20], 2, 1).state_dict().items()}
        torch.manual_seed(1)
        tsd = {k: v.clone() for k, v in Net([20], 1, 0).state_dict().items()}
        weight = torch.tensor(fx.WEIGHT_BDD)
        shards = [fx.make_batch(1, 16, 32, 20, seed=70 + r) for r in range(world)]
        masks = [(O.draw_dropout_masks(1, torch.Generator().manual_seed(r)),
                  O.draw_dropout_masks(1, torch.Generator().manual_seed(10 + r))) for r in range(world)]

        def fresh():
            sd = {k: v.clone() for k, v in base.items()}
            for n in names:
                sd[n].requires_grad_(O.step2_trainable("module." + n, 1))
            return sd

        # --- this rank: its shard, local CE scaled by global_weighted_ce, + lambda * KLD ---
        sd = fresh()
        img, lab = shards[rank]
        out_new = O.net_forward(sd, img, 1, True, masks[rank][0])
        out_prev = O.net_forward(sd, img, 0, True, masks[rank][1])
        with torch.no_grad():
            out_t = O.net_forward({k: v.clone() for k, v in tsd.items()}, img, 0, False)
        ce = global_weighted_ce(O.ce2d(out_new, lab[:, 0], weight), lab[:, 0], weight)
        (ce + 0.1 * O.kld_prob(out_prev, out_t)).backward()
        flat = torch.cat([sd[n].grad.reshape(-1) for n in names if sd[n].grad is not None])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= world                                   # what GradExchange + Adam's 1/world do
        # --- the DataParallel computation: replicas forward their shards (per-replica BN), the
        # logits are gathered, ONE weighted CE / ONE KLD mean over the whole batch ---
        sd = fresh()
        news, prevs, ts, labs = [], [], [], []
        for r in range(world):
            news.append(O.net_forward(sd, shards[r][0], 1, True, masks[r][0]))
            prevs.append(O.net_forward(sd, shards[r][0], 0, True, masks[r][1]))
            with torch.no_grad():
                ts.append(O.net_forward({k: v.clone() for k, v in tsd.items()}, shards[r][0], 0, False))
            labs.append(shards[r][1][:, 0])
        total = O.ce2d(torch.cat(news), torch.cat(labs), weight) + \
            0.1 * O.kld_prob(torch.cat(prevs), torch.cat(ts))
        total.backward()
        want = torch.cat([sd[n].grad.reshape(-1) for n in names if sd[n].grad is not None])
        torch.testing.assert_close(flat, want, rtol=2e-4, atol=1e-7)
        if rank == 0:
            out.put("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_global_weighted_ce_equals_dataparallel_loss_world2_gloo():
    """--dp-global-batch: rank-averaged gradients of the rescaled local losses == gradients of
    nn.DataParallel's single weighted mean over the gathered batch (train_new_task_step2.py:285-301)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_global_ce, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(280)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) == "ok"


def test_bench_gpus_flag_launches_that_many_ranks():
    """`python bench.py --gpus 2` without a launcher must BECOME two ranks (one process per GPU,
    train_new_task_step2.py:474-475) and report the process group's size.  Driven here with the
    gloo backend and the stand-in step (--stub-cpu): launch, barrier-bracketed timing,
    max-over-ranks and the single JSON line of rank 0 are the shipped code."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "3",
                        "--warmup", "1", "--stub-cpu"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout            # ONE line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    # 4 steps (1 warm-up + 3 timed) of g += rank + 1 followed by a SUM all-reduce over both ranks
    g = 0.0
    for _ in range(4):
        g = 2 * g + 3.0
    assert out["checksum"] == g
    # a launcher that already provides the ranks is not wrapped again
    p1 = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1", "--steps", "2",
                         "--warmup", "0", "--stub-cpu"], capture_output=True, text=True, timeout=300, env=env)
    assert p1.returncode == 0 and json.loads(p1.stdout.strip().splitlines()[-1])["n_gpus"] == 1
