"""Worker of tests/test_dp_gpu.py::test_two_ranks_on_one_device: one of two processes that share
cuda:0 and run Step2Engine for three iterations on DIFFERENT batches through a real 2-rank process
group (train_new_task_step2.py:474-475 replaced by engine.GradExchange): the decoder bucket and the
depth-staged shared-encoder buckets start from the backward hooks on the communication stream, the
compute stream joins before the fused Adam.  RCCL is tried first; it refuses two ranks on one
device ("Duplicate GPU detected"), in which case the same engine code runs over gloo with a staging
exchange (device bucket -> host all-reduce -> device bucket, enqueued on the same communication
stream with the same event waits).  Each rank writes what it saw to <out>/rank<r>.pt."""
import argparse
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import datetime
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--backend", default="gloo", choices=["nccl", "gloo"])
    ap.add_argument("--probe", action="store_true",
                    help="only find out whether RCCL accepts two ranks on one device (the parent bounds the time)")
    a = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)                    # BOTH ranks on the one GPU of the box
    dev = torch.device("cuda:0")
    tmo = datetime.timedelta(seconds=60)
    if a.probe:
        msg = "ok"
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
            t = torch.ones(4, device=dev)
            dist.all_reduce(t)                  # communicator creation happens here
            torch.cuda.synchronize()
            assert float(t[0]) == world
        except Exception as e:                  # noqa: BLE001 -- RCCL: duplicate GPU
            msg = f"{type(e).__name__}: {str(e)[:400]}"
        with open(os.path.join(a.out, f"probe{rank}.txt"), "w") as f:
            f.write(msg)
        os._exit(0)                             # no teardown of a possibly broken communicator
    backend, why = a.backend, ""
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)

    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import engine, ops
    from mdil_ss_amd import train_new_task_step2 as T
    from mdil_ss_amd.models.erfnet_RA_parallel import Net
    from oracle import fixtures as fx

    if backend == "gloo":
        # gloo reduces host memory: stage each bucket through the host ON the communication stream, so
        # the hooks / events / join of the shipped exchange are exercised unchanged
        class StagedExchange(engine.GradExchange):
            def start(self, bucket, after=(), pre=None):
                if self.comm_stream is None:
                    self.comm_stream = torch.cuda.Stream()
                self.comm_stream.wait_stream(torch.cuda.current_stream())
                for ev in after:
                    self.comm_stream.wait_event(ev)
                with torch.cuda.stream(self.comm_stream):
                    if pre is not None:
                        pre()
                    host = bucket.detach().to("cpu", non_blocking=False)     # syncs the comm stream only
                    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.pg)
                    bucket.copy_(host.to(dev, non_blocking=False))
        engine.GradExchange = StagedExchange

        def staged_broadcast(modules, process_group=None, src=0):
            with torch.no_grad():
                for m in modules:
                    if m is None:
                        continue
                    for t in list(m.parameters()) + list(m.buffers()):
                        h = t.detach().cpu()
                        dist.broadcast(h, src=src, group=process_group)
                        t.copy_(h)
            ops.refresh_packs()
        engine.broadcast_replicas = staged_broadcast

    torch.manual_seed(100 + rank)               # different seeds: the replicas must still end up identical
    student, teacher = Net([20, 20], 2, 1), Net([20], 1, 0)
    student.to(dev)
    teacher.to(dev)
    T.current_task = 1
    T.apply_step2_freeze(student, teacher, 1)
    eng = engine.Step2Engine(student, teacher, torch.tensor(fx.WEIGHT_BDD, device=dev), current_task=1,
                             lambdac=0.1, is_shared=T.is_shared, is_ds_curr=T.is_DS_curr)
    assert eng.world == 2
    starts = []
    real_start = eng.exchange.start

    def counting_start(bucket, after=(), pre=None):
        starts.append(int(bucket.numel()))
        return real_start(bucket, after=after, pre=pre)
    eng.exchange.start = counting_start
    losses = []
    for it in range(3):                          # iteration 1: one stream; 2 and 3: three streams + staged buckets
        img, lab = fx.make_batch(2, 64, 128, 20, seed=1000 + 10 * it + rank)     # this rank's shard
        _, ce, kld = eng.iteration(img.to(dev), lab.to(dev))
        losses.append((float(ce), float(kld)))
    torch.cuda.synchronize()
    flat = eng.optimizer.flat_param.detach().cpu()
    bufs = torch.cat([b.detach().float().reshape(-1).cpu() for b in student.buffers()])
    torch.save({"backend": backend, "why": why, "flat": flat, "bufs": bufs, "losses": losses,
                "starts": starts, "multi_stream": bool(getattr(eng, "multi_stream", False)),
                "stages": [list(r) for _, r in eng.shared_stages]},
               os.path.join(a.out, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
