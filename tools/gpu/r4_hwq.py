"""diagnostic: does a 5th / 6th stream (e.g. the communication stream of a multi-rank run) share a hardware queue with
the engine's streams, and does GPU_MAX_HW_QUEUES change that?  Dummy streams are created (and touched) first."""
import os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
import mdil_ss_amd
from mdil_ss_amd.engine import Step2Engine
dev = torch.device("cuda:0")
n_dummy = int(sys.argv[1])
dummies = [torch.cuda.Stream(priority=-1 if i % 2 else 0) for i in range(n_dummy)]
for st in dummies:
    with torch.cuda.stream(st):
        torch.zeros(16, device=dev).add_(1)
torch.cuda.synchronize()
pool = []
for i in range(4):
    g = torch.Generator().manual_seed(1234 + i)
    img = torch.rand(6, 3, 512, 1024, generator=g)
    lab = torch.randint(0, 20, (6, 1, 32, 64), generator=g).repeat_interleave(16, 2).repeat_interleave(16, 3).contiguous()
    pool.append((img.to(dev), lab.to(dev)))
student, teacher, T = bench.build_models(dev)
T.current_task = 1
eng = Step2Engine(student, teacher, torch.tensor(bench.WEIGHT_BDD, device=dev), current_task=1, lambdac=0.1,
                  is_shared=T.is_shared, is_ds_curr=T.is_DS_curr)
eng.optimizer.set_epoch(1, 150)
for i in range(8):
    eng.iteration(*pool[i % 4])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(30):
    eng.iteration(*pool[i % 4])
torch.cuda.synchronize()
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default')}, {n_dummy} streams used before the engine's: "
      f"{(time.perf_counter() - t0) / 30 * 1e3:.2f} ms/step", flush=True)
