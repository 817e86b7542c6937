import gc, os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
import mdil_ss_amd
from mdil_ss_amd import ops
from mdil_ss_amd.engine import Step2Engine
dev = torch.device("cuda:0")
pool = []
for i in range(4):
    g = torch.Generator().manual_seed(1234 + i)
    img = torch.rand(6, 3, 512, 1024, generator=g)
    lab = torch.randint(0, 20, (6, 1, 32, 64), generator=g).repeat_interleave(16, 2).repeat_interleave(16, 3).contiguous()
    pool.append((img.to(dev), lab.to(dev)))

def run(tag, stagger, cleanup):
    student, teacher, T = bench.build_models(dev)
    T.current_task = 1
    eng = Step2Engine(student, teacher, torch.tensor(bench.WEIGHT_BDD, device=dev), current_task=1, lambdac=0.1,
                      is_shared=T.is_shared, is_ds_curr=T.is_DS_curr)
    eng.stagger = stagger
    eng.optimizer.set_epoch(1, 150)
    for i in range(8):
        eng.iteration(*pool[i % 4])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(30):
        eng.iteration(*pool[i % 4])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30 * 1e3
    print(f"{tag}: {dt:.2f} ms/step  pack cache {len(ops._pack_cache)} jobs {len(ops._pack_jobs)} templates {len(ops._nb_templates)} "
          f"side streams {len(ops._side_streams)} tickets {len(ops._tickets)} ws {len(ops._ws)} mem {torch.cuda.memory_allocated() / 2**30:.1f} GiB", flush=True)
    del eng, student, teacher
    if cleanup:
        gc.collect()
        ops.invalidate_packs()
        torch.cuda.empty_cache()

mode = sys.argv[1]
run("engine 1 (stagger 8)", 8, mode == "clean")
run("engine 2 (lock step)", None, mode == "clean")
run("engine 3 (stagger 8)", 8, mode == "clean")
