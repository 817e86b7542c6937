#!/usr/bin/env python3
"""Context figure: what the reference's OWN stack does on this GPU.

Times the oracle's step-2 graph -- the reference's `nn` graph restated on stock torch ops
(`oracle/rap_oracle.py`: `F.conv2d`, `F.batch_norm`, `F.conv_transpose2d`, ... = ATen / MIOpen /
rocBLAS on PyTorch-ROCm) -- on DEVICE tensors at the bench's configuration (fp32, batch 6,
1024x512, 2 student forwards in train mode + frozen-model forward + CE + lambda KLD + backward +
a stand-in parameter pass).  Test / measurement infrastructure only: it imports `oracle/`, so it
lives in `tools/`, is never imported by the product and never runs inside `bench.py`'s timed
region.  Written to `profiles/r06_stock_rocm_baseline.txt` by `tools/gpu/r6/call1.sh`.

    python tools/stock_rocm_baseline.py [--batch 6] [--iters 10]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fixtures as fx  # noqa: E402
from oracle import rap_oracle as O  # noqa: E402
import mdil_ss_amd  # noqa: E402,F401
from mdil_ss_amd.models.erfnet_RA_parallel import Net  # noqa: E402
from bench import WEIGHT_BDD  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=6)
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--benchmark", action="store_true", help="torch.backends.cudnn.benchmark (MIOpen find mode)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = bool(a.benchmark)
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    torch.manual_seed(1)
    t_sd = {k: v.clone().to(dev) for k, v in Net([20], 1, 0).state_dict().items()}
    torch.manual_seed(0)
    net = Net([20, 20], 2, 1)
    s_sd = {k: v.clone() for k, v in net.state_dict().items()}
    for k, v in O.student_init_from_teacher({k: v.cpu() for k, v in t_sd.items()}, s_sd, 1).items():
        s_sd[k].copy_(v)
    s_sd = {k: v.to(dev) for k, v in s_sd.items()}
    names = [n for n, _ in net.named_parameters()]
    for n in names:
        s_sd[n].requires_grad_(O.step2_trainable("module." + n, 1))
    weight = torch.tensor(WEIGHT_BDD, device=dev)
    gen = torch.Generator(device="cpu")
    gen.manual_seed(7)
    batches = []
    for it in range(2):
        images, labels = fx.make_batch(a.batch, a.height, a.width, 20, seed=it, block=16)
        batches.append((images.to(dev), labels.to(dev)))

    def masks():
        return [m.to(dev) for m in O.draw_dropout_masks(a.batch, gen)]

    def iteration(i):
        images, labels = batches[i % 2]
        for n in names:
            s_sd[n].grad = None
        O.step2_iteration(s_sd, t_sd, images, labels, weight, 1, 0.1, masks(), masks())
        with torch.no_grad():
            for n in names:
                if s_sd[n].grad is not None:
                    s_sd[n].add_(s_sd[n].grad, alpha=-1e-6)   # stand-in for the optimizer's pass

    t0 = time.time()
    for i in range(a.warmup):
        iteration(i)
    torch.cuda.synchronize()
    warm = time.time() - t0
    t0 = time.time()
    for i in range(a.iters):
        iteration(i)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / a.iters
    print(f"stock PyTorch-ROCm {torch.__version__} (ATen / MIOpen, fp32, cudnn.benchmark={a.benchmark}): oracle step-2 "
          f"graph on device tensors, batch {a.batch} at {a.width}x{a.height}: {dt * 1e3:.1f} ms / iteration = "
          f"{a.batch / dt:.1f} img/s  ({a.warmup} warm-up iterations {warm:.1f} s, {a.iters} timed; "
          f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB)", flush=True)


if __name__ == "__main__":
    main()
