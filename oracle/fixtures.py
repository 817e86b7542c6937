"""Seeded synthetic inputs shared by ``tools/gen_golden.py`` and the tests.  TEST INFRASTRUCTURE
ONLY (see oracle/rap_oracle.py header).  No reference code is needed to run anything here."""
from __future__ import annotations

from typing import Dict

import torch

# weight_BDD of train_new_task_step2.py:125-127 (data, copied as constants) with [19]=0 (:134)
WEIGHT_BDD = [3.6525147483016243, 8.799815287822142, 4.781908267406055, 10.034828238618045,
              9.5567865464289, 9.645099012085169, 10.315292989325766, 10.163473632969513,
              4.791692009441432, 9.556915153488912, 4.142994047786311, 10.246903827488143,
              10.47145010979545, 6.006704177894196, 9.60620532303246, 9.964959813857726,
              10.478333987902301, 10.468010534454706, 10.440929141422366, 0.0]


def perturb_bn(state: Dict[str, torch.Tensor], seed: int) -> None:
    """Make a random-init state dict look like a trained checkpoint: every BN gets non-trivial
    affine parameters and running statistics (deterministic in ``seed`` and key order)."""
    g = torch.Generator().manual_seed(seed)
    for k in state:
        v = state[k]
        if k.endswith("running_mean"):
            v.copy_(0.1 * torch.randn(v.shape, generator=g))
        elif k.endswith("running_var"):
            v.copy_(0.5 + torch.rand(v.shape, generator=g))
        elif ("bn" in k) and k.endswith(".weight"):
            v.copy_(1.0 + 0.1 * torch.randn(v.shape, generator=g))
        elif ("bn" in k) and k.endswith(".bias"):
            v.copy_(0.1 * torch.randn(v.shape, generator=g))


def make_batch(n: int, h: int, w: int, n_classes: int, seed: int, block: int = 4):
    """images f32[n,3,h,w] in [0,1) (range of ToTensor, train_new_task_step2.py:76) and
    block-constant labels i64[n,1,h,w] in [0, n_classes) (class n_classes-1 = ignore)."""
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(n, 3, h, w, generator=g)
    lab = torch.randint(0, n_classes, (n, 1, (h + block - 1) // block, (w + block - 1) // block),
                        generator=g)
    lab = lab.repeat_interleave(block, 2).repeat_interleave(block, 3)[:, :, :h, :w].contiguous()
    return img, lab


def tensor_digest(t: torch.Tensor, n_samples: int = 64):
    """(sum, abs-sum, l2, strided sample) of a tensor in float64 -- a compact fingerprint."""
    f = t.detach().double().reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, min(n_samples, f.numel())).long()
    return torch.cat([torch.stack([f.sum(), f.abs().sum(), f.pow(2).sum().sqrt()]), f[idx]])
