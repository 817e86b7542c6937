"""Shared helpers for the parity tests: rebuild the golden scenario (tools/gen_golden.py) from
seeds + the oracle, without the reference."""
import numpy as np
import torch

from oracle import fixtures as fx
from oracle import rap_oracle as O


def seeded_state(num_classes, nb_tasks, seed, factory=None):
    """State dict with the reference's seed-``seed`` initial values.  ``factory`` builds a module
    whose construction order mirrors the reference (the product model); default = product Net."""
    if factory is None:
        import mdil_ss_amd  # noqa: F401
        from mdil_ss_amd.models.erfnet_RA_parallel import Net as factory
    torch.manual_seed(seed)
    net = factory(num_classes, nb_tasks, nb_tasks - 1)
    return {k: v.detach().clone() for k, v in net.state_dict().items()}


def golden_scenario(golden):
    """-> (teacher_state, student_state) exactly as tools/gen_golden.py prepared them."""
    teacher = seeded_state([20], 1, 1)
    fx.perturb_bn(teacher, seed=11)
    student = seeded_state([20, 20], 2, 0)
    new = O.student_init_from_teacher(teacher, student, 1)
    for k, v in new.items():
        student[k].copy_(v)
    g = torch.Generator().manual_seed(12)
    for k, v in student.items():
        if ".1.running_mean" in k and "encoder" in k:
            v.copy_(0.05 * torch.randn(v.shape, generator=g))
    return teacher, student


def golden_masks(golden, it, n_masks=13):
    new = [torch.from_numpy(golden[f"it{it}_mask{j}"])[:, :, None, None] for j in range(n_masks)]
    old = [torch.from_numpy(golden[f"it{it}_mask{j}"])[:, :, None, None]
           for j in range(n_masks, 2 * n_masks)]
    return new, old


def digest_rows(tensors):
    rows = []
    for p in tensors:
        if p is None:
            rows.append(np.full(67, np.nan))
            continue
        p = p.detach().cpu()
        d = fx.tensor_digest(p).numpy()
        rows.append(d if p.numel() >= 64 else np.pad(d, (0, 64 - p.numel())))
    return np.stack(rows)


def zero_grad_bias(name):
    """Conv biases that are immediately followed by a train-mode BatchNorm: BN subtracts the batch
    mean, so d loss / d bias == 0 analytically (models/erfnet_RA_parallel.py:23-24,95-100,105-109,
    159-160)."""
    if not name.endswith(".bias"):
        return False
    return ("conv1x3" in name) or ("parallel_conv" in name) or name.endswith(".conv.bias")
