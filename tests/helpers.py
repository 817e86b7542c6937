"""Shared helpers for the parity tests: rebuild the golden scenario (tools/gen_golden.py) from
seeds + the oracle, without the reference."""
import numpy as np
import torch

from oracle import fixtures as fx
from oracle import rap_oracle as O


def host_threads(cap=64):
    """CPU threads this process may really use: the cgroup CPU quota when there is one (the GPU
    box shows 256 logical CPUs but grants 16 CPUs' worth of time), else the visible CPU count."""
    import os
    n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


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


# ------------------------------------------------------------------------------------- step 3
WEIGHT_IDD = [3.235635601598852, 6.76221624390441, 9.458242359884549, 9.446818215454014,
              9.947040673126763, 9.789672819856547, 9.476665808564432, 10.465565126694731,
              9.59189547383129, 7.637805282159825, 8.990899026692638, 9.26222234098628,
              10.265657138809514, 9.386517631614392, 8.357391489170013, 9.910382864314824,
              10.389977663948363, 8.997422571963602, 10.418070541191673, 10.483262606962834,
              9.511436923349441, 7.597725385711079, 6.1734896019878205, 9.787631041755187,
              3.9178330193378708, 4.417448652936843, 0.0]


def step3_scenario():
    """-> (teacher_state [20,20], student_state [20,20,27]) as tools/gen_golden_step3.py made them."""
    teacher = seeded_state([20, 20], 2, 1)
    fx.perturb_bn(teacher, seed=21)
    student = seeded_state([20, 20, 27], 3, 0)
    for k, v in O.student_init_from_teacher(teacher, student, 2).items():
        student[k].copy_(v)
    return teacher, student


def step3_masks(g3):
    out = {}
    for key in ("new", "prev1", "prev0", "teach1", "teach0"):
        arr = g3["mask_" + key]                      # [13, N, 128] zero-padded on the channel axis
        widths = [64] * 5 + [128] * 8
        out[key] = [torch.from_numpy(arr[j][:, :w].copy())[:, :, None, None] for j, w in enumerate(widths)]
    return out


# --------------------------------------------------------------------------------- multi-task
def mt_scenario():
    """-> state dict of the multi-task model [20, 27] as tools/gen_golden_mt.py prepared it."""
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd.models.erfnet_multi_task import Net
    sd = seeded_state([20, 27], 2, 0, factory=Net)
    fx.perturb_bn(sd, seed=31)
    return sd


def mt_masks(gm, ind):
    widths = [64] * 5 + [128] * 8
    arr = gm[f"mask{ind}"]
    return [torch.from_numpy(arr[j][:, :w].copy())[:, :, None, None] for j, w in enumerate(widths)]


# ------------------------------------------------------------------ fine-tuning baseline (ftp2)
def ft_scenario():
    """-> state dict of models/erfnet_ftp2.Net(20, 20, 27) as tools/gen_golden_ft.py prepared it."""
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd.models.erfnet import NetFT2
    torch.manual_seed(0)
    sd = {k: v.detach().clone() for k, v in NetFT2(20, 20, 27).state_dict().items()}
    fx.perturb_bn(sd, seed=41)
    return sd


def ft_masks(gf):
    widths = [64] * 5 + [128] * 8
    return [torch.from_numpy(gf["mask"][j][:, :w].copy())[:, :, None, None] for j, w in enumerate(widths)]


def kernel_build_id():
    """Identity of the NUMERICS of the product path: sha1 over the device / C-ABI sources
    (mdil_ss_amd/csrc/*.hip, *.cpp, *.h and include/mdil_hip.h) with comments and whitespace removed,
    plus the HOST code that decides which launches run in which order -- ops.py, engine.py, _lib.py and
    the model containers -- as the dump of its syntax tree without docstrings (a comment or docstring
    edit keeps the id, any change of code does not; rounds 3-4 relied on a hand-bumped
    ``ops.NUMERICS_EPOCH`` for this half).
    Recorded mIoU samples carry the id of the build that produced them (tools/miou_hip_sample.py);
    tests/test_miou_parity.py uses only the samples of the build under test."""
    import ast
    import glob
    import hashlib
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "mdil_ss_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(root, "mdil_ss_amd", "csrc", "*.cpp")) +
                   glob.glob(os.path.join(root, "mdil_ss_amd", "csrc", "*.h")) +
                   [os.path.join(root, "include", "mdil_hip.h")])
    h = hashlib.sha1()
    for f in files:
        t = open(f).read()
        t = re.sub(r"/\*.*?\*/", "", t, flags=re.S)
        t = re.sub(r"//[^\n]*", "", t)
        t = re.sub(r"\s+", "", t)
        h.update(os.path.basename(f).encode() + b"\0" + t.encode() + b"\0")
    host = [os.path.join(root, "mdil_ss_amd", n) for n in ("ops.py", "engine.py", "_lib.py")] + \
        sorted(glob.glob(os.path.join(root, "mdil_ss_amd", "models", "*.py")))
    for f in host:
        tree = ast.parse(open(f).read())
        for node in ast.walk(tree):
            body = getattr(node, "body", None)
            if isinstance(body, list) and body and isinstance(body[0], ast.Expr) and \
                    isinstance(getattr(body[0], "value", None), ast.Constant) and isinstance(body[0].value.value, str):
                body.pop(0)
                if not body:
                    body.append(ast.Pass())
        h.update(os.path.basename(f).encode() + b"\0" + ast.dump(tree, annotate_fields=False).encode() + b"\0")
    return h.hexdigest()[:12]
