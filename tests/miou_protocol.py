"""Shared protocol of the mIoU-parity run: seeded procedural dataset, batch order,
dropout masks, initial states.  Used by tools/gen_miou_golden.py (reference, CPU) and
tests/test_miou_parity.py (HIP path, GPU).  Needs neither the reference nor a GPU by itself."""
import torch

from oracle import fixtures as fx
from oracle import rap_oracle as O

# A protocol whose outcome is a property of the method, not of fp32 rounding: a learnable task
# (well separated class colours + noise, all 19 evaluated classes present), the reference's batch
# size and hyper-parameters, LR schedules that decay to ~0, thousands of iterations (the network
# needs them at lr 5e-4: ~10 % mIoU after 600 iterations whatever the data) and a validation set
# of 512 images (1 M pixels per domain) so that single boundary pixels do not move the metric.
# 32x64 images keep the CPU reference run affordable (11.8 k iterations: ~50 min on 4 cores).
CONFIG = {"height": 32, "width": 64, "batch": 6, "n_train": 384, "n_val": 512, "epochs": 64,
          "epochs_step1": 120,
          "lambdac": 0.1, "n_rects": 6, "noise": 0.03, "classes_used": 19, "palette": "grid"}


def _dataset(n, seed, domain, n_classes=20):
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "_mdil_dataset", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                      "mdil_ss_amd", "dataset.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.ProceduralSeg(n, CONFIG["height"], CONFIG["width"], n_classes, seed=seed,
                             n_rects=CONFIG["n_rects"], noise=CONFIG["noise"], domain=domain,
                             classes_used=CONFIG["classes_used"], palette=CONFIG["palette"])


_cache = {}


def _items(kind):
    if kind not in _cache:
        ds = _dataset(CONFIG["n_train"] if kind.startswith("train") else CONFIG["n_val"],
                      {"train": 21, "val_new": 22, "val_old": 23, "train_old": 24}[kind],
                      domain=0 if kind.endswith("old") else 1)
        _cache[kind] = [ds[i] for i in range(len(ds))]
    return _cache[kind]


def train_batches(epoch, old_domain=False):
    """Batches of one epoch: the new domain (step 2) or, with ``old_domain``, the first domain
    (the step-1 pre-training that produces the teacher)."""
    items = _items("train_old" if old_domain else "train")
    perm = torch.randperm(len(items), generator=torch.Generator().manual_seed(
        (5000 if old_domain else 1000) + epoch)).tolist()
    b = CONFIG["batch"]
    for i in range(0, len(perm) - b + 1, b):
        idx = perm[i:i + b]
        yield torch.stack([items[j][0] for j in idx]), torch.stack([items[j][1] for j in idx])


def val_batches(task):
    items = _items("val_new" if task == 1 else "val_old")
    b = CONFIG["batch"]
    for i in range(0, len(items), b):
        yield (torch.stack([x[0] for x in items[i:i + b]]), torch.stack([x[1] for x in items[i:i + b]]))


def masks_for(iteration, n):
    g = torch.Generator().manual_seed(50000 + iteration)
    return O.draw_dropout_masks(n, g), O.draw_dropout_masks(n, g)


def step1_initial_state():
    """Seeded init of the step-1 model through the PRODUCT model's constructors (bit identical to
    the reference's under the same seed, tests/test_oracle_golden.py)."""
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd.models.erfnet_RA_parallel import Net
    torch.manual_seed(1)
    return {k: v.clone() for k, v in Net([20], 1, 0).state_dict().items()}


def step2_student_state(teacher):
    """Student of step 2: seeded init + the init rule applied to the trained step-1 state."""
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd.models.erfnet_RA_parallel import Net
    torch.manual_seed(0)
    student = {k: v.clone() for k, v in Net([20, 20], 2, 1).state_dict().items()}
    for k, v in O.student_init_from_teacher(teacher, student, 1).items():
        student[k].copy_(v)
    return student


# One-step checks "at a covering size" (tests/test_miou_parity.py): the protocol's 32x64 images put a
# 4x8 map in front of the deepest blocks, where only dilation 2 (and 4 along W) can form complete
# Winograd pairs -- 10 of the 16 dilated C=128 convs of a forward, their dgrads and weight gradients
# run the DIRECT kernels there.  The weights are fully convolutional: the same trained states are
# therefore also stepped once on a 256x512 batch (deepest map 32x64: every dilation 2..16 pairs up on
# both axes), which is what the full-size network launches.
COVER = {"height": 256, "width": 512, "batch": 2}


def covering_batch(seed, old_domain=False):
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "_mdil_dataset", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                      "mdil_ss_amd", "dataset.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ds = mod.ProceduralSeg(COVER["batch"], COVER["height"], COVER["width"], 20, seed=seed,
                           n_rects=4 * CONFIG["n_rects"], noise=CONFIG["noise"], domain=0 if old_domain else 1,
                           classes_used=CONFIG["classes_used"], palette=CONFIG["palette"])
    items = [ds[i] for i in range(len(ds))]
    return torch.stack([x[0] for x in items]), torch.stack([x[1] for x in items])
