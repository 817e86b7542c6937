#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING THE REFERENCE (build container only).

The reference (prachigarg23/MDIL-SS, mounted read-only at /root/reference) is pure Python on
torch; its model / metric modules import cleanly, its trainer imports with four stub modules
(torchvision, config_task, torch.utils.tensorboard, torchsummary -- SURVEY.md 8c).  This script
drives the reference's own ``Net``, ``CrossEntropyLoss2d``, ``is_shared``/``is_DS_curr``,
``iouEval`` + stock ``torch.optim.Adam`` / ``LambdaLR`` / ``KLDivLoss`` exactly the way
``train_new_task_step2.py:202-245,285-306`` does, on seeded tiny inputs, and dumps arrays only.
No reference source text enters the repository.

    python tools/gen_golden.py            # writes tests/golden/step2_tiny.npz, iou.npz, layout.npz
"""
import os
import sys
import types
import importlib

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, REPO)
from oracle import fixtures as fx  # noqa: E402

torch.set_num_threads(8)
torch.use_deterministic_algorithms(True)


def import_reference():
    sys.path.insert(0, REF)
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    for n in ["Compose", "CenterCrop", "Normalize", "Resize", "Pad", "ToTensor", "ToPILImage"]:
        setattr(tvt, n, type(n, (), {"__init__": lambda self, *a, **k: None}))
    tv.transforms = tvt
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = object
    ts = types.ModuleType("torchsummary")
    ts.summary = lambda *a, **k: None
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt,
                        "config_task": types.ModuleType("config_task"),
                        "torch.utils.tensorboard": tb, "torchsummary": ts})
    model_mod = importlib.import_module("models.erfnet_RA_parallel")
    trainer = importlib.import_module("train_new_task_step2")
    iou_mod = importlib.import_module("iouEval")
    return model_mod, trainer, iou_mod


class RecordingDropout(torch.nn.Module):
    """Stands in for ``blk.dropout`` on a reference *instance*: draws the mask the way
    nn.Dropout2d does on CPU and records it, so the very same masks can be replayed elsewhere."""

    def __init__(self, p, sink):
        super().__init__()
        self.p = p
        self.sink = sink

    def forward(self, x):
        if not self.training:
            return x
        m = torch.empty(x.shape[0], x.shape[1], 1, 1).bernoulli_(1 - self.p).div_(1 - self.p)
        self.sink.append(m.clone())
        return x * m


def strip(sd):
    return {k[len("module."):] if k.startswith("module.") else k: v for k, v in sd.items()}


def main():
    model_mod, trainer, iou_mod = import_reference()
    out_dir = os.path.join(REPO, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    G = {}

    # ---------------------------------------------------------------- models + init digests
    torch.manual_seed(1)
    teacher = model_mod.Net([20], 1, 0)
    torch.manual_seed(0)
    student = model_mod.Net([20, 20], 2, 1)
    t_sd, s_sd = teacher.state_dict(), student.state_dict()
    G["teacher_keys"] = np.array(list(t_sd.keys()))
    G["student_keys"] = np.array(list(s_sd.keys()))
    G["teacher_shapes"] = np.array([str(tuple(v.shape)) for v in t_sd.values()])
    G["student_shapes"] = np.array([str(tuple(v.shape)) for v in s_sd.values()])
    G["teacher_init_digest"] = np.stack([fx.tensor_digest(v, 8).numpy()[:11] if v.numel() >= 8
                                         else np.pad(fx.tensor_digest(v, 8).numpy(), (0, 11 - 3 - v.numel()))
                                         for v in t_sd.values()])
    G["student_init_digest"] = np.stack([fx.tensor_digest(v, 8).numpy()[:11] if v.numel() >= 8
                                         else np.pad(fx.tensor_digest(v, 8).numpy(), (0, 11 - 3 - v.numel()))
                                         for v in s_sd.values()])

    # "trained-looking" step-1 checkpoint: perturb BN affine + running stats of the teacher
    fx.perturb_bn(t_sd, seed=11)          # state_dict tensors alias the module's -> in place
    ckpt = {"module." + k: v.clone() for k, v in t_sd.items()}     # DataParallel naming (:474)

    # ---------------------------------------------------------------- student init (a19)
    # restated from train_new_task_step2.py:497-530 on ``module.``-prefixed dicts, then loaded
    # through the reference model's own load_state_dict(strict=False).
    import re
    t = 1
    s_keys = {"module." + k for k in s_sd}
    new = {k: v for k, v in ckpt.items() if k in s_keys}
    for k, v in ckpt.items():
        if "encoder" in k:
            if "parallel_conv" in k or "bn" in k:
                if ".{}.weight".format(t - 1) in k:
                    new[re.sub(".{}.weight".format(t - 1), ".{}.weight".format(t), k)] = v
                elif ".{}.bias".format(t - 1) in k:
                    new[re.sub(".{}.bias".format(t - 1), ".{}.bias".format(t), k)] = v
        elif "decoder" in k and "output_conv" not in k:
            new[re.sub("decoder.{}".format(t - 1), "decoder.{}".format(t), k)] = v
    G["init_loaded_keys"] = np.array(sorted(new.keys()))
    student.load_state_dict(strip(new), strict=False)
    # give the new-domain BN running stats non-trivial values too (they are NOT copied by the rule)
    g = torch.Generator().manual_seed(12)
    for k, v in student.state_dict().items():
        if ".1.running_mean" in k and "encoder" in k:
            v.copy_(0.05 * torch.randn(v.shape, generator=g))
    student_start = {k: v.clone() for k, v in student.state_dict().items()}

    # ---------------------------------------------------------------- trainer predicates
    trainer.current_task = 1
    names = ["module." + n for n, _ in student.named_parameters()]
    G["param_names"] = np.array(names)
    G["is_shared"] = np.array([bool(trainer.is_shared(n)) for n in names])
    G["is_ds_curr"] = np.array([bool(trainer.is_DS_curr(n)) for n in names])

    # freeze rule :202-215 (on the names as the trainer sees them)
    for p in teacher.parameters():
        p.requires_grad = False
    for name, m in student.named_parameters():
        name = "module." + name
        if "decoder" in name:
            if "decoder.{}".format(1) not in name:
                m.requires_grad = False
        elif "encoder" in name:
            if "bn" in name or "parallel_conv" in name:
                if ".{}.weight".format(1) in name or ".{}.bias".format(1) in name:
                    continue
                m.requires_grad = False
    G["requires_grad"] = np.array([p.requires_grad for _, p in student.named_parameters()])

    params = [("module." + n, p) for n, p in student.named_parameters()]
    groups = [{"params": [p for n, p in params if trainer.is_shared(n)], "lr": 5e-6},
              {"params": [p for n, p in params if trainer.is_DS_curr(n)]}]
    optimizer = torch.optim.Adam(groups, 5e-4, (0.9, 0.999), eps=1e-8, weight_decay=1e-4)
    num_epochs = 150
    sched = torch.optim.lr_scheduler.LambdaLR(
        optimizer, lr_lambda=lambda e: pow((1 - ((e - 1) / num_epochs)), 0.9))
    lrs = []
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for e in (1, 2, 75, 150):
            sched.step(e)
            lrs.append([g_["lr"] for g_ in optimizer.param_groups])
        sched.step(2)                      # run the two golden iterations at epoch-2 LR
    G["lr_epochs"] = np.array([1, 2, 75, 150])
    G["lr_values"] = np.array(lrs, dtype=np.float64)

    # ---------------------------------------------------------------- two hot-loop iterations
    N, H, W = 2, 32, 64
    weight = torch.tensor(fx.WEIGHT_BDD)
    criterion = trainer.CrossEntropyLoss2d(weight)
    kl = torch.nn.KLDivLoss()
    sink = []
    for li, blk in enumerate(student.encoder.layers):
        if hasattr(blk, "dropout"):
            blk.dropout = RecordingDropout(blk.dropout.p, sink)
    student.train()
    teacher.eval()
    acts = {}
    hooks = []

    def mk(name):
        def hook(_m, _i, o):
            acts[name] = o.detach().clone()
        return hook

    hooks.append(student.encoder.initial_block.register_forward_hook(mk("encoder.initial_block")))
    for li, blk in enumerate(student.encoder.layers):
        hooks.append(blk.register_forward_hook(mk(f"encoder.layers.{li}")))
    for li, blk in enumerate(student.decoder[1].layers):
        hooks.append(blk.register_forward_hook(mk(f"decoder.1.layers.{li}")))

    torch.manual_seed(1234)
    for it in range(2):
        images, labels = fx.make_batch(N, H, W, 20, seed=100 + it)
        G[f"it{it}_images"] = images.numpy()
        G[f"it{it}_labels"] = labels.numpy()
        del sink[:]
        acts.clear()
        outputs = student(images, 1)
        if it == 0:
            for h_ in hooks:
                h_.remove()
            for k, v in acts.items():
                G["it0_act_" + k] = v.numpy()
        outputs_prev_task = student(images, 0)
        outputs_prev_model = teacher(images, 0)
        ce = criterion(outputs, labels[:, 0])
        kld = kl(torch.nn.functional.softmax(outputs_prev_task, dim=1),
                 torch.nn.functional.softmax(outputs_prev_model, dim=1))
        total = ce + 0.1 * kld
        optimizer.zero_grad()
        total.backward()
        for j, m in enumerate(sink):
            G[f"it{it}_mask{j}"] = m.reshape(m.shape[0], m.shape[1]).numpy()
        G[f"it{it}_logits_new"] = outputs.detach().numpy()
        G[f"it{it}_logits_prev_task"] = outputs_prev_task.detach().numpy()
        G[f"it{it}_logits_prev_model"] = outputs_prev_model.detach().numpy()
        G[f"it{it}_losses"] = np.array([ce.item(), kld.item(), total.item()], dtype=np.float64)
        gd, full = [], {}
        for n_, p in student.named_parameters():
            if p.grad is None:
                gd.append(np.full(67, np.nan))
            else:
                gd.append(fx.tensor_digest(p.grad).numpy() if p.numel() >= 64 else
                          np.pad(fx.tensor_digest(p.grad).numpy(), (0, 64 - p.numel())))
                if p.numel() <= 4096 or "layers.1.conv3x1_1.weight" in n_ \
                        or "layers.14.conv1x3_2.weight" in n_ or "layers.9.parallel_conv_2.1.weight" in n_:
                    full[n_] = p.grad.detach().numpy().copy()
        G[f"it{it}_grad_digest"] = np.stack(gd)
        for n_, v in full.items():
            G[f"it{it}_grad_{n_}"] = v
        optimizer.step()
        pd = []
        for n_, p in student.named_parameters():
            pd.append(fx.tensor_digest(p).numpy() if p.numel() >= 64 else
                      np.pad(fx.tensor_digest(p).numpy(), (0, 64 - p.numel())))
        G[f"it{it}_param_digest"] = np.stack(pd)
        for k, v in student.state_dict().items():
            if fxbuf(k):
                G[f"it{it}_buf_{k}"] = v.numpy().copy()
    for k, v in student_start.items():
        if fxbuf(k) or ("bn" in k) or k.endswith(".bias"):
            G["start_" + k] = v.numpy()
    for k, v in ckpt.items():
        kk = k[len("module."):]
        if fxbuf(kk) or ("bn" in kk):
            G["teacher_" + kk] = v.numpy()

    # eval-mode student forward on both heads (for the eval path)
    student.eval()
    with torch.no_grad():
        images, _ = fx.make_batch(N, H, W, 20, seed=100)
        G["eval_logits_task1"] = student(images, 1).numpy()
        G["eval_logits_task0"] = student(images, 0).numpy()

    np.savez_compressed(os.path.join(out_dir, "step2_tiny.npz"), **G)
    print("step2_tiny.npz:", len(G), "arrays,",
          os.path.getsize(os.path.join(out_dir, "step2_tiny.npz")) / 1e6, "MB")
    print("losses it0:", G["it0_losses"], "it1:", G["it1_losses"])

    # ---------------------------------------------------------------- iouEval golden
    I = {}
    g = torch.Generator().manual_seed(7)
    pred = torch.randint(0, 20, (3, 1, 16, 24), generator=g)
    targ = torch.randint(0, 20, (3, 1, 16, 24), generator=g)
    ev = iou_mod.iouEval(20, 19)
    ev.addBatch(pred, targ)
    ev.addBatch(targ, targ)
    m, per = ev.getIoU()
    I.update(pred=pred.numpy(), targ=targ.numpy(), tp=ev.tp.numpy(), fp=ev.fp.numpy(),
             fn=ev.fn.numpy(), miou=np.array(m.item()), per=per.numpy())
    pred27 = torch.randint(0, 27, (2, 1, 24, 24), generator=g)
    targ27 = torch.randint(0, 27, (2, 1, 24, 24), generator=g)
    ev = iou_mod.iouEval(27, 26)
    ev.addBatch(pred27, targ27)
    m, per = ev.getIoU()
    I.update(pred27=pred27.numpy(), targ27=targ27.numpy(), tp27=ev.tp.numpy(), fp27=ev.fp.numpy(),
             fn27=ev.fn.numpy(), miou27=np.array(m.item()))
    np.savez_compressed(os.path.join(out_dir, "iou.npz"), **I)
    print("iou.npz written; mIoU", I["miou"], I["miou27"])


def fxbuf(k):
    return k.endswith(("running_mean", "running_var", "num_batches_tracked"))


if __name__ == "__main__":
    main()
