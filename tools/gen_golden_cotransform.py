#!/usr/bin/env python3
"""tests/golden/cotransform.npz: the reference's ``MyCoTransform`` (train_new_task_step2.py:48-81,
imported) run on synthetic PIL images.  torchvision is not installed here; the two names the
transform takes from it are bound to their published behaviour (Resize([h,w], interp) ->
PIL resize((w,h), interp); ToTensor -> uint8 HWC -> float CHW / 255) -- everything else (flip,
ImageOps.expand / crop translation, ToLabel, Relabel, the order of the random draws) is the
reference's own code.  Two sets: natural seeded draws (pins the RNG order) and every
(hflip, transX, transY) combination forced through a patched ``random`` (pins the fill rules)."""
import importlib
import os
import random
import sys
import types

import numpy as np
import torch
from PIL import Image

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


class Resize:
    def __init__(self, size, interpolation):
        self.size, self.interpolation = size, interpolation

    def __call__(self, img):
        return img.resize((self.size[1], self.size[0]), self.interpolation)


class ToTensor:
    def __call__(self, img):
        return torch.from_numpy(np.array(img, dtype=np.uint8)).permute(2, 0, 1).float().div(255)


def import_trainer():
    sys.path.insert(0, "/root/reference")
    tv = types.ModuleType("torchvision")
    tvt = types.ModuleType("torchvision.transforms")
    for n in ["Compose", "CenterCrop", "Normalize", "Pad", "ToPILImage"]:
        setattr(tvt, n, type(n, (), {"__init__": lambda self, *a, **k: None}))
    tvt.Resize, tvt.ToTensor = Resize, ToTensor
    tv.transforms = tvt
    tb = types.ModuleType("torch.utils.tensorboard")
    tb.SummaryWriter = object
    ts = types.ModuleType("torchsummary")
    ts.summary = lambda *a, **k: None
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt,
                        "config_task": types.ModuleType("config_task"),
                        "torch.utils.tensorboard": tb, "torchsummary": ts})
    return importlib.import_module("train_new_task_step2")


def source_pair(seed, hs, ws, n_classes):
    g = np.random.default_rng(seed)
    img = g.integers(0, 256, (hs, ws, 3), dtype=np.uint8)
    lab = g.integers(0, n_classes - 1, (hs // 4, ws // 4), dtype=np.uint8).repeat(4, 0).repeat(4, 1)
    lab[g.random((hs, ws)) < 0.05] = 255                   # void pixels
    return img, lab


def main():
    T = import_trainer()
    G = {}
    H, W = 24, 40
    T.NUM_CLASSES = 20
    co = T.MyCoTransform(augment=True, height=H, width=W)
    co_val = T.MyCoTransform(augment=False, height=H, width=W)
    # natural draws
    random.seed(1234)
    st = random.getstate()
    outs_x, outs_y, srcs_i, srcs_l = [], [], [], []
    for i in range(12):
        img, lab = source_pair(100 + i, 52, 92, 20)
        x, y = co(Image.fromarray(img), Image.fromarray(lab).convert("P"))
        srcs_i.append(img), srcs_l.append(lab), outs_x.append(x.numpy()), outs_y.append(y.numpy())
    G["nat_src_img"], G["nat_src_lab"] = np.stack(srcs_i), np.stack(srcs_l)
    G["nat_out_img"], G["nat_out_lab"] = np.stack(outs_x), np.stack(outs_y)
    random.setstate(st)
    G["nat_params"] = np.array([[int(random.random() < 0.5), random.randint(-2, 2), random.randint(-2, 2)]
                                for _ in range(12)], dtype=np.int32)
    # forced combinations (27-class labels: Relabel 255 -> 26)
    T.NUM_CLASSES = 27
    img, lab = source_pair(7, 48, 80, 27)
    G["frc_src_img"], G["frc_src_lab"] = img, lab
    real_random, real_randint = random.random, random.randint
    params, outs_x, outs_y = [], [], []
    try:
        for flip in (0, 1):
            for tx in range(-2, 3):
                for ty in range(-2, 3):
                    q = [tx, ty]
                    random.random = lambda f=flip: 0.25 if f else 0.75
                    random.randint = lambda a, b: q.pop(0)
                    x, y = co(Image.fromarray(img), Image.fromarray(lab).convert("P"))
                    params.append((flip, tx, ty)), outs_x.append(x.numpy()), outs_y.append(y.numpy())
    finally:
        random.random, random.randint = real_random, real_randint
    G["frc_params"] = np.array(params, dtype=np.int32)
    G["frc_out_img"], G["frc_out_lab"] = np.stack(outs_x), np.stack(outs_y)
    x, y = co_val(Image.fromarray(img), Image.fromarray(lab).convert("P"))
    G["val_out_img"], G["val_out_lab"] = x.numpy(), y.numpy()
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "cotransform.npz"), **G)
    print("cotransform.npz", {k: v.shape for k, v in G.items()})


if __name__ == "__main__":
    main()
