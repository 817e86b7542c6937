#!/usr/bin/env python3
"""GPU diagnostic: per-layer forward error of the student new-task forward vs golden activations."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as Hh
from tests.test_model_golden import _build

golden = np.load("tests/golden/step2_tiny.npz")
dev = torch.device("cuda:0")
student, teacher = _build(golden, dev)
m_new, m_old = Hh.golden_masks(golden, 0)
q = [m_new]
student.mask_provider = lambda n: q.pop(0)
student.train()
x = torch.from_numpy(golden["it0_images"]).to(dev).permute(0, 2, 3, 1).contiguous()
masks = student.draw_masks(2, dev)
enc = student.encoder
def rep(name, y):
    ref = torch.from_numpy(golden["it0_act_" + name])
    got = y.detach().permute(0, 3, 1, 2).cpu()
    err = (got - ref).abs()
    flips = int(((got > 0) != (ref > 0)).sum())
    print(f"{name:28s} max|ref| {float(ref.abs().max()):9.3e}  max err {float(err.max()):9.3e}  "
          f"rel {float(err.max() / ref.abs().max()):9.3e}  sign flips {flips}/{ref.numel()}")
y = enc.initial_block.run(x, 1, True); rep("encoder.initial_block", y)
k = 0
for li, layer in enumerate(enc.layers):
    if hasattr(layer, "bn_ini"):
        y = layer.run(y, 1, True)
    else:
        y = layer.run(y, 1, True, masks[k]); k += 1
    rep(f"encoder.layers.{li}", y)
for li, layer in enumerate(student.decoder[1].layers):
    y = layer.run(y, 0, True); rep(f"decoder.1.layers.{li}", y)
