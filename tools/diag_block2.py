#!/usr/bin/env python3
"""GPU diagnostic: inside decoder.1.layers.4, compare a1 (HIP) with F.conv2d on the SAME x."""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import helpers as Hh
from tests.test_model_golden import _build
from mdil_ss_amd import ops

golden = np.load("tests/golden/step2_tiny.npz")
dev = torch.device("cuda:0")
student, teacher = _build(golden, dev)
m_new, _ = Hh.golden_masks(golden, 0)
q = [m_new]
student.mask_provider = lambda n: q.pop(0)
student.train()
blk = student.decoder[1].layers[4]
cap = {}
orig = blk.run
def run(x, task, train, drop=None):
    cap["x"] = x.detach().clone()
    return orig(x, task, train, drop)
blk.run = run
with torch.no_grad():
    student(torch.from_numpy(golden["it0_images"]).to(dev), 1)
x = cap["x"]                                   # NHWC on device
N, H, W, C = x.shape
w, b = blk.conv3x1_1.weight.detach(), blk.conv3x1_1.bias.detach()
G = ops.make_geom(N, H, W, H, W, ops._taps_3x1(1), C, H, W, C)
a1 = ops.tapconv(G, C, C, x, None, ops.pack_conv(w, "fwd"), torch.empty_like(x), bias=b, relu=True)
pre = ops.tapconv(G, C, C, x, None, ops.pack_conv(w, "fwd"), torch.empty_like(x), bias=b, relu=False)
xc = x.permute(0, 3, 1, 2).cpu()
pre_cpu = F.conv2d(xc, w.cpu(), b.cpu(), padding=(1, 0))
pre_cpu64 = F.conv2d(xc.double(), w.cpu().double(), b.cpu().double(), padding=(1, 0))
pre_h = pre.permute(0, 3, 1, 2).cpu()
print("x: zeros fraction", float((xc == 0).float().mean()), " x.max", float(xc.max()))
print("pre-activation: max|hip-cpu32|", float((pre_h - pre_cpu).abs().max()), " max|hip-cpu64|",
      float((pre_h.double() - pre_cpu64).abs().max()), " max|cpu32-cpu64|", float((pre_cpu.double() - pre_cpu64).abs().max()))
flip = (pre_h > 0) != (pre_cpu > 0)
print("sign mismatches hip vs cpu32:", int(flip.sum()), "of", flip.numel())
idx = torch.nonzero(flip)
for i in idx[:10]:
    t = tuple(i.tolist())
    print("   at", t, "hip", float(pre_h[t]), "cpu32", float(pre_cpu[t]), "cpu64", float(pre_cpu64[t]))
small = (pre_cpu64.abs() < 1e-5)
print("elements with |pre| < 1e-5:", int(small.sum()), "; exactly-zero in cpu32:", int((pre_cpu == 0).sum()), " exactly-zero hip:", int((pre_h == 0).sum()))
