#!/usr/bin/env python3
"""GPU diagnostic: relu-gate flips of the intermediate a1/a2 activations, HIP model vs CPU oracle."""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import rap_oracle as O
from tests import helpers as Hh
from tests.test_model_golden import _build
from mdil_ss_amd import ops

golden = np.load("tests/golden/step2_tiny.npz")
dev = torch.device("cuda:0")
rec_cpu = []
def fp(S, p, idx, x, d):
    pre = F.conv2d(x, S[f"{p}.conv3x1_{idx}.weight"], S[f"{p}.conv3x1_{idx}.bias"], padding=(d, 0), dilation=(d, 1))
    rec_cpu.append((f"{p}.a{idx}", pre.detach()))
    return F.conv2d(F.relu(pre), S[f"{p}.conv1x3_{idx}.weight"], S[f"{p}.conv1x3_{idx}.bias"], padding=(0, d), dilation=(1, d))
O._factor_pair = fp
t_sd, s_sd = Hh.golden_scenario(golden)
m_new, _ = Hh.golden_masks(golden, 0)
img = torch.from_numpy(golden["it0_images"])
with torch.no_grad():
    O.net_forward(s_sd, img, 1, True, m_new)

rec_hip = []
orig = ops.tapconv
def tc(g, cin, cout, in0, in1, wpk, out, bias=None, scale=None, shift=None, res=None, res_gate=None, gate=None, relu=False):
    r = orig(g, cin, cout, in0, in1, wpk, out, bias, scale, shift, res, res_gate, gate, relu)
    if relu and scale is None and g.ntaps == 3 and cin == cout:
        rec_hip.append(out.detach().permute(0, 3, 1, 2).cpu())
    return r
ops.tapconv = tc
student, teacher = _build(golden, dev)
q = [m_new]
student.mask_provider = lambda n: q.pop(0)
student.train()
with torch.no_grad():
    student(img.to(dev), 1)
print(len(rec_cpu), len(rec_hip))
tot = 0
for (n, pre), a in zip(rec_cpu, rec_hip):
    f = (a > 0) != (pre > 0)
    err = (a - F.relu(pre)).abs().max()
    k = int(f.sum()); tot += k
    near = int((pre.abs() < 2 * err).sum())
    print(f"{n:34s} flips {k:3d}  max err {float(err):.2e}  elems with |pre|<2*err: {near}")
    for i in torch.nonzero(f)[:2]:
        t = tuple(i.tolist()); print("       ", t, "hip", float(a[t]), "cpu pre", float(pre[t]))
print("total flips", tot)
