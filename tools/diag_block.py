#!/usr/bin/env python3
"""GPU diagnostic: does decoder.1.layers.4's backward give the same parameter gradients in the
full model and in isolation (same x, same gy)?  And is the full-model backward deterministic?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fixtures as fx
from tests import helpers as Hh
from tests.test_model_golden import _build
from mdil_ss_amd import ops

golden = np.load("tests/golden/step2_tiny.npz")
dev = torch.device("cuda:0")
weight = torch.tensor(fx.WEIGHT_BDD, device=dev)
images = torch.from_numpy(golden["it0_images"]).to(dev)
labels = torch.from_numpy(golden["it0_labels"]).to(dev)

def full_run(capture=None):
    student, teacher = _build(golden, dev)
    m_new, m_old = Hh.golden_masks(golden, 0)
    q = [m_new, m_old]
    student.mask_provider = lambda n: q.pop(0)
    student.train(); teacher.eval()
    blk = student.decoder[1].layers[4]
    cap = {}
    if capture is not None:
        orig = blk.run
        def run(x, task, train, drop=None):
            x.retain_grad() if x.requires_grad else None
            cap["x"] = x
            y = orig(x, task, train, drop)
            y.retain_grad()
            cap["y"] = y
            return y
        blk.run = run
    out_new = student(images, 1)
    out_prev = student(images, 0)
    with torch.no_grad():
        out_t = teacher(images, 0)
    ce = ops.cross_entropy2d(out_new, labels[:, 0], weight)
    kld = ops.kld_prob(out_prev, out_t)
    (ce + 0.1 * kld).backward()
    torch.cuda.synchronize()
    grads = {n: p.grad.clone() for n, p in student.named_parameters() if p.grad is not None}
    return student, grads, cap

s1, g1, cap = full_run(capture=True)
s2, g2, _ = full_run()
nd = sum(int(not torch.equal(g1[n], g2[n])) for n in g1)
print("non-deterministic tensors between two identical full runs:", nd, "of", len(g1))
for n in g1:
    if not torch.equal(g1[n], g2[n]):
        print("   ", n, float((g1[n] - g2[n]).abs().max()))
        break
# isolated rerun of the block
blk = s1.decoder[1].layers[4]
x = cap["x"].detach().clone().requires_grad_(True)
gy = cap["y"].grad.detach().clone()
for p in blk.parameters():
    p.grad = None
y = blk.run(x, 0, True)
y.backward(gy)
torch.cuda.synchronize()
pref = "decoder.1.layers.4."
for n, p in blk.named_parameters():
    a, b = p.grad, g1[pref + n]
    print(f"{n:22s} isolated-vs-inmodel max|diff| {float((a - b).abs().max()):.3e}  (|g| max {float(b.abs().max()):.3e})")
print("gx isolated vs in-model:", float((x.grad - cap['x'].grad).abs().max()))
