"""diagnostic: is the async_wgrad schedule deterministic, and does the fake 2-rank exchange match it?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.test_dp_gpu import _run
golden = np.load("tests/golden/step2_tiny.npz")
dev = torch.device("cuda:0")


def where(eng, a, b):
    opt = eng.optimizer
    g0, g1 = opt.param_groups[0], opt.param_groups[1]
    d = (a - b).abs()
    out = []
    for name, (o, n) in (("shared", (g0["offset"], g0["numel"])), ("ds", (g1["offset"], g1["numel"]))):
        out.append(f"{name} {float(d[o:o + n].max()):.2e}")
    nd = eng.bucket_dec.numel()
    end = g1["offset"] + g1["numel"]
    out.append(f"dec {float(d[end - nd:end].max()):.2e}")
    return " ".join(out)


for sg in ("off", "8"):
    os.environ["MDIL_STAGGER"] = sg
    for asyncw in (False, True):
        e1, a = _run(golden, dev, 1, True, async_wgrad=asyncw)
        _, b = _run(golden, dev, 1, True, async_wgrad=asyncw)
        e2, c = _run(golden, dev, 2, True, async_wgrad=asyncw)
        print(f"stagger {sg} async {asyncw}: 1 vs 1 rank {float((a - b).abs().max()):.2e} | 1 vs fake 2: {where(e2, a, c)}", flush=True)
import mdil_ss_amd.ops as ops
ops.ASYNC_WGRAD = False
