"""diagnostic: is the async_wgrad schedule deterministic, and does the fake 2-rank exchange match it?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.test_dp_gpu import _run
golden = np.load("tests/golden/step2_tiny.npz")
dev = torch.device("cuda:0")
_, a = _run(golden, dev, 1, True, async_wgrad=True)
_, b = _run(golden, dev, 1, True, async_wgrad=True)
print("1 rank vs 1 rank (async):", float((a - b).abs().max()))
_, c = _run(golden, dev, 2, True, async_wgrad=True)
print("1 rank vs fake 2 ranks (async, stages off):", float((a - c).abs().max()))
os.environ["MDIL_ASYNC_STAGES"] = "1"
_, d = _run(golden, dev, 2, True, async_wgrad=True)
print("1 rank vs fake 2 ranks (async, stages on):", float((a - d).abs().max()))
_, e = _run(golden, dev, 1, True, async_wgrad=False)
print("async vs sync 1 rank:", float((a - e).abs().max()))
