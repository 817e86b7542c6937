"""diagnostic: the async_wgrad fake-2-rank comparison AFTER tests/test_model_golden.py ran in this process"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, pytest
which = sys.argv[1] if len(sys.argv) > 1 else "tests/test_model_golden.py"
pytest.main(["-q", "-m", "gpu", "-x", which, "-p", "no:cacheprovider"])
from tests.test_dp_gpu import _run
from mdil_ss_amd import ops
golden = np.load("tests/golden/step2_tiny.npz")
dev = torch.device("cuda:0")
print("side streams:", {k: v.cuda_stream for k, v in ops._side_streams.items()}, "BN_TAIL", ops.BN_TAIL, "ASYNC", ops.ASYNC_WGRAD)


def where(eng, a, b):
    opt = eng.optimizer
    d = (a - b).abs()
    out = []
    off = 0
    names = [n for n, _ in eng.student.named_parameters()]
    worst = []
    for gi, g in enumerate(opt.param_groups):
        o = g["offset"]
        for p in g["params"]:
            n = p.numel()
            m = float(d[o:o + n].max())
            if m > 0:
                nm = [nn for nn, pp in eng.student.named_parameters() if pp is p]
                worst.append((m, gi, nm[0] if nm else "?"))
            o += n
    worst.sort(reverse=True)
    return f"{len(worst)} tensors differ; worst: {worst[:6]}"


for asyncw in (True, False):
    e1, a = _run(golden, dev, 1, True, async_wgrad=asyncw)
    _, b = _run(golden, dev, 1, True, async_wgrad=asyncw)
    e2, c = _run(golden, dev, 2, True, async_wgrad=asyncw)
    _, c2 = _run(golden, dev, 2, True, async_wgrad=asyncw)
    print(f"async {asyncw}: 1 vs 1 rank {float((a - b).abs().max()):.2e} | 2 vs 2 {float((c - c2).abs().max()):.2e} | 1 vs fake 2: {where(e2, a, c)}", flush=True)
ops.ASYNC_WGRAD = False
