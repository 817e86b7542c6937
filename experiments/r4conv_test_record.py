"""GPU: the register-resident-weights form of the C = 128 3-tap convs (`r4conv_kernel`, csrc/w4conv.hip;
models/erfnet_RA_parallel.py:93-107: conv3x1_1 / conv3x1_2 and the dgrads of conv1x3_1 / conv1x3_2) against

  * `w4conv_kernel` (the same F(4,3) arithmetic with the weights in LDS; `MDIL_NO_R4CONV=1` in a child
    process, the switch is read once per process): BIT-IDENTICAL outputs -- same transform expressions, same
    accumulation order per accumulator -- for every epilogue the step launches and for ragged / small / odd
    shapes (tile queue shorter than the grid, last tile incomplete, non-power-of-two maps);
  * `F.conv2d` on the same inputs (fp32 on the device, the check of tests/test_hip_parity.py) at 1e-4 of the
    output's scale.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (N, H, W, axis, dilation, epilogue)
CASES = [
    (6, 64, 128, "w", 2, "bias_relu"),          # the bench's layer shape: conv3x1 / 1x3 forms of encoder.layers.7-14
    (6, 64, 128, "h", 2, "bias_relu"),
    (6, 64, 128, "h", 16, "bias_relu"),
    (6, 64, 128, "w", 16, "gate"),              # dgrad through conv1x3: gated by the ReLU input
    (6, 64, 128, "w", 8, "res_resgate"),        # dgrad + residual gradient gated by the block output
    (2, 32, 64, "h", 4, "folded_bn_relu"),      # eval mode: folded BatchNorm coefficients in the epilogue
    (2, 32, 64, "w", 1, "bias"),
    (2, 32, 64, "w", 4, "res_relu"),
    (1, 8, 16, "w", 2, "bias_relu"),            # 2 tiles: most work-groups of the grid have nothing to do
    (1, 8, 16, "h", 2, "gate"),
    (3, 24, 40, "w", 2, "bias_relu"),           # not powers of two: the integer-division pixel map; ragged last tile
    (3, 24, 40, "h", 2, "res_resgate"),
    (5, 16, 24, "w", 1, "gate"),
]


def _run_cases(path):
    """Every case through ops.tapconv (mdil_tapconv -> mdil_wconv -> w4conv / r4conv) -> npz of the outputs."""
    sys.path.insert(0, ROOT)
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import ops
    dev = torch.device("cuda:0")
    out = {}
    C = 128
    for i, (N, H, W, axis, d, epi) in enumerate(CASES):
        g = torch.Generator().manual_seed(1000 + i)
        x = torch.randn(N, H, W, C, generator=g).to(dev)
        w = (torch.randn(C, C, 3, 1, generator=g) * 0.05) if axis == "h" else (torch.randn(C, C, 1, 3, generator=g) * 0.05)
        w = w.to(dev)
        b = torch.randn(C, generator=g).to(dev)
        sc = (0.5 + torch.rand(C, generator=g)).to(dev)
        sh = torch.randn(C, generator=g).to(dev)
        r = torch.randn(N, H, W, C, generator=g).to(dev)
        r2 = torch.randn(N, H, W, C, generator=g).to(dev)
        taps = ops._taps_3x1(d) if axis == "h" else ops._taps_1x3(d)
        geom = ops.make_geom(N, H, W, H, W, taps, C, H, W, C)
        wp = ops.pack_conv(w, "fwd")
        y = torch.full((N, H, W, C), float("nan"), device=dev)
        kw = {"bias_relu": dict(bias=b, relu=True), "bias": dict(bias=b), "gate": dict(gate=r),
              "res_resgate": dict(res=r, res_gate=r2), "res_relu": dict(bias=b, res=r, relu=True),
              "folded_bn_relu": dict(bias=b, scale=sc, shift=sh, relu=True)}[epi]
        ops.tapconv(geom, C, C, x, None, wp, y, **kw)
        torch.cuda.synchronize()
        out[f"y{i}"] = y.cpu().numpy()
        # torch fp32 reference of the same op on the device
        xn = x.permute(0, 3, 1, 2)
        pad, dil = ((d, 0), (d, 1)) if axis == "h" else ((0, d), (1, d))
        z = torch.nn.functional.conv2d(xn, w, None, padding=pad, dilation=dil).permute(0, 2, 3, 1)
        if "bias" in kw:
            z = z + b
        if "scale" in kw:
            z = z * sc + sh
        if "res" in kw:
            z = z + (torch.where(r2 > 0, r, torch.zeros_like(r)) if "res_gate" in kw else r)
        if kw.get("relu"):
            z = z.relu()
        if "gate" in kw:
            z = torch.where(r > 0, z, torch.zeros_like(z))
        out[f"z{i}"] = z.cpu().numpy()
        ops.invalidate_packs()
    np.savez(path, **out)


def test_r4conv_is_bit_identical_to_w4conv(tmp_path):
    env = dict(os.environ, MDIL_NO_R4CONV="1", PYTHONPATH=ROOT)
    ref_path = str(tmp_path / "w4conv.npz")
    code = f"import sys; sys.path.insert(0, {ROOT!r}); from tests.test_r4conv_gpu import _run_cases; _run_cases({ref_path!r})"
    subprocess.run([sys.executable, "-c", code], check=True, env=env, timeout=600)
    assert os.environ.get("MDIL_NO_R4CONV") is None, "this process must run the shipped (register-resident) form"
    got_path = str(tmp_path / "r4conv.npz")
    _run_cases(got_path)
    ref, got = np.load(ref_path), np.load(got_path)
    for i, case in enumerate(CASES):
        y, y0, z = got[f"y{i}"], ref[f"y{i}"], got[f"z{i}"]
        assert np.isfinite(y).all(), case
        assert np.array_equal(y, y0), (case, "r4conv vs w4conv", float(np.abs(y - y0).max()))
        scale = float(np.abs(z).max())
        err = float(np.abs(y - z).max())
        assert err <= 1e-4 * scale, (case, "vs F.conv2d", err, scale)


if __name__ == "__main__":
    _run_cases(sys.argv[1])
