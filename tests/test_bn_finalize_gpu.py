"""GPU: the BatchNorm finalize steps inside their producing launches (TICKETS, include/mdil_hip.h;
csrc/bnfin.h) against the stand-alone finalize launch: same device code, same fixed merge order ->
BIT-IDENTICAL coefficients, running statistics and affine gradients, whichever work-group arrives
last.  The hand-off crosses work-groups on different XCDs (non-coherent L2s): it is exercised here
many times in a row, back to back with other launches (uneven load), at the layer shapes of the
network and at ragged ones, checking every word."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup():
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import _lib, ops
    return _lib, ops


@pytest.mark.parametrize("Cc,H,W,d", [(64, 32, 64, 1), (128, 16, 32, 2), (128, 32, 64, 16), (64, 12, 20, 1),
                                      (16, 24, 40, 1), (128, 64, 128, 4)])
def test_train_statistics_finalized_in_the_producer_are_bit_identical(Cc, H, W, d):
    _lib, ops = _setup()
    dev = torch.device("cuda:0")
    torch.manual_seed(Cc + H)
    N = 3
    x = torch.randn(N, H, W, Cc, device=dev).relu_()
    x2 = torch.randn(N, H, W, Cc, device=dev)
    w13 = torch.randn(Cc, Cc, 1, 3, device=dev) * 0.05
    pw = torch.randn(Cc, Cc, 1, 1, device=dev) * 0.05
    b1, b2 = torch.randn(Cc, device=dev), torch.randn(Cc, device=dev)
    gamma, beta = torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev)
    rap = Cc != 16
    g = ops.make_geom(N, H, W, H, W, ops._taps_1x3(d) + ([(0, 0, 1)] if rap else []), Cc, H, W, Cc)
    wp = ops.pack_pair(w13, pw if rap else None, "fwd")
    noise = torch.randn(1 << 22, device=dev)

    def run(fused, rounds):
        ops.BN_FIN = fused
        outs = []
        rm, rv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
        nbt = torch.zeros((), dtype=torch.int64, device=dev)
        for r in range(rounds):
            out = torch.empty_like(x)
            if r % 3 == 1:
                noise.mul_(1.0001)              # another kernel in front: uneven arrival
            coef = ops.tapconv_bn(g, Cc, Cc, x, x2 if rap else None, wp, out, gamma, beta, rm, rv, nbt,
                                  bias=b1, bias2=b2 if rap else None)
            outs.append((coef.clone(), out))
        torch.cuda.synchronize()
        return outs, rm.clone(), rv.clone(), int(nbt)

    try:
        ref, rm0, rv0, n0 = run(False, 3)
        got, rm1, rv1, n1 = run(True, 3)
        for (c0, o0), (c1, o1) in zip(ref, got):
            assert torch.equal(c0, c1), float((c0 - c1).abs().max())
            assert torch.equal(o0, o1)
        assert torch.equal(rm0, rm1) and torch.equal(rv0, rv1) and n0 == n1 == 3
        # many launches in a row: the ticket must come back to zero every time and no launch may
        # read another launch's rows
        many, *_ = run(True, 40)
        for c, _ in many[1:]:
            assert torch.equal(c[:2], many[0][0][:2])       # same input -> same batch statistics
        for t in ops._tickets.values():
            assert int(t[0]) == 0
        # against torch: batch mean / biased variance of the conv output
        zo = many[0][1]
        mean = zo.double().mean(dim=(0, 1, 2))
        var = zo.double().var(dim=(0, 1, 2), unbiased=False)
        torch.testing.assert_close(many[0][0][0].double(), mean, rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(many[0][0][1].double(), 1.0 / torch.sqrt(var + 1e-3), rtol=1e-5, atol=1e-6)
    finally:
        ops.BN_FIN = True
        ops.invalidate_packs()


@pytest.mark.parametrize("Cc,H,W", [(64, 32, 64), (128, 16, 32), (16, 24, 40), (64, 128, 256)])
def test_bn_backward_reductions_finalized_in_the_producer_are_bit_identical(Cc, H, W):
    _lib, ops = _setup()
    dev = torch.device("cuda:0")
    torch.manual_seed(7 * Cc + W)
    N = 3
    gy = torch.randn(N, H, W, Cc, device=dev)
    y = torch.randn(N, H, W, Cc, device=dev)
    z = torch.randn(N, H, W, Cc, device=dev) * 2 + 0.3
    gamma, beta = torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev)
    rm, rv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    res = {}
    try:
        for fused in (False, True, True, True):
            ops.BN_FIN = fused
            coef = ops.bn_train_stats(z, gamma, beta, rm.clone(), rv.clone(), nbt)
            gz, dg, db = ops.bn_backward(gy, y, None, z, gamma, beta, coef, True)
            torch.cuda.synchronize()
            if fused in res:
                for a, b in zip(res[fused], (coef, gz, dg, db)):
                    assert torch.equal(a, b)
            res[fused] = (coef.clone(), gz.clone(), dg.clone(), db.clone())
        for a, b in zip(res[False], res[True]):
            assert torch.equal(a, b), float((a - b).abs().max())
    finally:
        ops.BN_FIN = True
