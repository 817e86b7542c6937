"""GPU parity tests: every HIP entry point / block Function against the CPU oracle on the same
seeded inputs (fp32; tolerances stated per test), plus the full model against the golden
fixtures generated from the reference.  Everything goes through libmdil_hip.so's C ABI."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import fixtures as fx
from oracle import rap_oracle as O
from tests import helpers as Hh

pytestmark = pytest.mark.gpu

RTOL, ATOL = 2e-4, 2e-5   # fp32 kernels vs fp32 CPU oracle (different summation order)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def close(got, want, rtol=RTOL, atol=ATOL, what="", flip_frac=0.0, flip_l2=5e-3):
    """Element-wise |got - want| <= atol * max|want| + rtol * |want|.
    ``flip_frac`` > 0 (backward quantities downstream of a ReLU only): a ReLU whose pre-activation
    is closer to zero than the fp32 forward error (~1e-6 relative) can take the other branch than
    the CPU oracle's; its gate is a step, so the handful of gradient elements behind it are off
    by O(1) whatever the kernel does.  Up to that fraction of elements may then exceed the bound,
    provided the relative L2 error of the whole tensor stays below ``flip_l2`` (a wrong tap, a
    missing term or a mis-scaled path is orders of magnitude above either limit)."""
    got = got.detach().cpu().double()
    want = want.detach().cpu().double()
    scale = max(1e-30, float(want.abs().max()))     # truly relative to the tensor's magnitude
    err = (got - want).abs()
    bound = atol * scale + rtol * want.abs()
    bad = err > bound
    if bad.any():
        frac = float(bad.sum()) / bad.numel()
        l2 = float(err.norm() / max(1e-30, float(want.norm())))
        if flip_frac > 0 and frac <= flip_frac and l2 <= flip_l2:
            return
        idx = torch.nonzero(bad)[0].tolist()
        raise AssertionError(f"{what}: {int(bad.sum())}/{bad.numel()} mismatches (rel L2 {l2:.2e}), max err "
                             f"{float(err.max()):.3e} (scale {scale:.3e}); first at {idx}: "
                             f"got {float(got[tuple(idx)]):.6e} want {float(want[tuple(idx)]):.6e}")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale)


# ------------------------------------------------------------------------------------------------
# raw tap convolution + wgrad against F.conv2d
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("C,H,W,d,kind", [
    (64, 12, 20, 1, "3x1"), (64, 12, 20, 1, "1x3"), (128, 9, 16, 2, "3x1"), (128, 9, 16, 4, "1x3"),
    (128, 20, 24, 16, "1x3"), (128, 20, 24, 16, "3x1"), (16, 10, 36, 1, "3x1"), (16, 10, 36, 1, "1x3"),
    # ragged last tiles / many tiles per wave of the streaming kernels
    (128, 97, 130, 16, "3x1"), (128, 97, 130, 2, "1x3"), (64, 130, 129, 1, "3x1"), (64, 130, 129, 1, "1x3"),
    # W % 16 == 0: the streaming weight-gradient kernel (row descriptors, 16-pixel quads)
    (64, 12, 32, 1, "3x1"), (64, 12, 32, 1, "1x3"), (128, 20, 48, 16, "3x1"), (128, 20, 48, 16, "1x3"),
    (128, 33, 64, 8, "1x3"), (128, 33, 64, 4, "3x1"), (64, 67, 96, 1, "1x3"), (128, 7, 16, 2, "3x1"),
    # C = 16: streaming weight gradient (wgrad16): pixel counts off the 4-pixel step, dilated taps
    (16, 9, 7, 2, "3x1"), (16, 9, 7, 2, "1x3"), (16, 33, 130, 1, "1x3"), (16, 33, 130, 1, "3x1"),
])
def test_tapconv_factorised(dev, C, H, W, d, kind):
    from mdil_ss_amd import ops
    N = 2
    x = rnd(N, C, H, W, seed=1)
    k = (3, 1) if kind == "3x1" else (1, 3)
    w = rnd(C, C, *k, seed=2, scale=0.1)
    b = rnd(C, seed=3, scale=0.1)
    pad = (d, 0) if kind == "3x1" else (0, d)
    dil = (d, 1) if kind == "3x1" else (1, d)
    want = F.relu(F.conv2d(x, w, b, padding=pad, dilation=dil))
    xd, wd, bd = nhwc(x).to(dev), w.to(dev), b.to(dev)
    taps = ops._taps_3x1(d) if kind == "3x1" else ops._taps_1x3(d)
    g = ops.make_geom(N, H, W, H, W, taps, C, H, W, C)
    out = ops.tapconv(g, C, C, xd, None, ops.pack_conv(wd, "fwd"), torch.empty_like(xd), bias=bd,
                      relu=True)
    close(nchw(out), want, what=f"tapconv {kind} C{C} d{d}")
    # dgrad: conv^T(g) -- compare with autograd
    xg = x.clone().requires_grad_(True)
    y = F.conv2d(xg, w, None, padding=pad, dilation=dil)
    go = rnd(*y.shape, seed=4)
    y.backward(go)
    tf = ops._taps_3x1(d, True) if kind == "3x1" else ops._taps_1x3(d, True)
    g2 = ops.make_geom(N, H, W, H, W, tf, C, H, W, C)
    gin = ops.tapconv(g2, C, C, nhwc(go).to(dev), None, ops.pack_conv(wd, "dgrad"),
                      torch.empty_like(xd))
    close(nchw(gin), xg.grad, what=f"dgrad {kind} C{C} d{d}")
    # wgrad
    wg = w.clone().requires_grad_(True)
    bg = b.clone().requires_grad_(True)
    F.conv2d(x, wg, bg, padding=pad, dilation=dil).backward(go)
    dw, db = ops.wgrad(g, C, C, xd, None, nhwc(go).to(dev), (0, 1, 2), C * 3, 3, wd, bd)
    close(dw, wg.grad, rtol=5e-4, atol=5e-5, what=f"wgrad {kind} C{C} d{d}")
    close(db, bg.grad, rtol=5e-4, atol=5e-5, what=f"bgrad {kind} C{C} d{d}")
    ops.invalidate_packs()


# ------------------------------------------------------------------------------------------------
# blocks (forward + backward) against the oracle's functional blocks
# ------------------------------------------------------------------------------------------------
def _bn_state(prefix, c, seed):
    g = torch.Generator().manual_seed(seed)
    return {prefix + ".weight": 1 + 0.1 * torch.randn(c, generator=g),
            prefix + ".bias": 0.1 * torch.randn(c, generator=g),
            prefix + ".running_mean": 0.1 * torch.randn(c, generator=g),
            prefix + ".running_var": 0.5 + torch.rand(c, generator=g),
            prefix + ".num_batches_tracked": torch.tensor(3)}


def _to_dev(S, dev):
    return {k: v.clone().to(dev) for k, v in S.items()}


def _grad_check(S_cpu, S_dev, names, what):
    for n in names:
        gc, gd = S_cpu[n].grad, S_dev[n].grad
        assert (gc is None) == (gd is None), f"{what}: grad presence of {n}"
        if gc is None:
            continue
        if Hh.zero_grad_bias(n):
            ref_mag = max(float(S_cpu[m].grad.abs().max()) for m in names if S_cpu[m].grad is not None)
            assert float(gd.abs().max()) < 1e-4 * ref_mag, n     # analytically zero
            continue
        close(gd, gc, rtol=1e-3, atol=1e-4, what=f"{what}: grad {n}")


@pytest.mark.parametrize("C,H,W,d,rap", [(64, 16, 24, 1, True), (128, 8, 24, 2, True),
                                         (128, 12, 20, 8, True), (128, 36, 40, 16, True),
                                         (64, 16, 24, 1, False), (16, 24, 40, 1, False),
                                         (128, 98, 132, 4, True), (64, 132, 130, 1, True),
                                         # W % 16 == 0: streaming weight gradients (adapter as 4th tap)
                                         (64, 24, 48, 1, True), (128, 20, 32, 16, True),
                                         (128, 40, 64, 4, True), (64, 24, 48, 1, False)])
@pytest.mark.parametrize("train", [True, False])
def test_nb_block(dev, C, H, W, d, rap, train):
    from mdil_ss_amd import ops
    N = 2
    p = "blk"
    S = {}
    for i, (kk, nm) in enumerate([((3, 1), "conv3x1_1"), ((1, 3), "conv1x3_1"), ((3, 1), "conv3x1_2"),
                                  ((1, 3), "conv1x3_2")]):
        S[f"{p}.{nm}.weight"] = rnd(C, C, *kk, seed=10 + i, scale=(1.0 / (3 * C)) ** 0.5)
        S[f"{p}.{nm}.bias"] = rnd(C, seed=20 + i, scale=0.1)
    if rap:
        for j in (1, 2):
            S[f"{p}.parallel_conv_{j}.0.weight"] = rnd(C, C, 1, 1, seed=30 + j, scale=(1.0 / C) ** 0.5)
            S[f"{p}.parallel_conv_{j}.0.bias"] = rnd(C, seed=40 + j, scale=0.1)
            S.update(_bn_state(f"{p}.bns_{j}.0", C, 50 + j))
    else:
        for j in (1, 2):
            S.update(_bn_state(f"{p}.bn{j}", C, 50 + j))
    x = F.relu(rnd(N, C, H, W, seed=5))
    mask = None
    if rap and train:
        mask = torch.empty(N, C, 1, 1).bernoulli_(0.7, generator=torch.Generator().manual_seed(9)).div_(0.7)
    names = [k for k in S if k.endswith((".weight", ".bias"))]
    Sd = _to_dev(S, dev)
    for n in names:
        S[n].requires_grad_(True)
        Sd[n].requires_grad_(True)
    xc = x.clone().requires_grad_(True)
    xd = nhwc(x).to(dev).requires_grad_(True)
    bn1, bn2 = (f"{p}.bns_1.0", f"{p}.bns_2.0") if rap else (f"{p}.bn1", f"{p}.bn2")
    bufs = tuple(Sd[f"{b}.{s}"] for b in (bn1, bn2) for s in ("running_mean", "running_var", "num_batches_tracked"))
    pw = (lambda j, s: Sd[f"{p}.parallel_conv_{j}.0.{s}"]) if rap else (lambda j, s: None)
    ops.GATE_LOG = [] if train else None
    got = ops.NbFn.apply(xd, Sd[f"{p}.conv3x1_1.weight"], Sd[f"{p}.conv3x1_1.bias"],
                         Sd[f"{p}.conv1x3_1.weight"], Sd[f"{p}.conv1x3_1.bias"], pw(1, "weight"),
                         pw(1, "bias"), Sd[bn1 + ".weight"], Sd[bn1 + ".bias"],
                         Sd[f"{p}.conv3x1_2.weight"], Sd[f"{p}.conv3x1_2.bias"],
                         Sd[f"{p}.conv1x3_2.weight"], Sd[f"{p}.conv1x3_2.bias"], pw(2, "weight"),
                         pw(2, "bias"), Sd[bn2 + ".weight"], Sd[bn2 + ".bias"], bufs,
                         None if mask is None else mask.reshape(N, C).to(dev), d, train, None, None)
    # train mode: the oracle replays the ReLU gates the HIP forward took (ops.GATE_LOG), so the
    # backward comparison below is element-wise tight for every tensor -- no allowance for
    # pre-activations that round to different sides of zero in the two implementations
    gates = None
    if train:
        gates = [g.cpu() for g in ops.GATE_LOG]
        assert len(gates) == 4
    ops.GATE_LOG = None
    want = O._rap(S, p, xc, 0, train, d, mask, gates) if rap else O._nb1d_d(S, p, xc, train, d, gates)
    assert not gates
    what = f"nb C{C} d{d} rap{rap} train{train}"
    close(nchw(got), want, what=what + " fwd")
    if train:
        for b in (bn1, bn2):
            for s_ in ("running_mean", "running_var", "num_batches_tracked"):
                close(Sd[f"{b}.{s_}"].float(), S[f"{b}.{s_}"].float(), what=f"{what} {b}.{s_}")
        go = rnd(*want.shape, seed=6)
        want.backward(go)
        got.backward(nhwc(go).to(dev))
        close(nchw(xd.grad), xc.grad, rtol=1e-3, atol=1e-4, what=what + " gx")
        _grad_check(S, Sd, names, what)
    ops.invalidate_packs()


@pytest.mark.parametrize("C,H,W,d,rap,frozen", [
    (64, 24, 48, 1, True, ()), (128, 20, 32, 4, True, ()), (16, 24, 40, 1, False, ()),
    (128, 12, 20, 8, True, ()),                                  # W % 16 != 0: LDS-tiled wgrad
    # step-2 freezing (train_new_task_step2.py:222-232): shared convs frozen, adapters + BN train
    (128, 20, 32, 2, True, ("conv3x1_1", "conv1x3_1", "conv3x1_2", "conv1x3_2")),
    # everything frozen: the backward only propagates the input gradient (old-domain KD graph)
    (64, 24, 48, 1, True, ("conv3x1_1", "conv1x3_1", "conv3x1_2", "conv1x3_2", "pc1", "pc2", "bn1", "bn2")),
])
def test_nb_block_abi_is_the_per_launch_path(dev, C, H, W, d, rap, frozen):
    """mdil_nb_block_forward / _backward (one foreign call per block) enqueue the same launches
    as the per-launch host orchestration: bit-identical outputs, gradients and running stats,
    in train and eval mode, with any subset of the parameters frozen."""
    from mdil_ss_amd import ops
    N = 3

    def run(block_abi, train):
        ops.invalidate_packs()
        old = ops.BLOCK_ABI
        ops.BLOCK_ABI = block_abi
        try:
            P = {}
            for i, (kk, nm) in enumerate([((3, 1), "conv3x1_1"), ((1, 3), "conv1x3_1"),
                                          ((3, 1), "conv3x1_2"), ((1, 3), "conv1x3_2")]):
                P[nm + ".w"] = rnd(C, C, *kk, seed=10 + i, scale=(1.0 / (3 * C)) ** 0.5)
                P[nm + ".b"] = rnd(C, seed=20 + i, scale=0.1)
            for j in (1, 2):
                P[f"pc{j}.w"] = rnd(C, C, 1, 1, seed=30 + j, scale=(1.0 / C) ** 0.5) if rap else None
                P[f"pc{j}.b"] = rnd(C, seed=40 + j, scale=0.1) if rap else None
                P[f"bn{j}.w"] = 1 + rnd(C, seed=50 + j, scale=0.1)
                P[f"bn{j}.b"] = rnd(C, seed=60 + j, scale=0.1)
            P = {k: (None if v is None else v.to(dev).requires_grad_(k.split(".")[0] not in frozen))
                 for k, v in P.items()}
            bufs = tuple(t for j in (1, 2) for t in (
                rnd(C, seed=70 + j, scale=0.1).to(dev), (1 + rnd(C, seed=80 + j, scale=0.1).abs()).to(dev),
                torch.zeros((), dtype=torch.int64, device=dev)))
            x = nhwc(F.relu(rnd(N, C, H, W, seed=5))).to(dev).requires_grad_(True)
            drop = None
            if rap and train:
                drop = torch.empty(N, C).bernoulli_(0.7, generator=torch.Generator().manual_seed(9)).div_(0.7).to(dev)
            out = ops.NbFn.apply(x, P["conv3x1_1.w"], P["conv3x1_1.b"], P["conv1x3_1.w"], P["conv1x3_1.b"],
                                 P["pc1.w"], P["pc1.b"], P["bn1.w"], P["bn1.b"], P["conv3x1_2.w"],
                                 P["conv3x1_2.b"], P["conv1x3_2.w"], P["conv1x3_2.b"], P["pc2.w"],
                                 P["pc2.b"], P["bn2.w"], P["bn2.b"], bufs, drop, d, train, None, None)
            res = {"out": out.detach().clone()}
            if train:
                out.backward(nhwc(rnd(N, C, H, W, seed=6)).to(dev))
                res["gx"] = x.grad.clone()
                for k, v in P.items():
                    if v is not None:
                        assert (v.grad is not None) == (k.split(".")[0] not in frozen), k
                        if v.grad is not None:
                            res["d" + k] = v.grad.clone()
                for i, t in enumerate(bufs):
                    res[f"buf{i}"] = t.clone()
            return res
        finally:
            ops.BLOCK_ABI = old

    for train in (True, False):
        a, b = run(True, train), run(False, train)
        assert a.keys() == b.keys()
        for k in a:
            assert torch.equal(a[k], b[k]), (k, train, (a[k].double() - b[k].double()).abs().max().item())
    ops.invalidate_packs()


@pytest.mark.parametrize("C,H,W,d,axis", [
    (64, 8, 64, 1, "w"), (128, 4, 64, 2, "w"), (128, 4, 96, 4, "w"), (128, 6, 64, 8, "w"), (128, 4, 128, 16, "w"),
    (64, 8, 32, 1, "h"), (128, 12, 32, 2, "h"), (128, 16, 48, 4, "h"), (128, 32, 16, 8, "h"), (128, 64, 16, 16, "h"),
    (128, 20, 32, 2, "h"),     # H % 4 == 0; (128, 20, 32, 16, "h") would not pair up -> direct kernel
    (128, 20, 32, 16, "h"),
    (64, 4, 12, 1, "w"), (128, 8, 12, 2, "h"),      # F(4,3) with a ragged last wave tile (24 / 48 quads)
])
def test_three_tap_conv_and_weight_gradient_winograd_shapes(dev, C, H, W, d, axis):
    """The 3-tap convs whose axis splits into complete quads / pairs take the Winograd F(4,3) / F(2,3)
    kernels (w4conv.hip / wconv.hip; weight gradients wgradw / wgradx in wgrad.hip): forward, weight and
    bias gradient against an fp64 reference, every dilation of the network on both axes, edges included
    (N = 2 so image boundaries are crossed); H % 4d != 0 falls to F(2,3), H % 2d != 0 to the direct form."""
    from mdil_ss_amd import ops
    N = 2
    kk = (1, 3) if axis == "w" else (3, 1)
    pad, dil = ((0, d), (1, d)) if axis == "w" else ((d, 0), (d, 1))
    x = rnd(N, C, H, W, seed=3)
    w = rnd(C, C, *kk, seed=4, scale=(1.0 / (3 * C)) ** 0.5)
    b = rnd(C, seed=5, scale=0.1)
    go = rnd(N, C, H, W, seed=6)
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    want = F.conv2d(xr, wr, br, padding=pad, dilation=dil)
    want.backward(go.double())
    taps = ops._taps_1x3(d) if axis == "w" else ops._taps_3x1(d)
    g = ops.make_geom(N, H, W, H, W, taps, C, H, W, C)
    xd, wd, bd, gd = nhwc(x).to(dev), w.to(dev), b.to(dev), nhwc(go).to(dev)
    out = ops.tapconv(g, C, C, xd, None, ops.pack_conv(wd, "fwd"), torch.empty_like(xd), bias=bd)
    close(nchw(out), want.float(), what=f"conv {axis} d{d} fwd")
    dw, db = ops.wgrad(g, C, C, xd, None, gd, (0, 1, 2), C * 3, 3, wd, bd)
    close(dw, wr.grad.float(), rtol=1e-4, atol=2e-5, what=f"conv {axis} d{d} dW")
    close(db, br.grad.float(), rtol=1e-4, atol=2e-5, what=f"conv {axis} d{d} db")
    # dgrad = the same kernel on mirrored taps
    gt = ops.make_geom(N, H, W, H, W, ops._taps_1x3(d, flip=True) if axis == "w" else ops._taps_3x1(d, flip=True),
                       C, H, W, C)
    gx = ops.tapconv(gt, C, C, gd, None, ops.pack_conv(wd, "dgrad"), torch.empty_like(xd))
    close(nchw(gx), xr.grad.float(), what=f"conv {axis} d{d} dgrad")
    ops.invalidate_packs()


@pytest.mark.parametrize("cin,cout,H,W", [(3, 16, 16, 24), (16, 64, 12, 40), (64, 128, 8, 12)])
@pytest.mark.parametrize("train", [True, False])
def test_down_block(dev, cin, cout, H, W, train):
    from mdil_ss_amd import ops
    N, p = 2, "d"
    S = {f"{p}.conv.weight": rnd(cout - cin, cin, 3, 3, seed=1, scale=(1.0 / (9 * cin)) ** 0.5),
         f"{p}.conv.bias": rnd(cout - cin, seed=2, scale=0.1)}
    S.update(_bn_state(f"{p}.bn_ini.0", cout, 3))
    x = rnd(N, cin, H, W, seed=4)
    if cin != 3:
        x = F.relu(x)       # post-ReLU inputs: exercises max-pool ties at 0
    names = [k for k in S if k.endswith((".weight", ".bias"))]
    Sd = _to_dev(S, dev)
    for n in names:
        S[n].requires_grad_(True)
        Sd[n].requires_grad_(True)
    xc = x.clone().requires_grad_(cin != 3)
    want = O._down(S, p, xc, 0, train)
    xd = nhwc(x).to(dev).requires_grad_(cin != 3)
    b = f"{p}.bn_ini.0"
    got = ops.DownFn.apply(xd, Sd[f"{p}.conv.weight"], Sd[f"{p}.conv.bias"], Sd[b + ".weight"],
                           Sd[b + ".bias"], Sd[b + ".running_mean"], Sd[b + ".running_var"],
                           Sd[b + ".num_batches_tracked"], train, None)
    what = f"down {cin}->{cout} train{train}"
    close(nchw(got), want, what=what + " fwd")
    if train:
        close(Sd[b + ".running_var"], S[b + ".running_var"], what=what + " running_var")
        go = rnd(*want.shape, seed=6)
        want.backward(go)
        got.backward(nhwc(go).to(dev))
        if cin != 3:
            close(nchw(xd.grad), xc.grad, rtol=1e-3, atol=1e-4, what=what + " gx")
        _grad_check(S, Sd, names, what)
    ops.invalidate_packs()


@pytest.mark.parametrize("cin,cout,H,W", [(128, 64, 6, 10), (64, 16, 10, 12)])
@pytest.mark.parametrize("train", [True, False])
def test_up_block(dev, cin, cout, H, W, train):
    from mdil_ss_amd import ops
    N, p = 2, "u"
    S = {f"{p}.conv.weight": rnd(cin, cout, 3, 3, seed=1, scale=(1.0 / (9 * cin)) ** 0.5),
         f"{p}.conv.bias": rnd(cout, seed=2, scale=0.1)}
    S.update(_bn_state(f"{p}.bn", cout, 3))
    x = F.relu(rnd(N, cin, H, W, seed=4))
    names = [k for k in S if k.endswith((".weight", ".bias"))]
    Sd = _to_dev(S, dev)
    for n in names:
        S[n].requires_grad_(True)
        Sd[n].requires_grad_(True)
    xc = x.clone().requires_grad_(True)
    want = O._up(S, p, xc, train)
    xd = nhwc(x).to(dev).requires_grad_(True)
    b = f"{p}.bn"
    got = ops.UpFn.apply(xd, Sd[f"{p}.conv.weight"], Sd[f"{p}.conv.bias"], Sd[b + ".weight"],
                         Sd[b + ".bias"], Sd[b + ".running_mean"], Sd[b + ".running_var"],
                         Sd[b + ".num_batches_tracked"], train, None)
    what = f"up {cin}->{cout} train{train}"
    close(nchw(got), want, what=what + " fwd")
    if train:
        go = rnd(*want.shape, seed=6)
        want.backward(go)
        got.backward(nhwc(go).to(dev))
        close(nchw(xd.grad), xc.grad, rtol=1e-3, atol=1e-4, what=what + " gx")
        _grad_check(S, Sd, names, what)
    ops.invalidate_packs()


@pytest.mark.parametrize("nc", [20, 27])
def test_output_conv(dev, nc):
    """nc = 27 (IDD head): logits rows are padded to 28 floats inside the HIP path."""
    from mdil_ss_amd import ops
    N, H, W = 2, 10, 12
    w = rnd(16, nc, 2, 2, seed=1, scale=0.2).requires_grad_(True)
    b = rnd(nc, seed=2, scale=0.1).requires_grad_(True)
    x = F.relu(rnd(N, 16, H, W, seed=3)).requires_grad_(True)
    want = F.conv_transpose2d(x, w, b, stride=2)
    wd, bd = w.detach().to(dev).requires_grad_(True), b.detach().to(dev).requires_grad_(True)
    xd = nhwc(x.detach()).to(dev).requires_grad_(True)
    got = ops.OutFn.apply(xd, wd, bd)
    assert tuple(got.shape) == (N, 2 * H, 2 * W, nc)
    close(nchw(got), want, what="output_conv fwd")
    go = rnd(*want.shape, seed=4)
    want.backward(go)
    got.backward(nhwc(go).to(dev))
    close(nchw(xd.grad), x.grad, rtol=1e-3, atol=1e-4, what="output_conv gx")
    close(wd.grad, w.grad, rtol=1e-3, atol=1e-4, what="output_conv dw")
    close(bd.grad, b.grad, rtol=1e-3, atol=1e-4, what="output_conv db")
    ops.invalidate_packs()


# ------------------------------------------------------------------------------------------------
# losses / metric / optimizer
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nc", [20, 27])
def test_losses(dev, nc):
    from mdil_ss_amd import ops
    N, H, W = 2, 24, 40
    s = rnd(N, nc, H, W, seed=1, scale=2.0).requires_grad_(True)
    t = rnd(N, nc, H, W, seed=2, scale=2.0)
    _, lab = fx.make_batch(N, H, W, nc, seed=3)
    weight = torch.tensor(fx.WEIGHT_BDD) if nc == 20 else torch.cat(
        [1.0 + 9.0 * torch.rand(nc - 1, generator=torch.Generator().manual_seed(4)), torch.zeros(1)])
    ce = O.ce2d(s, lab[:, 0], weight)
    kld = O.kld_prob(s, t)
    (ce * 1.0 + 0.1 * kld).backward()
    sd = s.detach().to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    td = t.to(dev).contiguous(memory_format=torch.channels_last)
    ce_d = ops.cross_entropy2d(sd, lab[:, 0].to(dev), weight.to(dev))
    kld_d = ops.kld_prob(sd, td)
    assert float(ce_d) == pytest.approx(float(ce), rel=1e-5)
    assert float(kld_d) == pytest.approx(float(kld), rel=1e-4, abs=1e-7)
    (ce_d + 0.1 * kld_d).backward()
    close(sd.grad, s.grad, rtol=1e-4, atol=1e-5, what="d(ce+0.1*kld)/dlogits")


@pytest.mark.parametrize("nc", [20, 27])
@pytest.mark.parametrize("N,H,W", [(2, 12, 20), (1, 9, 7), (3, 16, 48)])
def test_fused_head_and_loss(dev, nc, N, H, W):
    """ops.head_ce / ops.head_kld (csrc/head.hip): Decoder.output_conv fused with the loss that
    consumes its logits (models/erfnet_RA_parallel.py:179-180,188 + train_new_task_step2.py:84-92,
    241,293-297), logits never materialised, backward recomputes them -- against stock torch:
    conv_transpose2d + the oracle's losses + autograd.  Pixel counts that are not multiples of the
    64-pixel wave batch; ignore labels; an upstream gradient scale; the optional logits output."""
    from mdil_ss_amd import ops
    g = torch.Generator().manual_seed(10 * nc + H)
    w = (torch.randn(16, nc, 2, 2, generator=g) * 0.3).requires_grad_(True)
    b = (torch.randn(nc, generator=g) * 0.2).requires_grad_(True)
    wt = torch.randn(16, nc, 2, 2, generator=g) * 0.3
    bt = torch.randn(nc, generator=g) * 0.2
    x = F.relu(torch.randn(N, 16, H, W, generator=g)).requires_grad_(True)
    xt = F.relu(torch.randn(N, 16, H, W, generator=g))
    _, lab = fx.make_batch(N, 2 * H, 2 * W, nc, seed=3)
    weight = torch.tensor(fx.WEIGHT_BDD) if nc == 20 else torch.cat(
        [1.0 + 9.0 * torch.rand(nc - 1, generator=torch.Generator().manual_seed(4)), torch.zeros(1)])
    # ---- oracle
    logits = F.conv_transpose2d(x, w, b, stride=2)
    ce = O.ce2d(logits, lab[:, 0], weight)
    kld = O.kld_prob(logits, F.conv_transpose2d(xt, wt, bt, stride=2))
    (0.7 * ce + 0.1 * kld).backward()
    # ---- HIP, CE and KLD separately (each with its own leaf tensors), then summed like the oracle
    def leaf(t, perm=False):
        t = t.detach()
        return (nhwc(t) if perm else t.clone()).to(dev).requires_grad_(True)
    xd1, wd1, bd1 = leaf(x, True), leaf(w), leaf(b)
    ce_d, lg = ops.head_ce(xd1, wd1, bd1, lab[:, 0].to(dev), weight.to(dev), True)
    assert tuple(lg.shape) == (N, nc, 2 * H, 2 * W) and not lg.requires_grad
    close(lg, logits, what="logits output of the fused head")
    ce_only = ops.head_ce(xd1.detach(), wd1.detach(), bd1.detach(), lab[:, 0].to(dev), weight.to(dev))
    assert float(ce_only) == float(ce_d)                     # same kernel with / without the logits store
    assert float(ce_d) == pytest.approx(float(ce), rel=2e-5)
    (0.7 * ce_d).backward()
    xd2, wd2, bd2 = leaf(x, True), leaf(w), leaf(b)
    kld_d = ops.head_kld(xd2, wd2, bd2, nhwc(xt).to(dev), wt.to(dev), bt.to(dev))
    assert float(kld_d) == pytest.approx(float(kld), rel=1e-4, abs=1e-7)
    (0.1 * kld_d).backward()
    close(nchw(xd1.grad + xd2.grad), x.grad, rtol=1e-3, atol=1e-4, what="fused head gx")
    close(wd1.grad + wd2.grad, w.grad, rtol=1e-3, atol=1e-4, what="fused head dw")
    close(bd1.grad + bd2.grad, b.grad, rtol=1e-3, atol=1e-4, what="fused head db")
    # the KLD term alone against its own oracle gradient (it is 1e-2 of the CE term above)
    x2 = x.detach().clone().requires_grad_(True)
    w2, b2 = w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    O.kld_prob(F.conv_transpose2d(x2, w2, b2, stride=2), F.conv_transpose2d(xt, wt, bt, stride=2)).backward()
    close(nchw(xd2.grad), 0.1 * x2.grad, rtol=1e-3, atol=1e-4, what="fused KLD gx")
    close(wd2.grad, 0.1 * w2.grad, rtol=1e-3, atol=1e-4, what="fused KLD dw")
    close(bd2.grad, 0.1 * b2.grad, rtol=1e-3, atol=2e-4, what="fused KLD db")
    # frozen head (the old-domain decoder of step 2): only the feature gradient
    xd3 = leaf(x, True)
    ops.head_kld(xd3, w.detach().to(dev), b.detach().to(dev), nhwc(xt).to(dev), wt.to(dev), bt.to(dev)).backward()
    close(nchw(xd3.grad), x2.grad, rtol=1e-3, atol=1e-4, what="fused KLD gx, frozen head")


def test_argmax_confusion(dev, golden_iou):
    from mdil_ss_amd import ops
    I = golden_iou
    pred, targ = torch.from_numpy(I["pred"]), torch.from_numpy(I["targ"])
    counts = torch.zeros(3, 20, dtype=torch.int64, device=dev)
    for p_ in (pred, targ):
        logits = rnd(3, 20, 16, 24, seed=5)
        logits.scatter_(1, p_, 10.0)          # argmax == pred
        ops.argmax_confusion(logits.to(dev).contiguous(memory_format=torch.channels_last),
                             targ[:, 0].to(dev), 19, counts)
    c = counts.cpu().numpy()
    np.testing.assert_array_equal(c[0, :19], I["tp"])
    np.testing.assert_array_equal(c[1, :19], I["fp"])
    np.testing.assert_array_equal(c[2, :19], I["fn"])
    # 27 classes (ignore = 26): logits arrive as an arbitrary [N,27,H,W] tensor and are re-laid
    # out into 28-float rows by the host side
    pred, targ = torch.from_numpy(I["pred27"]), torch.from_numpy(I["targ27"])
    logits = rnd(*pred.shape[:1], 27, *pred.shape[2:], seed=6)
    logits.scatter_(1, pred, 10.0)
    counts = torch.zeros(3, 27, dtype=torch.int64, device=dev)
    ops.argmax_confusion(logits.to(dev), targ[:, 0].to(dev), 26, counts)
    c = counts.cpu().numpy()
    np.testing.assert_array_equal(c[0, :26], I["tp27"])
    np.testing.assert_array_equal(c[1, :26], I["fp27"])
    np.testing.assert_array_equal(c[2, :26], I["fn27"])


def test_adam(dev):
    from mdil_ss_amd import ops
    n = 100003
    p, g = rnd(n, seed=1), rnd(n, seed=2, scale=1e-3)
    m, v = torch.zeros(n), torch.zeros(n)
    pd, md, vd = p.clone().to(dev), m.clone().to(dev), v.clone().to(dev)
    for step in (1, 2, 3):
        gi = g * step
        O.adam_l2_step(p, gi, m, v, step, 5e-4)
        ops.adam_step(pd, gi.to(dev), md, vd, step, 5e-4, weight_decay=1e-4)
    close(pd, p, rtol=1e-6, atol=1e-7, what="adam param")
    close(vd, v, rtol=1e-5, atol=1e-12, what="adam v")


def test_out_of_range_labels_raise(dev):
    """A label outside [0, C) (an un-relabelled 255) must not index past the class-weight table or
    the confusion histogram: the pixel is dropped and the next ``ops.check_labels()`` raises, where
    the reference's nll_loss / scatter_ raise a device assert (train_new_task_step2.py:92,
    iouEval.py:33)."""
    from mdil_ss_amd import ops
    N, H, W, nc = 2, 8, 16, 20
    s = rnd(N, nc, H, W, seed=1, scale=2.0)
    _, lab = fx.make_batch(N, H, W, nc, seed=3)
    w = torch.tensor(fx.WEIGHT_BDD)
    good = ops.cross_entropy2d(s.to(dev), lab[:, 0].to(dev), w.to(dev))
    ops.check_labels()                                        # clean so far
    bad_lab = lab.clone()
    bad_lab[0, 0, 0, :5] = 255
    bad_lab[1, 0, 3, 2] = -1
    loss = ops.cross_entropy2d(s.to(dev), bad_lab[:, 0].to(dev), w.to(dev))
    assert bool(torch.isfinite(loss))
    with pytest.raises(RuntimeError, match="outside"):
        ops.check_labels()
    ops.check_labels()                                        # the counter was reset
    # the dropped pixels behave like weight-0 pixels: same value as the oracle on the valid ones
    keep = (bad_lab[:, 0] >= 0) & (bad_lab[:, 0] < nc)
    ref_lab = torch.where(keep, bad_lab[:, 0], torch.full_like(bad_lab[:, 0], nc - 1))   # w[19] = 0
    assert float(loss) == pytest.approx(float(O.ce2d(s, ref_lab, w)), rel=1e-5)
    counts = torch.zeros(3, nc, dtype=torch.int64, device=dev)
    ops.argmax_confusion(s.to(dev), bad_lab[:, 0].to(dev), 19, counts)
    with pytest.raises(RuntimeError, match="outside"):
        ops.check_labels()
    with pytest.raises(RuntimeError, match="int64"):
        ops.cross_entropy2d(s.to(dev), lab[:, 0].to(dev).int(), w.to(dev))
    assert bool(torch.isfinite(good))


def test_packed_weights_follow_inplace_changes(dev):
    """Packed weight images are caches keyed by the parameter's storage.  A weight torch modifies
    in place (p.data.copy_/mul_, a foreign optimizer, EMA) bumps its version; the next use must see
    the new values (the stale image used to be served silently)."""
    from mdil_ss_amd import ops
    ops.invalidate_packs()
    N, H, W, C = 2, 8, 16, 64
    x = nhwc(rnd(N, C, H, W, seed=1)).to(dev)
    w = rnd(C, C, 3, 1, seed=2, scale=0.1).to(dev)
    g = ops.make_geom(N, H, W, H, W, ops._taps_3x1(1), C, H, W, C)
    y1 = ops.tapconv(g, C, C, x, None, ops.pack_conv(w, "fwd"), torch.empty_like(x)).clone()
    w.mul_(2.0)                                              # in place: same storage, new version
    y2 = ops.tapconv(g, C, C, x, None, ops.pack_conv(w, "fwd"), torch.empty_like(x))
    close(y2, 2.0 * y1, rtol=1e-6, atol=1e-7, what="conv after an in-place weight change")
    ops.invalidate_packs()


def test_nchw_to_nhwc_and_dropout_factors(dev):
    """The two small device kernels in front of the stem: layout conversion of the reference's NCHW
    float images and the Dropout2d factors from one uniform draw (bit-exact against torch)."""
    from mdil_ss_amd import ops
    x = rnd(3, 3, 20, 36, seed=1).to(dev)
    y = ops.to_nhwc(x)
    assert y.shape == (3, 20, 36, 3) and y.is_contiguous()
    assert torch.equal(y, x.permute(0, 2, 3, 1).contiguous())
    assert ops.to_nhwc(y.permute(0, 3, 1, 2)).data_ptr() == y.data_ptr()      # channels-last storage: no copy
    x5 = rnd(2, 5, 7, 9, seed=2).to(dev)
    assert torch.equal(ops.to_nhwc(x5), x5.permute(0, 2, 3, 1).contiguous())
    g = torch.Generator(device=dev).manual_seed(7)
    u = torch.rand(4096, device=dev, generator=g)
    keep = torch.cat([torch.full((1024,), 0.97), torch.full((3072,), 0.7)]).to(dev)
    inv = 1.0 / keep
    got = ops.dropout_factors(u, keep, inv)
    assert torch.equal(got, (u < keep).to(torch.float32) * inv)
    assert 0.2 < float((got[1024:] == 0).float().mean()) < 0.4


def test_optimizer_repack_follows_its_flat_buffer_not_requires_grad(dev):
    """the fused optimizer re-packs the images of what it just rewrote.  A parameter frozen AFTER the
    optimizer was built still moves in the flat buffer (weight decay) and the C-ABI update does not bump
    ``._version``: its packed image must follow all the same (selection by storage, not by ``requires_grad``),
    while the image of a tensor outside the optimizer's buffer is left alone."""
    from mdil_ss_amd import ops
    from mdil_ss_amd.engine import FlatAdam
    ops.invalidate_packs()
    N, H, W, C = 2, 8, 16, 64
    x = nhwc(rnd(N, C, H, W, seed=1)).to(dev)
    w = torch.nn.Parameter(rnd(C, C, 3, 1, seed=2, scale=0.1).to(dev))
    frozen = rnd(C, C, 3, 1, seed=3, scale=0.1).to(dev)                      # a teacher's weight: never in the optimizer
    opt = FlatAdam([{"params": [w]}], lr=1e-2, weight_decay=0.5)
    g = ops.make_geom(N, H, W, H, W, ops._taps_3x1(1), C, H, W, C)
    conv = lambda t: ops.tapconv(g, C, C, x, None, ops.pack_conv(t, "fwd"), torch.empty_like(x)).clone()
    y_w, y_f = conv(w), conv(frozen)          # (the Parameter object itself is the image's source, as in the models)
    w.requires_grad_(False)                                                     # frozen after construction
    w0 = w.data.clone()
    opt.flat_grad.fill_(0.25)
    opt.step()                                                                  # C-ABI Adam: w moves, no version bump
    assert not torch.equal(w.data, w0)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w.data, None, padding=(1, 0)).permute(0, 2, 3, 1)
    y_w2, y_f2 = conv(w), conv(frozen)
    close(y_w2, ref, rtol=1e-4, atol=1e-5, what="conv on the re-packed image of a parameter frozen after the optimizer was built")
    assert not torch.equal(y_w2, y_w) and torch.equal(y_f2, y_f)
    ops.invalidate_packs()


def test_input_gradient_is_not_cut_silently(dev):
    """``to_nhwc`` is a raw kernel into a fresh buffer -- for an input image that requires a gradient it
    would cut the graph silently (``img.grad`` stays None, nobody notices).  It now takes the differentiable ATen path
    for such an input, so the request reaches the stem's backward -- which has no input-gradient kernel (3-channel
    dgrad of the stride-2 stem conv: nothing in the reference's training asks for d loss / d image) and says so
    LOUDLY instead of returning nothing."""
    from mdil_ss_amd import ops
    from mdil_ss_amd.models.erfnet_RA_parallel import Net
    ops.invalidate_packs()
    torch.manual_seed(0)
    net = Net([20], 1, 0).to(dev).train()         # (the block operators keep what a backward needs in train mode only)
    img = rnd(2, 3, 32, 64, seed=5).to(dev).requires_grad_(True)
    y = net(img, 0)
    with pytest.raises(RuntimeError, match="no tile configuration"):
        y.float().square().mean().backward()
    torch.cuda.synchronize()
    assert ops.to_nhwc(img.detach()).requires_grad is False
    x = ops.to_nhwc(img)
    assert x.requires_grad and x.grad_fn is not None          # the ATen path: still attached to the image
    ops.invalidate_packs()
