"""GPU, BASELINE.json's full size (N=6, 512x1024): size-independent properties of the HIP path,
plus one full-resolution single-image step against the CPU oracle.

  * linearity / additivity of the convolution, its data gradient and its weight gradient at the
    real layer shapes (C=64 at 128x256, C=128 at 64x128, N=6);
  * sampled-pixel check of the convolution against a direct fp64 evaluation;
  * eval-mode forward: a batch of 6 equals six single-image forwards (no cross-image coupling);
  * losses / confusion counts: the batch value recombines from its halves;
  * BN batch statistics at 196,608 pixels against fp64;
  * one full-resolution (1x3x512x1024) step-2 forward+backward against the oracle on the host."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import fixtures as fx
from oracle import rap_oracle as O
from tests.test_hip_parity import close, nchw, nhwc

pytestmark = pytest.mark.gpu
N = 6


@pytest.fixture(scope="module")
def dev():
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _randn(*shape, seed, dev, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev)


@pytest.mark.parametrize("C,H,W,d,kind", [(64, 128, 256, 1, "1x3"), (64, 128, 256, 1, "3x1"),
                                          (128, 64, 128, 16, "1x3"), (128, 64, 128, 2, "3x1")])
def test_conv_linearity_and_samples(dev, C, H, W, d, kind):
    from mdil_ss_amd import ops
    ops.invalidate_packs()
    x, y = _randn(N, H, W, C, seed=1, dev=dev), _randn(N, H, W, C, seed=2, dev=dev)
    k = (3, 1) if kind == "3x1" else (1, 3)
    w = _randn(C, C, *k, seed=3, dev=dev, scale=0.1)
    taps = ops._taps_3x1(d) if kind == "3x1" else ops._taps_1x3(d)
    g = ops.make_geom(N, H, W, H, W, taps, C, H, W, C)
    wp = ops.pack_conv(w, "fwd")
    conv = lambda t: ops.tapconv(g, C, C, t, None, wp, torch.empty_like(t))
    cx, cy = conv(x), conv(y)
    cz = conv(2.0 * x - 0.5 * y)
    close(cz, 2.0 * cx - 0.5 * cy, rtol=1e-4, atol=2e-6, what="conv linearity")
    # sampled pixels against a direct fp64 contraction
    gs = torch.Generator().manual_seed(9)
    xc, wc = x.cpu().double(), w.cpu().double()
    for _ in range(24):
        n, h, ww = (int(torch.randint(0, m, (1,), generator=gs)) for m in (N, H, W))
        acc = torch.zeros(C, dtype=torch.float64)
        for t in range(3):
            hh = h + (t - 1) * d if kind == "3x1" else h
            wc_ = ww + (t - 1) * d if kind == "1x3" else ww
            if 0 <= hh < H and 0 <= wc_ < W:
                wt = wc[:, :, t, 0] if kind == "3x1" else wc[:, :, 0, t]
                acc += wt @ xc[n, hh, wc_]
        np.testing.assert_allclose(cx[n, h, ww].cpu().double().numpy(), acc.numpy(), rtol=1e-4, atol=1e-5)
    # weight gradient: additive over the batch, linear in the output gradient
    go = _randn(N, H, W, C, seed=4, dev=dev)
    b = torch.zeros(C, device=dev)
    dw_all, db_all = ops.wgrad(g, C, C, x, None, go, (0, 1, 2), C * 3, 3, w, b)
    g1 = ops.make_geom(1, H, W, H, W, taps, C, H, W, C)
    dw_sum, db_sum = torch.zeros_like(dw_all), torch.zeros_like(db_all)
    for n in range(N):
        a, c = ops.wgrad(g1, C, C, x[n:n + 1].contiguous(), None, go[n:n + 1].contiguous(), (0, 1, 2),
                         C * 3, 3, w, b)
        dw_sum += a
        db_sum += c
    close(dw_all, dw_sum, rtol=2e-4, atol=2e-5, what="wgrad additivity over images")
    close(db_all, db_sum, rtol=2e-4, atol=2e-5, what="bias-grad additivity over images")
    np.testing.assert_allclose(db_all.cpu().double().numpy(),
                               go.cpu().double().sum((0, 1, 2)).numpy(), rtol=1e-4, atol=1e-2)
    ops.invalidate_packs()


def test_bn_statistics_fullsize(dev):
    from mdil_ss_amd import ops
    C, H, W = 64, 128, 256
    z = _randn(N, H, W, C, seed=5, dev=dev) * 1.7 + 0.3
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    nbt = torch.zeros((), dtype=torch.int64, device=dev)
    coef = ops.bn_train_stats(z, gamma, beta, rm, rv, nbt)
    zd = z.cpu().double().reshape(-1, C)
    mean, var = zd.mean(0), zd.var(0, unbiased=False)
    np.testing.assert_allclose(coef[0].cpu().double().numpy(), mean.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(coef[1].cpu().double().numpy(), (var + 1e-3).rsqrt().numpy(), rtol=1e-5)
    n = zd.shape[0]
    np.testing.assert_allclose(rv.cpu().double().numpy(), (0.9 + 0.1 * var * n / (n - 1)).numpy(), rtol=1e-5)
    assert int(nbt) == 1


def test_eval_forward_batch_equals_single_images(dev):
    from mdil_ss_amd import ops
    from mdil_ss_amd.models.erfnet_RA_parallel import Net
    ops.invalidate_packs()
    torch.manual_seed(0)
    model = Net([20, 20], 2, 1)
    sd = model.state_dict()
    fx.perturb_bn(sd, seed=3)
    model.load_state_dict(sd)
    model.to(dev).eval()
    x = torch.rand(N, 3, 512, 1024, generator=torch.Generator().manual_seed(1234)).to(dev)
    with torch.no_grad():
        full = model(x, 1)
        assert tuple(full.shape) == (N, 20, 512, 1024) and bool(torch.isfinite(full).all())
        for n in (0, 3, 5):
            one = model(x[n:n + 1], 1)
            assert torch.equal(one[0], full[n]), "eval forward must not couple the images of a batch"


def test_losses_and_confusion_recombine_from_halves(dev):
    from mdil_ss_amd import ops
    H, W, C = 512, 1024, 20
    logits = _randn(N, H, W, C, seed=6, dev=dev).permute(0, 3, 1, 2)
    teach = _randn(N, H, W, C, seed=7, dev=dev).permute(0, 3, 1, 2)
    target = torch.randint(0, C, (N, H, W), generator=torch.Generator().manual_seed(8)).to(dev)
    w = torch.tensor(fx.WEIGHT_BDD, device=dev)
    ce = ops.cross_entropy2d(logits, target, w)
    parts, wsum = [], []
    for sl in (slice(0, 3), slice(3, 6)):
        parts.append(ops.cross_entropy2d(logits[sl], target[sl], w))
        wsum.append(w[target[sl]].double().sum())
    recombined = (parts[0].double() * wsum[0] + parts[1].double() * wsum[1]) / (wsum[0] + wsum[1])
    np.testing.assert_allclose(float(ce), float(recombined), rtol=2e-6)
    kl = ops.kld_prob(logits, teach)
    halves = [ops.kld_prob(logits[sl], teach[sl]) for sl in (slice(0, 3), slice(3, 6))]
    np.testing.assert_allclose(float(kl), 0.5 * (float(halves[0]) + float(halves[1])), rtol=5e-6)
    # sampled pixels of the CE against the closed form in fp64
    lp = torch.log_softmax(logits[0, :, :4, :4].double().cpu(), 0)
    t0 = target[0, :4, :4].cpu()
    manual = -(lp.gather(0, t0[None])[0] * w.cpu().double()[t0]).sum() / w.cpu().double()[t0].sum()
    small = ops.cross_entropy2d(logits[:1, :, :4, :4].contiguous(memory_format=torch.channels_last),
                                t0.to(dev)[None], w)
    np.testing.assert_allclose(float(small), float(manual), rtol=1e-5)
    cnt = lambda lg, tg: ops.argmax_confusion(lg, tg, C - 1, torch.zeros(3, C, dtype=torch.int64, device=dev))
    counts = cnt(logits, target[:, None])
    a = cnt(logits[:3], target[:3, None])
    b = cnt(logits[3:], target[3:, None])
    assert torch.equal(counts, a + b)
    tp, fp, fn = counts[0], counts[1], counts[2]
    assert int((tp + fn)[:C - 1].sum()) == int((target != C - 1).sum())     # every kept pixel once
    pred = logits.argmax(1)
    assert int(tp[:C - 1].sum()) == int(((pred == target) & (target != C - 1)).sum())


def test_fullres_single_image_step_against_oracle(dev):
    """1x3x512x1024: three forwards (train / train / eval) + CE + 0.1*KLD + backward, HIP vs the
    oracle on the host.  With 32K..524K pixels behind every BatchNorm the statistics are stable,
    so (unlike the tiny goldens) gradients agree tightly across the whole network."""
    from mdil_ss_amd import ops
    from mdil_ss_amd.models.erfnet_RA_parallel import Net
    from tests import helpers as Hh
    ops.invalidate_packs()
    torch.manual_seed(1)
    teacher = Net([20], 1, 0)
    torch.manual_seed(0)
    student = Net([20, 20], 2, 1)
    tsd = {k: v.clone() for k, v in teacher.state_dict().items()}
    fx.perturb_bn(tsd, seed=11)
    ssd = {k: v.clone() for k, v in student.state_dict().items()}
    for k, v in O.student_init_from_teacher(tsd, ssd, 1).items():
        ssd[k].copy_(v)
    teacher.load_state_dict(tsd)
    student.load_state_dict(ssd)
    student.to(dev).train()
    teacher.to(dev).eval()
    names = [n for n, _ in student.named_parameters()]
    for n, p in student.named_parameters():
        p.requires_grad = O.step2_trainable("module." + n, 1)
    for p in teacher.parameters():
        p.requires_grad = False
    images, labels = fx.make_batch(1, 512, 1024, 20, seed=77)
    gen = torch.Generator().manual_seed(5)
    m_new, m_old = O.draw_dropout_masks(1, gen), O.draw_dropout_masks(1, gen)
    q = [m_new, m_old]
    student.mask_provider = lambda n: q.pop(0)
    w = torch.tensor(fx.WEIGHT_BDD)
    xi, yi = images.to(dev), labels.to(dev)
    out_new, out_prev = student(xi, 1), student(xi, 0)
    with torch.no_grad():
        out_t = teacher(xi, 0)
    ce = ops.cross_entropy2d(out_new, yi[:, 0], w.to(dev))
    kld = ops.kld_prob(out_prev, out_t)
    (ce + 0.1 * kld).backward()
    # oracle
    torch.set_num_threads(min(32, torch.get_num_threads()))
    S = {k: v.clone() for k, v in ssd.items()}
    for n in names:
        S[n].requires_grad_(O.step2_trainable("module." + n, 1))
    o_new = O.net_forward(S, images, 1, True, m_new)
    o_prev = O.net_forward(S, images, 0, True, m_old)
    with torch.no_grad():
        o_t = O.net_forward(tsd, images, 0, False)
    o_ce, o_kld = O.ce2d(o_new, labels[:, 0], w), O.kld_prob(o_prev, o_t)
    (o_ce + 0.1 * o_kld).backward()
    close(out_t, o_t, rtol=5e-4, atol=5e-5, what="teacher logits (eval)")
    close(out_new, o_new.detach(), rtol=5e-4, atol=1e-4, what="student new-task logits")
    close(out_prev, o_prev.detach(), rtol=5e-4, atol=1e-4, what="student old-task logits")
    np.testing.assert_allclose([ce.item(), kld.item()], [o_ce.item(), o_kld.item()], rtol=2e-5)
    params = dict(student.named_parameters())
    rel, tail = [], []
    for n in names:
        gd, gc = params[n].grad, S[n].grad
        assert (gd is None) == (gc is None), n
        if gc is None or Hh.zero_grad_bias(n):
            continue
        num = float((gd.cpu().double() - gc.double()).norm())
        r = num / (float(gc.double().norm()) + 1e-12)
        rel.append(r)
        if n.startswith("decoder.1.output_conv"):
            tail.append(r)
    rel = np.array(rel)
    # ||g_hip - g_oracle|| / ||g_oracle|| per tensor.  The output conv (first in backward, no ReLU
    # crossed yet) agrees to ~1e-5.  Every ReLU the gradient then crosses flips the gates of the ~1e-5 fraction of its
    # 0.5-8 M pre-activations that lie within the fp32 forward error of zero; after the ~40 ReLU
    # layers down to the stem that is a fraction f ~ 3e-4 of changed paths, i.e. a relative
    # vector error ~ sqrt(f) ~ 1.5 % (measured 1.2-1.7 %), with both sides equally "right".
    assert max(tail) < 1e-4, tail
    assert np.median(rel) < 3e-2 and rel.max() < 6e-2, (np.median(rel), rel.max())


def test_fused_head_equals_the_unfused_path_fullsize(dev):
    """BASELINE config 3's head: features [6,256,512,16] -> logits [6,20,512,1024] -> CE / KLD.
    The fused operators (csrc/head.hip: no logits in memory, backward recomputes them, weight
    gradient on MFMA over 786,432 pixels in 1,024 block partials) against the unfused HIP path
    (mdil_outconv_fwd + mdil_ce_loss / mdil_kld_loss + tap-conv dgrad / wgrad, each verified
    against the oracle at small sizes): losses, feature gradients, weight and bias gradients; and
    the losses recombine from the batch's halves (fixed-order partial sums)."""
    from mdil_ss_amd import ops
    ops.invalidate_packs()
    H, W, nc = 256, 512, 20
    x = F.relu(_randn(N, H, W, 16, seed=1, dev=dev))
    xt = F.relu(_randn(N, H, W, 16, seed=2, dev=dev))
    w, b = _randn(16, nc, 2, 2, seed=3, dev=dev, scale=0.3), _randn(nc, seed=4, dev=dev, scale=0.2)
    wt, bt = _randn(16, nc, 2, 2, seed=5, dev=dev, scale=0.3), _randn(nc, seed=6, dev=dev, scale=0.2)
    _, lab = fx.make_batch(N, 2 * H, 2 * W, nc, seed=7, block=16)
    lab = lab[:, 0].to(dev)
    weight = torch.tensor(fx.WEIGHT_BDD, device=dev)
    res = {}
    for fused in (True, False):
        xs, ws, bs = (t.clone().requires_grad_(True) for t in (x, w, b))
        if fused:
            ce = ops.head_ce(xs, ws, bs, lab, weight)
            kld = ops.head_kld(xs, ws, bs, xt, wt, bt)
        else:
            logits = ops.OutFn.apply(xs, ws, bs).permute(0, 3, 1, 2)
            with torch.no_grad():
                lt = ops.OutFn.apply(xt, wt, bt).permute(0, 3, 1, 2)
            ce = ops.cross_entropy2d(logits, lab, weight)
            kld = ops.kld_prob(logits, lt)
        (ce + 0.1 * kld).backward()
        res[fused] = (float(ce), float(kld), xs.grad, ws.grad, bs.grad)
    ops.invalidate_packs()
    assert res[True][0] == pytest.approx(res[False][0], rel=1e-5)
    assert res[True][1] == pytest.approx(res[False][1], rel=1e-4, abs=1e-8)
    close(res[True][2], res[False][2], rtol=1e-3, atol=1e-4, what="full-size fused head gx")
    close(res[True][3], res[False][3], rtol=1e-3, atol=1e-4, what="full-size fused head dw")
    close(res[True][4], res[False][4], rtol=1e-3, atol=2e-4, what="full-size fused head db")
    # the KLD mean over the batch = mean of the halves' means (equal sizes)
    k1 = float(ops.head_kld(x[:3].contiguous(), w, b, xt[:3].contiguous(), wt, bt))
    k2 = float(ops.head_kld(x[3:].contiguous(), w, b, xt[3:].contiguous(), wt, bt))
    assert 0.5 * (k1 + k2) == pytest.approx(res[True][1], rel=1e-4, abs=1e-8)
