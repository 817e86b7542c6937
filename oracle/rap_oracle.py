"""CPU oracle for the ERFNet-RAP step-2 training path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (fp32, CPU) *restatement* of the algorithm the reference
implements with stock ``nn.Module``s.  It is the checker, never the product: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The shipped path (``mdil_ss_amd``) never imports anything under ``oracle/`` and
fails loudly when its HIP extension is missing.

Pinning: the reference holds no tests / golden vectors of its own, so the
oracle is pinned against fixtures produced by *importing the reference itself* in the build
container (``tools/gen_golden.py`` -> ``tests/golden/*.npz``); ``tests/test_oracle_golden.py``
checks every function here against them.

Everything is functional: parameters and buffers travel in flat ``dict[str, Tensor]`` keyed by
the reference's state-dict names (without the DataParallel ``module.`` prefix).

Reference citations are ``file:line`` into the upstream repo (prachigarg23/MDIL-SS).
"""
from __future__ import annotations

import math
import re
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

BN_EPS = 1e-3          # models/erfnet_RA_parallel.py:19,36,44,77,86,157
BN_MOMENTUM = 0.1      # nn.BatchNorm2d default

# Encoder layout, models/erfnet_RA_parallel.py:123-141:  index in ``encoder.layers``.
#   layers[0] = down 16->64, layers[1..5] = RAP(64, p=.03, d=1), layers[6] = down 64->128,
#   layers[7..14] = RAP(128, p=.3, d=2,4,8,16,2,4,8,16)
ENC_RAP = [(i, 64, 0.03, 1) for i in range(1, 6)] + \
          [(7 + i, 128, 0.3, d) for i, d in enumerate([2, 4, 8, 16, 2, 4, 8, 16])]
ENC_DOWN = [(0, 16, 64), (6, 64, 128)]
# Decoder layout, models/erfnet_RA_parallel.py:165-180
DEC_UP = [(0, 128, 64), (3, 64, 16)]
DEC_NB = [(1, 64), (2, 64), (4, 16), (5, 16)]


# ----------------------------------------------------------------------------------------------
# state-dict layout
# ----------------------------------------------------------------------------------------------
def _bn_entries(prefix: str, c: int):
    return [(prefix + ".weight", (c,)), (prefix + ".bias", (c,)),
            (prefix + ".running_mean", (c,)), (prefix + ".running_var", (c,)),
            (prefix + ".num_batches_tracked", ())]


def state_layout(num_classes: Sequence[int], nb_tasks: int) -> List[Tuple[str, tuple]]:
    """Ordered (name, shape) list == ``Net(num_classes, nb_tasks).state_dict()`` of the
    reference (models/erfnet_RA_parallel.py:194-205; registration order of the sub-modules)."""
    out: List[Tuple[str, tuple]] = []

    def down(prefix, cin, cout):
        out.append((prefix + ".conv.weight", (cout - cin, cin, 3, 3)))
        out.append((prefix + ".conv.bias", (cout - cin,)))
        for t in range(nb_tasks):
            out.extend(_bn_entries(f"{prefix}.bn_ini.{t}", cout))

    def conv(prefix, c, kh, kw):
        out.append((prefix + ".weight", (c, c, kh, kw)))
        out.append((prefix + ".bias", (c,)))

    down("encoder.initial_block", 3, 16)
    for li in range(15):
        p = f"encoder.layers.{li}"
        if li == 0:
            down(p, 16, 64)
        elif li == 6:
            down(p, 64, 128)
        else:
            c = 64 if li < 6 else 128
            conv(p + ".conv3x1_1", c, 3, 1)
            conv(p + ".conv1x3_1", c, 1, 3)
            for t in range(nb_tasks):
                conv(f"{p}.parallel_conv_1.{t}", c, 1, 1)
            for t in range(nb_tasks):
                out.extend(_bn_entries(f"{p}.bns_1.{t}", c))
            conv(p + ".conv3x1_2", c, 3, 1)
            conv(p + ".conv1x3_2", c, 1, 3)
            for t in range(nb_tasks):
                conv(f"{p}.parallel_conv_2.{t}", c, 1, 1)
            for t in range(nb_tasks):
                out.extend(_bn_entries(f"{p}.bns_2.{t}", c))
    for t in range(nb_tasks):
        d = f"decoder.{t}"
        for li in range(6):
            p = f"{d}.layers.{li}"
            if li in (0, 3):
                cin, cout = (128, 64) if li == 0 else (64, 16)
                out.append((p + ".conv.weight", (cin, cout, 3, 3)))
                out.append((p + ".conv.bias", (cout,)))
                out.extend(_bn_entries(p + ".bn", cout))
            else:
                c = 64 if li < 3 else 16
                conv(p + ".conv3x1_1", c, 3, 1)
                conv(p + ".conv1x3_1", c, 1, 3)
                out.extend(_bn_entries(p + ".bn1", c))
                conv(p + ".conv3x1_2", c, 3, 1)
                conv(p + ".conv1x3_2", c, 1, 3)
                out.extend(_bn_entries(p + ".bn2", c))
        out.append((d + ".output_conv.weight", (16, num_classes[t], 2, 2)))
        out.append((d + ".output_conv.bias", (num_classes[t],)))
    return out


def is_buffer(name: str) -> bool:
    return name.endswith(("running_mean", "running_var", "num_batches_tracked"))


# ----------------------------------------------------------------------------------------------
# forward
# ----------------------------------------------------------------------------------------------
def _bn(S: Dict[str, torch.Tensor], prefix: str, x: torch.Tensor, train: bool) -> torch.Tensor:
    """nn.BatchNorm2d(eps=1e-3): batch statistics + running update (unbiased var) in train mode,
    running statistics in eval mode.  Buffers in ``S`` are updated in place like the module's."""
    rm, rv = S[prefix + ".running_mean"], S[prefix + ".running_var"]
    if train:
        S[prefix + ".num_batches_tracked"] += 1
    return F.batch_norm(x, rm, rv, S[prefix + ".weight"], S[prefix + ".bias"],
                        training=train, momentum=BN_MOMENTUM, eps=BN_EPS)


def _act(x, gates):
    """ReLU.  ``gates`` (tests only): a list of recorded 0/1 masks, consumed in call order -- the
    ReLU then lets through exactly the recorded elements (``x * mask``) instead of ``x > 0``.
    A pre-activation closer to zero than the fp32 forward error of another implementation can
    sit on the other side of zero there; forcing that implementation's gates removes this
    (step-function) ambiguity from a backward comparison without touching anything else."""
    if gates is None:
        return F.relu(x)
    return x * gates.pop(0).to(x.dtype)


def _down(S, p, x, task, train, gates=None):
    # models/erfnet_RA_parallel.py:21-25 : cat([conv3x3 s2 p1 (x), maxpool2x2 (x)], C) -> bn -> relu
    y = torch.cat([F.conv2d(x, S[p + ".conv.weight"], S[p + ".conv.bias"], stride=2, padding=1),
                   F.max_pool2d(x, 2, stride=2)], 1)
    return _act(_bn(S, f"{p}.bn_ini.{task}", y, train), gates)


def _factor_pair(S, p, idx, x, d, gates=None):
    # 3x1 (dilated along H) -> relu -> 1x3 (dilated along W); :72-73,79-82 / :93-95,103-105
    a = _act(F.conv2d(x, S[f"{p}.conv3x1_{idx}.weight"], S[f"{p}.conv3x1_{idx}.bias"],
                      padding=(d, 0), dilation=(d, 1)), gates)
    return F.conv2d(a, S[f"{p}.conv1x3_{idx}.weight"], S[f"{p}.conv1x3_{idx}.bias"],
                    padding=(0, d), dilation=(1, d))


def _rap(S, p, x, task, train, d, mask, gates=None):
    # models/erfnet_RA_parallel.py:90-113
    z1 = _factor_pair(S, p, 1, x, 1, gates) + F.conv2d(x, S[f"{p}.parallel_conv_1.{task}.weight"],
                                                       S[f"{p}.parallel_conv_1.{task}.bias"])
    u = _act(_bn(S, f"{p}.bns_1.{task}", z1, train), gates)
    z2 = _factor_pair(S, p, 2, u, d, gates) + F.conv2d(u, S[f"{p}.parallel_conv_2.{task}.weight"],
                                                       S[f"{p}.parallel_conv_2.{task}.bias"])
    y = _bn(S, f"{p}.bns_2.{task}", z2, train)
    if train and mask is not None:          # Dropout2d: per-(n,c) keep/(1-p) scale, :110-111
        y = y * mask
    return _act(y + x, gates)


def _nb1d_d(S, p, x, train, d=1, gates=None):
    # models/erfnet_RA_parallel.py:48-64 (decoder blocks: dropprob 0 -> dropout skipped :61)
    u = _act(_bn(S, p + ".bn1", _factor_pair(S, p, 1, x, 1, gates), train), gates)
    y = _bn(S, p + ".bn2", _factor_pair(S, p, 2, u, d, gates), train)
    return _act(y + x, gates)


def _nb1d(S, p, x, train, gates=None):
    return _nb1d_d(S, p, x, train, 1, gates)


def _up(S, p, x, train, gates=None):
    # models/erfnet_RA_parallel.py:159-162
    y = F.conv_transpose2d(x, S[p + ".conv.weight"], S[p + ".conv.bias"], stride=2, padding=1,
                           output_padding=1)
    return _act(_bn(S, p + ".bn", y, train), gates)


def draw_dropout_masks(n: int, generator: Optional[torch.Generator] = None) -> List[torch.Tensor]:
    """The 13 encoder Dropout2d masks of one train-mode forward, drawn exactly as
    ``nn.Dropout2d`` does on CPU: ``empty(N,C,1,1).bernoulli_(1-p).div_(1-p)`` in block order
    (probed)."""
    masks = []
    for _, c, p, _ in ENC_RAP:
        m = torch.empty(n, c, 1, 1).bernoulli_(1 - p, generator=generator).div_(1 - p)
        masks.append(m)
    return masks


def net_forward(S: Dict[str, torch.Tensor], x: torch.Tensor, task: int, train: bool,
                masks: Optional[List[torch.Tensor]] = None,
                collect: Optional[Dict[str, torch.Tensor]] = None,
                gates: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
    """``Net.forward(input, task)`` (models/erfnet_RA_parallel.py:207-212).  ``gates``: see
    ``_act`` (recorded ReLU masks of another implementation's forward, in call order)."""
    y = _down(S, "encoder.initial_block", x, task, train, gates)
    if collect is not None:
        collect["encoder.initial_block"] = y
    rap = {li: (c, p, d) for li, c, p, d in ENC_RAP}
    k = 0
    for li in range(15):
        p = f"encoder.layers.{li}"
        if li in (0, 6):
            y = _down(S, p, y, task, train, gates)
        else:
            y = _rap(S, p, y, task, train, rap[li][2], None if masks is None else masks[k], gates)
            k += 1
        if collect is not None:
            collect[p] = y
    dp = f"decoder.{task}"
    for li in range(6):
        p = f"{dp}.layers.{li}"
        y = _up(S, p, y, train, gates) if li in (0, 3) else _nb1d(S, p, y, train, gates)
        if collect is not None:
            collect[p] = y
    return F.conv_transpose2d(y, S[dp + ".output_conv.weight"], S[dp + ".output_conv.bias"],
                              stride=2)


# ----------------------------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------------------------
def ce2d(logits: torch.Tensor, target: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """CrossEntropyLoss2d (train_new_task_step2.py:84-92): weighted-mean NLL of log_softmax.
    = -sum_p w[y_p] * logp_p[y_p] / sum_p w[y_p]."""
    logp = F.log_softmax(logits, dim=1)
    picked = logp.gather(1, target.unsqueeze(1)).squeeze(1)
    w = weight[target]
    return -(w * picked).sum() / w.sum()


def kld_prob(student_logits: torch.Tensor, teacher_logits: torch.Tensor) -> torch.Tensor:
    """train_new_task_step2.py:241,296-297: ``KLDivLoss()(softmax(s), softmax(t))`` -- the
    *probabilities* (not log-probs) are passed as input, reduction 'mean' over ALL elements:
    mean( t * (log t - p_s) )."""
    ps = F.softmax(student_logits, dim=1)
    t = F.softmax(teacher_logits, dim=1)
    return (t * (t.log() - ps)).mean()


# ----------------------------------------------------------------------------------------------
# trainer logic
# ----------------------------------------------------------------------------------------------
def is_shared(n: str) -> bool:                         # train_new_task_step2.py:95-96
    return "encoder" in n and "parallel_conv" not in n and "bn" not in n


def is_ds_curr(n: str, t: int) -> bool:                # train_new_task_step2.py:99-105
    if f"decoder.{t}" in n:
        return True
    if "encoder" in n and ("bn" in n or "parallel_conv" in n):
        return f".{t}.weight" in n or f".{t}.bias" in n
    return False


def step2_trainable(n: str, t: int) -> bool:           # train_new_task_step2.py:205-215
    if "decoder" in n:
        return f"decoder.{t}" in n
    if "encoder" in n and ("bn" in n or "parallel_conv" in n):
        return f".{t}.weight" in n or f".{t}.bias" in n
    return True


def poly_lr(base: float, epoch: int, num_epochs: int) -> float:     # :244-245,254
    return base * pow(1 - (epoch - 1) / num_epochs, 0.9)


def student_init_from_teacher(old: Dict[str, torch.Tensor], student: Dict[str, torch.Tensor],
                              t: int) -> Dict[str, torch.Tensor]:
    """train_new_task_step2.py:497-530 -- returns the dict that is ``load_state_dict``-ed
    (strict=False) into the student: common keys, DS(t-1)->DS(t) weight/bias (running stats are
    NOT copied), decoder(t-1)->decoder(t) except output_conv (buffers included).  Keys may carry
    any common prefix (e.g. ``module.``)."""
    new = {k: v for k, v in old.items() if k in student}
    for k, v in old.items():
        if "encoder" in k:
            if "parallel_conv" in k or "bn" in k:
                if f".{t-1}.weight" in k:
                    new[re.sub(f".{t-1}.weight", f".{t}.weight", k)] = v
                elif f".{t-1}.bias" in k:
                    new[re.sub(f".{t-1}.bias", f".{t}.bias", k)] = v
        elif "decoder" in k and "output_conv" not in k:
            new[re.sub(f"decoder.{t-1}", f"decoder.{t}", k)] = v
    return new


def adam_l2_step(p, g, m, v, step: int, lr: float, wd: float = 1e-4, b1: float = 0.9,
                 b2: float = 0.999, eps: float = 1e-8):
    """torch.optim.Adam (L2 weight decay added to the gradient, not AdamW), single tensor,
    in place.  train_new_task_step2.py:237-239."""
    g = g + wd * p
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def iou_counts(pred: torch.Tensor, target: torch.Tensor, n_classes: int, ignore: int):
    """iouEval.addBatch (iouEval.py:21-70) for index inputs: per-class tp/fp/fn (float64) over the
    first ``ignore`` classes, pixels whose target == ignore dropped from fp."""
    pred = pred.reshape(-1)
    target = target.reshape(-1)
    k = ignore if ignore >= 0 else n_classes
    tp = torch.zeros(k, dtype=torch.float64)
    fp = torch.zeros(k, dtype=torch.float64)
    fn = torch.zeros(k, dtype=torch.float64)
    for c in range(k):
        pc, tc = pred == c, target == c
        tp[c] = (pc & tc).sum()
        fp[c] = (pc & ~tc & (target != ignore)).sum() if ignore >= 0 else (pc & ~tc).sum()
        fn[c] = (~pc & tc).sum()
    return tp, fp, fn


def miou(tp, fp, fn):                                  # iouEval.py:72-77
    iou = tp / (tp + fp + fn + 1e-15)
    return iou.mean(), iou


def step2_iteration(student: Dict[str, torch.Tensor], teacher: Dict[str, torch.Tensor],
                    images: torch.Tensor, labels: torch.Tensor, weight: torch.Tensor, t: int,
                    lambdac: float, masks_new, masks_old, gates_new=None, gates_old=None,
                    ce_scale: float = 1.0):
    """One hot-loop iteration up to the gradients (train_new_task_step2.py:285-304).
    ``student`` tensors that are trainable must have requires_grad=True.  Returns
    (ce, kld, total, logits_new, logits_prev_task, logits_prev_model); grads land in .grad."""
    out_new = net_forward(student, images, t, True, masks_new, gates=gates_new)
    out_prev_task = net_forward(student, images, t - 1, True, masks_old, gates=gates_old)
    with torch.no_grad():
        out_prev_model = net_forward(teacher, images, t - 1, False)
    ce = ce2d(out_new, labels[:, 0], weight)
    kld = kld_prob(out_prev_task, out_prev_model)
    total = ce_scale * ce + lambdac * kld      # ce_scale = 0: the KD graph's backward alone
    total.backward()
    return ce, kld, total, out_new, out_prev_task, out_prev_model


# ----------------------------------------------------------------------------------------------
# step 3 (two old domains): train_new_task_step3.py:303-356
# ----------------------------------------------------------------------------------------------
def step3_iteration(student, teacher, images, labels, weight, t, lambdac, masks, opt_step):
    """One step-3 iteration = TWO optimizer steps (train_new_task_step3.py:317-356):
      (A) CE on the new task -> backward -> optimizer step;
      (B) KD on both old tasks, lambdac*(kld(t-1)+kld(t-2)) -> backward -> optimizer step.
    Quirks preserved: the old model is never put in eval mode in that file (only ``model.train()``
    at :301), so the teacher runs with batch statistics, updates its own running statistics and
    has dropout active; ``opt_step(tag)`` is called after each backward with .grad populated (under
    torch>=2 ``zero_grad`` sets grads to None, so step (B) only touches parameters the KD graph
    reaches).  ``masks``: dict with the 13-mask lists 'new', 'prev1', 'prev0', 'teach1', 'teach0'.
    -> (ce, kld_{t-1}, kld_{t-2}, logits_new)."""
    names = [n for n in student if not is_buffer(n)]

    def zero():
        for n in names:
            student[n].grad = None

    out = net_forward(student, images, t, True, masks["new"])
    ce = ce2d(out, labels[:, 0], weight)
    zero()
    ce.backward()
    opt_step("ce")
    p1 = net_forward(student, images, t - 1, True, masks["prev1"])
    p0 = net_forward(student, images, t - 2, True, masks["prev0"])
    with torch.no_grad():
        t1 = net_forward(teacher, images, t - 1, True, masks["teach1"])
        t0 = net_forward(teacher, images, t - 2, True, masks["teach0"])
    k1, k0 = kld_prob(p1, t1), kld_prob(p0, t0)
    kd = lambdac * (k1 + k0)
    zero()
    kd.backward()
    opt_step("kd")
    return ce.detach(), k1.detach(), k0.detach(), out.detach()


# ----------------------------------------------------------------------------------------------
# multi-task joint model (models/erfnet_multi_task.py) and its round-robin loop
# ----------------------------------------------------------------------------------------------
def mt_forward(S, x, task, train, masks=None, dec_prefix=None):
    """``Net.forward(input, task)`` of models/erfnet_multi_task.py:153-160: shared encoder
    (plain non_bottleneck_1d blocks with Dropout2d, one BatchNorm per layer), decoder ``task``.
    ``dec_prefix`` selects another head by state-dict prefix: ``decoder`` (models/erfnet.py:141-149),
    ``decoder_old`` / ``decoder_new`` (models/erfnet_ftp1.py:134-151), ``decoder_old1`` /
    ``decoder_old2`` / ``decoder_new`` (models/erfnet_ftp2.py:134-152) -- same blocks everywhere."""
    def down(p, x):                                                   # :22-25
        y = torch.cat([F.conv2d(x, S[p + ".conv.weight"], S[p + ".conv.bias"], stride=2, padding=1),
                       F.max_pool2d(x, 2, stride=2)], 1)
        return F.relu(_bn(S, p + ".bn", y, train))

    def block(p, x, d, mask):                                         # :48-64
        u = F.relu(_bn(S, p + ".bn1", _factor_pair(S, p, 1, x, 1), train))
        y = _bn(S, p + ".bn2", _factor_pair(S, p, 2, u, d), train)
        if train and mask is not None:
            y = y * mask
        return F.relu(y + x)

    y = down("encoder.initial_block", x)
    dil = {li: d for li, _, _, d in ENC_RAP}
    k = 0
    for li in range(15):
        p = f"encoder.layers.{li}"
        if li in (0, 6):
            y = down(p, y)
        else:
            y = block(p, y, dil[li], None if masks is None else masks[k])
            k += 1
    dp = dec_prefix if dec_prefix is not None else f"decoder.{task}"
    for li in range(6):
        p = f"{dp}.layers.{li}"
        y = _up(S, p, y, train) if li in (0, 3) else _nb1d(S, p, y, train)
    return F.conv_transpose2d(y, S[dp + ".output_conv.weight"], S[dp + ".output_conv.bias"],
                              stride=2)


def mt_is_shared(n: str) -> bool:                      # train_multi_task.py:107
    return "encoder" in n


def mt_is_ds(n: str) -> bool:                          # train_multi_task.py:110
    return "decoder" in n


def mt_round(S, batches, weights, masks, opt_step):
    """One inner-loop pass of train_multi_task.py:249-265: for every dataset in order --
    forward through its head, zero_grad (grads -> None), CE, backward, optimizer step (only the
    encoder and that head have gradients).  ``batches[i] = (images, labels)``.  -> [ce_i]."""
    names = [n for n in S if not is_buffer(n)]
    out = []
    for ind, (images, labels) in enumerate(batches):
        logits = mt_forward(S, images, ind, True, masks[ind])
        for n in names:
            S[n].grad = None
        ce = ce2d(logits, labels[:, 0], weights[ind])
        ce.backward()
        opt_step(ind)
        out.append(ce.detach())
    return out
