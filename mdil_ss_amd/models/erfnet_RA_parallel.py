"""ERFNet with per-domain parallel residual adapters (RAP) and domain-specific BatchNorm --
MI355X-native implementation behind the reference's module surface.

Drop-in for ``models/erfnet_RA_parallel.py`` of prachigarg23/MDIL-SS (lines 13-212):
``Net(num_classes: list, nb_tasks, cur_task)``, ``forward(input[N,3,H,W], task) ->
logits[N, num_classes[task], H, W]``, identical parameter / buffer names, shapes, registration
order and initial values under the same ``torch.manual_seed`` (so reference checkpoints load
with ``load_state_dict`` and the trainer's name-substring freeze / grouping rules keep working).

What is different is everything underneath: the sub-modules below are parameter *containers*
(stock ``nn.Conv2d`` / ``nn.BatchNorm2d`` objects are instantiated only to reproduce the
reference's initialisers and state-dict layout; their ``forward`` is never called).  The math
runs in the HIP kernels of libmdil_hip.so through the block-level autograd Functions in
``mdil_ss_amd.ops`` on NHWC tensors.  There is no eager fallback.
"""
import torch
import torch.nn as nn

from .. import ops

current_task = 0  # kept for parity with the reference's module-level global (:11)


class _Holder(nn.Module):
    """A module that only owns parameters; calling it is a bug."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter container: the block's HIP path consumes these tensors")


def _bn(c):
    return nn.BatchNorm2d(c, eps=1e-3)


def _bn_bufs(bn):
    return bn.running_mean, bn.running_var, bn.num_batches_tracked


class DownsamplerBlock(_Holder):
    # reference :13-25
    def __init__(self, ninput, noutput, nb_tasks=1):
        super().__init__()
        self.conv = nn.Conv2d(ninput, noutput - ninput, (3, 3), stride=2, padding=1, bias=True)
        self.bn_ini = nn.ModuleList([_bn(noutput) for _ in range(nb_tasks)])

    def run(self, x, task, train, drop=None, links=(None, None)):
        """``links = (boundary in front, boundary behind)`` (ops.Boundary, optional): the explicit
        chain of the block-boundary BatchNorm-backward fusion."""
        bn = self.bn_ini[task]
        return ops.DownFn.apply(x, self.conv.weight, self.conv.bias, bn.weight, bn.bias,
                                *_bn_bufs(bn), train, links[1])


class non_bottleneck_1d(_Holder):
    # reference :28-64 (decoder blocks; dropprob is 0 there so dropout never fires, :61)
    def __init__(self, chann, dropprob, dilated):
        super().__init__()
        self.conv3x1_1 = nn.Conv2d(chann, chann, (3, 1), stride=1, padding=(1, 0), bias=True)
        self.conv1x3_1 = nn.Conv2d(chann, chann, (1, 3), stride=1, padding=(0, 1), bias=True)
        self.bn1 = _bn(chann)
        self.conv3x1_2 = nn.Conv2d(chann, chann, (3, 1), stride=1, padding=(1 * dilated, 0),
                                   bias=True, dilation=(dilated, 1))
        self.conv1x3_2 = nn.Conv2d(chann, chann, (1, 3), stride=1, padding=(0, 1 * dilated),
                                   bias=True, dilation=(1, dilated))
        self.bn2 = _bn(chann)
        self.dropout = nn.Dropout2d(dropprob)
        self.dilated = dilated

    def run(self, x, task, train, drop=None, links=(None, None)):
        if self.dropout.p != 0:
            raise RuntimeError("non_bottleneck_1d with dropout is not on the hot path")
        bufs = _bn_bufs(self.bn1) + _bn_bufs(self.bn2)
        return ops.NbFn.apply(
            x, self.conv3x1_1.weight, self.conv3x1_1.bias, self.conv1x3_1.weight,
            self.conv1x3_1.bias, None, None, self.bn1.weight, self.bn1.bias,
            self.conv3x1_2.weight, self.conv3x1_2.bias, self.conv1x3_2.weight,
            self.conv1x3_2.bias, None, None, self.bn2.weight, self.bn2.bias, bufs, None,
            self.dilated, train, links[0], links[1])


class non_bottleneck_1d_RAP(_Holder):
    # reference :67-113
    def __init__(self, chann, dropprob, dilated, nb_tasks=1):
        super().__init__()
        self.conv3x1_1 = nn.Conv2d(chann, chann, (3, 1), stride=1, padding=(1, 0), bias=True)
        self.conv1x3_1 = nn.Conv2d(chann, chann, (1, 3), stride=1, padding=(0, 1), bias=True)
        self.parallel_conv_1 = nn.ModuleList(
            [nn.Conv2d(chann, chann, kernel_size=1, stride=1, padding=0, bias=True)
             for _ in range(nb_tasks)])
        self.bns_1 = nn.ModuleList([_bn(chann) for _ in range(nb_tasks)])
        self.conv3x1_2 = nn.Conv2d(chann, chann, (3, 1), stride=1, padding=(1 * dilated, 0),
                                   bias=True, dilation=(dilated, 1))
        self.conv1x3_2 = nn.Conv2d(chann, chann, (1, 3), stride=1, padding=(0, 1 * dilated),
                                   bias=True, dilation=(1, dilated))
        self.parallel_conv_2 = nn.ModuleList(
            [nn.Conv2d(chann, chann, kernel_size=1, stride=1, padding=0, bias=True)
             for _ in range(nb_tasks)])
        self.bns_2 = nn.ModuleList([_bn(chann) for _ in range(nb_tasks)])
        self.dropout = nn.Dropout2d(dropprob)
        self.dilated = dilated
        self.chann = chann

    def _operands(self, task):
        p1, p2 = self.parallel_conv_1[task], self.parallel_conv_2[task]
        b1, b2 = self.bns_1[task], self.bns_2[task]
        return (self.conv3x1_1.weight, self.conv3x1_1.bias, self.conv1x3_1.weight,
                self.conv1x3_1.bias, p1.weight, p1.bias, b1.weight, b1.bias,
                self.conv3x1_2.weight, self.conv3x1_2.bias, self.conv1x3_2.weight,
                self.conv1x3_2.bias, p2.weight, p2.bias, b2.weight, b2.bias,
                _bn_bufs(b1) + _bn_bufs(b2))

    def run(self, x, task, train, drop=None, links=(None, None)):
        if not (train and self.dropout.p != 0):
            drop = None
        return ops.NbFn.apply(x, *self._operands(task), drop, self.dilated, train, links[0], links[1])


class Encoder(_Holder):
    # reference :123-149
    def __init__(self, nb_tasks=1):
        super().__init__()
        self.initial_block = DownsamplerBlock(3, 16, nb_tasks)
        self.layers = nn.ModuleList()
        self.layers.append(DownsamplerBlock(16, 64, nb_tasks))
        for _ in range(5):
            self.layers.append(non_bottleneck_1d_RAP(64, 0.03, 1, nb_tasks))
        self.layers.append(DownsamplerBlock(64, 128, nb_tasks))
        for _ in range(2):
            for d in (2, 4, 8, 16):
                self.layers.append(non_bottleneck_1d_RAP(128, 0.3, d, nb_tasks))

    def dropout_blocks(self):
        return [m for m in self.layers if isinstance(m, non_bottleneck_1d_RAP)]

    def run(self, x, task, train, masks, links=None):
        """``links``: ops.boundaries(len(layers)); links[-1] is handed on to the decoder."""
        L = links if links is not None else ops.boundaries(len(self.layers))
        y = self.initial_block.run(x, task, train, links=(None, L[0]))
        k = 0
        for i, layer in enumerate(self.layers):
            if isinstance(layer, DownsamplerBlock):
                y = layer.run(y, task, train, links=(L[i], L[i + 1]))
            else:
                y = layer.run(y, task, train, None if masks is None else masks[k], links=(L[i], L[i + 1]))
                k += 1
        return y


class UpsamplerBlock(_Holder):
    # reference :152-162
    def __init__(self, ninput, noutput):
        super().__init__()
        self.conv = nn.ConvTranspose2d(ninput, noutput, 3, stride=2, padding=1, output_padding=1,
                                       bias=True)
        self.bn = _bn(noutput)

    def run(self, x, task, train, drop=None, links=(None, None)):
        return ops.UpFn.apply(x, self.conv.weight, self.conv.bias, self.bn.weight, self.bn.bias,
                              *_bn_bufs(self.bn), train, links[1])


class Decoder(_Holder):
    # reference :165-190
    def __init__(self, num_classes):
        super().__init__()
        self.layers = nn.ModuleList()
        self.layers.append(UpsamplerBlock(128, 64))
        self.layers.append(non_bottleneck_1d(64, 0, 1))
        self.layers.append(non_bottleneck_1d(64, 0, 1))
        self.layers.append(UpsamplerBlock(64, 16))
        self.layers.append(non_bottleneck_1d(16, 0, 1))
        self.layers.append(non_bottleneck_1d(16, 0, 1))
        self.output_conv = nn.ConvTranspose2d(16, num_classes, 2, stride=2, padding=0,
                                              output_padding=0, bias=True)

    def run(self, x, train, link_in=None):
        y = x
        L = [link_in] + ops.boundaries(len(self.layers) - 1)
        for i, layer in enumerate(self.layers):
            y = layer.run(y, 0, train, links=(L[i], L[i + 1]))
        return ops.OutFn.apply(y, self.output_conv.weight, self.output_conv.bias)


class Net(nn.Module):
    """ERFNet-RAP.  ``forward(input, task)`` keeps the reference contract (:207-212); dropout
    masks come from ``mask_provider`` (default: torch's device RNG, one Bernoulli draw per
    encoder block like nn.Dropout2d) so tests can replay recorded masks."""

    def __init__(self, num_classes=[20], nb_tasks=1, cur_task=0):
        super().__init__()
        global current_task
        current_task = cur_task
        self.encoder = Encoder(nb_tasks)
        self.decoder = nn.ModuleList([Decoder(num_classes[i]) for i in range(nb_tasks)])
        self.mask_provider = None
        self.mask_generator = None      # engines install a per-rank generator under data parallelism

    def draw_masks(self, n, device):
        """13 Dropout2d masks [N, C] (already divided by 1-p), block order."""
        if self.mask_provider is not None:
            return [m.to(device=device, dtype=torch.float32).reshape(n, -1).contiguous()
                    for m in self.mask_provider(n)]
        # one uniform draw for all 13 blocks (3 small launches instead of 26): element e of block
        # b is kept with probability 1 - p_b and scaled by 1 / (1 - p_b), as nn.Dropout2d does
        key = (n, str(device))
        plan = self._mask_plan.get(key) if hasattr(self, "_mask_plan") else None
        if plan is None:
            if not hasattr(self, "_mask_plan"):
                self._mask_plan = {}
            blocks = self.encoder.dropout_blocks()
            keep = torch.cat([torch.full((n * b.chann,), 1.0 - b.dropout.p) for b in blocks]).to(device)
            sizes = [n * b.chann for b in blocks]
            plan = self._mask_plan[key] = (keep, 1.0 / keep, sizes, [b.chann for b in blocks])
        keep, inv, sizes, chans = plan
        u = torch.rand(keep.numel(), device=device, generator=self.mask_generator)
        # (host tensors: only the multi-process CPU tests of the per-rank generator draw masks there)
        flat = ops.dropout_factors(u, keep, inv) if u.is_cuda else (u < keep).to(torch.float32).mul_(inv)
        return [m.view(n, c) for m, c in zip(flat.split(sizes), chans)]

    def plan(self, task, masks=None, head=True):
        """The forward pass as a list of block callables ``y = f(y)`` on NHWC tensors (the last
        one yields NHWC logits), so a scheduler can advance several forwards in lock step on
        different streams (engine.Step2Engine).  ``head=False`` stops before ``output_conv``: the
        plan then yields the decoder's 16-channel features for the fused head + loss operators
        (ops.head_ce / ops.head_kld), which take ``head_params(task)``."""
        train = self.training
        enc, dec = self.encoder, self.decoder[task]
        # one boundary per block boundary of THIS forward pass (ops.Boundary): B[j] sits behind step j
        B = ops.boundaries(len(enc.layers) + len(dec.layers))
        steps = [lambda y: enc.initial_block.run(y, task, train, links=(None, B[0]))]
        k = 0
        for i, layer in enumerate(enc.layers):
            if isinstance(layer, DownsamplerBlock):
                steps.append(lambda y, L=layer, ln=(B[i], B[i + 1]): L.run(y, task, train, links=ln))
            else:
                steps.append(lambda y, L=layer, m=(None if masks is None else masks[k]), ln=(B[i], B[i + 1]):
                             L.run(y, task, train, m, links=ln))
                k += 1
        ne = len(enc.layers)
        for i, layer in enumerate(dec.layers):
            steps.append(lambda y, L=layer, ln=(B[ne + i], B[ne + i + 1]): L.run(y, 0, train, links=ln))
        if head:
            steps.append(lambda y: ops.OutFn.apply(y, dec.output_conv.weight, dec.output_conv.bias))
        return steps

    def head_params(self, task):
        oc = self.decoder[task].output_conv
        return oc.weight, oc.bias

    def features(self, input, task):
        """forward() without ``output_conv``: NHWC decoder features [N, H/2, W/2, 16]."""
        if not input.is_cuda:
            raise RuntimeError("mdil_ss_amd.Net runs on MI355X only (input must be a cuda tensor); "
                               "there is no CPU fallback in the product path")
        x = ops.to_nhwc(input)
        masks = self.draw_masks(x.shape[0], x.device) if self.training else None
        for f in self.plan(task, masks, head=False):
            x = f(x)
        return x

    def forward(self, input, task):
        global current_task
        current_task = task
        if not input.is_cuda:
            raise RuntimeError("mdil_ss_amd.Net runs on MI355X only (input must be a cuda tensor); "
                               "there is no CPU fallback in the product path")
        train = self.training
        x = ops.to_nhwc(input)
        masks = self.draw_masks(x.shape[0], x.device) if train else None
        links = ops.boundaries(len(self.encoder.layers))
        y = self.encoder.run(x, task, train, masks, links)
        y = self.decoder[task].run(y, train, links[-1])             # [N, H, W, nc]
        return y.permute(0, 3, 1, 2)                                # NCHW view, channels-last storage

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        ops.refresh_packs()
        return out
