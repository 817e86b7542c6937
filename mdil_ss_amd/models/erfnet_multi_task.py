"""ERFNet multi-task joint model (one shared encoder, one decoder head per domain) on the HIP
path -- drop-in for ``models/erfnet_multi_task.py`` of the reference (lines 14-160): same
``Net(num_classes, nb_tasks, cur_task)`` / ``forward(input, task)`` surface, same parameter and
buffer names, shapes, order and seeded initial values.  Like the RAP model, sub-modules are
parameter containers; the math runs through ``mdil_ss_amd.ops`` (no adapters, shared BatchNorm,
encoder blocks with Dropout2d)."""
import torch
import torch.nn as nn

from .. import ops
from .erfnet_RA_parallel import Decoder, _Holder, _bn, _bn_bufs      # decoder is identical (:115-146)

current_task = 0


class DownsamplerBlock(_Holder):
    # reference :14-25
    def __init__(self, ninput, noutput):
        super().__init__()
        self.conv = nn.Conv2d(ninput, noutput - ninput, (3, 3), stride=2, padding=1, bias=True)
        self.bn = _bn(noutput)

    def run(self, x, task, train, drop=None, links=(None, None)):
        return ops.DownFn.apply(x, self.conv.weight, self.conv.bias, self.bn.weight, self.bn.bias,
                                *_bn_bufs(self.bn), train, links[1])


class non_bottleneck_1d(_Holder):
    # reference :28-64 (encoder use: dilation + Dropout2d)
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
        self.chann = chann

    def run(self, x, task, train, drop=None, links=(None, None)):
        if not (train and self.dropout.p != 0):
            drop = None
        bufs = _bn_bufs(self.bn1) + _bn_bufs(self.bn2)
        return ops.NbFn.apply(
            x, self.conv3x1_1.weight, self.conv3x1_1.bias, self.conv1x3_1.weight,
            self.conv1x3_1.bias, None, None, self.bn1.weight, self.bn1.bias,
            self.conv3x1_2.weight, self.conv3x1_2.bias, self.conv1x3_2.weight,
            self.conv1x3_2.bias, None, None, self.bn2.weight, self.bn2.bias, bufs, drop,
            self.dilated, train, links[0], links[1])


class Encoder(_Holder):
    # reference :73-103
    def __init__(self):
        super().__init__()
        self.initial_block = DownsamplerBlock(3, 16)
        self.layers = nn.ModuleList()
        self.layers.append(DownsamplerBlock(16, 64))
        for _ in range(5):
            self.layers.append(non_bottleneck_1d(64, 0.03, 1))
        self.layers.append(DownsamplerBlock(64, 128))
        for _ in range(2):
            for d in (2, 4, 8, 16):
                self.layers.append(non_bottleneck_1d(128, 0.3, d))

    def dropout_blocks(self):
        return [m for m in self.layers if isinstance(m, non_bottleneck_1d)]


class Net(nn.Module):
    # reference :148-160
    def __init__(self, num_classes=[20], nb_tasks=1, cur_task=0):
        super().__init__()
        self.encoder = Encoder()
        self.decoder = nn.ModuleList([Decoder(num_classes[i]) for i in range(nb_tasks)])
        self.mask_provider = None
        self.mask_generator = None

    def draw_masks(self, n, device):
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
        u = torch.rand(keep.numel(), device=device, generator=getattr(self, "mask_generator", None))
        flat = ops.dropout_factors(u, keep, inv) if u.is_cuda else (u < keep).to(torch.float32).mul_(inv)
        return [m.view(n, c) for m, c in zip(flat.split(sizes), chans)]

    def plan(self, task, masks=None, head=True):
        train = self.training
        enc, dec = self.encoder, self.decoder[task]
        B = ops.boundaries(len(enc.layers) + len(dec.layers))      # explicit fusion chain (ops.Boundary)
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
        if head:        # head=False: the decoder's features for the fused head + loss (ops.head_ce)
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
        y = ops.to_nhwc(input)
        masks = self.draw_masks(y.shape[0], y.device) if self.training else None
        for f in self.plan(task, masks, head=False):
            y = f(y)
        return y

    def forward(self, input, task):
        global current_task
        current_task = task
        if not input.is_cuda:
            raise RuntimeError("mdil_ss_amd.Net runs on MI355X only (input must be a cuda tensor); "
                               "there is no CPU fallback in the product path")
        y = ops.to_nhwc(input)
        masks = self.draw_masks(y.shape[0], y.device) if self.training else None
        for f in self.plan(task, masks):
            y = f(y)
        return y.permute(0, 3, 1, 2)

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        ops.refresh_packs()
        return out
