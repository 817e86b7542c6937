"""Single-task ERFNet and the fine-tuning / feature-extraction baselines' multi-head variants on
the HIP path -- drop-ins for ``models/erfnet.py``, ``models/erfnet_ftp1.py`` and
``models/erfnet_ftp2.py`` (``Net``) of the reference (``Net``): same constructors, forward flags, parameter
/ buffer names, order and seeded initial values.  All three are the plain encoder of
``erfnet_multi_task`` with differently named decoder heads."""
import torch
import torch.nn as nn

from .. import ops
from .erfnet_multi_task import Encoder, DownsamplerBlock
from .erfnet_RA_parallel import Decoder


class _PlainBase(nn.Module):
    mask_provider = None
    mask_generator = None

    def draw_masks(self, n, device):
        if self.mask_provider is not None:
            return [m.to(device=device, dtype=torch.float32).reshape(n, -1).contiguous()
                    for m in self.mask_provider(n)]
        masks = []
        for blk in self.encoder.dropout_blocks():
            p = blk.dropout.p
            m = torch.empty(n, blk.chann, device=device, dtype=torch.float32)
            masks.append(m.bernoulli_(1 - p, generator=self.mask_generator).div_(1 - p))
        return masks

    def _run(self, input, dec):
        if not input.is_cuda:
            raise RuntimeError("mdil_ss_amd models run on MI355X only (input must be a cuda tensor); "
                               "there is no CPU fallback in the product path")
        train = self.training
        y = ops.to_nhwc(input)
        masks = self.draw_masks(y.shape[0], y.device) if train else None
        enc = self.encoder
        B = ops.boundaries(len(enc.layers))                         # explicit fusion chain (ops.Boundary)
        y = enc.initial_block.run(y, 0, train, links=(None, B[0]))
        k = 0
        for i, layer in enumerate(enc.layers):
            if isinstance(layer, DownsamplerBlock):
                y = layer.run(y, 0, train, links=(B[i], B[i + 1]))
            else:
                y = layer.run(y, 0, train, None if masks is None else masks[k], links=(B[i], B[i + 1]))
                k += 1
        return dec.run(y, train, B[-1]).permute(0, 3, 1, 2)

    def load_state_dict(self, state_dict, strict=True, **kw):
        out = super().load_state_dict(state_dict, strict=strict, **kw)
        ops.refresh_packs()
        return out


class Net(_PlainBase):
    # models/erfnet.py:140-149 (the encoder takes num_classes there but does not use it); the
    # evaluation notebook imports the same class under the name ``ERFNet`` (cell 2)
    def __init__(self, num_classes):
        super().__init__()
        self.encoder = Encoder()
        self.decoder = Decoder(num_classes)

    def forward(self, input):
        return self._run(input, self.decoder)


ERFNet = Net


class NetFT1(_PlainBase):
    # models/erfnet_ftp1.py:134-151
    def __init__(self, num_classes_old=20, num_classes_new=20):
        super().__init__()
        self.encoder = Encoder()
        self.decoder_old = Decoder(num_classes_old)
        self.decoder_new = Decoder(num_classes_new)

    def forward(self, input, decoder_old=False, decoder_new=True, finetune=False):
        if decoder_old:
            return self._run(input, self.decoder_old)
        if decoder_new:
            return self._run(input, self.decoder_new)
        raise RuntimeError("select decoder_old or decoder_new")   # the reference returns encoder features


class NetFT2(_PlainBase):
    # models/erfnet_ftp2.py:134-152
    def __init__(self, num_classes_old1=20, num_classes_old2=20, num_classes_new=27):
        super().__init__()
        self.encoder = Encoder()
        self.decoder_old1 = Decoder(num_classes_old1)
        self.decoder_old2 = Decoder(num_classes_old2)
        self.decoder_new = Decoder(num_classes_new)

    def forward(self, input, decoder_old1=False, decoder_old2=False, decoder_new=True):
        if decoder_old1:
            return self._run(input, self.decoder_old1)
        if decoder_old2:
            return self._run(input, self.decoder_old2)
        if decoder_new:
            return self._run(input, self.decoder_new)
        raise RuntimeError("select a decoder")
