"""mIoU metric with the reference's ``iouEval`` surface (iouEval.py:8-77), computed by a fused
argmax + confusion-count HIP kernel instead of materialising N x C x H x W one-hot tensors.

``addBatch(x, y)``: ``x`` = logits ``[N,C,H,W]`` (argmax is fused) or class indices ``[N,1,H,W]``;
``y`` = targets ``[N,1,H,W]`` (int64).  Pixels whose target is ``ignoreIndex`` are dropped and the
ignore class is excluded from the mean, exactly like the reference.  Counts are exact integers
(int64 on device), converted to float64 in ``getIoU``.
"""
import torch

from . import ops


class iouEval:
    def __init__(self, nClasses, ignoreIndex=19):
        self.nClasses = nClasses
        self.ignoreIndex = ignoreIndex if nClasses > ignoreIndex else -1
        self.reset()

    def reset(self):
        self.counts = None
        k = self.nClasses if self.ignoreIndex == -1 else self.nClasses - 1
        self.tp = torch.zeros(k).double()
        self.fp = torch.zeros(k).double()
        self.fn = torch.zeros(k).double()

    def addBatch(self, x, y):
        if not x.is_cuda:
            raise RuntimeError("iouEval (mdil_ss_amd) accumulates on the GPU; pass device tensors")
        if self.counts is None:
            self.counts = torch.zeros(3, self.nClasses, dtype=torch.int64, device=x.device)
        if x.size(1) == 1:                     # class indices -> one-hot logits (rare path)
            logits = torch.zeros(x.size(0), self.nClasses, x.size(2), x.size(3), device=x.device)
            logits = logits.contiguous(memory_format=torch.channels_last)
            logits.scatter_(1, x, 1.0)
            x = logits
        ops.argmax_confusion(x, y[:, 0], self.ignoreIndex, self.counts)

    def _sync(self):
        if self.counts is not None:
            k = self.tp.numel()
            c = self.counts.cpu().double()
            ops.check_labels()       # a target outside [0, nClasses) raises (scatter_ would, iouEval.py:33)
            self.tp, self.fp, self.fn = c[0, :k].clone(), c[1, :k].clone(), c[2, :k].clone()

    def getIoU(self):
        self._sync()
        iou = self.tp / (self.tp + self.fp + self.fn + 1e-15)
        return torch.mean(iou), iou
