"""Datasets for the trainers.  The reference reads Cityscapes / BDD100k / IDD from hard-coded
site paths (train_new_task_step2.py:140-142, dataset.py); none of them is available offline, so
the MI355X build ships a seeded procedural dataset with the same sample contract
(image f32[3,H,W] in [0,1], label i64[1,H,W] with the ignore class = n_classes-1) used for
throughput and mIoU-parity runs.  Real-dataset loaders are a later row (8f-3)."""
import torch
from torch.utils.data import Dataset


class ProceduralSeg(Dataset):
    """Random axis-aligned class rectangles over a class-dependent colour + noise."""

    def __init__(self, n_items, height, width, n_classes=20, seed=0, n_rects=12, noise=0.08,
                 domain=0, classes_used=None):
        """``seed`` selects the images of a split, ``domain`` the class->colour palette (train and
        validation splits of one domain share it; different domains differ)."""
        self.n, self.h, self.w, self.c = n_items, height, width, n_classes
        self.seed, self.n_rects, self.noise = seed, n_rects, noise
        self.used = classes_used or n_classes          # labels are drawn from [0, used)
        g = torch.Generator().manual_seed(domain * 7919 + 17)
        self.palette = torch.rand(n_classes, 3, generator=g)

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000003 + i)
        lab = torch.full((self.h, self.w), int(torch.randint(0, min(self.used, self.c - 1), (1,), generator=g)),
                         dtype=torch.int64)
        for _ in range(self.n_rects):
            c = int(torch.randint(0, self.used, (1,), generator=g))
            y0 = int(torch.randint(0, self.h, (1,), generator=g))
            x0 = int(torch.randint(0, self.w, (1,), generator=g))
            hh = int(torch.randint(self.h // 8, self.h // 2 + 1, (1,), generator=g))
            ww = int(torch.randint(self.w // 8, self.w // 2 + 1, (1,), generator=g))
            lab[y0:y0 + hh, x0:x0 + ww] = c
        img = self.palette[lab].permute(2, 0, 1).contiguous()
        img = (img + self.noise * torch.randn(3, self.h, self.w, generator=g)).clamp_(0, 1)
        return img, lab.unsqueeze(0)
