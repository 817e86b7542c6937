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
                 domain=0, classes_used=None, palette="random"):
        """``seed`` selects the images of a split, ``domain`` the class->colour palette (train and
        validation splits of one domain share it; different domains differ).  ``palette``:
        "random" colours, or "grid" = well separated colours (a per-domain permutation of the
        3x3x3 grid {0.1, 0.5, 0.9}^3; needs n_classes <= 27)."""
        self.n, self.h, self.w, self.c = n_items, height, width, n_classes
        self.seed, self.n_rects, self.noise = seed, n_rects, noise
        self.used = classes_used or n_classes          # labels are drawn from [0, used)
        g = torch.Generator().manual_seed(domain * 7919 + 17)
        if palette == "grid":
            assert n_classes <= 27
            pts = torch.tensor([[r, gg, b] for r in (0.1, 0.5, 0.9) for gg in (0.1, 0.5, 0.9)
                                for b in (0.1, 0.5, 0.9)])
            self.palette = pts[torch.randperm(27, generator=g)[:n_classes]]
        else:
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


# ----------------------------------------------------------------------------------------------
# Real datasets: the reference's ``dataset.py`` surface (cityscapes :77-115, IDD :120-155,
# IDD_union :160-213, BDD100k :218-256).  File discovery and pairing rules are the reference's
# (sorted image list zipped with the sorted label list); decode is PIL on the host.  What the
# ``co_transform`` returns is up to the transform: the product's ``MyCoTransform`` yields the
# resized BYTES plus the three augmentation draws and leaves flip / shift / float conversion /
# relabel to ``ops.augment_batch`` on the GPU.
# ----------------------------------------------------------------------------------------------
import os  # noqa: E402

import numpy as np  # noqa: E402
from PIL import Image  # noqa: E402

EXTENSIONS = [".jpg", ".png"]


def load_image(file):
    return Image.open(file)


def is_image(filename):
    return any(filename.endswith(ext) for ext in EXTENSIONS)


def is_label_city(filename):
    return filename.endswith("_labelTrainIds.png")


def is_label_IDD(filename):
    return filename.endswith("_labellevel3Ids.png")


def is_label_BDD(filename):
    return filename.endswith("_train_id.png")


def _walk(root, keep):
    out = [os.path.join(dp, f) for dp, _, fn in os.walk(os.path.expanduser(root)) for f in fn if keep(f)]
    out.sort()
    return out


class _PairedSeg(torch.utils.data.Dataset):
    """image list + label list (both sorted), RGB / 'P' decode, optional label remap, co-transform."""

    label_map = None

    def __init__(self, co_transform=None):
        self.co_transform = co_transform

    def __getitem__(self, index):
        with open(self.filenames[index], "rb") as f:
            image = load_image(f).convert("RGB")
        with open(self.filenamesGt[index], "rb") as f:
            label = load_image(f).convert("P")
        if self.label_map is not None:
            label = Image.fromarray(np.uint8(self.label_map[np.array(label)]))
        if self.co_transform is not None:
            return self.co_transform(image, label)
        return image, label

    def __len__(self):
        return len(self.filenames)


class cityscapes(_PairedSeg):
    def __init__(self, root, co_transform=None, subset="train"):
        super().__init__(co_transform)
        self.images_root = os.path.join(root, "leftImg8bit/") + subset
        self.labels_root = os.path.join(root, "gtFine/") + subset
        print(self.images_root)
        self.filenames = _walk(self.images_root, is_image)
        self.filenamesGt = _walk(self.labels_root, is_label_city)


class IDD(_PairedSeg):
    def __init__(self, root, co_transform=None, subset="train"):
        super().__init__(co_transform)
        self.images_root = os.path.join(root, "leftImg8bit/") + subset
        self.labels_root = os.path.join(root, "gtFine/") + subset
        print(self.images_root)
        self.filenames = _walk(self.images_root, is_image)
        self.filenamesGt = _walk(self.labels_root, is_label_IDD)


class IDD_union(IDD):
    """IDD level-3 ids remapped into the Cityscapes-union label space (dataset.py:171-173,199-203)."""
    MAP_dict = {0: 0, 1: 19, 2: 1, 3: 20, 4: 11, 5: 12, 6: 17, 7: 18, 8: 21, 9: 13, 10: 14, 11: 15,
                12: 22, 13: 23, 14: 3, 15: 4, 16: 24, 17: 25, 18: 7, 19: 6, 20: 5, 21: 26, 22: 2,
                23: 27, 24: 8, 25: 10, 255: 255}

    def __init__(self, root, co_transform=None, subset="train"):
        super().__init__(root, co_transform, subset)
        k = np.array(list(self.MAP_dict.keys()))
        v = np.array(list(self.MAP_dict.values()))
        self.label_map = np.zeros(k.max() + 1, dtype=v.dtype)
        self.label_map[k] = v


class BDD100k(_PairedSeg):
    def __init__(self, root, co_transform=None, subset="train"):
        super().__init__(co_transform)
        self.images_root = os.path.join(root, "images/") + subset
        self.labels_root = os.path.join(root, "labels/") + subset
        print(self.images_root)
        self.filenames = sorted(os.path.join(self.images_root, f)
                                for f in os.listdir(self.images_root) if is_image(f))
        self.filenamesGt = sorted(os.path.join(self.labels_root, f)
                                  for f in os.listdir(self.labels_root) if is_label_BDD(f))


class MyCoTransform(object):
    """Host half of the reference's ``MyCoTransform`` (train_new_task_step2.py:48-81): PIL resize
    (bilinear image / nearest label, :56-57) and the three random draws in the reference's order
    (:62-69).  Returns ``(uint8 [H,W,3], uint8 [H,W], int32 [3] = hflip, transX, transY)``; the
    flip, shift, fill, ``ToTensor``, ``ToLabel`` and ``Relabel`` run on the GPU for the whole batch
    (``ops.augment_batch``), so the host ships 4 bytes per pixel instead of 20."""

    def __init__(self, augment=True, height=512, width=1024):
        self.augment, self.height, self.width = augment, height, width

    def __call__(self, input, target):
        import random
        input = input.resize((self.width, self.height), Image.BILINEAR)
        target = target.resize((self.width, self.height), Image.NEAREST)
        flip = tx = ty = 0
        if self.augment:
            flip = int(random.random() < 0.5)
            tx = random.randint(-2, 2)
            ty = random.randint(-2, 2)
        return (torch.from_numpy(np.array(input, dtype=np.uint8)),
                torch.from_numpy(np.array(target, dtype=np.uint8)),
                torch.tensor([flip, tx, ty], dtype=torch.int32))


def to_device_batch(batch, device, num_classes):
    """A collated loader batch -> (images f32 [N,3,H,W] (NHWC storage), labels i64 [N,1,H,W]) on
    ``device``.  3-tuples come from ``MyCoTransform`` (bytes + draws: finished on the GPU);
    2-tuples (procedural dataset) are already float / long."""
    from . import ops
    if len(batch) == 3:
        img, lab, params = batch
        return ops.augment_batch(img.to(device, non_blocking=True), lab.to(device, non_blocking=True),
                                 params.to(device, non_blocking=True), num_classes)
    images, labels = batch
    return images.to(device, non_blocking=True), labels.to(device, non_blocking=True)


# dataset roots hard-coded by the reference trainers (train_new_task_step2.py:140-142); override
# with --cs-datadir / --bdd-datadir / --idd-datadir or MDIL_{CS,BDD,IDD}_DATADIR
DATA_ROOTS = {"cityscapes": "/ssd_scratch/cvit/prachigarg/cityscapes/",
              "BDD": "/ssd_scratch/cvit/prachigarg/bdd100k/seg/",
              "IDD": "/ssd_scratch/cvit/prachigarg/IDD_Segmentation/"}
_ALIASES = {"CS": "cityscapes", "cityscapes": "cityscapes", "BDD": "BDD", "IDD": "IDD"}
_CLASSES = {"cityscapes": cityscapes, "BDD": BDD100k, "IDD": IDD}


def add_datadir_flags(parser):
    parser.add_argument("--cs-datadir", default=os.getenv("MDIL_CS_DATADIR", DATA_ROOTS["cityscapes"]))
    parser.add_argument("--bdd-datadir", default=os.getenv("MDIL_BDD_DATADIR", DATA_ROOTS["BDD"]))
    parser.add_argument("--idd-datadir", default=os.getenv("MDIL_IDD_DATADIR", DATA_ROOTS["IDD"]))


def open_dataset(name, subset, args, augment):
    """The reference's dataset object for ``name`` ('cityscapes'|'CS'|'BDD'|'IDD') with the
    product's host-side co-transform (bytes + draws; the rest runs in ops.augment_batch)."""
    key = _ALIASES[name]
    root = {"cityscapes": args.cs_datadir, "BDD": args.bdd_datadir, "IDD": args.idd_datadir}[key]
    if not os.path.isdir(root):
        flag = {"cityscapes": "--cs-datadir", "BDD": "--bdd-datadir", "IDD": "--idd-datadir"}[key]
        raise RuntimeError(f"dataset root for {name} not found: {root} (set {flag} or run with "
                           "--synthetic N)")
    return _CLASSES[key](root, MyCoTransform(augment, args.height, args.width), subset)
