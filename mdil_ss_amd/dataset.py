"""Datasets for the trainers.  The reference reads Cityscapes / BDD100k / IDD from hard-coded
site paths (train_new_task_step2.py:140-142, dataset.py); none of them is available offline, so
the MI355X build ships a seeded procedural dataset with the same sample contract
(image f32[3,H,W] in [0,1], label i64[1,H,W] with the ignore class = n_classes-1) used for
throughput and mIoU-parity runs.  The real-dataset loaders (8f-3) follow below,
with a resize cache (host, memory-mapped) and an HBM-resident form of it."""
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

    def resize_bytes(self, input, target):
        """The deterministic part (:56-57): the bytes PIL's resize produces, as uint8 arrays."""
        input = input.resize((self.width, self.height), Image.BILINEAR)
        target = target.resize((self.width, self.height), Image.NEAREST)
        return np.array(input, dtype=np.uint8), np.array(target, dtype=np.uint8)

    def draw(self):
        """The random part (:62-69), in the reference's order: flip, transX, transY."""
        import random
        flip = tx = ty = 0
        if self.augment:
            flip = int(random.random() < 0.5)
            tx = random.randint(-2, 2)
            ty = random.randint(-2, 2)
        return flip, tx, ty

    def __call__(self, input, target):
        img, lab = self.resize_bytes(input, target)
        return torch.from_numpy(img), torch.from_numpy(lab), torch.tensor(self.draw(), dtype=torch.int32)


# ----------------------------------------------------------------------------------------------
# Resize cache.  PNG/JPEG decode + PIL resize is deterministic per image and costs
# 10-30 ms of one core (29 img/s per worker on 2048x1024 PNGs: 233 img/s on the GPU box's 16 CPUs,
# below what ONE MI355X consumes, profiles/r02_loader_throughput.txt).  ``ResizedCache`` keeps the
# post-Resize uint8 bytes (what ``MyCoTransform.resize_bytes`` returns: 2 MiB per 1024x512 sample)
# in memory-mapped files, filled on first touch by whichever loader worker meets a sample first;
# every later epoch (and every later run on the same directory) reads bytes instead of decoding.
# ``DeviceResizedCache`` goes one step further, MI355X-first: a whole split is 6-15 GB of uint8
# (Cityscapes train 2,975 x 2 MiB = 6.2 GB, BDD 7,000 x 2 MiB = 14.7 GB) -- it lives in the 288 GB
# of HBM, an epoch is a permutation + three draws per sample on the host, a gather + the augment
# kernel on the device, and the host ships 12 bytes per sample.
# ----------------------------------------------------------------------------------------------
class ResizedCache:
    """uint8 [n,H,W,3] images + uint8 [n,H,W] labels + uint8 [n] filled flags, memory-mapped under
    ``directory``.  Keyed by ``identity`` (the split's files with size and mtime, dataset class /
    label remap, resize filters) and the target size: anything else in the directory is ignored."""

    def __init__(self, directory, name, n, identity, height, width):
        """``n`` samples; ``identity``: what the bytes are a function of besides the size (the
        split's image and label file lists)."""
        import hashlib
        import json
        os.makedirs(directory, exist_ok=True)
        key = hashlib.sha1(json.dumps([n, identity, height, width]).encode()).hexdigest()[:16]
        base = os.path.join(directory, f"{name}_{width}x{height}_{key}")
        self.paths = [base + s for s in (".img.u8", ".lab.u8", ".ok.u8")]
        shapes = [(n, height, width, 3), (n, height, width), (n,)]
        self.n, self.height, self.width = n, height, width
        arrs = []
        import time
        for pth, shp in zip(self.paths, shapes):
            size = int(np.prod(shp))
            try:
                # exactly ONE process creates a file (the ranks of a data-parallel run open the same
                # cache at the same time): everybody must map the same inode, or a filled flag could
                # be seen by a process whose image file never received the bytes
                fd = os.open(pth, os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o644)
                os.ftruncate(fd, size)                # sparse: pages appear as samples are written
                os.close(fd)
            except FileExistsError:
                t0 = time.time()
                while os.path.getsize(pth) != size:   # the creator is between open and ftruncate
                    if time.time() - t0 > 30:
                        raise RuntimeError(f"resize cache file {pth} has the wrong size "
                                           f"({os.path.getsize(pth)} != {size}); remove it")
                    time.sleep(0.05)
            arrs.append(np.memmap(pth, dtype=np.uint8, mode="r+", shape=shp))
        self.img, self.lab, self.ok = arrs

    def filled(self):
        return int(np.count_nonzero(self.ok))

    def get(self, i):
        return (self.img[i], self.lab[i]) if self.ok[i] else None

    def put(self, i, img, lab):
        self.img[i] = img
        self.lab[i] = lab
        self.ok[i] = 1                                # after the data (x86 stores stay in order)


class CachedSeg(torch.utils.data.Dataset):
    """``base`` (a dataset class above, built with ``co_transform=None``) behind a ResizedCache:
    same sample contract as ``base`` with ``MyCoTransform`` -- (uint8 image bytes, uint8 label
    bytes, int32 draws) -- and bit-identical bytes (tests/test_input_pipeline.py)."""

    def __init__(self, base, cache, co_transform):
        self.base, self.cache, self.co = base, cache, co_transform
        assert base.co_transform is None and len(base) == cache.n

    def __len__(self):
        return self.cache.n

    def bytes(self, i):
        hit = self.cache.get(i)
        if hit is None:
            img, lab = self.co.resize_bytes(*self.base[i])
            self.cache.put(i, img, lab)
            return img, lab
        return hit

    def __getitem__(self, i):
        img, lab = self.bytes(i)
        return (torch.from_numpy(np.ascontiguousarray(img)), torch.from_numpy(np.ascontiguousarray(lab)),
                torch.tensor(self.co.draw(), dtype=torch.int32))


class _Indexed(torch.utils.data.Dataset):
    """(index, image bytes, label bytes) without draws: what the device cache is filled from."""

    def __init__(self, cached):
        self.c = cached

    def __len__(self):
        return len(self.c)

    def __getitem__(self, i):
        img, lab = self.c.bytes(i)
        return i, torch.from_numpy(np.ascontiguousarray(img)), torch.from_numpy(np.ascontiguousarray(lab))


class DeviceResizedCache:
    """A split's post-Resize bytes resident in HBM + a loader-shaped iterator over it.
    ``loader(batch_size, ...)`` yields ``(images f32 [B,3,H,W] (NHWC storage), labels i64 [B,1,H,W])``
    on the device, already augmented -- the same values ``to_device_batch`` produces from a
    ``DataLoader`` over ``CachedSeg`` / ``MyCoTransform`` given the same indices and draws."""

    def __init__(self, cached, device, num_workers=0, fill_batch=16):
        self.cached, self.device = cached, device
        n, H, W = len(cached), cached.cache.height, cached.cache.width
        self.img = torch.empty(n, H, W, 3, dtype=torch.uint8, device=device)
        self.lab = torch.empty(n, H, W, dtype=torch.uint8, device=device)
        fill = torch.utils.data.DataLoader(_Indexed(cached), batch_size=fill_batch, shuffle=False,
                                           num_workers=num_workers)
        for idx, img, lab in fill:                    # first touch decodes; later runs read bytes
            idx = idx.to(device)
            self.img[idx] = img.to(device, non_blocking=True)
            self.lab[idx] = lab.to(device, non_blocking=True)

    def __len__(self):
        return self.img.shape[0]

    def loader(self, batch_size, num_classes, shuffle, drop_last=False, rank=0, world=1, seed=None):
        """``seed``: of the epoch permutations; default = torch's initial seed (``torch.manual_seed``),
        so runs differ unless seeded, like ``DataLoader(shuffle=True)`` -- and all ranks of a run,
        seeded alike by the trainer, draw the same permutation and take their own slice of it."""
        if seed is None:
            seed = torch.initial_seed() % (1 << 31)
        return _DeviceLoader(self, batch_size, num_classes, shuffle, drop_last, rank, world, seed)


class _DeviceLoader:
    """Iterable with the bits of the DataLoader surface the trainers use (len, .sampler.set_epoch)."""

    def __init__(self, cache, batch_size, num_classes, shuffle, drop_last, rank, world, seed):
        self.c, self.bs, self.nc = cache, batch_size, num_classes
        self.shuffle, self.drop_last, self.rank, self.world, self.seed = shuffle, drop_last, rank, world, seed
        self.epoch = 0
        self.sampler = self                           # trainers call loader.sampler.set_epoch(e)

    def set_epoch(self, epoch):
        self.epoch = epoch

    def _indices(self):
        n = len(self.c)
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            idx = torch.randperm(n, generator=g)
        else:
            idx = torch.arange(n)
        if self.world > 1:
            if self.shuffle:                          # DistributedSampler: pad to a multiple, stride
                pad = (-n) % self.world
                idx = torch.cat([idx, idx[:pad]])
            idx = idx[self.rank::self.world]
        return idx

    def __len__(self):
        n = len(self._indices())
        return n // self.bs if self.drop_last else (n + self.bs - 1) // self.bs

    def __iter__(self):
        from . import ops
        idx = self._indices()
        co = self.c.cached.co
        for b in range(len(self)):
            sel = idx[b * self.bs:(b + 1) * self.bs]
            prm = torch.tensor([co.draw() for _ in range(len(sel))], dtype=torch.int32)
            sel_d = sel.to(self.c.device, non_blocking=True)
            yield ops.augment_batch(self.c.img.index_select(0, sel_d), self.c.lab.index_select(0, sel_d),
                                    prm.to(self.c.device, non_blocking=True), self.nc)


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
    add_cache_flags(parser)
    parser.add_argument("--cs-datadir", default=os.getenv("MDIL_CS_DATADIR", DATA_ROOTS["cityscapes"]))
    parser.add_argument("--bdd-datadir", default=os.getenv("MDIL_BDD_DATADIR", DATA_ROOTS["BDD"]))
    parser.add_argument("--idd-datadir", default=os.getenv("MDIL_IDD_DATADIR", DATA_ROOTS["IDD"]))


def open_dataset(name, subset, args, augment):
    """The reference's dataset object for ``name`` ('cityscapes'|'CS'|'BDD'|'IDD') with the
    product's host-side co-transform (bytes + draws; the rest runs in ops.augment_batch).
    With ``--cache-resized DIR`` the split sits behind a ResizedCache (decode + resize once)."""
    key = _ALIASES[name]
    root = {"cityscapes": args.cs_datadir, "BDD": args.bdd_datadir, "IDD": args.idd_datadir}[key]
    if not os.path.isdir(root):
        flag = {"cityscapes": "--cs-datadir", "BDD": "--bdd-datadir", "IDD": "--idd-datadir"}[key]
        raise RuntimeError(f"dataset root for {name} not found: {root} (set {flag} or run with "
                           "--synthetic N)")
    co = MyCoTransform(augment, args.height, args.width)
    cache_dir = getattr(args, "cache_resized", None)
    if not cache_dir:
        return _CLASSES[key](root, co, subset)
    base = _CLASSES[key](root, None, subset)

    def stamp(paths):                 # re-exported / relabelled files under the same names: another cache
        out = []
        for p in paths:
            st = os.stat(p)
            out.append([p, st.st_size, st.st_mtime_ns])
        return out
    # what the cached bytes are a function of: the files (path, size, mtime), the dataset class (its
    # label remap, e.g. IDD_union) and the resize filters MyCoTransform applies
    identity = [stamp(base.filenames), stamp(base.filenamesGt), type(base).__name__,
                None if getattr(base, "label_map", None) is None else np.asarray(base.label_map).tolist(),
                "resize: image BILINEAR, label NEAREST (PIL)"]
    cache = ResizedCache(cache_dir, f"{key}_{subset}", len(base), identity, args.height, args.width)
    return CachedSeg(base, cache, co)


def add_cache_flags(parser):
    parser.add_argument("--cache-resized", default=os.getenv("MDIL_CACHE_RESIZED"),
                        help="directory for the post-Resize uint8 bytes of every split (filled on first "
                             "touch, reused by later epochs and runs): decode + PIL resize happen once")
    parser.add_argument("--cache-device", action="store_true",
                        help="with --cache-resized: keep the splits' bytes resident in HBM (2 MiB per "
                             "1024x512 sample) and run the whole input pipeline on the GPU from the "
                             "second touch on")
