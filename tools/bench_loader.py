#!/usr/bin/env python3
"""Host input-pipeline throughput: how many images/sec the product's loader path
(dataset.py classes + host half of MyCoTransform: PIL decode, PIL resize to 1024x512, three
random draws; the flip / shift / float conversion run on the GPU in ops.augment_batch) delivers
per worker, against what one MI355X consumes (~200 img/s in step 2).

    python tools/bench_loader.py [--workers 1 2 4 8] [--images 48] [--cached] [--device]

``--cached``: the same passes through the resize cache (``--cache-resized``: decode + PIL resize on
the first touch only, memory-mapped bytes afterwards); ``--device``: the HBM-resident form of it
(``--cache-device``: gather + augment kernel on the GPU, 12 bytes per sample from the host).

Writes a throw-away Cityscapes-layout tree of synthetic 2048x1024 PNGs (street-scene-like low
frequency content + noise, so PNG decode costs what it costs on photographs, ~2 MB per file) and
a BDD-layout tree of 1280x720 JPEGs under a temp dir, then times full passes through
torch.utils.data.DataLoader (batch 6).  No GPU needed."""
import argparse
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mdil_ss_amd  # noqa: E402,F401
from mdil_ss_amd import dataset as D  # noqa: E402


def _photo(h, w, rng):
    base = rng.random((h // 32 + 1, w // 32 + 1, 3)).astype(np.float32)
    img = np.asarray(Image.fromarray((base * 255).astype(np.uint8)).resize((w, h), Image.BILINEAR),
                     dtype=np.float32)
    img += rng.normal(0, 6.0, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def _labels(h, w, rng):
    lab = rng.integers(0, 19, (h // 64 + 1, w // 64 + 1)).astype(np.uint8)
    return np.asarray(Image.fromarray(lab).resize((w, h), Image.NEAREST))


def make_trees(root, n):
    rng = np.random.default_rng(0)
    cs_i = os.path.join(root, "cs", "leftImg8bit", "train", "city")
    cs_l = os.path.join(root, "cs", "gtFine", "train", "city")
    bd_i = os.path.join(root, "bdd", "images", "train")
    bd_l = os.path.join(root, "bdd", "labels", "train")
    for d in (cs_i, cs_l, bd_i, bd_l):
        os.makedirs(d)
    for i in range(n):
        Image.fromarray(_photo(1024, 2048, rng)).save(os.path.join(cs_i, f"city_{i:06d}_leftImg8bit.png"))
        Image.fromarray(_labels(1024, 2048, rng)).save(os.path.join(cs_l, f"city_{i:06d}_gtFine_labelTrainIds.png"))
        Image.fromarray(_photo(720, 1280, rng)).save(os.path.join(bd_i, f"{i:06d}.jpg"), quality=90)
        Image.fromarray(_labels(720, 1280, rng)).save(os.path.join(bd_l, f"{i:06d}_train_id.png"))
    return os.path.join(root, "cs"), os.path.join(root, "bdd")


def rate(ds, workers, passes=2):
    loader = torch.utils.data.DataLoader(ds, batch_size=6, shuffle=True, num_workers=workers,
                                         drop_last=False, persistent_workers=workers > 0)
    n = 0
    for _ in loader:           # warm-up pass (worker start-up, page cache)
        pass
    t0 = time.perf_counter()
    for _ in range(passes):
        for batch in loader:
            n += batch[0].shape[0]
    return n / (time.perf_counter() - t0)


def rate_device(loader, passes=4):
    n = 0
    for _ in loader:
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for e in range(passes):
        loader.set_epoch(e)
        for x, y in loader:
            n += x.shape[0]
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--images", type=int, default=48)
    ap.add_argument("--cached", action="store_true")
    ap.add_argument("--device", action="store_true")
    a = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="mdil_loader_")
    try:
        cs, bdd = make_trees(tmp, a.images)
        sz = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(tmp) for f in fs)
        print(f"{a.images} images per dataset, {sz / 2**20:.0f} MiB on disk (page-cache resident); "
              f"{os.cpu_count()} logical CPUs visible")
        tf = D.MyCoTransform(True, 512, 1024)
        for name, ds in (("cityscapes 2048x1024 png", D.cityscapes(cs, tf, "train")),
                         ("BDD100k   1280x720  jpg", D.BDD100k(bdd, tf, "train"))):
            for w in a.workers:
                r = rate(ds, w)
                print(f"{name}  workers {w:2d}: {r:7.1f} img/s  ({r / max(w, 1):6.1f} per worker)  -> "
                      f"{200.0 / (r / max(w, 1)):5.1f} workers feed one GPU at 200 img/s", flush=True)
        if a.cached or a.device:
            cdir = os.path.join(tmp, "cache")
            for name, cls, root in (("cityscapes 2048x1024 png", D.cityscapes, cs), ("BDD100k   1280x720  jpg", D.BDD100k, bdd)):
                base = cls(root, None, "train")
                cache = D.ResizedCache(cdir, name.split()[0], len(base), [base.filenames, base.filenamesGt], 512, 1024)
                ds = D.CachedSeg(base, cache, tf)
                t0 = time.perf_counter()
                for _ in torch.utils.data.DataLoader(ds, batch_size=6, num_workers=max(a.workers)):
                    pass
                print(f"{name}  resize cache, first touch (decode + resize + store), workers {max(a.workers)}: "
                      f"{len(ds) / (time.perf_counter() - t0):7.1f} img/s", flush=True)
                if a.cached:
                    for w in a.workers:
                        r = rate(ds, w, passes=6)
                        print(f"{name}  resize cache, later epochs (memory-mapped bytes), workers {w:2d}: {r:7.1f} img/s  "
                              f"({r / max(w, 1):6.1f} per worker)", flush=True)
                if a.device and torch.cuda.is_available():
                    res = D.DeviceResizedCache(ds, torch.device("cuda:0"), num_workers=max(a.workers))
                    r = rate_device(res.loader(6, 20, shuffle=True), passes=40)
                    print(f"{name}  HBM-resident cache ({res.img.numel() + res.lab.numel() >> 20} MiB on the GPU), "
                          f"gather + augment kernel, 0 workers: {r:7.1f} img/s", flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
