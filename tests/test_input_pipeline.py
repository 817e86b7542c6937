"""Input pipeline: MyCoTransform / dataset.py.
CPU: the oracle restatement against the golden produced by the reference's own MyCoTransform; the
host half of the product (PIL resize + draws) against the oracle; dataset file discovery.
GPU: ``ops.augment_batch`` (mdil_augment_batch) bit-exact against the oracle and the golden."""
import os
import random

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import cotransform as CT

GOLD = os.path.join(os.path.dirname(__file__), "golden", "cotransform.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _pil_pair(img, lab):
    return Image.fromarray(img), Image.fromarray(lab).convert("P")


def test_oracle_cotransform_matches_reference(gold):
    H, W = 24, 40
    for i in range(12):                                            # natural draws, 20 classes
        x, y = CT.co_transform(*_pil_pair(gold["nat_src_img"][i], gold["nat_src_lab"][i]), H, W, 20,
                               tuple(gold["nat_params"][i]))
        assert np.array_equal(x.numpy(), gold["nat_out_img"][i]), i
        assert np.array_equal(y.numpy(), gold["nat_out_lab"][i]), i
    random.seed(1234)                                              # the draw order itself
    drawn = [CT.draw_params(True) for _ in range(12)]
    assert np.array_equal(np.array(drawn, dtype=np.int32), gold["nat_params"])
    for k, prm in enumerate(gold["frc_params"]):                   # all 50 combinations, 27 classes
        x, y = CT.co_transform(*_pil_pair(gold["frc_src_img"], gold["frc_src_lab"]), H, W, 27, tuple(prm))
        assert np.array_equal(x.numpy(), gold["frc_out_img"][k]), prm
        assert np.array_equal(y.numpy(), gold["frc_out_lab"][k]), prm
    x, y = CT.co_transform(*_pil_pair(gold["frc_src_img"], gold["frc_src_lab"]), H, W, 27, None,
                           augment=False)
    assert np.array_equal(x.numpy(), gold["val_out_img"]) and np.array_equal(y.numpy(), gold["val_out_lab"])
    # the crop overhang of a negative shift is filled with 0 in the LABEL too (class 0, not void)
    k = [tuple(p) for p in gold["frc_params"]].index((0, -2, 0))
    assert np.all(gold["frc_out_lab"][k][0, :, -2:] == 0)
    k = [tuple(p) for p in gold["frc_params"]].index((0, 2, 0))
    assert np.all(gold["frc_out_lab"][k][0, :, :2] == 26)           # expand border 255 -> C-1


def test_host_cotransform_draws_and_bytes(gold):
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd.dataset import MyCoTransform
    co = MyCoTransform(True, 24, 40)
    random.seed(1234)
    for i in range(12):
        u8, l8, prm = co(*_pil_pair(gold["nat_src_img"][i], gold["nat_src_lab"][i]))
        assert u8.dtype == torch.uint8 and tuple(u8.shape) == (24, 40, 3) and tuple(l8.shape) == (24, 40)
        assert np.array_equal(prm.numpy(), gold["nat_params"][i])
        ri, rl = CT.resize_pair(*_pil_pair(gold["nat_src_img"][i], gold["nat_src_lab"][i]), 24, 40)
        assert np.array_equal(u8.numpy(), np.array(ri)) and np.array_equal(l8.numpy(), np.array(rl))
    st = random.getstate()
    MyCoTransform(False, 24, 40)(*_pil_pair(gold["frc_src_img"], gold["frc_src_lab"]))
    assert random.getstate() == st, "validation transform must not consume random draws"


def _write(path, arr):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(arr).save(path)


def test_dataset_discovery_and_decode(tmp_path):
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import dataset as D
    g = np.random.default_rng(0)
    rgb = lambda: g.integers(0, 256, (8, 12, 3), dtype=np.uint8)
    lab = lambda: g.integers(0, 19, (8, 12), dtype=np.uint8)
    cs = tmp_path / "cityscapes"
    labs = {}
    for city, stem in (("zurich", "zurich_000001_000019"), ("aachen", "aachen_000000_000019"),
                       ("aachen", "aachen_000003_000019")):
        _write(str(cs / "leftImg8bit/train" / city / f"{stem}_leftImg8bit.png"), rgb())
        labs[stem] = lab()
        _write(str(cs / "gtFine/train" / city / f"{stem}_gtFine_labelTrainIds.png"), labs[stem])
        _write(str(cs / "gtFine/train" / city / f"{stem}_gtFine_labelIds.png"), lab())   # must be ignored
    ds = D.cityscapes(str(cs) + "/", None, "train")
    assert len(ds) == 3 and [os.path.basename(f)[:6] for f in ds.filenames] == ["aachen", "aachen", "zurich"]
    assert all(f.endswith("_labelTrainIds.png") for f in ds.filenamesGt)
    img, lb = ds[2]
    assert img.mode == "RGB" and lb.mode == "P" and np.array_equal(np.array(lb), labs["zurich_000001_000019"])
    bdd = tmp_path / "bdd"
    for stem in ("b1", "a7"):
        _write(str(bdd / "images/val" / f"{stem}.jpg"), rgb())
        _write(str(bdd / "labels/val" / f"{stem}_train_id.png"), lab())
    ds = D.BDD100k(str(bdd) + "/", D.MyCoTransform(False, 8, 16), "val")
    assert len(ds) == 2 and os.path.basename(ds.filenames[0]) == "a7.jpg"
    u8, l8, prm = ds[0]
    assert tuple(u8.shape) == (8, 16, 3) and tuple(prm.tolist()) == (0, 0, 0)
    idd = tmp_path / "idd"
    raw = np.array([[0, 1, 2, 3], [22, 23, 25, 255]], dtype=np.uint8)
    _write(str(idd / "leftImg8bit/train/0/000_leftImg8bit.png"), rgb())
    _write(str(idd / "gtFine/train/0/000_gtFine_labellevel3Ids.png"), raw)
    assert np.array_equal(np.array(D.IDD(str(idd) + "/", None, "train")[0][1]), raw)
    uni = np.array(D.IDD_union(str(idd) + "/", None, "train")[0][1])
    assert np.array_equal(uni, np.array([[0, 19, 1, 20], [2, 27, 10, 255]], dtype=np.uint8))


def _golden_tree(tmp_path, gold):
    """The golden's 12 source images / labels as a Cityscapes-layout PNG tree (lossless)."""
    cs = tmp_path / "cs"
    for i in range(12):
        _write(str(cs / f"leftImg8bit/train/g/g_{i:03d}_leftImg8bit.png"), gold["nat_src_img"][i])
        _write(str(cs / f"gtFine/train/g/g_{i:03d}_gtFine_labelTrainIds.png"), gold["nat_src_lab"][i])
    return str(cs) + "/"


def test_resized_cache_is_byte_identical_to_the_pil_path(tmp_path, gold):
    """--cache-resized (dataset.py:75-113 + train_new_task_step2.py:56-57): the bytes
    served from the cache -- first touch (decode + resize + store), second touch (memory map) and
    a later run on the same directory -- equal what the uncached MyCoTransform path produces, and
    the draws are consumed in the same order."""
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import dataset as D
    root = _golden_tree(tmp_path, gold)
    H, W = 24, 40
    plain = D.cityscapes(root, D.MyCoTransform(True, H, W), "train")
    random.seed(1234)
    want = [plain[i] for i in range(12)]
    for i in range(12):                                   # the golden of the reference's own transform
        ri, rl = CT.resize_pair(*_pil_pair(gold["nat_src_img"][i], gold["nat_src_lab"][i]), H, W)
        assert np.array_equal(want[i][0].numpy(), np.array(ri)) and np.array_equal(want[i][1].numpy(), np.array(rl))
        assert np.array_equal(want[i][2].numpy(), gold["nat_params"][i])
    cdir = str(tmp_path / "cache")

    def cached():
        base = D.cityscapes(root, None, "train")
        cache = D.ResizedCache(cdir, "cityscapes_train", 12, [base.filenames, base.filenamesGt], H, W)
        return D.CachedSeg(base, cache, D.MyCoTransform(True, H, W)), cache

    ds, cache = cached()
    assert cache.filled() == 0
    for touch in ("first", "second"):
        random.seed(1234)
        for i in range(12):
            u8, l8, prm = ds[i]
            assert torch.equal(u8, want[i][0]) and torch.equal(l8, want[i][1]) and torch.equal(prm, want[i][2]), (touch, i)
        assert cache.filled() == 12
    # a later run: a new object over the same directory serves the bytes without touching the files
    ds2, cache2 = cached()
    assert cache2.filled() == 12
    ds2.base.filenames = ["/nonexistent"] * 12            # decoding would raise
    random.seed(1234)
    assert all(torch.equal(ds2[i][0], want[i][0]) and torch.equal(ds2[i][1], want[i][1]) for i in range(12))
    # another target size is another cache (never a stale hit)
    base = D.cityscapes(root, None, "train")
    other = D.ResizedCache(cdir, "cityscapes_train", 12, [base.filenames, base.filenamesGt], H, W + 8)
    assert other.filled() == 0 and other.paths[0] != cache.paths[0]
    # through DataLoader workers (two processes filling one memory-mapped cache)
    cdir2 = str(tmp_path / "cache2")
    base = D.cityscapes(root, None, "train")
    c3 = D.ResizedCache(cdir2, "cityscapes_train", 12, [base.filenames, base.filenamesGt], H, W)
    ds3 = D.CachedSeg(base, c3, D.MyCoTransform(False, H, W))
    got = [b for b in torch.utils.data.DataLoader(ds3, batch_size=4, num_workers=2)]
    assert c3.filled() == 12
    assert torch.equal(torch.cat([b[0] for b in got]), torch.stack([w[0] for w in want]))
    assert torch.equal(torch.cat([b[1] for b in got]), torch.stack([w[1] for w in want]))


def _cache_rank(rank, cdir, barrier, n):
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import dataset as D
    barrier.wait()                                   # both ranks open the cache at the same moment
    c = D.ResizedCache(cdir, "split", n, ["ident"], 8, 12)
    for i in range(rank, n, 2):                      # each fills its own half
        c.put(i, np.full((8, 12, 3), i, dtype=np.uint8), np.full((8, 12), 100 + i, dtype=np.uint8))


def test_resized_cache_opened_by_two_ranks_at_once(tmp_path):
    """Data parallel: every rank opens the same cache directory at start-up.  Exactly one of them
    may create a file -- all must map the same inode, or a 'filled' flag written by one rank would
    vouch for bytes another rank's file never received."""
    import multiprocessing as mp
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import dataset as D
    ctx = mp.get_context("spawn")
    for trial in range(3):
        cdir, n = str(tmp_path / f"c{trial}"), 40
        barrier = ctx.Barrier(2)
        ps = [ctx.Process(target=_cache_rank, args=(r, cdir, barrier, n)) for r in range(2)]
        for p_ in ps:
            p_.start()
        for p_ in ps:
            p_.join(60)
        assert all(p_.exitcode == 0 for p_ in ps)
        c = D.ResizedCache(cdir, "split", n, ["ident"], 8, 12)
        assert c.filled() == n
        for i in range(n):
            img, lab = c.get(i)
            assert (img == i).all() and (lab == 100 + i).all(), (trial, i)


@pytest.mark.gpu
def test_device_resident_cache_loader_matches_the_host_path(tmp_path, gold):
    """--cache-device: gather from the HBM-resident bytes + augment kernel == DataLoader over the
    PIL path + collate + to_device_batch, for the same indices and draws; bit-exact against the
    golden of the reference's own MyCoTransform."""
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import dataset as D
    dev = torch.device("cuda:0")
    root = _golden_tree(tmp_path, gold)
    H, W = 24, 40
    base = D.cityscapes(root, None, "train")
    cache = D.ResizedCache(str(tmp_path / "c"), "cityscapes_train", 12, [base.filenames, base.filenamesGt], H, W)
    resident = D.DeviceResizedCache(D.CachedSeg(base, cache, D.MyCoTransform(True, H, W)), dev, num_workers=2)
    assert cache.filled() == 12 and len(resident) == 12
    # validation order (no shuffle): sample i gets the golden's draws i -> the golden's outputs
    random.seed(1234)
    xs, ys = zip(*list(resident.loader(5, 20, shuffle=False)))
    assert [x.shape[0] for x in xs] == [5, 5, 2]
    assert torch.equal(torch.cat(xs).cpu(), torch.from_numpy(gold["nat_out_img"]))
    assert torch.equal(torch.cat(ys).cpu(), torch.from_numpy(gold["nat_out_lab"]))
    assert xs[0].permute(0, 2, 3, 1).is_contiguous()
    # shuffled epoch: same values as the host path fed the same permutation and draws
    ld = resident.loader(4, 20, shuffle=True, drop_last=True, seed=3)
    ld.set_epoch(2)
    perm = torch.randperm(12, generator=torch.Generator().manual_seed(5))
    random.seed(77)
    got = list(ld)
    random.seed(77)
    plain = D.cityscapes(root, D.MyCoTransform(True, H, W), "train")
    for b, (x, y) in enumerate(got):
        items = [plain[int(i)] for i in perm[4 * b:4 * b + 4]]
        xw, yw = D.to_device_batch(torch.utils.data.default_collate(items), dev, 20)
        assert torch.equal(x, xw) and torch.equal(y, yw), b
    # two ranks: the epoch's samples are split without overlap, every rank makes the same number of steps
    a = resident.loader(3, 20, True, True, rank=0, world=2, seed=1)._indices()
    b_ = resident.loader(3, 20, True, True, rank=1, world=2, seed=1)._indices()
    assert len(a) == len(b_) == 6 and sorted(torch.cat([a, b_]).tolist()) == list(range(12))
    va = resident.loader(5, 20, False, False, rank=1, world=2)._indices()
    assert va.tolist() == list(range(1, 12, 2))


@pytest.mark.gpu
def test_augment_batch_bit_exact(gold):
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import ops
    dev = torch.device("cuda:0")
    H, W = 24, 40
    ri, rl = CT.resize_pair(*_pil_pair(gold["frc_src_img"], gold["frc_src_lab"]), H, W)
    n = len(gold["frc_params"])
    img = torch.from_numpy(np.array(ri)).unsqueeze(0).repeat(n, 1, 1, 1).to(dev)
    lab = torch.from_numpy(np.array(rl)).unsqueeze(0).repeat(n, 1, 1).to(dev)
    x, y = ops.augment_batch(img, lab, torch.from_numpy(gold["frc_params"]).to(dev), 27)
    assert tuple(x.shape) == (n, 3, H, W) and x.permute(0, 2, 3, 1).is_contiguous()
    assert torch.equal(x.cpu(), torch.from_numpy(gold["frc_out_img"]))
    assert torch.equal(y.cpu(), torch.from_numpy(gold["frc_out_lab"]))
    # natural draws through the host half + collate + device half, like the trainer does
    from mdil_ss_amd.dataset import MyCoTransform, to_device_batch
    co = MyCoTransform(True, H, W)
    random.seed(1234)
    items = [co(*_pil_pair(gold["nat_src_img"][i], gold["nat_src_lab"][i])) for i in range(12)]
    batch = torch.utils.data.default_collate(items)
    x, y = to_device_batch(batch, dev, 20)
    assert torch.equal(x.cpu(), torch.from_numpy(gold["nat_out_img"]))
    assert torch.equal(y.cpu(), torch.from_numpy(gold["nat_out_lab"]))
    # full-size random batch against the oracle (size-independent property: every draw combination)
    g = np.random.default_rng(5)
    Hf, Wf = 512, 1024
    src = g.integers(0, 256, (Hf, Wf, 3), dtype=np.uint8)
    sl = g.integers(0, 20, (Hf, Wf), dtype=np.uint8)
    sl[g.random((Hf, Wf)) < 0.03] = 255
    prm = np.array([[1, -2, 2], [0, 2, -1], [1, 0, 0]], dtype=np.int32)
    x, y = ops.augment_batch(torch.from_numpy(src).unsqueeze(0).repeat(3, 1, 1, 1).to(dev),
                             torch.from_numpy(sl).unsqueeze(0).repeat(3, 1, 1).to(dev),
                             torch.from_numpy(prm).to(dev), 20)
    for i in range(3):
        xo, yo = CT.co_transform(*_pil_pair(src, sl), Hf, Wf, 20, tuple(prm[i]))
        assert torch.equal(x[i].cpu(), xo) and torch.equal(y[i].cpu(), yo), prm[i]


@pytest.mark.gpu
@pytest.mark.parametrize("cache", ["none", "host", "device"])
def test_step2_trainer_on_disk_datasets(tmp_path, monkeypatch, cache):
    """train_new_task_step2 end to end WITHOUT --synthetic: PNG/JPG trees in the reference's
    directory layout -> dataset classes -> host co-transform -> DataLoader collate ->
    ops.augment_batch -> Step2Engine; validation on the new and the old dataset.  Also through the
    resize cache (--cache-resized) and its HBM-resident form (--cache-device), two epochs so that
    the second one is served from the cache."""
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import ops
    from mdil_ss_amd import train_new_task_step2 as T
    from mdil_ss_amd.models.erfnet_RA_parallel import Net
    ops.invalidate_packs()
    g = np.random.default_rng(3)

    def pair(h, w, ncls):
        lab = g.integers(0, ncls - 1, (h // 8, w // 8), dtype=np.uint8).repeat(8, 0).repeat(8, 1)
        lab[:2] = 255
        pal = g.integers(0, 256, (256, 3), dtype=np.uint8)
        return pal[lab], lab

    cs, bdd = tmp_path / "cs", tmp_path / "bdd"
    for sub, n in (("train", 2), ("val", 2)):
        for i in range(n):
            im, lb = pair(48, 96, 20)
            _write(str(cs / f"leftImg8bit/{sub}/c/c_{i:03d}_leftImg8bit.png"), im)
            _write(str(cs / f"gtFine/{sub}/c/c_{i:03d}_gtFine_labelTrainIds.png"), lb)
    for sub, n in (("train", 4), ("val", 2)):
        for i in range(n):
            im, lb = pair(40, 72, 20)
            _write(str(bdd / f"images/{sub}/{i:04d}.jpg"), im)
            _write(str(bdd / f"labels/{sub}/{i:04d}_train_id.png"), lb)
    work = tmp_path / "run"
    work.mkdir()
    monkeypatch.chdir(work)
    torch.manual_seed(1)
    ckpt = tmp_path / "step1.pth.tar"
    torch.save({"state_dict": {"module." + k: v for k, v in Net([20], 1, 0).state_dict().items()}}, ckpt)
    args = T.build_parser().parse_args([
        "--savedir", "disk", "--num-epochs", "1" if cache == "none" else "2", "--batch-size", "2", "--state", str(ckpt),
        "--dataset", "BDD", "--dataset_old", "cityscapes", "--num-classes", "20", "20",
        "--current_task", "1", "--nb_tasks", "2", "--num-classes-old", "20", "--height", "32",
        "--width", "64", "--num-workers", "0", "--steps-loss", "1",
        "--cs-datadir", str(cs) + "/", "--bdd-datadir", str(bdd) + "/"] + (
        [] if cache == "none" else ["--cache-resized", str(tmp_path / "cache")] + (
            ["--cache-device"] if cache == "device" else [])))
    random.seed(0)
    T.main(args)
    log = (tmp_path / "save" / "disk" / "automated_log.txt").read_text().splitlines()
    assert len(log) == (2 if cache == "none" else 3) and np.isfinite(float(log[-1].split("\t\t")[1]))
    if cache != "none":
        files = sorted(os.listdir(tmp_path / "cache"))
        assert len(files) == 9 and all(f.endswith(".u8") for f in files), files     # 3 splits x (img, lab, ok)
        for f in files:
            if f.endswith(".ok.u8"):
                assert np.fromfile(tmp_path / "cache" / f, dtype=np.uint8).all(), f     # every sample was cached


def test_class_weights_and_named_datasets(tmp_path):
    import types
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import cal_class_weights as CW
    from mdil_ss_amd import dataset_custom as DC
    g = np.random.default_rng(1)
    root = tmp_path / "cs"
    total = np.zeros(20)
    for i in range(3):
        lab = g.integers(0, 19, (16, 24), dtype=np.uint8)
        lab[g.random((16, 24)) < 0.1] = 255
        _write(str(root / f"gtFine/train/x/x_{i}_gtFine_labelTrainIds.png"), lab)
        _write(str(root / f"leftImg8bit/train/x/x_{i}_leftImg8bit.png"),
               g.integers(0, 256, (16, 24, 3), dtype=np.uint8))
        c = np.bincount(lab.reshape(-1), minlength=256)
        total[:19] += c[:19]
        total[19] += c[255]
    w = CW.calc_weights(types.SimpleNamespace(datadir=str(root) + "/", dataset="cityscapes", num_classes=20))
    p = (total + 1) / (total + 1).sum()
    want = 1.0 / np.log(p + 1.1)
    want[19] = 0
    np.testing.assert_allclose(w, want, rtol=1e-12)
    ds = DC.cityscapes(str(root) + "/", None, None, "train")
    img, lb, fn, fg = ds[1]
    assert img.mode == "RGB" and lb.mode == "P" and fn.endswith("x_1_leftImg8bit.png")
    assert fg.endswith("x_1_gtFine_labelTrainIds.png") and len(ds) == 3
