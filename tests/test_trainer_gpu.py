"""GPU: the trainer mirror end to end on the procedural dataset: train() + eval() + checkpoint
files with the reference's names and dict layout (train_new_task_step2.py:368-393,441-446)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_step2_trainer_end_to_end(tmp_path, monkeypatch):
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import ops
    from mdil_ss_amd import train_new_task_step2 as T
    from mdil_ss_amd.models.erfnet_RA_parallel import Net
    ops.invalidate_packs()
    work = tmp_path / "run"
    work.mkdir()
    monkeypatch.chdir(work)                     # the trainer writes to ../save/<savedir>
    # a step-1 checkpoint in the reference's format (DataParallel 'module.' prefix)
    torch.manual_seed(1)
    step1 = Net([20], 1, 0)
    ckpt = tmp_path / "step1.pth.tar"
    torch.save({"state_dict": {"module." + k: v for k, v in step1.state_dict().items()}}, ckpt)
    args = T.build_parser().parse_args([
        "--savedir", "t/CS1_BDD2", "--num-epochs", "2", "--batch-size", "2", "--state", str(ckpt),
        "--dataset", "BDD", "--dataset_old", "cityscapes", "--num-classes", "20", "20",
        "--current_task", "1", "--nb_tasks", "2", "--num-classes-old", "20", "--height", "32",
        "--width", "64", "--synthetic", "8", "--num-workers", "0", "--steps-loss", "2",
        "--model-name-suffix", "ours-CS1-BDD2", "--iouTrain"])
    model = T.main(args)
    save = tmp_path / "save" / "t" / "CS1_BDD2"
    for f in ("opts.txt", "model.txt", "automated_log.txt", "best.txt",
              "checkpoint_BDD_erfnet_RA_parallel_2_2ours-CS1-BDD2_step2.pth.tar",
              "model_best_BDD_erfnet_RA_parallel_2_2ours-CS1-BDD2_step2.pth.tar"):
        assert (save / f).exists(), f
    log = (save / "automated_log.txt").read_text().splitlines()
    assert log[0].startswith("Epoch\t\tTrain-loss") and len(log) == 3
    assert 0.0 < float(log[1].split("\t\t")[3]) < 1.0            # --iouTrain: train-IoU column filled
    # epoch-wise TensorBoard scalars (train_new_task_step2.py:115-117,351-355): 7 tags x 2 epochs
    import glob
    from mdil_ss_amd.scalar_log import read_scalars
    ev = glob.glob(str(work / "Adaptations" / "runs_BDD_erfnet_RA_parallel_2_2ours-CS1-BDD2_step2" / "events.out.tfevents.*"))
    assert len(ev) == 1, ev
    sc = read_scalars(ev[0])
    assert sorted({t for _, t, _ in sc}) == sorted(["total_train_loss", "KLD_loss_train", "ce_loss_train", "val_loss_BDD",
                                                    "val_acc_BDD", "val_loss_cityscapes", "val_acc_cityscapes"]), sc
    assert sorted({st for st, _, _ in sc}) == [1, 2] and len(sc) == 14
    ck = torch.load(save / "checkpoint_BDD_erfnet_RA_parallel_2_2ours-CS1-BDD2_step2.pth.tar",
                    map_location="cpu", weights_only=False)
    assert set(ck) == {"epoch", "arch", "state_dict", "best_acc", "optimizer"} and ck["epoch"] == 3
    assert all(k.startswith("module.") for k in ck["state_dict"]) and len(ck["state_dict"]) == 680
    # the frozen old-domain parameters did not move; trained ones did
    new = {k[7:]: v for k, v in ck["state_dict"].items()}
    old = step1.state_dict()
    assert torch.equal(new["decoder.0.output_conv.weight"], old["decoder.0.output_conv.weight"])
    assert torch.equal(new["encoder.layers.1.parallel_conv_1.0.weight"],
                       old["encoder.layers.1.parallel_conv_1.0.weight"])
    assert not torch.equal(new["encoder.layers.1.conv3x1_1.weight"], old["encoder.layers.1.conv3x1_1.weight"])
    assert not torch.equal(new["encoder.layers.1.parallel_conv_1.1.weight"],
                           old["encoder.layers.1.parallel_conv_1.0.weight"])


def test_step1_trainer_then_step2_chain(tmp_path, monkeypatch):
    """train_RAPFT_step1 (1 epoch) -> its checkpoint feeds train_new_task_step2 (--state), like the
    reference's trainer_OURS.sh:50,56."""
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import ops
    from mdil_ss_amd import train_RAPFT_step1 as T1
    from mdil_ss_amd import train_new_task_step2 as T2
    ops.invalidate_packs()
    work = tmp_path / "run"
    work.mkdir()
    monkeypatch.chdir(work)
    common = ["--batch-size", "2", "--height", "32", "--width", "64", "--synthetic", "8",
              "--num-workers", "0", "--steps-loss", "0", "--num-epochs", "1"]
    T1.main(T1.build_parser().parse_args(["--savedir", "s1", "--num-classes", "20",
                                         "--current_task", "0", "--dataset", "cityscapes"] + common))
    ck = tmp_path / "save" / "s1" / "model_best_cityscapes_erfnet_RA_parallel_1_2RAP_FT_step1.pth.tar"
    assert ck.exists()
    sd = torch.load(ck, map_location="cpu", weights_only=False)["state_dict"]
    assert len(sd) == 395 and all(k.startswith("module.") for k in sd)
    # step-1 TensorBoard scalars (train_RAPFT_step1.py:107-109,340-344)
    import glob
    from mdil_ss_amd.scalar_log import read_scalars
    ev = glob.glob(str(work / "Adaptations" / "runs_cityscapes_erfnet_RA_parallel_1_2RAP_FT_step1" / "events.out.tfevents.*"))
    assert len(ev) == 1, ev
    assert sorted({t for _, t, _ in read_scalars(ev[0])}) == ["train_loss", "val_acc_cityscapes", "val_loss_cityscapes"]
    ops.invalidate_packs()
    T2.main(T2.build_parser().parse_args(["--savedir", "s2", "--state", str(ck), "--dataset", "BDD",
                                         "--dataset_old", "cityscapes", "--num-classes", "20", "20",
                                         "--current_task", "1", "--nb_tasks", "2",
                                         "--num-classes-old", "20"] + common))
    assert (tmp_path / "save" / "s2" / "checkpoint_BDD_erfnet_RA_parallel_1_2RAPFT_KLD_step2.pth.tar").exists()


def test_rccl_exchange_path_single_rank():
    """The data-parallel exchange exactly as the N>1 bench/trainer drive it -- process group on
    backend 'nccl' (= RCCL), bucketed all-reduce of flat-gradient slices on the side stream, join,
    Adam with 1/world -- executed with a one-rank group on the one GPU of the test box.  The
    collective is an identity there, but every call (init with device_id, stream hand-off,
    all_reduce on views of the flat buffer, destroy) goes through RCCL."""
    import os
    import socket
    import torch.distributed as dist
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd.engine import GradExchange
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        ex = GradExchange()
        ex.world = 2                      # take the multi-rank code path
        flat = torch.arange(1 << 20, device=dev, dtype=torch.float32)
        want = flat.clone()
        ex.start(flat[1000:])
        ex.start(flat[:1000])
        ex.join()
        torch.cuda.synchronize()
        assert torch.equal(flat, want)
        assert ex.comm_stream is not None
        # the engine's 3-stream iteration with the collective in the loop: the new decoder's
        # bucket is reduced from a tensor hook while the encoder backward is still running
        from mdil_ss_amd import ops
        from mdil_ss_amd import train_new_task_step2 as T
        from mdil_ss_amd.engine import Step2Engine
        from mdil_ss_amd.models.erfnet_RA_parallel import Net
        ops.invalidate_packs()
        torch.manual_seed(0)
        student, teacher = Net([20, 20], 2, 1).to(dev), Net([20], 1, 0).to(dev)
        T.current_task = 1
        T.apply_step2_freeze(student, teacher, 1)
        eng = Step2Engine(student, teacher, torch.ones(20, device=dev), current_task=1,
                          is_shared=T.is_shared, is_ds_curr=T.is_DS_curr)
        eng.world = eng.exchange.world = 2          # multi-rank code path (grad scale 1/2)
        img = torch.rand(2, 3, 32, 64, device=dev)
        lab = torch.randint(0, 19, (2, 1, 32, 64), device=dev)
        seen = []
        inner = eng.exchange.start

        def spy(bucket, **kw):
            seen.append((bucket.data_ptr(), torch.cuda.current_stream().cuda_stream))
            inner(bucket, **kw)
        eng.exchange.start = spy
        for _ in range(3):
            total, ce, kld = eng.iteration(img, lab)
        torch.cuda.synchronize()
        assert eng.multi_stream and eng._dec_reduced
        # the decoder bucket's collective must be ordered after the new-task graph's stream (the
        # stream its weight-gradient kernels ran on), not after the default stream
        dec = [st for ptr, st in seen if ptr == eng.bucket_dec.data_ptr()]
        assert dec and dec[-1] == eng.s_new.cuda_stream, (dec, eng.s_new.cuda_stream)
        # the two deep stages of the shared-encoder bucket went out from hooks inside the backward
        # (on one of the two graph streams, after BOTH graphs had passed the stage)
        sb = eng.bucket_shared.data_ptr()
        for _, (a, b) in eng.shared_stages:
            st = [s_ for ptr, s_ in seen if ptr == sb + 4 * a]
            assert st and st[-1] in (eng.s_new.cuda_stream, eng.s_old.cuda_stream), (a, st)
        assert eng._stages_sent == [0, 1]
        assert all(bool(torch.isfinite(v)) for v in (total, ce, kld))
    finally:
        dist.destroy_process_group()


def test_evaluate_cli_on_step3_checkpoint(tmp_path):
    """mdil_ss_amd.evaluate (Evaluation_Notebook cells 5, 11): strict load of a module.-prefixed
    3-task checkpoint, per-task mIoU; eval() keeps the notebook's (iou_classes, iouVal) order."""
    import json
    import mdil_ss_amd  # noqa: F401
    from mdil_ss_amd import evaluate as E
    from mdil_ss_amd import ops
    from mdil_ss_amd.models.erfnet_RA_parallel import Net
    ops.invalidate_packs()
    torch.manual_seed(4)
    net = Net([20, 20, 27], 3, 2)
    ck = tmp_path / "m.pth.tar"
    torch.save({"state_dict": {"module." + k: v for k, v in net.state_dict().items()}}, ck)
    out = tmp_path / "r.json"
    rep = E.main(E.build_parser().parse_args([
        "--state", str(ck), "--num-classes", "20", "20", "27", "--datasets", "cityscapes", "BDD", "IDD",
        "--synthetic", "4", "--height", "32", "--width", "64", "--batch-size", "2", "--num-workers", "0",
        "--json", str(out)]))
    assert set(rep) == {"cityscapes", "BDD", "IDD"} and len(rep["IDD"]["iou_classes"]) == 26
    assert all(0.0 <= r["mIoU"] <= 1.0 for r in rep.values())
    assert json.loads(out.read_text())["BDD"]["task"] == 1
