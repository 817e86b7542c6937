"""Step-3 incremental trainer (third domain, two old domains distilled) on MI355X.

Mirrors ``train_new_task_step3.py`` of the reference: ``is_shared`` / ``is_DS_curr`` (:88-101),
the freeze rule (:229-241), the two-step hot loop (:303-356, ``engine.Step3Engine``), validation
of every dataset in ``--datasets`` at epoch 1 and every 10th epoch (:392-399), checkpoint naming
(:436-451) and the CLI (:606-651: ``--dataset-new``, ``--datasets`` replace step 2's
``--dataset`` / ``--dataset_old``).  ``eval`` takes the class count of the validated task as an
int here, as in that file (:464).

Deliberate differences: ``--synthetic N`` procedural data (no datasets offline); one process per
GPU + RCCL instead of DataParallel over devices [0,1,2] with the old model on device 3 (:497-498);
tensorboard scalars (:424-425) are printed, not written; the per-iteration ``.item()`` calls are
replaced by a host read every ``--steps-loss`` iterations.
"""
import os
import time
from argparse import ArgumentParser

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader

from . import train_new_task_step2 as S2
from .dataset import (MyCoTransform, ProceduralSeg, add_datadir_flags,  # noqa: F401
                      open_dataset, to_device_batch)
from . import ops
from .engine import Step3Engine
from . import engine as _engine
from .iouEval import iouEval
from .models.erfnet_RA_parallel import Net as Net_RAP
from .train_new_task_step2 import (CrossEntropyLoss2d, class_weights, is_shared,  # noqa: F401
                                   save_checkpoint, student_init_dict, _strip, _prefixed, _rank,
                                   _is_dist)

NUM_CLASSES = 27
current_task = 2


def is_DS_curr(n):
    t = current_task
    if "decoder.{}".format(t) in n:
        return True
    if "encoder" in n and ("bn" in n or "parallel_conv" in n):
        return ".{}.weight".format(t) in n or ".{}.bias".format(t) in n
    return False


def make_loaders(args):
    world = dist.get_world_size() if _is_dist() else 1
    t = args.datasets.index(args.dataset_new)
    if args.synthetic:
        tr = ProceduralSeg(args.synthetic, args.height, args.width, args.num_classes[t], seed=11,
                           domain=t)
    else:                                   # reference :155-171
        tr = open_dataset(args.dataset_new, "train", args, augment=True)
    resident = None
    if getattr(args, "cache_device", False) and not args.synthetic:
        # --cache-resized DIR --cache-device: the splits' post-Resize bytes live in HBM (dataset.py)
        if not getattr(args, "cache_resized", None):
            raise RuntimeError("--cache-device needs --cache-resized DIR")
        from .dataset import DeviceResizedCache
        dev = torch.device("cuda", torch.cuda.current_device())
        rank = dist.get_rank() if _is_dist() else 0
        resident = lambda ds: DeviceResizedCache(ds, dev, args.num_workers)
    sampler = None
    if resident is not None:
        loader = resident(tr).loader(args.batch_size, args.num_classes[t], True, True, rank, world)
    else:
        if world > 1:
            sampler = torch.utils.data.distributed.DistributedSampler(tr, shuffle=True, seed=0)
        loader = DataLoader(tr, num_workers=args.num_workers, batch_size=args.batch_size,
                            shuffle=sampler is None, sampler=sampler, drop_last=True)
    loader_val = {}
    for ind, d in enumerate(args.datasets):
        if args.synthetic:
            va = ProceduralSeg(max(args.synthetic // 4, args.batch_size), args.height, args.width,
                               args.num_classes[ind], seed=12 + ind, domain=ind)
        else:                               # reference :196-212
            va = open_dataset(d, "val", args, augment=False)
        if resident is not None:            # (every rank scores the whole set here, as without the cache)
            loader_val[d] = resident(va).loader(args.batch_size, args.num_classes[ind], False)
        else:
            loader_val[d] = DataLoader(va, num_workers=args.num_workers, batch_size=args.batch_size)
    return loader, loader_val


def train(args, model, model_old):
    global NUM_CLASSES
    dev = next(model.parameters()).device
    savedir = f"../save/{args.savedir}"
    criterion_val = {d: CrossEntropyLoss2d(class_weights(d).to(dev))
                     for d in ("cityscapes", "IDD", "BDD")}
    weight = criterion_val[args.dataset_new].weight
    loader, loader_val = make_loaders(args)
    print("global current_task: ", current_task)
    S2.apply_step2_freeze(model, model_old, current_task)          # same rule, :229-241
    log_path = savedir + "/automated_log.txt"
    if _rank() == 0:
        if not os.path.exists(log_path):
            with open(log_path, "a") as f:
                f.write("Epoch\t\tTrain-loss\t\tTest-loss\t\tTrain-IoU\t\tTest-IoU\t\tlearningRate")
        with open(savedir + "/model.txt", "w") as f:
            f.write(str(model))
    engine = Step3Engine(model, model_old, weight, current_task=current_task,
                         lambdac=args.lambdac, is_shared=is_shared, is_ds_curr=is_DS_curr,
                         teacher_train=not args.eval_teacher, legacy_zero_grad=args.legacy_zero_grad)
    optimizer = engine.optimizer
    best_acc = 0
    tag = "{}_{}_{}_{}{}_step{}".format(args.dataset_new, args.model, args.num_epochs,
                                        args.batch_size, args.model_name_suffix,
                                        len(args.num_classes))
    from .scalar_log import add_scalars, close_writer, open_writer
    writer = open_writer("Adaptations/runs_" + tag, _rank())          # :116-118
    for epoch in range(1, args.num_epochs + 1):
        NUM_CLASSES = args.num_classes[args.current_task]
        print("-----TRAINING - EPOCH---", epoch, "-----")
        optimizer.set_epoch(epoch, args.num_epochs)
        for g in optimizer.param_groups:
            print("LEARNING RATE: ", g["lr"])
        if hasattr(loader.sampler, "set_epoch"):
            loader.sampler.set_epoch(epoch)
        sums = torch.zeros(3, device=dev)
        n_it = 0
        t_epoch = time.time()
        for step, batch in enumerate(loader):
            images, labels = to_device_batch(batch, dev, NUM_CLASSES)
            ce, kld_prev1, kld_prev0 = engine.iteration(images, labels)
            kd = args.lambdac * (kld_prev1 + kld_prev0)
            sums += torch.stack([ce + kd, ce, kd])                  # :358-360
            n_it += 1
            if args.steps_loss > 0 and step % args.steps_loss == 0:
                avg = float(sums[0]) / n_it
                ops.check_labels()      # raises like torch's device assert if a label was out of range
                dt = (time.time() - t_epoch) / n_it / args.batch_size
                print(f"loss: {avg:0.4} (epoch: {epoch}, step: {step})",
                      "// Avg time/img: %.4f s" % dt)
        print("epoch took: ", time.time() - t_epoch)
        average_loss_val = {d: 0.0 for d in args.datasets}
        val_acc = {d: 0.0 for d in args.datasets}
        if epoch == 1 or epoch % 10 == 0:
            print("----- VALIDATING - EPOCH", epoch, "-----")
            for ind, d in enumerate(args.datasets):
                print("validate: ", d)
                average_loss_val[d], val_acc[d] = eval(model, loader_val[d], criterion_val[d], ind,
                                                       args.num_classes[ind], epoch)
        info = {}
        for d in args.datasets:
            info["val_acc_{}".format(d)] = val_acc[d]
            info["val_loss_{}".format(d)] = average_loss_val[d]
        print(info)
        add_scalars(writer, info, epoch)                                   # :425-426
        if val_acc[args.dataset_new] == 0:
            current_acc = -average_loss_val[args.dataset_new]
        else:
            current_acc = val_acc[args.dataset_new]
        is_best = current_acc > best_acc
        best_acc = max(current_acc, best_acc)
        if _rank() == 0:
            save_checkpoint({
                "epoch": epoch + 1, "arch": str(model),
                "state_dict": _prefixed(model.state_dict()),
                "best_acc": best_acc, "optimizer": optimizer.state_dict(),
            }, is_best, savedir + f"/checkpoint_{tag}.pth.tar", savedir + f"/model_best_{tag}.pth.tar")
            if is_best:
                with open(savedir + "/best.txt", "w") as f:
                    f.write("Best epoch is %d, with Val-IoU= %.4f" % (epoch, val_acc[args.dataset_new]))
    close_writer(writer)
    return model


def eval(model, dataset_loader, criterion, task, num_classes, epoch):
    """Validation pass (:464-504); ``num_classes`` is the class count of ``task``."""
    global NUM_CLASSES
    model.eval()
    _engine.broadcast_buffers(model)     # the model that is scored = the model rank 0 checkpoints
    dev = next(model.parameters()).device
    NUM_CLASSES = num_classes
    print("number of classes in current task: ", num_classes)
    print("validating task: ", task)
    meter = iouEval(num_classes, num_classes - 1)
    loss_sum = torch.zeros((), device=dev)
    n = 0
    with torch.no_grad():
        for step, batch in enumerate(dataset_loader):
            inputs, targets = to_device_batch(batch, dev, num_classes)
            outputs = model(inputs, task)
            loss_sum += criterion(outputs, targets[:, 0])
            n += 1
            meter.addBatch(outputs, targets)
    iou_val, _ = meter.getIoU()
    avg = float(loss_sum) / max(n, 1)
    ops.check_labels()      # raises like torch's device assert if a label was out of range
    print("EPOCH IoU on VAL set: ", "{:0.2f}".format(float(iou_val) * 100), "%")
    print("check val fn, loss, acc: ", avg, float(iou_val))
    return avg, float(iou_val)


def main(args):
    global current_task
    current_task = args.current_task
    S2.current_task = args.current_task
    print("\ndataset: ", args.dataset_new)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 and not _is_dist():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    savedir = f"../save/{args.savedir}"
    if _rank() == 0:
        os.makedirs(savedir, exist_ok=True)
        with open(savedir + "/opts.txt", "w") as f:
            f.write(str(args))
    assert args.model == "erfnet_RA_parallel", "Error: model definition not found"
    print(args.num_classes, args.num_classes_old, args.nb_tasks, args.dataset_new)
    model = Net_RAP(args.num_classes, args.nb_tasks, args.current_task)
    model_old = Net_RAP(args.num_classes_old, args.nb_tasks - 1, args.current_task - 1)
    if args.state:
        saved = torch.load(args.state, map_location="cpu")["state_dict"]
        model_old.load_state_dict(_strip(saved), strict=False)
        print("loading previous step weights - CS-RAPs, BDD-RAPs and shared weights from previous step.")
        keys = {"module." + k for k in model.state_dict()}
        saved = saved if any(k.startswith("module.") for k in saved) else _prefixed(saved)
        model.load_state_dict(_strip(student_init_dict(saved, keys, current_task)), strict=False)
        print("loaded model from checkpoint provided.")
    model.to(dev)
    model_old.to(dev)
    model = train(args, model, model_old)
    print("========== TRAINING FINISHED ===========")
    return model


def build_parser():
    p = ArgumentParser()
    p.add_argument("--cuda", action="store_true", default=True)
    p.add_argument("--model", default="erfnet_RA_parallel")
    p.add_argument("--dataset-new", default="IDD")
    p.add_argument("--datasets", nargs="+", required=True, default=["IDD", "CS", "BDD"],
                   help="pass list of datasets in order")
    p.add_argument("--num-classes", type=int, nargs="+", required=True, default=[20, 20, 27])
    p.add_argument("--num-classes-old", type=int, nargs="+", required=True, default=[20])
    p.add_argument("--nb_tasks", type=int, default=3)
    p.add_argument("--current_task", type=int, default=2)
    p.add_argument("--state")
    p.add_argument("--lambdac", type=float, default=0.1)
    p.add_argument("--port", type=int, default=8097)
    p.add_argument("--datadir", default=os.getenv("HOME", "") + "/datasets/cityscapes/")
    p.add_argument("--height", type=int, default=512)
    p.add_argument("--width", type=int, default=1024)
    p.add_argument("--num-epochs", type=int, default=150)
    p.add_argument("--num-workers", type=int, default=4)
    p.add_argument("--batch-size", type=int, default=6)
    p.add_argument("--steps-loss", type=int, default=50)
    p.add_argument("--steps-plot", type=int, default=50)
    p.add_argument("--epochs-save", type=int, default=0)
    p.add_argument("--savedir", required=True)
    p.add_argument("--decoder", action="store_true")
    p.add_argument("--pretrainedEncoder")
    p.add_argument("--iouTrain", action="store_true", default=False)
    p.add_argument("--iouVal", action="store_true", default=True)
    p.add_argument("--resume", action="store_true")
    p.add_argument("--model-name-suffix", default="RAPFT_KLD")
    p.add_argument("--synthetic", type=int, default=0,
                   help="train on N seeded procedural images (MI355X build extension)")
    add_datadir_flags(p)
    p.add_argument("--eval-teacher", action="store_true",
                   help="run the previous model in eval mode (the reference leaves it in train mode)")
    p.add_argument("--legacy-zero-grad", action="store_true",
                   help="torch<=1.x zero_grad semantics: the DS group also steps after the KD backward")
    return p


if __name__ == "__main__":
    main(build_parser().parse_args())
