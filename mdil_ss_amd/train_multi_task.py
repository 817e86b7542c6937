"""Multi-task joint trainer (all domains at once, shared encoder + one head per domain) on MI355X.

Mirrors ``train_multi_task.py`` of the reference: ``is_shared`` / ``is_DS_curr`` (:107,110), the
optimizer groups (``5e-4/nb_tasks`` for the encoder, :212-220), the round-robin inner loop
(:249-265, ``engine.MultiTaskEngine``), validation of every dataset at epoch 1 and every 5th epoch
(:281-291), best-model rule on the mean IoU (:306-313), file names (:318-323) and the CLI
(:430-468).  The reference file does not import as shipped (``elif`` without ``if`` at :403); the
model selection below is what that line intends.  ``--synthetic N`` as in the other trainers.
"""
import os
import time
from argparse import ArgumentParser

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader

from .dataset import (MyCoTransform, ProceduralSeg, add_datadir_flags,  # noqa: F401
                      open_dataset, to_device_batch)
from . import ops
from .engine import MultiTaskEngine
from . import engine as _engine
from .iouEval import iouEval
from .models.erfnet_multi_task import Net as Net_MT
from .train_new_task_step2 import (CrossEntropyLoss2d, class_weights, save_checkpoint,  # noqa: F401
                                   _strip, _prefixed, _rank, _is_dist)
import re

NUM_CLASSES = 20
current_task = 0
DATASET_WEIGHTS = {"CS": "cityscapes", "BDD": "BDD", "IDD": "IDD"}      # :158-175


def is_shared(n):
    return "encoder" in n


def is_DS_curr(n):
    return "decoder" in n


def make_loaders(args):
    world = dist.get_world_size() if _is_dist() else 1
    loader_train, loader_val = {}, {}
    for ind, d in enumerate(args.datasets):
        if args.synthetic:
            tr = ProceduralSeg(args.synthetic, args.height, args.width, args.num_classes[ind],
                               seed=11 + 10 * ind, domain=ind)
            va = ProceduralSeg(max(args.synthetic // 4, 2), args.height, args.width,
                               args.num_classes[ind], seed=12 + 10 * ind, domain=ind)
        else:                               # reference :158-175
            tr = open_dataset(d, "train", args, augment=True)
            va = open_dataset(d, "val", args, augment=False)
        sampler = None
        if world > 1:
            sampler = torch.utils.data.distributed.DistributedSampler(tr, shuffle=True, seed=ind)
        loader_train[d] = DataLoader(tr, num_workers=args.num_workers, batch_size=args.batch_size,
                                     shuffle=sampler is None, sampler=sampler)
        loader_val[d] = DataLoader(va, num_workers=args.num_workers, batch_size=2, drop_last=True)
    return loader_train, loader_val


def train(args, model):
    global NUM_CLASSES
    print("datasets: ", args.datasets)
    print("nb_tasks: ", args.nb_tasks)
    print("dataset_name: ", args.dataset)
    print("num_classes: ", args.num_classes)
    dev = next(model.parameters()).device
    savedir = f"../save/{args.savedir}"
    ce_loss = {d: CrossEntropyLoss2d(class_weights(DATASET_WEIGHTS[d]).to(dev)) for d in args.datasets}
    loader_train, loader_val = make_loaders(args)
    log_path = savedir + "/automated_log.txt"
    if _rank() == 0:
        if not os.path.exists(log_path):
            with open(log_path, "a") as f:
                f.write("Epoch\t\tTrain-loss\t\tTest-loss\t\tTrain-IoU\t\tTest-IoU\t\tlearningRate")
        with open(savedir + "/model.txt", "w") as f:
            f.write(str(model))
    print("\nusing learning rate this for the W_s params", 5e-4 / args.nb_tasks, "\n")
    print("using 5e-4 lr for W_t")
    engine = MultiTaskEngine(model, [ce_loss[d].weight for d in args.datasets])
    optimizer = engine.optimizer
    best_acc = 0
    n_iters = min(len(loader_train[d]) for d in args.datasets)
    print("n_iters ", n_iters)
    tag = "{}_{}_{}_{}{}_step{}".format(args.dataset, args.model, args.num_epochs, args.batch_size,
                                        args.model_name_suffix, len(args.num_classes))
    from .scalar_log import add_scalars, close_writer, open_writer
    writer = open_writer("Adaptations/runs_" + tag, _rank())          # :122-124
    for epoch in range(1, args.num_epochs + 1):
        print("-----TRAINING - EPOCH---", epoch, "-----")
        optimizer.set_epoch(epoch, args.num_epochs)
        for g in optimizer.param_groups[:2]:
            print("LEARNING RATE: ", g["lr"])
        iterator = {}
        for d in args.datasets:
            if hasattr(loader_train[d].sampler, "set_epoch"):
                loader_train[d].sampler.set_epoch(epoch)
            iterator[d] = iter(loader_train[d])
        sums = torch.zeros(len(args.datasets), device=dev)
        t_epoch = time.time()
        model.train()
        for itr in range(n_iters):
            for ind, d in enumerate(args.datasets):
                NUM_CLASSES = args.num_classes[ind]
                images, labels = to_device_batch(next(iterator[d]), dev, NUM_CLASSES)
                loss = engine.sub_step(ind, images, labels)
                sums[ind] += loss
        average_epoch_loss_train = {d: float(sums[i]) / max(n_iters, 1)
                                    for i, d in enumerate(args.datasets)}
        print("epoch took: ", time.time() - t_epoch)
        average_loss_val = {d: 0.0 for d in args.datasets}
        val_acc = {d: 0.0 for d in args.datasets}
        if epoch % 5 == 0 or epoch == 1:
            for ind, d in enumerate(args.datasets):
                print("validate: ", d)
                average_loss_val[d], val_acc[d] = eval(model, loader_val[d], ce_loss[d], ind,
                                                       args.num_classes, epoch)
        info = {}
        for d in args.datasets:
            info["val_acc_{}".format(d)] = val_acc[d]
            info["val_loss_{}".format(d)] = average_loss_val[d]
            info["train_loss_{}".format(d)] = average_epoch_loss_train[d]
        print(info)
        add_scalars(writer, info, epoch)                                   # :300-301
        temp_acc = sum(val_acc[k] for k in args.datasets)
        current_acc = -0.0 if temp_acc == 0 else temp_acc / len(args.datasets)
        is_best = current_acc > best_acc
        best_acc = max(current_acc, best_acc)
        if _rank() == 0:
            save_checkpoint({
                "epoch": epoch + 1, "arch": str(model),
                "state_dict": _prefixed(model.state_dict()),
                "best_acc": best_acc, "optimizer": optimizer.state_dict(),
            }, is_best, savedir + f"/checkpoint_{tag}.pth.tar", savedir + f"/model_best_{tag}.pth.tar")
    close_writer(writer)
    return model


def eval(model, dataset_loader, criterion, task, num_classes, epoch):
    """Validation pass (:331-371); ``num_classes`` is the list, indexed by ``task``."""
    global NUM_CLASSES
    model.eval()
    _engine.broadcast_buffers(model)     # the model that is scored = the model rank 0 checkpoints
    dev = next(model.parameters()).device
    num_cls = num_classes[task]
    NUM_CLASSES = num_cls
    print("number of classes in current task: ", num_cls)
    print("validating task: ", task)
    meter = iouEval(num_cls, num_cls - 1)
    loss_sum = torch.zeros((), device=dev)
    n = 0
    with torch.no_grad():
        for step, batch in enumerate(dataset_loader):
            inputs, targets = to_device_batch(batch, dev, num_cls)
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
    print("\ndataset: ", args.dataset)
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
    assert args.model == "erfnet_multi_task", "Error: model definition not found"
    print(args.num_classes, args.nb_tasks, args.dataset)
    model = Net_MT(args.num_classes, args.nb_tasks, args.current_task)
    if args.state:
        saved = torch.load(args.state, map_location="cpu")["state_dict"]
        print("loading ImageNet pre-trained enc")
        new = {re.sub("module.features", "module", k): v for k, v in saved.items()}   # :418-420
        model.load_state_dict(_strip(new), strict=False)
    print("loaded\n")
    model.to(dev)
    model = train(args, model)
    print("========== TRAINING FINISHED ===========")
    return model


def build_parser():
    p = ArgumentParser()
    p.add_argument("--cuda", action="store_true", default=True)
    p.add_argument("--model", default="erfnet_multi_task")
    p.add_argument("--dataset", default="CSBDD")
    p.add_argument("--datasets", nargs="+", required=True, default=["CS", "BDD"])
    p.add_argument("--dlr", type=float, default=100.0)
    p.add_argument("--num-classes", type=int, nargs="+", required=True, default=[20])
    p.add_argument("--nb_tasks", type=int, default=1)
    p.add_argument("--current_task", type=int, default=0)
    p.add_argument("--state")
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
    p.add_argument("--model-name-suffix", default="RAP_FT")
    p.add_argument("--synthetic", type=int, default=0,
                   help="train on N seeded procedural images (MI355X build extension)")
    add_datadir_flags(p)
    return p


if __name__ == "__main__":
    main(build_parser().parse_args())
