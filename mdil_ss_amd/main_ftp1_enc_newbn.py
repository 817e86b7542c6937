"""Fine-tuning / feature-extraction baseline, first increment (one old + one new decoder head) on
MI355X -- mirror of the reference's ``main_ftp1_enc_newbn.py``: flags (:466-498), checkpoint
loading with ``decoder -> decoder_old`` key remap (:213-221), freeze rule and optimizers
(:228-243), validation of the new and the old dataset every epoch (:318-324), 8-column
``automated_log.txt`` row (:359-361), file names (:339-344).  The hot loop is
``engine.FineTuneEngine``.  ``--synthetic N`` as in the other trainers."""
import os
import re
import time
from argparse import ArgumentParser

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader

from .dataset import (MyCoTransform, ProceduralSeg, add_datadir_flags,  # noqa: F401
                      open_dataset, to_device_batch)
from .engine import FineTuneEngine
from . import engine as _engine
from .iouEval import iouEval
from .models.erfnet import NetFT1 as Net_ftp1
from .train_new_task_step2 import (CrossEntropyLoss2d, class_weights, save_checkpoint,  # noqa: F401
                                   _strip, _prefixed, _rank, _is_dist)

NUM_CLASSES = 20
NUM_CLASSES_old = 20
NUM_CLASSES_new = 20


def _loader(ds, args, train):
    world = dist.get_world_size() if _is_dist() else 1
    sampler = None
    if train and world > 1:
        sampler = torch.utils.data.distributed.DistributedSampler(ds, shuffle=True, seed=0)
    return DataLoader(ds, num_workers=args.num_workers, batch_size=args.batch_size,
                      shuffle=train and sampler is None, sampler=sampler, drop_last=train)


def make_loaders(args, names_classes, new_index):
    """names_classes: [(dataset name, class count)] in task order -> (train loader of the new
    dataset, {name: val loader})."""
    val = {}
    for ind, (name, nc) in enumerate(names_classes):
        if args.synthetic:
            ds = ProceduralSeg(max(args.synthetic // 4, args.batch_size), args.height, args.width, nc,
                               seed=12 + ind, domain=ind)
        else:
            ds = open_dataset(name, "val", args, augment=False)
        val[name] = _loader(ds, args, False)
    name, nc = names_classes[new_index]
    if args.synthetic:
        tr = ProceduralSeg(args.synthetic, args.height, args.width, nc, seed=11, domain=new_index)
    else:
        tr = open_dataset(name, "train", args, augment=True)
    return _loader(tr, args, True), val


def run_epochs(args, model, engine, loader, evaluate, tag, log_row, scalars=None):
    """Epoch loop shared by both fine-tuning trainers (reference :253-361).  ``scalars(avg_train) -> dict``:
    the epoch's TensorBoard scalars, written under 'Finetuning_Baselines/runs_<model>_<epochs>_<batch><suffix>'
    like the reference's ``writer`` (:109-111 / main_FT2_flexible_new.py:108-110)."""
    from .scalar_log import add_scalars, close_writer, open_writer
    writer = open_writer("Finetuning_Baselines/runs_{}_{}_{}{}".format(
        args.model, args.num_epochs, args.batch_size, args.model_name_suffix), _rank())
    dev = next(model.parameters()).device
    savedir = f"../save/{args.savedir}"
    log_path = savedir + "/automated_log.txt"
    if _rank() == 0 and not os.path.exists(log_path):
        with open(log_path, "a") as f:
            f.write("Epoch\t\tTrain-loss\t\tTest-loss\t\tTrain-IoU\t\tTest-IoU\t\tlearningRate")
    optimizer = engine.optimizer
    best_acc = 0
    for epoch in range(1, args.num_epochs + 1):
        print("----- TRAINING - EPOCH", epoch, "-----")
        optimizer.set_epoch(epoch, args.num_epochs)
        used_lr = float(optimizer.param_groups[0]["lr"])
        print("LEARNING RATE: ", used_lr)
        if hasattr(loader.sampler, "set_epoch"):
            loader.sampler.set_epoch(epoch)
        iou_train = iouEval(NUM_CLASSES_new, NUM_CLASSES_new - 1) if args.iouTrain else None
        loss_sum, n_it, t0 = torch.zeros((), device=dev), 0, time.time()
        for step, batch in enumerate(loader):
            images, labels = to_device_batch(batch, dev, NUM_CLASSES_new)
            loss_sum += engine.iteration(images, labels)
            n_it += 1
            if iou_train is not None:
                iou_train.addBatch(engine.last_outputs, labels)
            if args.steps_loss > 0 and step % args.steps_loss == 0:
                print(f"loss: {float(loss_sum) / n_it:0.4} (epoch: {epoch}, step: {step})",
                      "// Avg time/img: %.4f s" % ((time.time() - t0) / n_it / args.batch_size))
        avg_train = float(loss_sum) / max(n_it, 1)
        print("epoch took: ", time.time() - t0)
        iouTrain = float(iou_train.getIoU()[0]) if iou_train is not None else 0
        val_new_loss, val_new_acc, row = evaluate(epoch)
        if scalars is not None:
            add_scalars(writer, scalars(avg_train), epoch)
        current_acc = -val_new_loss if val_new_acc == 0 else val_new_acc
        is_best = current_acc > best_acc
        best_acc = max(current_acc, best_acc)
        if _rank() == 0:
            save_checkpoint({"epoch": epoch + 1, "arch": str(model),
                             "state_dict": _prefixed(model.state_dict()), "best_acc": best_acc,
                             "optimizer": optimizer.state_dict()}, is_best,
                            savedir + f"/checkpoint_{tag}.pth.tar", savedir + f"/model_best_{tag}.pth.tar")
            if is_best:
                with open(savedir + "/best.txt", "w") as f:
                    f.write("Best epoch is %d, with Val-IoU= %.4f" % (epoch, val_new_acc))
            log_row(log_path, epoch, avg_train, iouTrain, row, used_lr)
    close_writer(writer)
    return model


def train(args, finetune=False):
    global NUM_CLASSES
    print("old dataset: ", args.dataset_old)
    print("new dataset: ", args.dataset_new)
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    model = Net_ftp1(NUM_CLASSES_old, NUM_CLASSES_new)
    if args.state:
        saved = torch.load(args.state, map_location="cpu", weights_only=False)["state_dict"]
        new = {re.sub("decoder", "decoder_old", k): v for k, v in saved.items()}       # :218-220
        model.load_state_dict(_strip(new), strict=False)
        print("\nLOADED SAVED CITYSCAPES ENC -> ENC, DECODER --> OLD_DECODER for finetuning "
              "multi-head model on {}\n".format(args.dataset_new))
    model.to(dev)
    print("args.finetune: ", args.finetune)
    weight = class_weights(args.dataset_new).to(dev)
    criterion = CrossEntropyLoss2d(weight)
    criterion_old = CrossEntropyLoss2d(class_weights(args.dataset_old).to(dev))
    loader, val = make_loaders(args, [(args.dataset_old, NUM_CLASSES_old),
                                      (args.dataset_new, NUM_CLASSES_new)], 1)
    engine = FineTuneEngine(model, weight, finetune,
                            lambda x: model(x, decoder_old=False, decoder_new=True))
    print("finetuning optimizer" if finetune else "non-finetuning optimizer")
    if _rank() == 0:
        with open(f"../save/{args.savedir}/model.txt", "w") as f:
            f.write(str(model))

    def evaluate(epoch):
        print("----- VALIDATING - EPOCH", epoch, "--current---")
        ln, an = eval(model, val[args.dataset_new], criterion, NUM_CLASSES_new, epoch, task=1)
        print("----- VALIDATING - EPOCH", epoch, "--old----")
        lo, ao = eval(model, val[args.dataset_old], criterion_old, NUM_CLASSES_old, epoch, task=0)
        last["row"] = (ln, lo, an, ao)
        return ln, an, (ln, lo, an, ao)

    last = {}

    def scalars(avg_train):                                                     # :327-332
        ln, lo, an, ao = last["row"]
        return {"train_loss": avg_train,
                "val_loss_{}".format(args.dataset_new): ln, "val_accuracy_{}".format(args.dataset_new): an,
                "val_loss_{}".format(args.dataset_old): lo, "val_accuracy_{}".format(args.dataset_old): ao}

    def log_row(path, epoch, avg_train, iou_train, row, lr):
        with open(path, "a") as f:                                              # :359-361
            f.write("\n%d\t\t%.4f\t\t%.4f\t\t%.4f\t\t%.4f\t\t%.4f\t\t%.4f\t\t%.8f" % (
                epoch, avg_train, row[0], row[1], iou_train, row[2], row[3], lr))

    tag = "{}_{}_{}_{}".format(args.model, args.num_epochs, args.batch_size, args.model_name_suffix)
    return run_epochs(args, model, engine, loader, evaluate, tag, log_row, scalars)


def eval(model, dataset_loader, criterion, num_classes, epoch, task=1):
    """:365-409 -- task 1 = new decoder, task 0 = old decoder."""
    global NUM_CLASSES
    model.eval()
    _engine.broadcast_buffers(model)     # the model that is scored = the model rank 0 checkpoints
    dev = next(model.parameters()).device
    NUM_CLASSES = num_classes
    decoder_old, decoder_new = (False, True) if task == 1 else (True, False)
    print("num_classes: ", NUM_CLASSES, "decoder_old: ", decoder_old, "decoder_new: ", decoder_new)
    meter = iouEval(num_classes, num_classes - 1)
    loss_sum, n = torch.zeros((), device=dev), 0
    with torch.no_grad():
        for batch in dataset_loader:
            inputs, targets = to_device_batch(batch, dev, num_classes)
            outputs = model(inputs, decoder_old, decoder_new)
            loss_sum += criterion(outputs, targets[:, 0])
            n += 1
            meter.addBatch(outputs, targets)
    iou_val, _ = meter.getIoU()
    print("EPOCH IoU on VAL set: ", "{:0.2f}".format(float(iou_val) * 100), "%")
    return float(loss_sum) / max(n, 1), float(iou_val)


def _init_dist():
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1 and not _is_dist():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))


def main(args):
    global NUM_CLASSES_old, NUM_CLASSES_new
    NUM_CLASSES_old, NUM_CLASSES_new = args.num_classes_old, args.num_classes_new
    _init_dist()
    savedir = f"../save/{args.savedir}"
    if _rank() == 0:
        os.makedirs(savedir, exist_ok=True)
        with open(savedir + "/opts.txt", "w") as f:
            f.write(str(args))
    print("====== FINETUNING TRAINING OF NEW_DECODER & SHARED ENCODER ========")
    model = train(args, args.finetune)
    print("========== TRAINING FINISHED ===========")
    return model


def add_common_flags(p):
    p.add_argument("--port", type=int, default=8097)
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
    p.add_argument("--synthetic", type=int, default=0,
                   help="train on N seeded procedural images (MI355X build extension)")
    add_datadir_flags(p)


def build_parser():
    p = ArgumentParser()
    p.add_argument("--cuda", action="store_true", default=True)
    p.add_argument("--model", default="erfnet_ftp1")
    p.add_argument("--dataset-old", default="cityscapes")
    p.add_argument("--dataset-new", default="BDD")
    p.add_argument("--num-classes-old", type=int, default=20)
    p.add_argument("--num-classes-new", type=int, default=20)
    p.add_argument("--state")
    p.add_argument("--finetune", action="store_true")
    add_common_flags(p)
    p.add_argument("--model-name-suffix", default="Finetune-CStoBDD-final")
    return p


if __name__ == "__main__":
    main(build_parser().parse_args())
