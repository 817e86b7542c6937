"""Step-1 trainer (first domain, RAP-FT model) on MI355X: mirror of the reference's
``train_RAPFT_step1.py`` for ``--model erfnet_RA_parallel`` -- same entry points (``train``, ``eval``,
``save_checkpoint``, ``main``), flags (:513-549), freeze rule (:177-190), single-group Adam + poly
LR (:260-272), checkpoint dict / file names, ``module.``-prefixed keys and the ImageNet-encoder key
remap (:482-491).  The hot loop is ``engine.Step1Engine`` (HIP kernels, one process per GPU,
RCCL gradient all-reduce).  The other ablation models of the reference trainer
(``erfnet_RA_series`` / ``erfnet_RCM`` / ``erfnet_bn`` / ``erfnet_onlyRAP``, :21-26) are not in this
repository and are refused explicitly."""
import os
import re
import time
from argparse import ArgumentParser

import torch
import torch.distributed as dist

from .dataset import MyCoTransform, add_datadir_flags, to_device_batch  # noqa: F401
from .engine import Step1Engine
from .models.erfnet_RA_parallel import Net as Net_RAP
from . import train_new_task_step2 as S2
from .train_new_task_step2 import (CrossEntropyLoss2d, class_weights, save_checkpoint,  # noqa: F401
                                   _prefixed, _strip, _rank, _is_dist)

NUM_CLASSES = 20


def apply_step1_freeze(model, current_task):
    """:177-190 -- only decoder ``t`` and the encoder's domain-``t`` bn / parallel_conv weight+bias
    train among the domain-specific parameters; shared encoder convs always train."""
    for name, p in model.named_parameters():
        if "decoder" in name:
            p.requires_grad = "decoder.{}".format(current_task) in name
        elif "encoder" in name and ("bn" in name or "parallel_conv" in name):
            p.requires_grad = (".{}.weight".format(current_task) in name or
                               ".{}.bias".format(current_task) in name)


def eval(model, dataset_loader, criterion, task, num_classes, epoch):
    return S2.eval(model, dataset_loader, criterion, task, num_classes, epoch)


def train(args, model):
    global NUM_CLASSES
    t = args.current_task
    NUM_CLASSES = args.num_classes[t]
    dev = next(model.parameters()).device
    savedir = f"../save/{args.savedir}"
    weight = class_weights(args.dataset).to(dev)
    criterion = CrossEntropyLoss2d(weight)
    S2.current_task = t
    a2 = type("A", (), dict(vars(args)))()
    loader, loader_val, _ = S2.make_loaders(a2)
    apply_step1_freeze(model, t)
    log_path = savedir + "/automated_log.txt"
    if _rank() == 0:
        if not os.path.exists(log_path):
            with open(log_path, "a") as f:
                f.write("Epoch\t\tTrain-loss\t\tTest-loss\t\tTrain-IoU\t\tTest-IoU\t\tlearningRate")
        with open(savedir + "/model.txt", "w") as f:
            f.write(str(model))
    engine = Step1Engine(model, weight, current_task=t)
    optimizer = engine.optimizer
    best_acc = 0
    tag = "{}_{}_{}_{}{}_step{}".format(args.dataset, args.model, args.num_epochs, args.batch_size,
                                        args.model_name_suffix, len(args.num_classes))
    from .scalar_log import add_scalars, close_writer, open_writer
    writer = open_writer("Adaptations/runs_" + tag, _rank())          # :107-109
    for epoch in range(1, args.num_epochs + 1):
        print("----- TRAINING - EPOCH", epoch, "-----")
        optimizer.set_epoch(epoch, args.num_epochs)
        used_lr = float(optimizer.param_groups[0]["lr"])
        print("LEARNING RATE: ", used_lr)
        if hasattr(loader.sampler, "set_epoch"):
            loader.sampler.set_epoch(epoch)
        loss_sum = torch.zeros((), device=dev)
        n_it, t0 = 0, time.time()
        for step, batch in enumerate(loader):
            loss_sum += engine.iteration(*to_device_batch(batch, dev, NUM_CLASSES))
            n_it += 1
            if args.steps_loss > 0 and step % args.steps_loss == 0:
                print(f"loss: {float(loss_sum) / n_it:0.4} (epoch: {epoch}, step: {step})",
                      "// Avg time/img: %.4f s" % ((time.time() - t0) / n_it / args.batch_size))
        avg_train = float(loss_sum) / max(n_it, 1)
        print("----- VALIDATING - EPOCH", epoch, "-----")
        loss_val, val_acc = eval(model, loader_val, criterion, t, args.num_classes, epoch)
        add_scalars(writer, {"train_loss": avg_train, f"val_loss_{args.dataset}": loss_val,
                             f"val_acc_{args.dataset}": val_acc}, epoch)         # :340-344
        current_acc = -loss_val if val_acc == 0 else val_acc
        is_best = current_acc > best_acc
        best_acc = max(current_acc, best_acc)
        if _rank() == 0:
            save_checkpoint({"epoch": epoch + 1, "arch": str(model),
                             "state_dict": _prefixed(model.state_dict()), "best_acc": best_acc,
                             "optimizer": optimizer.state_dict()}, is_best,
                            savedir + f"/checkpoint_{tag}.pth.tar", savedir + f"/model_best_{tag}.pth.tar")
            if is_best:
                with open(savedir + "/best.txt", "w") as f:
                    f.write("Best epoch is %d, with Val-IoU= %.4f" % (epoch, val_acc))
            with open(log_path, "a") as f:
                f.write("\n%d\t\t%.4f\t\t%.4f\t\t%.4f\t\t%.4f\t\t%.8f" % (
                    epoch, avg_train, loss_val, 0, val_acc, used_lr))
    close_writer(writer)
    return model


def main(args):
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
    if args.model != "erfnet_RA_parallel":
        raise SystemExit(f"model '{args.model}' is not part of the MI355X build (only erfnet_RA_parallel)")
    model = Net_RAP(args.num_classes, args.nb_tasks, args.current_task)
    if args.state:
        saved = torch.load(args.state, map_location="cpu")["state_dict"]
        if args.current_task == 0:
            print("loading ImageNet pre-trained enc")        # :482-491
            saved = {re.sub("module.features", "module", k): v for k, v in saved.items()}
        else:
            print("loading previous step weights")
        model.load_state_dict(_strip(saved), strict=False)
        print("loaded model from checkpoint provided.")
    model.to(dev)
    model = train(args, model)
    print("========== TRAINING FINISHED ===========")
    return model


def build_parser():
    p = ArgumentParser()
    p.add_argument("--cuda", action="store_true", default=True)
    p.add_argument("--model", default="erfnet_RA_parallel")
    p.add_argument("--dataset", default="cityscapes")
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
