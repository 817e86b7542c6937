"""Step-2 incremental trainer (proposed method: RAP + domain-adaptive KD) on MI355X.

Mirrors the entry points, flags, file outputs and state-dict conventions of the reference's
``train_new_task_step2.py`` (prachigarg23/MDIL-SS): ``CrossEntropyLoss2d``, ``is_shared``,
``is_DS_curr``, ``train``, ``eval``, ``save_checkpoint``, ``main`` and the CLI of :541-587.
Underneath, the hot loop (:273-313) is ``engine.Step2Engine`` -- HIP kernels, one process per
GPU, RCCL gradient all-reduce -- instead of nn.DataParallel over ATen/cuDNN.

Differences that are deliberate (and flagged):
  * ``--synthetic N`` trains on the seeded procedural dataset (no dataset ships offline);
    ``--datadir`` style real loaders are a later row.
  * the per-iteration ``.item()`` x3 and ``torch.cuda.empty_cache()`` (:308-313) are replaced by
    a host read every ``--steps-loss`` iterations.
  * checkpoints keep the DataParallel ``module.`` key prefix so they interchange with the
    reference's (:441-446, :483-530).
"""
import os
import re
import time
from argparse import ArgumentParser

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader

from . import ops
from .dataset import (MyCoTransform, ProceduralSeg, add_datadir_flags,  # noqa: F401
                      open_dataset, to_device_batch)
from .engine import Step2Engine, poly_factor
from . import engine as _engine
from .iouEval import iouEval
from .models.erfnet_RA_parallel import Net as Net_RAP

NUM_CLASSES = 20
current_task = 0   # module global read by is_DS_curr, like the reference (:45,99-105)

# class weights hard-coded by the reference (:121-131), copied as data
WEIGHTS = {
    "IDD": [3.235635601598852, 6.76221624390441, 9.458242359884549, 9.446818215454014,
            9.947040673126763, 9.789672819856547, 9.476665808564432, 10.465565126694731,
            9.59189547383129, 7.637805282159825, 8.990899026692638, 9.26222234098628,
            10.265657138809514, 9.386517631614392, 8.357391489170013, 9.910382864314824,
            10.389977663948363, 8.997422571963602, 10.418070541191673, 10.483262606962834,
            9.511436923349441, 7.597725385711079, 6.1734896019878205, 9.787631041755187,
            3.9178330193378708, 4.417448652936843, 10.313160683418731],
    "BDD": [3.6525147483016243, 8.799815287822142, 4.781908267406055, 10.034828238618045,
            9.5567865464289, 9.645099012085169, 10.315292989325766, 10.163473632969513,
            4.791692009441432, 9.556915153488912, 4.142994047786311, 10.246903827488143,
            10.47145010979545, 6.006704177894196, 9.60620532303246, 9.964959813857726,
            10.478333987902301, 10.468010534454706, 10.440929141422366, 3.960822533003462],
    "cityscapes": [2.8159904084894922, 6.9874672455551075, 3.7901719017455604, 9.94305485286704,
                   9.77037625072462, 9.511470001589007, 10.310780572569994, 10.025305236316246,
                   4.6341256102158805, 9.561389195953845, 7.869695292372276, 9.518873463871952,
                   10.374050047877898, 6.662394711556909, 10.26054487392723, 10.28786101490449,
                   10.289883605859952, 10.405463349170795, 10.138502340710136, 5.131658171724055],
}


def class_weights(name):
    w = torch.tensor(WEIGHTS[name], dtype=torch.float32)
    w[-1] = 0            # ignore class carries zero weight (:133-135)
    return w


class CrossEntropyLoss2d(torch.nn.Module):
    """NLLLoss2d(weight)(log_softmax(outputs, 1), targets) (:84-92) as one fused HIP kernel."""

    def __init__(self, weight=None):
        super().__init__()
        self.weight = weight

    def forward(self, outputs, targets):
        w = self.weight
        if w is None:
            w = torch.ones(outputs.shape[1], device=outputs.device)
        return ops.cross_entropy2d(outputs, targets, w.to(outputs.device))


def is_shared(n):
    return "encoder" in n and "parallel_conv" not in n and "bn" not in n


def is_DS_curr(n):
    t = current_task
    if "decoder.{}".format(t) in n:
        return True
    if "encoder" in n and ("bn" in n or "parallel_conv" in n):
        return ".{}.weight".format(t) in n or ".{}.bias".format(t) in n
    return False


def apply_step2_freeze(model, model_old, t):
    """Freeze rule of :202-215: the whole old model; in the student every decoder but ``t`` and
    every encoder bn / parallel_conv that is not domain ``t``'s weight / bias."""
    for p in model_old.parameters():
        p.requires_grad = False
    for name, p in model.named_parameters():
        if "decoder" in name:
            if "decoder.{}".format(t) not in name:
                p.requires_grad = False
        elif "encoder" in name and ("bn" in name or "parallel_conv" in name):
            if not (".{}.weight".format(t) in name or ".{}.bias".format(t) in name):
                p.requires_grad = False


def student_init_dict(saved, student_keys, t):
    """Initialisation of the step-``t`` student from the step-(t-1) checkpoint (:497-530):
    common keys as they are; encoder DS(t-1) weight/bias -> DS(t) (running stats are NOT
    copied); decoder(t-1) -> decoder(t) except output_conv."""
    new = {k: v for k, v in saved.items() if k in student_keys}
    prev_w, prev_b = ".{}.weight".format(t - 1), ".{}.bias".format(t - 1)
    for k, v in saved.items():
        if "encoder" in k:
            if "parallel_conv" in k or "bn" in k:
                if prev_w in k:
                    new[re.sub(prev_w, ".{}.weight".format(t), k)] = v
                elif prev_b in k:
                    new[re.sub(prev_b, ".{}.bias".format(t), k)] = v
        elif "decoder" in k and "output_conv" not in k:
            new[re.sub("decoder.{}".format(t - 1), "decoder.{}".format(t), k)] = v
    return new


def _strip(sd):
    return {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}


def _prefixed(sd):
    return {"module." + k: v for k, v in sd.items()}


def _is_dist():
    return dist.is_available() and dist.is_initialized()


def _rank():
    return dist.get_rank() if _is_dist() else 0


def make_loaders(args):
    n_cls = args.num_classes[args.current_task]
    # the old-domain validation set is scored as TASK 0 (eval(..., 0, ...) below, reference :343-347):
    # its ignore label is relabelled to THAT head's last class -- whatever current_task is
    n_old = args.num_classes[0]
    world = dist.get_world_size() if _is_dist() else 1
    dom, dom_old = args.current_task, max(args.current_task - 1, 0)
    if args.synthetic:
        tr = ProceduralSeg(args.synthetic, args.height, args.width, n_cls, seed=11, domain=dom)
        va = ProceduralSeg(max(args.synthetic // 4, args.batch_size), args.height, args.width, n_cls,
                           seed=12, domain=dom)
        vo = ProceduralSeg(max(args.synthetic // 4, args.batch_size), args.height, args.width, n_old,
                           seed=13, domain=dom_old)
    else:                                   # reference :136-181
        tr = open_dataset(args.dataset, "train", args, augment=True)
        va = open_dataset(args.dataset, "val", args, augment=False)
        old = getattr(args, "dataset_old", None)            # the step-1 trainer has no old dataset
        vo = open_dataset(old, "val", args, augment=False) if old else va
    per_rank = args.batch_size
    if world > 1 and getattr(args, "dp_global_batch", False):
        # nn.DataParallel semantics: --batch-size is the GLOBAL batch, scattered over the GPUs
        assert args.batch_size % world == 0, "--dp-global-batch needs --batch-size divisible by the world size"
        per_rank = args.batch_size // world
    if getattr(args, "cache_device", False) and not args.synthetic:
        # --cache-resized DIR --cache-device: the splits' post-Resize bytes live in HBM; an epoch is
        # a permutation + three draws per sample on the host, a gather + the augment kernel on the GPU
        if not getattr(args, "cache_resized", None):
            raise RuntimeError("--cache-device needs --cache-resized DIR")
        from .dataset import DeviceResizedCache
        dev = torch.device("cuda", torch.cuda.current_device())
        caches = {}

        def resident(ds):
            if id(ds) not in caches:
                caches[id(ds)] = DeviceResizedCache(ds, dev, args.num_workers)
            return caches[id(ds)]
        return (resident(tr).loader(per_rank, n_cls, True, world > 1, _rank(), world),
                resident(va).loader(args.batch_size, n_cls, False, False, _rank(), world),
                resident(vo).loader(args.batch_size, n_old if vo is not va else n_cls, False, False, _rank(), world))
    sampler = None
    if world > 1:
        # every rank must run the same number of iterations (one gradient exchange each): the
        # sampler pads the shuffled index list to a multiple of the world size
        sampler = torch.utils.data.distributed.DistributedSampler(tr, shuffle=True, seed=0)
        # validation is sharded without padding (rank r takes images r, r+world, ...): the counts
        # are summed over the ranks in eval(), so every image is scored exactly once
        va = torch.utils.data.Subset(va, range(_rank(), len(va), world))
        vo = torch.utils.data.Subset(vo, range(_rank(), len(vo), world))
    # the last, smaller batch of an epoch is trained on, as in the reference (:150-152: no
    # drop_last); under data parallelism it is dropped so that ranks stay in step
    loader = DataLoader(tr, num_workers=args.num_workers, batch_size=per_rank,
                        shuffle=sampler is None, sampler=sampler, drop_last=world > 1)
    loader_val = DataLoader(va, num_workers=args.num_workers, batch_size=args.batch_size)
    loader_val_old = DataLoader(vo, num_workers=args.num_workers, batch_size=args.batch_size)
    return loader, loader_val, loader_val_old


def train(args, model, model_old):
    global NUM_CLASSES
    NUM_CLASSES = args.num_classes[args.current_task]
    dev = next(model.parameters()).device
    savedir = f"../save/{args.savedir}"
    weight = class_weights(args.dataset).to(dev)
    weight_old = class_weights(args.dataset_old).to(dev)
    criterion = CrossEntropyLoss2d(weight)
    criterion_old = CrossEntropyLoss2d(weight_old)
    loader, loader_val, loader_val_old = make_loaders(args)

    apply_step2_freeze(model, model_old, current_task)
    log_path = savedir + "/automated_log.txt"
    if _rank() == 0:
        if not os.path.exists(log_path):
            with open(log_path, "a") as f:
                f.write("Epoch\t\tTrain-loss\t\tTest-loss\t\tTrain-IoU\t\tTest-IoU\t\tlearningRate")
        with open(savedir + "/model.txt", "w") as f:
            f.write(str(model))

    engine = Step2Engine(model, model_old, weight, current_task=current_task,
                         lambdac=args.lambdac, is_shared=is_shared, is_ds_curr=is_DS_curr,
                         global_ce=getattr(args, "dp_global_batch", False))
    engine.want_logits = bool(args.iouTrain)     # only --iouTrain reads the training logits (:317-320)
    optimizer = engine.optimizer
    best_acc = 0
    tag = "{}_{}_{}_{}{}_step{}".format(args.dataset, args.model, args.num_epochs, args.batch_size,
                                        args.model_name_suffix, len(args.num_classes))
    from .scalar_log import add_scalars, close_writer, open_writer
    writer = open_writer("Adaptations/runs_" + tag, _rank())     # :115-117: SummaryWriter('Adaptations/runs_...')
    for epoch in range(1, args.num_epochs + 1):
        print("-----TRAINING - EPOCH---", epoch, "-----")
        optimizer.set_epoch(epoch, args.num_epochs)      # LambdaLR.step(epoch), :244-254
        used_lr = 0
        for g in optimizer.param_groups:
            print("LEARNING RATE: ", g["lr"])
            used_lr = float(g["lr"])
        if hasattr(loader.sampler, "set_epoch"):
            loader.sampler.set_epoch(epoch)
        sums = torch.zeros(3, device=dev)
        n_it = 0
        t_epoch = time.time()
        iou_train = iouEval(NUM_CLASSES, NUM_CLASSES - 1) if args.iouTrain else None
        for step, batch in enumerate(loader):
            images, labels = to_device_batch(batch, dev, NUM_CLASSES)
            total, ce, kld = engine.iteration(images, labels)
            sums += torch.stack([total, ce, kld])
            n_it += 1
            if iou_train is not None:                              # :317-320
                iou_train.addBatch(engine.last_outputs, labels)
            if args.steps_loss > 0 and step % args.steps_loss == 0:
                avg = float(sums[0]) / n_it                     # the only host sync in the loop
                ops.check_labels()      # raises like torch's device assert if a label was out of range
                dt = (time.time() - t_epoch) / n_it / args.batch_size
                print(f"loss: {avg:0.4} (epoch: {epoch}, step: {step})",
                      "// Avg time/img: %.4f s" % dt)
        avg_total, avg_ce, avg_kld = (sums / max(n_it, 1)).tolist()
        print("epoch took: ", time.time() - t_epoch)
        iouTrain = 0
        if iou_train is not None:                                  # :329-333
            iouTrain = float(iou_train.getIoU()[0])
            print("EPOCH IoU on TRAIN set: ", "{:0.2f}".format(iouTrain * 100), "%")

        print("----- VALIDATING - EPOCH", epoch, "-----")
        loss_val, val_acc = eval(model, loader_val, criterion, current_task, args.num_classes, epoch)
        loss_val_old, val_acc_old = eval(model, loader_val_old, criterion_old, 0, args.num_classes,
                                         epoch)
        print("old-task loss and acc: ", loss_val_old, val_acc_old)
        add_scalars(writer, {"total_train_loss": avg_total, "KLD_loss_train": avg_kld, "ce_loss_train": avg_ce,
                             f"val_loss_{args.dataset}": loss_val, f"val_acc_{args.dataset}": val_acc,
                             f"val_loss_{args.dataset_old}": loss_val_old,
                             f"val_acc_{args.dataset_old}": val_acc_old}, epoch)   # :351-355: epoch-wise scalars

        current_acc = -loss_val if val_acc == 0 else val_acc
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
                    f.write("Best epoch is %d, with Val-IoU= %.4f" % (epoch, val_acc))
            with open(log_path, "a") as f:
                f.write("\n%d\t\t%.4f\t\t%.4f\t\t%.4f\t\t%.4f\t\t%.8f" % (
                    epoch, avg_total, loss_val, iouTrain, val_acc, used_lr))
    close_writer(writer)
    return model


def eval(model, dataset_loader, criterion, task, num_classes, epoch):
    """Validation pass (:398-438): eval-mode forward, CE, fused argmax + confusion counts."""
    global NUM_CLASSES
    model.eval()
    _engine.broadcast_buffers(model)     # the model that is scored = the model rank 0 checkpoints
    dev = next(model.parameters()).device
    num_cls = num_classes[task]
    NUM_CLASSES = num_cls
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
    if _is_dist() and dist.get_world_size() > 1:
        # validation images are sharded over the ranks (make_loaders): sum the confusion counts
        # and the loss over the shards -> the metric of the whole validation set on every rank
        if meter.counts is None:
            meter.counts = torch.zeros(3, num_cls, dtype=torch.int64, device=dev)
        dist.all_reduce(meter.counts, op=dist.ReduceOp.SUM)
        ln = torch.stack([loss_sum.double(), torch.tensor(float(n), dtype=torch.float64, device=dev)])
        dist.all_reduce(ln, op=dist.ReduceOp.SUM)
        loss_sum, n = ln[0], int(ln[1].item())
    iou_val, _ = meter.getIoU()
    avg = float(loss_sum) / max(n, 1)
    ops.check_labels()      # raises like torch's device assert if a label was out of range
    print("EPOCH IoU on VAL set: ", "{:0.2f}".format(float(iou_val) * 100), "%")
    return avg, float(iou_val)


def save_checkpoint(state, is_best, filenameCheckpoint, filenameBest):
    torch.save(state, filenameCheckpoint)
    print("Saving model: ", filenameCheckpoint)
    if is_best:
        print("Saving model as best: ", filenameBest)
        torch.save(state, filenameBest)


def main(args):
    global current_task
    current_task = args.current_task
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
    model = Net_RAP(args.num_classes, args.nb_tasks, args.current_task)
    model_old = Net_RAP(args.num_classes_old, args.nb_tasks - 1, args.current_task - 1)
    if args.state:
        saved = torch.load(args.state, map_location="cpu")["state_dict"]
        model_old.load_state_dict(_strip(saved), strict=False)
        print("loading previous step weights - {}-RAPs and shared weights from previous step."
              .format(args.dataset_old))
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
    p.add_argument("--dataset", default="cityscapes")
    p.add_argument("--dataset_old", default="IDD")
    p.add_argument("--num-classes", type=int, nargs="+", required=True, default=[20])
    p.add_argument("--num-classes-old", type=int, nargs="+", required=True, default=[20])
    p.add_argument("--nb_tasks", type=int, default=1)
    p.add_argument("--current_task", type=int, default=0)
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
    p.add_argument("--dp-global-batch", action="store_true",
                   help="data parallel: treat --batch-size as the GLOBAL batch (scattered over the "
                        "GPUs like nn.DataParallel, BN over batch-size/world images per GPU) and take "
                        "the cross entropy as one weighted mean over the whole batch; default: "
                        "--batch-size images per GPU (BASELINE: 'batch 6/GPU'), rank-mean of per-shard "
                        "weighted means")
    p.add_argument("--synthetic", type=int, default=0,
                   help="train on N seeded procedural images (MI355X build extension)")
    add_datadir_flags(p)
    return p


if __name__ == "__main__":
    main(build_parser().parse_args())
