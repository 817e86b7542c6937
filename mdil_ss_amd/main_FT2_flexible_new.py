"""Fine-tuning / feature-extraction baseline, second increment (two old decoder heads + a new one)
on MI355X -- mirror of the reference's ``main_FT2_flexible_new.py``: flags (:455-490), checkpoint
loading with ``decoder_old -> decoder_old1`` / ``decoder_new -> decoder_old2`` remap (:203-212),
freeze rule and optimizers (:220-235), validation of every dataset at epoch 1 and every 10th
(:305-311), file names (:335-340).  The reference imports ``models.erfnet_ft2``, which does not
exist in its tree; the model it means is ``models/erfnet_ftp2.py``.  Hot loop:
``engine.FineTuneEngine``."""
import os
import re
from argparse import ArgumentParser

import torch

from . import main_ftp1_enc_newbn as F1
from .dataset import MyCoTransform, to_device_batch  # noqa: F401
from .engine import FineTuneEngine
from . import engine as _engine
from .iouEval import iouEval
from .models.erfnet import NetFT2 as Net_ft2
from .train_multi_task import DATASET_WEIGHTS
from .train_new_task_step2 import (CrossEntropyLoss2d, class_weights, save_checkpoint,  # noqa: F401
                                   _strip, _rank)

NUM_CLASSES = 27
DATASET_WEIGHTS = dict(DATASET_WEIGHTS, cityscapes="cityscapes")


def train(args, finetune=False):
    print("datasets: ", args.datasets)
    print("new dataset: ", args.dataset_new)
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    classes = args.num_classes
    model = Net_ft2(classes[0], classes[1], classes[2])
    if args.state:
        saved = torch.load(args.state, map_location="cpu", weights_only=False)["state_dict"]
        new = {}
        for k, v in saved.items():                                              # :205-211
            if "decoder_old" in k:
                k = re.sub("decoder_old", "decoder_old1", k)
            elif "decoder_new" in k:
                k = re.sub("decoder_new", "decoder_old2", k)
            new[k] = v
        model.load_state_dict(_strip(new), strict=False)
        print("\nLOADED SAVED CS-BDD ENC -> ENC, Dold->D1, Dnew->D2 for finetuning multi-head "
              "model on {}\n".format(args.dataset_new))
    model.to(dev)
    print("args.finetune: ", args.finetune)
    ce_loss = {d: CrossEntropyLoss2d(class_weights(DATASET_WEIGHTS[d]).to(dev)) for d in args.datasets}
    new_index = args.datasets.index(args.dataset_new)
    F1.NUM_CLASSES_new = classes[new_index]
    loader, val = F1.make_loaders(args, list(zip(args.datasets, classes)), new_index)
    engine = FineTuneEngine(model, ce_loss[args.dataset_new].weight, finetune,
                            lambda x: model(x, decoder_old1=False, decoder_old2=False, decoder_new=True))
    if _rank() == 0:
        with open(f"../save/{args.savedir}/model.txt", "w") as f:
            f.write(str(model))

    def evaluate(epoch):
        loss = {d: 0.0 for d in args.datasets}
        acc = {d: 0.0 for d in args.datasets}
        if epoch % 10 == 0 or epoch == 1:
            print("----- VALIDATING - EPOCH", epoch)
            for ind, d in enumerate(args.datasets):
                print("validate: ", d)
                loss[d], acc[d] = eval(model, val[d], ce_loss[d], classes[ind], epoch, ind)
        info = {}
        for d in args.datasets:
            info["val_acc_{}".format(d)] = acc[d]
            info["val_loss_{}".format(d)] = loss[d]
        print(info)
        last["info"] = info
        return loss[args.dataset_new], acc[args.dataset_new], None

    last = {}

    tag = "{}_{}_{}_{}".format(args.model, args.num_epochs, args.batch_size, args.model_name_suffix)
    return F1.run_epochs(args, model, engine, loader, evaluate, tag, lambda *a: None,
                         lambda avg_train: last["info"])                       # :313-322


def eval(model, dataset_loader, criterion, num_classes, epoch, task=2):
    """:362-420 -- task 2 = new decoder, 1 = decoder_old2, 0 = decoder_old1."""
    global NUM_CLASSES
    model.eval()
    _engine.broadcast_buffers(model)     # the model that is scored = the model rank 0 checkpoints
    dev = next(model.parameters()).device
    NUM_CLASSES = num_classes
    flags = {2: (False, False, True), 1: (False, True, False), 0: (True, False, False)}[task]
    print("num_classes: ", NUM_CLASSES, "decoder_old1: ", flags[0], "decoder_old2: ", flags[1],
          "decoder_new: ", flags[2])
    meter = iouEval(num_classes, num_classes - 1)
    loss_sum, n = torch.zeros((), device=dev), 0
    with torch.no_grad():
        for batch in dataset_loader:
            inputs, targets = to_device_batch(batch, dev, num_classes)
            outputs = model(inputs, *flags)
            loss_sum += criterion(outputs, targets[:, 0])
            n += 1
            meter.addBatch(outputs, targets)
    iou_val, _ = meter.getIoU()
    print("EPOCH IoU on VAL set: ", "{:0.2f}".format(float(iou_val) * 100), "%")
    return float(loss_sum) / max(n, 1), float(iou_val)


def main(args):
    F1._init_dist()
    savedir = f"../save/{args.savedir}"
    if _rank() == 0:
        os.makedirs(savedir, exist_ok=True)
        with open(savedir + "/opts.txt", "w") as f:
            f.write(str(args))
    print("====== FINETUNING TRAINING OF NEW_DECODER & SHARED ENCODER ========")
    model = train(args, args.finetune)
    print("========== TRAINING FINISHED ===========")
    return model


def build_parser():
    p = ArgumentParser()
    p.add_argument("--cuda", action="store_true", default=True)
    p.add_argument("--model", default="erfnet_ftp2")
    p.add_argument("--dataset-new", default="IDD")
    p.add_argument("--datasets", nargs="+", required=True, default=["IDD", "CS", "BDD"],
                   help="pass list of datasets in order")
    p.add_argument("--current_task", type=int, default=2)
    p.add_argument("--nb_tasks", type=int, default=3)
    p.add_argument("--num-classes", type=int, nargs="+", required=True, default=[20, 20, 27])
    p.add_argument("--state")
    p.add_argument("--finetune", action="store_true")
    p.add_argument("--datadir", default=os.getenv("HOME", "") + "/datasets/cityscapes/")
    F1.add_common_flags(p)
    p.add_argument("--model-name-suffix", default="FE-CSBDDtoIDD-oldencBN")
    return p


if __name__ == "__main__":
    main(build_parser().parse_args())
