"""Checkpoint evaluation on MI355X -- the proposed-model cells of the reference's
``Evaluation_Notebook.ipynb`` (cells 4-11) as a script: build ``Net_RAP(num_classes, nb_tasks,
nb_tasks-1)``, load a (``module.``-prefixed or plain) checkpoint strictly, and report per-class
IoU and mIoU of every task on its validation set through the fused argmax + confusion kernel.

    python -m mdil_ss_amd.evaluate --state model_best_....pth.tar --num-classes 20 20 27 \
        --datasets cityscapes BDD IDD [--cs-datadir ... --bdd-datadir ... --idd-datadir ...]

``eval(model, loader, criterion, task, num_classes) -> (iou_classes, iouVal)`` keeps the notebook's
signature and return order (cell 5).  ``--synthetic N`` evaluates on the procedural dataset."""
import json
from argparse import ArgumentParser

import torch
from torch.utils.data import DataLoader

from .dataset import ProceduralSeg, add_datadir_flags, open_dataset, to_device_batch
from .iouEval import iouEval
from .models.erfnet_RA_parallel import Net as Net_RAP
from .train_new_task_step2 import CrossEntropyLoss2d, class_weights, _strip

WEIGHT_NAME = {"cityscapes": "cityscapes", "CS": "cityscapes", "BDD": "BDD", "IDD": "IDD"}


def eval(model, dataset_loader, criterion, task, num_classes):
    model.eval()
    dev = next(model.parameters()).device
    num_cls = num_classes[task]
    meter = iouEval(num_cls, num_cls - 1)
    loss_sum, n = torch.zeros((), device=dev), 0
    with torch.no_grad():
        for batch in dataset_loader:
            inputs, targets = to_device_batch(batch, dev, num_cls)
            outputs = model(inputs, task)
            loss_sum += criterion(outputs, targets[:, 0])
            n += 1
            meter.addBatch(outputs, targets)
    iou_val, iou_classes = meter.getIoU()
    eval.last_loss = float(loss_sum) / max(n, 1)
    return iou_classes, float(iou_val)


def main(args):
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    nb = len(args.num_classes)
    assert len(args.datasets) == nb, "--datasets and --num-classes must list the same tasks"
    model = Net_RAP(args.num_classes, nb, nb - 1)
    if args.state:
        saved = torch.load(args.state, map_location="cpu", weights_only=False)
        model.load_state_dict(_strip(saved["state_dict"]), strict=True)
    model.to(dev)
    report = {}
    for task, name in enumerate(args.datasets):
        if args.synthetic:
            ds = ProceduralSeg(args.synthetic, args.height, args.width, args.num_classes[task],
                               seed=12 + task, domain=task)
        else:
            ds = open_dataset(name, "val", args, augment=False)
        loader = DataLoader(ds, num_workers=args.num_workers, batch_size=args.batch_size)
        criterion = CrossEntropyLoss2d(class_weights(WEIGHT_NAME[name]).to(dev))
        iou_classes, miou = eval(model, loader, criterion, task, args.num_classes)
        report[name] = {"task": task, "mIoU": miou, "val_loss": eval.last_loss,
                        "iou_classes": [float(v) for v in iou_classes]}
        print(f"{name} (task {task}): mIoU {miou * 100:.2f} %  val-loss {eval.last_loss:.4f}")
    if args.json:
        with open(args.json, "w") as f:
            json.dump(report, f, indent=1)
    return report


def build_parser():
    p = ArgumentParser()
    p.add_argument("--state", help="checkpoint written by the trainers (or by the reference)")
    p.add_argument("--num-classes", type=int, nargs="+", required=True)
    p.add_argument("--datasets", nargs="+", required=True)
    p.add_argument("--height", type=int, default=512)
    p.add_argument("--width", type=int, default=1024)
    p.add_argument("--batch-size", type=int, default=6)
    p.add_argument("--num-workers", type=int, default=4)
    p.add_argument("--synthetic", type=int, default=0)
    p.add_argument("--json", help="write the report here")
    add_datadir_flags(p)
    return p


if __name__ == "__main__":
    main(build_parser().parse_args())
