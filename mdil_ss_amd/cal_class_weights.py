"""Class-weight calculator with the surface of the reference's ``cal_class_weights.py``
(``calc_weights(args, enc=False)``, :21-70): pixel counts of the training labels (255 counted as
the last class), ``w_c = 1 / ln(p_c + 1.1)`` (1.2 for the encoder-only variant), last class
zeroed.  Labels are decoded with PIL (the reference uses ``cv2.imread(file, 0)``; single-channel
id PNGs read the same) and counted with one ``bincount`` per file."""
import os
from argparse import ArgumentParser

import numpy as np
from PIL import Image

from .dataset import _walk, is_label_BDD, is_label_IDD, is_label_city  # noqa: F401


def calc_weights(args, enc=False):
    datapath, dataset, num_classes = args.datadir, args.dataset, args.num_classes
    if dataset in ("cityscapes", "IDD"):
        datapath = os.path.join(datapath, "gtFine/train/")
        files = _walk(datapath, is_label_IDD if dataset == "IDD" else is_label_city)
    elif dataset == "BDD":
        datapath = os.path.join(datapath, "labels/train/")
        files = [os.path.join(datapath, f) for f in os.listdir(datapath)]       # every file, :47-48
    else:
        raise ValueError(dataset)
    print("calculating weights for {} with {} classes, located in root dir: {}".format(
        dataset, num_classes, datapath))
    counts = np.zeros(num_classes)
    for file in files:
        try:
            label = np.array(Image.open(file).convert("L"))
        except OSError:
            continue                                                            # cv2 returns None, :53
        c = np.bincount(label.reshape(-1), minlength=256)
        counts[num_classes - 1] += c[255]
        counts[:num_classes] += c[:num_classes]
        if c[num_classes:255].any():
            raise IndexError("label id outside [0, num_classes) and != 255 in " + file)
    counts += 1
    prob = counts / counts.sum() + (1.2 if enc else 1.1)
    weight = np.reciprocal(np.log(prob))
    weight[num_classes - 1] = 0
    return weight


if __name__ == "__main__":
    p = ArgumentParser()
    p.add_argument("--datadir", required=True)
    p.add_argument("--dataset", default="cityscapes")
    p.add_argument("--num-classes", type=int, default=20)
    print(list(calc_weights(p.parse_args())))
