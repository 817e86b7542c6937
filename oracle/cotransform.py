"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's input transform.

``co_transform`` follows ``MyCoTransform.__call__`` (train_new_task_step2.py:55-81) operation by
operation with PIL, the library the reference itself drives.  The reference reaches PIL through
``torchvision.transforms`` (not installed here, not vendored by the reference, no version pinned:
README.md:12-13).  Its published behaviour for the two calls on the path is restated:
  * ``Resize([h, w], interp)(img)``  == ``img.resize((w, h), interp)``  (torchvision functional_pil.resize
    for a 2-element size);
  * ``ToTensor()(img)`` for an 8-bit RGB PIL image == ``uint8 HWC -> CHW float32 .div(255)``.
``ToLabel`` / ``Relabel`` are transform.py:62-80.  The three random draws are taken by the caller
in the reference's order (``random.random()``, ``random.randint(-2, 2)`` twice, :62-69) so they
can be replayed on the device path.
"""
import random

import numpy as np
import torch
from PIL import Image, ImageOps


def draw_params(augment=True, rng=random):
    """-> (hflip, transX, transY) consuming the RNG exactly like :62-69 (no draw when not augmenting)."""
    if not augment:
        return 0, 0, 0
    hflip = rng.random()
    tx = rng.randint(-2, 2)
    ty = rng.randint(-2, 2)
    return int(hflip < 0.5), tx, ty


def resize_pair(image, target, height, width):
    """:56-57"""
    return (image.resize((width, height), Image.BILINEAR),
            target.resize((width, height), Image.NEAREST))


def co_transform(image, target, height, width, num_classes, params, augment=True):
    """PIL RGB image + PIL 'P' label -> (float32 [3,H,W] in [0,1], int64 [1,H,W]).
    ``params`` = (hflip, transX, transY) from ``draw_params``; ignored when ``augment`` is False."""
    flip, tx, ty = params if augment else (0, 0, 0)
    image, target = resize_pair(image, target, height, width)
    if augment:
        if flip:
            image = image.transpose(Image.FLIP_LEFT_RIGHT)
            target = target.transpose(Image.FLIP_LEFT_RIGHT)
        image = ImageOps.expand(image, border=(tx, ty, 0, 0), fill=0)             # :71
        target = ImageOps.expand(target, border=(tx, ty, 0, 0), fill=255)         # :72-73
        image = image.crop((0, 0, image.size[0] - tx, image.size[1] - ty))        # :74
        target = target.crop((0, 0, target.size[0] - tx, target.size[1] - ty))    # :75
    x = torch.from_numpy(np.array(image, dtype=np.uint8)).permute(2, 0, 1).float().div(255)
    y = torch.from_numpy(np.array(target)).long().unsqueeze(0)                    # ToLabel
    y[y == 255] = num_classes - 1                                                 # Relabel
    return x, y
