"""Evaluation-time datasets with the surface of the reference's ``dataset_custom.py``
(``cityscapes`` :75-117, ``IDD`` :122-164, ``BDD`` :167-209): separate input / target transforms
and ``(image, label, filename, filenameGt)`` items, as the evaluation and t-SNE notebooks use
them.  File discovery is shared with ``dataset.py``."""
import os

from .dataset import (_walk, is_image, is_label_BDD, is_label_IDD, is_label_city,  # noqa: F401
                      load_image)
import torch


class _Named(torch.utils.data.Dataset):
    def __init__(self, input_transform=None, target_transform=None):
        self.input_transform, self.target_transform = input_transform, target_transform

    def __getitem__(self, index):
        filename, filenameGt = self.filenames[index], self.filenamesGt[index]
        with open(os.path.join(self.images_root, filename), "rb") as f:
            image = load_image(f).convert("RGB")
        with open(os.path.join(self.labels_root, filenameGt), "rb") as f:
            label = load_image(f).convert("P")
        if self.input_transform is not None:
            image = self.input_transform(image)
        if self.target_transform is not None:
            label = self.target_transform(label)
        return image, label, filename, filenameGt

    def __len__(self):
        return len(self.filenames)


class cityscapes(_Named):
    label_filter = staticmethod(is_label_city)

    def __init__(self, root, input_transform=None, target_transform=None, subset="train"):
        super().__init__(input_transform, target_transform)
        self.images_root = os.path.join(root, "leftImg8bit/") + subset
        self.labels_root = os.path.join(root, "gtFine/") + subset
        print(self.images_root)
        self.filenames = _walk(self.images_root, is_image)
        self.filenamesGt = _walk(self.labels_root, self.label_filter)


class IDD(cityscapes):
    label_filter = staticmethod(is_label_IDD)


class BDD(_Named):
    def __init__(self, root, input_transform=None, target_transform=None, subset="train"):
        super().__init__(input_transform, target_transform)
        self.images_root = os.path.join(root, "images/") + subset
        self.labels_root = os.path.join(root, "labels/") + subset
        print(self.images_root)
        self.filenames = sorted(f for f in os.listdir(self.images_root) if is_image(f))
        self.filenamesGt = sorted(f for f in os.listdir(self.labels_root) if is_label_BDD(f))
