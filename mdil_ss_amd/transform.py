"""Label / colour helpers with the surface of the reference's ``transform.py``: ``Relabel``,
``ToLabel``, ``Colorize``, ``colormap_cityscapes``, ``colormap`` (:7-105).  Used by the host half
of the input pipeline and by visualisation; the training path relabels on the device
(``ops.augment_batch``)."""
import numpy as np
import torch

# transform.py:8-44, as data: train-id -> RGB (19 Cityscapes classes + IDD level-3 extras)
_CITYSCAPES_RGB = [
    (128, 64, 128), (244, 35, 232), (70, 70, 70), (102, 102, 156), (190, 153, 153),
    (153, 153, 153), (250, 170, 30), (220, 220, 0), (107, 142, 35), (152, 251, 152),
    (70, 130, 180), (220, 20, 60), (255, 0, 0), (0, 0, 142), (0, 0, 70), (0, 60, 100),
    (0, 80, 100), (0, 0, 230), (119, 11, 32), (229, 23, 142), (156, 60, 200), (99, 250, 80),
    (82, 92, 214), (196, 209, 152), (180, 165, 180), (37, 58, 77), (11, 35, 88), (150, 100, 100),
    (255, 255, 255)]


def colormap_cityscapes(n):
    cmap = np.zeros([n, 3], dtype=np.uint8)
    k = min(n, len(_CITYSCAPES_RGB))
    cmap[:k] = np.array(_CITYSCAPES_RGB[:k], dtype=np.uint8)
    return cmap


def colormap(n):
    """PASCAL-VOC bit-interleaved colour map (transform.py:47-60)."""
    i = np.arange(n, dtype=np.int64)
    rgb = np.zeros((n, 3), dtype=np.int64)
    for j in range(8):
        for c in range(3):
            rgb[:, c] += (1 << (7 - j)) * ((i >> (3 * j + c)) & 1)
    return rgb.astype(np.uint8)


class Relabel:
    def __init__(self, olabel, nlabel):
        self.olabel, self.nlabel = olabel, nlabel

    def __call__(self, tensor):
        assert tensor.dtype in (torch.int64, torch.uint8), "tensor needs to be LongTensor"
        tensor[tensor == self.olabel] = self.nlabel
        return tensor


class ToLabel:
    def __call__(self, image):
        return torch.from_numpy(np.array(image)).long().unsqueeze(0)


class Colorize:
    def __init__(self, n=22):
        cmap = colormap_cityscapes(256)
        cmap[n] = cmap[-1]
        self.cmap = torch.from_numpy(cmap[:n])

    def __call__(self, gray_image):
        """[1,H,W] integer labels -> uint8 [3,H,W] (one gather instead of a per-class loop)."""
        idx = gray_image[0].long().clamp(0, len(self.cmap))
        table = torch.cat([self.cmap, torch.zeros(1, 3, dtype=torch.uint8)]).to(idx.device)
        valid = gray_image[0].long() < len(self.cmap)
        idx = torch.where(valid, idx, torch.full_like(idx, len(self.cmap)))
        return table[idx].permute(2, 0, 1).contiguous()
