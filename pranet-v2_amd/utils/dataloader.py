"""Host side of the input path: the names MyTrain_med.py / MyTest_med.py import (`get_loader`, `test_dataset`) in front of the device-side transform.

Only the hand-over matters here: files are decoded to uint8 pixels on the host and `pn2.input.DeviceTransform` does Resize -> ToTensor ->
Normalize on the GPU, bit-exact with torchvision-on-PIL (tests/test_gpu_input.py); batches arrive as GPU tensors, so the `.cuda()` calls of
MyTrain_med.py:62-63 are no-ops.  File discovery keeps the reference's observable behaviour (dataloader.py:92-150, 165-205): sorted listing of
.jpg / .png images and .png (test: .tif / .png) masks, training pairs whose two files differ in size are skipped."""
import os

import numpy as np
import torch
from PIL import Image
from torch.utils.data import DataLoader, Dataset

from pn2.input import DeviceTransform


def _listing(root, suffixes):
    return sorted(root + name for name in os.listdir(root) if name.endswith(suffixes))


def _pixels(path, mode):
    """decoded file as a uint8 tensor: HxWx3 for mode 'RGB', HxW for 'L'"""
    with Image.open(path) as im:
        return torch.from_numpy(np.array(im.convert(mode), dtype=np.uint8))


def _same_size(a, b):
    with Image.open(a) as ia, Image.open(b) as ib:
        return ia.size == ib.size


class PolypDataset(Dataset):
    """(image uint8 HxWx3, mask uint8 HxW) pairs of a training directory."""

    def __init__(self, image_root, gt_root, trainsize):
        images, masks = _listing(image_root, ('.jpg', '.png')), _listing(gt_root, ('.png',))
        assert len(images) == len(masks), f"{len(images)} images but {len(masks)} masks"          # (the reference asserts too)
        pairs = [(i, m) for i, m in zip(images, masks) if _same_size(i, m)]
        self.trainsize = trainsize
        self.images, self.gts = [p[0] for p in pairs], [p[1] for p in pairs]
        self.size = len(pairs)

    def __len__(self):
        return self.size

    def __getitem__(self, index):
        return _pixels(self.images[index], 'RGB'), _pixels(self.gts[index], 'L')


def _keep_separate(items):
    # images of one batch differ in size until the device transform has resized them: no stacking on the host
    return [it[0] for it in items], [it[1] for it in items]


class _DeviceBatches:
    """iterates a host DataLoader and hands every batch through the device transform"""

    def __init__(self, loader, transform):
        self.loader, self.transform = loader, transform

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for images, masks in self.loader:
            yield self.transform([t.cuda(non_blocking=True) for t in images], [t.cuda(non_blocking=True) for t in masks])


def get_loader(image_root, gt_root, batchsize, trainsize, shuffle=True, num_workers=4, pin_memory=True):
    """Same signature as the reference's; iterating yields (images [N][3][S][S], gts [N][1][S][S]) fp32 GPU tensors."""
    host = DataLoader(PolypDataset(image_root, gt_root, trainsize), batch_size=batchsize, shuffle=shuffle, num_workers=num_workers,
                      pin_memory=pin_memory, collate_fn=_keep_separate)
    return _DeviceBatches(host, DeviceTransform(trainsize))


class test_dataset:
    """load_data() -> (image [1][3][S][S] normalised, on the GPU; the mask as a PIL 'L' image; output file name), one image per call."""

    def __init__(self, image_root, gt_root, testsize):
        self.testsize = testsize
        self.images, self.gts = _listing(image_root, ('.jpg', '.png')), _listing(gt_root, ('.tif', '.png'))
        self.transform = DeviceTransform(testsize)
        self.size = len(self.images)
        self.index = 0

    def __len__(self):
        return self.size

    def load_data(self):
        path = self.images[self.index]
        image = self.transform([_pixels(path, 'RGB').cuda()])
        with Image.open(self.gts[self.index]) as im:
            gt = im.convert('L')
        stem, ext = os.path.splitext(os.path.basename(path))
        self.index += 1
        return image, gt, (stem + '.png' if ext == '.jpg' else stem + ext)
