"""Mirror of binary_seg/utils/dataloader.py (get_loader / PolypDataset / test_dataset) with the per-image transform moved to the GPU:
worker processes only decode files (PIL -> uint8), `pn2.input.DeviceTransform` does Resize -> ToTensor -> Normalize on the device, bit-exact with
the reference's torchvision-on-PIL pipeline.  Batches come back as GPU tensors (the `.cuda()` calls of MyTrain_med.py:62-63 become no-ops)."""
import os

import numpy as np
import torch
import torch.utils.data as data
from PIL import Image

from pn2.input import DeviceTransform


class PolypDataset(data.Dataset):
    """dataloader.py:92-150: same file discovery, sorting and size filter; __getitem__ returns the decoded uint8 pixels."""

    def __init__(self, image_root, gt_root, trainsize):
        self.trainsize = trainsize
        self.images = sorted(image_root + f for f in os.listdir(image_root) if f.endswith('.jpg') or f.endswith('.png'))
        self.gts = sorted(gt_root + f for f in os.listdir(gt_root) if f.endswith('.png'))
        self.filter_files()
        self.size = len(self.images)

    def __getitem__(self, index):
        return torch.from_numpy(np.asarray(self.rgb_loader(self.images[index])).copy()), torch.from_numpy(np.asarray(self.binary_loader(self.gts[index])).copy())

    def filter_files(self):
        assert len(self.images) == len(self.gts)
        images, gts = [], []
        for img_path, gt_path in zip(self.images, self.gts):
            if Image.open(img_path).size == Image.open(gt_path).size:
                images.append(img_path); gts.append(gt_path)
        self.images, self.gts = images, gts

    def rgb_loader(self, path):
        with open(path, 'rb') as f:
            return Image.open(f).convert('RGB')

    def binary_loader(self, path):
        with open(path, 'rb') as f:
            return Image.open(f).convert('L')

    def __len__(self):
        return self.size


class _DeviceBatches:
    def __init__(self, loader, transform):
        self.loader, self.transform = loader, transform

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for images, gts in self.loader:
            yield self.transform([im.cuda(non_blocking=True) for im in images], [g.cuda(non_blocking=True) for g in gts])


def get_loader(image_root, gt_root, batchsize, trainsize, shuffle=True, num_workers=4, pin_memory=True):
    """dataloader.py:153-161; iterating yields (images [N][3][S][S], gts [N][1][S][S]) fp32 GPU tensors."""
    dataset = PolypDataset(image_root, gt_root, trainsize)
    loader = data.DataLoader(dataset=dataset, batch_size=batchsize, shuffle=shuffle, num_workers=num_workers, pin_memory=pin_memory,
                             collate_fn=lambda items: ([i for i, _ in items], [g for _, g in items]))
    return _DeviceBatches(loader, DeviceTransform(trainsize))


class test_dataset:
    """dataloader.py:165-205: load_data() -> (image [1][3][S][S] normalised, on the GPU; gt as the PIL 'L' image; name)."""

    def __init__(self, image_root, gt_root, testsize):
        self.testsize = testsize
        self.images = sorted(image_root + f for f in os.listdir(image_root) if f.endswith('.jpg') or f.endswith('.png'))
        self.gts = sorted(gt_root + f for f in os.listdir(gt_root) if f.endswith('.tif') or f.endswith('.png'))
        self.transform = DeviceTransform(testsize)
        self.size = len(self.images)
        self.index = 0

    def load_data(self):
        image = torch.from_numpy(np.asarray(self.rgb_loader(self.images[self.index])).copy()).cuda()
        image = self.transform([image])
        gt = self.binary_loader(self.gts[self.index])
        name = self.images[self.index].split('/')[-1]
        if name.endswith('.jpg'):
            name = name.split('.jpg')[0] + '.png'
        self.index += 1
        return image, gt, name

    def rgb_loader(self, path):
        with open(path, 'rb') as f:
            return Image.open(f).convert('RGB')

    def binary_loader(self, path):
        with open(path, 'rb') as f:
            return Image.open(f).convert('L')

    def __len__(self):
        return self.size
