"""Device-side input transform: what PolypDataset / test_dataset apply to every decoded image
(binary_seg/utils/dataloader.py:104-111, 176-181): Resize((S, S)) -> ToTensor() -> Normalize(mean, std) for the image,
Resize((S, S)) -> ToTensor() for the mask.  File decoding stays on the host (PIL); everything after the uint8 pixels runs on the GPU and is
bit-exact with torchvision-on-PIL (the resize is Pillow's antialiased bilinear, two uint8 passes with 22-bit fixed-point taps)."""
import ctypes as C

import numpy as np
import torch

from . import capi
from .capi import call

_MEAN, _STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class DeviceTransform:
    """t = DeviceTransform(352); x, gt = t(images_u8, masks_u8) with lists of uint8 tensors [H][W][3] / [H][W] (any sizes)
    -> x fp32 [N][3][S][S] normalised, gt fp32 [N][1][S][S] in [0, 1]: the batch PolypDataset + DataLoader would collate."""

    def __init__(self, size, mean=_MEAN, std=_STD, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("pranet-v2_amd runs on MI355X only: no GPU visible and there is no CPU fallback")
        capi.load()
        self.size = int(size)
        self.dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.mean = torch.tensor(mean, dtype=torch.float32, device=self.dev)
        self.std = torch.tensor(std, dtype=torch.float32, device=self.dev)
        self._coeffs = {}

    def _taps(self, in_size):
        hit = self._coeffs.get(in_size)
        if hit is None:
            S = self.size
            ks = call.pn2_resize_ksize(in_size, S)
            xmin, cnt, kk = np.zeros(S, np.int32), np.zeros(S, np.int32), np.zeros((S, ks), np.int32)
            call.pn2_resize_coeffs(in_size, S, C.c_void_p(xmin.ctypes.data), C.c_void_p(cnt.ctypes.data), C.c_void_p(kk.ctypes.data))
            hit = self._coeffs[in_size] = tuple(torch.from_numpy(a).to(self.dev) for a in (xmin, cnt, kk)) + (ks,)
        return hit

    def resize(self, img):
        """uint8 [H][W][C] or [H][W] on the device -> uint8 [S][S][C]: PIL.Image.resize((S, S), BILINEAR)."""
        if not img.is_cuda or img.dtype != torch.uint8:
            raise RuntimeError("DeviceTransform needs uint8 GPU tensors (no CPU fallback)")
        if img.dim() == 2:
            img = img[:, :, None]
        img = img.contiguous()
        H, W, Cc = img.shape
        S, st = self.size, _stream()
        if W != S:                                           # Pillow's order: width pass, then height pass
            xmin, cnt, kk, ks = self._taps(W)
            tmp = torch.empty((H, S, Cc), dtype=torch.uint8, device=self.dev)
            call.pn2_resize_u8_pass(_p(img), _p(tmp), H, W, Cc, S, 1, _p(xmin), _p(cnt), _p(kk), ks, st)
            img, W = tmp, S
        if H != S:
            xmin, cnt, kk, ks = self._taps(H)
            out = torch.empty((S, S, Cc), dtype=torch.uint8, device=self.dev)
            call.pn2_resize_u8_pass(_p(img), _p(out), H, W, Cc, S, 0, _p(xmin), _p(cnt), _p(kk), ks, st)
            img = out
        return img

    def __call__(self, images, masks=None):
        S, st = self.size, _stream()
        x = torch.empty((len(images), 3, S, S), dtype=torch.float32, device=self.dev)
        for i, im in enumerate(images):
            r = self.resize(im)
            if r.shape[2] != 3:
                raise RuntimeError("images must be RGB (rgb_loader converts to 'RGB', dataloader.py:133-136)")
            call.pn2_u8_to_tensor(_p(r), _p(x[i]), S, S, 3, _p(self.mean), _p(self.std), st)
        if masks is None:
            return x
        g = torch.empty((len(masks), 1, S, S), dtype=torch.float32, device=self.dev)
        for i, m in enumerate(masks):
            r = self.resize(m)
            call.pn2_u8_to_tensor(_p(r), _p(g[i]), S, S, 1, C.c_void_p(0), C.c_void_p(0), st)
        return x, g
