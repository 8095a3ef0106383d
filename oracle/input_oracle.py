"""CPU restatement of the reference's input transform (TEST INFRASTRUCTURE ONLY - nothing under pranet-v2_amd/ imports this).

binary_seg/utils/dataloader.py:104-111 (PolypDataset) and :176-181 (test_dataset):
    img_transform = Resize((S, S)) -> ToTensor() -> Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    gt_transform  = Resize((S, S)) -> ToTensor()
torchvision's Resize on a PIL image is PIL.Image.resize(size, BILINEAR); that algorithm lives in Pillow (third-party, not under /root/reference):
src/libImaging/Resample.c of the Pillow pinned by pranet2.yaml (9.4.0; the routine is unchanged through 12.x, which is what the golden vectors
were generated with).  Restated here from its published behaviour:
  * separable, horizontal pass then vertical pass, each rounding to uint8;
  * triangle filter whose support grows with the down-scaling factor (antialiasing): support = max(scale, 1), taps re-normalised to sum 1;
  * taps converted to fixed point with 22 fractional bits (round half away from zero), accumulation starts at 1 << 21, result >> 22, clipped.
Pinned by tests/golden/input_pipeline.npz (PIL outputs for up-/down-scaling, non-square and 1-channel inputs; generator tests/golden/make_golden_input.py).
"""
import numpy as np

PRECISION_BITS = 32 - 8 - 2
MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def resize_coeffs(in_size, out_size):
    """-> (xmin[out], count[out], kk[out][ksize] int32) exactly as Pillow's precompute_coeffs + normalize_coeffs_8bpc (bilinear)."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, np.int32); cnt = np.zeros(out_size, np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = int(center - support + 0.5)
        lo = max(lo, 0)
        hi = int(center + support + 0.5)
        hi = min(hi, in_size)
        n = hi - lo
        w = np.zeros(n, np.float64)
        for x in range(n):
            a = abs((x + lo - center + 0.5) * ss)
            w[x] = 1.0 - a if a < 1.0 else 0.0
        ww = w.sum() if n else 0.0
        # Pillow accumulates ww in a loop; a sequential sum reproduces its rounding
        ww = 0.0
        for x in range(n):
            ww += w[x]
        if ww != 0.0:
            w = w / ww
        for x in range(n):
            v = w[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w[x] < 0 else int(0.5 + v)
        xmin[xx] = lo; cnt[xx] = n
    return xmin, cnt, kk


def _pass(img, xmin, cnt, kk, axis):
    """One separable pass over `axis` of a uint8 [H][W][C] image."""
    out_size = len(xmin)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], np.int64)
    for xx in range(out_size):
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(cnt[xx]):
            acc += src[xmin[xx] + x] * int(kk[xx, x])
        out[xx] = acc >> PRECISION_BITS
    return np.moveaxis(np.clip(out, 0, 255).astype(np.uint8), 0, axis)


def pil_resize_bilinear_u8(img, out_h, out_w):
    """img uint8 [H][W][C] (or [H][W]) -> uint8 [out_h][out_w][C], PIL.Image.resize((out_w, out_h), BILINEAR)."""
    squeeze = img.ndim == 2
    if squeeze:
        img = img[:, :, None]
    H, W, _ = img.shape
    if W != out_w:
        img = _pass(img, *resize_coeffs(W, out_w), axis=1)
    if H != out_h:
        img = _pass(img, *resize_coeffs(H, out_h), axis=0)
    return img[:, :, 0] if squeeze else img


def train_transform(img_u8, gt_u8, size):
    """dataloader.py:104-120 -> (image fp32 [3][S][S] normalised, gt fp32 [1][S][S] in [0, 1])."""
    im = pil_resize_bilinear_u8(img_u8, size, size).astype(np.float32) / np.float32(255.0)
    im = (im - MEAN) / STD
    gt = pil_resize_bilinear_u8(gt_u8, size, size).astype(np.float32) / np.float32(255.0)
    return np.ascontiguousarray(im.transpose(2, 0, 1)), gt[None]
