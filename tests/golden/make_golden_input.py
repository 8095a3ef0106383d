#!/usr/bin/env python3
"""Golden vectors for the input transform of binary_seg/utils/dataloader.py:104-120 (Resize -> ToTensor -> Normalize).
torchvision is not installed here; its Resize on a PIL image is PIL.Image.resize(size, BILINEAR) and ToTensor / Normalize are the two
arithmetic lines restated below, so the vectors are produced with Pillow itself.  Run in the build container; only data is committed."""
import os
import numpy as np
from PIL import Image

rng = np.random.default_rng(7)
out = {}
cases = [("down", 211, 257, 3, 96), ("up", 61, 47, 3, 96), ("mixed", 90, 150, 3, 96), ("gt", 211, 257, 1, 96), ("small", 37, 53, 3, 64), ("same", 64, 64, 3, 64), ("big", 160, 120, 3, 352)]
mean = np.array([0.485, 0.456, 0.406], np.float32); std = np.array([0.229, 0.224, 0.225], np.float32)
for name, H, W, C, S in cases:
    # smooth + noisy content so both the antialiasing taps and the rounding are exercised
    yy, xx = np.mgrid[0:H, 0:W]
    base = 127 + 100 * np.sin(yy / 17.0)[..., None] * np.cos(xx / 23.0)[..., None] + rng.normal(0, 25, (H, W, C))
    img = np.clip(base, 0, 255).astype(np.uint8)
    pil = Image.fromarray(img if C == 3 else img[:, :, 0], "RGB" if C == 3 else "L")
    res = np.asarray(pil.resize((S, S), Image.BILINEAR))
    out[name + ".in"] = img if C == 3 else img[:, :, 0]
    out[name + ".resized"] = res
    if name == "big":          # the full-size case pins the resize only (keeps the fixture small)
        continue
    t = res.astype(np.float32) / np.float32(255.0)                     # ToTensor
    if C == 3:
        t = (t - mean) / std                                           # Normalize (sub then div, fp32)
        out[name + ".tensor"] = np.ascontiguousarray(t.transpose(2, 0, 1))
    else:
        out[name + ".tensor"] = t[None]
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "input_pipeline.npz"), **out)
print({k: v.shape for k, v in out.items()})
