#!/usr/bin/env python3
"""Golden vectors for the bf16 whole-model gate.  Runs ONLY in the build container (needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_bf16.py

The benchmarked precision is bf16 storage + bf16 matrix cores with fp32 accumulation.  On the whole-model fixtures (random init, train-mode
BatchNorm over a 2-image batch) rounding noise is amplified ~500x between the first conv and the logits (the reference's own fp32 run is
1e-3 away from its float64 run), so ANY bf16 execution lands O(1) relative L2 away from the float64 logits.  The yardstick stored here is
the imported reference itself under PyTorch's own bf16 policy, torch.autocast("cpu", dtype=torch.bfloat16) (convs in bf16, BatchNorm in
fp32): the 8 output maps and the 4 pair losses, same seeds / weights / sub-sampling as pranet_v2_{96,352}.npz.
"""
import os, sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE); sys.path.insert(0, ROOT)
from _ref_import import import_reference            # noqa: E402
from oracle import weights as W                      # noqa: E402

torch.set_num_threads(8)
R = import_reference()
out = {}
for tag, size, full in (("96", 96, True), ("352", 352, False)):
    model = R.pranet.PraNet_V2(num_class=1)
    model.load_state_dict(W.make_state_dict(W.manifest_pranet_v2(1), seed=0), strict=True)
    model.train()
    x, mask = W.synthetic_batch(2, size, seed=1234)
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        outs = [o.float() for o in model(x)]
    losses = [float(R.train.structure_loss(outs[i], outs[i + 4], mask, 1 - mask)) for i in range(4)]
    out[f"{tag}.losses"] = np.array(losses)
    for i, o in enumerate(outs):
        out[f"{tag}.out{i}"] = (o if full else o[:, :, ::4, ::4]).numpy().astype(np.float32)
np.savez_compressed(os.path.join(HERE, "pranet_v2_bf16ref.npz"), **out)
print("wrote pranet_v2_bf16ref.npz", len(out), "arrays")
