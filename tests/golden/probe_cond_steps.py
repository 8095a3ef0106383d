#!/usr/bin/env python3
"""Build-container probe (imports /root/reference): does a short reference training run condition a random-init PraNet-V2 enough for a literal 1e-4 fixture?
Trains the reference N Adam steps (lr 1e-4, clip 0.5, 8 x 96^2 synthetic batches) and prints max |fp32 - float64| of the 8 train-mode maps, both runs from the SAME weights.
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/probe_cond_steps.py 40
Measured: 8.3e-4 (0 steps), 6.1e-4 (5), 4.7e-4 (10), 4.3e-4 (20), 2.7e-4 (40) - never below 1e-4."""
import os, sys, torch
sys.path.insert(0, '/root/repo/tests/golden'); sys.path.insert(0, '/root/repo')
from _ref_import import import_reference
from oracle import weights as W
R = import_reference()
torch.set_num_threads(8)
import importlib
def structure_loss(pred, mask):   # MyTrain_med.py:19-38 restated through the oracle's own function would import oracle; use reference's if importable
    import torch.nn.functional as F
    weit = 1 + 5 * torch.abs(F.avg_pool2d(mask, kernel_size=31, stride=1, padding=15) - mask)
    wbce = F.binary_cross_entropy_with_logits(pred, mask, reduction='none')
    wbce = (weit * wbce).sum(dim=(2, 3)) / weit.sum(dim=(2, 3))
    pred = torch.sigmoid(pred)
    inter = ((pred * mask) * weit).sum(dim=(2, 3)); union = ((pred + mask) * weit).sum(dim=(2, 3))
    wiou = 1 - (inter + 1) / (union - inter + 1)
    return (wbce + wiou).mean()
m = R.pranet.PraNet_V2(num_class=1)
m.load_state_dict(W.make_state_dict(W.manifest_pranet_v2(1), seed=0), strict=True)
m.train()
opt = torch.optim.Adam(m.parameters(), 1e-4)
def gap(m):
    import copy
    x, _ = W.synthetic_batch(2, 96, seed=1234)
    m32 = copy.deepcopy(m).train(); m64 = copy.deepcopy(m).double().train()
    with torch.no_grad():
        o32 = m32(x); o64 = m64(x.double())
    return max(float((a.double() - b).abs().max()) for a, b in zip(o32, o64))
print("steps 0: max |ref32 - ref64| on the 8 maps", gap(m), flush=True)
for i in range(int(sys.argv[1])):
    x, g = W.synthetic_batch(8, 96, seed=500 + i)
    opt.zero_grad()
    o = m(x)
    bg = 1 - g
    loss = sum(structure_loss(o[j], g) + structure_loss(o[j + 4], bg) for j in range(4))
    loss.backward()
    for grp in opt.param_groups:
        for p in grp['params']:
            if p.grad is not None: p.grad.data.clamp_(-0.5, 0.5)
    opt.step()
    if (i + 1) in (5, 10, 20, 40):
        print(f"steps {i+1}: loss {float(loss):.3f} max |ref32 - ref64|", gap(m), flush=True)
