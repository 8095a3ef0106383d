#!/usr/bin/env python3
"""Gradient-probe accuracy of the fp32 path on the conditioned fixture (tests/golden/pranet_v2_cond.npz): per probe rel-L2 against the reference's float64
gradient next to the reference's own fp32 deviation.  `python tools/cond_probe.py t96 [fp32|bf16]`; environment knobs (PN2_BNB_EPILOGUE=0, ...) bisect a deviation."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
os.environ.setdefault("PN2_NO_PRETRAINED", "1")
import numpy as np, torch
import pn2
from lib.pranet import PraNet_V2
from pn2.loss import structure_loss
from oracle import weights as W

tag = sys.argv[1] if len(sys.argv) > 1 else "t96"
fp32 = (sys.argv[2] if len(sys.argv) > 2 else "fp32") == "fp32"
z = np.load(os.path.join(ROOT, "tests", "golden", "pranet_v2_cond.npz"))
pn2.set_compute_dtype("fp32" if fp32 else "bf16")
model = PraNet_V2(num_class=1)
model.load_state_dict(W.make_state_dict(W.manifest_pranet_v2(1), seed=0, bn3_gamma=float(z["bn3_gamma"])), strict=True)
model = model.cuda().train()
x, mask = W.synthetic_batch(int(z[f"{tag}.n"]), int(z[f"{tag}.size"]), seed=4242)
x, mask = x.cuda(), mask.cuda()
outs = model(x)
losses = [structure_loss(outs[i], outs[i + 4], mask, 1 - mask) for i in range(4)]
(losses[3] + losses[2] + losses[1] + losses[0]).backward()
named = dict(model.named_parameters())
T = lambda a: torch.from_numpy(np.asarray(a)).double()
rl = lambda a, b: float((a - b).norm() / (b.norm() + 1e-30))
knobs = {k: v for k, v in os.environ.items() if k.startswith("PN2_") and k not in ("PN2_NO_PRETRAINED",)}
print(f"== {tag} {'fp32' if fp32 else 'bf16'} {knobs}")
for f in z.files:
    if f.startswith(f"{tag}.graw."):
        k = f[len(tag) + 6:]
        r32, r64 = T(z[f]), T(z[f"{tag}.f64.graw." + k])
        got = named[k].grad.reshape(-1)[:256].double().cpu()
        print(f"{k:45s} ours {rl(got, r64):.2e}  ref32 {rl(r32, r64):.2e}  |g64| {float(r64.norm()):.2e}")
