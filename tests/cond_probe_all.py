#!/usr/bin/env python3
"""Diagnostic (not a test): every parameter gradient of the fp32 / bf16 path on the conditioned weights against the CPU oracle run in float64, in network order.
    python tests/cond_probe_all.py [n] [size] [fp32|bf16]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
os.environ.setdefault("PN2_NO_PRETRAINED", "1")
import torch
import pn2
from lib.pranet import PraNet_V2
from pn2.loss import structure_loss
from oracle import weights as W
from oracle import pranet_oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
size = int(sys.argv[2]) if len(sys.argv) > 2 else 96
dtn = sys.argv[3] if len(sys.argv) > 3 else "fp32"
sd = W.make_state_dict(W.manifest_pranet_v2(1), seed=0, bn3_gamma=0.05)
x, mask = W.synthetic_batch(n, size, seed=4242)
P = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
for k in O.params_of(P):
    P[k].requires_grad_(True)
ro = O.pranet_v2_forward(P, x.double(), True)
O.total_loss(ro, mask.double()).backward()
pn2.set_compute_dtype(dtn)
model = PraNet_V2(num_class=1)
model.load_state_dict(sd, strict=True)
model = model.cuda().train()
xg, mg = x.cuda(), mask.cuda()
outs = model(xg)
losses = [structure_loss(outs[i], outs[i + 4], mg, 1 - mg) for i in range(4)]
(losses[3] + losses[2] + losses[1] + losses[0]).backward()
print("logits max abs err", max(float((o.detach().cpu().double() - r.detach()).abs().max()) for o, r in zip(outs, ro)))
for k, p in model.named_parameters():
    if p.grad is None or P[k].grad is None:
        continue
    g, r = p.grad.double().cpu(), P[k].grad
    print(f"{k:50s} rel-L2 {float((g - r).norm() / (r.norm() + 1e-30)):.2e}   |g| {float(r.norm()):.2e}")
