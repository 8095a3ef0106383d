#!/usr/bin/env python3
"""Per-kernel timing of one eager PVT_PraNet_V2 training step (config 4: bs=16, 352x352).  GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
os.environ.setdefault("PN2_NO_PRETRAINED", "1")
import torch
import pn2
from pn2.trainer import Trainer
from pn2.profile import Recorder
from lib.pranet import PVT_PraNet_V2
from bench import synthetic

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
pn2.set_compute_dtype(sys.argv[2] if len(sys.argv) > 2 else "bf16")
torch.manual_seed(0)
model = PVT_PraNet_V2(num_class=1).cuda().train()
tr = Trainer(model)
x, m = synthetic(bs, 352, 1234, "cuda")
for _ in range(3):
    tr.step(x, m)
torch.cuda.synchronize()
with Recorder() as rec:
    tr.step(x, m)
agg = rec.summary(detail=False)
tot = sum(d["ms"] for d in agg.values())
print(f"total kernel ms {tot:.2f}")
for k, d in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
    extra = f"{d['flops'] / d['ms'] / 1e9:8.1f} TF/s" if d["flops"] else ""
    print(f"{k:34s} {d['ms']:8.3f} ms {d['launches']:5d} launches {extra}")
print()
for k, d in sorted(rec.summary(detail=True).items(), key=lambda kv: -kv[1]["ms"])[:48]:
    extra = f"{d['flops'] / d['ms'] / 1e9:8.1f} TF/s" if d["flops"] else ""
    print(f"{k:84s} {d['ms']:8.3f} ms x{d['launches']:3d} {extra}")
