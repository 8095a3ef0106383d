#!/usr/bin/env python3
"""Per-kernel / per-conv-shape timing of one eager training step (HIP events around every launch).  GPU box only."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
os.environ.setdefault("PN2_NO_PRETRAINED", "1")
import torch
import pn2
from pn2.trainer import Trainer
from pn2.profile import Recorder
from lib.pranet import PraNet_V2
from bench import synthetic

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dt = sys.argv[2] if len(sys.argv) > 2 else "bf16"
pn2.set_compute_dtype(dt)
torch.manual_seed(0)
model = PraNet_V2(num_class=1).cuda().train()
tr = Trainer(model)
x, m = synthetic(bs, 352, 1234, "cuda")
for _ in range(2):
    tr.step(x, m)
with Recorder() as rec:
    tr.step(x, m)
agg = rec.summary(detail=True)
tot = sum(d["ms"] for d in agg.values())
print(f"total kernel ms {tot:.2f}")
by_kernel = {}
for name, d in agg.items():
    k = name.split(" ")[0]
    e = by_kernel.setdefault(k, [0.0, 0, 0, 0]); e[0] += d["ms"]; e[1] += d["launches"]; e[2] += d["flops"]; e[3] += d["bytes"]
for k, (ms, n, fl, by) in sorted(by_kernel.items(), key=lambda kv: -kv[1][0]):
    extra = f"{fl / ms / 1e9:8.1f} TF/s" if fl else (f"{by / ms / 1e6:8.1f} GB/s" if by else "")
    print(f"{k:34s} {ms:8.3f} ms {n:5d} launches {extra}")
print()
for name, d in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
    if d["flops"]:
        print(f"{name:70s} {d['ms']:7.3f} ms x{d['launches']:3d} {d['flops'] / d['ms'] / 1e9:8.1f} TF/s")
print()
for name, d in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:400]:
    if not d["flops"] and d["ms"] > 0.04:
        bw = f"{d['bytes'] / d['ms'] / 1e6:8.1f} GB/s" if d["bytes"] else ""
        print(f"{name:80s} {d['ms']:7.3f} ms x{d['launches']:3d} {bw}")

rq = tr.grad_queue
if rq is not None:
    tot = sum(t.numel() * 4 for t in rq.slabs.values())
    print(f"\nwgrad slabs: {len(rq.slabs)} tensors, {tot / 1e6:.1f} MB total")
    hist = {}
    for t in rq.slabs.values():
        k = tuple(t.shape)
        hist.setdefault(k, [0, 0]); hist[k][0] += 1; hist[k][1] += t.numel() * 4
    for k, (n, b) in sorted(hist.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"  slab {k}: x{n}  {b / 1e6:.1f} MB")
    print("tuned wgrad choices:", {k[1:]: v for k, v in tr.tuner.items() if k[0] == "w"} if tr.tuner else None)
