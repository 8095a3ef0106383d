#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: EMCADNet(dual, K=9, pvt_v2_b2) forward + the reference trainer's 15-subset loss (torch ops, as trainer.py:106-140
runs them) + backward + optimizer step through the nn.Module surface.  Usage: emcad_bench.py [batch] [size] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
os.environ.setdefault("PN2_NO_PRETRAINED", "1")
import torch
import torch.nn.functional as F
import pn2
from pn2.profile import Recorder
from lib.networks import EMCADNet

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
size = int(sys.argv[2]) if len(sys.argv) > 2 else 512
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
pn2.set_compute_dtype("bf16")
torch.manual_seed(0)
model = EMCADNet(num_classes=9, kernel_sizes=[1, 3, 5], expansion_factor=2, activation="relu6", encoder="pvt_v2_b2", pretrain=False, dual=True).cuda().train()
GRAPH = os.environ.get("EMCAD_GRAPH", "0") == "1"          # replay the whole step (forward, loss, backward, AdamW) from one hipGraph
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-4, capturable=GRAPH)
x = torch.randn(bs, 1, size, size, device="cuda")
label = torch.randint(0, 9, (bs, size // 16, size // 16), device="cuda")
label = F.interpolate(label[:, None].float(), size=(size, size), mode="nearest")[:, 0].long()
bg = torch.stack([(label != k).float() for k in range(9)], 1)
subsets = [s for s in __import__("itertools").chain.from_iterable(__import__("itertools").combinations(range(4), r) for r in range(1, 5))]


def dice(logits, target):
    prob = torch.softmax(logits, 1); loss = 0.0
    for i in range(9):
        t = (target == i).float(); s = prob[:, i]
        loss = loss + (1 - (2 * (s * t).sum() + 1e-5) / ((s * s).sum() + (t * t).sum() + 1e-5))
    return loss / 9


FUSED = os.environ.get("EMCAD_TORCH_LOSS", "0") != "1"
from pn2.loss import mutation_loss


TRAINER = os.environ.get("EMCAD_TRAINER", "0") == "1"      # pn2.trainer.Trainer(loss="mutation"): arena, deferred table launches, AdamW kernel, hipGraph
if TRAINER:
    from pn2.trainer import Trainer
    tr = Trainer(model, lr=1e-4, clip=None, weight_decay=1e-4, loss="mutation", hot=model.hot_parameters(True))
    for _ in range(3):
        l = tr.step(x, (label, bg))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        l = tr.step(x, (label, bg))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"EMCADNet dual K=9 bs={bs} {size}x{size} bf16 (Trainer, eager): {1e3 * dt:.1f} ms/step  {bs / dt:.1f} img/s  loss {float(l[0]):.3f}")
    tr.capture(x, (label, bg), warmup=2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        l = tr.replay()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"EMCADNet dual K=9 bs={bs} {size}x{size} bf16 (Trainer, hipGraph replay): {1e3 * dt:.1f} ms/step  {bs / dt:.1f} img/s  loss {float(l[0]):.3f}")
    with Recorder() as rec:
        tr.step(x, (label, bg))
    agg = rec.summary()
    print(f"pn2 kernel time (event-bracketed, eager) {sum(d['ms'] for d in agg.values()):.1f} ms")
    for k, d in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:28]:
        print(f"{k:34s} {d['ms']:8.3f} ms {d['launches']:5d} launches")
    print()
    for k, d in sorted(rec.summary(detail=True).items(), key=lambda kv: -kv[1]["ms"])[:60]:
        extra = f"{d['flops'] / d['ms'] / 1e9:8.1f} TF/s" if d["flops"] else ""
        print(f"{k:90s} {d['ms']:8.3f} ms x{d['launches']:3d} {extra}")
    sys.exit(0)


def step():
    P = model(x, mode="train")
    if FUSED:
        loss = mutation_loss(P, label, bg)
    else:
        loss = 0.0
        for s in subsets:
            iout = sum(P[i] for i in s); ibg = sum(P[4 + i] for i in s)
            loss = loss + 0.5 * F.cross_entropy(iout, label) + 0.7 * dice(iout, label) + 0.3 * F.binary_cross_entropy_with_logits(ibg, bg)
    opt.zero_grad(); loss.backward(); opt.step()
    return loss


if GRAPH:
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            l = step()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        l = step()
    run = graph.replay
else:
    run = step
    for _ in range(2):
        l = step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    r = run()
    l = l if GRAPH else r
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print(f"EMCADNet dual K=9 bs={bs} {size}x{size} bf16 ({'fused pn2.loss.mutation_loss' if FUSED else 'torch loss'}{', hipGraph replay' if GRAPH else ''}): {1e3 * dt:.1f} ms/step  {bs / dt:.1f} img/s  loss {float(l.detach()):.3f}")
if GRAPH:
    sys.exit(0)
with Recorder() as rec:
    l = step()
agg = rec.summary()
torch.cuda.synchronize()
print(f"pn2 kernel time (event-bracketed, eager) {sum(d['ms'] for d in agg.values()):.1f} ms")
for k, d in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:24]:
    print(f"{k:34s} {d['ms']:8.3f} ms {d['launches']:5d} launches")
print()
for k, d in sorted(rec.summary(detail=True).items(), key=lambda kv: -kv[1]["ms"])[:45]:
    print(f"{k:90s} {d['ms']:8.3f} ms x{d['launches']:3d}")
