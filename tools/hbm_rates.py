#!/usr/bin/env python3
"""Achievable HBM rates of this part for the three traffic mixes the streaming kernels have: pure write (fill), copy (1 read : 1 write), pure read (sum).  GPU box only."""
import torch
def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (128, 512, 2048):
    n = mb << 20
    a = torch.empty(n // 4, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
    a.normal_()
    tw = t(lambda: a.fill_(1.0)); tc = t(lambda: b.copy_(a)); tr = t(lambda: a.sum())
    print(f"{mb:5d} MB: write {n / tw / 1e12:.2f} TB/s   copy {2 * n / tc / 1e12:.2f} TB/s (read+write)   read {n / tr / 1e12:.2f} TB/s")
