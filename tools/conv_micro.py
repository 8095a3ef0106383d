#!/usr/bin/env python3
"""Micro-benchmark of the implicit-GEMM conv kernel on chosen shapes (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
import torch, torch.nn as nn
from pn2 import BF16
from pn2.engine import Engine

def bench(N, H, W, Cin, Cout, k, pad=0, stride=1, reps=20):
    eng = Engine(BF16, True, need_grad=False)
    x = eng.new_act(N, H, W, Cin); x.t.normal_()
    conv = nn.Conv2d(Cin, Cout, k, stride, pad, bias=False).cuda()
    for _ in range(3):
        eng.conv_bn_act(x, conv, None)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        eng.conv_bn_act(x, conv, None)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps      # includes pack_weight + affine copy; subtract by timing those? keep: they are small for big shapes
    M = N * ((H + 2 * pad - k) // stride + 1) ** 2
    fl = 2 * M * Cout * Cin * k * k
    print(f"{N}x{H}x{W} {Cin}->{Cout} k{k}: {ms*1e3:8.1f} us  {fl/ms/1e9:8.1f} TF/s (incl. pack+copy)")

if __name__ == "__main__":
    bench(1, 64, 64, 4096, 4096, 1)
    bench(4, 64, 64, 4096, 4096, 1)
    bench(32, 22, 22, 416, 1024, 1)
    bench(32, 22, 22, 1024, 416, 1)
    bench(32, 88, 88, 104, 256, 1)
    bench(32, 22, 22, 104, 104, 3, 1)
    bench(32, 11, 11, 256, 256, 5, 2)
