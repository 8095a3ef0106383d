#!/usr/bin/env python3
"""What one node of a captured step costs at least: N launches of a trivial kernel (pn2_copy of 64 elements) captured into one hipGraph, time per node at replay;
the same N launches issued eagerly on the stream.  (GPU box)"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
import torch
from pn2.capi import call, BF16

P = lambda t: C.c_void_p(t.data_ptr())
a = torch.zeros(64, 8, device="cuda", dtype=torch.bfloat16); b = torch.zeros_like(a)
N = 1000


def body():
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(N):
        call.pn2_copy(BF16, P(a), 8, BF16, P(b), 8, 64, 8, 0, st)


s = torch.cuda.Stream()
with torch.cuda.stream(s):
    body(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); body(); e1.record(); torch.cuda.synchronize()
    print(f"eager: {e0.elapsed_time(e1) * 1e3 / N:.2f} us per launch (host-bound if large)")
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    g.replay()
e1.record(); torch.cuda.synchronize()
print(f"hipGraph replay: {e0.elapsed_time(e1) * 1e3 / (5 * N):.2f} us per node")
