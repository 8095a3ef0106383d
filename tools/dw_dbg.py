#!/usr/bin/env python3
"""Ablation of the depth-wise 3x3 window walk: needs a library built with -DPN2_DW_ABLATE (make -C pranet-v2_amd/csrc FLAGS+=-DPN2_DW_ABLATE, or PN2_LIB=<that build>):
bit 1 of `flip` switches the halo rows off, bit 2 the stores.  Result on MI355X (16 x 88 x 88 x 512): 54.7 us full, 52.5 us with neither - the walk is not memory-bound."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pranet-v2_amd"))
import torch
from pn2.capi import call, BF16
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (N, H, W, Cc) in [(16, 88, 88, 512), (16, 44, 44, 1024)]:
    M = N * H * W
    x = torch.randn(M, Cc, device="cuda").bfloat16(); z = torch.empty_like(x); w = torch.randn(Cc, 9, device="cuda")
    for flip, what in ((1, "full"), (3, "centre row only"), (5, "no stores"), (7, "centre row, no stores")):
        t = timeit(lambda: call.pn2_dwconv3x3(BF16, P(x), P(w), None, P(z), None, N, H, W, Cc, flip, 0, st))
        print(N, H, W, Cc, f"{what:24s} {t:7.1f} us")
    y = torch.empty_like(x)
    t = timeit(lambda: y.copy_(x)); print("   torch copy", f"{t:7.1f} us  {2*M*Cc*2/t/1e3:.0f} GB/s")
