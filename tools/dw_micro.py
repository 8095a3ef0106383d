#!/usr/bin/env python3
"""Micro-benchmark of the depth-wise 3x3 kernels on the Mlp shapes of PVTv2-B2 (bs=16, 352^2)."""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
import torch
from pn2.capi import call, BF16

P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def bench(N, H, W, Cc):
    dev = "cuda"; st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    M = N * H * W
    x = torch.randn(M, Cc, device=dev).bfloat16(); dz = torch.randn(M, Cc, device=dev).bfloat16()
    z = torch.empty_like(x); y = torch.empty_like(x)
    w = torch.randn(Cc, 9, device=dev); b = torch.randn(Cc, device=dev)
    nb = call.pn2_dwconv3x3_wgrad_blocks(BF16, N, H, W, Cc)
    part = torch.empty(nb, Cc * 10, device=dev)
    t_f = timeit(lambda: call.pn2_dwconv3x3(BF16, P(x), P(w), P(b), P(z), P(y), N, H, W, Cc, 0, 0, st))
    t_d = timeit(lambda: call.pn2_dwconv3x3(BF16, P(dz), P(w), None, P(z), None, N, H, W, Cc, 1, 0, st))
    t_w = timeit(lambda: call.pn2_dwconv3x3_wgrad(BF16, P(dz), P(x), P(part), nb, N, H, W, Cc, None, None, st))
    t_wg = timeit(lambda: call.pn2_dwconv3x3_wgrad(BF16, P(dz), P(x), P(part), nb, N, H, W, Cc, P(x), P(z), st))
    by = M * Cc * 2
    print(f"{N}x{H}x{W}x{Cc:5d} ({by/1e6:6.1f} MB): fwd+gelu {t_f:7.1f} us {3*by/t_f/1e3:6.0f} GB/s | dgrad {t_d:7.1f} us {2*by/t_d/1e3:6.0f} GB/s | wgrad(nb{nb}) {t_w:7.1f} us {2*by/t_w/1e3:6.0f} GB/s | +gelu' {t_wg:7.1f} us {4*by/t_wg/1e3:6.0f} GB/s")


if __name__ == "__main__":
    for shp in [(16, 88, 88, 512), (16, 44, 44, 1024), (16, 22, 22, 1280), (16, 11, 11, 2048), (16, 128, 128, 512), (16, 64, 64, 1024)]:
        bench(*shp)
