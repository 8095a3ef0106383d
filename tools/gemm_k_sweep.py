#!/usr/bin/env python3
"""Time pn2_conv_gemm (1x1, bf16) against the contraction length K at fixed M x N: separates the per-workgroup fixed cost (launch, first
LDS-DMA round trip, epilogue) from the per-K-step cost of the pipeline.  GPU box only."""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
import torch
from pn2 import capi
from pn2.capi import call, BF16

P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def run(N, H, W, K, Cout, code, stats=0):
    dev = "cuda"; st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    M = N * H * W
    Kp = (K + 127) // 128 * 128
    x = torch.randn(M, K, device=dev).bfloat16()
    wp = torch.randn((Cout + 127) // 128 * 128, Kp, device=dev).bfloat16()
    out = torch.empty(M, Cout, device=dev, dtype=torch.bfloat16)
    d = capi.ConvDesc()
    d.N, d.H, d.W, d.OH, d.OW = N, H, W, H, W
    d.Cin_p, d.ld_in, d.Cout, d.ld_out = K, K, Cout, Cout
    d.KH, d.KW, d.stride, d.pad_h, d.pad_w, d.dil_h, d.dil_w = 1, 1, 1, 0, 0, 1, 1
    d.transposed, d.Kp, d.flags = 0, Kp, (code << 8) | stats
    ps = pq = None
    if stats:
        nb = call.pn2_conv_stat_blocks(M, Cout, BF16) * 2
        ps, pq = torch.empty(nb, Cout, device=dev), torch.empty(nb, Cout, device=dev)
    return timeit(lambda: call.pn2_conv_gemm(BF16, P(x), P(wp), P(out), P(ps), P(pq), C.byref(d), st))


if __name__ == "__main__":
    for (N, H, W, Cout) in [(32, 22, 22, 1024), (32, 22, 22, 128), (32, 88, 88, 128), (32, 11, 11, 2048)]:
        M = N * H * W
        for name, code in (("dma 64x128", 2 | (1 << 2) | (3 << 4)), ("dma 128x128", 2 | (2 << 2) | (3 << 4)), ("dma 64x64", 2 | (1 << 2) | (2 << 4)),
                           ("dma2 64x128", 3 | (1 << 2) | (3 << 4)), ("dma2 128x128", 3 | (2 << 2) | (3 << 4)), ("dma2 64x64", 3 | (1 << 2) | (2 << 4))):
            row = []
            for K in (64, 128, 256, 512, 1024, 2048):
                t = run(N, H, W, K, Cout, code)
                row.append(f"K{K}: {t:6.1f}us {2.0 * M * K * Cout / t / 1e6:6.0f}TF")
            print(f"M{M:6d} N{Cout:5d} {name:14s} " + " | ".join(row))
