#!/usr/bin/env python3
"""Micro-benchmark of the BN streaming kernels (affine / bwd_reduce / bwd_apply) on the shapes of the training step (GPU box)."""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
import torch
from pn2 import capi
from pn2.capi import call, BF16

P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def timeit(fn, reps=50):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def bench(M, Cc, ld_dy, ld_x):
    dev = "cuda"
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    dy = torch.randn(M, ld_dy, device=dev).bfloat16()
    x = torch.randn(M, ld_x, device=dev).bfloat16()
    y = torch.randn(M, ld_dy, device=dev).bfloat16()
    dx = torch.empty(M, ld_x, device=dev, dtype=torch.bfloat16)
    mean = torch.zeros(Cc, device=dev); invstd = torch.ones(Cc, device=dev)
    coef = torch.ones(3 * Cc, device=dev)
    sc = torch.ones(Cc, device=dev); sh = torch.zeros(Cc, device=dev)
    nb = call.pn2_bn_bwd_blocks(M, Cc, BF16)
    p1 = torch.empty(nb, Cc, device=dev); p2 = torch.empty(nb, Cc, device=dev)
    t_aff = timeit(lambda: call.pn2_affine_act(BF16, P(x), ld_x, BF16, P(y), ld_dy, M, Cc, P(sc), P(sh), C.c_void_p(0), 0, 1, st))
    # (the step's form: mask recomputed from the raw conv output, no stored y)
    t_red = timeit(lambda: call.pn2_bn_bwd_reduce(BF16, BF16, P(dy), ld_dy, Cc, C.c_void_p(0), 0, BF16, P(x), ld_x, M, Cc, P(mean), P(invstd), P(p1), P(p2), nb,
                                                  P(sc), P(sh), 0, st))
    # the step's dominant form: ReLU mask recomputed from the raw conv output (msc / msh), no stored y
    t_app = timeit(lambda: call.pn2_bn_bwd_apply(BF16, BF16, P(dy), ld_dy, Cc, C.c_void_p(0), 0, BF16, P(x), ld_x, M, Cc, P(mean), P(invstd), P(coef), P(dx), ld_x,
                                                 C.c_void_p(0), 0, 0, P(sc), P(sh), 0, st))
    b = M * Cc * 2
    print(f"M{M:7d} C{Cc:5d} lddy{ld_dy:5d}: affine {t_aff:6.1f} us {2*b/t_aff/1e3:6.0f} GB/s | reduce(nb{nb:4d}) {t_red:6.1f} us {2*b/t_red/1e3:6.0f} GB/s | apply {t_app:6.1f} us {3*b/t_app/1e3:6.0f} GB/s")


if __name__ == "__main__":
    for M, Cc, lddy, ldx in [(15488, 104, 416, 104), (61952, 56, 224, 56), (247808, 32, 128, 32), (3872, 208, 832, 208), (15488, 1024, 1024, 1024),
                             (61952, 512, 512, 512), (247808, 256, 256, 256), (3872, 2048, 2048, 2048), (15488, 416, 416, 416), (61952, 32, 32, 32)]:
        bench(M, Cc, lddy, ldx)
