#!/usr/bin/env python3
"""fp32fast 1x1-conv GEMM (conv_gather_gemm<f32f_t>): time against K for fixed M, N and every tile code - intercept = prologue + epilogue, slope = steady-state rate.
GPU box: python tools/probes/f32f_gemm_scan.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
import torch
from pn2 import capi
from pn2.capi import call

capi.load()
dev = "cuda"
P = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(M, K, N, bm, bn, reps=20):
    x = torch.randn(M, K, device=dev)
    Kp = (K + 127) // 128 * 128
    wp = torch.randn((N + 127) // 128 * 128, Kp, device=dev) * 0.05
    out = torch.empty(M, N, device=dev)
    d = capi.ConvDesc()
    d.KH = d.KW = d.stride = d.dil_h = d.dil_w = 1
    d.N, d.H, d.W, d.OH, d.OW = 1, M, 1, M, 1
    d.Cin_p, d.ld_in, d.Cout, d.ld_out, d.Kp = K, K, N, N, Kp
    d.flags = (1 | (bm << 2) | (bn << 4)) << 8
    for _ in range(3):
        call.pn2_conv_gemm(capi.F32F, P(x), P(wp), P(out), C.c_void_p(0), C.c_void_p(0), C.byref(d), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call.pn2_conv_gemm(capi.F32F, P(x), P(wp), P(out), C.c_void_p(0), C.c_void_p(0), C.byref(d), st)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3          # us


for M in (247808, 15488):
    for N in (64, 256):
        print(f"M={M} N={N}")
        for bm, bn in ((2, 2), (1, 2), (2, 1), (2, 3)):
            row = []
            for K in (64, 128, 256, 512, 1024, 2048):
                if M * K * 4 > 3e9:
                    continue
                us = run(M, K, N, bm, bn)
                tf = 2.0 * M * K * N / us * 1e-6
                gb = (M * K + M * N) * 4 / us * 1e-3
                row.append(f"K={K}: {us:7.1f}us {tf:5.1f}TF {gb:5.0f}GB/s")
            print(f"  tile {64*bm}x{[0,32,64,128][bn]}: " + " | ".join(row))
