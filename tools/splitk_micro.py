#!/usr/bin/env python3
"""Split-K against the plain launch for convs with few output rows and a long contraction (cold operands: a 512 MB fill before every timed launch)."""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
import torch
from pn2 import capi
from pn2.capi import call, BF16

P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
junk = None


def cold_time(fn, reps=5):
    global junk
    if junk is None:
        junk = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    fn(); ts = []
    for _ in range(reps):
        junk.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return min(ts)


def run(N, H, W, Cin, Cout, k, pad):
    dev = "cuda"; st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    M = N * H * W
    K = k * k * Cin; Kp = (K + 127) // 128 * 128
    x = torch.randn(M, Cin, device=dev).bfloat16()
    wp = (torch.randn((Cout + 127) // 128 * 128, Kp, device=dev) * 0.05).bfloat16()
    wp[:, K:] = 0
    out = torch.empty(M, Cout, device=dev, dtype=torch.bfloat16); out2 = torch.empty_like(out)
    nb = (M + 63) // 64
    ps, pq = torch.empty(nb, Cout, device=dev), torch.empty(nb, Cout, device=dev)
    ps2, pq2 = torch.empty(nb, Cout, device=dev), torch.empty(nb, Cout, device=dev)
    d = capi.ConvDesc()
    d.N, d.H, d.W, d.OH, d.OW = N, H, W, H, W
    d.Cin_p, d.ld_in, d.Cout, d.ld_out = Cin, Cin, Cout, Cout
    d.KH, d.KW, d.stride, d.pad_h, d.pad_w, d.dil_h, d.dil_w = k, k, 1, pad, pad, 1, 1
    d.transposed, d.Kp = 0, Kp
    res = []
    for kern in (2, 3):
        for bn in (2, 3):
            code = kern | (1 << 2) | (bn << 4)
            d.flags = (code << 8) | capi.CONV_STATS
            t = cold_time(lambda: call.pn2_conv_gemm(BF16, P(x), P(wp), P(out), P(ps), P(pq), C.byref(d), st))
            res.append((t, f"plain k{kern} 64x{32 << (bn - 1)}"))
    base = min(res)
    print(f"{N}x{H}x{W} {Cin}->{Cout} k{k}: M {M} K {K}  best plain {base[0]:.1f} us ({base[1]})")
    code = 3 | (1 << 2) | (3 << 4)
    d.flags = (code << 8) | capi.CONV_STATS
    call.pn2_conv_gemm(BF16, P(x), P(wp), P(out), P(ps), P(pq), C.byref(d), st)
    for S in (2, 4, 8):
        ws = torch.empty(S, M, Cout, device=dev)
        for kern in (2, 3):
            for bn in (2, 3):
                code = kern | (1 << 2) | (bn << 4)
                d.flags = (code << 8) | (S << 16)

                def both():
                    call.pn2_conv_gemm(BF16, P(x), P(wp), P(out2), P(ws), C.c_void_p(0), C.byref(d), st)
                    call.pn2_conv_splitk_reduce(BF16, P(ws), S, M, Cout, P(out2), Cout, C.c_void_p(0), P(ps2), P(pq2), 0, st)
                t = cold_time(both)
                torch.cuda.synchronize()
                e_o = float((out2.float() - out.float()).abs().max() / out.float().abs().max())
                e_s = float((ps2 - ps).abs().max() / ps.abs().max())
                print(f"   split {S} k{kern} 64x{32 << (bn - 1)}: {t:6.1f} us   rel diff out {e_o:.1e} stats {e_s:.1e}")


if __name__ == "__main__":
    run(32, 11, 11, 256, 256, 5, 2)
    run(16, 11, 11, 256, 256, 5, 2)
    run(32, 11, 11, 208, 208, 3, 1)
    run(32, 11, 11, 2048, 832, 1, 0)
