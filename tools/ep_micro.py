#!/usr/bin/env python3
"""Micro-benchmark of the dgrad GEMM with / without the BatchNorm-backward statistics epilogue (pn2_conv_gemm vs pn2_conv_gemm_ep), cold operands.
Usage: python tools/ep_micro.py   (GPU box)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
import torch
from pn2 import capi
from pn2.capi import call, BF16

dev = "cuda"
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
thr = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def bench(fn, reps=5):
    ts = []
    for _ in range(reps):
        thr.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return min(ts)


def run(M, Cin, Cout, codes, accum=False, masky=False, dual=False):
    """1x1 'dgrad' GEMM: in [M][Cin] -> out [M][Cout]"""
    Kp = (Cin + 127) // 128 * 128
    x = torch.randn(M, Cin, device=dev).bfloat16()
    wp = (torch.randn((Cout + 127) // 128 * 128, Kp, device=dev) * 0.05).bfloat16()
    out = torch.zeros(M, Cout, device=dev, dtype=torch.bfloat16)
    outb = torch.zeros(M, Cout, device=dev, dtype=torch.bfloat16)
    raw = torch.randn(M, Cout, device=dev).bfloat16(); y = torch.randn(M, Cout, device=dev).bfloat16()
    rawb = torch.randn(M, Cout, device=dev).bfloat16()
    par = torch.randn(4, Cout, device=dev)
    d = capi.ConvDesc()
    d.N, d.H, d.W, d.OH, d.OW = 1, M, 1, M, 1
    d.Cin_p, d.ld_in, d.Cout, d.ld_out = Cin, Cin, Cout, Cout
    d.KH, d.KW, d.stride, d.pad_h, d.pad_w, d.dil_h, d.dil_w = 1, 1, 1, 0, 0, 1, 1
    d.transposed, d.Kp = 1, Kp
    res = []
    for code in codes:
        bm = 64 if (code >> 2) & 3 == 1 else 128
        nb = (M + bm - 1) // bm
        p1, p2, p3, p4 = (torch.zeros(nb, Cout, device=dev) for _ in range(4))
        d.flags = (code << 8) | (capi.CONV_ACCUM if accum else 0)
        t0 = bench(lambda: call.pn2_conv_gemm(BF16, P(x), P(wp), P(out), C.c_void_p(0), C.c_void_p(0), C.byref(d), st))
        ep = capi.ConvEp()
        ep.a.mode = capi.BNB_STATS | (capi.BNB_MASK_Y if masky else capi.BNB_MASK_RAW) | int(os.environ.get("EPDBG", "0"))
        ep.a.raw, ep.a.ld_raw, ep.a.par, ep.a.ps = raw.data_ptr(), Cout, par.data_ptr(), Cout
        ep.a.y, ep.a.ld_y = y.data_ptr(), Cout
        ep.a.p1, ep.a.p2, ep.a.ldp = p1.data_ptr(), p2.data_ptr(), Cout
        if dual:
            ep.b.out, ep.b.ld_out, ep.b.mode = outb.data_ptr(), Cout, capi.BNB_STATS | capi.BNB_MASK_RAW
            ep.b.raw, ep.b.ld_raw, ep.b.par, ep.b.ps = rawb.data_ptr(), Cout, par.data_ptr(), Cout
            ep.b.p1, ep.b.p2, ep.b.ldp = p3.data_ptr(), p4.data_ptr(), Cout
        t1 = bench(lambda: call.pn2_conv_gemm_ep(BF16, P(x), P(wp), P(out), C.byref(d), C.byref(ep), st))
        res.append((code, t0, t1))
    return res


if __name__ == "__main__":
    cases = [("layer1 conv3 dgrad 256->128", 247808, 256, 128, False, False, False),
             ("layer1 conv1 dgrad 128->256 accum+masky", 247808, 128, 256, True, True, False),
             ("layer2 conv3 dgrad 512->224", 61952, 512, 224, False, False, False),
             ("layer3 conv1 dgrad 416->1024 accum+masky", 15488, 416, 1024, True, True, False),
             ("narrow dual 32->32 (1x1 stand-in)", 247808, 32, 32, True, False, True)]
    codes = [k | (bm << 2) | (bn << 4) for k in (1, 2, 3) for bm in (1, 2) for bn in (1, 2, 3)]
    if os.environ.get("EPCODES"):
        codes = [int(c, 16) for c in os.environ["EPCODES"].split(",")]
        cases = cases[:2]
    for name, M, Cin, Cout, acc, my, dual in cases:
        cs = [c for c in codes if not (((c >> 4) & 3) == 3 and Cout <= 64) and not (((c >> 4) & 3) == 2 and Cout <= 32)]
        r = run(M, Cin, Cout, cs, acc, my, dual)
        best0 = min(r, key=lambda t: t[1]); best1 = min(r, key=lambda t: t[2])
        mb = M * (Cin + Cout) * 2 / 1e6
        print(f"{name}: in+out {mb:.0f} MB | plain best code {best0[0]:#x} {best0[1]:.1f} us | ep best code {best1[0]:#x} {best1[2]:.1f} us (ep at plain's code {best0[2]:.1f} us)")
        for code, t0, t1 in r:
            print(f"    code {code:#04x} k{code & 3} bm{(code >> 2) & 3} bn{(code >> 4) & 3}: plain {t0:7.1f} us   ep {t1:7.1f} us")
