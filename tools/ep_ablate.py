#!/usr/bin/env python3
"""Ablation: the same forward conv launches (shipped tile, cold operands) with and without the BatchNorm statistics in the epilogue (PN2_CONV_STATS) - how
much of a short-K conv is epilogue instruction issue.  GPU box only."""
import ast, ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
import torch
from pn2 import capi
from pn2.capi import call, BF16
from pn2.engine import _thrash
TABLE = {ast.literal_eval(k): v for k, v in json.load(open(os.path.join(ROOT, "pranet-v2_amd", "pn2", "tuned_gfx950.json"))).items()}
rup = lambda v, m: (v + m - 1) // m * m
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
KEYS = [('g', 32, 88, 88, 88, 88, 256, 256, 128, 1, 1, 1, 0, 0, 1, 1, 0), ('g', 32, 88, 88, 88, 88, 128, 128, 256, 1, 1, 1, 0, 0, 1, 1, 0),
        ('g', 32, 88, 88, 88, 88, 32, 128, 32, 3, 3, 1, 1, 1, 1, 1, 0), ('g', 32, 44, 44, 44, 44, 56, 224, 56, 3, 3, 1, 1, 1, 1, 1, 0),
        ('g', 32, 22, 22, 22, 22, 104, 416, 104, 3, 3, 1, 1, 1, 1, 1, 0), ('g', 32, 22, 22, 22, 22, 416, 416, 1024, 1, 1, 1, 0, 0, 1, 1, 0),
        ('g', 32, 22, 22, 22, 22, 1024, 1024, 416, 1, 1, 1, 0, 0, 1, 1, 0), ('g', 32, 176, 176, 176, 176, 32, 32, 32, 3, 3, 1, 1, 1, 1, 1, 0)]
for key in KEYS:
    _, N, H, W, OH, OW, Cin_p, ld_in, Cout, KH, KW, s, ph, pw, dh, dw, tr = key
    taps, M = KH * KW, N * OH * OW
    d = capi.ConvDesc()
    d.N, d.H, d.W, d.OH, d.OW = N, H, W, OH, OW
    d.Cin_p, d.ld_in, d.Cout, d.ld_out = Cin_p, ld_in, Cout, Cout
    d.KH, d.KW, d.stride, d.pad_h, d.pad_w, d.dil_h, d.dil_w = KH, KW, s, ph, pw, dh, dw
    d.transposed, d.Kp = tr, rup(taps * Cin_p, 128)
    x = torch.randn(N * H * W, ld_in, device="cuda").bfloat16()
    wp = (torch.randn(rup(Cout, 128), d.Kp, device="cuda") * 0.05).bfloat16()
    out = torch.empty(M, Cout, dtype=torch.bfloat16, device="cuda")
    psum = torch.empty((M + 63) // 64, Cout, device="cuda"); psq = torch.empty_like(psum)
    code = TABLE[key]
    res = []
    for flags, a, b in ((capi.CONV_STATS, psum, psq), (0, None, None)):
        d.flags = flags | (code << 8)
        ts = []
        for rep in range(5):
            _thrash()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); call.pn2_conv_gemm(BF16, P(x), P(wp), P(out), P(a), P(b), C.byref(d), st()); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        res.append(min(ts[1:]))
    print(f"{Cin_p:4d}->{Cout:4d} k{KH}x{KW} M{M:7d} code {code:#04x}: with stats {res[0]:6.1f} us   without {res[1]:6.1f} us   ({100 * (res[0] - res[1]) / res[0]:.0f} %)")
