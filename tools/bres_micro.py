#!/usr/bin/env python3
"""Times the conv GEMM kernels on real layer shapes of the benchmark (C ABI, cold operands): the shipped tile vs the persistent B-resident kernel (code | 0x40)
on the same tile and on the other tiles.  GPU box only.   python tools/bres_micro.py"""
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
only = sys.argv[1:] 
keys = [k for k in TABLE if k[0] == "g" and k[1] == 32 and len(k) == 17]
sel = [k for k in keys if (k[9], k[10]) in ((1, 1), (3, 3)) and k[12] in (0, 1) and k[14] == 1 and k[11] == 1]
sel.sort(key=lambda k: -(k[1] * k[4] * k[5] * k[6] * k[8] * k[9] * k[10]))
for key in sel[:48]:
    _, N, H, W, OH, OW, Cin_p, ld_in, Cout, KH, KW, s, ph, pw, dh, dw, tr = key
    taps = KH * KW
    M = N * OH * OW
    ksteps = (taps * Cin_p + 63) // 64
    d = capi.ConvDesc()
    d.N, d.H, d.W, d.OH, d.OW = N, H, W, OH, OW
    d.Cin_p, d.ld_in, d.Cout, d.ld_out = Cin_p, ld_in, Cout, Cout
    d.KH, d.KW, d.stride, d.pad_h, d.pad_w, d.dil_h, d.dil_w = KH, KW, s, ph, pw, dh, dw
    d.transposed, d.Kp = tr, rup(taps * Cin_p, 128)
    x = torch.randn(N * H * W, ld_in, device="cuda").bfloat16()
    wp = (torch.randn(rup(Cout, 128), d.Kp, device="cuda") * 0.05).bfloat16()
    out = torch.empty(M, Cout, dtype=torch.bfloat16, device="cuda")
    nb64 = (M + 63) // 64
    psum = torch.empty(nb64, Cout, device="cuda"); psq = torch.empty(nb64, Cout, device="cuda")
    base = TABLE[key]
    cands = [base]
    for kern in (2, 3):
        for bm in (1, 2):
            for bn in (1, 2, 3):
                if (bn == 2 and Cout <= 32) or (bn == 3 and Cout <= 64) or (bm == 2 and M <= 64):
                    continue
                cands.append(kern | (bm << 2) | (bn << 4) | 0x40)
    res = []
    for code in cands:
        d.flags = (capi.CONV_STATS if not tr else 0) | (code << 8)
        ts = []
        for rep in range(4):
            _thrash()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            call.pn2_conv_gemm(BF16, P(x), P(wp), P(out), P(psum) if not tr else P(None), P(psq) if not tr else P(None), C.byref(d), st())
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        res.append((min(ts[1:]), code))
    t0 = res[0][0]
    best = min(res[1:])
    fl = 2 * M * Cout * Cin_p * taps
    print(f"{'dgrad' if tr else 'fwd  '} {Cin_p:4d}->{Cout:4d} k{KH}x{KW} M{M:7d} ld{ld_in:4d} ksteps{ksteps:3d}: table code {base:#04x} {t0:7.1f} us ({fl / t0 / 1e6:6.0f} TF/s) | best bres {best[1]:#04x} {best[0]:7.1f} us  x{t0 / best[0]:.2f}", flush=True)
