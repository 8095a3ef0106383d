#!/usr/bin/env python3
"""Times the conv GEMM kernels on the real layer shapes of the benchmark step (C ABI, cold operands): the shipped (kernel, tile) of pn2/tuned_gfx950.json against
the window-form kernel (tuning-code bit 6: persistent workgroups, rolling input window + weights resident in LDS) on its tiles.  Plain forward / dgrad launches (with the forward BatchNorm
statistics); GPU box only.   python tools/direct_micro.py [max shapes]"""
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
nmax = int(sys.argv[1]) if len(sys.argv) > 1 else 64
keys = [k for k in TABLE if k[0] == "g" and k[1] == 32 and len(k) == 17]
sel = [k for k in keys if k[11] == 1 and k[9] * k[10] > 1 and (k[2], k[3]) == (k[4], k[5])]
sel.sort(key=lambda k: (-(k[9] * k[10] > 1), -(k[1] * k[4] * k[5] * k[6] * k[8] * k[9] * k[10])))
tot0 = tot1 = 0.0
for key in sel[:nmax]:
    _, N, H, W, OH, OW, Cin_p, ld_in, Cout, KH, KW, s, ph, pw, dh, dw, tr = key
    taps = KH * KW
    M = N * OH * OW
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
    for bm in (1, 2, 3):
        for bn in (1, 2, 3):
            if (bn == 2 and Cout <= 32) or bn == 3 or bm == 1 or Cin_p > 64:
                continue
            cands.append(0x42 | (bm << 2) | (bn << 4))
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
    if len(res) < 2:
        continue
    best = min(res[1:])
    tot0 += t0; tot1 += min(t0, best[0])
    fl = 2 * M * Cout * Cin_p * taps
    print(f"{'dgrad' if tr else 'fwd  '} {Cin_p:4d}->{Cout:4d} k{KH}x{KW} s{s} d{dh} M{M:7d} ld{ld_in:4d}: table {base:#04x} {t0:7.1f} us ({fl / t0 / 1e6:5.0f} TF/s) | window " +
          " ".join(f"{c:#04x}:{t:6.1f}" for t, c in res[1:]) + f" | best x{t0 / best[0]:.2f}", flush=True)
print(f"sum over shapes: table {tot0:.0f} us, min(table, window) {tot1:.0f} us")
