#!/usr/bin/env python3
"""Every tile / kernel candidate of the conv GEMM on chosen shapes, cold operands (what the per-shape tuner sees), as a table: code = kernel | bm << 2 | bn << 4
(kernel 1 register-staged, 2 LDS-DMA 3-stage ring, 3 LDS-DMA 2-stage ring; bm 1/2 = 64/128 rows; bn 1/2/3 = 32/64/128 columns).  GPU box only.
Usage: gemm_codes_micro.py [reps]"""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
os.environ["PN2_TUNE_TABLE"] = "0"
import torch, torch.nn as nn
from pn2 import BF16, core, capi
from pn2.capi import call
from pn2.engine import Engine, _p, _stream
from pn2 import ops_conv

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 9
NAMES = {1: "reg", 2: "dma3", 3: "dma2"}



def run_all(self, t, key, cd, in_ptr, wp, M, Cout, ep):
    st = _stream()
    nul = C.c_void_p(0)
    scratch = torch.empty((M, Cout), dtype=torch.bfloat16, device=self.dev)
    d2 = capi.ConvDesc()
    C.memmove(C.byref(d2), C.byref(cd), C.sizeof(capi.ConvDesc))
    d2.ld_out, d2.Cout = Cout, Cout
    rows = []
    for kern in (1, 2, 3):
        for bm in (1, 2):
            for bn in (1, 2, 3):
                if (bn == 2 and Cout <= 32) or (bn == 3 and Cout <= 64) or (bm == 2 and M <= 64):
                    continue
                for ks in (0, 0x40, 0x80):
                    if ks and (kern < 2 or bn < 2):
                        continue
                    if ks == 0x80 and not (bm == 1 and bn == 2):
                        continue
                    code = kern | (bm << 2) | (bn << 4) | ks
                    d2.flags = code << 8
                    try:
                        call.pn2_conv_gemm(self.dt, in_ptr, _p(wp), _p(scratch), nul, nul, C.byref(d2), st)
                    except RuntimeError:
                        continue
                    res = {}
                    for cold in (True, False):
                        evs = []
                        for _ in range(REPS):
                            if cold:
                                core._thrash()
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record(); call.pn2_conv_gemm(self.dt, in_ptr, _p(wp), _p(scratch), nul, nul, C.byref(d2), st); e1.record()
                            evs.append((e0, e1))
                        torch.cuda.synchronize()
                        ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
                        res[cold] = (ts[0], ts[len(ts) // 2])
                    tiles = ((M + (64 << (bm - 1)) - 1) // (64 << (bm - 1))) * ((Cout + (16 << bn) - 1) // (16 << bn))
                    rows.append((code, NAMES[kern] + ("/ks2" if ks == 0x40 else "/ks4" if ks else ""), 64 << (bm - 1), 16 << bn, tiles, res[True], res[False]))
    fl = self._flops
    print(f"  {'code':>4} {'kern':>8} {'BM':>4} {'BN':>4} {'tiles':>6} | cold min / med us (TF/s at min) | warm min / med us")
    best = min(rows, key=lambda r: r[5][0])
    for r in rows:
        mark = " <-- best cold" if r is best else ""
        print(f"  {r[0]:4d} {r[1]:>8} {r[2]:4d} {r[3]:4d} {r[4]:6d} | {r[5][0]:7.1f} / {r[5][1]:7.1f} ({fl / r[5][0] / 1e6:6.1f}) | {r[6][0]:7.1f} / {r[6][1]:7.1f}{mark}")
    t[key] = best[0]
    return best[0]


ops_conv.ConvOps._tune_gemm_run = run_all if hasattr(ops_conv, "ConvOps") else None


def bench(N, H, W, Cin, Cout, k, pad=0):
    eng = Engine(BF16, True, need_grad=False, tuner={})
    x = eng.new_act(N, H, W, Cin); x.t.normal_()
    conv = nn.Conv2d(Cin, Cout, k, 1, pad, bias=False).cuda()
    M = N * H * W
    eng._flops = 2 * M * Cout * Cin * k * k
    print(f"{N}x{H}x{W} {Cin}->{Cout} k{k}  M={M}  {eng._flops / 1e9:.2f} GFLOP  in+out {(M * (Cin + Cout) * 2) / 1e6:.1f} MB")
    eng.conv_bn_act(x, conv, None)


if __name__ == "__main__":
    for cls in (c for c in vars(ops_conv).values() if isinstance(c, type) and hasattr(c, "_tune_gemm_run")):
        cls._tune_gemm_run = run_all
    bench(32, 22, 22, 104, 104, 3, 1)
    bench(32, 11, 11, 208, 208, 3, 1)
    bench(32, 11, 11, 2048, 832, 1)
    bench(32, 11, 11, 256, 256, 5, 2)
