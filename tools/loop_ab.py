#!/usr/bin/env python3
"""A/B of two builds of libpn2_hip.so on the conv GEMM (cold operands): python tools/loop_ab.py libA.so libB.so   (GPU box)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
import torch
from pn2 import capi

dev = "cuda"
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
libs = [C.CDLL(os.path.abspath(p)) for p in sys.argv[1:3]]
for l in libs:
    l.pn2_conv_gemm.argtypes = capi.SIGNATURES["pn2_conv_gemm"]; l.pn2_conv_gemm.restype = C.c_int
thr = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def bench(fn, reps=5):
    ts = []
    for _ in range(reps):
        thr.fill_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return min(ts)


def case(N, H, W, Cin, Cout, k, pad):
    M = N * H * W
    Kp = (k * k * Cin + 127) // 128 * 128
    x = torch.randn(M, Cin, device=dev).bfloat16()
    wp = (torch.randn((Cout + 127) // 128 * 128, Kp, device=dev) * 0.05).bfloat16()
    out = torch.zeros(M, Cout, device=dev, dtype=torch.bfloat16)
    d = capi.ConvDesc()
    d.N, d.H, d.W, d.OH, d.OW = N, H, W, H, W
    d.Cin_p, d.ld_in, d.Cout, d.ld_out = Cin, Cin, Cout, Cout
    d.KH, d.KW, d.stride, d.pad_h, d.pad_w, d.dil_h, d.dil_w = k, k, 1, pad, pad, 1, 1
    d.transposed, d.Kp = 0, Kp
    fl = 2 * M * Cin * Cout * k * k
    best = [1e9, 1e9]; rows = []
    for kern in (2, 3):
        for bm in (1, 2):
            for bn in (1, 2, 3):
                if (bn == 3 and Cout <= 64) or (bn == 2 and Cout <= 32):
                    continue
                code = kern | (bm << 2) | (bn << 4)
                d.flags = code << 8
                t = []
                for li, l in enumerate(libs):
                    rc = l.pn2_conv_gemm(capi.BF16, P(x), P(wp), P(out), C.c_void_p(0), C.c_void_p(0), C.byref(d), st)
                    assert rc == 0, rc
                    t.append(bench(lambda: l.pn2_conv_gemm(capi.BF16, P(x), P(wp), P(out), C.c_void_p(0), C.c_void_p(0), C.byref(d), st)))
                    best[li] = min(best[li], t[-1])
                rows.append((code, t))
    print(f"{N}x{H}x{W} {Cin}->{Cout} k{k}: best A {best[0]:.1f} us ({fl / best[0] / 1e6:.0f} TF/s)  best B {best[1]:.1f} us ({fl / best[1] / 1e6:.0f} TF/s)")
    for code, t in rows:
        print(f"    kern{code & 3} bm{(code >> 2) & 3} bn{(code >> 4) & 3}: A {t[0]:7.1f}  B {t[1]:7.1f}")


if __name__ == "__main__":
    case(32, 22, 22, 1024, 416, 1, 0)
    case(32, 22, 22, 416, 1024, 1, 0)
    case(32, 44, 44, 512, 208, 1, 0)
    case(32, 88, 88, 256, 104, 1, 0)
    case(32, 11, 11, 2048, 832, 1, 0)
    case(32, 22, 22, 104, 104, 3, 1)
    case(32, 44, 44, 56, 56, 3, 1)
    case(32, 11, 11, 256, 256, 5, 2)
    case(8, 64, 64, 4096, 4096, 1, 0)
