#!/usr/bin/env python3
"""Phase breakdown of ONE conv launch from in-kernel s_memtime stamps (debug library: make -C pranet-v2_amd/csrc stamp).  GPU box only.
   python tools/stamp_micro.py [name filter ...]        (CODE=<int> forces a tile code, default: the shipped table's)
Shapes are the benchmark's own (keys of tuned_gfx950.json)."""
import ast, ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("PN2_LIB", os.path.join(ROOT, "pranet-v2_amd", "csrc", "libpn2_stamp.so"))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
import numpy as np, torch
from pn2 import capi
from pn2.capi import call, BF16
from pn2.engine import _thrash

lib = C.CDLL(os.environ["PN2_LIB"])
lib.pn2_debug_stamps.argtypes = [C.c_void_p, C.c_int]
TABLE = {ast.literal_eval(k): v for k, v in json.load(open(os.path.join(ROOT, "pranet-v2_amd", "pn2", "tuned_gfx950.json"))).items()}
rup = lambda v, m: (v + m - 1) // m * m
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
SL = 16
NAMES = ["prologue", "issue", "1st-land", "k-loop", "epi-pre", "stats+stage", "merge", "store-issue", "(ret)"]


def run(key, code, cold):
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
    nb64 = (M + 63) // 64
    psum = torch.empty(nb64, Cout, device="cuda"); psq = torch.empty(nb64, Cout, device="cuda")
    d.flags = (capi.CONV_STATS if not tr else 0) | (code << 8)
    bm = 64 if ((code >> 2) & 3) == 1 else 128
    bn = {1: 32, 2: 64, 3: 128}.get((code >> 4) & 3, 128)
    us = []
    for rep in range(4):
        if cold: _thrash()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call.pn2_conv_gemm(BF16, P(x), P(wp), P(out), P(psum) if not tr else P(None), P(psq) if not tr else P(None), C.byref(d), st())
        e1.record(); torch.cuda.synchronize()
        us.append(e0.elapsed_time(e1) * 1e3)
    analyse(us, M, Cout, bm, bn, f"{'dgrad' if tr else 'fwd  '} {Cin_p}->{Cout} k{KH}x{KW} s{s} M={M} code {code:#04x} tile {bm}x{bn} ksteps {(taps * Cin_p + 63) // 64}  {'cold' if cold else 'warm'}",
            2 * M * Cout * Cin_p * taps, (M * Cout + N * H * W * Cin_p) * 2)


def analyse(us, M, Cout, bm, bn, title, flops, byts):
    nblk = min(((M + bm - 1) // bm) * ((Cout + bn - 1) // bn), 65536)
    buf = np.zeros((nblk, SL), dtype=np.uint64)
    rc = lib.pn2_debug_stamps(buf.ctypes.data, nblk)
    assert rc == 0, rc
    t = buf[:, :10].astype(np.int64)
    if (t[:, 9] == 0).any():
        print(f"   ({int((t[:, 9] == 0).sum())} of {nblk} workgroups left no stamp: the tile code does not map to {bm}x{bn}?)")
        t = t[t[:, 9] != 0]
    if os.environ.get("RAW"):
        print(buf[:4]); print("col min", buf.min(0)); print("col max", buf.max(0))
    # s_memtime bases differ between XCDs / CUs: only differences inside one CU are used.  Tick rate: the stamps of the busiest CUs span the launch.
    keep = buf[:, 9] != 0
    hw, xcc = buf[keep, 10].astype(np.int64), buf[keep, 11].astype(np.int64) & 15
    cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
    spans = []
    for c in np.unique(cu):
        sel = cu == c
        spans.append(t[sel, 9].max() - t[sel, 0].min())
    span = float(np.median(spans))
    tick_per_us = span / us[-1]
    t0 = 0
    phs = np.diff(t, axis=1)                      # [nblk, 9]
    tot = t[:, 9] - t[:, 0]
    ncu = len(np.unique(cu))
    print(f"\n{title}: {us[-1]:.1f} us "
          f"({flops / us[-1] / 1e6:.0f} TF/s, {byts / us[-1] / 1e6:.2f} TB/s algorithmic), {nblk} WGs on {ncu} CUs, {tick_per_us:.0f} ticks/us")
    print(f"   WG lifetime median {np.median(tot) / tick_per_us:6.2f} us  p90 {np.percentile(tot, 90) / tick_per_us:6.2f};  mean WGs alive per CU {tot.sum() / span / max(ncu, 1):.2f}  (CU span min {min(spans) / tick_per_us:.1f} max {max(spans) / tick_per_us:.1f} us)")
    line = "   "
    for i in range(8):
        line += f"{NAMES[i]} {np.median(phs[:, i]) / tick_per_us:5.2f}/{np.percentile(phs[:, i], 90) / tick_per_us:5.2f}  "
    print(line + " (median/p90 us)")
    b = buf[keep].astype(np.int64)
    print(f"   inside stats+stage: statistics {np.median(b[:, 12] - b[:, 5]) / tick_per_us:5.2f}  cvt + LDS stores {np.median(b[:, 13] - b[:, 12]) / tick_per_us:5.2f}  barrier {np.median(b[:, 6] - b[:, 13]) / tick_per_us:5.2f}")
    if b[:, 14].any() and not b[:, 15].any():
        print(f"   inside store-issue: copy-out {np.median(b[:, 14] - b[:, 7]) / tick_per_us:5.2f}  statistics on the matrix cores + partial stores {np.median(b[:, 8] - b[:, 14]) / tick_per_us:5.2f}")
    if b[:, 14].any() and b[:, 15].any():
        print(f"   inside store-issue (BN-backward epilogue): targets {np.median(b[:, 14] - b[:, 7]) / tick_per_us:5.2f}  barrier + LDS sums {np.median(b[:, 15] - b[:, 14]) / tick_per_us:5.2f}  partial stores {np.median(b[:, 8] - b[:, 15]) / tick_per_us:5.2f}")
    drain = phs[:, 8]
    print(f"   store drain (s_waitcnt vmcnt(0) after the last store) {np.median(drain) / tick_per_us:5.2f}/{np.percentile(drain, 90) / tick_per_us:5.2f}")


def run_ep(key, code, cold):
    """dgrad with the BatchNorm-backward epilogue (pn2_conv_gemm_ep), operands as in tests/test_gpu_baseline_shapes.py"""
    _, N, H, W, OH, OW, Cin_p, ld_in, Cout, KH, KW, s, ph, pw, dh, dw, tr, _x, amode, bmode, dual, accf = key
    taps, M = KH * KW, N * OH * OW
    d = capi.ConvDesc()
    d.N, d.H, d.W, d.OH, d.OW = N, H, W, OH, OW
    d.Cin_p, d.ld_in, d.Cout, d.ld_out = Cin_p, ld_in, Cout, Cout
    d.KH, d.KW, d.stride, d.pad_h, d.pad_w, d.dil_h, d.dil_w = KH, KW, s, ph, pw, dh, dw
    d.transposed, d.Kp = 1, rup(taps * Cin_p, 128)
    dy = torch.randn(N * H * W, ld_in, device="cuda").bfloat16()
    wp = (torch.randn(rup(Cout, 128), d.Kp, device="cuda") * 0.05).bfloat16()
    out = torch.randn(M, Cout, device="cuda").bfloat16()
    d.flags = (capi.CONV_ACCUM if accf else 0) | (code << 8)
    bm = 64 if ((code >> 2) & 3) == 1 else 128
    bn = {1: 32, 2: 64, 3: 128}.get((code >> 4) & 3, 128)
    nb = (M + bm - 1) // bm
    ep = capi.ConvEp(); keep = []
    def target(t, mode):
        raw = torch.randn(M, Cout, device="cuda").bfloat16(); par = torch.rand(4, Cout, device="cuda") + 0.5
        y = torch.randn(M, Cout, device="cuda").bfloat16() if mode & 4 else None
        p1 = torch.empty(nb, Cout, device="cuda"); p2 = torch.empty(nb, Cout, device="cuda")
        t.mode = mode; t.raw, t.ld_raw = raw.data_ptr(), Cout
        if y is not None: t.y, t.ld_y = y.data_ptr(), Cout
        t.par, t.ps = par.data_ptr(), Cout
        t.p1, t.p2, t.ldp = p1.data_ptr(), p2.data_ptr(), Cout
        keep.extend([raw, par, y, p1, p2])
    target(ep.a, amode)
    if dual:
        target(ep.b, bmode)
        ob = torch.empty(M, Cout, device="cuda").bfloat16(); keep.append(ob)
        ep.b.out, ep.b.ld_out = ob.data_ptr(), Cout
    us = []
    for rep in range(4):
        if cold: _thrash()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call.pn2_conv_gemm_ep(BF16, P(dy), P(wp), P(out), C.byref(d), C.byref(ep), st())
        e1.record(); torch.cuda.synchronize()
        us.append(e0.elapsed_time(e1) * 1e3)
    nt = 2 if dual else 1
    analyse(us, M, Cout, bm, bn, f"dgrad+BN-bwd {Cin_p}->{Cout} k{KH}x{KW} s{s} M={M} code {code:#04x} tile {bm}x{bn} ksteps {(taps * Cin_p + 63) // 64} amode {amode} dual {dual} accum {accf} {'cold' if cold else 'warm'}",
            2 * M * Cout * Cin_p * taps, (N * H * W * Cin_p + M * Cout * (1 + nt + (1 if accf else 0) + nt)) * 2)


flt = sys.argv[1:]
keys = [k for k in TABLE if k[0] == "g" and len(k) == 17 and k[14] == 1]
keys.sort(key=lambda k: -(k[1] * k[4] * k[5]))
want = [("pw64-256", lambda k: k[9] == 1 and k[6] == 64 and k[8] == 256 and k[16] == 0),
        ("pw128-256", lambda k: k[9] == 1 and k[6] == 128 and k[8] == 256 and k[16] == 0),
        ("pw256-128", lambda k: k[9] == 1 and k[6] == 256 and k[8] == 128 and k[16] == 0),
        ("c3-32-32", lambda k: k[9] == 3 and k[10] == 3 and k[6] == 32 and k[8] == 32 and k[16] == 0 and k[11] == 1 and k[4] == 88),
        ("c3-56-56", lambda k: k[9] == 3 and k[10] == 3 and k[6] == 56 and k[8] == 56 and k[16] == 0 and k[11] == 1),
        ("pw512-224", lambda k: k[9] == 1 and k[6] == 512 and k[8] == 224 and k[16] == 0),
        ("stem32-64", lambda k: k[9] == 3 and k[6] == 32 and k[8] == 64 and k[4] == 176)]
if any(a.startswith("ep") for a in flt):
    eps = [k for k in TABLE if k[0] == "g" and len(k) == 22]
    if os.environ.get("K3"):                       # only the 3x3 dgrads (the Res2Net branch convs)
        eps = [k for k in eps if k[9] == 3]
    eps.sort(key=lambda k: -(k[1] * k[4] * k[5] * k[8]))
    for key in eps[:int(os.environ.get("NEP", "10"))]:
        print(f"==== ep {key}")
        run_ep(key, int(os.environ.get("CODE", TABLE[key])), False)
    sys.exit(0)
for name, f in want:
    if flt and not any(a in name for a in flt): continue
    ks = [k for k in keys if f(k)]
    if not ks: print("no key for", name); continue
    key = ks[0]
    code = int(os.environ.get("CODE", TABLE[key]))
    print(f"==== {name} {key}")
    for cold in (True, False):
        run(key, code, cold)
