#!/usr/bin/env python3
"""Micro-benchmark of the fused DSRA tail (pn2_dsra_tail_fwd / _bwd) at the headline geometry (bs=32, 352x352, 8 lateral maps at
44/22/11/44 squared): band kernels (default) against the row-per-block kernels (PN2_TAIL_BAND=0), with the two paths compared
against each other (maps, loss sums, low-res gradients).  GPU box only.  Usage: tail_micro.py [N] [S] [align_corners]"""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
import torch
from pn2 import capi
from pn2.capi import call

P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def timeit(fn, reps=50, cold=None):
    for _ in range(3):
        fn()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    torch.cuda.synchronize()
    for e0, e1 in evs:
        if cold is not None:
            cold.add_(1.0)            # evict L2 / MALL: inside a step the tail runs on cold operands
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    ts = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    return ts[len(ts) // 2] * 1e3


def run(N, S, ac, band):
    """band: 0 row kernels, 1 band kernels (one block per group of <= 2 maps), 3 = the one-pass entry pn2_dsra_tail_fwd_bwd"""
    os.environ["PN2_TAIL_BAND"] = str(band) if band < 2 else "2"
    dev = "cuda"
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cpu").manual_seed(5)
    sizes = [S // 8, S // 16, S // 32, S // 8]
    Pn = 4
    srcs = [(torch.randn(N, h, h, generator=g) * 2).to(dev) for h in sizes + sizes]
    dsrcs = [torch.full_like(s, 0.25) for s in srcs]
    mask = (torch.rand(N, S, S, generator=g) < 0.3).float().to(dev)
    weit = torch.empty_like(mask)
    call.pn2_loss_weights(P(mask), P(weit), N, S, S, 31, st)
    d = capi.TailDesc()
    d.N, d.OH, d.OW, d.P, d.align_corners = N, S, S, Pn, ac
    for j, (s, ds) in enumerate(zip(srcs, dsrcs)):
        m = d.maps[j]
        h = s.shape[1]
        m.src, m.dsrc, m.h, m.w = s.data_ptr(), ds.data_ptr(), h, h
        m.rh = m.rw = ((h - 1) / (S - 1)) if ac else h / S
        m.accumulate = 1 if j % 3 == 0 else 0
    nb = call.pn2_dsra_tail_blocks(S)
    lat = torch.empty(2 * Pn, N, S, S, device=dev)
    partial = torch.empty(Pn, N, nb, 5, device=dev)
    sums = torch.empty(Pn, N, 4, device=dev); wsum = torch.empty(N, device=dev); loss = torch.empty(Pn + 1, device=dev)
    need = int(call.pn2_dsra_tail_scratch(C.byref(d)))
    scratch = torch.empty(max(need, 1), device=dev)
    fwd = lambda: call.pn2_dsra_tail_fwd(C.byref(d), P(lat), P(mask), P(weit), P(partial), P(sums), P(wsum), P(loss), st)
    bwd = lambda: call.pn2_dsra_tail_bwd(C.byref(d), P(mask), P(weit), P(wsum), P(sums), 1.0, P(scratch) if need else None, need, st)
    per = torch.empty(Pn, N, device=dev)
    isum = torch.zeros(Pn * N * 10, dtype=torch.int64, device=dev) if os.environ.get("PN2_TAIL_ISUM", "1") == "1" else None
    def both():          # (in the trainer the clear rides on the weights launch; here it is a fill in front of the timed region's kernels)
        if isum is not None:
            isum.zero_()
        call.pn2_dsra_tail_fwd_bwd(C.byref(d), P(lat), P(mask), P(weit), P(partial), P(sums), P(wsum), P(per), P(loss), 1.0, P(scratch), need, P(isum), st)
    alg = 17 * N * S * S * 4
    cold = torch.zeros(256 << 20, device=dev)
    if band == 3:
        assert int(call.pn2_dsra_tail_fused_ok(C.byref(d))) == 1
        need = int(call.pn2_dsra_tail_fused_scratch(C.byref(d)))
        scratch = torch.empty(need, device=dev)
        for ds in dsrcs:
            ds.fill_(0.25)
        both()
        torch.cuda.synchronize()
        out = (lat.clone(), sums.clone(), loss.clone(), [x.clone() for x in dsrcs])
        t, tc = timeit(both), timeit(both, cold=cold)
        print(f"N={N} S={S} ac={ac} one-pass entry (3 launches): fwd+bwd {t:7.1f} us (cold {tc:7.1f})  -> {alg / t / 1e6:6.2f} TB/s algorithmic, cold {alg / tc / 1e6:6.2f} TB/s")
        return out
    fwd()
    for ds in dsrcs:
        ds.fill_(0.25)
    bwd()
    torch.cuda.synchronize()
    out = (lat.clone(), sums.clone(), loss.clone(), [x.clone() for x in dsrcs])
    tf, tb = timeit(fwd), timeit(bwd)
    tfc, tbc = timeit(fwd, cold=cold), timeit(bwd, cold=cold)
    print(f"N={N} S={S} ac={ac} band={int(band)} scratch={need}: fwd {tf:7.1f} us  bwd {tb:7.1f} us (cold {tfc:7.1f} / {tbc:7.1f})  "
          f"-> {alg / (tf + tb) / 1e6:6.2f} TB/s algorithmic (17*S per image), cold {alg / (tfc + tbc) / 1e6:6.2f} TB/s")
    return out


def relmax(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 352
    ac = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    a = run(N, S, ac, 0)
    for mode in (1, 3):
        b = run(N, S, ac, mode)
        print(f"mode {mode} vs row kernels: maps relmax", relmax(b[0], a[0]), "sums relmax", relmax(b[1], a[1]), "loss", a[2].tolist(), b[2].tolist(),
              " dsrc relmax", max(relmax(x, y) for x, y in zip(b[3], a[3])))
