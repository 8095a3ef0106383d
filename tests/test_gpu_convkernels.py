"""The conv GEMM kernels one by one (C ABI, forced kernel / tile codes): the register-staged kernel, the LDS-DMA kernel with 3 and 2 ring stages
(activation operand through a buffer descriptor: 32-bit offsets, hardware zero fill for padding taps and missing rows / channel chunks),
and every tile shape must produce the SAME bf16 tensor bit for bit, forward gather and transposed (dgrad) gather, and that tensor must be the
convolution torch computes in float64 from the same bf16 operands, up to one rounding of the output.  Geometries: 1x1, 3x3, 1xk / kx1, 5x5,
dilation 3 / 5 / 7, stride 2 (forward and its dgrad), row counts that do not fill a tile, channel counts that straddle a 64-wide K-step."""
import ctypes as C
import os, sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
dev = "cuda"

# N, H, W, Cin, Cout, KH, KW, stride, pad_h, pad_w, dil
GEOMS = [
    (2, 11, 13, 72, 40, 1, 1, 1, 0, 0, 1),
    (2, 11, 13, 104, 104, 3, 3, 1, 1, 1, 1),
    (3, 9, 10, 24, 56, 3, 3, 1, 1, 1, 1),
    (2, 12, 12, 32, 32, 3, 3, 1, 3, 3, 3),
    (1, 15, 9, 32, 32, 3, 3, 1, 7, 7, 7),
    (2, 10, 14, 32, 32, 1, 7, 1, 0, 3, 1),
    (2, 10, 14, 32, 32, 5, 1, 1, 2, 0, 1),
    (1, 9, 9, 40, 48, 5, 5, 1, 2, 2, 1),
    (2, 16, 18, 56, 56, 3, 3, 2, 1, 1, 1),
    (2, 17, 15, 8, 32, 3, 3, 2, 1, 1, 1),
    (1, 5, 7, 200, 136, 1, 1, 1, 0, 0, 1),
    (2, 40, 36, 32, 32, 3, 3, 1, 1, 1, 1),          # many tiles per image
    (3, 20, 18, 56, 56, 3, 3, 1, 1, 1, 1),          # a 64-wide K-step straddles two taps
    (5, 30, 30, 16, 24, 5, 5, 1, 2, 2, 1),          # 25 taps, four taps per K-step
]


def _rup(v, m):
    return (v + m - 1) // m * m


@pytest.mark.parametrize("geom", GEOMS)
@pytest.mark.parametrize("transposed", [0, 1])
def test_conv_kernels_agree_and_match_float64(geom, transposed):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pn2 import capi
    from pn2.capi import call, BF16
    N, H, W, Cin, Cout, KH, KW, s, ph, pw, dil = geom
    OH = (H + 2 * ph - dil * (KH - 1) - 1) // s + 1
    OW = (W + 2 * pw - dil * (KW - 1) - 1) // s + 1
    g = torch.Generator(device="cpu").manual_seed(N * 1000 + H * 10 + KH + transposed)
    w = (torch.randn(Cout, Cin, KH, KW, generator=g) * 0.2).bfloat16()
    taps = KH * KW
    P = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    d = capi.ConvDesc()
    d.KH, d.KW, d.stride, d.pad_h, d.pad_w, d.dil_h, d.dil_w = KH, KW, s, ph, pw, dil, dil
    if not transposed:
        x = torch.randn(N, H, W, Cin, generator=g).bfloat16()
        ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), None, s, (ph, pw), dil).permute(0, 2, 3, 1)          # [N,OH,OW,Cout]
        Kp = _rup(taps * Cin, 128)
        wp = torch.zeros(_rup(Cout, 128), Kp, dtype=torch.bfloat16)
        wp[:Cout, :taps * Cin] = w.permute(0, 2, 3, 1).reshape(Cout, taps * Cin)          # wp[co][tap*Cin + ci]
        d.N, d.H, d.W, d.OH, d.OW = N, H, W, OH, OW
        d.Cin_p, d.ld_in, d.Cout, d.ld_out = Cin, Cin, Cout, Cout
        src, n_out, M = x, Cout, N * OH * OW
    else:
        dy = torch.randn(N, OH, OW, Cout, generator=g).bfloat16()
        ref = F.conv_transpose2d(dy.double().permute(0, 3, 1, 2), w.double(), None, s, (ph, pw),
                                 (H - ((OH - 1) * s - 2 * ph + dil * (KH - 1) + 1), W - ((OW - 1) * s - 2 * pw + dil * (KW - 1) + 1)), 1, dil).permute(0, 2, 3, 1)   # [N,H,W,Cin]
        Kp = _rup(taps * Cout, 128)
        wp = torch.zeros(_rup(Cin, 128), Kp, dtype=torch.bfloat16)
        wp[:Cin, :taps * Cout] = w.permute(1, 2, 3, 0).reshape(Cin, taps * Cout)          # wp[ci][tap*Cout + co]
        d.N, d.H, d.W, d.OH, d.OW = N, OH, OW, H, W
        d.Cin_p, d.ld_in, d.Cout, d.ld_out = Cout, Cout, Cin, Cin
        src, n_out, M = dy, Cin, N * H * W
    d.transposed, d.Kp = transposed, Kp
    src_g, wp_g = src.to(dev), wp.to(dev)
    outs, stats = {}, {}
    for kern in (1, 2, 3):
        for bm in (1, 2):
            for bn in (1, 2, 3):
                if (bn == 3 and n_out <= 64) or (bn == 2 and n_out <= 32):
                    continue
                code = kern | (bm << 2) | (bn << 4)
                d.flags = (code << 8) | (0 if transposed else capi.CONV_STATS)
                tm = 64 * bm
                out = torch.full((M, n_out), float("nan"), dtype=torch.bfloat16, device=dev)
                if transposed:
                    call.pn2_conv_gemm(BF16, P(src_g), P(wp_g), P(out), C.c_void_p(0), C.c_void_p(0), C.byref(d), st)
                else:
                    nblk = (M + tm - 1) // tm
                    ps = torch.full((nblk, n_out), float("nan"), device=dev); pq = torch.full((nblk, n_out), float("nan"), device=dev)
                    call.pn2_conv_gemm(BF16, P(src_g), P(wp_g), P(out), P(ps), P(pq), C.byref(d), st)
                    stats.setdefault(tm, {})[code] = (ps, pq)
                outs[code] = out
    torch.cuda.synchronize()
    first = next(iter(outs.values()))
    for code, o in outs.items():
        assert torch.equal(o.view(torch.int16), first.view(torch.int16)), f"kernel/tile code {code:#x} differs from code {next(iter(outs)):#x}"
    # BatchNorm partials (mean, M2 per row block and channel): every bf16 kernel takes them from the STORED tile on the matrix cores, so kernels with
    # the same row-block height agree bit for bit (the table-driven launches swap kernels under the engine), and their Chan merge is the
    # mean / biased variance of the stored tensor
    o64 = first.double().cpu()
    for tm, per in stats.items():
        c0, (ps0, pq0) = next(iter(per.items()))
        for code, (ps, pq) in per.items():
            assert torch.equal(ps, ps0) and torch.equal(pq, pq0), f"statistics of code {code:#x} differ from code {c0:#x}"
        nblk = ps0.shape[0]
        n_t = torch.full((nblk,), float(tm), dtype=torch.float64); n_t[-1] = M - (nblk - 1) * tm
        mean_t, m2_t = ps0.double().cpu(), pq0.double().cpu()
        mean = (mean_t * n_t[:, None]).sum(0) / M
        var = (m2_t.sum(0) + (n_t[:, None] * (mean_t - mean) ** 2).sum(0)) / M
        smean, svar = o64.mean(0), o64.var(0, unbiased=False)
        assert float(((mean - smean).abs() / svar.sqrt()).max()) < 2e-5
        assert float(((var - svar).abs() / svar).max()) < 5e-5
    got = first.double().cpu().reshape(ref.shape)
    # fp32 accumulation of exact bf16 products, one bf16 rounding of the result: within one output ulp of the float64 convolution
    tol = ref.abs() * 2.0 ** -8 + 1e-3 * float(ref.abs().max())
    assert bool(((got - ref).abs() <= tol).all()), float(((got - ref).abs() - tol).max())


WGEOMS = [
    (2, 11, 13, 272, 136, 1, 1, 1, 0, 0, 1),        # co tile 128 x 2, Kp = 384: a full and a half-empty 256-wide contraction tile
    (2, 9, 10, 40, 200, 3, 3, 1, 1, 1, 1),          # 3x3, co tile 128, Kp = 384
    (2, 11, 13, 72, 40, 1, 1, 1, 0, 0, 1),
    (2, 11, 13, 104, 104, 3, 3, 1, 1, 1, 1),
    (2, 12, 12, 32, 32, 3, 3, 1, 3, 3, 3),
    (2, 10, 14, 32, 32, 1, 7, 1, 0, 3, 1),
    (2, 16, 18, 56, 56, 3, 3, 2, 1, 1, 1),
    (3, 20, 20, 136, 200, 1, 1, 1, 0, 0, 1),
]


@pytest.mark.parametrize("geom", WGEOMS)
def test_wgrad_kernels_agree_and_match_float64(geom):
    """Weight gradient: register-staged kernel and LDS-DMA kernel (both operands through buffer descriptors), 1 and several pixel splits, slab
    reduction into the OIHW fp32 gradient - against torch float64 autograd on the same bf16 operands."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pn2 import capi
    from pn2.capi import call, BF16
    N, H, W, Cin, Cout, KH, KW, s, ph, pw, dil = geom
    OH = (H + 2 * ph - dil * (KH - 1) - 1) // s + 1
    OW = (W + 2 * pw - dil * (KW - 1) - 1) // s + 1
    g = torch.Generator(device="cpu").manual_seed(7 * N + H + KW)
    x = torch.randn(N, H, W, Cin, generator=g).bfloat16()
    dy = torch.randn(N, OH, OW, Cout, generator=g).bfloat16()
    wref = torch.zeros(Cout, Cin, KH, KW, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double().permute(0, 3, 1, 2), wref, None, s, (ph, pw), dil).backward(dy.double().permute(0, 3, 1, 2))
    ref = wref.grad
    P = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    tco = call.pn2_wgrad_tile_co(Cout)
    wd = capi.WgradDesc()
    wd.N, wd.H, wd.W, wd.OH, wd.OW = N, H, W, OH, OW
    wd.Cin_p, wd.ld_x, wd.Cout_p, wd.ld_dy = Cin, Cin, Cout, Cout
    wd.KH, wd.KW, wd.stride, wd.pad_h, wd.pad_w, wd.dil_h, wd.dil_w = KH, KW, s, ph, pw, dil, dil
    wd.Rp, wd.Kp = _rup(Cout, tco), _rup(KH * KW * Cin, 128)
    rd = capi.PackDesc()
    rd.Cout, rd.Cin, rd.KH, rd.KW = Cout, Cin, KH, KW
    rd.Cout_p, rd.gw_out, rd.gwp_out, rd.Cin_p, rd.gw_in, rd.gwp_in = Cout, Cout, Cout, Cin, Cin, Cin
    rd.Rp, rd.Kp, rd.transposed = wd.Rp, wd.Kp, 0
    xg, dyg = x.to(dev), dy.to(dev)
    res = {}
    for tune in (1, 2, 3):          # 3: the LDS-DMA kernel with 128 x 256 tiles (co tile 128 and >= 256 contraction columns; otherwise it IS kernel 2)
        for ns in (1, 3):
            wd.tune = tune
            slab = torch.full((ns, wd.Rp, wd.Kp), float("nan"), dtype=torch.float32, device=dev)
            gw = torch.full((Cout, Cin, KH, KW), float("nan"), dtype=torch.float32, device=dev)
            call.pn2_conv_wgrad(BF16, P(dyg), P(xg), P(slab), C.byref(wd), ns, st)
            call.pn2_wgrad_reduce(P(slab), P(gw), C.byref(rd), ns, 0, st)
            res[(tune, ns)] = gw.double().cpu()
    scale = float(ref.abs().max())
    for k, v in res.items():
        assert float((v - ref).abs().max()) <= 2e-5 * scale, (k, float((v - ref).abs().max()) / scale)


def test_table_driven_conv_launch_matches_single_launches_bitwise():
    """pn2_conv_gemm_multi (the lock-step launches of Engine.lockstep: one LDS-DMA kernel for every job of a tile shape) against one pn2_conv_gemm per job
    with each job's OWN kernel code (register-staged, 2- and 3-stage LDS-DMA): outputs and BatchNorm partials bit for bit.  Jobs as in the RFB tails:
    1xk / kx1 / dilated 3x3, inputs and outputs that are channel slices of wider buffers, row counts far below a tile."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pn2 import capi
    from pn2.capi import call, BF16
    from pn2.engine import _job_table, _p
    P = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cpu").manual_seed(77)

    def mk(N, H, W, Cin_p, Cout, KH, KW, ph, pw, code, dil=1, ld_in=None, ld_out=None, off_in=0, off_out=0):
        d = capi.ConvDesc()
        ld_in, ld_out = ld_in or Cin_p, ld_out or Cout
        d.N, d.H, d.W, d.OH, d.OW = N, H, W, H, W
        d.Cin_p, d.ld_in, d.Cout, d.ld_out = Cin_p, ld_in, Cout, ld_out
        d.KH, d.KW, d.stride, d.pad_h, d.pad_w, d.dil_h, d.dil_w = KH, KW, 1, ph, pw, dil, dil
        d.transposed, d.Kp = 0, _rup(KH * KW * Cin_p, 128)
        d.flags = capi.CONV_STATS | (code << 8)
        M = N * H * W
        x = torch.randn(M, ld_in, generator=g).bfloat16().to(dev)[:, off_in:]
        wp = (torch.randn(_rup(Cout, 128), d.Kp, generator=g) * 0.1).bfloat16().to(dev)
        return d, x, wp, M, ld_out, off_out

    for bmc, bnc, bm in ((1, 1, 64), (2, 1, 128), (1, 2, 64)):
        codes = [k | (bmc << 2) | (bnc << 4) for k in (1, 2, 3)]
        co = 32 if bnc == 1 else 64
        jobs = [mk(2, 12, 12, 32, co, 1, 3, 0, 1, codes[0], ld_in=224, off_in=32), mk(2, 12, 12, 32, co, 3, 3, 3, 3, codes[1], dil=3, ld_out=256, off_out=64),
                mk(2, 6, 6, 32, co, 3, 3, 5, 5, codes[2], dil=5, ld_out=256, off_out=128), mk(2, 3, 3, 32, co, 1, 7, 0, 3, codes[0], ld_in=416, off_in=96),
                mk(2, 6, 6, 64, co, 5, 1, 2, 0, codes[0]), mk(3, 24, 24, 64, co, 1, 1, 0, 0, codes[1])]
        single, multi, structs, nbs = [], [], [], []
        for d, x, wp, M, ld_out, off_out in jobs:
            nb = (M + bm - 1) // bm
            o = torch.zeros(M, ld_out, dtype=torch.bfloat16, device=dev)[:, off_out:]; ps = torch.zeros(nb, d.Cout, device=dev); pq = torch.zeros(nb, d.Cout, device=dev)
            call.pn2_conv_gemm(BF16, P(x), P(wp), P(o), P(ps), P(pq), C.byref(d), st)
            single.append((o, ps, pq))
            o2 = torch.zeros(M, ld_out, dtype=torch.bfloat16, device=dev)[:, off_out:]; ps2 = torch.zeros_like(ps); pq2 = torch.zeros_like(pq)
            j = capi.ConvJob()
            j.in_, j.wp, j.out, j.psum, j.psq = x.data_ptr(), wp.data_ptr(), o2.data_ptr(), ps2.data_ptr(), pq2.data_ptr()
            C.memmove(C.byref(j.d), C.byref(d), C.sizeof(d))
            tile = call.pn2_conv_gemm_tile(BF16, C.byref(j.d))
            assert tile >> 8 == bm
            structs.append(j); nbs.append(call.pn2_conv_gemm_job_blocks(BF16, C.byref(j), tile >> 8, tile & 255)); multi.append((o2, ps2, pq2))
        table, bstart, total = _job_table(capi.ConvJob, structs, nbs)
        call.pn2_conv_gemm_multi(BF16, tile >> 8, tile & 255, 0, _p(table), _p(bstart), len(structs), total, st)
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(zip(single, multi)):
            assert torch.equal(a[0], b[0]), (bm, i, "output")
            assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), (bm, i, "statistics")


KS_GEOMS = [  # long contractions (>= 6 K-steps of 64) where the intra-workgroup split-K kernels apply; few tiles, ragged last tile, uneven K shares
    (2, 11, 13, 104, 104, 3, 3, 1, 1, 1, 1),
    (1, 9, 9, 40, 48, 5, 5, 1, 2, 2, 1),            # 1000-long contraction: 16 K-steps, 4 groups x 4
    (2, 11, 11, 208, 208, 3, 3, 1, 1, 1, 1),        # 30 K-steps: 15 + 15 / 8 + 8 + 8 + 6
    (1, 5, 7, 456, 136, 1, 1, 1, 0, 0, 1),          # 1x1, 8 K-steps (the last one half empty)
    (3, 20, 18, 56, 72, 3, 3, 1, 1, 1, 1),          # a K-step straddles two taps at a group boundary
    (2, 16, 18, 56, 56, 3, 3, 2, 1, 1, 1),          # stride 2 and its transposed gather
]


@pytest.mark.parametrize("geom", KS_GEOMS)
@pytest.mark.parametrize("transposed", [0, 1])
def test_intra_workgroup_split_k_kernels_match_float64_and_the_plain_kernel(geom, transposed):
    """conv_dma_gemm_ks (tuning-code bits 6 / 7: two / four K groups of four waves share a tile's K loop, accumulators meet in LDS): the convolution of the
    same bf16 operands in float64 up to one rounding of the output, BatchNorm partials consistent with the STORED tile, and - another fp32 summation order, not
    another result - the plain kernel's tensor up to single bf16 roundings on a few elements."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pn2 import capi
    from pn2.capi import call, BF16
    N, H, W, Cin, Cout, KH, KW, s, ph, pw, dil = geom
    OH = (H + 2 * ph - dil * (KH - 1) - 1) // s + 1
    OW = (W + 2 * pw - dil * (KW - 1) - 1) // s + 1
    g = torch.Generator(device="cpu").manual_seed(N * 1000 + H * 10 + KH + transposed + 7)
    w = (torch.randn(Cout, Cin, KH, KW, generator=g) * 0.2).bfloat16()
    taps = KH * KW
    P = lambda t: C.c_void_p(t.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    d = capi.ConvDesc()
    d.KH, d.KW, d.stride, d.pad_h, d.pad_w, d.dil_h, d.dil_w = KH, KW, s, ph, pw, dil, dil
    if not transposed:
        x = torch.randn(N, H, W, Cin, generator=g).bfloat16()
        ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), None, s, (ph, pw), dil).permute(0, 2, 3, 1)
        Kp = _rup(taps * Cin, 128)
        wp = torch.zeros(_rup(Cout, 128), Kp, dtype=torch.bfloat16)
        wp[:Cout, :taps * Cin] = w.permute(0, 2, 3, 1).reshape(Cout, taps * Cin)
        d.N, d.H, d.W, d.OH, d.OW = N, H, W, OH, OW
        d.Cin_p, d.ld_in, d.Cout, d.ld_out = Cin, Cin, Cout, Cout
        src, n_out, M = x, Cout, N * OH * OW
    else:
        dy = torch.randn(N, OH, OW, Cout, generator=g).bfloat16()
        ref = F.conv_transpose2d(dy.double().permute(0, 3, 1, 2), w.double(), None, s, (ph, pw),
                                 (H - ((OH - 1) * s - 2 * ph + dil * (KH - 1) + 1), W - ((OW - 1) * s - 2 * pw + dil * (KW - 1) + 1)), 1, dil).permute(0, 2, 3, 1)
        Kp = _rup(taps * Cout, 128)
        wp = torch.zeros(_rup(Cin, 128), Kp, dtype=torch.bfloat16)
        wp[:Cin, :taps * Cout] = w.permute(1, 2, 3, 0).reshape(Cin, taps * Cout)
        d.N, d.H, d.W, d.OH, d.OW = N, OH, OW, H, W
        d.Cin_p, d.ld_in, d.Cout, d.ld_out = Cout, Cout, Cin, Cin
        src, n_out, M = dy, Cin, N * H * W
    d.transposed, d.Kp = transposed, Kp
    src_g, wp_g = src.to(dev), wp.to(dev)
    tol = ref.abs() * 2.0 ** -8 + 1e-3 * float(ref.abs().max())

    def run(code):
        d.flags = (code << 8) | (0 if transposed else capi.CONV_STATS)
        tm = 64 * ((code >> 2) & 3)
        out = torch.full((M, n_out), float("nan"), dtype=torch.bfloat16, device=dev)
        nblk = (M + tm - 1) // tm
        ps = torch.full((nblk, n_out), float("nan"), device=dev); pq = torch.full((nblk, n_out), float("nan"), device=dev)
        call.pn2_conv_gemm(BF16, P(src_g), P(wp_g), P(out), C.c_void_p(0) if transposed else P(ps), C.c_void_p(0) if transposed else P(pq), C.byref(d), st)
        torch.cuda.synchronize()
        return out, ps, pq, tm
    ran = 0
    for kern in (2, 3):
        for bm in (1, 2):
            for bn in (2, 3):
                if bn == 3 and n_out <= 64:
                    continue
                base = kern | (bm << 2) | (bn << 4)
                plain = run(base)[0]
                for bit in (0x40, 0x80):
                    if bit == 0x80 and not (bm == 1 and bn == 2):
                        continue
                    try:
                        out, ps, pq, tm = run(base | bit)
                    except RuntimeError as e:          # a tile whose rings do not fit 160 KB of LDS (128 x 128 with three stages)
                        assert "status -4" in str(e), e
                        continue
                    ran += 1
                    got = out.double().cpu().reshape(ref.shape)
                    assert bool(((got - ref).abs() <= tol).all()), (hex(base | bit), float(((got - ref).abs() - tol).max()))
                    dd = (out.float() - plain.float()).abs()
                    assert float((dd > 0).float().mean()) < 0.02 and bool((dd <= plain.float().abs() * 2.0 ** -7 + 1e-6).all()), hex(base | bit)
                    if not transposed:          # partials = mean / M2 of the stored tile rows
                        o64 = out.double().cpu()
                        nblk = ps.shape[0]
                        n_t = torch.full((nblk,), float(tm), dtype=torch.float64); n_t[-1] = M - (nblk - 1) * tm
                        mean_t, m2_t = ps.double().cpu(), pq.double().cpu()
                        mean = (mean_t * n_t[:, None]).sum(0) / M
                        var = (m2_t.sum(0) + (n_t[:, None] * (mean_t - mean) ** 2).sum(0)) / M
                        smean, svar = o64.mean(0), o64.var(0, unbiased=False)
                        assert float(((mean - smean).abs() / svar.sqrt()).max()) < 2e-5 and float(((var - svar).abs() / svar).max()) < 5e-5
    assert ran >= 6
