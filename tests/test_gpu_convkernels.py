"""The conv GEMM kernels one by one (C ABI, forced kernel / tile codes): the register-staged kernel, the LDS-DMA kernel with 3 and 2 ring stages
(activation operand through a buffer descriptor: 32-bit offsets, hardware zero fill for padding taps and missing rows / channel chunks), the
persistent kernel with the weight panel resident in LDS, and every tile shape must produce the SAME bf16 tensor bit for bit, forward gather and transposed (dgrad) gather, and that tensor must be the
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
    outs = {}
    for kern in (1, 2, 3, 2 | 0x40, 3 | 0x40):          # 0x40: persistent workgroups with the weight panel resident in LDS (conv_bres_gemm)
        for bm in (1, 2):
            for bn in (1, 2, 3):
                if (bn == 3 and n_out <= 64) or (bn == 2 and n_out <= 32):
                    continue
                code = kern | (bm << 2) | (bn << 4)
                d.flags = code << 8
                out = torch.full((M, n_out), float("nan"), dtype=torch.bfloat16, device=dev)
                call.pn2_conv_gemm(BF16, P(src_g), P(wp_g), P(out), C.c_void_p(0), C.c_void_p(0), C.byref(d), st)
                outs[code] = out
    torch.cuda.synchronize()
    first = next(iter(outs.values()))
    for code, o in outs.items():
        assert torch.equal(o.view(torch.int16), first.view(torch.int16)), f"kernel/tile code {code:#x} differs from code {next(iter(outs)):#x}"
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
