"""The BENCHMARKED launches against an oracle: real layer shapes of BASELINE config 2 (PraNet-V2 / Res2Net-50, N = 32, 352 x 352, bf16) run
through the C ABI with the tile the SHIPPED tuning table (pn2/tuned_gfx950.json) selects for exactly that launch, checked against torch float64
on the CPU computed from the same bf16 operands.

The fixtures of the other tests are 96^2 / 64^2 networks and 11 tiny geometries: a table entry only applies to the launch it was timed for, so
none of the 593 shipped entries is ever active there.  What only exists at the benchmarked size - grids of 1 000 .. 15 000 workgroups with the
XCD-aware tile remap, 32-bit buffer offsets on 60 .. 130 MB tensors, channel slices of wider buffers (ld_in > Cin_p), BatchNorm statistics over
247 808 rows, split-K at 3 872 x 6 400, 320 pixel splits in the weight gradient - is verified here:

  forward      out == conv_f64 within one bf16 rounding; PN2_CONV_STATS partial (mean, M2) rows merge to the float64 mean / variance
  dgrad        transposed gather == conv_transpose_f64 within one rounding
  dgrad + BatchNorm-backward epilogue (pn2_conv_gemm_ep)   stored gradient (plain / accumulated / masked), sum dz and sum dz * x_hat per channel
               against float64, single and dual target (Bottle2neck's sp + spx[i])
  wgrad        slabs + reduce == float64 autograd weight gradient
  split-K      the 5x5 / 256-channel convs of the ra4 branch (pranet.py:304-306) with 4 K-slices + pn2_conv_splitk_reduce

Each case asserts that its key IS in the shipped table and that Engine._tune_gemm / _tune_wgrad would return that entry without timing anything.
One extra case drives a >= 2 GB activation tensor through pn2_conv_gemm: the LDS-DMA kernels address `in` with 32-bit offsets, the library must
switch to the register-staged kernel there (INTEGRATION.md "Size limits")."""
import ast
import ctypes as C
import json
import os, sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
dev = "cuda"
TABLE = {ast.literal_eval(k): (tuple(v) if isinstance(v, list) else v)
         for k, v in json.load(open(os.path.join(ROOT, "pranet-v2_amd", "pn2", "tuned_gfx950.json"))).items()}


def _rup(v, m):
    return (v + m - 1) // m * m


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ref_conv(x, w, s, ph, pw, dh, dw):
    """x [N,H,W,Cin] f64, w [Cout,Cin,KH,KW] f64 -> [N,OH,OW,Cout] f64"""
    if w.shape[2] == 1 and w.shape[3] == 1 and s == 1:
        return x @ w[:, :, 0, 0].t()
    return F.conv2d(x.permute(0, 3, 1, 2), w, None, s, (ph, pw), (dh, dw)).permute(0, 2, 3, 1).contiguous()


def _ref_dgrad(dy, w, s, ph, pw, dh, dw, H, W):
    """dy [N,OH,OW,Cout] f64 -> dx [N,H,W,Cin] f64"""
    KH, KW = w.shape[2], w.shape[3]
    if KH == 1 and KW == 1 and s == 1:
        return dy @ w[:, :, 0, 0]
    OH, OW = dy.shape[1], dy.shape[2]
    op = (H - ((OH - 1) * s - 2 * ph + dh * (KH - 1) + 1), W - ((OW - 1) * s - 2 * pw + dw * (KW - 1) + 1))
    return F.conv_transpose2d(dy.permute(0, 3, 1, 2), w, None, s, (ph, pw), op, 1, (dh, dw)).permute(0, 2, 3, 1).contiguous()


def _one_rounding(got, ref, first=None):
    """got == ref up to one bf16 rounding; `first`: a term that was itself rounded to bf16 before it entered the sum `ref` (gradient accumulation:
    the GEMM tile is staged as bf16, then added to the bf16 gradient already in memory and rounded again)."""
    tol = ref.abs() * 2.0 ** -8 + 1e-3 * float(ref.abs().max())
    if first is not None:
        tol = tol + first.abs() * 2.0 ** -8
    bad = (got - ref).abs() > tol
    assert not bool(bad.any()), (int(bad.sum()), float(((got - ref).abs() - tol).max()))


def _tile_m(code):
    return 64 if (code >> 2) & 3 == 1 else 128


def _engine_would_pick(key, code):
    """The engine's lookup (Engine._tune_gemm builds exactly this key from the launch's pn2_conv_desc) returns the shipped entry, no timing."""
    from pn2 import engine
    assert engine.TUNER.get(key) == code, "the shipped table is not what the engine has loaded (PN2_TUNE_TABLE=0 / PN2_TUNE_CACHE set?)"


# ---- forward / dgrad GEMMs: keys of the shipped table ('g', N, H, W, OH, OW, Cin_p, ld_in, Cout, KH, KW, stride, pad_h, pad_w, dil_h, dil_w, transposed)
FWD = [
    ("stem 3->32 s2 @352 (Res2Net_v1b.py:102)", ('g', 32, 352, 352, 176, 176, 8, 8, 32, 3, 3, 2, 1, 1, 1, 1, 0)),
    ("layer1 conv1 64->4x26 @88 (:32)", ('g', 32, 88, 88, 88, 88, 64, 64, 128, 1, 1, 1, 0, 0, 1, 1, 0)),
    ("layer1 conv3 4x26->256 @88 (:49)", ('g', 32, 88, 88, 88, 88, 128, 128, 256, 1, 1, 1, 0, 0, 1, 1, 0)),
    ("layer1 conv1 256->4x26 @88", ('g', 32, 88, 88, 88, 88, 256, 256, 128, 1, 1, 1, 0, 0, 1, 1, 0)),
    ("layer1 branch 26->26 3x3 @88, slice of the 128-wide split buffer (:44,66-69)", ('g', 32, 88, 88, 88, 88, 32, 128, 32, 3, 3, 1, 1, 1, 1, 1, 0)),
    ("layer2.0 branch 52->52 3x3 s2 88->44 (stage block)", ('g', 32, 88, 88, 44, 44, 56, 224, 56, 3, 3, 2, 1, 1, 1, 1, 0)),
    ("layer3.0 branch 104->104 3x3 s2 44->22", ('g', 32, 44, 44, 22, 22, 104, 416, 104, 3, 3, 2, 1, 1, 1, 1, 0)),
    ("layer4.0 branch 208->208 3x3 s2 22->11", ('g', 32, 22, 22, 11, 11, 208, 832, 208, 3, 3, 2, 1, 1, 1, 1, 0)),
    ("layer3 branch 104->104 3x3 @22", ('g', 32, 22, 22, 22, 22, 104, 416, 104, 3, 3, 1, 1, 1, 1, 1, 0)),
    ("layer3 conv3 416->1024 @22", ('g', 32, 22, 22, 22, 22, 416, 416, 1024, 1, 1, 1, 0, 0, 1, 1, 0)),
    ("layer3 conv1 1024->416 @22", ('g', 32, 22, 22, 22, 22, 1024, 1024, 416, 1, 1, 1, 0, 0, 1, 1, 0)),
    ("layer4 conv3 832->2048 @11", ('g', 32, 11, 11, 11, 11, 832, 832, 2048, 1, 1, 1, 0, 0, 1, 1, 0)),
    ("layer4 conv1 2048->832 @11", ('g', 32, 11, 11, 11, 11, 2048, 2048, 832, 1, 1, 1, 0, 0, 1, 1, 0)),
    ("rfb3_1 branch3 3x3 dilation 7 @22 (pranet.py:70)", ('g', 32, 22, 22, 22, 22, 32, 32, 32, 3, 3, 1, 7, 7, 7, 7, 0)),
    ("rfb2_1 branch2 1x5 on a slice of the fused N=224 reducer output @44 (:62)", ('g', 32, 44, 44, 44, 44, 32, 224, 32, 1, 5, 1, 0, 2, 1, 1, 0)),
    ("fused 1x1 reducers of x3: 1024 -> 5x32 + 64 = 224 @22 (:52-73,308)", ('g', 32, 22, 22, 22, 22, 1024, 1024, 224, 1, 1, 1, 0, 0, 1, 1, 0)),
    ("dgrad layer4 conv3 2048->832 @11 (plain)", ('g', 32, 11, 11, 11, 11, 2048, 2048, 1024, 1, 1, 1, 0, 0, 1, 1, 1)),
    ("dgrad ra4 5x5 256->256 @11 (plain, split-K in the step)", ('g', 32, 11, 11, 11, 11, 256, 256, 256, 5, 5, 1, 2, 2, 1, 1, 1)),
]


@pytest.mark.parametrize("name,key", FWD, ids=[n.split(" (")[0].replace(" ", "_") for n, _ in FWD])
def test_forward_and_dgrad_launches_of_the_benchmark_with_shipped_tiles(name, key):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pn2 import capi
    from pn2.capi import call, BF16
    assert key in TABLE, f"{name}: not in the shipped table"
    code = TABLE[key]
    _engine_would_pick(key, code)
    _, N, H, W, OH, OW, Cin_p, ld_in, Cout, KH, KW, s, ph, pw, dh, dw, transposed = key
    taps = KH * KW
    g = torch.Generator(device="cpu").manual_seed(H * 7 + Cin_p + Cout + taps)
    # the conv reads channels [0, Cin_p) of rows with pitch ld_in (a channel slice of a wider buffer when ld_in > Cin_p); the rest is poison
    src = torch.randn(N, H, W, ld_in, generator=g).bfloat16()
    if ld_in > Cin_p:
        src[..., Cin_p:] = float("nan")
    d = capi.ConvDesc()
    d.N, d.H, d.W, d.OH, d.OW = N, H, W, OH, OW
    d.Cin_p, d.ld_in, d.Cout, d.ld_out = Cin_p, ld_in, Cout, Cout
    d.KH, d.KW, d.stride, d.pad_h, d.pad_w, d.dil_h, d.dil_w = KH, KW, s, ph, pw, dh, dw
    d.transposed = transposed
    M = N * OH * OW
    if not transposed:
        w = (torch.randn(Cout, Cin_p, KH, KW, generator=g) * (2.0 / (taps * Cin_p)) ** 0.5).bfloat16()
        ref = _ref_conv(src[..., :Cin_p].double(), w.double(), s, ph, pw, dh, dw).reshape(M, Cout)
        d.Kp = _rup(taps * Cin_p, 128)
        wp = torch.zeros(_rup(Cout, 128), d.Kp, dtype=torch.bfloat16)
        wp[:Cout, :taps * Cin_p] = w.permute(0, 2, 3, 1).reshape(Cout, taps * Cin_p)
        stats = True
    else:
        # dgrad of a forward conv  (Cout -> Cin_p channels... in the forward's terms: x has `Cout` channels, dy has `Cin_p`):  in = dy [N,H,W], out = dx [N,OH,OW]
        w = (torch.randn(Cin_p, Cout, KH, KW, generator=g) * (2.0 / (taps * Cin_p)) ** 0.5).bfloat16()         # forward weight [co = dy channels][ci = dx channels]
        ref = _ref_dgrad(src[..., :Cin_p].double(), w.double(), s, ph, pw, dh, dw, OH, OW).reshape(M, Cout)
        d.Kp = _rup(taps * Cin_p, 128)
        wp = torch.zeros(_rup(Cout, 128), d.Kp, dtype=torch.bfloat16)
        wp[:Cout, :taps * Cin_p] = w.permute(1, 2, 3, 0).reshape(Cout, taps * Cin_p)                          # wp[ci][tap*Cout_fwd + co]
        stats = False
    tm = _tile_m(code)
    nblk = (M + tm - 1) // tm
    out = torch.full((M, Cout), float("nan"), dtype=torch.bfloat16, device=dev)
    psum = torch.full((nblk, Cout), float("nan"), dtype=torch.float32, device=dev) if stats else None
    psq = torch.full((nblk, Cout), float("nan"), dtype=torch.float32, device=dev) if stats else None
    d.flags = (capi.CONV_STATS if stats else 0) | (code << 8)
    src_g, wp_g = src.to(dev), wp.to(dev)
    call.pn2_conv_gemm(BF16, P(src_g), P(wp_g), P(out), P(psum), P(psq), C.byref(d), _stream())
    torch.cuda.synchronize()
    _one_rounding(out.double().cpu(), ref)
    if stats:
        # (mean, M2) per tile and channel -> Chan merge in float64 == the float64 mean / biased variance of the conv output (what pn2_bn_finalize computes)
        n_t = torch.full((nblk,), float(tm), dtype=torch.float64); n_t[-1] = M - (nblk - 1) * tm
        mean_t, m2_t = psum.double().cpu(), psq.double().cpu()
        mean = (mean_t * n_t[:, None]).sum(0) / M
        var = (m2_t.sum(0) + (n_t[:, None] * (mean_t - mean) ** 2).sum(0)) / M
        # the LDS-DMA kernels take the statistics of the STORED (bf16-rounded) tensor - the one pn2_affine_act normalises, and what torch.autocast's
        # batch_norm sees: tight against the stored output, within the rounding noise (2^-9 / sqrt(M)-ish) of the float64 reference
        o = out.double().cpu()
        smean, svar = o.mean(0), o.var(0, unbiased=False)
        sd = svar.sqrt()
        assert float(((mean - smean).abs() / sd).max()) < 2e-5, float(((mean - smean).abs() / sd).max())
        assert float(((var - svar).abs() / svar).max()) < 5e-5, float(((var - svar).abs() / svar).max())
        rmean, rvar = ref.mean(0), ref.var(0, unbiased=False)
        assert float(((mean - rmean).abs() / sd).max()) < 1e-3, float(((mean - rmean).abs() / sd).max())
        assert float(((var - rvar).abs() / rvar).max()) < 2e-3, float(((var - rvar).abs() / rvar).max())
    if key == FWD[-1][1]:
        # the step runs this conv's forward and dgrad as split-K (Engine._ksplit: K = 6400, M = 3872): 4 K-slices leave fp32 partial tiles, the reduce sums them
        ks = 4
        ws = torch.full((ks, M, Cout), float("nan"), dtype=torch.float32, device=dev)
        out2 = torch.full((M, Cout), float("nan"), dtype=torch.bfloat16, device=dev)
        d.flags = ((2 | (1 << 2) | (3 << 4)) << 8) | (ks << 16)
        call.pn2_conv_gemm(BF16, P(src_g), P(wp_g), P(out2), P(ws), P(None), C.byref(d), _stream())
        call.pn2_conv_splitk_reduce(BF16, P(ws), ks, M, Cout, P(out2), Cout, P(None), P(None), P(None), 0, _stream())
        torch.cuda.synchronize()
        _one_rounding(out2.double().cpu(), ref)


# ---- dgrad GEMMs that carry the BatchNorm-backward epilogue: key + ('ep', a.mode, b.mode, dual, accumulate flag)
EP = [
    ("conv1 256->4x26 @88: gradient of bn3+residual+ReLU output (mask from stored y)", ('g', 32, 88, 88, 88, 88, 128, 128, 256, 1, 1, 1, 0, 0, 1, 1, 1, 'ep', 5, 0, 0, 2)),
    ("conv3 4x26->256 @88: gradient of the concat buffer (mask from raw)", ('g', 32, 88, 88, 88, 88, 256, 256, 128, 1, 1, 1, 0, 0, 1, 1, 1, 'ep', 3, 0, 0, 0)),
    ("conv1 1024->416 @22", ('g', 32, 22, 22, 22, 22, 416, 416, 1024, 1, 1, 1, 0, 0, 1, 1, 1, 'ep', 5, 0, 0, 2)),
    ("conv3 832->2048 @11", ('g', 32, 11, 11, 11, 11, 2048, 2048, 832, 1, 1, 1, 0, 0, 1, 1, 1, 'ep', 3, 0, 0, 0)),
    ("stage branch 104->104 3x3 s2 dgrad 22->44", ('g', 32, 22, 22, 44, 44, 104, 104, 104, 3, 3, 2, 1, 1, 1, 1, 1, 'ep', 3, 0, 0, 0)),
    ("normal branch 104->104 3x3 @22, dual target (sp + spx[i])", ('g', 32, 22, 22, 22, 22, 104, 104, 104, 3, 3, 1, 1, 1, 1, 1, 1, 'ep', 3, 3, 1, 2)),
    ("normal branch 26->26 3x3 @88, dual target", ('g', 32, 88, 88, 88, 88, 32, 32, 32, 3, 3, 1, 1, 1, 1, 1, 1, 'ep', 3, 3, 1, 2)),
    ("rfb dilated 3x3 d5 @44 dgrad, BN without ReLU", ('g', 32, 44, 44, 44, 44, 32, 32, 32, 3, 3, 1, 5, 5, 5, 5, 1, 'ep', 1, 0, 0, 0)),
]


def _bn_operands(g, M, Cc):
    raw = torch.randn(M, Cc, generator=g).bfloat16()
    par = torch.empty(4, Cc)
    par[0] = torch.rand(Cc, generator=g) * 0.8 + 0.6          # scale
    par[1] = torch.randn(Cc, generator=g) * 0.3               # shift
    par[2] = torch.randn(Cc, generator=g) * 0.2               # mean
    par[3] = torch.rand(Cc, generator=g) * 0.8 + 0.6          # invstd
    return raw, par


def _expected_ep(stored, raw, par, mode, y):
    """float64 restatement of the epilogue's sums for one target: dz = stored * mask, p1 = sum dz, p2 = invstd * (sum dz*raw - mean * sum dz)."""
    r = raw.double()
    if mode & 4:
        mask = y.double() > 0
    elif mode & 2:
        mask = (r * par[0].double() + par[1].double()) > 0
    else:
        mask = torch.ones_like(r, dtype=torch.bool)
    dz = stored * mask
    p1 = dz.sum(0)
    p2 = par[3].double() * ((dz * r).sum(0) - par[2].double() * p1)
    return dz, p1, p2


@pytest.mark.parametrize("name,key", EP, ids=[n.split(":")[0].split(",")[0].replace(" ", "_") for n, _ in EP])
@pytest.mark.parametrize("store_masked", [False, True])
def test_dgrad_with_batchnorm_backward_epilogue_at_benchmark_shapes(name, key, store_masked):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pn2 import capi
    from pn2.capi import call, BF16
    assert key in TABLE, f"{name}: not in the shipped table"
    code = TABLE[key]
    _engine_would_pick(key, code)
    _, N, H, W, OH, OW, Cin_p, ld_in, Cout, KH, KW, s, ph, pw, dh, dw, transposed, _, amode, bmode, dual, accf = key
    assert transposed == 1
    if store_masked and (dual or not (amode & 6)):
        pytest.skip("PN2_BNB_STORE_MASKED is only used on single-target ReLU gradients")
    taps = KH * KW
    M = N * OH * OW
    g = torch.Generator(device="cpu").manual_seed(H * 11 + Cin_p + Cout + taps + amode)
    dy = torch.randn(N, H, W, ld_in, generator=g).bfloat16()
    w = (torch.randn(Cin_p, Cout, KH, KW, generator=g) * (2.0 / (taps * Cin_p)) ** 0.5).bfloat16()
    ref = _ref_dgrad(dy[..., :Cin_p].double(), w.double(), s, ph, pw, dh, dw, OH, OW).reshape(M, Cout)
    d = capi.ConvDesc()
    d.N, d.H, d.W, d.OH, d.OW = N, H, W, OH, OW
    d.Cin_p, d.ld_in, d.Cout, d.ld_out = Cin_p, ld_in, Cout, Cout
    d.KH, d.KW, d.stride, d.pad_h, d.pad_w, d.dil_h, d.dil_w = KH, KW, s, ph, pw, dh, dw
    d.transposed, d.Kp = 1, _rup(taps * Cin_p, 128)
    wp = torch.zeros(_rup(Cout, 128), d.Kp, dtype=torch.bfloat16)
    wp[:Cout, :taps * Cin_p] = w.permute(1, 2, 3, 0).reshape(Cout, taps * Cin_p)
    d.flags = (capi.CONV_ACCUM if accf else 0) | (code << 8)
    tm = _tile_m(code)
    nblk = (M + tm - 1) // tm
    ep = capi.ConvEp()
    keep = []

    def target(t, mode):
        raw, par = _bn_operands(g, M, Cout)
        y = (torch.randn(M, Cout, generator=g).bfloat16() if mode & 4 else None)
        raw_g, par_g, y_g = raw.to(dev), par.to(dev), (y.to(dev) if y is not None else None)
        p1 = torch.full((nblk, Cout), float("nan"), dtype=torch.float32, device=dev)
        p2 = torch.full((nblk, Cout), float("nan"), dtype=torch.float32, device=dev)
        t.mode = mode
        t.raw, t.ld_raw = raw_g.data_ptr(), Cout
        if y_g is not None:
            t.y, t.ld_y = y_g.data_ptr(), Cout
        t.par, t.ps = par_g.data_ptr(), Cout
        t.p1, t.p2, t.ldp = p1.data_ptr(), p2.data_ptr(), Cout
        keep.append((raw_g, par_g, y_g))
        return raw, par, y, p1, p2

    ta = target(ep.a, amode | (capi.BNB_STORE_MASKED if store_masked else 0))
    prior = torch.randn(M, Cout, generator=g).bfloat16() if accf else None
    out = (prior.to(dev).clone() if accf else torch.full((M, Cout), float("nan"), dtype=torch.bfloat16, device=dev))
    tb = None
    if dual:
        tb = target(ep.b, bmode)
        out_b = torch.full((M, Cout), float("nan"), dtype=torch.bfloat16, device=dev)
        ep.b.out, ep.b.ld_out = out_b.data_ptr(), Cout
    dy_g, wp_g = dy.to(dev), wp.to(dev)
    call.pn2_conv_gemm_ep(BF16, P(dy_g), P(wp_g), P(out), C.byref(d), C.byref(ep), _stream())
    torch.cuda.synchronize()
    got = out.double().cpu()
    # target a: the GEMM's own destination (+= the prior gradient when accumulating); the statistics see the STORED (rounded) gradient
    want = ref + (prior.double() if accf else 0)
    if not store_masked:
        _one_rounding(got, want, ref if accf else None)
        stored = got
    else:
        # the stored tensor is dz itself; rebuild the unmasked stored value from the float64 result rounded like the kernel does (fp32 sum -> bf16)
        stored = ((ref.float().bfloat16().double() + prior.double()) if accf else want).float().bfloat16().double()
    raw, par, y, p1, p2 = ta
    dz, e1, e2 = _expected_ep(stored, raw, par, amode, y)
    if store_masked:
        # one rounding where the mask keeps the element, exact zero where it drops it (a handful of elements within rounding of the threshold may differ)
        diff = (got - dz).abs() > (want.abs() * 2.0 ** -7 + 1e-3 * float(want.abs().max()))
        assert int(diff.sum()) <= 8, int(diff.sum())
    scale1 = dz.abs().sum(0) + 1e-30
    s1, s2 = p1.double().cpu().sum(0), p2.double().cpu().sum(0)
    assert float(((s1 - e1).abs() / scale1).max()) < 2e-4, float(((s1 - e1).abs() / scale1).max())
    scale2 = par[3].double() * ((dz * raw.double()).abs().sum(0) + par[2].double().abs() * dz.abs().sum(0)) + 1e-30
    assert float(((s2 - e2).abs() / scale2).max()) < 2e-4, float(((s2 - e2).abs() / scale2).max())
    if dual:
        got_b = out_b.double().cpu()
        _one_rounding(got_b, ref)                                   # the second destination receives the plain result, never accumulated
        raw, par, y, p1, p2 = tb
        dz, e1, e2 = _expected_ep(got_b, raw, par, bmode, y)
        s1, s2 = p1.double().cpu().sum(0), p2.double().cpu().sum(0)
        assert float(((s1 - e1).abs() / (dz.abs().sum(0) + 1e-30)).max()) < 2e-4
        scale2 = par[3].double() * ((dz * raw.double()).abs().sum(0) + par[2].double().abs() * dz.abs().sum(0)) + 1e-30
        assert float(((s2 - e2).abs() / scale2).max()) < 2e-4


# ---- weight gradients: ('w', N, H, W, OH, OW, Cin_p, ld_x, Cout_p, ld_dy, KH, KW, stride, pad_h, pad_w, dil_h, dil_w, heuristic splits) -> (kernel, splits)
WG = [
    ("stem 3->32 s2 @352: 991 232 pixels, 320 splits", ('w', 32, 352, 352, 176, 176, 8, 8, 32, 32, 3, 3, 2, 1, 1, 1, 1, 640)),
    ("layer1 conv3 4x26->256 @88", ('w', 32, 88, 88, 88, 88, 128, 128, 256, 256, 1, 1, 1, 0, 0, 1, 1, 192)),
    ("layer1 branch 26->26 3x3 @88 on a slice (ld_x 128)", ('w', 32, 88, 88, 88, 88, 32, 128, 32, 32, 3, 3, 1, 1, 1, 1, 1, 214)),
    ("layer2.0 branch 52->52 3x3 s2", ('w', 32, 88, 88, 44, 44, 56, 224, 56, 56, 3, 3, 2, 1, 1, 1, 1, 160)),
    ("layer3 conv1 1024->416 @22", ('w', 32, 22, 22, 22, 22, 1024, 1024, 416, 416, 1, 1, 1, 0, 0, 1, 1, 12)),
    ("layer4 conv3 832->2048 @11", ('w', 32, 11, 11, 11, 11, 832, 832, 2048, 2048, 1, 1, 1, 0, 0, 1, 1, 3)),
    ("layer3 branch 104->104 3x3 @22", ('w', 32, 22, 22, 22, 22, 104, 416, 104, 104, 3, 3, 1, 1, 1, 1, 1, 48)),
    ("ra4 5x5 256->256 @11", ('w', 32, 11, 11, 11, 11, 256, 256, 256, 256, 5, 5, 1, 2, 2, 1, 1, 3)),
    ("rfb 1x7 on a slice @44", ('w', 32, 44, 44, 44, 44, 32, 224, 32, 32, 1, 7, 1, 0, 3, 1, 1, 320)),
]


@pytest.mark.parametrize("name,key", WG, ids=[n.split(":")[0].replace(" ", "_") for n, _ in WG])
def test_weight_gradient_launches_of_the_benchmark_with_shipped_kernel_and_splits(name, key):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pn2 import capi, engine
    from pn2.capi import call, BF16
    assert key in TABLE, f"{name}: not in the shipped table"
    kern, ns = TABLE[key]
    assert engine.TUNER.get(key) == (kern, ns)
    _, N, H, W, OH, OW, Cin_p, ld_x, Cout_p, ld_dy, KH, KW, s, ph, pw, dh, dw, _ = key
    g = torch.Generator(device="cpu").manual_seed(H * 5 + Cin_p + Cout_p + KH)
    x = torch.randn(N, H, W, ld_x, generator=g).bfloat16()
    if ld_x > Cin_p:
        x[..., Cin_p:] = float("nan")
    dy = torch.randn(N, OH, OW, ld_dy, generator=g).bfloat16()
    xr, dyr = x[..., :Cin_p].double(), dy.double()
    if KH == 1 and KW == 1 and s == 1:
        ref = (dyr.reshape(-1, Cout_p).t() @ xr.reshape(-1, Cin_p)).reshape(Cout_p, Cin_p, 1, 1)
    else:
        wref = torch.zeros(Cout_p, Cin_p, KH, KW, dtype=torch.float64, requires_grad=True)
        F.conv2d(xr.permute(0, 3, 1, 2), wref, None, s, (ph, pw), (dh, dw)).backward(dyr.permute(0, 3, 1, 2))
        ref = wref.grad
    tco = call.pn2_wgrad_tile_co(Cout_p)
    wd = capi.WgradDesc()
    wd.N, wd.H, wd.W, wd.OH, wd.OW = N, H, W, OH, OW
    wd.Cin_p, wd.ld_x, wd.Cout_p, wd.ld_dy = Cin_p, ld_x, Cout_p, ld_dy
    wd.KH, wd.KW, wd.stride, wd.pad_h, wd.pad_w, wd.dil_h, wd.dil_w = KH, KW, s, ph, pw, dh, dw
    wd.Rp, wd.Kp, wd.tune = _rup(Cout_p, tco), _rup(KH * KW * Cin_p, 128), kern
    rd = capi.PackDesc()
    rd.Cout, rd.Cin, rd.KH, rd.KW = Cout_p, Cin_p, KH, KW
    rd.Cout_p, rd.gw_out, rd.gwp_out, rd.Cin_p, rd.gw_in, rd.gwp_in = Cout_p, Cout_p, Cout_p, Cin_p, Cin_p, Cin_p
    rd.Rp, rd.Kp, rd.transposed = wd.Rp, wd.Kp, 0
    slab = torch.full((ns, wd.Rp, wd.Kp), float("nan"), dtype=torch.float32, device=dev)
    gw = torch.full((Cout_p, Cin_p, KH, KW), float("nan"), dtype=torch.float32, device=dev)
    xg, dyg = x.to(dev), dy.to(dev)
    call.pn2_conv_wgrad(BF16, P(dyg), P(xg), P(slab), C.byref(wd), ns, _stream())
    call.pn2_wgrad_reduce(P(slab), P(gw), C.byref(rd), ns, 0, _stream())
    torch.cuda.synchronize()
    err = float((gw.double().cpu() - ref).abs().max()) / float(ref.abs().max())
    assert err <= 5e-5, err                    # exact bf16 products, fp32 sums over up to 991 232 pixels in <= 320 slabs


def test_activation_tensor_of_2GB_takes_the_64bit_address_kernel():
    """INTEGRATION.md "Size limits": the LDS-DMA kernels address `in` with 32-bit byte offsets; when the gathered tensor's extent reaches 2 GB the
    library must serve the launch with the register-staged kernel (64-bit addresses) - even when the caller's tuning code asks for an LDS-DMA
    kernel.  A 1x1 conv over 2 113 536 pixels x 512 bf16 channels (2.16 GB): rows below AND above the 2 GB line against float64."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pn2 import capi
    from pn2.capi import call, BF16
    N, H, W, Cin, Cout = 4, 688, 768, 512, 64
    M = N * H * W
    assert M * Cin * 2 >= 2 ** 31
    g = torch.Generator(device="cpu").manual_seed(3)
    x = torch.empty(M, Cin, dtype=torch.bfloat16, device=dev)
    x.normal_(generator=torch.Generator(device=dev).manual_seed(3))
    w = (torch.randn(Cout, Cin, generator=g) * (2.0 / Cin) ** 0.5).bfloat16()
    d = capi.ConvDesc()
    d.N, d.H, d.W, d.OH, d.OW = N, H, W, H, W
    d.Cin_p, d.ld_in, d.Cout, d.ld_out = Cin, Cin, Cout, Cout
    d.KH, d.KW, d.stride, d.pad_h, d.pad_w, d.dil_h, d.dil_w = 1, 1, 1, 0, 0, 1, 1
    d.transposed, d.Kp = 0, Cin
    wp = torch.zeros(128, Cin, dtype=torch.bfloat16)
    wp[:Cout] = w
    wp_g = wp.to(dev)
    rows = torch.cat([torch.arange(0, 4096), torch.arange(M // 2 - 2048, M // 2 + 2048), torch.arange(M - 4096, M)])      # 2 GB line = row 2 097 152 = M/2 - 8192...
    rows = torch.cat([rows, torch.arange(2 ** 31 // (Cin * 2) - 2048, 2 ** 31 // (Cin * 2) + 2048)])
    ref = x[rows.to(dev)].double().cpu() @ w.double().t()
    for code in (0, 2 | (2 << 2) | (2 << 4), 3 | (1 << 2) | (2 << 4)):          # library heuristic, and callers insisting on the LDS-DMA kernels
        out = torch.full((M, Cout), float("nan"), dtype=torch.bfloat16, device=dev)
        d.flags = code << 8
        call.pn2_conv_gemm(BF16, P(x), P(wp_g), P(out), P(None), P(None), C.byref(d), _stream())
        torch.cuda.synchronize()
        _one_rounding(out[rows.to(dev)].double().cpu(), ref)
        assert not bool(torch.isnan(out.float()).any())
