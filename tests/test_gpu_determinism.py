"""Bit-reproducibility of the DEFAULT training path: the captured bs = 32 step of bench.py replayed 50 x from identical
state must give the same bits every time - loss[0:5], every full-resolution map and every parameter gradient.  The only place of the step whose summation
order is not fixed by construction is the per-image loss sums of the one-pass DSRA tail (pn2_tail.hip: tail_one_k adds every band's five sums into per-image
accumulators in arrival order).  Round 5 used fp64 hardware atomics there (exact only while the band sums of an image lie within ~2^23 of each other); they are
now two-word fixed-point INTEGER atomics (tail_isum_add), associative whatever the spread.  The second test drives that kernel with CONFIDENT logits (|z| ~ 20,
right on the upper half of each image and wrong on the lower: band sums of one image spread over > 30 binades - the case the fp64 form could not guarantee)."""
import ctypes as C
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
os.environ.setdefault("PN2_NO_PRETRAINED", "1")
dev = "cuda"


@pytest.fixture(autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import pn2
    pn2.load_library()      # fails loudly if the HIP extension is missing
    yield
    pn2.set_compute_dtype("bf16")


def test_captured_bs32_step_replays_bit_identically_50_times():
    """bench.py's object (Trainer.capture / replay at 32 x 3 x 352 x 352, bf16, random init) with lr = 0: the weights stay put, so every replay is the same
    computation - loss, maps and the whole gradient arena must not move by a bit over 50 replays."""
    import pn2
    from pn2.trainer import Trainer
    from lib.pranet import PraNet_V2
    from oracle import weights as W
    pn2.set_compute_dtype("bf16")
    torch.manual_seed(0)
    model = PraNet_V2(num_class=1).to(dev).train()
    x, mask = W.synthetic_batch(32, 352, seed=1234)
    x, mask = x.to(dev), mask.to(dev)
    tr = Trainer(model, lr=0.0, clip=0.5)
    tr.capture(x, mask, warmup=2)
    loss0 = tr.replay().clone()
    torch.cuda.synchronize()
    g0, maps0, w0 = tr.gflat.clone(), tr.last_outs.clone(), tr.flat.clone()
    assert torch.isfinite(loss0).all() and float(g0.abs().sum()) > 0
    for i in range(50):
        loss = tr.replay()
        torch.cuda.synchronize()
        assert torch.equal(loss, loss0), (i, loss.tolist(), loss0.tolist())
        assert torch.equal(tr.gflat, g0), (i, float((tr.gflat - g0).abs().max()))
        assert torch.equal(tr.last_outs, maps0), i
    assert torch.equal(tr.flat, w0)          # (lr = 0: the state the replays started from really was identical)


def test_one_pass_tail_is_order_independent_on_confident_logits():
    """pn2_dsra_tail_fwd_bwd with the image-sum accumulators, 40 launches on sources whose up-sampled logits are ~ +-20, agreeing with the mask on the upper half of every image
    and contradicting it on the lower (the band sums w*bce of one image then span ~1e-5 .. 1e5): loss, per-image sums and every low-res gradient bit-identical across launches, and the
    loss equal to the atomic-free three-launch form (isum = NULL) to fp32 rounding."""
    from pn2 import capi
    from pn2.capi import call
    from oracle import weights as W
    N, S = 8, 352
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    _, mask = W.synthetic_batch(N, S, seed=77)
    mask = mask.reshape(N, S, S).to(dev).contiguous()
    g = torch.Generator(device="cpu").manual_seed(5)
    sizes = [44, 44, 22, 11] * 2
    srcs = []
    for j, h in enumerate(sizes):
        sgn = torch.nn.functional.interpolate(mask[:, None].cpu(), size=(h, h), mode="nearest")[:, 0] * 2 - 1          # +1 on the mask, -1 off it
        if j >= 4:
            sgn = -sgn                                       # bg maps are judged against 1 - mask
        t = torch.randn(N, h, h, generator=g) * 0.05
        t[:, : h // 2] += 20.0 * sgn[:, : h // 2]            # upper half: confident and RIGHT (w*bce ~ e^-20 per pixel away from the mask's edge)
        t[:, h // 2:] -= 20.0 * sgn[:, h // 2:]              # lower half: confident and WRONG (w*bce ~ 20 w per pixel)
        srcs.append(t.to(dev).contiguous())
    weit = torch.empty_like(mask)
    nb = call.pn2_dsra_tail_blocks(S)

    def run(with_isum):
        dsrcs = [torch.zeros_like(t) for t in srcs]
        d = capi.TailDesc()
        d.N, d.OH, d.OW, d.P, d.align_corners = N, S, S, 4, 0
        for j, (s_, ds) in enumerate(zip(srcs, dsrcs)):
            m, h = d.maps[j], s_.shape[1]
            m.src, m.dsrc, m.h, m.w, m.accumulate = s_.data_ptr(), ds.data_ptr(), h, h, 0
            m.rh = m.rw = h / S
        assert int(call.pn2_dsra_tail_fused_ok(C.byref(d))) == 1
        isum = torch.full((4 * N * 10,), 3, dtype=torch.int64, device=dev) if with_isum else None
        if with_isum:
            call.pn2_loss_weights_clear(P(mask), P(weit), N, S, S, 31, P(isum), isum.numel(), st)
        else:
            call.pn2_loss_weights(P(mask), P(weit), N, S, S, 31, st)
        lat = torch.empty(8, N, S, S, device=dev)
        partial = torch.empty(4, N, nb, 5, device=dev)
        sums, wsum, loss, per = torch.empty(4, N, 4, device=dev), torch.empty(N, device=dev), torch.empty(5, device=dev), torch.empty(4, N, device=dev)
        need = int(call.pn2_dsra_tail_fused_scratch(C.byref(d)))
        scratch = torch.empty(need, device=dev)
        call.pn2_dsra_tail_fwd_bwd(C.byref(d), P(lat), P(mask), P(weit), P(partial), P(sums), P(wsum), P(per), P(loss), 1.0, P(scratch), need, P(isum), st)
        torch.cuda.synchronize()
        return loss, sums, wsum, dsrcs, partial

    ref = run(True)
    assert torch.isfinite(ref[0]).all()
    bands = ref[4][0, :, :, 0]          # w*bce band sums of pair 0: [image][band]
    spread = float((bands.max(1).values / bands.clamp_min(1e-30).min(1).values).max())
    assert spread > 2.0 ** 30, f"the fixture is meant to spread an image's band sums over > 30 binades (got 2^{__import__('math').log2(spread):.1f})"
    for i in range(40):
        got = run(True)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2]), i
        assert all(torch.equal(a, b) for a, b in zip(got[3], ref[3])), i
    plain = run(False)
    assert float((plain[0] - ref[0]).abs().max()) <= 2e-6 * float(ref[0].abs().max())
    for a, b in zip(plain[3], ref[3]):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max() + 1e-30)


def test_mixed_weight_gradient_table_is_bit_identical_to_the_two_tables_bs32_bf16(monkeypatch):
    """GradQueue._build with WGRAD_MIX (default): the k x k and the pointwise jobs of the 128 x 256 weight-gradient tile run from ONE table (conv_wgrad_dma_tab_mix, job ranges
    interleaved) instead of two launches - same per-job arithmetic: loss, gradient arena and parameters after three steps bit for bit, at the benchmarked size (the
    small fixtures have no job on that tile)."""
    import pn2
    from pn2 import core
    from pn2.trainer import Trainer
    from lib.pranet import PraNet_V2
    from oracle import weights as W
    pn2.set_compute_dtype("bf16")
    x, m = W.synthetic_batch(32, 352, seed=1234)
    x, m = x.to(dev), m.to(dev)
    res, variants = [], []
    real = core.call.pn2_conv_wgrad_multi
    for mix in (0.0, 2.0):
        monkeypatch.setattr(core, "WGRAD_MIX", mix)
        seen = set()
        monkeypatch.setattr(core.call, "pn2_conv_wgrad_multi", lambda dt, v, *a, _s=seen: (_s.add(v), real(dt, v, *a))[1], raising=False)
        torch.manual_seed(0)
        tr = Trainer(PraNet_V2(num_class=1).to(dev).train(), lr=1e-4, clip=0.5)
        for _ in range(3):
            loss = tr.step(x, m)
        torch.cuda.synchronize()
        res.append((loss.clone(), tr.gflat.clone(), tr.flat.clone()))
        variants.append(seen)
        monkeypatch.setattr(core.call, "pn2_conv_wgrad_multi", real, raising=False)
    assert {12, 13} <= variants[0] and 14 not in variants[0] and 14 in variants[1] and not ({12, 13} & variants[1])
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
