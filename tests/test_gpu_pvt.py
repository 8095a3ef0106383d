"""GPU parity tests of the PVTv2 encoder path (config 4): every new op through the C ABI against torch on the CPU (float64), one
transformer Block against the oracle, and the whole PVT_PraNet_V2 training forward/backward against the vectors the imported reference
produced (tests/golden/pvt_pranet_v2_96.npz, DropPath off).  Tolerances as in test_gpu_parity.py."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
os.environ.setdefault("PN2_NO_PRETRAINED", "1")
dev = "cuda"


@pytest.fixture(autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import pn2
    pn2.load_library()
    yield
    pn2.set_compute_dtype("bf16")


def relmax(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def rell2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-12))


def _run(dtn, build, ref, x, params):
    """build(eng, act) on the GPU vs ref(x64) in float64; returns after asserting output, input-gradient and parameter gradients."""
    from pn2 import F32, BF16
    from pn2.engine import Engine
    from pn2.graph import _seed_grad
    dt = F32 if dtn == "fp32" else BF16
    err, tol = (relmax, 5e-5) if dt == F32 else (rell2, 3e-2)
    eng = Engine(dt, True, need_grad=True)
    a = eng.from_nchw(x, requires_grad=True)
    y = build(eng, a)
    out = eng.to_nchw(y).clone()
    gy = torch.randn_like(out)
    _seed_grad(y, gy)
    eng.backward()
    xc = (x.bfloat16().float() if dt == BF16 else x).double().cpu().requires_grad_(True)
    p64 = [p.detach().double().cpu().requires_grad_(True) for p in params]
    r = ref(xc, *p64)
    r.backward(gy.double().cpu())
    assert err(out, r) < tol, "forward"
    assert err(a.grad[..., :x.shape[1]].float().permute(0, 3, 1, 2), xc.grad) < tol, "input gradient"
    for p, q in zip(params, p64):
        assert err(eng.pgrads.get(p), q.grad) < tol, "parameter gradient"


@pytest.mark.parametrize("dtn", ["fp32", "bf16"])
@pytest.mark.parametrize("C_", [64, 128, 320, 512])
def test_layernorm(dtn, C_):
    torch.manual_seed(C_)
    ln = nn.LayerNorm(C_, eps=1e-6).to(dev)
    ln.weight.data.uniform_(0.5, 1.5); ln.bias.data.normal_(0, 0.2)
    x = torch.randn(3, C_, 7, 5, device=dev) * 2 + 0.3
    _run(dtn, lambda e, a: e.layernorm(a, ln),
         lambda t, g, b: F.layer_norm(t.permute(0, 2, 3, 1), (C_,), g, b, 1e-6).permute(0, 3, 1, 2), x, [ln.weight, ln.bias])


@pytest.mark.parametrize("dtn", ["fp32", "bf16"])
def test_linear_bias_residual_and_strided_biased_convs(dtn):
    torch.manual_seed(3)
    lin = nn.Linear(64, 320).to(dev)
    x = torch.randn(2, 64, 6, 9, device=dev)
    _run(dtn, lambda e, a: e.linear(a, lin), lambda t, w, b: F.linear(t.permute(0, 2, 3, 1), w, b).permute(0, 3, 1, 2), x, [lin.weight, lin.bias])
    lin2 = nn.Linear(128, 128).to(dev)
    x2 = torch.randn(2, 128, 5, 5, device=dev)
    _run(dtn, lambda e, a: e.linear(a, lin2, residual=a), lambda t, w, b: F.linear(t.permute(0, 2, 3, 1), w, b).permute(0, 3, 1, 2) + t, x2, [lin2.weight, lin2.bias])
    for (cin, cout, k, s, p, H) in ((3, 64, 7, 4, 3, 33), (64, 128, 3, 2, 1, 13), (64, 64, 8, 8, 0, 24), (128, 128, 4, 4, 0, 12), (320, 320, 2, 2, 0, 6)):
        conv = nn.Conv2d(cin, cout, k, s, p).to(dev)
        conv.bias.data.normal_(0, 0.2)
        xx = torch.randn(2, cin, H, H, device=dev)
        _run(dtn, lambda e, a: e.conv_bias(a, conv), lambda t, w, b: F.conv2d(t, w, b, s, p), xx, [conv.weight, conv.bias])


@pytest.mark.parametrize("dtn", ["fp32", "bf16"])
def test_dwconv_gelu(dtn):
    torch.manual_seed(4)
    for C_, H, W in ((64, 9, 7), (512, 5, 6), (1280, 3, 3)):
        conv = nn.Conv2d(C_, C_, 3, 1, 1, groups=C_).to(dev)
        conv.weight.data.normal_(0, 0.4); conv.bias.data.normal_(0, 0.3)
        x = torch.randn(2, C_, H, W, device=dev)
        _run(dtn, lambda e, a: e.dwconv_gelu(a, conv), lambda t, w, b: F.gelu(F.conv2d(t, w, b, 1, 1, groups=C_)), x, [conv.weight, conv.bias])


@pytest.mark.parametrize("dtn", ["fp32", "bf16"])
@pytest.mark.parametrize("cfg", [(1, 12, 12, 3, 3), (2, 6, 6, 3, 3), (5, 4, 5, 11, 11), (8, 3, 3, 3, 3), (2, 16, 16, 14, 14), (1, 20, 13, 13, 12), (2, 9, 30, 16, 16)])
def test_attention(dtn, cfg):
    """q [B, Nq, heads*64] against kv [B, Nkv, 2*heads*64] (k then v, heads inner), softmax over Nkv (9..256 keys: every 64-key bucket, several query tiles per block)."""
    from pn2 import F32, BF16
    from pn2.engine import Engine
    from pn2.graph import _seed_grad
    heads, qh, qw, kh, kw = cfg
    dt = F32 if dtn == "fp32" else BF16
    err, tol = (relmax, 5e-5) if dt == F32 else (rell2, 3e-2)
    torch.manual_seed(heads)
    B, Cc = 2, heads * 64
    q = torch.randn(B, Cc, qh, qw, device=dev)
    kv = torch.randn(B, 2 * Cc, kh, kw, device=dev)
    eng = Engine(dt, True, need_grad=True)
    qa, kva = eng.from_nchw(q, True), eng.from_nchw(kv, True)
    o = eng.attention(qa, kva, heads)
    out = eng.to_nchw(o).clone()
    go = torch.randn_like(out)
    _seed_grad(o, go); eng.backward()
    cast = (lambda t: t.bfloat16().float()) if dt == BF16 else (lambda t: t)
    q64 = cast(q).double().cpu().requires_grad_(True); kv64 = cast(kv).double().cpu().requires_grad_(True)
    Nq, Nkv = qh * qw, kh * kw
    qq = q64.flatten(2).transpose(1, 2).reshape(B, Nq, heads, 64).permute(0, 2, 1, 3)
    kk = kv64.flatten(2).transpose(1, 2).reshape(B, Nkv, 2, heads, 64).permute(2, 0, 3, 1, 4)
    attn = ((qq @ kk[0].transpose(-2, -1)) * 64 ** -0.5).softmax(dim=-1)
    r = (attn @ kk[1]).transpose(1, 2).reshape(B, Nq, Cc).transpose(1, 2).reshape(B, Cc, qh, qw)
    r.backward(go.double().cpu())
    assert err(out, r) < tol
    assert err(qa.grad.float().permute(0, 3, 1, 2), q64.grad) < tol
    assert err(kva.grad.float().permute(0, 3, 1, 2), kv64.grad) < tol


def _pvt_model(fp32=True, seed=3):
    import pn2
    from lib.pranet import PVT_PraNet_V2
    from oracle import weights as W
    pn2.set_compute_dtype("fp32" if fp32 else "bf16")
    model = PVT_PraNet_V2(num_class=1)
    model.load_state_dict(W.make_state_dict(W.manifest_pvt_pranet_v2(1), seed=seed), strict=True)
    model.backbone.reset_drop_path(0.0)
    return model.to(dev).train()


def test_pvt_state_dict_manifest():
    import json
    from lib.pranet import PVT_PraNet_V2
    ref = json.load(open(os.path.join(G, "manifest_pvt.json")))["pvt_pranet_v2_k1"]
    model = PVT_PraNet_V2(num_class=1)
    assert [(k, list(v.shape)) for k, v in model.state_dict().items()] == list(ref.items())


def test_pvt_backbone_features_vs_reference():
    z = np.load(os.path.join(G, "pvt_pranet_v2_96.npz"))
    from oracle import weights as W
    model = _pvt_model(fp32=True)
    x, _ = W.synthetic_batch(2, 96, seed=4321)
    with torch.no_grad():
        feats = model.backbone(x.to(dev))
    for i, f in enumerate(feats):
        ref64 = torch.from_numpy(z[f"f64.feat{i}"])
        own = float((torch.from_numpy(z[f"feat{i}"]).double() - ref64).abs().max())
        assert float((f.double().cpu() - ref64).abs().max()) <= max(1e-4, 3 * own), i


@pytest.mark.parametrize("fp32", [True, False])
def test_pvt_pranet_v2_forward_backward_vs_reference(fp32):
    """Whole config-4 model, train mode, through nn.Module + torch autograd: 8 outputs, the 4-pair loss and 26 gradient probes."""
    from pn2.loss import structure_loss
    from oracle import weights as W
    z = np.load(os.path.join(G, "pvt_pranet_v2_96.npz"))
    model = _pvt_model(fp32=fp32)
    x, mask = W.synthetic_batch(2, 96, seed=4321)
    xg, mg = x.to(dev), mask.to(dev)
    outs = model(xg)
    losses = [structure_loss(outs[i], outs[i + 4], mg, 1 - mg) for i in range(4)]
    loss = losses[3] + losses[2] + losses[1] + losses[0]
    loss.backward()
    names = dict(model.named_parameters())
    if fp32:
        for i, o in enumerate(outs):
            ref64 = torch.from_numpy(z[f"f64.out{i}"])
            own = float((torch.from_numpy(z[f"out{i}"]).double() - ref64).abs().max())
            assert float((o.detach().double().cpu() - ref64).abs().max()) <= max(1e-4, 3 * own), i
        assert abs(float(loss) - float(z["f64.losses"].sum())) < max(1e-4, 3 * abs(float(z["loss"]) - float(z["f64.losses"].sum())))
        for k in z.files:
            if k.startswith("f64.grawnorm."):
                name = k[len("f64.grawnorm."):]
                g = names[name].grad
                r64, r32 = float(z[k]), float(z["grawnorm." + name])
                assert abs(float(g.norm()) - r64) <= max(2e-4 * r64, 3 * abs(r32 - r64)) + 1e-7, name
                h64 = torch.from_numpy(z["f64.graw." + name]).double()
                h32 = torch.from_numpy(z["graw." + name]).double()
                ours = g.detach().reshape(-1)[:h64.numel()].double().cpu()
                assert float((ours - h64).norm()) <= max(2e-4 * float(h64.norm()), 3 * float((h32 - h64).norm())) + 1e-7, name
    else:
        # bf16 storage through 16 transformer blocks + train-mode BN heads on 3x3..12x12 maps: a sanity band, not a parity bound
        for i, o in enumerate(outs):
            assert rell2(o, torch.from_numpy(z[f"f64.out{i}"])) < 0.15, i
        assert abs(float(loss) - float(z["loss"])) < 5e-2 * float(z["loss"])
        for k in z.files:
            if k.startswith("f64.grawnorm."):
                name = k[len("f64.grawnorm."):]
                assert abs(float(names[name].grad.norm()) - float(z[k])) < 0.3 * float(z[k]) + 5e-3, name      # (norm4.bias: exactly 0 analytically, BN cancels it)


def test_pvt_pranet_v2_one_channel_input_vs_reference():
    """1-channel slices take the conv(1->3)+BN+ReLU stem of pranet.py:190-191 first (its three parameters train; its bias gradient is ~0 under
    the train-mode BN); fp32 path against the float64 reference, gates relative to the reference's own fp32 error."""
    from pn2.loss import structure_loss
    from oracle import weights as W
    z = np.load(os.path.join(G, "pvt_pranet_v2_gray_64.npz"))
    model = _pvt_model(fp32=True, seed=11)
    x, mask = W.synthetic_batch(2, 64, seed=777)
    xg, mg = x[:, :1].contiguous().to(dev), mask.to(dev)
    outs = model(xg, segSize=None)
    losses = [structure_loss(outs[i], outs[i + 4], mg, 1 - mg) for i in range(4)]
    (losses[3] + losses[2] + losses[1] + losses[0]).backward()
    for i, o in enumerate(outs):
        ref64 = torch.from_numpy(z[f"f64.out{i}"])
        own = float((torch.from_numpy(z[f"out{i}"]).double() - ref64).abs().max())
        assert float((o.detach().double().cpu() - ref64).abs().max()) <= max(1e-4, 3 * own), i
    names = dict(model.named_parameters())
    for k in z.files:
        if k.startswith("f64.grawnorm."):
            name = k[len("f64.grawnorm."):]
            g = names[name].grad
            assert g is not None, name
            h64 = torch.from_numpy(z["f64.graw." + name]).double()
            h32 = torch.from_numpy(z["graw." + name]).double()
            ours = g.detach().reshape(-1)[:h64.numel()].double().cpu()
            assert float((ours - h64).norm()) <= max(2e-4 * float(h64.norm()), 3 * float((h32 - h64).norm())) + 1e-7, name
    assert float((model.conv[1].running_mean.double().cpu() - torch.from_numpy(z["f64.rm.conv.1"]).double()).abs().max()) < 1e-5
    assert float((model.conv[1].running_var.double().cpu() - torch.from_numpy(z["f64.rv.conv.1"]).double()).abs().max()) < 1e-5
    assert names["conv.0.bias"].grad is not None and float(names["conv.0.bias"].grad.abs().max()) < 1e-4


def test_pvt_trainer_step_and_graph_replay():
    """The fused trainer (arena, deferred wgrad tables, fused tail) drives the PVT model too; hipGraph replay == eager bit for bit."""
    from pn2.trainer import Trainer
    from oracle import weights as W
    x, mask = W.synthetic_batch(2, 96, seed=4321)
    xg, mg = x.to(dev), mask.to(dev)
    z = np.load(os.path.join(G, "pvt_pranet_v2_96.npz"))
    res = []
    for graph in (False, True):
        tr = Trainer(_pvt_model(fp32=False), lr=1e-4, clip=0.5)
        if graph:
            tr.capture(xg, mg, warmup=2)
            loss = tr.replay()
        else:
            for i in range(3):
                loss = tr.step(xg, mg)
                if i == 0:
                    assert abs(float(loss[-1]) - float(z["loss"])) < 5e-2 * float(z["loss"])
        torch.cuda.synchronize()
        res.append((loss.clone(), tr.flat.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


def test_fc1_bias_gradient_from_the_depthwise_data_gradient_walk(monkeypatch):
    """PN2_DW_COLSUM (default): the depth-wise conv's data-gradient kernel IS the producer of Mlp.fc1's output gradient (pvtv2.py:49-56), so it also leaves that
    tensor's column sums - fc1's bias gradient - as per-workgroup partial rows (pn2_dwconv3x3_colsum) and the deferred column-sum pass does not read the block's
    largest gradient tensor again.  One backward with the switch on and off: every gradient identical except fc1.bias up to the fp32 order of the sums."""
    from pn2 import core
    from pn2.capi import call
    from pn2.trainer import Trainer
    from oracle import weights as W
    x, mask = W.synthetic_batch(2, 96, seed=99)
    xg, mg = x.to(dev), mask.to(dev)
    res = []
    for on in (False, True):
        monkeypatch.setattr(core, "DW_COLSUM", on)
        n = {"c": 0}
        real = call.pn2_dwconv3x3_colsum
        monkeypatch.setattr(call, "pn2_dwconv3x3_colsum", lambda *a: (n.__setitem__("c", n["c"] + 1), real(*a))[1], raising=False)
        model = _pvt_model(fp32=False)
        tr = Trainer(model, lr=1e-4, clip=0.5)
        tr.forward_backward(xg, mg)
        torch.cuda.synchronize()
        monkeypatch.setattr(call, "pn2_dwconv3x3_colsum", real, raising=False)
        fc1b = torch.zeros_like(tr.gflat, dtype=torch.bool)
        for k, p in model.named_parameters():
            if k.endswith("mlp.fc1.bias") and id(p) in tr.off:
                off, cnt = tr.off[id(p)]
                fc1b[off:off + cnt] = True
        res.append((tr.gflat.clone(), fc1b, n["c"]))
    (g0, m0, n0), (g1, m1, n1) = res
    assert n0 == 0 and n1 == 16 and int(m0.sum()) > 0 and torch.equal(m0, m1), (n0, n1)          # one launch per block of PVTv2-B2
    assert torch.equal(g0[~m0], g1[~m0]), "only fc1's bias gradients may differ"
    assert float((g0[m0] - g1[m0]).abs().max()) <= 2e-6 * float(g0[m0].abs().max()) + 1e-12


def test_drop_path_plan_draws_and_residual_gradient_alias():
    """Engine.drop_path_plan: the DropPath draws of a forward from ONE bernoulli over a [K][N] table of keep probabilities (pvtv2.py:125,148-149 asks for two
    per block); drop_path_add takes its rows in order, a request that does not match draws for itself; the residual's gradient shares dy's buffer."""
    from pn2 import BF16
    from pn2.engine import Engine
    from pn2.graph import _seed_grad
    torch.manual_seed(3)
    N = 4096
    eng = Engine(BF16, True, need_grad=True)
    eng.drop_path_plan([0.0, 0.1, 0.1, 0.3, 0.3], N)
    table, ps = eng._dp_rows
    assert ps == [0.1, 0.1, 0.3, 0.3] and tuple(table.shape) == (4, N)
    for k, p in enumerate(ps):
        vals = torch.unique(table[k]).tolist()
        assert len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - 1.0 / (1.0 - p)) < 1e-6
        assert abs(float((table[k] > 0).float().mean()) - (1.0 - p)) < 4 * (p * (1 - p) / N) ** 0.5          # keep rate within 4 sigma
    assert not torch.equal(table[0] > 0, table[1] > 0)
    res = torch.randn(N, 16, 2, 2, device=dev); x = torch.randn(N, 16, 2, 2, device=dev)
    ra, xa = eng.from_nchw(res, requires_grad=True), eng.from_nchw(x, requires_grad=True)
    y = eng.drop_path_add(ra, xa, 0.1)
    assert eng._dp_next == 1
    sc = table[0].view(N, 1, 1, 1)
    want = torch.addcmul(res.bfloat16().double(), x.bfloat16().double(), sc.double())
    assert float((eng.to_nchw(y).double() - want).abs().max()) <= 2 ** -8 * float(want.abs().max())          # one bf16 rounding of the result
    xb = eng.from_nchw(torch.randn(N, 16, 2, 2, device=dev), requires_grad=True)
    z = eng.drop_path_add(y, xb, 0.2)          # not in the plan at this position: own draw, the plan's cursor stays
    assert eng._dp_next == 1
    gy = torch.randn(N, 16, 2, 2, device=dev)
    _seed_grad(z, gy)
    eng.backward()
    gz = gy.bfloat16()
    assert y.grad.data_ptr() == z.grad.data_ptr() and ra.grad.data_ptr() == z.grad.data_ptr(), "the residual stream's gradient is one buffer"
    assert torch.equal(ra.grad[..., :16].permute(0, 3, 1, 2), gz)
    eng2 = Engine(BF16, False, need_grad=False)
    eng2.drop_path_plan([0.1, 0.1], N)
    assert eng2._dp_rows is None          # eval: nothing is drawn


@pytest.mark.parametrize("dtn", ["fp32", "bf16"])
def test_sliding_window_depthwise_edge_geometries(dtn):
    """The row-segment walks of the depth-wise 3x3 (+GELU) kernels at awkward sizes: one-pixel and one-row images, widths that are not a
    multiple of the segment length, channel counts that force the narrow vector variants, large tensors that take the wide ones."""
    torch.manual_seed(14)
    for C_, H, W in ((8, 1, 1), (8, 1, 37), (16, 23, 1), (24, 5, 33), (40, 9, 9), (72, 17, 19), (512, 40, 47)):
        conv = nn.Conv2d(C_, C_, 3, 1, 1, groups=C_).to(dev)
        conv.weight.data.normal_(0, 0.4); conv.bias.data.normal_(0, 0.3)
        x = torch.randn(2, C_, H, W, device=dev)
        _run(dtn, lambda e, a: e.dwconv_gelu(a, conv), lambda t, w, b: F.gelu(F.conv2d(t, w, b, 1, 1, groups=C_)), x, [conv.weight, conv.bias])


@pytest.mark.parametrize("dtn", ["fp32", "bf16"])
def test_patchify_and_few_channel_strided_data_gradients(dtn):
    """Data gradients that do not run the transposed-gather GEMM: kernel == stride convs (GEMM over the patches + depth-to-space, also when the
    image is not a multiple of the stride: the remainder rows / columns get zero) and strided convs with <= 4 input channels (per-pixel walk
    over the taps that land on an output pixel)."""
    torch.manual_seed(15)
    cases = [(64, 64, 8, 8, 0, 24, 24), (64, 64, 4, 4, 0, 22, 18), (32, 48, 2, 2, 0, 11, 13),      # patchify, incl. sizes with a remainder
             (3, 64, 7, 4, 3, 40, 36), (3, 32, 3, 2, 1, 21, 19), (1, 16, 5, 4, 2, 33, 30), (4, 24, 3, 2, 1, 16, 16)]   # few input channels
    for cin, cout, k, s, p, H, W in cases:
        conv = nn.Conv2d(cin, cout, k, s, p).to(dev)
        x = torch.randn(2, cin, H, W, device=dev)
        _run(dtn, lambda e, a: e.conv_bias(a, conv), lambda t, w, b: F.conv2d(t, w, b, s, p), x, [conv.weight, conv.bias])


def test_deferred_column_sums_and_splitk_match_immediate_launches(monkeypatch):
    """pn2_colsum_multi + pn2_colsum_finalize_multi (one table-driven launch per flush) against one pn2_colsum / pn2_colsum_finalize per
    bias, LayerNorm and depth-wise parameter: bit for bit over a PVT-PraNet-V2 training step; and the split-K launch of a few-row / long-
    contraction conv (fp32 partial tiles + reduce with the BatchNorm statistics) against the plain launch."""
    import ctypes as C
    from pn2 import engine, capi, core
    from pn2.capi import call, BF16
    from pn2.trainer import Trainer
    from oracle import weights as W
    x, m = W.synthetic_batch(2, 96, seed=31)
    x, m = x.to(dev), m.to(dev)
    res = []
    for defer in (False, True):
        monkeypatch.setattr(core, "DEFER_COLSUM", defer)
        tr = Trainer(_pvt_model(fp32=False))
        for _ in range(3):
            loss = tr.forward_backward(x, m)
        torch.cuda.synchronize()
        res.append((loss.clone(), tr.gflat.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    # ---- split-K
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.manual_seed(2)
    N, H, Wd, Cin, Cout, k = 4, 11, 11, 256, 200, 5
    M, K = N * H * Wd, k * k * Cin
    Kp = (K + 127) // 128 * 128
    xin = torch.randn(M, Cin, device=dev).bfloat16()
    wp = (torch.randn((Cout + 127) // 128 * 128, Kp, device=dev) * 0.03).bfloat16()
    wp[Cout:] = 0; wp[:, K:] = 0
    d = capi.ConvDesc()
    d.N, d.H, d.W, d.OH, d.OW = N, H, Wd, H, Wd
    d.Cin_p, d.ld_in, d.Cout, d.ld_out = Cin, Cin, Cout, Cout
    d.KH, d.KW, d.stride, d.pad_h, d.pad_w, d.dil_h, d.dil_w = k, k, 1, 2, 2, 1, 1
    d.transposed, d.Kp = 0, Kp
    nb = (M + 63) // 64
    code = 2 | (1 << 2) | (3 << 4)
    o1 = torch.empty(M, Cout, device=dev, dtype=torch.bfloat16); o2 = torch.empty_like(o1)
    ps1, pq1, ps2, pq2 = (torch.zeros(nb, Cout, device=dev) for _ in range(4))
    d.flags = (code << 8) | capi.CONV_STATS
    call.pn2_conv_gemm(BF16, P(xin), P(wp), P(o1), P(ps1), P(pq1), C.byref(d), st)
    for S in (2, 4, 7):
        ws = torch.empty(S, M, Cout, device=dev)
        d.flags = (code << 8) | (S << 16)
        call.pn2_conv_gemm(BF16, P(xin), P(wp), P(o2), P(ws), C.c_void_p(0), C.byref(d), st)
        call.pn2_conv_splitk_reduce(BF16, P(ws), S, M, Cout, P(o2), Cout, C.c_void_p(0), P(ps2), P(pq2), 0, st)
        torch.cuda.synchronize()
        assert relmax(o2.float(), o1.float()) < 1e-2                      # one bf16 ulp where the fp32 sums round differently
        # the plain launch leaves (mean, M2) of its 64-row tiles (Chan partials), the split-K reduce raw moments of the same blocks
        nt = torch.full((nb, 1), 64.0, device=dev); nt[-1] = M - 64 * (nb - 1)
        s1 = (ps1.double() * nt).sum(0); s2 = (pq1.double() + nt * ps1.double() ** 2).sum(0)
        # both take the moments of the STORED (bf16-rounded) values: exact against the tensor the reduce wrote, and within the few one-ulp
        # differences between the two outputs of the plain launch's statistics
        o64 = o2.double()
        assert rell2(ps2.double().sum(0), o64.sum(0)) < 2e-6 and rell2(pq2.double().sum(0), (o64 * o64).sum(0)) < 2e-6
        assert rell2(ps2.double().sum(0), s1) < 2e-4 and rell2(pq2.double().sum(0), s2) < 2e-4


def test_full_size_properties_config4_bs16_352_bf16():
    """BASELINE config 4 shape (PVT-PraNet-V2, bs=16 per GPU, 352x352, bf16, DropPath 0): size-independent properties instead of an oracle run -
    finite, deterministic (bit for bit across two passes), output geometry, batch-permutation invariance of the batch-mean loss (BatchNorm
    statistics do not depend on the image order; LayerNorm / attention are per image), total == sum of the pair losses, hipGraph replay == eager."""
    from pn2.trainer import Trainer
    from oracle import weights as W
    x, mask = W.synthetic_batch(16, 352, seed=5)
    xg, mg = x.to(dev), mask.to(dev)
    tr = Trainer(_pvt_model(fp32=False), lr=1e-4, clip=0.5)
    l1 = tr.forward_backward(xg, mg).clone(); g1 = tr.gflat.clone(); o1 = tr.last_outs.clone()
    l2 = tr.forward_backward(xg, mg).clone(); g2 = tr.gflat.clone()
    assert torch.isfinite(l1).all() and torch.isfinite(g1).all() and float(g1.abs().max()) > 0
    assert torch.equal(l1, l2) and torch.equal(g1, g2), "kernels must be deterministic (no atomics on the data path)"
    assert o1.shape == (8, 16, 352, 352, 1)
    perm = torch.randperm(16, device=dev)
    l3 = tr.forward_backward(xg[perm], mg[perm])
    assert abs(float(l3[-1]) - float(l1[-1])) < 2e-2 * abs(float(l1[-1]))
    assert abs(float(l1[:4].sum()) - float(l1[4])) < 1e-5 * abs(float(l1[4]))
    # one optimizer step moves the weights and lowers nothing to NaN; replaying the captured step == the eager step, bit for bit
    tr.optimizer_step()
    ref = Trainer(_pvt_model(fp32=False), lr=1e-4, clip=0.5)
    for _ in range(3):
        le = ref.step(xg, mg)
    cap = Trainer(_pvt_model(fp32=False), lr=1e-4, clip=0.5)
    cap.capture(xg, mg, warmup=2)
    lg = cap.replay()
    torch.cuda.synchronize()
    assert torch.equal(le, lg) and torch.equal(ref.flat, cap.flat)
