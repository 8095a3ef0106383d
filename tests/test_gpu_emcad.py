"""GPU parity tests of the EMCAD decoder path (BASELINE config 5): each decoder block through the C ABI against the oracle's restatement
(float64 on the CPU), then the whole EMCADNet(dual, K=9) training forward/backward against the vectors of the imported reference."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

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


def _randomize(mod, seed):
    g = torch.Generator().manual_seed(seed)
    for n, p in mod.named_parameters():
        if p.dim() > 1:
            fan = p[0].numel()
            p.data = torch.randn(p.shape, generator=g) * (1.0 / fan) ** 0.5
        elif n.endswith("weight"):
            p.data = torch.rand(p.shape, generator=g) * 0.8 + 0.6
        else:
            p.data = torch.randn(p.shape, generator=g) * 0.1
    return mod.to(dev).train()


def _check(dtn, mod, build, ref, xs):
    """build(eng, *acts) vs ref(P64, *x64) where P64 is the module's state_dict in float64 (oracle conventions)."""
    from pn2 import F32, BF16
    from pn2.engine import Engine
    from pn2.graph import _seed_grad
    dt = F32 if dtn == "fp32" else BF16
    err, tol = (relmax, 1e-4) if dt == F32 else (rell2, 4e-2)
    eng = Engine(dt, True, need_grad=True)
    acts = [eng.from_nchw(x, requires_grad=True) for x in xs]
    y = build(eng, *acts)
    out = eng.to_nchw(y).clone()
    torch.manual_seed(1234)
    gy = torch.randn_like(out)
    _seed_grad(y, gy)
    eng.backward()
    cast = (lambda t: t.bfloat16().float()) if dt == BF16 else (lambda t: t)
    x64 = [cast(x).double().cpu().requires_grad_(True) for x in xs]
    P = {k: (v.detach().double().cpu().clone() if v.dtype.is_floating_point else v.detach().cpu().clone()) for k, v in mod.state_dict().items()}
    for k, v in P.items():
        if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    r = ref(P, *x64)
    r.backward(gy.double().cpu())
    assert err(out, r) < tol, "forward"
    for a, x, xr in zip(acts, xs, x64):
        assert err(a.grad[..., :x.shape[1]].float().permute(0, 3, 1, 2), xr.grad) < (tol if dt == F32 else 8e-2), "input gradient"
    names = dict(mod.named_parameters())
    for k, p in names.items():
        g = eng.pgrads.get(p)
        assert g is not None, k
        scale = float(P[k].grad.abs().max())
        if scale < 1e-6:          # biases in front of a train-mode BN: analytically zero
            assert float(g.abs().max()) < (1e-4 if dt == F32 else 5e-2), k
        else:
            assert err(g, P[k].grad) < (tol if dt == F32 else 0.2), k      # bf16: ReLU / ReLU6 mask flips at 0 and 6 through three stacked BNs


@pytest.mark.parametrize("dtn", ["fp32", "bf16"])
def test_mscb_and_eucb(dtn):
    from lib.decoders import MSCB, EUCB
    from oracle import emcad_oracle as E
    from oracle.pranet_oracle import Ctx
    m = _randomize(MSCB(64, 64, 1, kernel_sizes=[1, 3, 5], expansion_factor=2, activation="relu6"), 1)
    x = torch.randn(2, 64, 9, 7, device=dev) * 1.5
    _check(dtn, m, lambda e, a: m._build(e, a), lambda P, t: E.mscb(P, "", t, Ctx(True)), [x])
    u = _randomize(EUCB(64, 32), 2)
    _check(dtn, u, lambda e, a: u._build(e, a), lambda P, t: E.eucb(P, "", t, Ctx(True)), [torch.randn(2, 64, 5, 6, device=dev)])


@pytest.mark.parametrize("dtn", ["fp32", "bf16"])
def test_gates_lgag_cab_sab(dtn):
    from lib.decoders import LGAG, CAB, SAB
    from oracle import emcad_oracle as E
    from oracle.pranet_oracle import Ctx
    l = _randomize(LGAG(64, 64, 32, kernel_size=3, groups=32), 3)
    g, x = torch.randn(2, 64, 7, 6, device=dev), torch.randn(2, 64, 7, 6, device=dev)
    _check(dtn, l, lambda e, a, b: l._build(e, a, b), lambda P, a, b: E.lgag(P, "", a, b, Ctx(True)), [g, x])
    c = _randomize(CAB(128), 4)
    xc = torch.randn(3, 128, 6, 5, device=dev)
    _check(dtn, c, lambda e, a: c._build_gated(e, a), lambda P, t: E.cab(P, "", t) * t, [xc])
    s = _randomize(SAB(), 5)
    _check(dtn, s, lambda e, a: s._build_gated(e, a), lambda P, t: E.sab(P, "", t) * t, [torch.randn(2, 64, 9, 8, device=dev)])


def test_emcad_dual_decoder_vs_oracle_fp32():
    """The whole decoder (4 stages, LGAG/CAB/SAB gates, K=9 DSRA heads) on well-sized random encoder features: forward and all parameter gradients
    against the oracle in float64.  (The 64x64 whole-model vectors below run its deepest stage on 2x2 maps, where train-mode BN is ill-conditioned.)"""
    from pn2 import F32
    from pn2.engine import Engine
    from pn2.graph import _seed_grad
    from lib.decoders import EMCAD_dual
    from oracle import emcad_oracle as E
    from oracle.pranet_oracle import Ctx
    dec = _randomize(EMCAD_dual(channels=[512, 320, 128, 64], kernel_sizes=[1, 3, 5], expansion_factor=2, activation="relu6", num_class=9), 7)
    torch.manual_seed(8)
    feats = [torch.randn(3, c, s, s, device=dev) for c, s in ((512, 6), (320, 12), (128, 24), (64, 48))]
    eng = Engine(F32, True, need_grad=True)
    acts = [eng.from_nchw(f, requires_grad=True) for f in feats]
    outs = dec._build(eng, acts[0], acts[1:])
    o_t = [eng.to_nchw(o).clone() for o in outs]
    torch.manual_seed(99)
    gys = [torch.randn_like(o) for o in o_t]
    for o, g in zip(outs, gys):
        _seed_grad(o, g)
    eng.backward()
    def oracle(dtype):
        P = {k: (v.detach().to(dtype).cpu().clone() if v.dtype.is_floating_point else v.detach().cpu().clone()) for k, v in dec.state_dict().items()}
        for k, v in P.items():
            if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var")):
                v.requires_grad_(True)
        fx = [f.to(dtype).cpu().requires_grad_(True) for f in feats]
        ref = E.emcad_dual(P, "", fx[0], fx[1:], Ctx(True))
        sum((r * g.to(dtype).cpu()).sum() for r, g in zip(ref, gys)).backward()
        return P, fx, ref
    P, f64, ref = oracle(torch.float64)
    P32, f32, _ = oracle(torch.float32)
    # ReLU6 / max-pool / gate discontinuities after train-mode BN over ~100 samples make the deep-stage gradients ill-conditioned: the oracle's own
    # fp32 run is up to ~1e-2 away from its float64 run here, so (as for the whole models) the bound is a multiple (8x: the
    # error grows stage by stage, 2e-4 at the shallowest input to 8e-3 at the deepest) of that measured distance
    for i, (o, r) in enumerate(zip(o_t, ref)):
        assert relmax(o, r) < 2e-4, i
    for a, f, fr, fr32 in zip(acts, feats, f64, f32):
        assert rell2(a.grad[..., :f.shape[1]].permute(0, 3, 1, 2), fr.grad) < max(1e-2, 8 * rell2(fr32.grad, fr.grad))
    for k, p in dec.named_parameters():
        g = eng.pgrads.get(p)
        scale = float(P[k].grad.abs().max())
        if scale < 1e-6:          # conv biases in front of a train-mode BN: analytically zero, other gradients here are O(1e2)
            assert float(g.abs().max()) < 2e-3, k
        else:
            # floor: a single ReLU6 mask flip (|x - 6| ~ 1e-6) moves a bias gradient by ~1e-3; the one-element BatchNorm of the LGAG psi branch
            # (one number summed over every pixel behind those flips) moves by up to ~1.5e-2 when only the summation order of the statistics changes
            floor = 2.5e-2 if p.numel() == 1 else 1e-2
            assert rell2(g, P[k].grad) < max(floor, 8 * rell2(P32[k].grad, P[k].grad)), k


def test_mutation_loss_kernels_vs_reference_formula():
    """pn2.loss.mutation_loss (one forward pass + one backward pass over the 8 maps) against the oracle's restatement of trainer.py:106-140
    evaluated in float64: value and the gradient of every map."""
    from pn2.loss import mutation_loss
    from oracle import emcad_oracle as E
    torch.manual_seed(3)
    N, K, H, W = 3, 9, 20, 28
    maps = [(torch.randn(N, K, H, W, device=dev) * 1.5).requires_grad_(True) for _ in range(8)]
    label = torch.randint(0, K, (N, H, W), device=dev)
    bg = torch.stack([(label != k).float() for k in range(K)], 1)
    loss = mutation_loss(maps, label, bg)
    loss.backward()
    m64 = [m.detach().double().cpu().requires_grad_(True) for m in maps]
    ref = E.mutation_loss(m64, label.cpu(), bg.double().cpu())
    ref.backward()
    assert abs(float(loss) - float(ref)) < 1e-5 * float(ref)
    for a, b in zip(maps, m64):
        assert relmax(a.grad, b.grad) < 2e-5


def _model(fp32=True):
    import pn2
    from lib.networks import EMCADNet
    from oracle import weights as W
    pn2.set_compute_dtype("fp32" if fp32 else "bf16")
    m = EMCADNet(num_classes=9, kernel_sizes=[1, 3, 5], expansion_factor=2, dw_parallel=True, add=True, lgag_ks=3, activation="relu6", encoder="pvt_v2_b2",
                 pretrain=False, dual=True)
    m.load_state_dict(W.make_state_dict(W.manifest_emcadnet(9), seed=5), strict=True)
    m.backbone.reset_drop_path(0.0)
    return m.to(dev).train()


def test_emcad_state_dict_manifest():
    import json
    from lib.networks import EMCADNet
    ref = json.load(open(os.path.join(G, "manifest_emcad.json")))["emcadnet_dual_k9"]
    m = EMCADNet(num_classes=9, activation="relu6", pretrain=False, dual=True)
    assert [(k, list(v.shape)) for k, v in m.state_dict().items()] == list(ref.items())


@pytest.mark.parametrize("fp32", [True, False])
def test_emcadnet_forward_backward_vs_reference(fp32):
    """EMCADNet.forward(dual) + the reference trainer's 15-subset CE + Dice + BCE loss (torch ops on the module outputs, as trainer.py does) + backward."""
    from oracle import emcad_oracle as E
    z = np.load(os.path.join(G, "emcad_64.npz"))
    model = _model(fp32)
    x = torch.from_numpy(z["x"]).to(dev); label = torch.from_numpy(z["label"]).to(dev); bg = torch.from_numpy(z["bg_mask"]).to(dev)
    outs = model(x, mode="train")
    from pn2.loss import mutation_loss
    loss = mutation_loss(outs, label, bg)            # trainer.py:106-140 as the fused kernel pair
    loss.backward()
    names = dict(model.named_parameters())
    if fp32:
        for i, o in enumerate(outs):
            ref64 = torch.from_numpy(z[f"f64.out{i}"])
            own = float((torch.from_numpy(z[f"out{i}"]).double() - ref64).abs().max())
            assert float((o.detach().double().cpu() - ref64).abs().max()) <= max(1e-4, 3 * own), i
        assert abs(float(loss) - float(z["f64.loss"])) < max(1e-4, 3 * abs(float(z["loss"]) - float(z["f64.loss"])))
        for k in z.files:
            if k.startswith("f64.grawnorm."):
                name = k[len("f64.grawnorm."):]
                g = names[name].grad
                r64, r32 = float(z[k]), float(z["grawnorm." + name])
                # 64x64 inputs put the deepest decoder stage on 2x2 maps (8 samples per BatchNorm channel) behind ReLU6 / max-pool / gate
                # discontinuities: the reference's own fp32 gradients are 0.5-4 % away from its float64 run on these probes, so this is a
                # sanity band; the tight bound is test_emcad_dual_decoder_vs_oracle_fp32 (8x the oracle's fp32 distance at healthy sizes)
                assert abs(float(g.norm()) - r64) <= max(1e-2 * r64, 3 * abs(r32 - r64)) + 2e-6, name
                h64 = torch.from_numpy(z["f64.graw." + name]).double(); h32 = torch.from_numpy(z["graw." + name]).double()
                ours = g.detach().reshape(-1)[:h64.numel()].double().cpu()
                assert float((ours - h64).norm()) <= max(6e-2 * float(h64.norm()), 3 * float((h32 - h64).norm())) + 2e-6, name
    else:
        rels = [rell2(o, torch.from_numpy(z[f"f64.out{i}"])) for i, o in enumerate(outs)]
        print("\nEMCADNet bf16 64x64, rel-L2 of the 8 maps against the reference's float64 run:", " ".join(f"{v:.3f}" for v in rels))
        # bf16 on 2x2 .. 16x16 train-mode-BN maps: sanity band only.  The last map of each head moves between 0.22 and 0.27 with bit-level choices that change no
        # arithmetic (PN2_KS2=0: 0.218 -> 0.247; the depth-wise window kernels, which only fuse a few more multiply-adds: 0.221 -> 0.254; both: 0.274)
        for i, v in enumerate(rels):
            assert v < 0.35, i
        assert abs(float(loss) - float(z["loss"])) < 5e-2 * float(z["loss"])


def test_trainer_mutation_step_matches_module_surface_and_adamw():
    """pn2.trainer.Trainer(loss="mutation") - step arena, deferred table-driven launches, loss kernels writing the map gradients in place,
    AdamW kernel, hipGraph replay - against the nn.Module surface + pn2.loss.mutation_loss + torch.optim.AdamW on the same weights and batch
    (fp32 compute): loss, every parameter gradient, the parameters after two steps, and eager == graph replay bit for bit."""
    from pn2.loss import mutation_loss
    from pn2.trainer import Trainer
    z = np.load(os.path.join(G, "emcad_64.npz"))
    x = torch.from_numpy(z["x"]).to(dev); label = torch.from_numpy(z["label"]).to(dev); bg = torch.from_numpy(z["bg_mask"]).to(dev)
    lr, wd = 1e-3, 1e-2
    # ---- module surface
    ma = _model(True)
    hot_a = ma.hot_parameters(x.shape[1] == 1)
    opt = torch.optim.AdamW(hot_a, lr=lr, weight_decay=wd)
    losses_a = []
    for _ in range(2):
        loss = mutation_loss(ma(x, mode="train"), label, bg)
        opt.zero_grad(); loss.backward()
        if not losses_a:
            g_a = [p.grad.clone() for p in hot_a]
        opt.step(); losses_a.append(float(loss))
    # ---- trainer
    mb = _model(True)
    hot_b = mb.hot_parameters(x.shape[1] == 1)
    tr = Trainer(mb, lr=lr, clip=None, weight_decay=wd, loss="mutation", hot=hot_b)
    l1 = tr.forward_backward(x, (label, bg))
    torch.cuda.synchronize()
    assert abs(float(l1[0]) - losses_a[0]) < 1e-6 * abs(losses_a[0])
    for (n, _), pa, pb in zip([(n, p) for n, p in mb.named_parameters() if any(p is q for q in hot_b)], g_a, hot_b):
        gb = tr._grad_view(pb)
        scale = float(pa.abs().max())
        # same kernels; the launch grouping differs, and the table-driven wgrads run fewer pixel splits (GradQueue.table_splits): another fp32 summation order
        # (absolute floor: conv.0.weight's three gradients are ~4e-6, sums of O(0.1) terms that cancel - 2.4e-7 apart between the two groupings)
        assert float((gb - pa).abs().max()) <= 1e-5 * max(scale, 1e-3) + 1e-6, n
    tr.optimizer_step()
    l2 = tr.step(x, (label, bg))
    torch.cuda.synchronize()
    assert abs(float(l2[0]) - losses_a[1]) < 1e-5 * abs(losses_a[1])
    # Adam divides by sqrt(v): where a gradient is rounding noise (conv biases in front of a train-mode BN: analytically zero) the update is +-lr
    # whatever the noise says, so individual elements may differ by up to 2 lr per step; everything else agrees to fp32 accuracy
    bad = tot = 0
    for pa, pb in zip(hot_a, hot_b):
        d = (pb.data - pa.data).abs()
        assert float(d.max()) <= 4.5 * lr
        bad += int((d > 0.05 * lr).sum()); tot += d.numel()          # an update that is off by more than 5 % of the step size
    assert bad <= 1e-2 * tot, (bad, tot)
    # ---- hipGraph replay == eager, bit for bit (bf16, two trainers from identical weights)
    res = []
    for graph in (False, True):
        m2 = _model(False)
        tr2 = Trainer(m2, lr=lr, clip=None, weight_decay=wd, loss="mutation", hot=m2.hot_parameters(True))
        if graph:
            tr2.capture(x, (label, bg), warmup=2)
            out = tr2.replay(x, (label, bg))
        else:
            for _ in range(3):
                out = tr2.step(x, (label, bg))
        torch.cuda.synchronize()
        res.append((out.clone(), tr2.flat.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("dtn", ["fp32", "bf16"])
def test_depthwise_kxk_and_pair_conv_kernels_at_edge_geometries(dtn):
    """pn2_dwconv (K = 1, 3, 5: forward with BatchNorm partial rows, mirrored data gradient, weight gradient) and pn2_pairconv3x3_* straight
    through the C ABI against torch, at sizes that stress the row-segment walks: single pixels / rows / columns, widths that are not a multiple
    of the segment length, channel counts that select every vector width."""
    import ctypes as C
    import torch.nn.functional as F
    from pn2.capi import call, F32, BF16
    dt, tdt = (F32, torch.float32) if dtn == "fp32" else (BF16, torch.bfloat16)
    tol = 2e-5 if dt == F32 else 2e-2
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator(device="cpu").manual_seed(9)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous()
    for N, H, W, Cc in ((1, 1, 1, 8), (2, 1, 29, 16), (1, 21, 1, 8), (2, 7, 33, 24), (2, 19, 18, 40), (1, 40, 52, 128)):
        for K in (1, 3, 5):
            x = torch.randn(N, Cc, H, W, generator=g).to(dev).to(tdt).float()
            w = (torch.randn(Cc, 1, K, K, generator=g) * 0.4).to(dev)
            dz = torch.randn(N, Cc, H, W, generator=g).to(dev).to(tdt).float()
            xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
            zr = F.conv2d(xr, wr, None, 1, K // 2, groups=Cc)
            zr.backward(dz)
            xh, dzh = nhwc(x).to(tdt), nhwc(dz).to(tdt)
            z = torch.empty_like(xh); dx = torch.empty_like(xh)
            nb = call.pn2_dwconv_blocks(dt, N, H, W, Cc, K, 0)
            ps, pq = torch.zeros(nb, Cc, device=dev), torch.zeros(nb, Cc, device=dev)
            wf = w.reshape(Cc, K * K).contiguous()
            call.pn2_dwconv(dt, P(xh), P(wf), P(z), N, H, W, Cc, K, 0, 0, P(ps), P(pq), st)
            call.pn2_dwconv(dt, P(dzh), P(wf), P(dx), N, H, W, Cc, K, 1, 0, C.c_void_p(0), C.c_void_p(0), st)
            nbw = call.pn2_dwconv_blocks(dt, N, H, W, Cc, K, 1)
            part = torch.zeros(nbw, Cc * K * K, device=dev)
            call.pn2_dwconv_wgrad(dt, P(dzh), P(xh), P(part), N, H, W, Cc, K, st)
            torch.cuda.synchronize()
            tag = (N, H, W, Cc, K)
            assert rell2(z.float().permute(0, 3, 1, 2), zr) < tol, tag
            assert rell2(dx.float().permute(0, 3, 1, 2), xr.grad) < tol, tag
            assert rell2(part.sum(0).reshape(Cc, 1, K, K), wr.grad) < tol, tag
            zs = z.float().reshape(-1, Cc)
            assert rell2(ps.sum(0), zs.sum(0)) < 1e-4 and rell2(pq.sum(0), (zs * zs).sum(0)) < 1e-4, tag
    for N, H, W, Fo in ((1, 1, 1, 8), (2, 1, 31, 8), (1, 17, 1, 16), (2, 9, 35, 24), (1, 33, 40, 64)):
        x = torch.randn(N, 2 * Fo, H, W, generator=g).to(dev).to(tdt).float()
        w = (torch.randn(Fo, 2, 3, 3, generator=g) * 0.3).to(dev)
        dz = torch.randn(N, Fo, H, W, generator=g).to(dev).to(tdt).float()
        xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
        zr = F.conv2d(xr, wr, None, 1, 1, groups=Fo)
        zr.backward(dz)
        xh, dzh = nhwc(x).to(tdt), nhwc(dz).to(tdt)
        z = torch.empty(N, H, W, Fo, device=dev, dtype=tdt); dx = torch.empty_like(xh)
        nb = call.pn2_pairconv_blocks(dt, N, H, W, Fo)
        ps, pq = torch.zeros(nb, Fo, device=dev), torch.zeros(nb, Fo, device=dev)
        wf = w.reshape(Fo, 18).contiguous()
        call.pn2_pairconv3x3_fwd(dt, P(xh), P(wf), P(z), N, H, W, Fo, P(ps), P(pq), st)
        call.pn2_pairconv3x3_dgrad(dt, P(dzh), P(wf), P(dx), N, H, W, Fo, 0, st)
        part = torch.zeros(nb, Fo * 18, device=dev)
        call.pn2_pairconv3x3_wgrad(dt, P(dzh), P(xh), P(part), N, H, W, Fo, st)
        torch.cuda.synchronize()
        tag = ("pair", N, H, W, Fo)
        assert rell2(z.float().permute(0, 3, 1, 2), zr) < tol, tag
        assert rell2(dx.float().permute(0, 3, 1, 2), xr.grad) < tol, tag
        assert rell2(part.sum(0).reshape(Fo, 2, 3, 3), wr.grad) < tol, tag
        zs = z.float().reshape(-1, Fo)
        assert rell2(ps.sum(0), zs.sum(0)) < 1e-4 and rell2(pq.sum(0), (zs * zs).sum(0)) < 1e-4, tag


def test_full_size_properties_config5_bs16_512_k9_bf16():
    """BASELINE config 5 shape (EMCADNet dual, K = 9, bs=16 per GPU, 512x512 1-channel slices, bf16, the 15-subset loss + AdamW): finite,
    deterministic, output geometry (8 maps of [16, 512, 512, 9]), sample-permutation invariance of the batch-mean loss, hipGraph replay == eager."""
    from pn2.trainer import Trainer
    g = torch.Generator(device="cpu").manual_seed(77)
    N, S, K = 16, 512, 9
    x = torch.randn(N, 1, S, S, generator=g).to(dev)
    lab = torch.randint(0, K, (N, S // 16, S // 16), generator=g).to(dev)
    lab = torch.nn.functional.interpolate(lab[:, None].float(), size=(S, S), mode="nearest")[:, 0].long()
    bg = torch.stack([(lab != k).float() for k in range(K)], 1)

    def trainer():
        m = _model(False)
        return Trainer(m, lr=1e-4, clip=None, weight_decay=1e-4, loss="mutation", hot=m.hot_parameters(True))
    tr = trainer()
    l1 = tr.forward_backward(x, (lab, bg)).clone(); g1 = tr.gflat.clone()
    l2 = tr.forward_backward(x, (lab, bg)).clone(); g2 = tr.gflat.clone()
    assert torch.isfinite(l1).all() and torch.isfinite(g1).all() and float(g1.abs().max()) > 0
    assert torch.equal(l1, l2) and torch.equal(g1, g2), "kernels must be deterministic (no atomics on the data path)"
    perm = torch.randperm(N, device=dev)
    l3 = tr.forward_backward(x[perm], (lab[perm], bg[perm]))
    assert abs(float(l3[0]) - float(l1[0])) < 2e-2 * abs(float(l1[0]))
    ref = trainer()
    for _ in range(3):
        le = ref.step(x, (lab, bg))
    cap = trainer()
    cap.capture(x, (lab, bg), warmup=2)
    lg = cap.replay(x, (lab, bg))
    torch.cuda.synchronize()
    assert torch.equal(le, lg) and torch.equal(ref.flat, cap.flat)
