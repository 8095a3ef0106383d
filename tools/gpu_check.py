#!/usr/bin/env python3
"""Bring-up script (GPU box): runs every engine op against the CPU oracle / plain torch fp32 and prints max errors.
Not a test (tests/ has the asserting versions) — it never stops at the first failure so one gpurun call shows everything."""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
os.environ.setdefault("PN2_NO_PRETRAINED", "1")
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from pn2 import F32, BF16
from pn2.engine import Engine
from pn2.graph import _seed_grad, set_compute_dtype
from oracle import pranet_oracle as O
from oracle import weights as W

dev = "cuda"
torch.manual_seed(0)
RESULTS = []


def report(name, err, tol):
    ok = err <= tol
    RESULTS.append((name, err, tol, ok))
    print(f"{'OK ' if ok else 'BAD'} {name:60s} err={err:.3e} tol={tol:.1e}", flush=True)


L2 = False   # bf16 runs: relative L2 error (ReLU-mask flips of near-zero activations make max-norm meaningless)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    if L2:
        return float((a - b).norm() / (b.norm() + 1e-12))
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def case(fn):
    try:
        fn()
    except Exception:
        print("EXC in", fn.__name__); traceback.print_exc(); RESULTS.append((fn.__name__, float("nan"), 0, False))


def conv_case(dt, N, Cin, Cout, k, stride, pad, dil, H, Wd, bn=False, relu=False):
    name = f"conv dt={dt} {Cin}->{Cout} k={k} s={stride} p={pad} d={dil} {N}x{H}x{Wd} bn={bn} relu={relu}"
    global L2
    L2 = dt == BF16
    tol = 2e-5 if dt == F32 else 3e-2
    conv = nn.Conv2d(Cin, Cout, k, stride, pad, dil, bias=False).to(dev)
    bnm = nn.BatchNorm2d(Cout).to(dev) if bn else None
    if bn:
        bnm.weight.data.uniform_(0.5, 1.5); bnm.bias.data.normal_(0, 0.2)
    x = torch.randn(N, Cin, H, Wd, device=dev)
    eng = Engine(dt, True, need_grad=True)
    a = eng.from_nchw(x, requires_grad=True)
    y = eng.conv_bn_act(a, conv, bnm, relu=relu)
    out = eng.to_nchw(y).clone()
    gy = torch.randn_like(out)
    _seed_grad(y, gy)
    eng.backward()
    torch.cuda.synchronize()
    gx = a.grad[..., :Cin].float().permute(0, 3, 1, 2)
    gw = eng.pgrads.get(conv.weight)
    # reference on CPU fp64
    xc = x.double().cpu().requires_grad_(True)
    wc = conv.weight.detach().double().cpu().requires_grad_(True)
    r = F.conv2d(xc, wc, None, stride, pad, dil)
    if bn:
        g_ = bnm.weight.detach().double().cpu().requires_grad_(True); b_ = bnm.bias.detach().double().cpu().requires_grad_(True)
        r = F.batch_norm(r, None, None, g_, b_, True, 0.1, 1e-5)
    if relu:
        r = F.relu(r)
    r.backward(gy.double().cpu())
    report(name + " fwd", rel(out, r), tol)
    report(name + " dgrad", rel(gx, xc.grad), tol)
    report(name + " wgrad", rel(gw, wc.grad), tol)
    if bn:
        report(name + " dgamma", rel(eng.pgrads.get(bnm.weight), g_.grad), tol)
        report(name + " dbeta", rel(eng.pgrads.get(bnm.bias), b_.grad), tol)


def convs():
    for dt in (F32, BF16):
        conv_case(dt, 2, 3, 32, 3, 2, 1, 1, 38, 38)
        conv_case(dt, 2, 32, 64, 3, 1, 1, 1, 19, 21)
        conv_case(dt, 2, 64, 256, 1, 1, 0, 1, 17, 17)
        conv_case(dt, 2, 256, 104, 1, 1, 0, 1, 9, 9, bn=True, relu=True)
        conv_case(dt, 3, 56, 56, 3, 2, 1, 1, 15, 15, bn=True, relu=True)
        conv_case(dt, 2, 256, 256, 5, 1, 2, 1, 11, 11, bn=True)
        conv_case(dt, 2, 32, 32, (1, 5), 1, (0, 2), 1, 11, 11, bn=True)
        conv_case(dt, 2, 32, 32, (7, 1), 1, (3, 0), 1, 11, 11, bn=True)
        conv_case(dt, 2, 32, 32, 3, 1, 5, 5, 22, 22, bn=True)
        conv_case(dt, 2, 128, 32, 3, 1, 1, 1, 22, 22, bn=True, relu=True)
        conv_case(dt, 1, 2048, 832, 1, 1, 0, 1, 11, 11)


def module_vs_oracle(name, mod, oracle_fn, inputs, dt, tol):
    """mod: pn2 nn.Module (train mode); oracle_fn(P, *cpu_inputs) -> tensor(s)."""
    global L2
    L2 = dt == BF16
    set_compute_dtype(dt)
    mod = mod.to(dev).train()
    P = {k: v.detach().cpu().clone() for k, v in mod.state_dict().items()}
    xs = [x.to(dev).requires_grad_(True) for x in inputs]
    outs = mod(*xs)
    outs = outs if isinstance(outs, (tuple, list)) else (outs,)
    gs = [torch.randn_like(o) for o in outs]
    torch.autograd.backward(list(outs), gs)
    torch.cuda.synchronize()
    keys = O.params_of(P)
    for k in keys:
        P[k].requires_grad_(True)
    xc = [x.detach().cpu().requires_grad_(True) for x in inputs]
    ro = oracle_fn(P, *xc)
    ro = ro if isinstance(ro, (tuple, list)) else (ro,)
    torch.autograd.backward(list(ro), [g.cpu() for g in gs])
    for i, (o, r) in enumerate(zip(outs, ro)):
        report(f"{name} dt={dt} out{i}", rel(o, r), tol)
    for i, (x, c) in enumerate(zip(xs, xc)):
        report(f"{name} dt={dt} dx{i}", rel(x.grad, c.grad), tol)
    worst, wk = 0.0, None
    named = dict(mod.named_parameters())
    for k in keys:
        if P[k].grad is None:
            continue
        e = rel(named[k].grad, P[k].grad)
        if e > worst:
            worst, wk = e, k
    report(f"{name} dt={dt} worst param grad ({wk})", worst, tol * 5)
    sd = mod.state_dict()
    worst = max(rel(sd[k].float(), P[k].detach().float()) for k in sd if "running" in k)
    report(f"{name} dt={dt} running stats", worst, tol)


def blocks():
    from lib.Res2Net_v1b import Bottle2neck
    from lib.pranet import RFB_modified, aggregation, BasicConv2d
    for dt, tol in ((F32, 5e-5), (BF16, 6e-2)):
        def rnd(m):
            for p in m.parameters():
                p.data = torch.randn_like(p) * (0.2 if p.ndim > 1 else 0.3) + (1.0 if p.ndim == 1 else 0.0)
            return m
        b = rnd(Bottle2neck(64, 16))
        case(lambda: module_vs_oracle("bottle2neck normal w6", b, lambda P, x: O.bottle2neck(P, "", x, O.Ctx(True), 1, False, False), [torch.randn(2, 64, 12, 12)], dt, tol))
        b = rnd(Bottle2neck(256, 64))
        case(lambda: module_vs_oracle("bottle2neck normal w26", b, lambda P, x: O.bottle2neck(P, "", x, O.Ctx(True), 1, False, False), [torch.randn(2, 256, 10, 10)], dt, tol))
        down = nn.Sequential(nn.AvgPool2d(2, 2, ceil_mode=True, count_include_pad=False), nn.Conv2d(64, 128, 1, bias=False), nn.BatchNorm2d(128))
        b = rnd(Bottle2neck(64, 32, stride=2, downsample=down, stype="stage"))
        case(lambda: module_vs_oracle("bottle2neck stage s2 odd", b, lambda P, x: O.bottle2neck(P, "", x, O.Ctx(True), 2, True, True), [torch.randn(2, 64, 13, 13)], dt, tol))
        r = rnd(RFB_modified(48, 32))
        case(lambda: module_vs_oracle("rfb", r, lambda P, x: O.rfb(P, "", x, O.Ctx(True)), [torch.randn(2, 48, 11, 11)], dt, tol))
        a = rnd(aggregation(32, 1))
        case(lambda: module_vs_oracle("aggregation", a, lambda P, x1, x2, x3: O.aggregation(P, "", x1, x2, x3, O.Ctx(True)),
                                      [torch.randn(2, 32, 3, 3), torch.randn(2, 32, 6, 6), torch.randn(2, 32, 12, 12)], dt, tol))


def misc_ops():
    global L2
    for dt, tol in ((F32, 1e-5), (BF16, 2e-2)):
        L2 = dt == BF16
        eng = Engine(dt, True, need_grad=True)
        x = torch.randn(2, 16, 13, 15, device=dev)
        a = eng.from_nchw(x, True)
        y = eng.maxpool3x3s2(a)
        o = eng.to_nchw(y).clone(); g = torch.randn_like(o); _seed_grad(y, g); eng.backward()
        xc = x.cpu().requires_grad_(True)
        if dt == BF16: xc = x.bfloat16().float().cpu().requires_grad_(True)
        r = F.max_pool2d(xc, 3, 2, 1); r.backward(g.cpu())
        report(f"maxpool dt={dt} fwd", rel(o, r), tol); report(f"maxpool dt={dt} bwd", rel(a.grad.float().permute(0, 3, 1, 2), xc.grad), tol)
        for (k, s, p, ceil, inc, H) in ((3, 1, 1, False, True, 12), (3, 2, 1, False, True, 13), (2, 2, 0, True, False, 13), (2, 2, 0, True, False, 12)):
            eng = Engine(dt, True, need_grad=True)
            x = torch.randn(2, 8, H, H + 1, device=dev); a = eng.from_nchw(x, True)
            y = eng.avgpool(a, k, s, p, ceil, inc)
            o = eng.to_nchw(y).clone(); g = torch.randn_like(o); _seed_grad(y, g); eng.backward()
            xc = x.cpu().requires_grad_(True)
            r = F.avg_pool2d(xc, k, s, p, ceil, inc); r.backward(g.cpu())
            report(f"avgpool k{k}s{s}p{p} ceil={ceil} H={H} dt={dt} fwd", rel(o, r), tol)
            report(f"avgpool k{k}s{s}p{p} ceil={ceil} H={H} dt={dt} bwd", rel(a.grad.float().permute(0, 3, 1, 2), xc.grad), tol)
        for (scale, ac, C, H) in ((2, True, 32, 11), (2, False, 8, 11), (0.25, False, 8, 44), (8, False, 8, 11), (32, False, 8, 5)):
            eng = Engine(dt, True, need_grad=True)
            x = torch.randn(2, C, H, H, device=dev); a = eng.from_nchw(x, True)
            y = eng.bilinear(a, scale, ac)
            o = eng.to_nchw(y).clone(); g = torch.randn_like(o); _seed_grad(y, g); eng.backward()
            xc = x.cpu().requires_grad_(True)
            r = F.interpolate(xc, scale_factor=scale, mode="bilinear", align_corners=ac); r.backward(g.cpu())
            report(f"bilinear x{scale} ac={ac} C={C} dt={dt} fwd", rel(o, r), tol)
            report(f"bilinear x{scale} ac={ac} C={C} dt={dt} bwd", rel(a.grad.float().permute(0, 3, 1, 2), xc.grad), tol)


def tail_ops():
    global L2
    L2 = False
    from pn2.engine import Act
    z = np.load(os.path.join(ROOT, "tests/golden/dsra_k9.npz"))
    for sm, tag in ((True, "sm"), (False, "nosm")):
        eng = Engine(F32, True, need_grad=True)
        mk = lambda k: Act(eng, torch.from_numpy(z[k]).to(dev).permute(0, 2, 3, 1).contiguous(), 9, 9, 9, F32)
        fg, cf, cb = mk("fg"), mk("crop_fg"), mk("crop_bg")
        y = eng.dsra_fuse(fg, cf, cb, sm)
        _seed_grad(y, torch.from_numpy(z["gout"]).to(dev)); eng.backward()
        report(f"dsra K=9 {tag} fwd", rel(y.t.permute(0, 3, 1, 2), torch.from_numpy(z[tag + "_y"])), 1e-5)
        for a, k in ((fg, "gfg"), (cf, "gcf"), (cb, "gcb")):
            report(f"dsra K=9 {tag} {k}", rel(a.grad.permute(0, 3, 1, 2), torch.from_numpy(z[f"{tag}_{k}"])), 1e-5)
    # structure loss vs golden
    from pn2.loss import structure_loss_multi
    z = np.load(os.path.join(ROOT, "tests/golden/structure_loss.npz"))
    for tag in ("rand", "zeros", "ones"):
        pred = torch.from_numpy(z[f"{tag}_pred"]).to(dev).requires_grad_(True)
        pbg = torch.from_numpy(z[f"{tag}_pred_bg"]).to(dev).requires_grad_(True)
        mask = torch.from_numpy(z[f"{tag}_mask"]).to(dev)
        loss = structure_loss_multi([pred], [pbg], mask)
        loss.backward()
        report(f"structure_loss {tag} value", abs(float(loss) - float(z[f"{tag}_loss"])), 2e-6)
        report(f"structure_loss {tag} dpred", rel(pred.grad, torch.from_numpy(z[f"{tag}_gpred"])), 2e-5)
        report(f"structure_loss {tag} dpred_bg", rel(pbg.grad, torch.from_numpy(z[f"{tag}_gpred_bg"])), 2e-5)


def whole_model(tag, dt, tol):
    global L2
    L2 = False
    from lib.pranet import PraNet_V2
    from pn2.loss import structure_loss_multi
    set_compute_dtype(dt)
    z = np.load(os.path.join(ROOT, f"tests/golden/pranet_v2_{tag}.npz"))
    size, n = int(z["size"]), int(z["n"])
    model = PraNet_V2(num_class=1)
    model.load_state_dict(W.make_state_dict(W.manifest_pranet_v2(1), seed=0), strict=True)
    model = model.to(dev).train()
    x, mask = W.synthetic_batch(n, size, seed=1234)
    x, mask = x.to(dev), mask.to(dev)
    t0 = time.time()
    outs = model(x)
    loss = structure_loss_multi(list(outs[:4]), list(outs[4:]), mask)
    loss.backward()
    torch.cuda.synchronize()
    print(f"  model {tag} dt={dt}: fwd+bwd wall {time.time() - t0:.2f}s loss={float(loss):.6f} (golden {float(z['s1.loss']):.6f})")
    report(f"model {tag} dt={dt} loss", abs(float(loss) - float(z["s1.loss"])), tol * 10)
    full = tag == "96"
    for i, o in enumerate(outs):
        r32 = torch.from_numpy(z[f"s1.out{i}"]).double(); r64 = torch.from_numpy(z[f"f64.out{i}"])
        got = (o.detach().cpu() if full else o.detach().cpu()[:, :, ::4, ::4]).double()
        own = float((r32 - r64).abs().max())
        report(f"model {tag} dt={dt} out{i} |ours-ref64| (ref32-ref64={own:.1e})", float((got - r64).abs().max()), max(tol, 3 * own) if dt == F32 else tol)
    named = dict(model.named_parameters())
    for f in z.files:
        if f.startswith("graw."):
            k = f[5:]
            r32 = torch.from_numpy(z[f]).double(); r64 = torch.from_numpy(z["f64." + f]).double()
            got = named[k].grad.reshape(-1)[:256].cpu().double()
            sc = float(r64.abs().max()) + 1e-12
            own = float((r32 - r64).abs().max()) / sc
            report(f"model {tag} dt={dt} grad {k} (ref32 own {own:.1e})", float((got - r64).abs().max()) / sc, max(1e-4, 3 * own) if dt == F32 else 0.25)


if __name__ == "__main__":
    which = sys.argv[1:] or ["convs", "misc", "tail", "blocks", "m96", "m352"]
    print("device:", torch.cuda.get_device_name(0))
    if "convs" in which: case(convs)
    if "misc" in which: case(misc_ops)
    if "tail" in which: case(tail_ops)
    if "blocks" in which: case(blocks)
    if "m96" in which: case(lambda: whole_model("96", F32, 1e-4))
    if "m352" in which: case(lambda: whole_model("352", F32, 1e-4))
    if "m352bf" in which: case(lambda: whole_model("352", BF16, 0.3))
    bad = [r for r in RESULTS if not r[3]]
    print(f"\n{len(RESULTS) - len(bad)} ok, {len(bad)} bad")
    for r in bad:
        print("  BAD", r[0], r[1])
