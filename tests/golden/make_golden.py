#!/usr/bin/env python3
"""Golden-vector generator.  Runs ONLY in the build container (needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Imports the reference (ai4colonoscopy/PraNet-V2, /root/reference/binary_seg) through
tests/golden/_ref_import.py, drives ITS classes/functions on seeded inputs and writes the
inputs' recipe + expected outputs as small .npz/.json fixtures next to this file.  Only
data is stored — no reference source.  Weights are regenerated on both sides from
oracle/weights.py (seeded per-key generators), so the 130 MB state_dict is never stored.
"""
import json, os, sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE); sys.path.insert(0, ROOT)
from _ref_import import import_reference            # noqa: E402
from oracle import weights as W                      # noqa: E402

torch.set_num_threads(8)
R = import_reference()


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrs):
    np.savez_compressed(os.path.join(HERE, name), **arrs)
    print("wrote", name, len(arrs), "arrays")


PROBE_PARAMS = [
    "backbone.conv1.0.weight", "backbone.conv1.1.weight", "backbone.bn1.bias",
    "backbone.layer1.0.conv1.weight", "backbone.layer1.0.convs.1.weight", "backbone.layer1.0.downsample.1.weight",
    "backbone.layer1.2.bns.2.weight", "backbone.layer2.0.convs.0.weight", "backbone.layer2.3.conv3.weight",
    "backbone.layer3.0.downsample.1.weight", "backbone.layer3.5.convs.2.weight", "backbone.layer4.0.convs.1.weight",
    "backbone.layer4.2.conv3.weight", "backbone.layer4.2.bn3.weight",
    "rfb2_1.branch0.0.conv.weight", "rfb3_1.branch2.1.conv.weight", "rfb3_1.branch2.2.conv.weight",
    "rfb4_1.branch3.3.conv.weight", "rfb4_1.conv_cat.conv.weight", "rfb2_1.conv_res.bn.bias",
    "agg1.conv_upsample1.conv.weight", "agg1.conv_upsample5.conv.weight", "agg1.conv_concat3.conv.weight",
    "agg1.conv4.bn.weight", "agg1.conv5_fg.weight", "agg1.conv5_fg.bias", "agg1.conv5_bg.weight",
    "ra4_conv1.conv.weight", "ra4_conv3.conv.weight", "ra4_conv5_fg.conv.weight", "ra4_conv5_bg.bn.bias",
    "ra3_conv2.conv.weight", "ra3_conv4_fg.conv.weight", "ra3_conv4_bg.conv.weight",
    "ra2_conv1.conv.weight", "ra2_conv4_fg.conv.weight", "ra2_conv4_fg.bn.weight", "ra2_conv4_bg.bn.weight",
]
PROBE_BUFFERS = ["backbone.bn1.running_mean", "backbone.bn1.running_var", "backbone.layer3.2.bns.1.running_var",
                 "rfb4_1.conv_cat.bn.running_mean", "ra2_conv4_fg.bn.running_var", "agg1.conv4.bn.running_mean"]
NPROBE = 256


def head(t):
    return npy(t.reshape(-1)[:NPROBE].clone())


# ---------------------------------------------------------------------------- manifests
def gen_manifest():
    m2 = R.pranet.PraNet_V2(num_class=1)
    ref = {k: list(v.shape) for k, v in m2.state_dict().items()}
    ours = {k: list(v) for k, v in W.manifest_pranet_v2(1).items()}
    assert list(ref.items()) == list(ours.items()), "manifest mismatch (V2)"
    m1 = R.v1.PraNet()
    ref1 = {k: list(v.shape) for k, v in m1.state_dict().items()}
    ours1 = {k: list(v) for k, v in W.manifest_pranet_v1().items()}
    assert list(ref1.items()) == list(ours1.items()), "manifest mismatch (V1)"
    m9 = R.pranet.PraNet_V2(num_class=9)
    assert {k: list(v.shape) for k, v in m9.state_dict().items()} == {k: list(v) for k, v in W.manifest_pranet_v2(9).items()}
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump({"pranet_v2_k1": ref, "pranet_v1": ref1,
                   "n_params_v2": sum(p.numel() for p in m2.parameters()),
                   "n_params_v1": sum(p.numel() for p in m1.parameters())}, f)
    print("manifest ok:", len(ref), len(ref1))


# ---------------------------------------------------------------------------- structure_loss
def gen_structure_loss():
    g = torch.Generator().manual_seed(7)
    out = {}
    for tag, maskfn in (("rand", None), ("zeros", 0.0), ("ones", 1.0)):
        pred = (torch.randn(2, 1, 64, 64, generator=g) * 3).requires_grad_(True)
        pred_bg = (torch.randn(2, 1, 64, 64, generator=g) * 3).requires_grad_(True)
        if maskfn is None:
            _, mask = W.synthetic_batch(2, 64, seed=11)
        else:
            mask = torch.full((2, 1, 64, 64), maskfn)
        loss = R.train.structure_loss(pred, pred_bg, mask, 1 - mask)
        loss.backward()
        out.update({f"{tag}_pred": npy(pred), f"{tag}_pred_bg": npy(pred_bg), f"{tag}_mask": npy(mask),
                    f"{tag}_loss": npy(loss), f"{tag}_gpred": npy(pred.grad), f"{tag}_gpred_bg": npy(pred_bg.grad)})
    save("structure_loss.npz", **out)


# ---------------------------------------------------------------------------- DSRA fusion K=9 (formula of pranet.py:365-368)
def gen_dsra():
    g = torch.Generator().manual_seed(9)
    fg = torch.randn(2, 9, 12, 12, generator=g).requires_grad_(True)
    cf = torch.randn(2, 9, 12, 12, generator=g).requires_grad_(True)
    cb = torch.randn(2, 9, 12, 12, generator=g).requires_grad_(True)
    go = torch.randn(2, 9, 12, 12, generator=g)
    out = {"fg": npy(fg), "crop_fg": npy(cf), "crop_bg": npy(cb), "gout": npy(go)}
    for tag, sm in (("sm", True), ("nosm", False)):
        for t in (fg, cf, cb):
            t.grad = None
        if sm:
            y = fg + fg.mul(torch.nn.functional.softmax(cf - cb, dim=1))
        else:
            y = fg + fg.mul(cf - cb)
        y.backward(go)
        out.update({f"{tag}_y": npy(y), f"{tag}_gfg": npy(fg.grad), f"{tag}_gcf": npy(cf.grad), f"{tag}_gcb": npy(cb.grad)})
    save("dsra_k9.npz", **out)


# ---------------------------------------------------------------------------- sub-blocks
def gen_blocks():
    torch.manual_seed(3)
    out = {}
    # Bottle2neck normal + stage (Res2Net_v1b.py:15-91) with small planes
    B = R.res2.Bottle2neck
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, 12, 12, generator=g)
    blk = B(64, 16, baseWidth=26, scale=4)             # width 6, normal, no downsample
    for p in blk.parameters():
        p.data = torch.randn(p.shape, generator=g) * (0.3 if p.ndim > 1 else 0.5) + (1.0 if p.ndim == 1 else 0.0)
    blk.train(); y = blk(x)
    out["b2n_x"] = npy(x); out["b2n_y"] = npy(y)
    for k, v in blk.state_dict().items():
        out["b2n_sd." + k] = npy(v)                     # buffers AFTER the train-mode forward
    import torch.nn as nn
    down = nn.Sequential(nn.AvgPool2d(2, 2, ceil_mode=True, count_include_pad=False), nn.Conv2d(64, 128, 1, bias=False), nn.BatchNorm2d(128))
    blk2 = B(64, 32, stride=2, downsample=down, stype="stage")
    x2 = torch.randn(2, 64, 13, 13, generator=g)        # odd size exercises ceil_mode
    for p in blk2.parameters():
        p.data = torch.randn(p.shape, generator=g) * (0.3 if p.ndim > 1 else 0.5) + (1.0 if p.ndim == 1 else 0.0)
    blk2.train(); y2 = blk2(x2)
    out["b2s_x"] = npy(x2); out["b2s_y"] = npy(y2)
    for k, v in blk2.state_dict().items():
        out["b2s_sd." + k] = npy(v)
    # RFB_modified (pranet.py:46-83) and aggregation (:86-125)
    r = R.pranet.RFB_modified(48, 32)
    for p in r.parameters():
        p.data = torch.randn(p.shape, generator=g) * (0.2 if p.ndim > 1 else 0.5) + (1.0 if p.ndim == 1 else 0.0)
    xr = torch.randn(2, 48, 11, 11, generator=g)
    r.train(); yr = r(xr)
    out["rfb_x"] = npy(xr); out["rfb_y"] = npy(yr)
    for k, v in r.state_dict().items():
        out["rfb_sd." + k] = npy(v)
    a = R.pranet.aggregation(32, 1)
    for p in a.parameters():
        p.data = torch.randn(p.shape, generator=g) * (0.2 if p.ndim > 1 else 0.5) + (1.0 if p.ndim == 1 else 0.0)
    a1, a2, a3 = (torch.randn(2, 32, s, s, generator=g) for s in (3, 6, 12))
    a.train(); fgo, bgo = a(a1, a2, a3)
    out.update(agg_x1=npy(a1), agg_x2=npy(a2), agg_x3=npy(a3), agg_fg=npy(fgo), agg_bg=npy(bgo))
    for k, v in a.state_dict().items():
        out["agg_sd." + k] = npy(v)
    save("blocks.npz", **out)


# ---------------------------------------------------------------------------- whole model
def gen_model(size, n, tag, full_maps):
    man = W.manifest_pranet_v2(1)
    sd0 = W.make_state_dict(man, seed=0)
    model = R.pranet.PraNet_V2(num_class=1)
    model.load_state_dict(sd0, strict=True)
    model.train()
    x, mask = W.synthetic_batch(n, size, seed=1234)
    opt = torch.optim.Adam(model.parameters(), 1e-4)                       # MyTrain_med.py:149
    out = {"size": np.array(size), "n": np.array(n)}
    names = dict(model.named_parameters())
    bufs = dict(model.named_buffers())
    for step in (1, 2):
        opt.zero_grad()
        outs = model(x)
        bg = 1 - mask
        losses = [R.train.structure_loss(outs[i], outs[i + 4], mask, bg) for i in range(4)]   # :78-81 (order l2,l3,l4,l5)
        loss = losses[3] + losses[2] + losses[1] + losses[0]
        loss.backward()
        if step == 1:
            for k in PROBE_PARAMS:
                out["graw." + k] = head(names[k].grad)                        # before clamp
                out["grawnorm." + k] = npy(names[k].grad.norm())
            out["nograd"] = np.array([k for k, p in names.items() if p.grad is None])
        R.utils.clip_gradient(opt, 0.5)                                       # :85
        opt.step()                                                            # :86
        s = f"s{step}."
        out[s + "losses"] = np.array([float(l) for l in losses])
        out[s + "loss"] = npy(loss)
        for i, o in enumerate(outs):
            o = o.detach()
            out[s + f"out{i}.stats"] = np.array([float(o.mean()), float(o.abs().mean()), float(o.abs().max())])
            out[s + f"out{i}"] = npy(o if full_maps else o[:, :, ::4, ::4])
        for k in PROBE_PARAMS:
            out[s + "param." + k] = head(names[k])
        for k in PROBE_BUFFERS:
            out[s + "buf." + k] = head(bufs[k])
    # The same reference classes run in float64: the well-conditioned "truth" of step 1.  Train-mode BN over small
    # batches makes this network ill-conditioned in fp32 (the reference's own fp32 result differs from its fp64 result
    # by ~1e-3 on the logits), so parity tests bound |ours - ref64| by a multiple of |ref32 - ref64| measured here.
    m64 = R.pranet.PraNet_V2(num_class=1)
    m64.load_state_dict(sd0, strict=True)
    m64 = m64.double().train()
    o64 = m64(x.double())
    l64 = [R.train.structure_loss(o64[i], o64[i + 4], mask.double(), 1 - mask.double()) for i in range(4)]
    (l64[3] + l64[2] + l64[1] + l64[0]).backward()
    out["f64.losses"] = np.array([float(l) for l in l64])
    n64 = dict(m64.named_parameters())
    for i, o in enumerate(o64):
        o = o.detach()
        out[f"f64.out{i}"] = npy(o if full_maps else o[:, :, ::4, ::4]).astype(np.float64)
    for k in PROBE_PARAMS:
        out["f64.graw." + k] = head(n64[k].grad)
        out["f64.grawnorm." + k] = npy(n64[k].grad.norm())
    del m64, o64
    # eval-mode forward with the populated running stats + MyTest_med.py:104-111 tail
    model.eval()
    with torch.no_grad():
        xe = x[:1]
        outs = model(xe)
        res = outs[0] + outs[1] + outs[2] + outs[3]
        gt_shape = (size + 8, size - 6)                                        # GT size differs from test size
        res = torch.nn.functional.interpolate(res, size=gt_shape, mode="bilinear", align_corners=False)
        res = res.sigmoid().data.cpu().numpy().squeeze()
        res = (res - res.min()) / (res.max() - res.min() + 1e-8)
        u8 = (res * 255).astype(np.uint8)
        gt = torch.nn.functional.interpolate(mask[:1], size=gt_shape, mode="nearest")[0, 0].numpy()
        sys.path.insert(0, "/root/reference/binary_seg")
        from utils.eval_functions import Fmeasure_calu                          # noqa
        thr = np.linspace(1, 0, 256)
        dic = np.mean([Fmeasure_calu(u8.astype(np.float64) / 255, (gt > 0.5).astype(np.float64), t)[3] for t in thr])
        for i, o in enumerate(outs):
            out[f"eval.out{i}"] = npy(o if full_maps else o[:, :, ::4, ::4])
        out["eval.u8"] = u8; out["eval.gt"] = gt.astype(np.uint8); out["eval.meanDic"] = np.array(dic)
    save(f"pranet_v2_{tag}.npz", **out)


V1_PROBES = ("ra4_conv1.conv.weight", "ra3_conv1.conv.weight", "ra2_conv4.conv.weight", "agg1.conv5.weight", "rfb3_1.branch2.2.conv.weight", "ra4_conv5.bn.weight")


def _v1_case(make_model, sd0, x, out, probes):
    """outputs + gradient probes of a V1 model in fp32 and, from the same weights, in float64 (f64.*: the well-conditioned truth)."""
    for tag, dt in (("", torch.float32), ("f64.", torch.float64)):
        model = make_model()
        model.load_state_dict(sd0, strict=True)
        model = model.to(dt).train()
        outs = model(x.to(dt))
        for i, o in enumerate(outs):
            out[f"{tag}out{i}"] = npy(o)
        sum(o.square().mean() for o in outs).backward()
        names = dict(model.named_parameters())
        for k in probes:
            out[tag + "graw." + k] = head(names[k].grad)
            out[tag + "grawnorm." + k] = npy(names[k].grad.norm())


def gen_v1(size=96, n=2):
    man = W.manifest_pranet_v1()
    sd0 = W.make_state_dict(man, seed=1)
    x, _ = W.synthetic_batch(n, size, seed=77)
    out = {}
    _v1_case(lambda: R.v1.PraNet(), sd0, x, out, ("resnet.layer4.2.conv3.weight",) + V1_PROBES)
    save("pranet_v1_96.npz", **out)


def _ref_pvt_v1_model():
    """Reference PVT_PraNet (PraNet_Res2Net.py:188-224) without its checkpoint file: torch.load patched to {} around construction; DropPath off."""
    orig = torch.load
    torch.load = lambda *a, **k: {}
    try:
        m = R.v1.PVT_PraNet()
    finally:
        torch.load = orig
    m.backbone.reset_drop_path(0.0)
    return m


def gen_pvt_v1(size=96, n=2):
    man = W.manifest_pvt_pranet_v1()
    model = _ref_pvt_v1_model()
    ref = {k: list(v.shape) for k, v in model.state_dict().items()}
    assert list(ref.items()) == [(k, list(v)) for k, v in man.items()], "manifest mismatch (PVT_PraNet)"
    with open(os.path.join(HERE, "manifest_pvt_v1.json"), "w") as f:
        json.dump({"pvt_pranet": ref, "n_params": sum(p.numel() for p in model.parameters())}, f)
    sd0 = W.make_state_dict(man, seed=7)
    x, _ = W.synthetic_batch(n, size, seed=78)
    out = {}
    _v1_case(_ref_pvt_v1_model, sd0, x, out, ("backbone.block4.0.attn.q.weight", "backbone.patch_embed1.proj.weight") + V1_PROBES)
    save("pvt_pranet_v1_96.npz", **out)


PVT_PROBES = [
    "backbone.patch_embed1.proj.weight", "backbone.patch_embed1.proj.bias", "backbone.patch_embed1.norm.weight",
    "backbone.block1.0.norm1.weight", "backbone.block1.0.attn.q.weight", "backbone.block1.0.attn.kv.bias", "backbone.block1.0.attn.sr.weight",
    "backbone.block1.0.attn.norm.bias", "backbone.block1.2.mlp.fc1.weight", "backbone.block1.2.mlp.dwconv.dwconv.weight",
    "backbone.block2.1.attn.proj.weight", "backbone.block2.1.mlp.dwconv.dwconv.bias", "backbone.block2.3.mlp.fc2.bias", "backbone.patch_embed3.proj.weight",
    "backbone.block3.2.attn.kv.weight", "backbone.block3.5.norm2.bias", "backbone.block4.0.attn.q.bias", "backbone.block4.0.mlp.fc1.weight",
    "backbone.block4.2.mlp.fc2.weight", "backbone.norm2.weight", "backbone.norm4.bias",
    "rfb2_1.branch0.0.conv.weight", "rfb4_1.conv_cat.conv.weight", "agg1.conv5_fg.weight", "ra4_conv1.conv.weight", "ra2_conv4_fg.conv.weight",
]


def _ref_pvt_model(dtype=torch.float32):
    """Reference PVT_PraNet_V2 without its checkpoint file: torch.load patched to {} around construction; DropPath off."""
    orig = torch.load
    torch.load = lambda *a, **k: {}
    try:
        m = R.pranet.PVT_PraNet_V2(num_class=1)
    finally:
        torch.load = orig
    m.backbone.reset_drop_path(0.0)          # stochastic depth is the one RNG-dependent op of the path; parity is pinned with it off
    return m.to(dtype)


def gen_pvt(size=96, n=2):
    man = W.manifest_pvt_pranet_v2(1)
    model = _ref_pvt_model()
    ref = {k: list(v.shape) for k, v in model.state_dict().items()}
    assert list(ref.items()) == [(k, list(v)) for k, v in man.items()], "manifest mismatch (PVT)"
    with open(os.path.join(HERE, "manifest_pvt.json"), "w") as f:
        json.dump({"pvt_pranet_v2_k1": ref, "n_params": sum(p.numel() for p in model.parameters())}, f)
    sd0 = W.make_state_dict(man, seed=3)
    model.load_state_dict(sd0, strict=True)
    model.train()
    x, mask = W.synthetic_batch(n, size, seed=4321)
    out = {"size": np.array(size), "n": np.array(n)}
    # backbone features alone (4 NCHW maps)
    with torch.no_grad():
        feats = model.backbone(x)
    for i, f_ in enumerate(feats):
        out[f"feat{i}"] = npy(f_)
    outs = model(x)
    losses = [R.train.structure_loss(outs[i], outs[i + 4], mask, 1 - mask) for i in range(4)]
    loss = losses[3] + losses[2] + losses[1] + losses[0]
    loss.backward()
    names = dict(model.named_parameters())
    out["losses"] = np.array([float(l) for l in losses]); out["loss"] = npy(loss)
    for i, o in enumerate(outs):
        out[f"out{i}"] = npy(o)
    for k in PVT_PROBES:
        out["graw." + k] = head(names[k].grad)
        out["grawnorm." + k] = npy(names[k].grad.norm())
    m64 = _ref_pvt_model()
    m64.load_state_dict(sd0, strict=True)
    m64 = m64.double().train()
    o64 = m64(x.double())
    l64 = [R.train.structure_loss(o64[i], o64[i + 4], mask.double(), 1 - mask.double()) for i in range(4)]
    (l64[3] + l64[2] + l64[1] + l64[0]).backward()
    n64 = dict(m64.named_parameters())
    out["f64.losses"] = np.array([float(l) for l in l64])
    for i, f_ in enumerate(m64.backbone(x.double())):
        out[f"f64.feat{i}"] = npy(f_).astype(np.float64)
    for i, o in enumerate(o64):
        out[f"f64.out{i}"] = npy(o).astype(np.float64)
    for k in PVT_PROBES:
        out["f64.graw." + k] = head(n64[k].grad)
        out["f64.grawnorm." + k] = npy(n64[k].grad.norm())
    save("pvt_pranet_v2_96.npz", **out)


def gen_pvt_gray(size=64, n=2):
    """PVT_PraNet_V2 on 1-channel input: the conv(1->3)+BN+ReLU stem of pranet.py:190-191 in front of the backbone (fp32 + float64)."""
    man = W.manifest_pvt_pranet_v2(1)
    sd0 = W.make_state_dict(man, seed=11)
    x, mask = W.synthetic_batch(n, size, seed=777)
    x = x[:, :1].contiguous()
    out = {"size": np.array(size), "n": np.array(n)}
    probes = ["conv.0.weight", "conv.1.weight", "conv.1.bias", "backbone.patch_embed1.proj.weight", "ra2_conv4_fg.conv.weight"]
    for tag, dt in (("", torch.float32), ("f64.", torch.float64)):
        model = _ref_pvt_model()
        model.load_state_dict(sd0, strict=True)
        model = model.to(dt).train()
        outs = model(x.to(dt))
        m = mask.to(dt)
        losses = [R.train.structure_loss(outs[i], outs[i + 4], m, 1 - m) for i in range(4)]
        (losses[3] + losses[2] + losses[1] + losses[0]).backward()
        names = dict(model.named_parameters())
        out[tag + "losses"] = np.array([float(l) for l in losses])
        for i, o in enumerate(outs):
            out[tag + f"out{i}"] = npy(o).astype(np.float64 if tag else np.float32)
        for k in probes:
            out[tag + "graw." + k] = head(names[k].grad)
            out[tag + "grawnorm." + k] = npy(names[k].grad.norm())
        out[tag + "rm.conv.1"] = npy(model.conv[1].running_mean); out[tag + "rv.conv.1"] = npy(model.conv[1].running_var)
    save("pvt_pranet_v2_gray_64.npz", **out)


def gen_eval_metrics():
    """Threshold-sweep metrics of eval.py:22-50 (Fmeasure_calu, eval_functions.py:131-166) + MAE on small synthetic maps."""
    import importlib
    cwd = os.getcwd(); os.chdir("/root/reference/binary_seg")
    try:
        ef = importlib.import_module("utils.eval_functions")
    finally:
        os.chdir(cwd)
    rng = np.random.default_rng(5)
    H, Wd = 64, 80
    yy, xx = np.mgrid[0:H, 0:Wd]
    blob = (((yy - 30) / 14.0) ** 2 + ((xx - 42) / 22.0) ** 2 < 1).astype(np.float64)
    smooth = np.clip(255 * (0.75 * blob + 0.25 * rng.random((H, Wd))) + 12 * rng.standard_normal((H, Wd)), 0, 255).astype(np.uint8)
    cases = {"blob": (smooth, blob), "zero_pred": (np.zeros((H, Wd), np.uint8), blob), "zero_gt": (smooth, np.zeros((H, Wd))),
             "exact": ((255 * blob).astype(np.uint8), blob), "full": (np.full((H, Wd), 255, np.uint8), np.ones((H, Wd)))}
    out = {}
    thr = np.linspace(1, 0, 256)
    for tag, (pred, gt) in cases.items():
        gt_mask = (gt > 0.5).astype(np.float64)
        pm = pred.astype(np.float64) / 255
        cols = np.array([ef.Fmeasure_calu(pm, gt_mask, t) for t in thr], dtype=np.float64)    # Pr, Rec, Spe, Dice, F, IoU
        out[tag + "_pred"] = pred; out[tag + "_gt"] = gt.astype(np.float32)
        out[tag + "_means"] = cols.mean(axis=0)
        out[tag + "_curves"] = cols
        out[tag + "_mae"] = np.float64(np.mean(np.abs(gt_mask - pm)))
    save("eval_metrics.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["manifest", "loss", "dsra", "blocks", "m96", "m352", "v1", "pvtv1", "evalm", "pvt", "pvtgray"]
    if "evalm" in which: gen_eval_metrics()
    if "pvt" in which: gen_pvt()
    if "manifest" in which: gen_manifest()
    if "loss" in which: gen_structure_loss()
    if "dsra" in which: gen_dsra()
    if "blocks" in which: gen_blocks()
    if "m96" in which: gen_model(96, 2, "96", True)
    if "m352" in which: gen_model(352, 2, "352", False)
    if "v1" in which: gen_v1()
    if "pvtv1" in which: gen_pvt_v1()
    if "pvtgray" in which: gen_pvt_gray()
