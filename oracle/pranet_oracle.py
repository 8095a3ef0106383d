"""TEST INFRASTRUCTURE — CPU oracle for the PraNet-V2 hot path.  NOT the product.

A functional, plain-PyTorch-CPU fp32 restatement of the algorithm the reference
runs between `optimizer.zero_grad()` and `optimizer.step()`
(/root/reference/binary_seg/MyTrain_med.py:59-86).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this file; the product path
(pranet-v2_amd/) never does and raises if its HIP library is missing.

Pinning: tests/test_oracle_golden.py checks every function here against vectors
produced by the *imported reference itself* in the build container
(tests/golden/make_golden.py -> tests/golden/*.npz).  The reference ships no tests
or golden vectors for this path, so those generated vectors are the pin.

All functions take a flat state_dict `P` (key -> tensor, reference names) and a key
prefix, so no nn.Module of ours is involved: this file shares no code with the product.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1


class Ctx:
    """Mode + optional side effects (running-stat updates) of one forward."""

    def __init__(self, training):
        self.training = training


def bn(P, name, x, ctx):
    # nn.BatchNorm2d defaults (pranet.py:37, Res2Net_v1b.py:33): eps 1e-5, momentum 0.1,
    # biased var for normalisation, unbiased for running_var; F.batch_norm updates P[...] in place.
    y = F.batch_norm(x, P[name + ".running_mean"], P[name + ".running_var"], P[name + ".weight"], P[name + ".bias"],
                     ctx.training, BN_MOMENTUM, BN_EPS)
    if ctx.training:
        P[name + ".num_batches_tracked"] += 1
    return y


def basic_conv(P, name, x, ctx, stride=1, padding=0, dilation=1):
    # BasicConv2d.forward = bn(conv(x)), bias-free, NO ReLU  (pranet.py:40-43)
    x = F.conv2d(x, P[name + ".conv.weight"], None, stride, padding, dilation)
    return bn(P, name + ".bn", x, ctx)


# --------------------------------------------------------------------------------------
# Res2Net-50 v1b 26w x 4s  (Res2Net_v1b.py)
# --------------------------------------------------------------------------------------
def bottle2neck(P, p, x, ctx, stride, stage, has_down, scale=4):
    # Res2Net_v1b.py:58-91
    out = F.relu(bn(P, p + "bn1", F.conv2d(x, P[p + "conv1.weight"]), ctx))
    width = out.shape[1] // scale
    spx = torch.split(out, width, 1)
    outs = []
    sp = None
    for i in range(scale - 1):
        sp = spx[i] if (i == 0 or stage) else sp + spx[i]
        sp = F.conv2d(sp, P[p + f"convs.{i}.weight"], None, stride, 1)
        sp = F.relu(bn(P, p + f"bns.{i}", sp, ctx))
        outs.append(sp)
    if stage:
        outs.append(F.avg_pool2d(spx[scale - 1], 3, stride, 1))        # :40, :80
    else:
        outs.append(spx[scale - 1])
    out = torch.cat(outs, 1)
    out = bn(P, p + "bn3", F.conv2d(out, P[p + "conv3.weight"]), ctx)
    if has_down:
        # AvgPool2d(k=stride, s=stride, ceil_mode=True, count_include_pad=False) -> 1x1 conv -> BN (:127-136)
        r = F.avg_pool2d(x, stride, stride, 0, True, False)
        r = bn(P, p + "downsample.2", F.conv2d(r, P[p + "downsample.1.weight"]), ctx)
    else:
        r = x
    return F.relu(out + r)


def res2net_features(P, p, x, ctx, layers=(3, 4, 6, 3)):
    # stem + maxpool + layer1..4, as PraNet_V2.forward drives it (pranet.py:331-341)
    x = F.relu(bn(P, p + "conv1.1", F.conv2d(x, P[p + "conv1.0.weight"], None, 2, 1), ctx))
    x = F.relu(bn(P, p + "conv1.4", F.conv2d(x, P[p + "conv1.3.weight"], None, 1, 1), ctx))
    x = F.conv2d(x, P[p + "conv1.6.weight"], None, 1, 1)
    x = F.relu(bn(P, p + "bn1", x, ctx))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for li, nblk in enumerate(layers):
        stride = 1 if li == 0 else 2
        for b in range(nblk):
            q = f"{p}layer{li + 1}.{b}."
            x = bottle2neck(P, q, x, ctx, stride if b == 0 else 1, b == 0, b == 0)
        feats.append(x)
    return feats  # x1..x4


# --------------------------------------------------------------------------------------
# heads  (pranet.py)
# --------------------------------------------------------------------------------------
def rfb(P, p, x, ctx):
    # RFB_modified.forward pranet.py:75-83
    x0 = basic_conv(P, p + "branch0.0", x, ctx)
    br = [x0]
    for bi, k in ((1, 3), (2, 5), (3, 7)):
        q = f"{p}branch{bi}."
        y = basic_conv(P, q + "0", x, ctx)
        y = basic_conv(P, q + "1", y, ctx, padding=(0, k // 2))
        y = basic_conv(P, q + "2", y, ctx, padding=(k // 2, 0))
        y = basic_conv(P, q + "3", y, ctx, padding=k, dilation=k)
        br.append(y)
    x_cat = basic_conv(P, p + "conv_cat", torch.cat(br, 1), ctx, padding=1)
    return F.relu(x_cat + basic_conv(P, p + "conv_res", x, ctx))


def up2_ac(x):
    # nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)  (pranet.py:93)
    return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)


def aggregation(P, p, x1, x2, x3, ctx, v1=False):
    # pranet.py:109-125 (V2) / PraNet_Res2Net.py:83-98 (V1)
    c = lambda n, t: basic_conv(P, p + n, t, ctx, padding=1)
    x2_1 = c("conv_upsample1", up2_ac(x1)) * x2
    x3_1 = c("conv_upsample2", up2_ac(up2_ac(x1))) * c("conv_upsample3", up2_ac(x2)) * x3
    x2_2 = c("conv_concat2", torch.cat((x2_1, c("conv_upsample4", up2_ac(x1))), 1))
    x3_2 = c("conv_concat3", torch.cat((x3_1, c("conv_upsample5", up2_ac(x2_2))), 1))
    x = c("conv4", x3_2)
    if v1:
        return F.conv2d(x, P[p + "conv5.weight"], P[p + "conv5.bias"])
    return (F.conv2d(x, P[p + "conv5_fg.weight"], P[p + "conv5_fg.bias"]),
            F.conv2d(x, P[p + "conv5_bg.weight"], P[p + "conv5_bg.bias"]))


def interp(x, scale):
    # F.interpolate(x, scale_factor=s, mode='bilinear') -> align_corners=False, given scale used
    return F.interpolate(x, scale_factor=scale, mode="bilinear")


def dsra_fuse(fg, crop_fg, crop_bg, use_softmax=True):
    # pranet.py:365-368 — bg is NOT modified
    if use_softmax:
        return fg + fg * F.softmax(crop_fg - crop_bg, dim=1)
    return fg + fg * (crop_fg - crop_bg)


# ------------------------------------------------------------------------------------------------ PVTv2 encoder (lib/pvtv2.py)
def _ln(P, name, x, eps):
    return F.layer_norm(x, (x.shape[-1],), P[name + ".weight"], P[name + ".bias"], eps)


def _lin(P, name, x):
    return F.linear(x, P[name + ".weight"], P[name + ".bias"])


def pvt_attention(P, p, x, H, W, heads, sr):
    """Attention.forward (pvtv2.py:90-111): spatial-reduction attention; dropout rates are 0 in pvt_v2_b2 (:402-406)."""
    B, N, C = x.shape
    hd = C // heads
    q = _lin(P, p + "q", x).reshape(B, N, heads, hd).permute(0, 2, 1, 3)
    if sr > 1:
        x_ = x.permute(0, 2, 1).reshape(B, C, H, W)
        x_ = F.conv2d(x_, P[p + "sr.weight"], P[p + "sr.bias"], stride=sr).reshape(B, C, -1).permute(0, 2, 1)
        x_ = _ln(P, p + "norm", x_, 1e-5)                      # nn.LayerNorm(dim) default eps (:71)
    else:
        x_ = x
    kv = _lin(P, p + "kv", x_).reshape(B, -1, 2, heads, hd).permute(2, 0, 3, 1, 4)
    k, v = kv[0], kv[1]
    attn = ((q @ k.transpose(-2, -1)) * hd ** -0.5).softmax(dim=-1)
    return _lin(P, p + "proj", (attn @ v).transpose(1, 2).reshape(B, N, C))


def pvt_mlp(P, p, x, H, W):
    """Mlp.forward (pvtv2.py:42-49) with DWConv (:368-374) and exact GELU."""
    B, N, _ = x.shape
    x = _lin(P, p + "fc1", x)
    Ch = x.shape[-1]
    x = F.conv2d(x.transpose(1, 2).reshape(B, Ch, H, W), P[p + "dwconv.dwconv.weight"], P[p + "dwconv.dwconv.bias"], padding=1, groups=Ch)
    x = F.gelu(x.flatten(2).transpose(1, 2))
    return _lin(P, p + "fc2", x)


def pvt_block(P, p, x, H, W, heads, sr):
    """Block.forward (pvtv2.py:147-151) with DropPath = identity (drop_path forced to 0 for parity)."""
    x = x + pvt_attention(P, p + "attn.", _ln(P, p + "norm1", x, 1e-6), H, W, heads, sr)
    return x + pvt_mlp(P, p + "mlp.", _ln(P, p + "norm2", x, 1e-6), H, W)


def pvt_features(P, p, x, cfg=None):
    """PyramidVisionTransformerImpr.forward_features (pvtv2.py:307-341) for pvt_v2_b2 (:399-406): four NCHW feature maps."""
    from oracle.weights import PVT_B2
    cfg = cfg or PVT_B2
    B = x.shape[0]
    outs = []
    for i in range(4):
        k, s = (7, 4) if i == 0 else (3, 2)
        x = F.conv2d(x, P[f"{p}patch_embed{i + 1}.proj.weight"], P[f"{p}patch_embed{i + 1}.proj.bias"], stride=s, padding=k // 2)
        H, W = x.shape[2:]
        x = _ln(P, f"{p}patch_embed{i + 1}.norm", x.flatten(2).transpose(1, 2), 1e-5)        # OverlapPatchEmbed.norm default eps (:169)
        for j in range(cfg["depths"][i]):
            x = pvt_block(P, f"{p}block{i + 1}.{j}.", x, H, W, cfg["num_heads"][i], cfg["sr_ratios"][i])
        x = _ln(P, f"{p}norm{i + 1}", x, 1e-6)
        x = x.reshape(B, H, W, -1).permute(0, 3, 1, 2).contiguous()
        outs.append(x)
    return outs


def pvt_pranet_v2_forward(P, x, training, use_softmax=True, sem_downsample=1):
    """PVT_PraNet_V2.forward (pranet.py:205-263): PVTv2-B2 features, then exactly the PraNet_V2 heads."""
    def features(P_, x_, ctx):
        if x_.shape[1] == 1:            # pranet.py:190-191 + :139-143: conv(1->3, k1, bias) + BN + ReLU for 1-channel slices
            x_ = F.relu(bn(P_, "conv.1", F.conv2d(x_, P_["conv.0.weight"], P_["conv.0.bias"]), ctx))
        return pvt_features(P_, "backbone.", x_)
    return pranet_v2_forward(P, x, training, use_softmax, sem_downsample, features=features)


def pranet_v2_forward(P, x, training, use_softmax=True, sem_downsample=1, features=None):
    """PraNet_V2.forward (pranet.py:329-417).  Mutates BN running stats in P when training."""
    ctx = Ctx(training)
    x1, x2, x3, x4 = features(P, x, ctx) if features is not None else res2net_features(P, "backbone.", x, ctx)
    x2_rfb = rfb(P, "rfb2_1.", x2, ctx)
    x3_rfb = rfb(P, "rfb3_1.", x3, ctx)
    x4_rfb = rfb(P, "rfb4_1.", x4, ctx)
    ra5_fg, ra5_bg = aggregation(P, "agg1.", x4_rfb, x3_rfb, x2_rfb, ctx)
    sd = sem_downsample
    l5_fg, l5_bg = interp(ra5_fg, 8 / sd), interp(ra5_bg, 8 / sd)
    # ---- DSRA3
    c4_fg, c4_bg = interp(ra5_fg, 0.25), interp(ra5_bg, 0.25)
    t = basic_conv(P, "ra4_conv1", x4, ctx)
    for i in (2, 3, 4):
        t = F.relu(basic_conv(P, f"ra4_conv{i}", t, ctx, padding=2))
    ra4_fg = basic_conv(P, "ra4_conv5_fg", t, ctx)
    ra4_bg = basic_conv(P, "ra4_conv5_bg", t, ctx)
    ra4_fg = dsra_fuse(ra4_fg, c4_fg, c4_bg, use_softmax)
    l4_fg, l4_bg = interp(ra4_fg, 32 / sd), interp(ra4_bg, 32 / sd)
    # ---- DSRA2 / DSRA1
    prev_fg, prev_bg = ra4_fg, ra4_bg
    lat = {}
    for s, xs, up in ((3, x3, 16), (2, x2, 8)):
        c_fg, c_bg = interp(prev_fg, 2), interp(prev_bg, 2)
        t = basic_conv(P, f"ra{s}_conv1", xs, ctx)
        t = F.relu(basic_conv(P, f"ra{s}_conv2", t, ctx, padding=1))
        t = F.relu(basic_conv(P, f"ra{s}_conv3", t, ctx, padding=1))
        r_fg = basic_conv(P, f"ra{s}_conv4_fg", t, ctx, padding=1)
        r_bg = basic_conv(P, f"ra{s}_conv4_bg", t, ctx, padding=1)
        r_fg = dsra_fuse(r_fg, c_fg, c_bg, use_softmax)
        lat[s] = (interp(r_fg, up / sd), interp(r_bg, up / sd))
        prev_fg, prev_bg = r_fg, r_bg
    return lat[2][0], lat[3][0], l4_fg, l5_fg, lat[2][1], lat[3][1], l4_bg, l5_bg


def pvt_pranet_v1_forward(P, x, training):
    """PVT_PraNet.forward (PraNet_Res2Net.py:226-273): PVTv2-B2 features, then exactly the PraNet reverse-attention heads."""
    return pranet_v1_forward(P, x, training, features=lambda P_, x_, ctx: pvt_features(P_, "backbone.", x_))


def pranet_v1_forward(P, x, training, features=None):
    """PraNet.forward (PraNet_Res2Net.py:130-186): reverse attention = (1 - sigmoid(crop)) gate."""
    ctx = Ctx(training)
    x1, x2, x3, x4 = features(P, x, ctx) if features is not None else res2net_features(P, "resnet.", x, ctx)
    x2_rfb = rfb(P, "rfb2_1.", x2, ctx)
    x3_rfb = rfb(P, "rfb3_1.", x3, ctx)
    x4_rfb = rfb(P, "rfb4_1.", x4, ctx)
    ra5 = aggregation(P, "agg1.", x4_rfb, x3_rfb, x2_rfb, ctx, v1=True)
    l5 = interp(ra5, 8)
    crop = interp(ra5, 0.25)
    t = (-1 * torch.sigmoid(crop) + 1).expand(-1, x4.shape[1], -1, -1).mul(x4)
    t = basic_conv(P, "ra4_conv1", t, ctx)
    for i in (2, 3, 4):
        t = F.relu(basic_conv(P, f"ra4_conv{i}", t, ctx, padding=2))
    x = basic_conv(P, "ra4_conv5", t, ctx) + crop
    l4 = interp(x, 32)
    lats = {}
    for s, xs, up in ((3, x3, 16), (2, x2, 8)):
        crop = interp(x, 2)
        t = (-1 * torch.sigmoid(crop) + 1).expand(-1, xs.shape[1], -1, -1).mul(xs)
        t = basic_conv(P, f"ra{s}_conv1", t, ctx)
        t = F.relu(basic_conv(P, f"ra{s}_conv2", t, ctx, padding=1))
        t = F.relu(basic_conv(P, f"ra{s}_conv3", t, ctx, padding=1))
        x = basic_conv(P, f"ra{s}_conv4", t, ctx, padding=1) + crop
        lats[s] = interp(x, up)
    return l5, l4, lats[3], lats[2]


# --------------------------------------------------------------------------------------
# loss / optimiser / eval tail
# --------------------------------------------------------------------------------------
def structure_loss(pred, pred_bg, mask_fg, mask_bg):
    # MyTrain_med.py:19-38
    weit = 1 + 5 * torch.abs(F.avg_pool2d(mask_fg, kernel_size=31, stride=1, padding=15) - mask_fg)
    wsum = weit.sum(dim=(2, 3))
    wbce = (weit * F.binary_cross_entropy_with_logits(pred, mask_fg, reduction="none")).sum(dim=(2, 3)) / wsum
    wbce2 = (weit * F.binary_cross_entropy_with_logits(pred_bg, mask_bg, reduction="none")).sum(dim=(2, 3)) / wsum
    p = torch.sigmoid(pred)
    inter = ((p * mask_fg) * weit).sum(dim=(2, 3))
    union = ((p + mask_fg) * weit).sum(dim=(2, 3))
    wiou = 1 - (inter + 1) / (union - inter + 1)
    return (wbce + wiou + 0.8 * wbce2).mean()


def total_loss(outs, gts):
    # MyTrain_med.py:74-82 : four (fg,bg) pairs, same masks
    bg = 1 - gts
    return sum(structure_loss(outs[i], outs[i + 4], gts, bg) for i in range(4))


def params_of(P):
    """Keys that nn.Module.parameters() would yield (everything except BN buffers)."""
    return [k for k in P if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))]


def train_step(P, opt_state, x, gts, lr=1e-4, clip=0.5, betas=(0.9, 0.999), eps=1e-8, forward=pranet_v2_forward):
    """One MyTrain_med.py:59-86 step on CPU: fwd, 4x structure_loss, bwd, clamp(+-clip), Adam.

    Mutates P (params + BN buffers) and opt_state {"step", "m", "v"}; returns (loss, outs, grads).
    """
    keys = params_of(P)
    for k in keys:
        P[k].requires_grad_(True); P[k].grad = None
    outs = forward(P, x, True)
    loss = total_loss(outs, gts)
    loss.backward()
    grads = {}
    opt_state["step"] = opt_state.get("step", 0) + 1
    t = opt_state["step"]
    with torch.no_grad():
        for k in keys:
            g = P[k].grad
            P[k].requires_grad_(False)
            if g is None:
                continue
            g = g.clamp(-clip, clip)                           # utils/utils.py:14-17
            grads[k] = g
            m = opt_state.setdefault("m", {}).setdefault(k, torch.zeros_like(g))
            v = opt_state.setdefault("v", {}).setdefault(k, torch.zeros_like(g))
            m.mul_(betas[0]).add_(g, alpha=1 - betas[0])       # torch.optim.Adam, no wd / amsgrad
            v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
            bc1, bc2 = 1 - betas[0] ** t, 1 - betas[1] ** t
            denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
            P[k].addcdiv_(m, denom, value=-lr / bc1)
            P[k].grad = None
    return loss.detach(), [o.detach() for o in outs], grads


def test_postprocess(outs, gt_shape):
    # MyTest_med.py:104-111 : sum of the 4 fg maps -> bilinear to GT size -> sigmoid -> min-max -> uint8
    res = outs[0] + outs[1] + outs[2] + outs[3]
    res = F.interpolate(res, size=gt_shape, mode="bilinear", align_corners=False)
    res = res.sigmoid().squeeze()
    res = (res - res.min()) / (res.max() - res.min() + 1e-8)
    return (res.numpy() * 255).astype("uint8")


def mean_dice(pred_u8, gt):
    """meanDic of eval.py:22,44-50 + eval_functions.py:131-166 (Fmeasure_calu Dice over 256 thresholds)."""
    import numpy as np
    pred = pred_u8.astype(np.float64) / 255.0
    gt = gt > 0.5
    dice = np.zeros(256)
    thr = np.linspace(1, 0, 256)
    for i, t in enumerate(thr):
        lab = pred >= t if t <= 1 else np.zeros_like(gt)
        num_rec, num_no_rec = lab.sum(), (~lab).sum()
        lab_and = lab & gt
        num_and, num_obj = lab_and.sum(), gt.sum()
        if num_and == 0:
            dice[i] = 0
        else:
            dice[i] = 2.0 * num_and / (num_obj + num_rec)
    return float(dice.mean())


def threshold_metrics(pred_u8, gt):
    """Threshold sweep of eval.py:22-50: for each of the 256 thresholds linspace(1, 0, 256) the (precision, recall, specificity, Dice,
    F-measure, IoU) of Fmeasure_calu (eval_functions.py:131-166), plus MAE (eval.py:33).  Returns (curves[256, 6], mae)."""
    import numpy as np
    pm = pred_u8.astype(np.float64) / 255
    g = (gt > 0.5).astype(np.float64)
    cols = np.zeros((256, 6))
    for i, t in enumerate(np.linspace(1, 0, 256)):
        lab = pm >= min(t, 1)
        num_rec, num_no_rec = int(lab.sum()), int((~lab).sum())
        num_and, num_obj = int((lab & (g == 1)).sum()), g.sum()
        fn, fp = num_obj - num_and, num_rec - num_and
        tn = num_no_rec - fn
        if num_and == 0:
            continue
        pre, rec = num_and / num_rec, num_and / num_obj
        cols[i] = (pre, rec, tn / (tn + fp), 2 * num_and / (num_obj + num_rec), (2.0 * pre * rec) / (pre + rec), num_and / (fn + num_rec))
    return cols, float(np.mean(np.abs(g - pm)))


def clone_sd(sd):
    return OrderedDict((k, v.clone()) for k, v in sd.items())


# ---------------------------------------------------------------------------------------------- the remaining metrics of eval_for_testAllInOne
# (binary_seg/eval.py:18-66: Sm = StructureMeasure, wFm = original_WFb, meanEm = mean over the 256 thresholds of EnhancedMeasure;
#  utils/eval_functions.py:5-129,168-192).  numpy restatement; the only third-party arithmetic the reference uses here is scipy.ndimage's exact Euclidean
#  feature transform (distance_transform_edt(return_indices=True)) and its `convolve(mode='nearest')`: both are restated below (edt_nearest / conv_nearest)
#  and pinned against scipy itself in tests/test_oracle_golden.py, tie-breaking included.
def edt_nearest(fg):
    """For every pixel the nearest pixel with fg != 0 (Euclidean), as scipy.ndimage.distance_transform_edt(1 - fg, return_indices=True) returns it - ties
    included: scipy runs Maurer's Voronoi sweep along axis 0 and then along axis 1 and keeps the EARLIER site while the next one is not strictly closer, i.e.
    (1) in a column the nearest fg row, the smaller row on a tie; (2) over the columns' candidates (r(i, j'), j') the smallest squared distance, the smaller
    column on a tie.  Returns (dist float64 [H, W], ri int [H, W], rj int [H, W]); a map without any fg pixel gives dist = inf... scipy's answer for that case
    is never used by the reference (StructureMeasure / original_WFb are only meaningful with a non-empty gt; eval.py calls them anyway: see full_metrics)."""
    import numpy as np
    fg = np.asarray(fg) != 0
    H, W = fg.shape
    big = 1 << 40
    rows = np.arange(H)
    rcol = np.full((H, W), -1, dtype=np.int64)          # nearest fg row of every (i, j) within column j
    for j in range(W):
        sites = rows[fg[:, j]]
        if sites.size:
            d = np.abs(rows[:, None] - sites[None, :])
            rcol[:, j] = sites[np.argmin(d, axis=1)]     # argmin takes the first (= smaller row) of equal distances
    dist2 = np.full((H, W), big, dtype=np.int64)
    ri = np.zeros((H, W), dtype=np.int64); rj = np.zeros((H, W), dtype=np.int64)
    cols = np.arange(W)
    for i in range(H):
        r = rcol[i]                                      # candidate row per column
        ok = r >= 0
        if not ok.any():
            continue
        dv = np.where(ok, (r - i) ** 2, big)             # [W]
        d = dv[None, :] + (cols[None, :] - cols[:, None]) ** 2          # [j, j']
        k = np.argmin(d, axis=1)                         # first (= smaller column) of equal distances
        dist2[i] = d[cols, k]; ri[i] = r[k]; rj[i] = k
    return np.sqrt(dist2.astype(np.float64)), ri, rj


def conv_nearest(x, k):
    """scipy.ndimage.convolve(x, k, mode='nearest') for an odd square kernel: out[i, j] = sum_{a, b} k[a, b] * x[clamp(i + c - a), clamp(j + c - b)]
    (convolution = correlation with the flipped kernel; borders replicate the edge pixel), accumulated in float64 in scipy's order (footprint row-major)."""
    import numpy as np
    K = k.shape[0]; c = K // 2
    xp = np.pad(x, c, mode="edge")
    H, W = x.shape
    out = np.zeros((H, W), dtype=np.float64)
    kf = k[::-1, ::-1]
    for a in range(K):
        for b in range(K):
            out += kf[a, b] * xp[a:a + H, b:b + W]
    return out


def full_metrics(pred_u8, gt):
    """eval_for_testAllInOne(opt, pred, gt) for opt["metrics"] = every name eval.py:52-60 defines that is not commented out:
    -> dict(meanDic, meanIoU, meanEm, mae, Sm, wFm) + the threshold curve of EnhancedMeasure ("E")."""
    import numpy as np
    eps = np.finfo(np.float64).eps
    pred = pred_u8.astype(np.float64) / 255                       # eval.py:28
    g = (np.asarray(gt) > 0.5).astype(np.float64)                 # :26
    cols, mae = threshold_metrics(pred_u8, gt)
    # ---- EnhancedMeasure per threshold (eval_functions.py:168-192)
    E = np.zeros(256)
    for i, t in enumerate(np.linspace(1, 0, 256)):
        b = (pred >= t).astype(np.float64)
        if g.sum() == 0:
            em = 1 - b
        elif (1 - g).sum() == 0:
            em = b.copy()
        else:
            ap, ag = b - b.mean(), g - g.mean()
            al = 2 * (ag * ap) / (ag ** 2 + ap ** 2 + eps)
            em = ((al + 1) ** 2) / 4
        E[i] = em.sum() / (g.size - 1 + eps)
    # ---- StructureMeasure (eval_functions.py:5-94)
    y = g.mean()
    if y == 0:
        sm = 1 - pred.mean()
    elif y == 1:
        sm = pred.mean()
    else:
        def obj(p, m):
            x = p[m == 1].mean(); s = p[m == 1].std()
            return 2.0 * x / (x ** 2 + 1 + s + eps)
        pf = pred.copy(); pf[g != 1] = 0.0
        pb = 1 - pred; pb[g == 1] = 0.0
        u = g.mean()
        s_obj = u * obj(pf, g) + (1 - u) * obj(pb, 1 - g)
        if g.sum() == 0:
            cx, cy = g.shape[0] // 2, g.shape[1] // 2
        else:
            xs, ys = np.where(g == 1)
            cx, cy = int(xs.mean().round()), int(ys.mean().round())

        def ssim(p, m):
            if p.size == 0:                                       # an empty quadrant: numpy's mean of an empty slice is nan -> the reference returns nan * 0-weight
                return np.nan
            x, yy, n = p.mean(), m.mean(), p.size
            sx = ((p - x) ** 2 / (n - 1 + eps)).sum(); sy = ((m - yy) ** 2 / (n - 1 + eps)).sum()
            sxy = ((p - x) * (m - yy) / (n - 1 + eps)).sum()
            al, be = 4 * x * yy * sxy, (x ** 2 + yy ** 2) * (sx + sy)
            return al / (be + eps) if al != 0 else (1 if be == 0 else 0)
        quads = [(slice(None, cx), slice(None, cy)), (slice(cx, None), slice(None, cy)), (slice(None, cx), slice(cy, None)), (slice(cx, None), slice(cy, None))]
        s_reg = 0.0
        for a, b_ in quads:
            wq = g[a, b_].size / g.size
            s_reg = s_reg + ssim(pred[a, b_], g[a, b_]) * wq
        sm = 0.5 * s_obj + 0.5 * s_reg
        if sm < 0:
            sm = 0
    # ---- original_WFb (eval_functions.py:96-129)
    Eabs = np.abs(pred - g)
    if g.sum() == 0:
        wfm = float("nan")          # scipy's feature transform of an all-background map returns out-of-range indices; the reference's value is undefined there
    else:
        dst, ri, rj = edt_nearest(g)
        xk, yk = np.mgrid[-7 // 2 + 1:7 // 2 + 1, -7 // 2 + 1:7 // 2 + 1]
        K = np.exp(-((xk ** 2 + yk ** 2) / (2.0 * 5 ** 2))); K = K / K.sum()
        Et = Eabs.copy()
        Et[g != 1] = Eabs[ri[g != 1], rj[g != 1]]
        EA = conv_nearest(Et, K)
        mn = Eabs.copy()
        sel = (g == 1) & (EA < Eabs)
        mn[sel] = EA[sel]
        B = np.ones_like(g)
        B[g != 1] = 2.0 - 1 * np.exp(np.log(1 - 0.5) / 5 * dst[g != 1])
        Ew = mn * B
        TPw = g.sum() - Ew[g == 1].sum(); FPw = Ew[g != 1].sum()
        R = 1 - Ew[g == 1].mean(); Pq = TPw / (TPw + FPw + eps)
        wfm = 2 * R * Pq / (R + Pq + eps)
    m = cols.mean(axis=0)
    return {"meanDic": float(m[3]), "meanIoU": float(m[5]), "meanEm": float(E.mean()), "mae": mae, "Sm": float(sm), "wFm": float(wfm), "E": E}
