"""TEST INFRASTRUCTURE — CPU restatement of the reference's multi-class path (BASELINE config 5), not part of the product.

EMCADNet(dual=True, encoder='pvt_v2_b2') = PVTv2-B2 encoder + EMCAD_dual decoder with the K=9 dual-supervised reverse-attention heads:
  /root/reference/multiclass_seg/EMCAD/lib/networks.py:10-128, lib/decoders.py:14-22,69-77,83-251,407-526.
Functional torch-CPU code over a flat state_dict P (same conventions as oracle/pranet_oracle.py); the trainer's 15-subset loss is
EMCAD/trainer.py:106-140 with utils/utils.py:102-138 (DiceLoss).
"""
import math
from itertools import chain, combinations

import torch
import torch.nn.functional as F

from oracle.pranet_oracle import Ctx, bn, pvt_features


def _act(x, name):
    return F.relu6(x) if name == "relu6" else F.relu(x)


def channel_shuffle(x, groups):                                  # decoders.py:69-77
    b, c, h, w = x.shape
    return x.view(b, groups, c // groups, h, w).transpose(1, 2).contiguous().view(b, -1, h, w)


def mscb(P, p, x, ctx, kernel_sizes=(1, 3, 5), activation="relu6"):
    """MSCB.forward (decoders.py:141-160) with MSDC (:83-99), stride 1, add=True, dw_parallel=True, in == out channels."""
    c = x.shape[1]
    t = _act(bn(P, p + "pconv1.1", F.conv2d(x, P[p + "pconv1.0.weight"]), ctx), activation)
    ex = t.shape[1]
    dout = 0
    for i, k in enumerate(kernel_sizes):
        q = p + f"msdc.dwconvs.{i}."
        dout = dout + _act(bn(P, q + "1", F.conv2d(t, P[q + "0.weight"], None, 1, k // 2, 1, ex), ctx), activation)
    dout = channel_shuffle(dout, math.gcd(ex, c))
    out = bn(P, p + "pconv2.1", F.conv2d(dout, P[p + "pconv2.0.weight"]), ctx)
    return x + out


def eucb(P, p, x, ctx):
    """EUCB.forward (:182-186): nearest x2 -> depth-wise 3x3 + BN + ReLU -> channel_shuffle(groups = channels: identity) -> biased 1x1."""
    c = x.shape[1]
    t = F.interpolate(x, scale_factor=2)
    t = F.relu(bn(P, p + "up_dwc.2", F.conv2d(t, P[p + "up_dwc.1.weight"], None, 1, 1, 1, c), ctx))
    t = channel_shuffle(t, c)
    return F.conv2d(t, P[p + "pwc.0.weight"], P[p + "pwc.0.bias"])


def lgag(P, p, g, x, ctx):
    """LGAG.forward (:208-214): grouped 3x3 convs (groups = F_int) + BN on g and x, ReLU of the sum, 1x1 -> BN -> sigmoid gate on x."""
    fi = P[p + "W_g.0.weight"].shape[0]
    g1 = bn(P, p + "W_g.1", F.conv2d(g, P[p + "W_g.0.weight"], P[p + "W_g.0.bias"], 1, 1, 1, fi), ctx)
    x1 = bn(P, p + "W_x.1", F.conv2d(x, P[p + "W_x.0.weight"], P[p + "W_x.0.bias"], 1, 1, 1, fi), ctx)
    psi = F.relu(g1 + x1)
    psi = torch.sigmoid(bn(P, p + "psi.1", F.conv2d(psi, P[p + "psi.0.weight"], P[p + "psi.0.bias"]), ctx))
    return x * psi


def cab(P, p, x):
    """CAB.forward (:233-241): sigmoid(fc2(relu(fc1(avgpool))) + fc2(relu(fc1(maxpool))))."""
    a = F.adaptive_avg_pool2d(x, 1); mx = F.adaptive_max_pool2d(x, 1)
    f = lambda t: F.conv2d(F.relu(F.conv2d(t, P[p + "fc1.weight"])), P[p + "fc2.weight"])
    return torch.sigmoid(f(a) + f(mx))


def sab(P, p, x):
    """SAB.forward (:252-258): sigmoid(conv7x7([mean_c, max_c]))."""
    t = torch.cat([x.mean(1, keepdim=True), x.max(1, keepdim=True)[0]], 1)
    return torch.sigmoid(F.conv2d(t, P[p + "conv.weight"], None, 1, 3))


def _head(P, p, x, ctx, pad):
    return bn(P, p + ".bn", F.conv2d(x, P[p + ".conv.weight"], None, 1, pad), ctx)       # BasicConv2d (:14-22): no ReLU


def emcad_dual(P, p, x4, skips, ctx):
    """EMCAD_dual.forward (decoders.py:441-526)."""
    def stage(d, lvl):
        d = cab(P, p + f"cab{lvl}.", d) * d
        d = sab(P, p + "sab.", d) * d
        return mscb(P, p + f"mscb{lvl}.0.", d, ctx)
    d = stage(x4, 4)
    fg, bg = _head(P, p + "ConvBlock4_fg", d, ctx, 0), _head(P, p + "ConvBlock4_bg", d, ctx, 0)
    fgs, bgs = [fg], [bg]
    for lvl, skip in ((3, skips[0]), (2, skips[1]), (1, skips[2])):
        d = eucb(P, p + f"eucb{lvl}.", d, ctx)
        up_fg = F.interpolate(fg, size=d.shape[2:], mode="bilinear"); up_bg = F.interpolate(bg, size=d.shape[2:], mode="bilinear")
        d = d + lgag(P, p + f"lgag{lvl}.", d, skip, ctx)
        d = stage(d, lvl)
        fg, bg = _head(P, p + f"ConvBlock{lvl}_fg", d, ctx, 1), _head(P, p + f"ConvBlock{lvl}_bg", d, ctx, 1)
        fg = fg + fg.mul(F.softmax(up_fg - up_bg, dim=1))
        fgs.append(fg); bgs.append(bg)
    return fgs + bgs


def emcadnet_forward(P, x, training):
    """EMCADNet.forward, dual branch (networks.py:100-118): 8 full-resolution K-channel maps [p11, p12, p13, p14, p11_bg, ...]."""
    ctx = Ctx(training)
    if x.shape[1] == 1:
        x = F.relu(bn(P, "conv.1", F.conv2d(x, P["conv.0.weight"], P["conv.0.bias"]), ctx))
    x1, x2, x3, x4 = pvt_features(P, "backbone.", x)
    outs = emcad_dual(P, "decoder.", x4, [x3, x2, x1], ctx)
    scales = [32, 16, 8, 4] * 2
    return [F.interpolate(o, scale_factor=s, mode="bilinear") for o, s in zip(outs, scales)]


def powerset(items):                                              # EMCAD/utils/utils.py powerset
    return chain.from_iterable(combinations(items, r) for r in range(len(items) + 1))


def dice_loss(logits, target, n_classes):                         # utils/utils.py:102-138 with softmax=True
    prob = torch.softmax(logits, dim=1)
    loss = 0.0
    for i in range(n_classes):
        t = (target == i).float(); s = prob[:, i]
        loss = loss + (1 - (2 * (s * t).sum() + 1e-5) / ((s * s).sum() + (t * t).sum() + 1e-5))
    return loss / n_classes


def mutation_loss(outs, label, bg_mask, n_classes=9):
    """trainer.py:106-140 (dual, supervision='mutation'): over the 15 non-empty subsets of the 4 scales,
    0.5 CE(sum fg) + 0.7 Dice(softmax(sum fg)) + 0.3 BCEWithLogits(sum bg, bg_mask)."""
    P_fg, P_bg = outs[:4], outs[4:]
    loss = 0.0
    for s in powerset(range(4)):
        if not s:
            continue
        iout = sum(P_fg[i] for i in s); ibg = sum(P_bg[i] for i in s)
        loss = loss + 0.5 * F.cross_entropy(iout, label.long()) + 0.7 * dice_loss(iout, label, n_classes) + 0.3 * F.binary_cross_entropy_with_logits(ibg, bg_mask)
    return loss
