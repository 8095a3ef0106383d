"""PraNet-V2 (DSRA) models on the pn2 engine — same nn.Module surface as the reference's lib/pranet.py.

Class names, constructor signatures, sub-module names (hence state_dict keys) and the 8-tuple returned by
forward() match /root/reference/binary_seg/lib/pranet.py (BasicConv2d :31-43, RFB_modified :46-83,
aggregation :86-125, PVT_PraNet_V2 :129-263, PraNet_V2 :268-417), so MyTrain_med.py / MyTest_med.py import
this file unchanged.  forward() runs hand-written gfx950 kernels (csrc/) through pn2.engine; the
nn.Conv2d / nn.BatchNorm2d leaves only hold parameters.
"""
import math
import os

import torch
import torch.nn as nn

from lib.nn import SynchronizedBatchNorm2d
from pn2.capi import F32
from pn2.engine import rup
from pn2.graph import run_module
from .Res2Net_v1b import res2net50_v1b_26w_4s

BatchNorm2d = SynchronizedBatchNorm2d


def conv3x3_bn_relu(in_planes, out_planes, stride=1):
    "3x3 convolution + BN + relu (kept for import compatibility; nothing on the hot path calls it)"
    return nn.Sequential(nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False),
                         BatchNorm2d(out_planes), nn.ReLU(inplace=True))


class BasicConv2d(nn.Module):
    """conv -> BN, bias-free, and NO ReLU in forward (the `relu` member is unused, as in the reference :31-43)."""

    def __init__(self, in_planes, out_planes, kernel_size, stride=1, padding=0, dilation=1):
        super().__init__()
        self.conv = nn.Conv2d(in_planes, out_planes, kernel_size=kernel_size, stride=stride, padding=padding, dilation=dilation, bias=False)
        self.bn = nn.BatchNorm2d(out_planes)
        self.relu = nn.ReLU(inplace=True)

    def _build(self, eng, x, relu=False, residual=None, out=None, head=False, x_last=False, gate=None):
        """x_last: this conv is x's first consumer in forward order, so its data gradient completes x's gradient (engine.conv_bn_act).
        gate: V1 reverse-attention logits in front of this (1x1) conv, applied in the GEMM epilogue (engine.conv_bn_act)."""
        if head:      # K-channel fp32 head map; raw conv output keeps 8-channel padded rows
            K = self.conv.out_channels
            return eng.conv_bn_act(x, self.conv, self.bn, relu=relu, out_map=(K, rup(K, 8)), y_dt=F32, y_C=K, x_last=x_last)
        return eng.conv_bn_act(x, self.conv, self.bn, relu=relu, residual=residual, out=out, x_last=x_last, gate=gate)

    def forward(self, x):
        return run_module(lambda e, a: [self._build(e, a)], [x], list(self.parameters()), self.training)[0]


class RFB_modified(nn.Module):
    def __init__(self, in_channel, out_channel):
        super().__init__()
        self.relu = nn.ReLU(True)
        self.branch0 = nn.Sequential(BasicConv2d(in_channel, out_channel, 1))

        def branch(k):
            return nn.Sequential(
                BasicConv2d(in_channel, out_channel, 1),
                BasicConv2d(out_channel, out_channel, kernel_size=(1, k), padding=(0, k // 2)),
                BasicConv2d(out_channel, out_channel, kernel_size=(k, 1), padding=(k // 2, 0)),
                BasicConv2d(out_channel, out_channel, 3, padding=k, dilation=k))
        self.branch1, self.branch2, self.branch3 = branch(3), branch(5), branch(7)
        self.conv_cat = BasicConv2d(4 * out_channel, out_channel, 3, padding=1)
        self.conv_res = BasicConv2d(in_channel, out_channel, 1)

    def _build(self, eng, x, extra=()):
        """reference :75-83.  The five Cin->32 1x1 reducers (branch0..3 heads, conv_res) — plus any `extra` 1x1 BasicConv2d
        that reads the same map (the RA stage's conv1) — run as ONE fused GEMM over x (eng.conv_bn_multi); the branch tails write
        straight into the channel slices of the concat buffer, and relu(x_cat + conv_res(x)) is the residual epilogue of
        conv_cat's BN-apply pass.  Returns the RFB output (and the extra outputs, if any)."""
        c = self.conv_cat.conv.out_channels
        heads = eng.conv_bn_multi(x, [self.branch0[0], self.branch1[0], self.branch2[0], self.branch3[0], self.conv_res] + list(extra))
        cat = eng.new_act(x.N, x.H, x.W, 4 * c)
        eng.copy_into(heads[0], cat.slice(0, c))
        for bi, br in enumerate((self.branch1, self.branch2, self.branch3), start=1):
            t = br[1]._build(eng, heads[bi])
            t = br[2]._build(eng, t, x_last=True)                  # single-consumer chain
            br[3]._build(eng, t, out=cat.slice(bi * c, (bi + 1) * c), x_last=True)
        y = self.conv_cat._build(eng, cat, relu=True, residual=heads[4])
        return (y, heads[5:]) if extra else y

    def forward(self, x):
        return run_module(lambda e, a: [self._build(e, a)], [x], list(self.parameters()), self.training)[0]


def rfb_group(eng, rfbs, xs, extras):
    """The RFB modules of a model (reference :75-83, three per model) in LOCK STEP: they do not depend on each other, nor do the three branch tails
    inside each.  Phase A: the fused 1x1 reducer GEMM of every module; phase B: the 9 branch tails (1xk -> kx1 -> 3x3 dilated, k = 3, 5, 7) and the
    branch0 pass-through copies; phase C: conv_cat + residual + ReLU.  Each phase advances position by position with one table-driven launch per
    kernel kind (Engine.lockstep), forward and backward.  -> [(y, extra outputs)] like RFB_modified._build(extra=...)."""
    c = rfbs[0].conv_cat.conv.out_channels
    heads = eng.lockstep("rfb.heads", [lambda r=r, x=x, e=e: eng.conv_bn_multi(x, [r.branch0[0], r.branch1[0], r.branch2[0], r.branch3[0], r.conv_res] + list(e))
                                       for r, x, e in zip(rfbs, xs, extras)])
    cats = [eng.new_act(x.N, x.H, x.W, 4 * c) for x in xs]

    def tail(r, h, cat, bi):
        br = (r.branch1, r.branch2, r.branch3)[bi - 1]
        t = br[1]._build(eng, h[bi])
        t = br[2]._build(eng, t, x_last=True)
        br[3]._build(eng, t, out=cat.slice(bi * c, (bi + 1) * c), x_last=True)
    fns = [lambda r=r, h=h, cat=cat, bi=bi: tail(r, h, cat, bi) for r, h, cat in zip(rfbs, heads, cats) for bi in (1, 2, 3)]
    fns += [lambda h=h, cat=cat: eng.copy_into(h[0], cat.slice(0, c)) for h, cat in zip(heads, cats)]
    eng.lockstep("rfb.tails", fns)
    ys = eng.lockstep("rfb.cat", [lambda r=r, h=h, cat=cat: r.conv_cat._build(eng, cat, relu=True, residual=h[4]) for r, h, cat in zip(rfbs, heads, cats)])
    return [(y, h[5:]) for y, h in zip(ys, heads)]


class aggregation(nn.Module):
    def __init__(self, channel, num_class):
        super().__init__()
        self.relu = nn.ReLU(True)
        self.upsample = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
        self.conv_upsample1 = BasicConv2d(channel, channel, 3, padding=1)
        self.conv_upsample2 = BasicConv2d(channel, channel, 3, padding=1)
        self.conv_upsample3 = BasicConv2d(channel, channel, 3, padding=1)
        self.conv_upsample4 = BasicConv2d(channel, channel, 3, padding=1)
        self.conv_upsample5 = BasicConv2d(2 * channel, 2 * channel, 3, padding=1)
        self.conv_concat2 = BasicConv2d(2 * channel, 2 * channel, 3, padding=1)
        self.conv_concat3 = BasicConv2d(3 * channel, 3 * channel, 3, padding=1)
        self.conv4 = BasicConv2d(3 * channel, 3 * channel, 3, padding=1)
        self.conv5_fg = nn.Conv2d(3 * channel, num_class, 1)
        self.conv5_bg = nn.Conv2d(3 * channel, num_class, 1)

    def _build_trunk(self, eng, x1, x2, x3):
        """reference :109-121 up to conv4 (shared by the V1 single-head aggregation)."""
        c = x1.C
        up = lambda a: eng.bilinear(a, 2, align_corners=True)
        u1 = up(x1)                                   # reference recomputes upsample(x1) three times; same values
        cat2 = eng.new_act(x2.N, x2.H, x2.W, 2 * c)
        cat3 = eng.new_act(x3.N, x3.H, x3.W, 3 * c)
        # three independent conv+BN chains (conv_upsample1 on u1, conv_upsample2 on up(u1), conv_upsample3 on up(x2)) in lock step; conv_upsample4
        # also reads u1 - its data gradient accumulates into the same buffer as conv_upsample1's, so it cannot share their launches
        uu1, ux2 = up(u1), up(x2)
        cu1, cu2, cu3 = eng.lockstep("agg.up", [lambda: self.conv_upsample1._build(eng, u1),
                                                lambda: self.conv_upsample2._build(eng, uu1),
                                                lambda: self.conv_upsample3._build(eng, ux2)])
        self.conv_upsample4._build(eng, u1, out=cat2.slice(c, 2 * c))
        eng.mul(cu1, x2, out=cat2.slice(0, c))                                                       # x2_1
        t = eng.mul(cu2, cu3)
        eng.mul(t, x3, out=cat3.slice(0, c))                                                         # x3_1
        x2_2 = self.conv_concat2._build(eng, cat2)
        self.conv_upsample5._build(eng, up(x2_2), out=cat3.slice(c, 3 * c))
        x3_2 = self.conv_concat3._build(eng, cat3)
        return self.conv4._build(eng, x3_2, x_last=True)

    def _build(self, eng, x1, x2, x3):
        x = self._build_trunk(eng, x1, x2, x3)
        def head(conv):
            K = conv.out_channels
            return eng.conv_bn_act(x, conv, None, out_map=(K, rup(K, 8)), y_dt=F32, y_C=K, bias=conv.bias, x_last=conv is self.conv5_fg)   # fg: x's first consumer
        return [head(self.conv5_fg), head(self.conv5_bg)]

    def forward(self, x1, x2, x3):
        return run_module(lambda e, a, b, c: self._build(e, a, b, c), [x1, x2, x3], list(self.parameters()), self.training)


def _dsra_tail(m, eng, t1, ra5_fg, ra5_bg):
    """DSRA3/2/1 of reference :349-417 (identical in PVT_PraNet_V2 :205-263); returns the 8 lateral maps.
    t1[s] = ra{s}_conv1(x_s), already computed inside the fused reducer GEMM of the matching RFB."""
    sd = m.sem_downsample
    up = lambda a, s: eng.bilinear(a, s)

    def final(a, s, j):   # full-resolution lateral map j of the returned 8-tuple
        OH, OW = int(math.floor(a.H * s)), int(math.floor(a.W * s))
        return eng.bilinear(a, s, out=eng.lateral_out(j, 8, a.N, OH, OW, a.C))
    l5_fg, l5_bg = final(ra5_fg, 8 / sd, 3), final(ra5_bg, 8 / sd, 7)
    # The conv stacks of DSRA3/2/1 only depend on the encoder (t1[s]); the stages meet in the fusion with the up-sampled maps of the stage below.
    # So the three stacks - ra4: 3 x (5x5 + ReLU) + two 1x1 heads; ra3 / ra2: 2 x (3x3 + ReLU) + two 3x3 heads - advance in lock step, then the
    # cheap sequential chain crop -> fg + fg * softmax(crop_fg - crop_bg) -> up-sample follows.
    def stack(s):
        t = t1[s]
        if s == 4:
            t = m.ra4_conv2._build(eng, t, relu=True)
            t = m.ra4_conv3._build(eng, t, relu=True, x_last=True)
            t = m.ra4_conv4._build(eng, t, relu=True, x_last=True)
            return m.ra4_conv5_fg._build(eng, t, head=True, x_last=True), m.ra4_conv5_bg._build(eng, t, head=True)      # fg head: t's first consumer
        t = getattr(m, f"ra{s}_conv2")._build(eng, t, relu=True)
        t = getattr(m, f"ra{s}_conv3")._build(eng, t, relu=True, x_last=True)
        return getattr(m, f"ra{s}_conv4_fg")._build(eng, t, head=True, x_last=True), getattr(m, f"ra{s}_conv4_bg")._build(eng, t, head=True)
    (f4, b4), (f3, b3), (f2, b2) = eng.lockstep("dsra.stacks", [lambda: stack(4), lambda: stack(3), lambda: stack(2)])
    # K = 1 with the softmax gate (every model the binary scripts build): softmax over ONE channel is identically 1 whatever the crop maps hold, the fusion is fg + fg
    # and no gradient reaches the crops - so they are not resampled at all (six bilinear launches per step; the fusion kernel gets fg itself as a stand-in operand and
    # computes softmax(fg - fg) = 1: the same bits).  PN2_ZERO_CROP_SKIP=0 keeps the reference's data flow.
    from pn2 import core as _core
    if f4.C == 1 and m.use_softmax and _core.ZERO_CROP_SKIP:
        crop = lambda a, s, stand_in: stand_in
    else:
        crop = lambda a, s, stand_in: up(a, s)
    # ---- DSRA3
    f = eng.dsra_fuse(f4, crop(ra5_fg, 0.25, f4), crop(ra5_bg, 0.25, f4), m.use_softmax)
    b = b4
    l4_fg, l4_bg = final(f, 32 / sd, 2), final(b, 32 / sd, 6)
    lat = {}
    for s, u, fs, bs in ((3, 16, f3, b3), (2, 8, f2, b2)):
        f = eng.dsra_fuse(fs, crop(f, 2, fs), crop(b, 2, fs), m.use_softmax)
        b = bs
        lat[s] = (final(f, u / sd, s - 2), final(b, u / sd, s + 2))     # slot = position in the returned 8-tuple
    return [lat[2][0], lat[3][0], l4_fg, l5_fg, lat[2][1], lat[3][1], l4_bg, l5_bg]


class PraNet_V2(nn.Module):
    """Res2Net-50 encoder + RFBs + partial decoder + three dual-supervised reverse-attention stages."""

    def __init__(self, channel=32, num_class=3, sem_downsample=1, use_softmax=True):
        super().__init__()
        self.idx = range(10)
        self.num_class = num_class
        self.sem_downsample = sem_downsample
        self.use_softmax = use_softmax
        self.conv = nn.Sequential(nn.Conv2d(1, 3, kernel_size=1), nn.BatchNorm2d(3), nn.ReLU(inplace=True))   # unused by forward (reference :278-282)
        self.backbone = res2net50_v1b_26w_4s(pretrained=True)
        self.rfb2_1 = RFB_modified(512, channel)
        self.rfb3_1 = RFB_modified(1024, channel)
        self.rfb4_1 = RFB_modified(2048, channel)
        self.agg1 = aggregation(channel, self.num_class)
        self.ra4_conv1 = BasicConv2d(2048, 256, kernel_size=1)
        self.ra4_conv2 = BasicConv2d(256, 256, kernel_size=5, padding=2)
        self.ra4_conv3 = BasicConv2d(256, 256, kernel_size=5, padding=2)
        self.ra4_conv4 = BasicConv2d(256, 256, kernel_size=5, padding=2)
        self.ra4_conv5_fg = BasicConv2d(256, num_class, kernel_size=1)
        self.ra4_conv5_bg = BasicConv2d(256, num_class, kernel_size=1)
        self.ra3_conv1 = BasicConv2d(1024, 64, kernel_size=1)
        self.ra3_conv2 = BasicConv2d(64, 64, kernel_size=3, padding=1)
        self.ra3_conv3 = BasicConv2d(64, 64, kernel_size=3, padding=1)
        self.ra3_conv4_fg = BasicConv2d(64, num_class, kernel_size=3, padding=1)
        self.ra3_conv4_bg = BasicConv2d(64, num_class, kernel_size=3, padding=1)
        self.ra2_conv1 = BasicConv2d(512, 64, kernel_size=1)
        self.ra2_conv2 = BasicConv2d(64, 64, kernel_size=3, padding=1)
        self.ra2_conv3 = BasicConv2d(64, 64, kernel_size=3, padding=1)
        self.ra2_conv4_fg = BasicConv2d(64, num_class, kernel_size=3, padding=1)
        self.ra2_conv4_bg = BasicConv2d(64, num_class, kernel_size=3, padding=1)

    def hot_parameters(self):
        """Parameters that forward() touches (backbone.fc and self.conv never receive gradients — reference :329-417)."""
        return [p for n, p in self.named_parameters() if not (n.startswith('conv.') or n.startswith('backbone.fc.'))]

    def _build(self, eng, x):
        x1, x2, x3, x4 = self.backbone._build_features(eng, x)
        # the three RFB modules are independent of each other: they advance in lock step (table-driven launches, rfb_group)
        (x2_rfb, (t2,)), (x3_rfb, (t3,)), (x4_rfb, (t4,)) = rfb_group(eng, [self.rfb2_1, self.rfb3_1, self.rfb4_1], [x2, x3, x4],
                                                                  [[self.ra2_conv1], [self.ra3_conv1], [self.ra4_conv1]])
        ra5_fg, ra5_bg = self.agg1._build(eng, x4_rfb, x3_rfb, x2_rfb)
        return _dsra_tail(self, eng, {2: t2, 3: t3, 4: t4}, ra5_fg, ra5_bg)

    def forward(self, x, segSize=None):
        return run_module(self._build, [x], self.hot_parameters(), self.training)


class PVT_PraNet_V2(nn.Module):
    """PVTv2-B2 variant (reference :129-263): the transformer encoder of lib/pvtv2.py feeding exactly the PraNet_V2 heads."""

    def __init__(self, channel=32, num_class=3, sem_downsample=1, use_softmax=True):
        super().__init__()
        self.idx = range(10)
        self.num_class = num_class
        self.sem_downsample = sem_downsample
        self.use_softmax = use_softmax
        self.conv = nn.Sequential(nn.Conv2d(1, 3, kernel_size=1), nn.BatchNorm2d(3), nn.ReLU(inplace=True))
        from lib.pvtv2 import pvt_v2_b2
        self.backbone = pvt_v2_b2()
        path = './models/pvt_v2_b2.pth'
        if os.path.exists(path) or os.environ.get('PN2_NO_PRETRAINED', '0') != '1':
            save_model = torch.load(path)
            model_dict = self.backbone.state_dict()
            model_dict.update({k: v for k, v in save_model.items() if k in model_dict.keys()})
            self.backbone.load_state_dict(model_dict)
        self.rfb2_1 = RFB_modified(128, channel)
        self.rfb3_1 = RFB_modified(320, channel)
        self.rfb4_1 = RFB_modified(512, channel)
        self.agg1 = aggregation(channel, self.num_class)
        self.ra4_conv1 = BasicConv2d(512, 256, kernel_size=1)
        self.ra4_conv2 = BasicConv2d(256, 256, kernel_size=5, padding=2)
        self.ra4_conv3 = BasicConv2d(256, 256, kernel_size=5, padding=2)
        self.ra4_conv4 = BasicConv2d(256, 256, kernel_size=5, padding=2)
        self.ra4_conv5_fg = BasicConv2d(256, num_class, kernel_size=1)
        self.ra4_conv5_bg = BasicConv2d(256, num_class, kernel_size=1)
        self.ra3_conv1 = BasicConv2d(320, 64, kernel_size=1)
        self.ra3_conv2 = BasicConv2d(64, 64, kernel_size=3, padding=1)
        self.ra3_conv3 = BasicConv2d(64, 64, kernel_size=3, padding=1)
        self.ra3_conv4_fg = BasicConv2d(64, num_class, kernel_size=3, padding=1)
        self.ra3_conv4_bg = BasicConv2d(64, num_class, kernel_size=3, padding=1)
        self.ra2_conv1 = BasicConv2d(128, 64, kernel_size=1)
        self.ra2_conv2 = BasicConv2d(64, 64, kernel_size=3, padding=1)
        self.ra2_conv3 = BasicConv2d(64, 64, kernel_size=3, padding=1)
        self.ra2_conv4_fg = BasicConv2d(64, num_class, kernel_size=3, padding=1)
        self.ra2_conv4_bg = BasicConv2d(64, num_class, kernel_size=3, padding=1)

    def hot_parameters(self, gray=False):
        """Parameters that forward() touches (self.conv only serves 1-channel inputs — reference :190-191)."""
        return [p for n, p in self.named_parameters() if gray or not n.startswith('conv.')]

    def _build(self, eng, x):
        """1-channel inputs go through conv(1->3)+BN+ReLU first (reference :190-191); then backbone._build_features -> RFBs -> aggregation ->
        DSRA tail: identical head code (:193-263)"""
        if x.C == 1:
            x = eng.conv_bn_act(x, self.conv[0], self.conv[1], relu=True, bias=self.conv[0].bias)
        return PraNet_V2._build(self, eng, x)

    def forward(self, x, segSize=None):
        """segSize is accepted and unused, as in the reference (:189)."""
        return run_module(self._build, [x], self.hot_parameters(x.shape[1] == 1), self.training)
