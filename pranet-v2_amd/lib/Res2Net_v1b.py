"""Res2Net-50 v1b (26w x 4s) encoder on the pn2 engine.

Same class / function names, constructor arguments, attribute names and state_dict keys as the
reference's lib/Res2Net_v1b.py (Bottle2neck :15-91, Res2Net :94-167, factories :170-226), so
checkpoints and `model.backbone.layerN` style access carry over.  nn.Conv2d / nn.BatchNorm2d leaves
are parameter containers; the arithmetic runs in the gfx950 kernels via `_build` (NHWC, the 26/52-wide
Res2Net splits stored in 32/56-channel padded groups so every pixel row stays 16-byte aligned).
"""
import math
import os

import torch
import torch.nn as nn

from pn2.engine import rup
from pn2.graph import run_module

__all__ = ['Res2Net', 'res2net50_v1b', 'res2net101_v1b', 'res2net50_v1b_26w_4s']


ALIAS_CAT_GRAD = os.environ.get("PN2_ALIAS_CAT_GRAD", "1") == "1"


def TEE_CONCAT():
    from pn2 import core
    return core.TEE_CONCAT


def POOL_FUSE():
    from pn2 import core
    return core.POOL_FUSE


class Bottle2neck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, baseWidth=26, scale=4, stype='normal'):
        super().__init__()
        width = int(math.floor(planes * (baseWidth / 64.0)))
        self.conv1 = nn.Conv2d(inplanes, width * scale, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(width * scale)
        self.nums = 1 if scale == 1 else scale - 1
        if stype == 'stage':
            self.pool = nn.AvgPool2d(kernel_size=3, stride=stride, padding=1)
        self.convs = nn.ModuleList(nn.Conv2d(width, width, kernel_size=3, stride=stride, padding=1, bias=False) for _ in range(self.nums))
        self.bns = nn.ModuleList(nn.BatchNorm2d(width) for _ in range(self.nums))
        self.conv3 = nn.Conv2d(width * scale, planes * self.expansion, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stype = stype
        self.scale = scale
        self.width = width

    def _build(self, eng, x):
        """Reference Bottle2neck.forward (Res2Net_v1b.py:58-91) expressed in engine ops."""
        assert self.scale > 1
        w, sc = self.width, self.scale
        wp = rup(w, 8)
        stage = self.stype == 'stage'
        stride = self.convs[0].stride[0]
        # x_last: conv1 is the first consumer of the block input, so its data gradient is the last contribution to that gradient - the dgrad
        # GEMM takes the BatchNorm-backward statistics of the previous block's bn3 in its epilogue (engine.conv_bn_act)
        OH = (x.H + 2 - 3) // stride + 1              # (conv1 is 1x1, stride 1: out1 has x's extent)
        OW = (x.W + 2 - 3) // stride + 1
        cat = eng.new_act(x.N, OH, OW, w * sc, w, wp)
        last = cat.slice(self.nums * wp, sc * wp, w, w, wp)
        # a normal block passes spx[3] through unchanged (Res2Net_v1b.py:78-79): in a training pass the launch that writes out1 writes that slice into the concat buffer as well
        tee = (last, self.nums * wp) if (not stage and eng.training and TEE_CONCAT()) else None
        out1 = eng.conv_bn_act(x, self.conv1, self.bn1, relu=True, out_map=(w, wp), x_last=True, tee=tee)
        spx = [out1.slice(i * wp, (i + 1) * wp, w, w, wp) for i in range(sc)]
        # the branch convs leave their raw outputs and BatchNorm rows in channel slices of two shared buffers, so that conv3's dgrad can take
        # the backward statistics of bns[0..2] (and of bn1's last slice, which the concat buffer copies) in one epilogue
        rawcat = eng.empty(out1.N, OH, OW, sc * wp)
        pcat = eng.fbuf(4, sc * wp)
        def branch(i, s_in, nxt):
            return eng.conv_bn_act(s_in, self.convs[i], self.bns[i], relu=True, out=cat.slice(i * wp, (i + 1) * wp, w, w, wp), out_map=(w, wp), sum_with=nxt,
                                   raw_out=rawcat[..., i * wp:(i + 1) * wp], par_out=pcat[:, i * wp:(i + 1) * wp], x_last=True)
        if stage:
            # the branches of a stage block are independent (sp = spx[i], Res2Net_v1b.py:66-69): they advance in lock step, one table-driven
            # launch per kernel kind (conv, finalize, BN-apply; and their backward) for all three, next to the pooled pass-through slice
            eng.lockstep("b2n.stage", [lambda i=i: branch(i, spx[i], None) for i in range(self.nums)] + [lambda: eng.avgpool(spx[self.nums], 3, stride, 1, out=last)])
            eng.concat_bnb(cat, rawcat, pcat, split=self.nums * wp, tail=None)
        else:
            s_in = spx[0]
            for i in range(self.nums):
                # sp_i = relu(bn(conv(s_in))) goes into the concat buffer; the next branch's input sp_i + spx[i+1] (Res2Net_v1b.py:66-68) is written
                # by the same pass (spx[i+1] feeds only that sum: its slice of conv1's gradient doubles as the sum's gradient buffer)
                nxt = spx[i + 1] if i + 1 < self.nums else None
                r = branch(i, s_in, nxt)
                if nxt is not None:
                    s_in = r[1]
            eng.copy_into(spx[self.nums], last, forward=tee is None)
            eng.concat_bnb(cat, rawcat, pcat, split=self.nums * wp, tail=spx[self.nums])
            if ALIAS_CAT_GRAD and cat.t.shape == out1.t.shape:
                # d(cat) and d(out1) share ONE buffer: slice i of d(cat) (the gradient of sp_i) is dead once bns[i]'s backward has formed dz_i, which is
                # before conv_i's dgrad writes slice i of d(out1); the last slice IS d(spx[3]) - the copy's backward has nothing left to move
                cat.galias = out1
        if self.downsample is not None:
            pool, dconv, dbn = self.downsample[0], self.downsample[1], self.downsample[2]
            k = pool.kernel_size if isinstance(pool.kernel_size, int) else pool.kernel_size[0]
            # (fold_bwd: conv1 above is x's first consumer - its dgrad runs last in the backward pass and adds this pool's gradient in its epilogue)
            r = x if k == 1 else eng.avgpool(x, k, k, 0, ceil_mode=True, count_include_pad=False, fold_bwd=True)
            res = eng.conv_bn_act(r, dconv, dbn, relu=False)
        else:
            res = x
        return eng.conv_bn_act(cat, self.conv3, self.bn3, relu=True, residual=res, x_last=True)

    def forward(self, x):
        return run_module(lambda e, a: [self._build(e, a)], [x], list(self.parameters()), self.training)[0]


class Res2Net(nn.Module):
    def __init__(self, block, layers, baseWidth=26, scale=4, num_classes=1000):
        self.inplanes = 64
        super().__init__()
        self.baseWidth = baseWidth
        self.scale = scale
        self.conv1 = nn.Sequential(
            nn.Conv2d(3, 32, 3, 2, 1, bias=False), nn.BatchNorm2d(32), nn.ReLU(inplace=True),
            nn.Conv2d(32, 32, 3, 1, 1, bias=False), nn.BatchNorm2d(32), nn.ReLU(inplace=True),
            nn.Conv2d(32, 64, 3, 1, 1, bias=False))
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU()
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.AvgPool2d(kernel_size=stride, stride=stride, ceil_mode=True, count_include_pad=False),
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=1, bias=False),
                nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample=downsample, stype='stage', baseWidth=self.baseWidth, scale=self.scale)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, baseWidth=self.baseWidth, scale=self.scale))
        return nn.Sequential(*layers)

    def _build_features(self, eng, x):
        """stem + maxpool + layer1..4 -> (x1, x2, x3, x4), as PraNet_V2.forward drives the backbone (pranet.py:331-341)."""
        c = self.conv1
        x = eng.conv_bn_act(x, c[0], c[1], relu=True)
        x = eng.conv_bn_act(x, c[3], c[4], relu=True, x_last=True)          # single-consumer chain: see Bottle2neck._build
        if eng.training and POOL_FUSE() and rup(c[6].out_channels, 8) == c[6].out_channels:
            # bn1 -> relu -> maxpool (Res2Net_v1b.py:137-139) as one op: the 176 x 176 x 64 BatchNorm output is consumed by the pool alone and never written
            x = eng.conv_bn_act(x, c[6], self.bn1, relu=True, x_last=True, pool=True)
        else:
            x = eng.conv_bn_act(x, c[6], self.bn1, relu=True, x_last=True)
            x = eng.maxpool3x3s2(x)
        feats = []
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                x = blk._build(eng, x)
            feats.append(x)
        return feats

    def forward(self, x):
        """Classification forward of the reference (Res2Net_v1b.py:150-167): features on the engine, the
        1000-way pooled head (never used by PraNet) through torch."""
        ps = [p for n, p in self.named_parameters() if not n.startswith('fc.')]
        x4 = run_module(lambda e, a: [self._build_features(e, a)[3]], [x], ps, self.training)[0]
        return self.fc(self.avgpool(x4).flatten(1))


def _load_pretrained(model, path):
    """Reference behaviour (Res2Net_v1b.py:196-200): torch.load of a fixed relative path.  Missing file is an
    error unless PN2_NO_PRETRAINED=1 (benchmarks / tests use random init: the checkpoint is not redistributable here)."""
    if os.environ.get('PN2_NO_PRETRAINED', '0') == '1' and not os.path.exists(path):
        return model
    model.load_state_dict(torch.load(path))
    return model


def res2net50_v1b(pretrained=False, **kwargs):
    model = Res2Net(Bottle2neck, [3, 4, 6, 3], baseWidth=26, scale=4, **kwargs)
    return _load_pretrained(model, '../models/res2net50_v1b_26w_4s-3cf99910.pth') if pretrained else model


def res2net101_v1b(pretrained=False, **kwargs):
    model = Res2Net(Bottle2neck, [3, 4, 23, 3], baseWidth=26, scale=4, **kwargs)
    return _load_pretrained(model, '../models/res2net101_v1b_26w_4s-0812c246.pth') if pretrained else model


def res2net50_v1b_26w_4s(pretrained=False, **kwargs):
    model = Res2Net(Bottle2neck, [3, 4, 6, 3], baseWidth=26, scale=4, **kwargs)
    return _load_pretrained(model, '../models/res2net50_v1b_26w_4s-3cf99910.pth') if pretrained else model


def res2net101_v1b_26w_4s(pretrained=False, **kwargs):
    model = Res2Net(Bottle2neck, [3, 4, 23, 3], baseWidth=26, scale=4, **kwargs)
    return _load_pretrained(model, '../models/res2net101_v1b_26w_4s-0812c246.pth') if pretrained else model


def res2net152_v1b_26w_4s(pretrained=False, **kwargs):
    model = Res2Net(Bottle2neck, [3, 8, 36, 3], baseWidth=26, scale=4, **kwargs)
    return _load_pretrained(model, '../models/res2net152_v1b_26w_4s-0812c246.pth') if pretrained else model
