"""PraNet (V1, reverse attention) on the pn2 engine — surface of the reference's lib/PraNet_Res2Net.py.

Names / state_dict keys follow /root/reference/binary_seg/lib/PraNet_Res2Net.py (aggregation :64-98, PraNet :101-186,
PVT_PraNet :188-273).  The (1 - sigmoid(crop)) gate multiply is one fused kernel (pn2_ra_gate_*), not a materialised
expand().mul() copy.
"""
import os

import torch
import torch.nn as nn

from pn2.capi import F32
from pn2.engine import rup
from pn2.graph import run_module
from .Res2Net_v1b import res2net50_v1b_26w_4s
from .pranet import BasicConv2d, RFB_modified, aggregation as _aggregation_v2


class aggregation(_aggregation_v2):
    """Single-head partial decoder (reference :64-98): conv5 instead of conv5_fg / conv5_bg."""

    def __init__(self, channel):
        nn.Module.__init__(self)
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
        self.conv5 = nn.Conv2d(3 * channel, 1, 1)

    def _build(self, eng, x1, x2, x3):
        x = self._build_trunk(eng, x1, x2, x3)
        return [eng.conv_bn_act(x, self.conv5, None, out_map=(1, 8), y_dt=F32, y_C=1, bias=self.conv5.bias)]

    def forward(self, x1, x2, x3):
        return run_module(lambda e, a, b, c: self._build(e, a, b, c), [x1, x2, x3], list(self.parameters()), self.training)[0]


class PraNet(nn.Module):
    def __init__(self, channel=32):
        super().__init__()
        self.resnet = res2net50_v1b_26w_4s(pretrained=True)
        self.rfb2_1 = RFB_modified(512, channel)
        self.rfb3_1 = RFB_modified(1024, channel)
        self.rfb4_1 = RFB_modified(2048, channel)
        self.agg1 = aggregation(channel)
        self.ra4_conv1 = BasicConv2d(2048, 256, kernel_size=1)
        self.ra4_conv2 = BasicConv2d(256, 256, kernel_size=5, padding=2)
        self.ra4_conv3 = BasicConv2d(256, 256, kernel_size=5, padding=2)
        self.ra4_conv4 = BasicConv2d(256, 256, kernel_size=5, padding=2)
        self.ra4_conv5 = BasicConv2d(256, 1, kernel_size=1)
        self.ra3_conv1 = BasicConv2d(1024, 64, kernel_size=1)
        self.ra3_conv2 = BasicConv2d(64, 64, kernel_size=3, padding=1)
        self.ra3_conv3 = BasicConv2d(64, 64, kernel_size=3, padding=1)
        self.ra3_conv4 = BasicConv2d(64, 1, kernel_size=3, padding=1)
        self.ra2_conv1 = BasicConv2d(512, 64, kernel_size=1)
        self.ra2_conv2 = BasicConv2d(64, 64, kernel_size=3, padding=1)
        self.ra2_conv3 = BasicConv2d(64, 64, kernel_size=3, padding=1)
        self.ra2_conv4 = BasicConv2d(64, 1, kernel_size=3, padding=1)

    def hot_parameters(self):
        return [p for n, p in self.named_parameters() if not n.startswith('resnet.fc.')]

    def _build(self, eng, x):
        """reference :130-186"""
        x1, x2, x3, x4 = self.resnet._build_features(eng, x)
        return _ra_heads(self, eng, x2, x3, x4)

    def forward(self, x):
        return run_module(self._build, [x], self.hot_parameters(), self.training)


def _ra_heads(m, eng, x2, x3, x4):
    """RFBs, partial decoder and the three reverse-attention branches shared by PraNet (:141-186) and PVT_PraNet (:229-273):
    x = (1 - sigmoid(crop)).expand(C) * x_l  ->  conv stack  ->  + crop  ->  up-sample.  The gate sits in front of a 1x1 conv, so it is applied to
    that GEMM's accumulator rows (conv(g * x) = g * conv(x) for a per-pixel g): the gated copy of x_l is never written."""
    x2_rfb = m.rfb2_1._build(eng, x2)
    x3_rfb = m.rfb3_1._build(eng, x3)
    x4_rfb = m.rfb4_1._build(eng, x4)
    ra5 = m.agg1._build(eng, x4_rfb, x3_rfb, x2_rfb)[0]
    l5 = eng.bilinear(ra5, 8)
    crop = eng.bilinear(ra5, 0.25)
    fuse = os.environ.get("PN2_RA_GATE_FUSED", "1") == "1"           # 0: the gate as its own pass over x_l (pn2_ra_gate_fwd / _bwd)
    gated = (lambda conv, xl, c: conv._build(eng, xl, gate=c)) if fuse else (lambda conv, xl, c: conv._build(eng, eng.ra_gate(xl, c)))
    t = gated(m.ra4_conv1, x4, crop)
    t = m.ra4_conv2._build(eng, t, relu=True)
    t = m.ra4_conv3._build(eng, t, relu=True)
    t = m.ra4_conv4._build(eng, t, relu=True)
    x = eng.add(m.ra4_conv5._build(eng, t, head=True), crop)
    l4 = eng.bilinear(x, 32)
    lat = {}
    for s, xs, u in ((3, x3, 16), (2, x2, 8)):
        crop = eng.bilinear(x, 2)
        t = gated(getattr(m, f"ra{s}_conv1"), xs, crop)
        t = getattr(m, f"ra{s}_conv2")._build(eng, t, relu=True)
        t = getattr(m, f"ra{s}_conv3")._build(eng, t, relu=True)
        x = eng.add(getattr(m, f"ra{s}_conv4")._build(eng, t, head=True), crop)
        lat[s] = eng.bilinear(x, u)
    return [l5, l4, lat[3], lat[2]]


class PVT_PraNet(nn.Module):
    """reference :188-273: the PVTv2-B2 encoder of lib/pvtv2.py feeding the same reverse-attention heads as PraNet."""

    def __init__(self, channel=32):
        super().__init__()
        from lib.pvtv2 import pvt_v2_b2
        self.backbone = pvt_v2_b2()
        path = './models/pvt_v2_b2.pth'
        if os.path.exists(path) or os.environ.get('PN2_NO_PRETRAINED', '0') != '1':      # reference :196-201 (hard-loads the checkpoint)
            save_model = torch.load(path)
            model_dict = self.backbone.state_dict()
            model_dict.update({k: v for k, v in save_model.items() if k in model_dict.keys()})
            self.backbone.load_state_dict(model_dict)
        self.rfb2_1 = RFB_modified(128, channel)
        self.rfb3_1 = RFB_modified(320, channel)
        self.rfb4_1 = RFB_modified(512, channel)
        self.agg1 = aggregation(channel)
        self.ra4_conv1 = BasicConv2d(512, 256, kernel_size=1)
        self.ra4_conv2 = BasicConv2d(256, 256, kernel_size=5, padding=2)
        self.ra4_conv3 = BasicConv2d(256, 256, kernel_size=5, padding=2)
        self.ra4_conv4 = BasicConv2d(256, 256, kernel_size=5, padding=2)
        self.ra4_conv5 = BasicConv2d(256, 1, kernel_size=1)
        self.ra3_conv1 = BasicConv2d(320, 64, kernel_size=1)
        self.ra3_conv2 = BasicConv2d(64, 64, kernel_size=3, padding=1)
        self.ra3_conv3 = BasicConv2d(64, 64, kernel_size=3, padding=1)
        self.ra3_conv4 = BasicConv2d(64, 1, kernel_size=3, padding=1)
        self.ra2_conv1 = BasicConv2d(128, 64, kernel_size=1)
        self.ra2_conv2 = BasicConv2d(64, 64, kernel_size=3, padding=1)
        self.ra2_conv3 = BasicConv2d(64, 64, kernel_size=3, padding=1)
        self.ra2_conv4 = BasicConv2d(64, 1, kernel_size=3, padding=1)

    def hot_parameters(self):
        return list(self.parameters())

    def _build(self, eng, x):
        """reference :226-273"""
        x1, x2, x3, x4 = self.backbone._build_features(eng, x)
        return _ra_heads(self, eng, x2, x3, x4)

    def forward(self, x):
        return run_module(self._build, [x], self.hot_parameters(), self.training)
