"""EMCAD decoder with dual-supervised K-class heads (reference: multiclass_seg/EMCAD/lib/decoders.py) on the gfx950 kernels.

Same class names, constructor signatures and parameter names (state_dict keys) as the reference; the computation is expressed in
engine ops (pn2/engine.py + ops_*.py): 1x1 / 3x3 / 7x7 convs on the implicit-GEMM kernels, BatchNorm on pn2_bn_*, and the depth-wise, grouped,
gating, pooling, shuffle and up-sampling pieces on csrc/pn2_emcad.hip.  Only the configuration the reference trains (train_synapse.py:
kernel_sizes [1,3,5], expansion 2, dw_parallel, add, stride 1, lgag_ks 3, relu6 in the MSCBs) is built; other switches raise.
No PyTorch fallback: forward needs the GPU library.
"""
import math

import torch
import torch.nn as nn

from pn2 import F32
from pn2.graph import run_module


def gcd(a, b):
    while b:
        a, b = b, a % b
    return a


class BasicConv2d(nn.Module):
    """conv -> BN, no ReLU in forward (decoders.py:14-22)."""

    def __init__(self, in_planes, out_planes, kernel_size, stride=1, padding=0, dilation=1):
        super().__init__()
        self.conv = nn.Conv2d(in_planes, out_planes, kernel_size=kernel_size, stride=stride, padding=padding, dilation=dilation, bias=False)
        self.bn = nn.BatchNorm2d(out_planes)
        self.relu = nn.ReLU(inplace=True)

    def _build(self, eng, x):      # K-channel fp32 head map
        K = self.conv.out_channels
        return eng.conv_bn_act(x, self.conv, self.bn, out_map=(K, (K + 7) // 8 * 8), y_dt=F32, y_C=K)


def _init_weights(module, scheme='normal'):            # decoders.py:24-54 with the scheme every block passes
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            if scheme == 'normal':
                nn.init.normal_(m.weight, std=.02)
            else:
                fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
                nn.init.normal_(m.weight, 0, math.sqrt(2.0 / fan_out))
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, (nn.BatchNorm2d, nn.LayerNorm)):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)


def act_layer(act, inplace=False):
    act = act.lower()
    if act == 'relu':
        return nn.ReLU(inplace)
    if act == 'relu6':
        return nn.ReLU6(inplace)
    raise NotImplementedError('activation layer [%s] is not built (relu / relu6 only)' % act)


def _act_code(m):
    return 2 if isinstance(m, nn.ReLU6) else True


class MSDC(nn.Module):
    """Multi-scale depth-wise convolutions (:83-102): parallel k x k depth-wise conv + BN + act for every k."""

    def __init__(self, in_channels, kernel_sizes, stride, activation='relu6', dw_parallel=True):
        super().__init__()
        assert stride == 1 and dw_parallel, "only the stride-1 parallel configuration is built"
        self.in_channels, self.kernel_sizes, self.activation, self.dw_parallel = in_channels, kernel_sizes, activation, dw_parallel
        self.dwconvs = nn.ModuleList([
            nn.Sequential(nn.Conv2d(in_channels, in_channels, k, stride, k // 2, groups=in_channels, bias=False), nn.BatchNorm2d(in_channels), act_layer(activation, inplace=True))
            for k in kernel_sizes])
        _init_weights(self)

    def _build(self, eng, x):
        return [eng.dwconv_bn_act(x, seq[0], seq[1], relu=_act_code(seq[2])) for seq in self.dwconvs]


class MSCB(nn.Module):
    """Multi-scale convolution block (:104-164)."""

    def __init__(self, in_channels, out_channels, stride, kernel_sizes=[1, 3, 5], expansion_factor=2, dw_parallel=True, add=True, activation='relu6'):
        super().__init__()
        assert stride == 1 and add and in_channels == out_channels, "only the stride-1 / add / in == out configuration is built"
        self.in_channels, self.out_channels, self.stride, self.kernel_sizes = in_channels, out_channels, stride, kernel_sizes
        self.expansion_factor, self.dw_parallel, self.add, self.activation = expansion_factor, dw_parallel, add, activation
        self.n_scales = len(kernel_sizes)
        self.use_skip_connection = True
        self.ex_channels = int(in_channels * expansion_factor)
        self.pconv1 = nn.Sequential(nn.Conv2d(in_channels, self.ex_channels, 1, 1, 0, bias=False), nn.BatchNorm2d(self.ex_channels), act_layer(activation, inplace=True))
        self.msdc = MSDC(self.ex_channels, kernel_sizes, stride, activation, dw_parallel=dw_parallel)
        self.combined_channels = self.ex_channels
        self.pconv2 = nn.Sequential(nn.Conv2d(self.combined_channels, out_channels, 1, 1, 0, bias=False), nn.BatchNorm2d(out_channels))
        _init_weights(self)

    def _build(self, eng, x):
        t = eng.conv_bn_act(x, self.pconv1[0], self.pconv1[1], relu=_act_code(self.pconv1[2]))
        dout = eng.shuffled_sum(self.msdc._build(eng, t), gcd(self.combined_channels, self.out_channels))
        return eng.conv_bn_act(dout, self.pconv2[0], self.pconv2[1], residual=x)


def MSCBLayer(in_channels, out_channels, n=1, stride=1, kernel_sizes=[1, 3, 5], expansion_factor=2, dw_parallel=True, add=True, activation='relu6'):
    convs = [MSCB(in_channels, out_channels, stride, kernel_sizes=kernel_sizes, expansion_factor=expansion_factor, dw_parallel=dw_parallel, add=add, activation=activation)]
    for _ in range(1, n):
        convs.append(MSCB(out_channels, out_channels, 1, kernel_sizes=kernel_sizes, expansion_factor=expansion_factor, dw_parallel=dw_parallel, add=add, activation=activation))
    return nn.Sequential(*convs)


class EUCB(nn.Module):
    """Efficient up-convolution block (:166-186): nearest x2, depth-wise 3x3 + BN + ReLU, channel_shuffle(groups = channels) = identity, biased 1x1."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, activation='relu'):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.up_dwc = nn.Sequential(nn.Upsample(scale_factor=2),
                                    nn.Conv2d(in_channels, in_channels, kernel_size=kernel_size, stride=stride, padding=kernel_size // 2, groups=in_channels, bias=False),
                                    nn.BatchNorm2d(in_channels), act_layer(activation, inplace=True))
        self.pwc = nn.Sequential(nn.Conv2d(in_channels, out_channels, kernel_size=1, stride=1, padding=0, bias=True))
        _init_weights(self)

    def _build(self, eng, x):
        t = eng.dwconv_bn_act(eng.upsample2x(x), self.up_dwc[1], self.up_dwc[2], relu=_act_code(self.up_dwc[3]))
        return eng.conv_bias(t, self.pwc[0])


class LGAG(nn.Module):
    """Large-kernel grouped attention gate (:188-214)."""

    def __init__(self, F_g, F_l, F_int, kernel_size=3, groups=1, activation='relu'):
        super().__init__()
        if kernel_size == 1:
            groups = 1
        assert kernel_size == 3 and groups == F_int and F_g == 2 * F_int and F_l == 2 * F_int and activation == 'relu', "only the EMCAD configuration (3x3, groups = F_int = C/2) is built"
        self.W_g = nn.Sequential(nn.Conv2d(F_g, F_int, kernel_size=kernel_size, stride=1, padding=kernel_size // 2, groups=groups, bias=True), nn.BatchNorm2d(F_int))
        self.W_x = nn.Sequential(nn.Conv2d(F_l, F_int, kernel_size=kernel_size, stride=1, padding=kernel_size // 2, groups=groups, bias=True), nn.BatchNorm2d(F_int))
        self.psi = nn.Sequential(nn.Conv2d(F_int, 1, kernel_size=1, stride=1, padding=0, bias=True), nn.BatchNorm2d(1), nn.Sigmoid())
        self.activation = act_layer(activation, inplace=True)
        _init_weights(self)

    def _build(self, eng, g, x):
        g1 = eng.pairconv_bn(g, self.W_g[0], self.W_g[1])
        t = eng.pairconv_bn(x, self.W_x[0], self.W_x[1], relu=True, residual=g1)             # relu(g1 + x1)
        pre = eng.conv_bn_act(t, self.psi[0], self.psi[1], bias=self.psi[0].bias, out_map=(1, 8), y_dt=F32, y_C=1)
        return eng.sigmoid_gate(x, pre, 1)


class CAB(nn.Module):
    """Channel attention block (:216-241): returns the gated input (the reference returns the gate and multiplies at the call site)."""

    def __init__(self, in_channels, out_channels=None, ratio=16, activation='relu'):
        super().__init__()
        self.in_channels = in_channels
        if in_channels < ratio:
            ratio = in_channels
        self.reduced_channels = in_channels // ratio
        self.out_channels = in_channels if out_channels is None else out_channels
        assert self.out_channels == in_channels
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.max_pool = nn.AdaptiveMaxPool2d(1)
        self.activation = act_layer(activation, inplace=True)
        self.fc1 = nn.Conv2d(in_channels, self.reduced_channels, 1, bias=False)
        self.fc2 = nn.Conv2d(self.reduced_channels, self.out_channels, 1, bias=False)
        self.sigmoid = nn.Sigmoid()
        _init_weights(self)

    def _build_gated(self, eng, x):
        """cab(x) * x"""
        avg, mx = eng.global_pool(x)
        f = lambda t: eng.conv_bn_act(eng.conv_bn_act(t, self.fc1, None, relu=True), self.fc2, None)
        return eng.sigmoid_gate(x, eng.add(f(avg), f(mx)), 0)


class SAB(nn.Module):
    """Spatial attention block (:243-258)."""

    def __init__(self, kernel_size=7):
        super().__init__()
        assert kernel_size in (3, 7, 11), 'kernel must be 3 or 7 or 11'
        self.conv = nn.Conv2d(2, 1, kernel_size, padding=kernel_size // 2, bias=False)
        self.sigmoid = nn.Sigmoid()
        _init_weights(self)

    def _build_gated(self, eng, x):
        """sab(x) * x"""
        pre = eng.conv_bn_act(eng.chan_stats(x), self.conv, None, out_map=(1, 8), y_dt=F32, y_C=1)
        return eng.sigmoid_gate(x, pre, 1)


class EMCAD_dual(nn.Module):
    """EMCAD decoder with the dual-supervised reverse-attention heads (:407-526)."""

    def __init__(self, channels=[512, 320, 128, 64], kernel_sizes=[1, 3, 5], expansion_factor=6, dw_parallel=True, add=True, lgag_ks=3, activation='relu6', num_class=None):
        super().__init__()
        assert num_class is not None
        eucb_ks = 3
        mk = lambda c: MSCBLayer(c, c, n=1, stride=1, kernel_sizes=kernel_sizes, expansion_factor=expansion_factor, dw_parallel=dw_parallel, add=add, activation=activation)
        self.mscb4 = mk(channels[0])
        for i, lvl in ((1, 3), (2, 2), (3, 1)):
            setattr(self, f"eucb{lvl}", EUCB(in_channels=channels[i - 1], out_channels=channels[i], kernel_size=eucb_ks, stride=eucb_ks // 2))
            setattr(self, f"lgag{lvl}", LGAG(F_g=channels[i], F_l=channels[i], F_int=channels[i] // 2, kernel_size=lgag_ks, groups=channels[i] // 2))
            setattr(self, f"mscb{lvl}", mk(channels[i]))
        for i, lvl in enumerate((4, 3, 2, 1)):
            setattr(self, f"cab{lvl}", CAB(channels[i]))
        self.sab = SAB()
        for i, lvl in enumerate((4, 3, 2, 1)):
            k, p = (1, 0) if lvl == 4 else (3, 1)
            setattr(self, f"ConvBlock{lvl}_fg", BasicConv2d(channels[i], num_class, kernel_size=k, padding=p))
            setattr(self, f"ConvBlock{lvl}_bg", BasicConv2d(channels[i], num_class, kernel_size=k, padding=p))

    def _stage(self, eng, d, lvl):
        d = getattr(self, f"cab{lvl}")._build_gated(eng, d)
        d = self.sab._build_gated(eng, d)
        return getattr(self, f"mscb{lvl}")[0]._build(eng, d)

    def _build(self, eng, x, skips):
        """forward :441-526 -> [d4_fg, d3_fg, d2_fg, d1_fg, d4_bg, d3_bg, d2_bg, d1_bg] (low-resolution K-channel fp32 maps)"""
        d = self._stage(eng, x, 4)
        fg, bg = self.ConvBlock4_fg._build(eng, d), self.ConvBlock4_bg._build(eng, d)
        fgs, bgs = [fg], [bg]
        for lvl, skip in ((3, skips[0]), (2, skips[1]), (1, skips[2])):
            d = getattr(self, f"eucb{lvl}")._build(eng, d)
            up_fg, up_bg = eng.resize_to(fg, d.H, d.W), eng.resize_to(bg, d.H, d.W)
            d = eng.add(d, getattr(self, f"lgag{lvl}")._build(eng, d, skip))
            d = self._stage(eng, d, lvl)
            fg, bg = getattr(self, f"ConvBlock{lvl}_fg")._build(eng, d), getattr(self, f"ConvBlock{lvl}_bg")._build(eng, d)
            fg = eng.dsra_fuse(fg, up_fg, up_bg, True)
            fgs.append(fg); bgs.append(bg)
        return fgs + bgs
