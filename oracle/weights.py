"""TEST INFRASTRUCTURE — not part of the product path.

Deterministic state_dict builder shared by the golden-vector generator
(tests/golden/make_golden.py, run in the build container next to the imported
reference) and by the parity tests (run on the GPU box, where the reference does
not exist).  The full PraNet-V2 state_dict is 130 MB, far too big to commit, so
fixtures carry only (seed, key->shape manifest, expected outputs) and both sides
regenerate the identical weights from this file.

Key names / shapes restate what the reference classes register:
  PraNet_V2            /root/reference/binary_seg/lib/pranet.py:268-325
  RFB_modified         pranet.py:46-74      aggregation  pranet.py:86-104
  Res2Net / Bottle2neck /root/reference/binary_seg/lib/Res2Net_v1b.py:15-56,95-146
  PraNet (V1)          /root/reference/binary_seg/lib/PraNet_Res2Net.py:101-128
"""
import math
import re
from collections import OrderedDict

import torch


def _conv(m, name, cout, cin, kh, kw, bias=False):
    m[name + ".weight"] = (cout, cin, kh, kw)
    if bias:
        m[name + ".bias"] = (cout,)


def _bn(m, name, c):
    m[name + ".weight"] = (c,)
    m[name + ".bias"] = (c,)
    m[name + ".running_mean"] = (c,)
    m[name + ".running_var"] = (c,)
    m[name + ".num_batches_tracked"] = ()


def _basic(m, name, cin, cout, k):
    kh, kw = (k, k) if isinstance(k, int) else k
    _conv(m, name + ".conv", cout, cin, kh, kw)
    _bn(m, name + ".bn", cout)


def _res2net(m, p, layers=(3, 4, 6, 3), base_width=26, scale=4):
    _conv(m, p + "conv1.0", 32, 3, 3, 3); _bn(m, p + "conv1.1", 32)
    _conv(m, p + "conv1.3", 32, 32, 3, 3); _bn(m, p + "conv1.4", 32)
    _conv(m, p + "conv1.6", 64, 32, 3, 3)
    _bn(m, p + "bn1", 64)
    inplanes = 64
    for li, (planes, nblk) in enumerate(zip((64, 128, 256, 512), layers)):
        stride = 1 if li == 0 else 2
        for b in range(nblk):
            q = f"{p}layer{li + 1}.{b}."
            width = int(math.floor(planes * (base_width / 64.0)))
            _conv(m, q + "conv1", width * scale, inplanes, 1, 1); _bn(m, q + "bn1", width * scale)
            for i in range(scale - 1):
                _conv(m, q + f"convs.{i}", width, width, 3, 3)
            for i in range(scale - 1):
                _bn(m, q + f"bns.{i}", width)
            _conv(m, q + "conv3", planes * 4, width * scale, 1, 1); _bn(m, q + "bn3", planes * 4)
            if b == 0 and (stride != 1 or inplanes != planes * 4):
                _conv(m, q + "downsample.1", planes * 4, inplanes, 1, 1); _bn(m, q + "downsample.2", planes * 4)
            inplanes = planes * 4
    m[p + "fc.weight"] = (1000, 2048)
    m[p + "fc.bias"] = (1000,)


def _rfb(m, p, cin, c):
    _basic(m, p + "branch0.0", cin, c, 1)
    for bi, k in ((1, 3), (2, 5), (3, 7)):
        q = f"{p}branch{bi}."
        _basic(m, q + "0", cin, c, 1)
        _basic(m, q + "1", c, c, (1, k))
        _basic(m, q + "2", c, c, (k, 1))
        _basic(m, q + "3", c, c, 3)
    _basic(m, p + "conv_cat", 4 * c, c, 3)
    _basic(m, p + "conv_res", cin, c, 1)


def _agg(m, p, c, num_class=None):
    for i in (1, 2, 3, 4):
        _basic(m, p + f"conv_upsample{i}", c, c, 3)
    _basic(m, p + "conv_upsample5", 2 * c, 2 * c, 3)
    _basic(m, p + "conv_concat2", 2 * c, 2 * c, 3)
    _basic(m, p + "conv_concat3", 3 * c, 3 * c, 3)
    _basic(m, p + "conv4", 3 * c, 3 * c, 3)
    if num_class is None:  # V1: single head
        _conv(m, p + "conv5", 1, 3 * c, 1, 1, bias=True)
    else:
        _conv(m, p + "conv5_fg", num_class, 3 * c, 1, 1, bias=True)
        _conv(m, p + "conv5_bg", num_class, 3 * c, 1, 1, bias=True)


def manifest_pranet_v2(num_class=1, channel=32):
    """key -> shape, in the order the reference's state_dict() yields them."""
    m = OrderedDict()
    _conv(m, "conv.0", 3, 1, 1, 1, bias=True); _bn(m, "conv.1", 3)
    _res2net(m, "backbone.")
    _rfb(m, "rfb2_1.", 512, channel); _rfb(m, "rfb3_1.", 1024, channel); _rfb(m, "rfb4_1.", 2048, channel)
    _agg(m, "agg1.", channel, num_class)
    _basic(m, "ra4_conv1", 2048, 256, 1)
    for i in (2, 3, 4):
        _basic(m, f"ra4_conv{i}", 256, 256, 5)
    _basic(m, "ra4_conv5_fg", 256, num_class, 1); _basic(m, "ra4_conv5_bg", 256, num_class, 1)
    for s, cin in ((3, 1024), (2, 512)):
        _basic(m, f"ra{s}_conv1", cin, 64, 1)
        _basic(m, f"ra{s}_conv2", 64, 64, 3); _basic(m, f"ra{s}_conv3", 64, 64, 3)
        _basic(m, f"ra{s}_conv4_fg", 64, num_class, 3); _basic(m, f"ra{s}_conv4_bg", 64, num_class, 3)
    return m


def manifest_pranet_v1(channel=32):
    m = OrderedDict()
    _res2net(m, "resnet.")
    _rfb(m, "rfb2_1.", 512, channel); _rfb(m, "rfb3_1.", 1024, channel); _rfb(m, "rfb4_1.", 2048, channel)
    _agg(m, "agg1.", channel, None)
    _basic(m, "ra4_conv1", 2048, 256, 1)
    for i in (2, 3, 4):
        _basic(m, f"ra4_conv{i}", 256, 256, 5)
    _basic(m, "ra4_conv5", 256, 1, 1)
    for s, cin in ((3, 1024), (2, 512)):
        _basic(m, f"ra{s}_conv1", cin, 64, 1)
        _basic(m, f"ra{s}_conv2", 64, 64, 3); _basic(m, f"ra{s}_conv3", 64, 64, 3)
        _basic(m, f"ra{s}_conv4", 64, 1, 3)
    return m


def _ln(m, name, c):
    m[name + ".weight"] = (c,)
    m[name + ".bias"] = (c,)


def _linear(m, name, cout, cin):
    m[name + ".weight"] = (cout, cin)
    m[name + ".bias"] = (cout,)


PVT_B2 = dict(embed_dims=(64, 128, 320, 512), num_heads=(1, 2, 5, 8), mlp_ratios=(8, 8, 4, 4), depths=(3, 4, 6, 3), sr_ratios=(8, 4, 2, 1))


def _pvt_v2(m, p, cfg=PVT_B2, in_chans=3):
    """PyramidVisionTransformerImpr registration order (lib/pvtv2.py:197-243; Block :114-130, Attention :52-72, Mlp :14-23,
    OverlapPatchEmbed :154-170, DWConv :363-366)."""
    dims = cfg["embed_dims"]
    for i in range(4):
        k = 7 if i == 0 else 3
        _conv(m, f"{p}patch_embed{i + 1}.proj", dims[i], in_chans if i == 0 else dims[i - 1], k, k, bias=True)
        _ln(m, f"{p}patch_embed{i + 1}.norm", dims[i])
    for i in range(4):
        d, hid, sr = dims[i], dims[i] * cfg["mlp_ratios"][i], cfg["sr_ratios"][i]
        for j in range(cfg["depths"][i]):
            q = f"{p}block{i + 1}.{j}."
            _ln(m, q + "norm1", d)
            _linear(m, q + "attn.q", d, d); _linear(m, q + "attn.kv", 2 * d, d); _linear(m, q + "attn.proj", d, d)
            if sr > 1:
                _conv(m, q + "attn.sr", d, d, sr, sr, bias=True)
                _ln(m, q + "attn.norm", d)
            _ln(m, q + "norm2", d)
            _linear(m, q + "mlp.fc1", hid, d)
            _conv(m, q + "mlp.dwconv.dwconv", hid, 1, 3, 3, bias=True)
            _linear(m, q + "mlp.fc2", d, hid)
        _ln(m, f"{p}norm{i + 1}", d)


def manifest_pvt_pranet_v2(num_class=1, channel=32):
    """PVT_PraNet_V2 (lib/pranet.py:129-203): key -> shape in state_dict() order."""
    m = OrderedDict()
    _conv(m, "conv.0", 3, 1, 1, 1, bias=True); _bn(m, "conv.1", 3)
    _pvt_v2(m, "backbone.")
    _rfb(m, "rfb2_1.", 128, channel); _rfb(m, "rfb3_1.", 320, channel); _rfb(m, "rfb4_1.", 512, channel)
    _agg(m, "agg1.", channel, num_class)
    _basic(m, "ra4_conv1", 512, 256, 1)
    for i in (2, 3, 4):
        _basic(m, f"ra4_conv{i}", 256, 256, 5)
    _basic(m, "ra4_conv5_fg", 256, num_class, 1); _basic(m, "ra4_conv5_bg", 256, num_class, 1)
    for s, cin in ((3, 320), (2, 128)):
        _basic(m, f"ra{s}_conv1", cin, 64, 1)
        _basic(m, f"ra{s}_conv2", 64, 64, 3); _basic(m, f"ra{s}_conv3", 64, 64, 3)
        _basic(m, f"ra{s}_conv4_fg", 64, num_class, 3); _basic(m, f"ra{s}_conv4_bg", 64, num_class, 3)
    return m


def _emcad_decoder(m, p, channels=(512, 320, 128, 64), kernel_sizes=(1, 3, 5), expansion=2, num_class=9):
    """EMCAD_dual registration order (multiclass_seg/EMCAD/lib/decoders.py:407-440; MSCB :104-140, MSDC :83-99, EUCB :166-181, LGAG :190-207,
    CAB :216-232, SAB :244-251, BasicConv2d :14-22)."""
    def mscb(q, c):
        ex = c * expansion
        _conv(m, q + "0.pconv1.0", ex, c, 1, 1); _bn(m, q + "0.pconv1.1", ex)
        for i, k in enumerate(kernel_sizes):
            _conv(m, q + f"0.msdc.dwconvs.{i}.0", ex, 1, k, k); _bn(m, q + f"0.msdc.dwconvs.{i}.1", ex)
        _conv(m, q + "0.pconv2.0", c, ex, 1, 1); _bn(m, q + "0.pconv2.1", c)

    def eucb(q, cin, cout):
        _conv(m, q + "up_dwc.1", cin, 1, 3, 3); _bn(m, q + "up_dwc.2", cin)
        _conv(m, q + "pwc.0", cout, cin, 1, 1, bias=True)

    def lgag(q, c):
        fi = c // 2
        for w in ("W_g", "W_x"):
            _conv(m, q + w + ".0", fi, 2, 3, 3, bias=True); _bn(m, q + w + ".1", fi)       # groups = F_int: 2 input channels per group
        _conv(m, q + "psi.0", 1, fi, 1, 1, bias=True); _bn(m, q + "psi.1", 1)
    c = channels
    mscb(p + "mscb4.", c[0])
    for i, lvl in ((1, 3), (2, 2), (3, 1)):
        eucb(p + f"eucb{lvl}.", c[i - 1], c[i]); lgag(p + f"lgag{lvl}.", c[i]); mscb(p + f"mscb{lvl}.", c[i])
    for i, lvl in enumerate((4, 3, 2, 1)):
        r = c[i] // 16 if c[i] >= 16 else 1
        m[p + f"cab{lvl}.fc1.weight"] = (r, c[i], 1, 1); m[p + f"cab{lvl}.fc2.weight"] = (c[i], r, 1, 1)
    m[p + "sab.conv.weight"] = (1, 2, 7, 7)
    for i, lvl in enumerate((4, 3, 2, 1)):
        k = 1 if lvl == 4 else 3
        _basic(m, p + f"ConvBlock{lvl}_fg", c[i], num_class, k); _basic(m, p + f"ConvBlock{lvl}_bg", c[i], num_class, k)


def manifest_pvt_pranet_v1(channel=32):
    """PVT_PraNet (lib/PraNet_Res2Net.py:188-224): key -> shape in state_dict() order."""
    m = OrderedDict()
    _pvt_v2(m, "backbone.")
    _rfb(m, "rfb2_1.", 128, channel); _rfb(m, "rfb3_1.", 320, channel); _rfb(m, "rfb4_1.", 512, channel)
    _agg(m, "agg1.", channel, None)
    _basic(m, "ra4_conv1", 512, 256, 1)
    for i in (2, 3, 4):
        _basic(m, f"ra4_conv{i}", 256, 256, 5)
    _basic(m, "ra4_conv5", 256, 1, 1)
    for s, cin in ((3, 320), (2, 128)):
        _basic(m, f"ra{s}_conv1", cin, 64, 1)
        _basic(m, f"ra{s}_conv2", 64, 64, 3); _basic(m, f"ra{s}_conv3", 64, 64, 3)
        _basic(m, f"ra{s}_conv4", 64, 1, 3)
    return m


def manifest_emcadnet(num_classes=9):
    """EMCADNet(dual=True, encoder='pvt_v2_b2') (multiclass_seg/EMCAD/lib/networks.py:10-98): key -> shape in state_dict() order."""
    m = OrderedDict()
    _conv(m, "conv.0", 3, 1, 1, 1, bias=True); _bn(m, "conv.1", 3)
    _pvt_v2(m, "backbone.")
    _emcad_decoder(m, "decoder.", num_class=num_classes)
    for i, cch in zip((4, 3, 2, 1), (512, 320, 128, 64)):
        _conv(m, f"out_head{i}", num_classes, cch, 1, 1, bias=True)
    return m


_BN3 = re.compile(r"layer\d\.\d+\.bn3\.weight$")


def make_state_dict(manifest, seed=0, bn3_gamma=None):
    """Deterministic non-trivial weights: every tensor from its own CPU generator.

    conv/linear: N(0, sqrt(2/fan_in))·0.9 ; BN gamma U(0.6,1.4), beta N(0,0.1),
    running_mean N(0,0.1), running_var U(0.6,1.4).  Values are chosen so that
    activations stay O(1) through ~60 layers in both train and eval mode.

    bn3_gamma (the CONDITIONED fixtures, tests/golden/make_golden_cond.py): the gamma of the last BatchNorm of every
    Bottle2neck (`layerL.B.bn3.weight`, Res2Net_v1b.py:50,84) is multiplied by this factor.  A random-init BN-ReLU
    residual network is chaotic - each block multiplies a perturbation by ~1.2, so fp32 rounding noise reaches 1e-3 on
    O(1) logits after 16 blocks, in eval mode with calibrated statistics just as in train mode (measured on the imported
    reference).  Trained checkpoints do not behave like that: their residual branches are small corrections (the usual
    zero-init-residual regime).  With bn3 gamma x 0.05 the reference's own fp32 run agrees with its float64 run to
    ~2e-5 on the logits, which is what lets a test gate north_star's literal "1e-4 abs on fp32 logits".
    """
    sd = OrderedDict()
    for idx, (k, shape) in enumerate(manifest.items()):
        g = torch.Generator(device="cpu").manual_seed(seed * 100003 + idx)
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.long)
        elif k.endswith("running_var"):
            sd[k] = torch.rand(shape, generator=g) * 0.8 + 0.6
        elif k.endswith("running_mean"):
            sd[k] = torch.randn(shape, generator=g) * 0.1
        elif len(shape) == 1 and k.endswith(".weight"):      # BN gamma
            sd[k] = torch.rand(shape, generator=g) * 0.8 + 0.6
            if bn3_gamma is not None and _BN3.search(k):
                sd[k] = sd[k] * bn3_gamma
        elif len(shape) == 1:                                  # biases (BN beta, conv/fc bias)
            sd[k] = torch.randn(shape, generator=g) * 0.1
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            sd[k] = torch.randn(shape, generator=g) * (0.9 * math.sqrt(2.0 / fan_in))
    return sd


def synthetic_batch(n, size, seed=1234):
    """Images N(0,1) and polyp-like masks (1-3 filled ellipses, float {0,1}) — SURVEY §8(d)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn((n, 3, size, size), generator=g)
    yy, xx = torch.meshgrid(torch.arange(size, dtype=torch.float32), torch.arange(size, dtype=torch.float32), indexing="ij")
    mask = torch.zeros((n, 1, size, size))
    for i in range(n):
        k = int(torch.randint(1, 4, (1,), generator=g))
        for _ in range(k):
            cy, cx = (torch.rand(2, generator=g) * 0.6 + 0.2) * size
            ry, rx = (torch.rand(2, generator=g) * 0.16 + 0.08) * size
            mask[i, 0][((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0] = 1.0
    return x, mask
