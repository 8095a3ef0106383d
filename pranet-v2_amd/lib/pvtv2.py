"""PVTv2 encoder (reference: lib/pvtv2.py) on the gfx950 kernels.

Same class names, constructor signatures, parameter names (state_dict keys) and init as the reference; the computation is
expressed in engine ops (pn2/engine.py + ops_*.py): the nn.Linear layers run as 1x1 implicit-GEMM convolutions on NHWC tokens, LayerNorm,
the depth-wise conv + GELU and the spatial-reduction attention are the kernels of csrc/pn2_vit.hip.  Tokens [B, N, C] of the
reference are NHWC pixels here, so the reshapes/permutes of the reference (:92-107, :191, :316-338, :370-372) cost nothing.
There is no PyTorch fallback: forward needs the GPU library.
"""
import math
from functools import partial

import torch
import torch.nn as nn

from pn2.graph import run_module


def trunc_normal_(t, std=1.0):
    return nn.init.trunc_normal_(t, std=std, a=-2.0, b=2.0)


def _init_weights(m):                       # pvtv2.py:27-40 (identical in every class of the file)
    if isinstance(m, nn.Linear):
        trunc_normal_(m.weight, std=.02)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.LayerNorm):
        nn.init.constant_(m.bias, 0)
        nn.init.constant_(m.weight, 1.0)
    elif isinstance(m, nn.Conv2d):
        fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
        fan_out //= m.groups
        m.weight.data.normal_(0, math.sqrt(2.0 / fan_out))
        if m.bias is not None:
            m.bias.data.zero_()


class DropPath(nn.Module):
    """Stochastic depth per sample (timm.models.layers.DropPath, used at pvtv2.py:125)."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob


class DWConv(nn.Module):
    def __init__(self, dim=768):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, 3, 1, 1, bias=True, groups=dim)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        assert act_layer is nn.GELU and drop == 0., "only the pvt_v2 configuration (GELU, no dropout) is built"
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.dwconv = DWConv(hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)
        self.apply(_init_weights)

    def _build(self, eng, x, residual=None):
        """fc2(gelu(dwconv(fc1(x)))) (+ residual)  — Mlp.forward :42-49"""
        t = eng.linear(x, self.fc1)
        t = eng.dwconv_gelu(t, self.dwconv.dwconv)
        return eng.linear(t, self.fc2, residual=residual)


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0., sr_ratio=1):
        super().__init__()
        assert dim % num_heads == 0, f"dim {dim} should be divided by num_heads {num_heads}."
        assert attn_drop == 0. and proj_drop == 0. and qk_scale is None, "only the pvt_v2 configuration (no dropout, default scale) is built"
        self.dim = dim
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.q = nn.Linear(dim, dim, bias=qkv_bias)
        self.kv = nn.Linear(dim, dim * 2, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.sr_ratio = sr_ratio
        if sr_ratio > 1:
            self.sr = nn.Conv2d(dim, dim, kernel_size=sr_ratio, stride=sr_ratio)
            self.norm = nn.LayerNorm(dim)
        self.apply(_init_weights)

    def _build(self, eng, x, residual=None):
        """Attention.forward :90-111 (+ residual add of Block.forward :148)"""
        q = eng.linear(x, self.q)
        src = x
        if self.sr_ratio > 1:
            src = eng.layernorm(eng.conv_bias(x, self.sr), self.norm)
        kv = eng.linear(src, self.kv)
        o = eng.attention(q, kv, self.num_heads)
        return eng.linear(o, self.proj, residual=residual)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm, sr_ratio=1):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop, sr_ratio=sr_ratio)
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = norm_layer(dim)
        mlp_hidden_dim = int(dim * mlp_ratio)
        self.mlp = Mlp(in_features=dim, hidden_features=mlp_hidden_dim, act_layer=act_layer, drop=drop)
        self.apply(_init_weights)

    def _build(self, eng, x):
        """Block.forward :147-151"""
        p = getattr(self.drop_path, "drop_prob", 0.0)
        if p > 0.0 and eng.training:
            x = eng.drop_path_add(x, self.attn._build(eng, eng.layernorm(x, self.norm1)), p)
            return eng.drop_path_add(x, self.mlp._build(eng, eng.layernorm(x, self.norm2)), p)
        x = self.attn._build(eng, eng.layernorm(x, self.norm1), residual=x)
        return self.mlp._build(eng, eng.layernorm(x, self.norm2), residual=x)


class OverlapPatchEmbed(nn.Module):
    """Image to Patch Embedding (:154-194): strided conv + LayerNorm over the embedding."""

    def __init__(self, img_size=224, patch_size=7, stride=4, in_chans=3, embed_dim=768):
        super().__init__()
        img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        patch_size = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.img_size = img_size
        self.patch_size = patch_size
        self.H, self.W = img_size[0] // patch_size[0], img_size[1] // patch_size[1]
        self.num_patches = self.H * self.W
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=stride, padding=(patch_size[0] // 2, patch_size[1] // 2))
        self.norm = nn.LayerNorm(embed_dim)
        self.apply(_init_weights)

    def _build(self, eng, x):
        return eng.layernorm(eng.conv_bias(x, self.proj), self.norm)


class PyramidVisionTransformerImpr(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dims=[64, 128, 256, 512],
                 num_heads=[1, 2, 4, 8], mlp_ratios=[4, 4, 4, 4], qkv_bias=False, qk_scale=None, drop_rate=0.,
                 attn_drop_rate=0., drop_path_rate=0., norm_layer=nn.LayerNorm, depths=[3, 4, 6, 3], sr_ratios=[8, 4, 2, 1]):
        super().__init__()
        self.num_classes = num_classes
        self.depths = depths
        self.patch_embed1 = OverlapPatchEmbed(img_size=img_size, patch_size=7, stride=4, in_chans=in_chans, embed_dim=embed_dims[0])
        self.patch_embed2 = OverlapPatchEmbed(img_size=img_size // 4, patch_size=3, stride=2, in_chans=embed_dims[0], embed_dim=embed_dims[1])
        self.patch_embed3 = OverlapPatchEmbed(img_size=img_size // 8, patch_size=3, stride=2, in_chans=embed_dims[1], embed_dim=embed_dims[2])
        self.patch_embed4 = OverlapPatchEmbed(img_size=img_size // 16, patch_size=3, stride=2, in_chans=embed_dims[2], embed_dim=embed_dims[3])
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(depths))]      # stochastic depth decay rule
        cur = 0
        for i in range(4):
            blocks = nn.ModuleList([Block(dim=embed_dims[i], num_heads=num_heads[i], mlp_ratio=mlp_ratios[i], qkv_bias=qkv_bias, qk_scale=qk_scale,
                                          drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[cur + j], norm_layer=norm_layer, sr_ratio=sr_ratios[i])
                                    for j in range(depths[i])])
            setattr(self, f"block{i + 1}", blocks)
            setattr(self, f"norm{i + 1}", norm_layer(embed_dims[i]))
            cur += depths[i]
        self.apply(_init_weights)

    def reset_drop_path(self, drop_path_rate):
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, sum(self.depths))]
        cur = 0
        for i in range(4):
            for j, blk in enumerate(getattr(self, f"block{i + 1}")):
                if isinstance(blk.drop_path, DropPath):
                    blk.drop_path.drop_prob = dpr[cur + j]
                elif dpr[cur + j] > 0:
                    blk.drop_path = DropPath(dpr[cur + j])
            cur += self.depths[i]

    def freeze_patch_emb(self):
        self.patch_embed1.requires_grad = False

    def _build_features(self, eng, x):
        """forward_features :307-341: four NHWC feature maps (64, 128, 320, 512 channels for b2)."""
        outs = []
        if eng.training:          # all DropPath draws of the forward at once (two per block with a non-zero rate, in block order)
            eng.drop_path_plan([p for i in range(4) for blk in getattr(self, f"block{i + 1}") for p in 2 * (getattr(blk.drop_path, "drop_prob", 0.0),)], x.N)
        for i in range(4):
            x = getattr(self, f"patch_embed{i + 1}")._build(eng, x)
            for blk in getattr(self, f"block{i + 1}"):
                x = blk._build(eng, x)
            x = eng.layernorm(x, getattr(self, f"norm{i + 1}"))
            outs.append(x)
        return outs

    def forward(self, x):
        return run_module(lambda e, a: self._build_features(e, a), [x], list(self.parameters()), self.training)


def _variant(name, embed_dims, num_heads, mlp_ratios, depths, sr_ratios=(8, 4, 2, 1)):
    class _V(PyramidVisionTransformerImpr):
        def __init__(self, **kwargs):
            super().__init__(patch_size=4, embed_dims=list(embed_dims), num_heads=list(num_heads), mlp_ratios=list(mlp_ratios), qkv_bias=True,
                             norm_layer=partial(nn.LayerNorm, eps=1e-6), depths=list(depths), sr_ratios=list(sr_ratios), drop_rate=0.0, drop_path_rate=0.1)
    _V.__name__ = _V.__qualname__ = name
    return _V


# pvtv2.py:378-436.  The attention kernel is built for head_dim 64, which b1..b5 use; b0 (head_dim 32) constructs but cannot run.
pvt_v2_b0 = _variant("pvt_v2_b0", (32, 64, 160, 256), (1, 2, 5, 8), (8, 8, 4, 4), (2, 2, 2, 2))
pvt_v2_b1 = _variant("pvt_v2_b1", (64, 128, 320, 512), (1, 2, 5, 8), (8, 8, 4, 4), (2, 2, 2, 2))
pvt_v2_b2 = _variant("pvt_v2_b2", (64, 128, 320, 512), (1, 2, 5, 8), (8, 8, 4, 4), (3, 4, 6, 3))
pvt_v2_b3 = _variant("pvt_v2_b3", (64, 128, 320, 512), (1, 2, 5, 8), (8, 8, 4, 4), (3, 4, 18, 3))
pvt_v2_b4 = _variant("pvt_v2_b4", (64, 128, 320, 512), (1, 2, 5, 8), (8, 8, 4, 4), (3, 8, 27, 3))
pvt_v2_b5 = _variant("pvt_v2_b5", (64, 128, 320, 512), (1, 2, 5, 8), (4, 4, 4, 4), (3, 6, 40, 3))
