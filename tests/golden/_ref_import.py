"""Import helper for the golden-vector generator (runs ONLY in the build container).

Imports the reference implementation from /root/reference/binary_seg with import-surface
stubs for packages the image lacks (torchvision, timm, thop, imageio), exactly as SURVEY.md
§8(c) describes.  Nothing here travels to the GPU box: only the .npz vectors it helps to
produce do.
"""
import os, sys, types, importlib

REF = "/root/reference/binary_seg"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    import torch
    import torch.nn as nn
    sys.dont_write_bytecode = True
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    if "torchvision" not in sys.modules:
        tv = _stub("torchvision")
        tv.utils = _stub("torchvision.utils", save_image=lambda *a, **k: None)
        tv.transforms = _stub("torchvision.transforms")
    if "timm" not in sys.modules:
        class DropPath(nn.Module):
            def __init__(self, drop_prob=0.0):
                super().__init__(); self.drop_prob = drop_prob
            def forward(self, x):
                if self.drop_prob == 0.0 or not self.training:
                    return x
                keep = 1 - self.drop_prob
                mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
                return x * mask / keep
        def to_2tuple(x):
            return tuple(x) if isinstance(x, (tuple, list)) else (x, x)
        def trunc_normal_(t, std=1.0, **k):
            return nn.init.trunc_normal_(t, std=std, a=-2, b=2)
        timm = _stub("timm")
        timm.models = _stub("timm.models")
        _stub("timm.models.layers", DropPath=DropPath, to_2tuple=to_2tuple, trunc_normal_=trunc_normal_)
        _stub("timm.models.registry", register_model=lambda f: f)
        _stub("timm.models.vision_transformer", _cfg=lambda **k: {})
    _stub("thop", profile=None, clever_format=None)
    _stub("imageio")
    if REF not in sys.path:
        sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        pranet = importlib.import_module("lib.pranet")
        v1 = importlib.import_module("lib.PraNet_Res2Net")
        res2 = importlib.import_module("lib.Res2Net_v1b")
        utils = importlib.import_module("utils.utils")
        # reference constructors hard-load a checkpoint that is not in the container
        orig = res2.res2net50_v1b_26w_4s
        pranet.res2net50_v1b_26w_4s = lambda pretrained=False, **k: orig(pretrained=False, **k)
        v1.res2net50_v1b_26w_4s = lambda pretrained=False, **k: orig(pretrained=False, **k)
        # structure_loss lives in the training script, whose module body needs dataloader stubs
        _stub("utils.dataloader", get_loader=None, test_dataset=None)
        _stub("MyTest_med", test_with_eval=None)
        train = importlib.import_module("MyTrain_med")
    finally:
        os.chdir(cwd)
    return types.SimpleNamespace(pranet=pranet, v1=v1, res2=res2, utils=utils, train=train)
