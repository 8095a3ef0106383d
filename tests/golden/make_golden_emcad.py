"""Golden vectors for BASELINE config 5 (EMCADNet dual, K=9) from the imported reference — build container only.
Runs in its own process: multiclass_seg/EMCAD has its own `lib` package, which would collide with binary_seg's."""
import json, os, sys, types
import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = "/root/reference/multiclass_seg/EMCAD"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


class DropPath(nn.Module):
    def __init__(self, drop_prob=0.0):
        super().__init__(); self.drop_prob = drop_prob
    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        raise RuntimeError("golden vectors are generated with DropPath off")


def named_apply(fn, module, name="", depth_first=True, include_root=False):
    if not depth_first and include_root:
        fn(module=module, name=name)
    for cn, cm in module.named_children():
        named_apply(fn=fn, module=cm, name=".".join((name, cn)) if name else cn, depth_first=depth_first, include_root=True)
    if depth_first and include_root:
        fn(module=module, name=name)
    return module


tn = lambda t, std=1.0, **k: nn.init.trunc_normal_(t, std=std, a=-2, b=2)
_stub("timm"); _stub("timm.models")
_stub("timm.models.layers", DropPath=DropPath, to_2tuple=lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x), trunc_normal_=tn, trunc_normal_tf_=tn)
_stub("timm.models.helpers", named_apply=named_apply)
_stub("timm.models.registry", register_model=lambda f: f)
_stub("timm.models.vision_transformer", _cfg=lambda **k: {})
sys.path.insert(0, REF)
cwd = os.getcwd(); os.chdir(REF)
try:
    from lib.networks import EMCADNet
finally:
    os.chdir(cwd)
from oracle import weights as W


def npy(t):
    return t.detach().cpu().numpy()


PROBES = ["conv.0.weight", "backbone.patch_embed1.proj.weight", "backbone.block3.2.attn.kv.weight", "backbone.norm4.weight",
          "decoder.mscb4.0.pconv1.0.weight", "decoder.mscb4.0.msdc.dwconvs.0.0.weight", "decoder.mscb4.0.msdc.dwconvs.2.0.weight", "decoder.mscb4.0.pconv2.0.weight",
          "decoder.mscb4.0.msdc.dwconvs.1.1.weight", "decoder.eucb3.up_dwc.1.weight", "decoder.eucb3.pwc.0.weight", "decoder.eucb3.pwc.0.bias",
          "decoder.lgag3.W_g.0.weight", "decoder.lgag3.W_x.0.bias", "decoder.lgag3.psi.0.weight", "decoder.lgag3.psi.1.weight", "decoder.lgag1.W_x.0.weight",
          "decoder.cab4.fc1.weight", "decoder.cab2.fc2.weight", "decoder.sab.conv.weight", "decoder.mscb1.0.pconv2.1.bias",
          "decoder.ConvBlock4_fg.conv.weight", "decoder.ConvBlock3_bg.conv.weight", "decoder.ConvBlock1_fg.conv.weight", "decoder.ConvBlock1_fg.bn.weight"]
NP = 256


def build(dtype):
    m = EMCADNet(num_classes=9, kernel_sizes=[1, 3, 5], expansion_factor=2, dw_parallel=True, add=True, lgag_ks=3, activation="relu6",
                 encoder="pvt_v2_b2", pretrain=False, dual=True)
    m.backbone.reset_drop_path(0.0)
    return m.to(dtype)


def main(size=64, n=2):
    sys.path.insert(0, REF)
    # the reference's own subset enumeration and DiceLoss (utils/utils.py:20-30,102-138); its module body imports plotting / medical-IO
    # packages the image lacks, none of which these two use
    for missing in ("medpy", "seaborn", "segmentation_mask_overlay", "SimpleITK", "thop", "ptflops"):
        _stub(missing, metric=None, overlay_masks=None, profile=None, clever_format=None, get_model_complexity_info=None)
    cwd = os.getcwd(); os.chdir(REF)
    try:
        from utils.utils import powerset, DiceLoss
    finally:
        os.chdir(cwd)
    man = W.manifest_emcadnet(9)
    model = build(torch.float32)
    ref = {k: list(v.shape) for k, v in model.state_dict().items()}
    assert list(ref.items()) == [(k, list(v)) for k, v in man.items()], "manifest mismatch (EMCADNet)"
    json.dump({"emcadnet_dual_k9": ref, "n_params": sum(p.numel() for p in model.parameters())}, open(os.path.join(HERE, "manifest_emcad.json"), "w"))
    sd0 = W.make_state_dict(man, seed=5)
    model.load_state_dict(sd0, strict=True)
    model.train()
    g = torch.Generator().manual_seed(77)
    x = torch.randn(n, 1, size, size, generator=g)
    label = torch.randint(0, 9, (n, size, size), generator=g)
    # blocky labels (organ-like regions) instead of per-pixel noise
    label = torch.nn.functional.interpolate(label[:, None, ::8, ::8].float(), size=(size, size), mode="nearest")[:, 0].long()
    bg_mask = torch.stack([(label != k).float() for k in range(9)], 1)         # dataset_synapse.py: per-class background masks
    out = {"x": npy(x), "label": npy(label), "bg_mask": npy(bg_mask)}

    def run(m, xx, bgm):
        P = m(xx, mode="train")
        ce = nn.CrossEntropyLoss(); dl = DiceLoss(9); bce = nn.BCEWithLogitsLoss()
        loss = 0.0
        for s in powerset(list(range(4))):
            if s == []:
                continue
            iout = sum(P[i] for i in s); ibg = sum(P[4 + i] for i in s)
            loss = loss + 0.5 * ce(iout, label.long()) + 0.7 * dl(iout, label, softmax=True) + 0.3 * bce(ibg, bgm)
        loss.backward()
        return P, loss
    P, loss = run(model, x, bg_mask)
    names = dict(model.named_parameters())
    for i, o in enumerate(P):
        out[f"out{i}"] = npy(o)
    out["loss"] = npy(loss)
    for k in PROBES:
        out["graw." + k] = npy(names[k].grad.reshape(-1)[:NP]); out["grawnorm." + k] = npy(names[k].grad.norm())
    m64 = build(torch.float32); m64.load_state_dict(sd0, strict=True); m64 = m64.double().train()
    P64, l64 = run(m64, x.double(), bg_mask.double())
    n64 = dict(m64.named_parameters())
    for i, o in enumerate(P64):
        out[f"f64.out{i}"] = npy(o)
    out["f64.loss"] = npy(l64)
    for k in PROBES:
        out["f64.graw." + k] = npy(n64[k].grad.reshape(-1)[:NP]); out["f64.grawnorm." + k] = npy(n64[k].grad.norm())
    np.savez_compressed(os.path.join(HERE, "emcad_64.npz"), **out)
    print("wrote emcad_64.npz", len(out), "arrays; loss", float(loss))


if __name__ == "__main__":
    main()
