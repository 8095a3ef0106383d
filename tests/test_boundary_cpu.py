"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/pn2.h declares,
the ctypes table matches the header, the nn.Module mirror has the reference's state_dict, and there is NO CPU fallback."""
import ctypes, json, os, re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("PN2_NO_PRETRAINED", "1")


def _header_functions():
    src = open(os.path.join(ROOT, "include", "pn2.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(pn2_\w+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from pn2 import capi
    assert os.path.exists(capi.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(capi.LIB_PATH)
    names = _header_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/pn2.h but not exported"
    assert sorted(capi.SIGNATURES) == names, "ctypes binding table and include/pn2.h disagree"


def test_header_argument_counts_match_binding():
    from pn2 import capi
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "pn2.h")).read(), flags=re.S)
    for name, args in capi.SIGNATURES.items():
        m = re.search(r"\bint\s+" + name + r"\s*\((.*?)\)\s*;", src, flags=re.S)
        assert m, name
        n = 0 if m.group(1).strip() in ("", "void") else m.group(1).count(",") + 1
        assert n == len(args), (name, n, len(args))


def test_value_helpers_run_without_gpu():
    from pn2.capi import call
    assert call.pn2_conv_tile_n(256) == 128 and call.pn2_conv_tile_n(32) == 32 and call.pn2_conv_tile_n(56) == 64
    assert call.pn2_wgrad_tile_co(32) == 32 and call.pn2_wgrad_tile_co(208) == 128
    assert call.pn2_conv_stat_blocks(129, 256, 1) == 3 and call.pn2_conv_stat_blocks(1 << 20, 256, 1) == 8192
    assert call.pn2_loss_blocks(352 * 352) == 31


def test_argument_errors_are_reported_not_swallowed():
    from pn2 import capi
    lib = capi.load()
    d = capi.ConvDesc()
    assert lib.pn2_conv_gemm(capi.BF16, None, None, None, None, None, ctypes.byref(d), None) == -1     # null pointers
    with pytest.raises(RuntimeError):
        capi.call.pn2_dsra_fuse_fwd(None, None, None, None, 1, 1, 1, None)
    # the streaming kernels index with 32 bits: element counts within one grid step of 2^32 are refused at the boundary (status -2), before any launch
    one = ctypes.c_void_p(16)
    assert lib.pn2_copy(capi.BF16, one, 8, capi.BF16, one, 8, 1 << 20, 4096, 0, None) == -2
    assert lib.pn2_binary(capi.BF16, 0, one, 8, one, 8, one, 8, 1 << 20, 4096, 0, None) == -2
    assert lib.pn2_nchw_to_nhwc(capi.BF16, one, one, 8, 1 << 16, 3, 1 << 16, 8, None) == -2
    assert lib.pn2_bilinear_fwd(capi.BF16, one, 8, one, 8, 64, 8, 8, 4096, 128, 128, 0, 1.0, 1.0, None) == -2


def test_state_dict_matches_reference_manifest():
    from lib.pranet import PraNet_V2
    from lib.PraNet_Res2Net import PraNet
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))
    m = PraNet_V2(num_class=1)
    assert [(k, list(v.shape)) for k, v in m.state_dict().items()] == [(k, v) for k, v in ref["pranet_v2_k1"].items()]
    assert sum(p.numel() for p in m.parameters()) == ref["n_params_v2"]
    assert len(m.hot_parameters()) == 478 - 6          # conv.0/1 and backbone.fc never receive gradients
    m1 = PraNet()
    assert [(k, list(v.shape)) for k, v in m1.state_dict().items()] == [(k, v) for k, v in ref["pranet_v1"].items()]
    # all four models MyTest_med.py:58-76 constructs load their checkpoints (V1 strict, :59,64)
    os.environ["PN2_NO_PRETRAINED"] = "1"
    from lib.PraNet_Res2Net import PVT_PraNet
    from oracle import weights as W
    refp = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest_pvt_v1.json")))
    mp = PVT_PraNet()
    assert [(k, list(v.shape)) for k, v in mp.state_dict().items()] == [(k, v) for k, v in refp["pvt_pranet"].items()]
    assert sum(p.numel() for p in mp.parameters()) == refp["n_params"]
    mp.load_state_dict(W.make_state_dict(W.manifest_pvt_pranet_v1(), seed=7), strict=True)
    # constructor signature / defaults of the reference (pranet.py:270)
    d = PraNet_V2()
    assert (d.num_class, d.sem_downsample, d.use_softmax) == (3, 1, True)
    # attribute access the reference forward relies on (pranet.py:331-341)
    for a in ("conv1", "bn1", "relu", "maxpool", "layer1", "layer2", "layer3", "layer4"):
        assert hasattr(m.backbone, a)


def test_reference_loads_fixture_weights_strict():
    from lib.pranet import PraNet_V2
    from oracle import weights as W
    m = PraNet_V2(num_class=1)
    m.load_state_dict(W.make_state_dict(W.manifest_pranet_v2(1), seed=0), strict=True)


def test_no_cpu_fallback():
    """The product path must fail loudly without a GPU instead of computing on the CPU."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from lib.pranet import PraNet_V2, BasicConv2d
    from pn2.loss import structure_loss
    with pytest.raises(RuntimeError):
        BasicConv2d(8, 8, 3, padding=1)(torch.randn(1, 8, 8, 8))
    with pytest.raises(RuntimeError):
        PraNet_V2(num_class=1)(torch.randn(1, 3, 64, 64))
    with pytest.raises(RuntimeError):
        structure_loss(torch.randn(1, 1, 8, 8), torch.randn(1, 1, 8, 8), torch.zeros(1, 1, 8, 8), torch.ones(1, 1, 8, 8))
    from lib.pranet import PVT_PraNet_V2
    os.environ["PN2_NO_PRETRAINED"] = "1"
    pvt = PVT_PraNet_V2(num_class=1)
    with pytest.raises(RuntimeError):
        pvt(torch.randn(1, 3, 64, 64))
    with pytest.raises(RuntimeError):
        pvt.backbone(torch.randn(1, 3, 64, 64))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "pranet-v2_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                s = open(os.path.join(dp, f)).read()
                assert "oracle" not in s.replace("no oracle", ""), f"{f} mentions the oracle"


def test_utils_match_reference_behaviour():
    from utils.utils import clip_gradient, adjust_lr, AvgMeter
    p = torch.nn.Parameter(torch.zeros(4)); p.grad = torch.tensor([-2.0, -0.1, 0.3, 9.0])
    opt = torch.optim.SGD([p], lr=1.0)
    clip_gradient(opt, 0.5)
    assert p.grad.tolist() == [-0.5, -0.10000000149011612, 0.30000001192092896, 0.5]
    adjust_lr(opt, 1.0, 60, 0.1, 50); adjust_lr(opt, 1.0, 61, 0.1, 50)
    assert abs(opt.param_groups[0]["lr"] - 0.01) < 1e-12      # compounding, as in the reference
    m = AvgMeter(num=2)
    for v in (1.0, 2.0, 4.0):
        m.update(torch.tensor(v))
    assert float(m.show()) == 3.0
