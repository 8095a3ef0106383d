"""Every remaining behaviour switch of the engine / trainer runs a training step: a 2 x 96^2 PraNet-V2 step on the fp32 path with the
switch flipped must reproduce the default step - same losses, same gradients up to the fp32 rounding of a different summation order.  A switch is either
an environment variable read when the object is built or a module constant read per call; both are flipped for one step here."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
os.environ.setdefault("PN2_NO_PRETRAINED", "1")
dev = "cuda"

# (where, name, value): "engine" (pn2/core.py) / "res2net" = module constant, "env" = environment variable read by Trainer / run_module at construction
SWITCHES = [
    ("engine", "BNB_EPILOGUE", False), ("engine", "MASKED_STORE", False), ("engine", "LOCKSTEP", False), ("engine", "GRAD_ALIAS", False),
    ("engine", "SPLITK", False), ("engine", "FUSE_BIAS", False), ("engine", "DEFER_COLSUM", False), ("engine", "ZERO_CROP_SKIP", False),
    ("engine", "PATCH_DGRAD", False), ("engine", "SMALL_CIN_DGRAD", False), ("engine", "WGRAD_SLAB_CAP", 0.0), ("engine", "WGRAD_ROTATE", False), ("engine", "TEE_CONCAT", False), ("engine", "POOL_FUSE", False), ("engine", "POOL_BWD_QUAD", False), ("engine", "POOL_FOLD", False), ("engine", "RES_STATS", False), ("engine", "MUL_BWD", False), ("engine", "WGRAD_MIX", 0.0),
    ("res2net", "ALIAS_CAT_GRAD", False), ("lockstep", "COALESCE", False), ("lockstep", "MIXED", False),
    ("env", "PN2_FUSED_TAIL", "0"), ("env", "PN2_DEFER_WGRAD", "1"), ("env", "PN2_DEFER_WGRAD", "0"), ("env", "PN2_STEP_ARENA", "0"), ("env", "PN2_AUTOTUNE", "0"),
]


@pytest.fixture(autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import pn2
    pn2.load_library()
    yield
    pn2.set_compute_dtype("bf16")


def _step():
    import pn2
    from pn2.trainer import Trainer
    from lib.pranet import PraNet_V2
    from oracle import weights as W
    pn2.set_compute_dtype("fp32")
    model = PraNet_V2(num_class=1)
    model.load_state_dict(W.make_state_dict(W.manifest_pranet_v2(1), seed=0, bn3_gamma=0.05), strict=True)
    model = model.to(dev).train()
    x, m = W.synthetic_batch(2, 96, seed=1234)
    tr = Trainer(model, lr=1e-4, clip=0.5)
    loss = tr.forward_backward(x.to(dev), m.to(dev))
    torch.cuda.synchronize()
    return loss.clone().cpu().double(), tr.gflat.clone().cpu().double(), tr.last_outs.clone().cpu().double()


_BASE = {}


@pytest.mark.parametrize("where,name,value", SWITCHES, ids=[f"{n}={v}" for _w, n, v in SWITCHES])
def test_step_under_switch_matches_default(where, name, value, monkeypatch):
    if "base" not in _BASE:
        _BASE["base"] = _step()
    l0, g0, o0 = _BASE["base"]
    if where == "env":
        monkeypatch.setenv(name, value)
    else:
        import importlib
        mod = importlib.import_module({"engine": "pn2.core", "res2net": "lib.Res2Net_v1b", "lockstep": "pn2.lockstep"}[where])
        assert hasattr(mod, name), name
        monkeypatch.setattr(mod, name, value)
    l1, g1, o1 = _step()
    assert float((l1 - l0).abs().max()) < 1e-5, (name, l1, l0)
    assert float((o1 - o0).abs().max()) < 2e-5                      # forward maps
    rel = float((g1 - g0).norm() / g0.norm())
    assert rel < 2e-4, (name, rel)                                   # same gradients up to another fp32 summation order (conditioned weights: no chaos)


def test_pool_backward_folded_into_the_dgrad_epilogue_is_bit_identical_bf16(monkeypatch):
    """PN2_POOL_FOLD (bf16, the benchmarked precision): AvgPool2d(2, 2)'s backward inside conv1's dgrad epilogue (pn2_conv_ep.pool; layer2.0 of the Res2Net encoder, where nothing
    else has written the block input's gradient) leaves the bits the separate pool-backward launch + `+=` leave: 1/4 is exact, the sum is rounded once either way."""
    import pn2
    from pn2 import core
    from pn2.trainer import Trainer
    from lib.pranet import PraNet_V2
    from oracle import weights as W

    def step(fold):
        monkeypatch.setattr(core, "POOL_FOLD", fold)
        pn2.set_compute_dtype("bf16")
        model = PraNet_V2(num_class=1)
        model.load_state_dict(W.make_state_dict(W.manifest_pranet_v2(1), seed=0, bn3_gamma=0.05), strict=True)
        model = model.to(dev).train()
        x, m = W.synthetic_batch(2, 96, seed=1234)
        tr = Trainer(model, lr=1e-4, clip=0.5)
        calls = []
        from pn2.capi import call
        real = call.pn2_avgpool_bwd
        monkeypatch.setattr(call, "pn2_avgpool_bwd", lambda *a: (calls.append(a[5:10]), real(*a))[1])
        loss = tr.forward_backward(x.to(dev), m.to(dev))
        torch.cuda.synchronize()
        monkeypatch.setattr(call, "pn2_avgpool_bwd", real)
        return loss.clone(), tr.gflat.clone(), len(calls)
    l0, g0, n0 = step(False)
    l1, g1, n1 = step(True)
    assert n1 == n0 - 1, (n0, n1)          # one pool-backward launch fewer (layer3.0 / layer4.0: their block inputs x2 / x3 already carry the heads' gradient -> plain launch)
    assert torch.equal(l0, l1) and torch.equal(g0, g1)
