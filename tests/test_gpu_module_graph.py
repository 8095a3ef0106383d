"""The nn.Module surface replays a training call site from hipGraphs after its first calls (pn2/graph.py).  The literal loop of MyTrain_med.py:59-86 -
model(images) -> 4 x structure_loss -> loss.backward() -> clip_gradient -> torch.optim.Adam.step() - must give the same training run with and without them, and the
calls the graphs cannot serve (a second forward before the backward, eval mode, no_grad) must keep their eager semantics."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
os.environ.setdefault("PN2_NO_PRETRAINED", "1")
dev = "cuda"


@pytest.fixture(autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import pn2
    from pn2 import graph as G
    pn2.load_library()
    yield
    pn2.set_compute_dtype("bf16")
    G.set_module_graph(True)


def _model():
    from lib.pranet import PraNet_V2
    from oracle import weights as W
    model = PraNet_V2(num_class=1)
    model.load_state_dict(W.make_state_dict(W.manifest_pranet_v2(1), seed=0, bn3_gamma=0.05), strict=True)
    return model.to(dev).train()


def _run(steps, graphs, dtype="fp32"):
    import pn2
    from pn2 import graph as G
    from pn2.loss import structure_loss
    from utils.utils import clip_gradient
    from oracle import weights as W
    pn2.set_compute_dtype(dtype)
    G.set_module_graph(graphs)
    model = _model()
    opt = torch.optim.Adam(model.parameters(), 1e-4)
    losses = []
    for i in range(steps):
        x, m = W.synthetic_batch(2, 96, seed=100 + i)
        x, m = x.to(dev), m.to(dev)
        bg = 1 - m
        opt.zero_grad()
        o = model(x)
        loss = structure_loss(o[3], o[7], m, bg) + structure_loss(o[2], o[6], m, bg) + structure_loss(o[1], o[5], m, bg) + structure_loss(o[0], o[4], m, bg)
        loss.backward()
        clip_gradient(opt, 0.5)
        opt.step()
        losses.append(float(loss))
    torch.cuda.synchronize()
    sd = {k: v.detach().clone().double().cpu() for k, v in model.state_dict().items()}
    return losses, sd, model


def test_training_loop_with_graph_replayed_call_site_matches_eager():
    l0, sd0, _ = _run(7, False)
    l1, sd1, model = _run(7, True)
    sites = next(iter(model.hot_parameters())).__dict__["_pn2_sites"]
    st = next(iter(sites.values()))
    assert st.graph_f is not None and st.graph_b is not None and st.arena_steps == 2 and st.calls == 2      # calls 5, 6, 7 were replays
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 2e-5 * max(1.0, abs(a)), (l0, l1)
    for k in sd0:
        if sd0[k].numel() > 1:
            # Adam moves a weight by ~lr per step whatever the size of its gradient: where a gradient is ~0 the two runs (another fp32 summation order in the
            # table-driven wgrads) may step in opposite directions - a few steps of 1e-4 on single elements, nothing on the bulk
            d = (sd0[k] - sd1[k]).abs()
            assert float(d.max()) <= 3e-4 + 1e-6 * float(sd0[k].abs().max()), (k, float(d.max()))
            assert float(d.norm()) <= 2e-3 * float(sd0[k].norm()) + 1e-6, (k, float(d.norm()), float(sd0[k].norm()))


def test_outputs_and_gradients_of_a_replayed_call_match_the_plain_pass():
    """Same weights, same batch: call 5 of a site (a replay) against the plain pass of a fresh model."""
    import pn2
    from pn2 import graph as G
    from oracle import weights as W
    pn2.set_compute_dtype("fp32")
    x, _ = W.synthetic_batch(2, 96, seed=7)
    x = x.to(dev)
    gs = [torch.randn(2, 1, 96, 96, device=dev, generator=torch.Generator(device=dev).manual_seed(i)) for i in range(8)]

    def one(model):
        model.zero_grad()
        outs = model(x)
        torch.autograd.backward([o for o in outs], gs)
        return [o.detach().clone() for o in outs], {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    G.set_module_graph(False)
    ref_model = _model()
    G.set_module_graph(True)
    model = _model()
    # BatchNorm running statistics differ after 4 warm calls, the train-mode outputs do not depend on them
    for _ in range(4):
        one(model)
    o1, g1 = one(model)
    st = next(iter(next(iter(model.hot_parameters())).__dict__["_pn2_sites"].values()))
    assert st.graph_f is not None
    G.set_module_graph(False)
    o0, g0 = one(ref_model)
    for a, b in zip(o0, o1):
        assert float((a - b).abs().max()) < 2e-5
    assert set(g0) == set(g1)
    num = sum(float((g0[k].double() - g1[k].double()).pow(2).sum()) for k in g0) ** 0.5
    den = sum(float(g0[k].double().pow(2).sum()) for k in g0) ** 0.5
    assert num / den < 2e-4, num / den
    # the handed-out outputs are private copies: a later call does not overwrite them
    keep = o1[0].clone()
    outs = model(x * 0.5)
    assert torch.equal(o1[0], keep) and not torch.equal(outs[0], keep)
    del outs


def test_second_forward_before_backward_falls_back_to_the_plain_pass():
    import pn2
    from pn2 import graph as G
    from oracle import weights as W
    pn2.set_compute_dtype("fp32")
    G.set_module_graph(True)
    model = _model()
    xa, _ = W.synthetic_batch(2, 96, seed=1)
    xb, _ = W.synthetic_batch(2, 96, seed=2)
    xa, xb = xa.to(dev), xb.to(dev)
    for _ in range(5):                         # warm the site up to replay
        model.zero_grad()
        sum(o.sum() for o in model(xa)).backward()
    st = next(iter(next(iter(model.hot_parameters())).__dict__["_pn2_sites"].values()))
    assert st.graph_f is not None
    model.zero_grad()
    oa = model(xa)                              # replay
    ob = model(xb)                              # the site is busy: plain pass, both graphs stay valid
    assert st.busy()
    (sum(o.sum() for o in oa) + sum(o.sum() for o in ob)).backward()
    g_both = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    assert not st.busy()
    G.set_module_graph(False)
    model.zero_grad()
    (sum(o.sum() for o in model(xa)) + sum(o.sum() for o in model(xb))).backward()
    num = sum(float((g_both[n].double() - p.grad.double()).pow(2).sum()) for n, p in model.named_parameters() if p.grad is not None) ** 0.5
    den = sum(float(p.grad.double().pow(2).sum()) for n, p in model.named_parameters() if p.grad is not None) ** 0.5
    assert num / den < 5e-4, num / den          # (train-mode BatchNorm: the running statistics moved in between, the gradients do not depend on them)
    # a forward whose result is dropped without a backward does not block the site
    model(xa)
    assert not st.busy()
    # eval mode and no_grad calls never touch the site
    n_gen = st.gen
    with torch.no_grad():
        model(xa)
    model.eval()
    model(xa)
    model.train()
    assert st.gen == n_gen


def test_clip_gradient_multi_tensor_form_is_the_per_tensor_clamp():
    """utils.clip_gradient (reference utils/utils.py:7-17) clamps GPU gradients with two multi-tensor launches: same values as the reference's per-parameter loop."""
    from utils.utils import clip_gradient
    g = torch.Generator(device=dev).manual_seed(0)
    ps = [torch.nn.Parameter(torch.zeros(s, device=dev)) for s in ((3, 5), (7,), (2, 3, 4, 5), (1,))] + [torch.nn.Parameter(torch.zeros(4, device=dev))]
    for p in ps[:-1]:
        p.grad = torch.randn(p.shape, device=dev, generator=g) * 2
    want = [None if p.grad is None else p.grad.clone().clamp_(-0.5, 0.5) for p in ps]
    clip_gradient(torch.optim.SGD(ps, 0.1), 0.5)
    for p, w in zip(ps, want):
        assert (p.grad is None and w is None) or torch.equal(p.grad, w)


@pytest.mark.parametrize("kind", ["pvt", "emcad"])
def test_pvt_and_emcad_call_sites_replay_from_graphs(kind):
    """The PVTv2 / EMCAD mirrors go through the same call-site machinery (attention, LayerNorm, depth-wise ops, CAB's shared weights).  Seven calls on constant weights: the
    plain pass (calls 1, 2), the arena passes (3, 4) and the replays (5, 6, 7) must give the same outputs and - up to the fp32 summation order of the table-driven
    weight gradients - the same gradients.  (A training trajectory is no yardstick here: on these random-init nets with train-mode BatchNorm over 2 x 2 maps a 1e-8 gradient
    difference grows to 1e-3 in the loss within two SGD steps.)"""
    import pn2, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from pn2 import graph as G
    pn2.set_compute_dtype("fp32")
    G.set_module_graph(True)
    g = torch.Generator().manual_seed(11)
    if kind == "pvt":
        from test_gpu_pvt import _pvt_model as mk
        x = torch.randn(2, 3, 96, 96, generator=g).to(dev)
        fwd = lambda m, t: m(t)
    else:
        from test_gpu_emcad import _model as mk
        x = torch.randn(2, 1, 64, 64, generator=g).to(dev)
        fwd = lambda m, t: m(t, mode="train")
    model = mk(True)
    gs, res = None, []
    for _ in range(7):
        model.zero_grad()
        outs = fwd(model, x)
        outs = list(outs) if isinstance(outs, (tuple, list)) else [outs]
        if gs is None:
            gs = [torch.randn(o.shape, generator=torch.Generator().manual_seed(50 + k)).to(dev) for k, o in enumerate(outs)]
        torch.autograd.backward(outs, gs)
        res.append(([o.detach().clone() for o in outs], torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.grad is not None]).clone()))
    sites = [st for p in model.parameters() if "_pn2_sites" in p.__dict__ for st in p.__dict__["_pn2_sites"].values()]
    assert sites and any(st.graph_f is not None and st.gen >= 3 for st in sites), "no call site reached graph replay"
    for i in range(1, 7):
        for a, b in zip(res[0][0], res[i][0]):
            assert float((a - b).abs().max()) <= 1e-6 * max(1.0, float(a.abs().max())), i
        assert float((res[0][1] - res[i][1]).norm() / res[0][1].norm()) <= 1e-6, i


def test_droppath_draws_fresh_masks_under_graph_replay():
    """DropPath (pvtv2.py:125) keeps drawing new per-sample masks when the call site is replayed: torch's graph-safe generator advances with every replay."""
    import pn2, sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_pvt import _pvt_model
    from pn2 import graph as G
    pn2.set_compute_dtype("fp32")
    G.set_module_graph(True)
    model = _pvt_model(True)
    model.backbone.reset_drop_path(0.5)
    x = torch.randn(4, 3, 96, 96, generator=torch.Generator().manual_seed(3)).to(dev)
    outs = []
    for _ in range(8):
        model.zero_grad()
        o = model(x)
        sum(t.sum() for t in o).backward()
        outs.append(o[0].detach().clone())
    st = [st for p in model.parameters() if "_pn2_sites" in p.__dict__ for st in p.__dict__["_pn2_sites"].values()]
    assert st and st[0].graph_f is not None
    # calls 6, 7, 8 are replays of one graph on the same input and (nearly) the same weights: different DropPath masks -> different outputs
    assert not torch.equal(outs[5], outs[6]) and not torch.equal(outs[6], outs[7])


def test_bf16_replayed_call_matches_the_plain_pass():
    """The default (benchmarked) precision through the call-site graphs (the tests above run fp32 only): a replayed bf16 call against the plain bf16 pass
    of a fresh model on the same weights and batch - the same kernels on the same operands, so the maps agree to the last bit and the gradients up to the fp32
    summation order of the table-driven weight gradients."""
    import pn2
    from pn2 import graph as G
    from oracle import weights as W
    pn2.set_compute_dtype("bf16")
    x, _ = W.synthetic_batch(2, 96, seed=7)
    x = x.to(dev)
    gs = [torch.randn(2, 1, 96, 96, device=dev, generator=torch.Generator(device=dev).manual_seed(i)) for i in range(8)]

    def one(model):
        model.zero_grad()
        outs = model(x)
        torch.autograd.backward([o for o in outs], gs)
        return [o.detach().clone() for o in outs], {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    G.set_module_graph(True)
    model = _model()
    for _ in range(4):
        one(model)
    o1, g1 = one(model)
    st = next(iter(next(iter(model.hot_parameters())).__dict__["_pn2_sites"].values()))
    assert st.graph_f is not None and st.graph_b is not None
    G.set_module_graph(False)
    o0, g0 = one(_model())
    for a, b in zip(o0, o1):
        assert torch.equal(a, b)
    num = sum(float((g0[k].double() - g1[k].double()).pow(2).sum()) for k in g0) ** 0.5
    den = sum(float(g0[k].double().pow(2).sum()) for k in g0) ** 0.5
    assert set(g0) == set(g1) and num / den < 2e-4, num / den


def test_gradients_accumulate_across_replayed_calls_without_zero_grad():
    """Two forward / backward pairs WITHOUT zero_grad in between (gradient accumulation): the second replayed backward hands out views of a NEW flat copy, and
    autograd must add them to the .grad tensors adopted from the first one - p.grad == g(batch a) + g(batch b), not the second gradient alone."""
    import pn2
    from pn2 import graph as G
    from oracle import weights as W
    pn2.set_compute_dtype("fp32")
    G.set_module_graph(True)
    model = _model()
    xa, _ = W.synthetic_batch(2, 96, seed=21)
    xb, _ = W.synthetic_batch(2, 96, seed=22)
    xa, xb = xa.to(dev), xb.to(dev)
    for _ in range(5):                         # warm the site up to replay
        model.zero_grad()
        sum(o.sum() for o in model(xa)).backward()
    st = next(iter(next(iter(model.hot_parameters())).__dict__["_pn2_sites"].values()))
    assert st.graph_f is not None

    def grads():
        return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    model.zero_grad()
    sum(o.sum() for o in model(xa)).backward()
    ga = grads()
    model.zero_grad()
    sum(o.sum() for o in model(xb)).backward()
    gb = grads()
    model.zero_grad()
    sum(o.sum() for o in model(xa)).backward()
    held = {n: p.grad for n, p in model.named_parameters() if p.grad is not None}          # the tensors adopted as .grad by the first backward
    sum(o.sum() for o in model(xb)).backward()                                               # no zero_grad: accumulates
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        assert p.grad is held[n], n                                                          # accumulated in place
        want = ga[n] + gb[n]
        assert float((p.grad - want).abs().max()) <= 1e-6 * max(1.0, float(want.abs().max())), n


def test_clip_and_adam_as_one_launch_each_match_the_stock_optimizer(monkeypatch):
    """utils.clip_gradient on the module surface: the gradients handed out by a call site are views of one flat buffer and its parameters live in one flat arena
    (pn2/optim.py), so clip = one clamp launch and torch.optim.Adam.step = pn2_clamp_adam (same update formula).  The training run must be the stock run:
    same losses, same weights, a usable optimizer.state_dict(), lr changes (utils.adjust_lr) followed, and the stock step back in charge when the layout breaks."""
    from pn2 import optim as PO
    from utils.utils import adjust_lr
    monkeypatch.setattr(PO, "FUSED_OPT", False)
    l0, sd0, _ = _run(7, True)
    monkeypatch.setattr(PO, "FUSED_OPT", True)
    l1, sd1, model = _run(7, True)
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 2e-5 * max(1.0, abs(a)), (l0, l1)
    for k in sd0:
        if sd0[k].numel() > 1:
            d = (sd0[k] - sd1[k]).abs()
            assert float(d.max()) <= 3e-4 + 1e-6 * float(sd0[k].abs().max()), (k, float(d.max()))
            assert float(d.norm()) <= 2e-3 * float(sd0[k].norm()) + 1e-6, (k, float(d.norm()), float(sd0[k].norm()))
    # ---- one more run, looking inside
    import pn2
    from pn2.loss import structure_loss
    from utils.utils import clip_gradient
    from oracle import weights as W
    pn2.set_compute_dtype("fp32")
    model = _model()
    ref = _model()
    opt, opt_ref = torch.optim.Adam(model.parameters(), 1e-4), torch.optim.Adam(ref.parameters(), 1e-4)
    x, m = W.synthetic_batch(2, 96, seed=3)
    x, m = x.to(dev), m.to(dev)

    def step(model, opt, fused):
        monkeypatch.setattr(PO, "FUSED_OPT", fused)
        opt.zero_grad()
        o = model(x)
        loss = sum(structure_loss(o[i], o[i + 4], m, 1 - m) for i in range(4))
        loss.backward()
        clip_gradient(opt, 0.5)
        gmax = max(float(p.grad.abs().max()) for p in model.parameters() if p.grad is not None)
        opt.step()
        return gmax
    for i in range(5):
        if i == 4:
            adjust_lr(opt, 1e-4, 30, 0.1, 30); adjust_lr(opt_ref, 1e-4, 30, 0.1, 30)          # lr *= 0.1 on both
        assert step(model, opt, True) <= 0.5 and step(ref, opt_ref, False) <= 0.5
    f = opt._pn2_fused
    assert f is not None and f.used == 3 and getattr(opt_ref, "_pn2_fused", None) is None          # (calls 1-2 are plain passes: separate gradient tensors, stock step)
    hot = list(model.hot_parameters())
    assert all(f.fp.holds(p) for p in hot)
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        assert float((p - q).abs().max()) <= 2.2e-5 + 1e-6 * float(q.abs().max()), n           # (lr 1e-5 in the last step; a ~zero gradient may step the other way)
    s, s_ref = opt.state_dict()["state"], opt_ref.state_dict()["state"]
    assert set(s) == set(s_ref)
    for k in s:
        assert float(s[k]["step"]) == float(s_ref[k]["step"]) == 5.0
        assert float((s[k]["exp_avg"] - s_ref[k]["exp_avg"]).abs().max()) <= 1e-6 + 1e-4 * float(s_ref[k]["exp_avg"].abs().max())
    # a parameter is re-allocated behind the arena's back (p.data = ...): the next call of the model re-homes it (pn2.optim.flat_params), the run goes on and keeps
    # matching the stock optimizer; with the arena gone for good (PN2_FUSED_OPT off) the stock step takes over and every state entry has a step counter of its own again
    with torch.no_grad():
        hot[3].data = hot[3].data.clone()
    step(model, opt, True); step(ref, opt_ref, False)
    assert all(f.fp.holds(p) for p in hot) and f.used == 4
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        assert float((p - q).abs().max()) <= 3.3e-5 + 1e-6 * float(q.abs().max()), n
    step(model, opt, False); step(ref, opt_ref, False)
    st7 = opt.state_dict()["state"]
    assert getattr(opt, "_pn2_fused", None) is None and all(float(v["step"]) == 7.0 for v in st7.values()) and len({id(v["step"]) for v in opt.state.values()}) == len(opt.state)
    for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        assert float((p - q).abs().max()) <= 4.4e-5 + 1e-6 * float(q.abs().max()), n


def test_verbatim_torch_op_structure_loss_on_the_module_surface():
    """What an unedited MyTrain_med.py runs: its own torch-op structure_loss (restated in bench.torch_structure_loss) on the outputs of the mirror model - same loss
    and same parameter gradients as the fused loss kernels."""
    import sys
    import pn2
    from pn2.loss import structure_loss
    from oracle import weights as W
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    pn2.set_compute_dtype("fp32")
    x, m = W.synthetic_batch(2, 96, seed=11)
    x, m = x.to(dev), m.to(dev)
    res = []
    for fn in (bench.torch_structure_loss, structure_loss):
        model = _model()
        o = model(x)
        loss = sum(fn(o[i], o[i + 4], m, 1 - m) for i in range(4))
        loss.backward()
        res.append((float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
    assert abs(res[0][0] - res[1][0]) <= 1e-5 * abs(res[1][0])
    num = sum(float((res[0][1][k].double() - res[1][1][k].double()).pow(2).sum()) for k in res[1][1]) ** 0.5
    den = sum(float(res[1][1][k].double().pow(2).sum()) for k in res[1][1]) ** 0.5
    assert num / den < 1e-4, num / den
