"""Pins oracle/ (the CPU restatement) against vectors produced by the imported reference
(tests/golden/make_golden.py).  CPU only; no HIP code involved."""
import json, os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import pranet_oracle as O
from oracle import weights as W

torch.set_num_threads(8)
G = os.path.join(os.path.dirname(__file__), "golden")
TOL = 2e-5   # oracle vs reference: same torch CPU kernels, different op order in places


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_manifest_matches_reference():
    ref = json.load(open(os.path.join(G, "manifest.json")))
    ours = W.manifest_pranet_v2(1)
    assert [(k, list(v)) for k, v in ours.items()] == [(k, v) for k, v in ref["pranet_v2_k1"].items()]
    assert len(ours) == 949
    ours1 = W.manifest_pranet_v1()
    assert [(k, list(v)) for k, v in ours1.items()] == [(k, v) for k, v in ref["pranet_v1"].items()]
    sd = W.make_state_dict(ours)
    nparam = sum(v.numel() for k, v in sd.items() if k in O.params_of(sd))
    assert nparam == ref["n_params_v2"] == 32548842


@pytest.mark.parametrize("tag", ["rand", "zeros", "ones"])
def test_structure_loss(tag):
    z = np.load(os.path.join(G, "structure_loss.npz"))
    pred = T(z[f"{tag}_pred"]).requires_grad_(True)
    pred_bg = T(z[f"{tag}_pred_bg"]).requires_grad_(True)
    mask = T(z[f"{tag}_mask"])
    loss = O.structure_loss(pred, pred_bg, mask, 1 - mask)
    loss.backward()
    assert abs(float(loss) - float(z[f"{tag}_loss"])) < 1e-6
    assert (pred.grad - T(z[f"{tag}_gpred"])).abs().max() < 1e-7
    assert (pred_bg.grad - T(z[f"{tag}_gpred_bg"])).abs().max() < 1e-7


@pytest.mark.parametrize("sm", [True, False])
def test_dsra_k9(sm):
    z = np.load(os.path.join(G, "dsra_k9.npz"))
    fg, cf, cb = (T(z[k]).requires_grad_(True) for k in ("fg", "crop_fg", "crop_bg"))
    y = O.dsra_fuse(fg, cf, cb, sm)
    y.backward(T(z["gout"]))
    tag = "sm" if sm else "nosm"
    assert (y - T(z[tag + "_y"])).abs().max() < 1e-6
    for t, k in ((fg, "gfg"), (cf, "gcf"), (cb, "gcb")):
        assert (t.grad - T(z[f"{tag}_{k}"])).abs().max() < 1e-6


def _block_sd(z, prefix):
    sd = OrderedDict()
    for k in z.files:
        if k.startswith(prefix):
            name = k[len(prefix):]
            v = T(z[k]).clone()
            if name.endswith("running_mean"): v.zero_()
            if name.endswith("running_var"): v.fill_(1.0)
            if name.endswith("num_batches_tracked"): v.zero_()
            sd[name] = v
    return sd


def test_blocks():
    z = np.load(os.path.join(G, "blocks.npz"))
    sd = _block_sd(z, "b2n_sd.")
    y = O.bottle2neck(sd, "", T(z["b2n_x"]), O.Ctx(True), 1, False, False)
    assert (y - T(z["b2n_y"])).abs().max() < TOL
    for k in sd:   # running stats after one train-mode forward
        assert (sd[k].float() - T(z["b2n_sd." + k]).float()).abs().max() < TOL, k
    sd = _block_sd(z, "b2s_sd.")
    y = O.bottle2neck(sd, "", T(z["b2s_x"]), O.Ctx(True), 2, True, True)
    assert (y - T(z["b2s_y"])).abs().max() < TOL
    for k in sd:
        assert (sd[k].float() - T(z["b2s_sd." + k]).float()).abs().max() < TOL, k
    sd = _block_sd(z, "rfb_sd.")
    y = O.rfb(sd, "", T(z["rfb_x"]), O.Ctx(True))
    assert (y - T(z["rfb_y"])).abs().max() < TOL
    sd = _block_sd(z, "agg_sd.")
    fg, bg = O.aggregation(sd, "", T(z["agg_x1"]), T(z["agg_x2"]), T(z["agg_x3"]), O.Ctx(True))
    assert (fg - T(z["agg_fg"])).abs().max() < 5e-4 * max(1.0, float(T(z["agg_fg"]).abs().max()))
    assert (bg - T(z["agg_bg"])).abs().max() < 5e-4 * max(1.0, float(T(z["agg_bg"]).abs().max()))


@pytest.mark.parametrize("tag,full", [("96", True), ("352", False)])
def test_model_train_steps(tag, full):
    z = np.load(os.path.join(G, f"pranet_v2_{tag}.npz"))
    size, n = int(z["size"]), int(z["n"])
    P = W.make_state_dict(W.manifest_pranet_v2(1), seed=0)
    x, mask = W.synthetic_batch(n, size, seed=1234)
    st = {}
    for step in (1, 2):
        loss, outs, grads = O.train_step(P, st, x, mask)
        s = f"s{step}."
        assert abs(float(loss) - float(z[s + "loss"])) < 2e-4
        for i, o in enumerate(outs):
            ref = T(z[s + f"out{i}"])
            got = o if full else o[:, :, ::4, ::4]
            assert (got - ref).abs().max() < 1e-4, (step, i)
        for k in [f[len(s + "param."):] for f in z.files if f.startswith(s + "param.")]:
            assert (P[k].reshape(-1)[:256] - T(z[s + "param." + k])).abs().max() < 2e-6, k
        for k in [f[len(s + "buf."):] for f in z.files if f.startswith(s + "buf.")]:
            assert (P[k].reshape(-1)[:256] - T(z[s + "buf." + k])).abs().max() < 1e-5, k
        if step == 1:
            assert sorted(k for k in O.params_of(P) if k not in grads) == sorted(str(s_) for s_ in z["nograd"])
    # eval forward + MyTest tail + Dice
    with torch.no_grad():
        outs = O.pranet_v2_forward(P, x[:1], False)
    for i, o in enumerate(outs):
        ref = T(z[f"eval.out{i}"])
        got = o if full else o[:, :, ::4, ::4]
        assert (got - ref).abs().max() < 1e-4 * max(1.0, float(ref.abs().max())), i
    u8 = O.test_postprocess(outs, tuple(z["eval.u8"].shape))
    assert np.abs(u8.astype(int) - z["eval.u8"].astype(int)).max() <= 1
    assert abs(O.mean_dice(u8, z["eval.gt"]) - float(z["eval.meanDic"])) < 1e-3


def test_grad_probes_96():
    z = np.load(os.path.join(G, "pranet_v2_96.npz"))
    P = W.make_state_dict(W.manifest_pranet_v2(1), seed=0)
    x, mask = W.synthetic_batch(2, 96, seed=1234)
    keys = O.params_of(P)
    for k in keys:
        P[k].requires_grad_(True)
    loss = O.total_loss(O.pranet_v2_forward(P, x, True), mask)
    loss.backward()
    for f in z.files:
        if f.startswith("graw."):
            k = f[5:]
            ref = T(z[f])
            got = P[k].grad.reshape(-1)[:256]
            scale = max(1e-3, float(ref.abs().max()))
            assert (got - ref).abs().max() < 2e-3 * scale, k
            assert abs(float(P[k].grad.norm()) - float(z["grawnorm." + k])) < 2e-3 * float(z["grawnorm." + k]) + 1e-6, k


def test_v1_forward():
    z = np.load(os.path.join(G, "pranet_v1_96.npz"))
    P = W.make_state_dict(W.manifest_pranet_v1(), seed=1)
    x, _ = W.synthetic_batch(2, 96, seed=77)
    outs = O.pranet_v1_forward(P, x, True)
    for i, o in enumerate(outs):
        assert (o - T(z[f"out{i}"])).abs().max() < 1e-4, i


def test_pvt_v1_forward_and_manifest():
    """PVT_PraNet (PraNet_Res2Net.py:188-273): the oracle's restatement against the imported reference's vectors; manifest == the reference's state_dict."""
    import json
    ref = json.load(open(os.path.join(G, "manifest_pvt_v1.json")))
    assert [(k, list(v)) for k, v in W.manifest_pvt_pranet_v1().items()] == [(k, v) for k, v in ref["pvt_pranet"].items()]
    z = np.load(os.path.join(G, "pvt_pranet_v1_96.npz"))
    P = W.make_state_dict(W.manifest_pvt_pranet_v1(), seed=7)
    x, _ = W.synthetic_batch(2, 96, seed=78)
    outs = O.pvt_pranet_v1_forward(P, x, True)
    for i, o in enumerate(outs):
        assert (o - T(z[f"out{i}"])).abs().max() < 1e-4, i


def test_threshold_metrics_oracle_matches_reference_curves():
    """oracle.threshold_metrics == the imported reference's Fmeasure_calu sweep (tests/golden/make_golden.py evalm), NaNs included."""
    import warnings
    z = np.load(os.path.join(G, "eval_metrics.npz"))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tag in ("blob", "zero_pred", "zero_gt", "exact", "full"):
            cols, mae = O.threshold_metrics(z[tag + "_pred"], z[tag + "_gt"])
            assert np.array_equal(cols, z[tag + "_curves"], equal_nan=True), tag
            assert mae == float(z[tag + "_mae"])
            assert abs(O.mean_dice(z[tag + "_pred"], z[tag + "_gt"]) - float(z[tag + "_means"][3])) < 1e-12


def test_pvt_manifest_and_oracle_match_reference():
    """PVT_PraNet_V2: state_dict manifest and the oracle's PVTv2-B2 restatement (features, 8 outputs, loss, gradient probes) against the
    vectors the imported reference produced (tests/golden/make_golden.py pvt; DropPath off)."""
    ref = json.load(open(os.path.join(G, "manifest_pvt.json")))
    man = W.manifest_pvt_pranet_v2(1)
    assert [(k, list(v)) for k, v in man.items()] == list(ref["pvt_pranet_v2_k1"].items())
    z = np.load(os.path.join(G, "pvt_pranet_v2_96.npz"))
    sd = W.make_state_dict(man, seed=3)
    x, mask = W.synthetic_batch(2, 96, seed=4321)
    P = O.clone_sd(sd)
    with torch.no_grad():
        for i, f in enumerate(O.pvt_features(P, "backbone.", x)):
            assert float((f - torch.from_numpy(z[f"feat{i}"])).abs().max()) < 1e-5
    for k, v in P.items():
        if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    outs = O.pvt_pranet_v2_forward(P, x, True)
    for i, o in enumerate(outs):
        assert float((o.detach() - torch.from_numpy(z[f"out{i}"])).abs().max()) < 1e-4
    loss = O.total_loss(outs, mask)
    assert abs(float(loss) - float(z["loss"])) < 1e-4
    loss.backward()
    for k in z.files:
        if k.startswith("grawnorm."):
            g = P[k[len("grawnorm."):]].grad
            assert abs(float(g.norm()) - float(z[k])) <= 2e-3 * float(z[k]) + 1e-6, k


def test_pvt_one_channel_stem_oracle_matches_reference():
    """1-channel input: conv(1->3)+BN+ReLU stem (pranet.py:190-191) in front of PVT_PraNet_V2 - outputs, loss and the stem's gradient probes."""
    z = np.load(os.path.join(G, "pvt_pranet_v2_gray_64.npz"))
    sd = W.make_state_dict(W.manifest_pvt_pranet_v2(1), seed=11)
    x, mask = W.synthetic_batch(2, 64, seed=777)
    P = O.clone_sd(sd)
    for k, v in P.items():
        if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    outs = O.pvt_pranet_v2_forward(P, x[:, :1].contiguous(), True)
    for i, o in enumerate(outs):
        assert float((o.detach() - torch.from_numpy(z[f"out{i}"])).abs().max()) < 1e-4
    loss = O.total_loss(outs, mask)
    assert abs(float(loss) - float(z["losses"].sum())) < 1e-4
    loss.backward()
    for k in z.files:
        if k.startswith("grawnorm."):
            g = P[k[len("grawnorm."):]].grad
            assert abs(float(g.norm()) - float(z[k])) <= 2e-3 * float(z[k]) + 1e-6, k
    assert float((P["conv.1.running_mean"] - torch.from_numpy(z["rm.conv.1"])).abs().max()) < 1e-6


def test_emcad_manifest_and_oracle_match_reference():
    """BASELINE config 5: EMCADNet(dual, K=9, pvt_v2_b2) manifest + oracle/emcad_oracle.py (8 outputs, the 15-subset CE+Dice+BCE loss, gradient
    probes) against the imported reference's vectors (tests/golden/make_golden_emcad.py)."""
    from oracle import emcad_oracle as E
    ref = json.load(open(os.path.join(G, "manifest_emcad.json")))
    man = W.manifest_emcadnet(9)
    assert [(k, list(v)) for k, v in man.items()] == list(ref["emcadnet_dual_k9"].items())
    z = np.load(os.path.join(G, "emcad_64.npz"))
    P = O.clone_sd(W.make_state_dict(man, seed=5))
    for k, v in P.items():
        if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
    outs = E.emcadnet_forward(P, torch.from_numpy(z["x"]), True)
    for i, o in enumerate(outs):
        assert float((o.detach() - torch.from_numpy(z[f"out{i}"])).abs().max()) < 1e-4
    loss = E.mutation_loss(outs, torch.from_numpy(z["label"]), torch.from_numpy(z["bg_mask"]))
    assert abs(float(loss) - float(z["loss"])) < 1e-4
    loss.backward()
    for k in z.files:
        if k.startswith("grawnorm."):
            g = P[k[len("grawnorm."):]].grad
            assert abs(float(g.norm()) - float(z[k])) <= 2e-3 * float(z[k]) + 2e-6, k


def test_input_transform_oracle_matches_pillow_vectors():
    """oracle.input_oracle (restatement of Pillow's antialiased bilinear resize + ToTensor + Normalize, dataloader.py:104-111) against the vectors
    Pillow itself produced (tests/golden/make_golden_input.py): the uint8 resize bit for bit, the tensors exactly."""
    from oracle import input_oracle as I
    z = np.load(os.path.join(G, "input_pipeline.npz"))
    names = sorted({k.split(".")[0] for k in z.files})
    assert {"down", "up", "mixed", "gt", "small", "same", "big"} <= set(names)
    for n in names:
        img = z[n + ".in"]; S = z[n + ".resized"].shape[0]
        assert np.array_equal(I.pil_resize_bilinear_u8(img, S, S), z[n + ".resized"]), n
        if n + ".tensor" in z.files and img.ndim == 3:
            t, _ = I.train_transform(img, img[:, :, 0], S)
            assert np.array_equal(t, z[n + ".tensor"]), n
    _, g = I.train_transform(z["down.in"], z["gt.in"], 96)
    assert np.array_equal(g, z["gt.tensor"])


def _cond_weights(z, calibrated):
    """Conditioned weights of tests/golden/make_golden_cond.py (bn3 gamma x 0.05), optionally with the calibrated running statistics."""
    P = W.make_state_dict(W.manifest_pranet_v2(1), seed=0, bn3_gamma=float(z["bn3_gamma"]))
    if calibrated:
        for f in z.files:
            if f.startswith("calib."):
                P[f[6:]] = T(z[f]).clone()
    return P


def test_conditioned_fixture_oracle_train_and_eval():
    """The oracle against the CONDITIONED reference vectors (pranet_v2_cond.npz): there the reference's own fp32 and float64 runs agree to
    ~2e-5 on the logits, so north_star's literal '1e-4 abs on fp32 logits / Dice within 1e-3' is what is asserted - train mode (8 x 96^2:
    logits, losses, gradient probes) and eval mode with calibrated BatchNorm statistics (1 x 352^2, 2 x 96^2: logits, uint8 map, meanDic)."""
    z = np.load(os.path.join(G, "pranet_v2_cond.npz"))
    assert float(np.max(z["t96.own_abs"])) < 3e-5 and float(np.max(z["e352.own_abs"])) < 5e-5      # the fixture is well-conditioned (reference vs itself)
    P = _cond_weights(z, False)
    x, mask = W.synthetic_batch(int(z["t96.n"]), int(z["t96.size"]), seed=4242)
    for k in O.params_of(P):
        P[k].requires_grad_(True)
    outs = O.pranet_v2_forward(P, x, True)
    losses = [O.structure_loss(outs[i], outs[i + 4], mask, 1 - mask) for i in range(4)]
    (losses[3] + losses[2] + losses[1] + losses[0]).backward()
    assert np.abs(np.array([float(l) for l in losses]) - z["t96.losses"]).max() < 1e-5
    for i, o in enumerate(outs):
        assert float((o.detach() - T(z[f"t96.out{i}"])).abs().max()) < 1e-4, i              # LITERAL north_star bound, against the reference's fp32 logits
        assert float((o.detach()[:, :, ::2, ::2].double() - T(z[f"t96.f64.out{i}"])).abs().max()) < 1e-4, i
    for f in z.files:
        if f.startswith("t96.graw."):
            k = f[9:]
            r32, r64 = T(z[f]).double(), T(z["t96.f64.graw." + k]).double()
            got = P[k].grad.reshape(-1)[:256].double()
            own = float((r32 - r64).norm() / (r64.norm() + 1e-30))
            e = float((got - r64).norm() / (r64.norm() + 1e-30))
            assert e <= max(1e-5, 3 * own), (k, e, own)                                      # two fp32 evaluations of the same graph
    assert sorted(k for k in O.params_of(P) if P[k].grad is None) == sorted(str(s) for s in z["t96.nograd"])
    # eval mode, calibrated running statistics (MyTest_med.py:98-111)
    Pc = _cond_weights(z, True)
    for tag in ("e96", "e352"):
        n, size, s32 = int(z[f"{tag}.n"]), int(z[f"{tag}.size"]), int(z[f"{tag}.stride32"])
        x, mask = W.synthetic_batch(n, size, seed=4242)
        with torch.no_grad():
            outs = O.pranet_v2_forward(Pc, x, False)
        for i, o in enumerate(outs):
            assert float((o[:, :, ::s32, ::s32] - T(z[f"{tag}.out{i}"])).abs().max()) < 1e-4, (tag, i)
        u8 = O.test_postprocess([o[:1] for o in outs], tuple(z[f"{tag}.u8"].shape))
        assert np.abs(u8.astype(int) - z[f"{tag}.u8"].astype(int)).max() <= 1
        assert abs(O.mean_dice(u8, z[f"{tag}.gt"]) - float(z[f"{tag}.meanDic"])) < 1e-3


VARIANTS = {"k3": (dict(), "v2", 0, 1234), "k3lin": (dict(use_softmax=False, sem_downsample=2), "v2", 0, 1234), "pvtk3": (dict(), "pvt", 3, 4321)}


@pytest.mark.parametrize("tag", sorted(VARIANTS))
def test_constructor_variants_oracle_matches_reference(tag):
    """The constructor paths the binary scripts never take - PraNet_V2() with its DEFAULT num_class=3 (pranet.py:270), use_softmax=False +
    sem_downsample=2 (pranet.py:349-350, 365-368), PVT_PraNet_V2(num_class=3) - against the reference's own classes (tests/golden/make_golden_variants.py):
    8 outputs, gradient heads / norms under fixed cotangents, BN buffers, the parameters without gradient."""
    z = np.load(os.path.join(G, "pranet_v2_variants.npz"))
    kw, fam, wseed, xseed = VARIANTS[tag]
    man = W.manifest_pranet_v2(3) if fam == "v2" else W.manifest_pvt_pranet_v2(3)
    P = W.make_state_dict(man, seed=wseed)
    x, _ = W.synthetic_batch(2, 96, seed=xseed)
    keys = O.params_of(P)
    for k in keys:
        P[k].requires_grad_(True)
    fwd = O.pranet_v2_forward if fam == "v2" else O.pvt_pranet_v2_forward
    outs = fwd(P, x, True, **kw)
    assert tuple(outs[0].shape) == tuple(int(v) for v in z[f"{tag}.shape"])
    g = torch.Generator().manual_seed(99)
    gs = [torch.randn(o.shape, generator=g) for o in outs]
    sum((o * c).sum() for o, c in zip(outs, gs)).backward()
    for i, o in enumerate(outs):
        ref = T(z[f"{tag}.out{i}"])
        assert float((o.detach() - ref).abs().max()) < 1e-4 * max(1.0, float(ref.abs().max())), (tag, i)
    for f in z.files:
        if f.startswith(f"{tag}.grawnorm."):
            k = f[len(f"{tag}.grawnorm."):]
            assert abs(float(P[k].grad.norm()) - float(z[f])) <= 3e-3 * float(z[f]) + 1e-6, k
            ref = T(z[f"{tag}.graw.{k}"])
            assert float((P[k].grad.reshape(-1)[:256] - ref).abs().max()) < 3e-3 * max(1e-3, float(ref.abs().max())), k
        if f.startswith(f"{tag}.buf."):
            k = f[len(f"{tag}.buf."):]
            assert float((P[k].detach().reshape(-1)[:256] - T(z[f])).abs().max()) < 1e-5, k
    assert sorted(k for k in keys if P[k].grad is None) == sorted(str(s_) for s_ in z[f"{tag}.nograd"])


def test_full_metrics_oracle_matches_reference_eval_for_testAllInOne():
    """oracle.full_metrics (Sm, wFm, meanEm next to meanDic / meanIoU / mae) == the imported reference's eval_for_testAllInOne (eval.py:18-66; vectors from
    tests/golden/make_golden_evalfull.py), and the oracle's restatement of scipy's exact Euclidean feature transform == scipy's own output INDEX BY INDEX on a
    map full of equidistant sites (the tie-breaking decides which error value original_WFb propagates to a background pixel)."""
    import warnings
    z = np.load(os.path.join(G, "eval_full.npz"))
    names = ["meanDic", "meanIoU", "wFm", "Sm", "meanEm", "mae"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tag in ("blob", "zero_pred", "exact", "full", "two", "rand352"):
            m = O.full_metrics(z[tag + "_pred"], z[tag + "_gt"])
            for k, ref in zip(names, z[tag + "_vals"]):
                assert abs(m[k] - float(ref)) <= 1e-12 * max(1.0, abs(float(ref))), (tag, k, m[k], float(ref))
            assert np.abs(m["E"] - z[tag + "_E"]).max() <= 1e-12, tag
    dst, ri, rj = O.edt_nearest(z["tie_gt"])
    assert np.array_equal(ri, z["tie_ri"]) and np.array_equal(rj, z["tie_rj"]) and np.array_equal(dst, z["tie_dst"])
    # ... and against scipy in this process, on random maps of several densities (scipy is a dependency of the reference, present in the image)
    from scipy.ndimage import distance_transform_edt
    rng = np.random.default_rng(3)
    for dens in (0.002, 0.05, 0.5):
        g = (rng.random((37, 61)) < dens).astype(np.float64)
        g[rng.integers(0, 37), rng.integers(0, 61)] = 1
        d_s, i_s = distance_transform_edt(1 - g, return_indices=True)
        d_o, ri, rj = O.edt_nearest(g)
        assert np.array_equal(d_o, d_s) and np.array_equal(ri, i_s[0]) and np.array_equal(rj, i_s[1]), dens
