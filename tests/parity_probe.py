#!/usr/bin/env python3
"""Where does the fp32 whole-model error come from?  (What's weak 1)

Runs PraNet-V2 (fp32 MFMA path) on the committed fixture inputs and prints, against the reference's float64 vectors
(tests/golden/pranet_v2_*.npz: f64.* = the imported reference run in float64, s1.* = the same reference in float32):
  * max |logit - ref64| per output next to the reference's own fp32-vs-f64 gap,
  * the relative L2 error of every gradient probe next to the reference's own,
  * encoder features x1..x4 (and their per-layer growth) against the CPU oracle run in float64 / float32 on the same weights.
Usage (GPU box):  python tests/parity_probe.py [96|352]
"""
import os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
os.environ.setdefault("PN2_NO_PRETRAINED", "1")
import numpy as np
import torch


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "96"
    import pn2
    from lib.pranet import PraNet_V2
    from pn2.loss import structure_loss
    from pn2.graph import run_module
    from oracle import weights as W
    from oracle import pranet_oracle as O
    z = np.load(os.path.join(ROOT, "tests", "golden", f"pranet_v2_{tag}.npz"))
    size, n = int(z["size"]), int(z["n"])
    pn2.set_compute_dtype("fp32")
    sd = W.make_state_dict(W.manifest_pranet_v2(1), seed=0)
    model = PraNet_V2(num_class=1); model.load_state_dict(sd, strict=True); model = model.cuda().train()
    x, mask = W.synthetic_batch(n, size, seed=1234)
    # ---- encoder features vs the oracle in float64 / float32
    feats = run_module(lambda e, a: model.backbone._build_features(e, a), [x.cuda()], model.hot_parameters(), True)
    for dt in (torch.float64, torch.float32):
        P = {k: (v.to(dt) if v.is_floating_point() else v.clone()) for k, v in sd.items()}
        ctx = O.Ctx(True) if hasattr(O, "Ctx") else None
        ref = O.res2net_features(P, "backbone.", x.to(dt), ctx)
        if dt == torch.float64:
            ref64 = ref
        for i, (f, r) in enumerate(zip(feats, ref)):
            e_ours = float((f.detach().cpu().double() - ref64[i].double()).abs().max())
            e_ref = float((r.double() - ref64[i].double()).abs().max())
            if dt == torch.float32:
                print(f"x{i + 1}: max|ours-ref64| {e_ours:.3e}   max|oracle32-ref64| {e_ref:.3e}   max|ref64| {float(ref64[i].abs().max()):.2f}")
    # fresh model (running statistics were touched by the feature pass)
    model = PraNet_V2(num_class=1); model.load_state_dict(sd, strict=True); model = model.cuda().train()
    outs = model(x.cuda())
    mg = mask.cuda()
    losses = [structure_loss(outs[i], outs[i + 4], mg, 1 - mg) for i in range(4)]
    (losses[3] + losses[2] + losses[1] + losses[0]).backward()
    print("losses: ours-ref64", [f"{float(l) - float(r):+.2e}" for l, r in zip(losses, z["f64.losses"])], " ref32-ref64", [f"{a - b:+.2e}" for a, b in zip(z["s1.losses"], z["f64.losses"])])
    full = tag == "96"
    for i, o in enumerate(outs):
        r32 = torch.from_numpy(z[f"s1.out{i}"]).double(); r64 = torch.from_numpy(z[f"f64.out{i}"])
        got = (o.detach().cpu() if full else o.detach().cpu()[:, :, ::4, ::4]).double()
        print(f"out{i}: max|ours-ref64| {float((got - r64).abs().max()):.3e}   own {float((r32 - r64).abs().max()):.3e}")
    named = dict(model.named_parameters())
    rows = []
    for f in z.files:
        if f.startswith("graw."):
            k = f[5:]
            r32 = torch.from_numpy(z[f]).double(); r64 = torch.from_numpy(z["f64." + f]).double()
            got = named[k].grad.reshape(-1)[:256].cpu().double()
            own = float((r32 - r64).norm() / (r64.norm() + 1e-30)); e = float((got - r64).norm() / (r64.norm() + 1e-30))
            rows.append((e / max(own, 1e-12), k, e, own))
    for ratio, k, e, own in sorted(rows, reverse=True):
        print(f"grad {k:45s} ours {e:.3e}  own {own:.3e}  ratio {ratio:6.2f}")


if __name__ == "__main__":
    main()
