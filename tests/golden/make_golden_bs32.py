#!/usr/bin/env python3
"""Golden vectors at the BENCHMARKED batch: the imported reference's PraNet_V2(num_class=1) train step (MyTrain_med.py:59-86: forward, 4 x structure_loss,
backward) at 32 x 3 x 352 x 352 - the shape bench.py times - so that the object bench.py measures (Trainer.capture / replay at bs=32) is pinned against
the reference itself and not only by property tests.  Runs ONLY in the build container (imports /root/reference through _ref_import.py); only data is
written (pranet_v2_bs32.npz, < 2 MB).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_bs32.py [cond|rand] [f32|f64|bf16] ...

Two weight sets (oracle.weights.make_state_dict, seed 0): "cond" = bn3 gamma x 0.05 (residual branches as small corrections, the regime of a trained
checkpoint: the carrier of north_star's literal 1e-4, see make_golden_cond.py) and "rand" = the plain default init bench.py runs on (chaotic: relative
gates only).  Three runs of the SAME reference classes per set: fp32, float64, and fp32 modules under torch.autocast("cpu", bfloat16) - the bf16 yardstick.

Stored per set: the 8 logit maps at stride 16 (fp32 run, and the float64 run rounded to fp32: every image for cond / every second image for rand),
the reference's own max |fp32 - float64| per map over the FULL maps, the 4 pair losses of every run, the 38 gradient probes (first 256 elements; fp32 and
float64), torch-autocast's rel-L2 per map / per probe / losses against float64 (full tensors), and the heads of every BatchNorm running_mean /
running_var after the step (fp32 run).

Memory: an fp32 train step of the reference at this batch holds ~30 GB of autograd state; the float64 run would not fit the 64 GB container, so its
sixteen Bottle2neck blocks run under torch.utils.checkpoint (non-reentrant: the block's forward is re-executed during backward - the same modules, the
same arithmetic, deterministic on the CPU; only BatchNorm's running-statistics side effect happens twice, and those buffers are taken from the fp32 run).
Every run is cached under /tmp/pn2_bs32 so that an interrupted generation resumes.
"""
import os, sys, time
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE); sys.path.insert(0, ROOT)
from _ref_import import import_reference            # noqa: E402
from oracle import weights as W                      # noqa: E402
from make_golden import PROBE_PARAMS, head, npy      # noqa: E402

torch.set_num_threads(8)
R = import_reference()
N, SIZE, SEED, STRIDE = 32, 352, 4242, 16
CACHE = "/tmp/pn2_bs32"
os.makedirs(CACHE, exist_ok=True)


def rell2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def run(which, kind):
    """-> dict: full outputs (kept only in the cache), losses, gradient probes, BatchNorm buffers."""
    path = os.path.join(CACHE, f"{which}_{kind}.pt")
    if os.path.exists(path):
        return torch.load(path)
    t0 = time.time()
    sd = W.make_state_dict(W.manifest_pranet_v2(1), seed=0, bn3_gamma=0.05 if which == "cond" else None)
    dt = torch.float64 if kind == "f64" else torch.float32
    model = R.pranet.PraNet_V2(num_class=1)
    model.load_state_dict(sd, strict=True)
    model = model.to(dt).train()
    x, mask = W.synthetic_batch(N, SIZE, seed=SEED)
    x, mask = x.to(dt), mask.to(dt)
    ckpt = kind == "f64"
    if ckpt:
        from torch.utils.checkpoint import checkpoint
        B2 = R.res2.Bottle2neck
        orig = B2.forward
        B2.forward = lambda self, inp: checkpoint(orig, self, inp, use_reentrant=False)
    try:
        if kind == "bf16":
            with torch.autocast("cpu", torch.bfloat16):
                outs = model(x)
            outs = [o.float() for o in outs]
        else:
            outs = model(x)
        losses = [R.train.structure_loss(outs[i], outs[i + 4], mask, 1 - mask) for i in range(4)]          # MyTrain_med.py:78-81
        (losses[3] + losses[2] + losses[1] + losses[0]).backward()                                           # :82-84
    finally:
        if ckpt:
            B2.forward = orig
    names = dict(model.named_parameters())
    res = {"outs": [o.detach() for o in outs], "losses": [float(l) for l in losses],
           "grads": {k: names[k].grad.detach().reshape(-1)[:256].clone() for k in PROBE_PARAMS},
           "gnorm": {k: float(names[k].grad.norm()) for k in PROBE_PARAMS},
           "nograd": sorted(k for k, p in names.items() if p.grad is None),
           "bufs": {k: v.detach().reshape(-1)[:32].clone() for k, v in model.state_dict().items() if k.endswith("running_mean") or k.endswith("running_var")},
           "seconds": time.time() - t0}
    torch.save(res, path)
    print(f"[{which} {kind}] {res['seconds']:.0f} s, losses {res['losses']}", flush=True)
    return res


def assemble():
    out = {"n": np.array(N), "size": np.array(SIZE), "seed": np.array(SEED), "stride": np.array(STRIDE), "bn3_gamma": np.array(0.05)}
    for which in ("cond", "rand"):
        have = {k: os.path.exists(os.path.join(CACHE, f"{which}_{k}.pt")) for k in ("f32", "f64", "bf16")}
        if not (have["f32"] and have["f64"]):
            print(f"{which}: fp32 / float64 runs missing, set skipped")
            continue
        r32, r64 = run(which, "f32"), run(which, "f64")
        step = 1 if which == "cond" else 2                      # images of the maps that are stored (fp32 and float64 runs alike)
        out[f"{which}.image_step"] = np.array(step)
        out[f"{which}.losses"] = np.array(r32["losses"]); out[f"{which}.f64.losses"] = np.array(r64["losses"])
        out[f"{which}.nograd"] = np.array(r32["nograd"])
        own = []
        for i in range(8):
            out[f"{which}.out{i}"] = npy(r32["outs"][i][::step, :, ::STRIDE, ::STRIDE])
            out[f"{which}.f64.out{i}"] = npy(r64["outs"][i][::step, :, ::STRIDE, ::STRIDE]).astype(np.float32)
            own.append(float((r32["outs"][i].double() - r64["outs"][i]).abs().max()))
        out[f"{which}.own_abs"] = np.array(own)
        for k in PROBE_PARAMS:
            out[f"{which}.graw." + k] = npy(r32["grads"][k]); out[f"{which}.f64.graw." + k] = npy(r64["grads"][k]).astype(np.float64)
        for k, v in r32["bufs"].items():
            out[f"{which}.buf." + k] = npy(v)
        print(which, "reference own |fp32 - f64| max %.2e" % max(own), " probes own rel-L2 median %.2e" %
              float(np.median([rell2(r32["grads"][k], r64["grads"][k]) for k in PROBE_PARAMS])))
        if have["bf16"]:
            rb = run(which, "bf16")
            out[f"{which}.bf16.losses"] = np.array(rb["losses"])
            out[f"{which}.bf16.rel"] = np.array([rell2(rb["outs"][i], r64["outs"][i]) for i in range(8)])
            out[f"{which}.bf16.grel"] = np.array([rell2(rb["grads"][k], r64["grads"][k]) for k in PROBE_PARAMS])
            print(which, "torch-autocast bf16 rel-L2 per map", ["%.3f" % v for v in out[f"{which}.bf16.rel"]],
                  " probes median %.3f max %.3f" % (float(np.median(out[f"{which}.bf16.grel"])), float(out[f"{which}.bf16.grel"].max())))
    path = os.path.join(HERE, "pranet_v2_bs32.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays,", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    args = sys.argv[1:]
    sets = [a for a in args if a in ("cond", "rand")] or ["cond", "rand"]
    kinds = [a for a in args if a in ("f32", "f64", "bf16")] or ["f32", "f64", "bf16"]
    if "assemble" not in args:
        for which in sets:
            for kind in kinds:
                run(which, kind)
    assemble()
