#!/usr/bin/env python3
"""Where the drop-in loop (MyTrain_med.py:59-86 on the mirror classes) spends its step: per phase, the host time to ISSUE it and the GPU time it takes when the
phases are separated by synchronisations; plus the un-synchronised step for both loss variants (the script's own torch-op structure_loss / pn2.loss).
GPU box: python tools/module_surface_breakdown.py [--verbatim]"""
import os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
os.environ.setdefault("PN2_NO_PRETRAINED", "1")
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    import pn2
    from lib.pranet import PraNet_V2
    from utils.utils import clip_gradient
    verbatim = "--verbatim" in sys.argv
    loss_fn = bench.torch_structure_loss if verbatim else __import__("pn2.loss", fromlist=["structure_loss"]).structure_loss
    dev = torch.device("cuda", 0)
    pn2.set_compute_dtype("bf16")
    torch.manual_seed(0)
    model = PraNet_V2(num_class=1).to(dev).train()
    opt = torch.optim.Adam(model.parameters(), 1e-4)
    x, m = bench.synthetic(32, 352, 1234, dev)
    bg = 1 - m
    phases = ["zero_grad", "forward", "loss", "backward", "clip", "adam"]
    host = {k: 0.0 for k in phases}
    gpu = {k: 0.0 for k in phases}

    def step(timed):
        def run(name, fn):
            if not timed:
                return fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            host[name] += t1 - t0
            gpu[name] += t2 - t0
            return r
        run("zero_grad", lambda: opt.zero_grad())
        o = run("forward", lambda: model(x))
        loss = run("loss", lambda: loss_fn(o[3], o[7], m, bg) + loss_fn(o[2], o[6], m, bg) + loss_fn(o[1], o[5], m, bg) + loss_fn(o[0], o[4], m, bg))
        run("backward", lambda: loss.backward())
        run("clip", lambda: clip_gradient(opt, 0.5))
        run("adam", lambda: opt.step())
        return loss
    for _ in range(8):
        step(False)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        step(False)
    torch.cuda.synchronize()
    free = (time.perf_counter() - t0) / n
    for _ in range(n):
        step(True)
    print(f"loss = {'torch ops (verbatim)' if verbatim else 'pn2.loss'}; un-synchronised step {1e3 * free:.3f} ms = {32 / free:.1f} images/s")
    print(f"{'phase':10s} {'host issue ms':>14s} {'issue+GPU ms':>14s}")
    for k in phases:
        print(f"{k:10s} {1e3 * host[k] / n:14.3f} {1e3 * gpu[k] / n:14.3f}")
    print(f"{'sum':10s} {1e3 * sum(host.values()) / n:14.3f} {1e3 * sum(gpu.values()) / n:14.3f}")


if __name__ == "__main__":
    main()
