#!/usr/bin/env python3
"""bench.py — images/sec of the full PraNet-V2 (Res2Net-50) training step at 352x352, bs=32 per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A step = forward + 4x structure loss + backward + clamp(0.5) + Adam(1e-4) [+ RCCL gradient all-reduce], i.e. everything the
reference runs between optimizer.zero_grad() and optimizer.step() (MyTrain_med.py:59-86), on synthetic data (images N(0,1),
1-3 random ellipses per mask) with random-init weights (the reference's default init, seed 0).  Rank 0 prints ONE JSON line.

Extra objects in that line:
  roofline     — the dominant kernel family (implicit-GEMM conv forward+dgrad on MFMA): algorithmic FLOPs / measured kernel
                 time, timed with HIP events on the launch stream inside an instrumented extra step.
  cpu_baseline — the CPU oracle (oracle/pranet_oracle.py, a torch-CPU port of the same step) timed on this host's cores on a
                 bounded sample (rank 0, N=1 only).
"""
import argparse, json, os, sys, time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
os.environ.setdefault("PN2_NO_PRETRAINED", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0
TRAIN_GFLOP_PER_IMG = 78.02   # 13.004 GMAC fwd x 2 x 3 (fwd + dgrad + wgrad)


def synthetic(n, size, seed, device):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn((n, 3, size, size), generator=g)
    yy, xx = torch.meshgrid(torch.arange(size, dtype=torch.float32), torch.arange(size, dtype=torch.float32), indexing="ij")
    m = torch.zeros((n, 1, size, size))
    for i in range(n):
        for _ in range(int(torch.randint(1, 4, (1,), generator=g))):
            cy, cx = (torch.rand(2, generator=g) * 0.6 + 0.2) * size
            ry, rx = (torch.rand(2, generator=g) * 0.16 + 0.08) * size
            m[i, 0][((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0] = 1.0
    return x.to(device), m.to(device)


def cpu_baseline(size, bs):
    """The oracle's train step (kind 'port') on the host cores: 1 warm-up + 2 timed steps at a reduced batch."""
    from oracle import pranet_oracle as O
    from oracle import weights as W
    cores = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(cores)
    P = W.make_state_dict(W.manifest_pranet_v2(1), seed=0)
    x, mask = W.synthetic_batch(bs, size, seed=1234)
    st = {}
    O.train_step(P, st, x, mask)
    t0 = time.perf_counter()
    reps = 2
    for _ in range(reps):
        O.train_step(P, st, x, mask)
    dt = (time.perf_counter() - t0) / reps
    return {"value": round(bs / dt, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"oracle train step (fwd+4x structure_loss+bwd+clamp+Adam) fp32, bs={bs} at {size}x{size}, 1 warm-up + {reps} timed steps, {dt:.2f} s/step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--size", type=int, default=352)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--model", default="res2net", choices=["res2net", "pvt", "emcad"],
                    help="res2net = BASELINE config 2/3 (headline); pvt = config 4 (PVT_PraNet_V2, use --batch 16); "
                         "emcad = config 5 (EMCADNet dual K=9 + the 15-subset loss + AdamW, use --batch 16 --size 512)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=4)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pg = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
        pg = dist.group.WORLD

    import pn2
    from pn2.trainer import Trainer
    from lib.pranet import PraNet_V2, PVT_PraNet_V2
    pn2.set_compute_dtype(args.dtype)
    torch.manual_seed(0)
    if args.model == "emcad":
        # EMCAD/trainer.py: 1-channel Synapse slices, 9 classes, AdamW(lr 1e-4, wd 1e-4), supervision='mutation' on the dual heads
        from lib.networks import EMCADNet
        model = EMCADNet(num_classes=9, kernel_sizes=[1, 3, 5], expansion_factor=2, activation="relu6", encoder="pvt_v2_b2", pretrain=False, dual=True).to(dev).train()
        tr = Trainer(model, lr=1e-4, clip=None, weight_decay=1e-4, loss="mutation", hot=model.hot_parameters(True), process_group=pg)
        g = torch.Generator(device="cpu").manual_seed(1234 + rank)
        x = torch.randn(args.batch, 1, args.size, args.size, generator=g).to(dev)
        lab = torch.randint(0, 9, (args.batch, args.size // 16, args.size // 16), generator=g).to(dev)
        lab = torch.nn.functional.interpolate(lab[:, None].float(), size=(args.size, args.size), mode="nearest")[:, 0].long()
        m = (lab, torch.stack([(lab != k).float() for k in range(9)], 1))
    else:
        model = (PraNet_V2 if args.model == "res2net" else PVT_PraNet_V2)(num_class=1).to(dev).train()
        tr = Trainer(model, lr=1e-4, clip=0.5, process_group=pg)
        x, m = synthetic(args.batch, args.size, 1234 + rank, dev)

    use_graph = not args.no_graph
    if use_graph:
        tr.capture(x, m, warmup=2)
        step = lambda: tr.replay()
    else:
        step = lambda: tr.step(x, m)
    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t)
    loss_v = [float(v) for v in loss.float().cpu()]

    # ---- instrumented extra step: per-kernel-family HIP-event timing (not part of the timed region)
    from pn2 import profile as prof
    roof = prof.measure_step(tr, x, m, args.dtype)
    names = {"res2net": "PraNet-V2 Res2Net50", "pvt": "PVT-PraNet-V2 (pvt_v2_b2, DropPath 0.1)", "emcad": "EMCADNet dual K=9 (pvt_v2_b2 encoder, EMCAD decoder)"}
    what = ("fwd+15-subset CE/Dice/BCE loss+bwd+AdamW" if args.model == "emcad" else "fwd+4x structure_loss+bwd+clamp+Adam")

    if rank == 0:
        ips = world * args.batch * args.steps / el
        out = {
            "metric": "images/sec (train fwd+bwd) at 352x352 bs=32/GPU",
            "value": round(ips, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * el / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{names[args.model]} training step ({what}), bs={args.batch}/GPU {args.size}x{args.size}, "
                                   f"random-init, synthetic {'block labels' if args.model == 'emcad' else 'ellipse masks'}", "global_batch": world * args.batch, "parallelism": f"dp{world}",
                       "launch": "hipGraph replay" if use_graph else "eager"},
            "loss": loss_v,
            "mfma_frac_whole_step": (None if args.model == "emcad" else
                                     round(ips / world * (TRAIN_GFLOP_PER_IMG if args.model == "res2net" else 72.3) / 1e3 / (PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS), 4)),
            "roofline": roof["roofline"], "kernels": roof["kernels"],
        }
        if world == 1 and not args.no_cpu_baseline and args.model != "emcad":
            out["cpu_baseline"] = cpu_baseline(args.size, args.cpu_batch)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
