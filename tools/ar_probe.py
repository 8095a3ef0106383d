import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/pranet-v2_amd")
os.environ.setdefault("PN2_NO_PRETRAINED", "1"); os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29455")
import torch, torch.distributed as dist
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.zeros(30_000_000, device=dev)
big = torch.zeros(256 << 20, dtype=torch.uint8, device=dev)
def busy():
    for _ in range(40):
        big.fill_(1)          # ~40 x 45 us of GPU work queued
for mode in ("allreduce", "allreduce_async", "none"):
    ts = []
    for it in range(6):
        torch.cuda.synchronize()
        busy()
        t0 = time.perf_counter()
        if mode == "allreduce":
            dist.all_reduce(x)
        elif mode == "allreduce_async":
            w = dist.all_reduce(x, async_op=True); w.wait()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append((1e3 * (t1 - t0), 1e3 * (t2 - t1)))
    print(mode, "host ms in call / remaining GPU drain ms:", [f"{a:.3f}/{b:.3f}" for a, b in ts[1:]])
dist.destroy_process_group()
