#!/usr/bin/env python3
"""Where the ~0.45 ms of a one-rank data-parallel replay go: times Trainer.replay() with the collectives replaced by (a) nothing, (b) a bare stream fork / join,
(c) the real RCCL all-reduce.  GPU box only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
os.environ.setdefault("PN2_NO_PRETRAINED", "1"); os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29466")
os.environ.setdefault("PN2_DP_BUCKET_MB", "128")
import torch, torch.distributed as dist
import pn2
from pn2.trainer import Trainer
from lib.pranet import PraNet_V2
from bench import synthetic
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
torch.manual_seed(0)
model = PraNet_V2(num_class=1).to(dev).train()
tr = Trainer(model, lr=1e-4, clip=0.5, process_group=dist.group.WORLD, force_dp=True)
x, m = synthetic(32, 352, 1234, dev)
tr.capture(x, m, warmup=2)
st = tr._cur
side = torch.cuda.Stream()
def timed(fn, n=60):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
def noex():
    for g, _ in st.segments: g.replay()
    st.graph_opt.replay()
def forkjoin():
    for g, _ in st.segments:
        g.replay()
        side.wait_stream(torch.cuda.current_stream())
    torch.cuda.current_stream().wait_stream(side)
    st.graph_opt.replay()
def forkjoin_work():
    for g, bs in st.segments:
        g.replay()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            tr.gflat[:16].add_(0)
    torch.cuda.current_stream().wait_stream(side)
    st.graph_opt.replay()
def event_only():
    for g, _ in st.segments:
        g.replay()
        e = torch.cuda.Event(); e.record()
    st.graph_opt.replay()
def host_time(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(1e3 * (time.perf_counter() - t0))
    torch.cuda.synchronize()
    ts.sort()
    return ts[len(ts) // 2], ts[-1]
def direct():
    for g, bs in st.segments:
        g.replay()
        for b in bs:
            a, e, _ = tr.buckets.buckets[b]
            w = dist.all_reduce(tr.gflat[a:e], async_op=True)
            w.wait()
    st.graph_opt.replay()
def sync_op():
    for g, bs in st.segments:
        g.replay()
        for b in bs:
            a, e, _ = tr.buckets.buckets[b]
            dist.all_reduce(tr.gflat[a:e])
    st.graph_opt.replay()
print("host ms per replay() call (median, max): no collectives %.3f %.3f | RCCL %.3f %.3f" % (host_time(noex) + host_time(tr.replay)))
print(f"direct all_reduce async+wait {timed(direct):.3f} ms")
print(f"all_reduce sync op  {timed(sync_op):.3f} ms")
print("segments", len(st.segments))
print(f"no collectives      {timed(noex):.3f} ms")
print(f"event record only   {timed(event_only):.3f} ms")
print(f"bare fork / join    {timed(forkjoin):.3f} ms")
print(f"fork / tiny kernel / join {timed(forkjoin_work):.3f} ms")
print(f"RCCL all-reduce     {timed(tr.replay):.3f} ms")
print(f"no collectives      {timed(noex):.3f} ms")
dist.destroy_process_group()
