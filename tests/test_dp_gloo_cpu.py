"""world_size-2 gloo test of the data-parallel gradient path (pn2/dp.py): bucket launch order follows backward,
every element is summed exactly once, and (sum * 1/world) equals the mean of the per-shard gradients."""
import os, sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pn2.dp import GradBuckets
    sizes = [1000, 24, 5000, 8, 300, 4096, 12]
    spans, off = [], 0
    for i, n in enumerate(sizes):
        spans.append((i, off, n)); off += n
    g = torch.Generator().manual_seed(100 + rank)
    gflat = torch.randn(off, generator=g)
    local = gflat.clone()
    bk = GradBuckets(gflat, spans, bucket_bytes=4 * 4000)
    assert len(bk.buckets) >= 3 and bk.buckets[0][0] == 0 and bk.buckets[-1][1] == off
    # backward produces the last parameters first
    written = set()
    bk.reset()
    for i in reversed(range(len(sizes))):
        written.add(i)
        bk.launch_ready(written)
    bk.finish()
    order = list(bk.order)
    # gather every rank's local gradient to check the sum
    allg = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(allg, local)
    ok_sum = torch.allclose(gflat, sum(allg), atol=1e-6)
    mean = gflat * (1.0 / world)
    ok_mean = torch.allclose(mean, torch.stack(allg).mean(0), atol=1e-6)
    # second step with a parameter that never gets a gradient: finish() must still reduce its bucket
    gflat.copy_(local)
    bk.reset()
    bk.launch_ready({len(sizes) - 1})
    bk.finish()
    ok_rest = torch.allclose(gflat, sum(allg), atol=1e-6)
    # third step: parameter 5 is applied twice per step (two contributions from different tape entries): its bucket must not leave
    # after the first one, and the launch order must still be identical on every rank
    gflat.copy_(local)
    bk.reset()
    expected = {i: (2 if i == 5 else 1) for i in range(len(sizes))}
    counts, seen5 = {}, []
    for i in [6, 5, 4, 3, 5, 2, 1, 0]:
        counts[i] = counts.get(i, 0) + 1
        bk.launch_ready(counts, expected=expected)
        if i == 5:
            seen5.append(any(5 in bk.buckets[b][2] for b in bk.order))
    bk.finish()
    ok_multi = seen5 == [False, True] and torch.allclose(gflat, sum(allg), atol=1e-6)
    # third step again through the O(1) form the trainer uses (expect() arms per-bucket counters, every gradient sink reports contribution()):
    # same launch order as the scan, nothing leaves early, every element summed once
    order_scan = list(bk.order)
    gflat.copy_(local)
    bk.reset()
    bk.expect(expected)
    counts, seen5 = {}, []
    for i in [6, 5, 4, 3, 5, 2, 1, 0]:
        counts[i] = counts.get(i, 0) + 1
        bk.contribution(i)
        bk.launch_ready(counts, expected=expected)
        if i == 5:
            seen5.append(any(5 in bk.buckets[b][2] for b in bk.order))
    bk.finish()
    ok_multi = ok_multi and seen5 == [False, True] and list(bk.order) == order_scan and torch.allclose(gflat, sum(allg), atol=1e-6)
    # fourth step: the record / replay flow of a captured step (Trainer._capture_segments): launches are noted while recording, issued by launch_async
    gflat.copy_(local)
    bk.reset()
    bk.record = []
    for i in reversed(range(len(sizes))):
        bk.launch_ready(set(range(i, len(sizes))))
    bk.finish()
    recorded, bk.record = list(bk.record), None
    ok_rec = recorded == order and torch.equal(gflat, local)            # nothing was sent while recording
    bk.reset()
    for b in recorded:
        bk.launch_async([b])
    bk.wait()
    ok_rec = ok_rec and torch.allclose(gflat, sum(allg), atol=1e-6)
    # fifth step: bf16 on the wire (PN2_DP_WIRE=bf16): half the bytes, a bf16 sum widened back into the fp32 arena
    g16 = local.clone()
    bw = GradBuckets(g16, spans, bucket_bytes=4 * 4000, wire_dtype=torch.bfloat16)
    bw.reduce_all()
    ref16 = sum(a.bfloat16().float() for a in allg)
    ok_wire = float((g16 - ref16).abs().max()) <= 2 ** -7 * float(ref16.abs().max()) and g16.dtype == torch.float32
    q.put((rank, order, ok_sum, ok_mean, ok_rest and ok_multi and ok_rec and ok_wire))
    dist.destroy_process_group()


def test_bucketed_allreduce_two_ranks():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(30)
    for rank, order, ok_sum, ok_mean, ok_rest in res:
        assert ok_sum and ok_mean and ok_rest, (rank, ok_sum, ok_mean, ok_rest)
        assert order == sorted(order, reverse=True), f"buckets must launch tail-first, got {order}"
    assert res[0][1] == res[1][1], "ranks must launch collectives in the same order"


def test_buckets_are_cut_from_the_tail_with_a_small_head_bucket():
    """Bucket layout (no communication): full buckets from the tail of the arena (what backward completes first), the head of the arena - the bucket that
    leaves last, with nothing left to overlap its all-reduce - at most head_bytes; every element in exactly one bucket, in arena order."""
    sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
    from pn2.dp import GradBuckets
    sizes = [300, 700, 1000, 2000, 4000, 1000, 3000, 2500, 500]
    spans, off = [], 0
    for i, n in enumerate(sizes):
        spans.append((i, off, n)); off += n
    bk = GradBuckets(torch.zeros(off), spans, bucket_bytes=4 * 5000, head_bytes=4 * 1200)
    b = bk.buckets
    assert b[0][0] == 0 and b[-1][1] == off and all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1))
    assert [sorted(k[2]) for k in b] == [[0, 1], [2, 3], [4, 5], [6, 7, 8]], [sorted(k[2]) for k in b]      # tail: 6000 >= 5000; 5000; rest 4000 -> head 1000 <= 1200 + 3000
    assert (b[0][1] - b[0][0]) * 4 <= 4 * 1200
    # everything below one bucket (15000 elements < 100000): the head bucket (spans that fit bucket_bytes / 5 = 20000 elements ... all of them) - i.e. ONE bucket
    one = GradBuckets(torch.zeros(off), spans, bucket_bytes=4 * 100000)
    assert [sorted(k[2]) for k in one.buckets] == [list(range(len(sizes)))]
    # below one bucket but above the head allowance: a head of <= head_bytes plus ONE rest bucket, in arena order
    two = GradBuckets(torch.zeros(off), spans, bucket_bytes=4 * 100000, head_bytes=4 * 1200)
    assert [sorted(k[2]) for k in two.buckets] == [[0, 1], list(range(2, len(sizes)))] and (two.buckets[0][1] - two.buckets[0][0]) * 4 <= 4 * 1200
