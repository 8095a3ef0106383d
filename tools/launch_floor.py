"""Per-node cost of DEPENDENT kernel nodes inside one hipGraph on this part (GPU box): N tiny kernels in a chain, replayed; the same with the chain
forked over 2 / 4 independent branches.  python tools/launch_floor.py"""
import torch, time
dev = "cuda"
def bench(build, reps=20):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        build()                      # warm
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            build()
    g.replay(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e6
N = 1000
for numel in (256, 65536, 4 << 20):
    x = torch.zeros(numel, device=dev)
    def chain():
        for _ in range(N): x.add_(1.0)
    us = bench(chain)
    print(f"chain of {N} dependent add_ on {numel:8d} floats: {us / N:6.2f} us per node   ({numel * 8 / (us / N) / 1e3:8.1f} GB/s)")
for br in (2, 4):
    xs = [torch.zeros(256, device=dev) for _ in range(br)]
    def fork():
        cur = torch.cuda.current_stream()
        ss = [torch.cuda.Stream() for _ in range(br)]
        e0 = torch.cuda.Event(); e0.record(cur)
        for b, st in enumerate(ss):
            st.wait_event(e0)
            with torch.cuda.stream(st):
                for _ in range(N // br): xs[b].add_(1.0)
            e = torch.cuda.Event(); e.record(st); cur.wait_event(e)
    us = bench(fork)
    print(f"{br} independent branches, {N} tiny nodes in total: {us / N:6.2f} us per node")
