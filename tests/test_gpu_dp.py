"""Two data-parallel ranks on ONE GPU (gloo backend moving the CUDA gradient buckets): exercises exactly the trainer control flow that the
8-GPU RCCL run uses — bucket hooks inside backward, deferred wgrad/reduce flush before each bucket, the split hipGraphs with the all-reduce
between them — and checks the DP parity statement of SURVEY 8(e): the 2-rank step on two N-image shards == a single process that runs
the two shards itself and averages the gradients."""
import os, sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(kind="res2net", autotune=False):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "pranet-v2_amd"))
    os.environ["PN2_NO_PRETRAINED"] = "1"
    # autotune off: every process launches the same kernels, so the 2-rank result can be compared tightly with the single process.  autotune on (the
    # SHIPPED configuration): every process times its own candidates and may pick other tiles; replicas still stay bit-identical (they apply the
    # same all-reduced gradient), only the comparison with the separately tuned single process needs slack on this ill-conditioned 2-image problem
    os.environ["PN2_AUTOTUNE"] = "1" if autotune else "0"
    import pn2
    from oracle import weights as W
    pn2.set_compute_dtype("bf16")
    if kind == "emcad":
        from lib.networks import EMCADNet
        model = EMCADNet(num_classes=9, kernel_sizes=[1, 3, 5], expansion_factor=2, dw_parallel=True, add=True, lgag_ks=3, activation="relu6", encoder="pvt_v2_b2",
                         pretrain=False, dual=True)
        model.load_state_dict(W.make_state_dict(W.manifest_emcadnet(9), seed=5), strict=True)
        model.backbone.reset_drop_path(0.0)
    else:
        from lib.pranet import PraNet_V2
        model = PraNet_V2(num_class=1)
        model.load_state_dict(W.make_state_dict(W.manifest_pranet_v2(1), seed=0), strict=True)
    return model.cuda().train(), W


def _batch(kind, W, rank):
    if kind == "emcad":
        g = torch.Generator(device="cpu").manual_seed(900 + rank)
        x = torch.randn(2, 1, 64, 64, generator=g).cuda()
        lab = torch.randint(0, 9, (2, 64, 64), generator=g).cuda()
        return x, (lab, torch.stack([(lab != k).float() for k in range(9)], 1))
    x, m = W.synthetic_batch(2, 96, seed=50 + rank)
    return x.cuda(), m.cuda()


def _trainer(kind, model, pg):
    from pn2.trainer import Trainer
    if kind == "emcad":
        # weights applied more than once per step (CAB fc1 / fc2, the shared sab conv) + 1 MB buckets: a bucket must wait for the LAST contribution
        return Trainer(model, lr=1e-4, clip=None, weight_decay=1e-4, loss="mutation", hot=model.hot_parameters(True), process_group=pg, bucket_bytes=1 << 20)
    return Trainer(model, lr=1e-4, clip=0.5, process_group=pg, bucket_bytes=8 << 20)


def _worker(rank, world, port, q, kind, autotune):
    try:
        _worker_body(rank, world, port, q, kind, autotune)
    except Exception:
        import traceback
        q.put((rank, "ERROR", traceback.format_exc()))


def _worker_body(rank, world, port, q, kind, autotune):
    import faulthandler, signal
    faulthandler.dump_traceback_later(240, exit=False)          # a rank that hangs says where (the parent shows the workers' stderr)
    faulthandler.register(signal.SIGUSR1, all_threads=True)      # ... and at once when the parent's queue times out
    say = lambda msg: print(f"[dp worker {rank}] {msg}", file=sys.stderr, flush=True)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model, W = _setup(kind, autotune)
    tr = _trainer(kind, model, dist.group.WORLD)
    x, m = _batch(kind, W, rank)
    say("first pass")
    loss = tr.forward_backward(x, m)          # first pass: counts the gradient contributions per parameter, buckets leave at the end
    torch.cuda.synchronize()
    g1 = tr.gflat.clone()
    tr.optimizer_step()
    p1 = tr.flat.clone()
    say("second pass")
    tr.step(x, m)                             # second pass: bucket hooks inside backward
    torch.cuda.synchronize()
    order = list(tr.buckets.order)
    say("capture")
    tr.capture(x, m, warmup=2)                # 2 more eager steps, then the split graphs
    say("replay")
    tr.replay(); tr.replay()
    torch.cuda.synchronize()
    say("done")
    faulthandler.cancel_dump_traceback_later()
    segs = [bs for _, bs in tr._cur.segments]      # the captured step is a chain of hipGraphs cut where buckets leave
    assert len(segs) >= 3 and [b for bs in segs for b in bs] == order, (segs, order)
    q.put((rank, g1.cpu().numpy(), p1.cpu().numpy(), tr.flat.clone().cpu().numpy(), order, float(loss[-1])))     # numpy: no fd passing
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,autotune", [("res2net", False), ("res2net", True), ("emcad", False)])
def test_two_ranks_match_single_process_with_averaged_gradients(kind, autotune):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    world = 2
    ctx = mp.get_context("spawn")
    import queue as _queue
    for attempt in (0, 1):
        q = ctx.Queue()
        port = 29600 + (os.getpid() + 7 * len(kind) + int(autotune) + 97 * attempt) % 2000
        ps = [ctx.Process(target=_worker, args=(r, world, port, q, kind, autotune)) for r in range(world)]
        for p in ps:
            p.start()
        res = []
        try:
            for _ in range(world):
                r = q.get(timeout=int(os.environ.get("PN2_TEST_DP_TIMEOUT", "300")))
                if isinstance(r[1], str) and r[1] == "ERROR":          # fail fast with the worker's traceback: the other rank is waiting in a collective that will never complete
                    raise AssertionError(f"rank {r[0]} failed:\n{r[2]}")
                res.append(r)
        except _queue.Empty:
            # two processes time-slicing one GPU behind gloo: on two boxes of the pool this test stalled in its first pass (4 runs in a row there) with code that
            # passed before and after on every other box (~15 runs).  One re-run on a TIME-OUT only (PN2_TEST_DP_RETRY=0: none); an error in a worker fails at once.
            # The stall is not silent: the workers dump their stacks (faulthandler, SIGUSR1) and the re-run shows up as a warning in the pytest summary.
            import signal, warnings
            for p in ps:
                if p.is_alive():
                    os.kill(p.pid, signal.SIGUSR1)          # faulthandler.register in _worker: every thread's stack to stderr
            import time as _t
            _t.sleep(2)
            if attempt == 1 or os.environ.get("PN2_TEST_DP_RETRY", "1") != "1":
                raise
            warnings.warn(f"two-rank data-parallel worker ({kind}, autotune={autotune}) TIMED OUT and was re-run once - worker stacks are on stderr; a repeat of this is a deadlock, not a slow box")
            continue
        finally:
            for p in ps:
                p.join(5 if len(res) < world else 60)
                if p.is_alive():
                    p.kill()
        break
    res.sort(key=lambda r: r[0])
    (_, g_a, p_a, pf_a, order_a, _), (_, g_b, p_b, pf_b, order_b, _) = res
    g_a, p_a, pf_a, g_b, p_b, pf_b = (torch.from_numpy(t) for t in (g_a, p_a, pf_a, g_b, p_b, pf_b))
    assert order_a == order_b and len(order_a) >= 3 and order_a[-1] == 0, "ranks must launch the bucket collectives in the same order, head bucket last"
    assert torch.equal(g_a, g_b), "both ranks must hold the same summed gradient"
    assert torch.equal(p_a, p_b) and torch.equal(pf_a, pf_b), "replicas must stay bit-identical (eager steps, then 2 eager + 2 replayed steps)"
    # single process: the two shards one after the other, gradients averaged by hand, same clamp+Adam
    saved = {k: os.environ.get(k) for k in ("PN2_AUTOTUNE", "PN2_NO_PRETRAINED")}
    try:
        model, W = _setup(kind, autotune)
        tr = _trainer(kind, model, None)
    finally:                      # _setup's switches are for the worker processes; this process runs the rest of the suite
        for k, v in saved.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    gs = []
    for rank in range(world):
        x, m = _batch(kind, W, rank)
        tr.forward_backward(x, m)
        gs.append(tr.gflat.clone())
    gsum = gs[0] + gs[1]
    torch.cuda.synchronize()
    ref = (gsum * 0.5).cpu()
    g_cmp = g_a * 0.5                                           # before the optimizer kernel the arena holds the plain all-reduced sum
    if not autotune:            # same kernels in the same order -> normally exact
        assert float((g_cmp - ref).norm() / ref.norm()) < 2e-2
    # autotune on: this process tuned its own tiles; on the 2-image fixture bf16 rounding noise is amplified to O(1) in the backbone gradients
    # (tests/test_gpu_parity.py::test_model_bf16_vs_reference_f64), so only the replica-consistency statements above and the Adam bound below hold
    tr.gflat.copy_(gsum * 0.5)
    tr.optimizer_step()
    torch.cuda.synchronize()
    assert float((p_a - tr.flat.cpu()).abs().max()) < 3e-4      # one Adam step of size lr=1e-4: sign flips of ~0 gradients move a weight by <= 2e-4


# ---------------------------------------------------------------------------------------------------------------------------------------
# RCCL itself (torch.distributed backend "nccl" IS RCCL on ROCm).  Two ranks cannot share one GPU under RCCL ("duplicate GPU"), so the real
# backend is exercised with a ONE-rank communicator and Trainer(force_dp=True): eager bucket hooks issuing ncclAllReduce on RCCL's stream next to
# the backward kernels, the collectives captured into the step's hipGraph (and, as the fallback, _capture_segments + launch_async between hipGraph
# segments) under the thread-local capture mode with RCCL's watchdog thread alive, PN2_DP_WIRE=bf16 (temporary bf16 buffers), two trainers in one process.  A 1-rank sum is the
# identity, so the result must be BIT-IDENTICAL to the trainer without a process group - any ordering bug between RCCL's stream and the compute
# stream (a bucket sent before its gradients are complete, an optimizer replay that does not wait for the collective) shows as a difference.
def _rccl_worker(port, q, wire, captured=True):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ["PN2_DP_WIRE"] = wire
        os.environ["PN2_DP_CAPTURE"] = "1" if captured else "0"      # (read when pn2.trainer is imported, below)
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        from pn2.trainer import Trainer
        model, W = _setup("res2net", False)
        x, m = _batch("res2net", W, 0)
        tr = Trainer(model, lr=1e-4, clip=0.5, process_group=dist.group.WORLD, bucket_bytes=8 << 20, force_dp=True)
        assert tr.dp and tr.world == 1
        tr.step(x, m); tr.step(x, m)                  # pass 1 counts contributions, pass 2 sends buckets from inside backward (eager RCCL calls)
        torch.cuda.synchronize()
        order = list(tr.buckets.order)
        tr.capture(x, m, warmup=2)
        tr.replay(); loss = tr.replay()
        torch.cuda.synchronize()
        if captured:        # ONE graph holds the step, its collectives (issued in the eager order) and the optimizer
            assert tr._cur.segments is None and tr._cur.graph_opt is None and list(tr.buckets.order) == order, (tr.buckets.order, order)
            segs = [order]
        else:
            segs = [bs for _, bs in tr._cur.segments]
            assert len(segs) >= 3 and [b for bs in segs for b in bs] == order, (segs, order)
        dp_flat = tr.flat.clone(); dp_loss = loss.clone()
        # the same six steps without a process group
        model2, _ = _setup("res2net", False)
        tr2 = Trainer(model2, lr=1e-4, clip=0.5)
        assert not tr2.dp
        tr2.step(x, m); tr2.step(x, m)
        tr2.capture(x, m, warmup=2)
        tr2.replay(); loss2 = tr2.replay()
        torch.cuda.synchronize()
        if wire == "fp32":
            same = bool(torch.equal(dp_flat, tr2.flat)) and bool(torch.equal(dp_loss, loss2))
            info = float((dp_flat - tr2.flat).abs().max())
        else:           # gradients rounded to bf16 on the wire: Adam's update direction is (almost) unchanged, weights agree to ~lr per step
            info = float((dp_flat - tr2.flat).abs().max())
            same = info < 6 * 2e-4 and abs(float(dp_loss[-1]) - float(loss2[-1])) < 5e-2 * abs(float(loss2[-1]))
        # a second data-parallel trainer in the same process (its own buckets, segments and collectives on the same communicator)
        model3, _ = _setup("res2net", False)
        tr3 = Trainer(model3, lr=1e-4, clip=0.5, process_group=dist.group.WORLD, bucket_bytes=8 << 20, force_dp=True)
        tr3.step(x, m); tr3.step(x, m)
        tr3.capture(x, m, warmup=2)
        tr3.replay(); tr3.replay(); tr.replay()
        torch.cuda.synchronize()
        same3 = bool(torch.equal(tr3.flat, dp_flat)) if wire == "fp32" else True
        q.put(("OK", same, same3, info, len(segs), len(tr.buckets.buckets), dist.get_backend()))
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put(("ERROR", traceback.format_exc()))


@pytest.mark.parametrize("captured", [True, False])
@pytest.mark.parametrize("wire", ["fp32", "bf16"])
def test_rccl_one_rank_dp_path_is_bit_identical_to_local(wire, captured):
    """captured: the bucket all-reduces are captured INTO the step's hipGraph (the default on RCCL); else: c10d asynchronous collectives between a chain
    of graph segments (the fallback, PN2_DP_CAPTURE=0)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31600 + (os.getpid() + (3 if wire == "bf16" else 0) + (7 if captured else 0)) % 2000
    p = ctx.Process(target=_rccl_worker, args=(port, q, wire, captured))
    p.start()
    import queue as _queue, time as _time
    res, t_end = None, _time.monotonic() + 900
    while res is None:                       # a worker that dies (e.g. c10d's watchdog terminating the process) must fail the test at once, not after the time-out
        try:
            res = q.get(timeout=2)
        except _queue.Empty:
            assert p.is_alive() or not q.empty(), f"the RCCL worker exited with code {p.exitcode} without a result"
            assert _time.monotonic() < t_end, "the RCCL worker timed out"
    p.join(60)
    assert res[0] == "OK", res[1]
    _, same, same3, info, nseg, nb, backend = res
    print(f"RCCL 1-rank [{wire}]: backend {backend}, {nb} buckets, {nseg} graph segments, max |w_dp - w_local| {info:.2e}")
    assert backend == "nccl"
    assert same, f"data-parallel path over RCCL differs from the local trainer (max weight difference {info:.3e})"
    assert same3, "a second data-parallel trainer in the same process diverged"
