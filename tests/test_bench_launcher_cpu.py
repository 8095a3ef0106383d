"""`bench.py --gpus N` starts its N ranks itself (the reference's counterpart is the one-line nn.DataParallel wrap of
multiclass_seg/EMCAD/trainer.py:75-77).  Driven here without a GPU: `--backend gloo --dry-run` runs everything of a multi-rank run except the model - the child
launcher, the rendezvous on 127.0.0.1, the gradient buckets of pn2/dp.py, the barrier + max-over-ranks timing protocol, rank 0's JSON line."""
import io, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _env():
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None); e.pop("RANK", None); e.pop("LOCAL_RANK", None)
    e["OMP_NUM_THREADS"] = "1"
    return e


def test_gpus_2_launches_two_ranks_and_reports_them():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout            # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dp"]["nccl_ranks"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["dry_run"] is True and out["value"] is None and out["dp"]["sum_correct"] is True and out["scaling"] == "weak"


def test_launcher_function_streams_child_output_and_returns_its_status():
    import bench
    os.environ.pop("WORLD_SIZE", None)
    buf = io.StringIO()
    rc = bench.launch_ranks(2, ["--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "2", "--warmup", "0"], backend="gloo", timeout=300, out=buf)
    assert rc == 0
    out = json.loads([ln for ln in buf.getvalue().splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 2
    # a failing rank is a non-zero status of the launcher, not a hang and not a silent success (argparse error in every rank: status != 0)
    rc = bench.launch_ranks(2, ["--gpus", "2", "--backend", "gloo", "--no-such-flag"], backend="gloo", timeout=300, out=io.StringIO())
    assert rc != 0


def test_refuses_more_ranks_than_gpus_without_starting_anything():
    # no GPU in the CPU container / one on a GPU box: 64 ranks can never be served
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1"], capture_output=True, text=True, timeout=120, env=_env())
    assert r.returncode == 2 and "needs 64 GPUs" in r.stderr and r.stdout.strip() == ""


def test_hung_ranks_are_killed_after_the_launch_timeout():
    import bench
    # the ranks wait on a rendezvous that can never complete (the test passes a world of 2 but only one process would join a 3-rank store): simulate a hang
    # cheaply with a child that sleeps - the launcher's own timeout path is what is under test
    import unittest.mock as um
    real = subprocess.Popen

    def sleeper(cmd, **kw):
        return real([sys.executable, "-c", "import time; time.sleep(60)"], **kw)
    with um.patch("subprocess.Popen", sleeper):
        rc = bench.launch_ranks(2, [], backend="gloo", timeout=1.0, out=io.StringIO())
    assert rc == 124


def test_refuses_to_launch_ranks_under_a_profiler():
    # rocprofv3 -- python3 bench.py --gpus 8: the profiler's preloaded library has initialised the GPU in the would-be launcher: status 2, nothing started
    for var, val in (("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so"), ("ROCP_TOOL_LIBRARIES", "/opt/rocm/lib/librocprofiler-sdk-tool.so"), ("ROCPROFILER_LOG_LEVEL", "1")):
        e = _env()
        import bench
        assert not bench.profiler_attached(e)
        e[var] = val
        assert bench.profiler_attached(e)
    e = _env()
    e["ROCPROF_OUTPUT_PATH"] = "/tmp/x"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "1"], capture_output=True, text=True, timeout=120, env=e)
    assert r.returncode == 2 and "profiled process" in r.stderr and r.stdout.strip() == ""


def test_gpu_count_comes_from_sysfs_not_from_hip(tmp_path):
    import bench
    for i, simd in enumerate((0, 256, 256, 0)):          # two CPU nodes, two GPU nodes
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text(f"cpu_cores_count {0 if simd else 64}\nsimd_count {simd}\nmem_banks_count 1\n")
    assert bench.count_gpus_sysfs(str(tmp_path)) == 2
    assert bench.count_gpus_sysfs(str(tmp_path / "missing")) in (0, None)


def test_sigterm_to_the_launcher_takes_the_rank_group_down(tmp_path):
    # a harness that SIGTERMs bench.py must not leave torchrun + ranks behind in their own session
    import signal, time
    marker = tmp_path / "pids"
    code = (
        "import sys, os, io, subprocess, unittest.mock as um\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "real = subprocess.Popen\n"
        "def sleeper(cmd, **kw):\n"
        "    p = real([sys.executable, '-c', 'import time; time.sleep(120)'], **kw)\n"
        f"    open({str(marker)!r}, 'w').write(str(p.pid))\n"
        "    return p\n"
        "with um.patch('subprocess.Popen', sleeper):\n"
        "    sys.exit(bench.launch_ranks(2, [], backend='gloo', timeout=100.0, out=io.StringIO()))\n")
    p = subprocess.Popen([sys.executable, "-c", code], env=_env())
    for _ in range(200):
        if marker.exists() and marker.read_text().strip():
            break
        time.sleep(0.1)
    child = int(marker.read_text())
    os.kill(child, 0)                      # alive
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=30) == 128 + signal.SIGTERM
    for _ in range(100):
        try:
            os.kill(child, 0)
        except ProcessLookupError:
            break
        # (a zombie still answers kill 0 until it is reaped by init: look at its state)
        try:
            if open(f"/proc/{child}/stat").read().split()[2] == "Z":
                break
        except OSError:
            break
        time.sleep(0.1)
    else:
        raise AssertionError("the launcher's child group survived SIGTERM")
