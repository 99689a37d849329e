"""bench.py's own rank launcher on the CPU: `--gpus N` must start N ranks (or refuse), never run fewer and
report fewer.  The engine needs a GPU, so the ranks run the exchange rehearsal (gloo, no engine)."""
import json
import os
import subprocess
import sys
import time
import uuid

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=300)


def _want_min():
    surface = ((np.arange(4096, dtype=np.float64) * 37.0 + 55.0) % 101.0) + 1.0
    return int(np.argmin(surface)), float(surface.min())


def test_launcher_starts_two_ranks_over_gloo():
    p = _run(["--gpus", "2", "--rehearse-cpu"])
    assert p.returncode == 0, p.stderr
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, "rank 0 alone prints: %r" % p.stdout
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["points_per_rank"] == [2048, 2048] and out["surface_ok"]
    assert (out["best_idx"], out["best_loss"]) == _want_min()


def test_launcher_three_ranks_uneven_shares():
    p = _run(["--gpus", "3", "--rehearse-cpu"])
    assert p.returncode == 0, p.stderr
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 3 and out["points_per_rank"] == [1366, 1365, 1365] and out["surface_ok"]
    assert (out["best_idx"], out["best_loss"]) == _want_min()


def _stragglers(mark):
    """Processes still alive that carry the test's mark in their environment (ranks the launcher left behind)."""
    import psutil
    left = []
    for p in psutil.process_iter():
        try:
            if p.pid != os.getpid() and p.environ().get("EPI_TEST_MARK") == mark and p.status() != psutil.STATUS_ZOMBIE:
                left.append(p.pid)
        except (psutil.Error, OSError):
            pass
    return left


def test_a_rank_that_dies_at_start_ends_the_run_in_seconds():
    """One rank of three exits before the rendezvous: without supervision the other two would wait in
    init_process_group for the full timeout (here 170 s).  The launcher must name the rank, stop the others and
    return its status at once."""
    mark = uuid.uuid4().hex
    t0 = time.monotonic()
    p = _run(["--gpus", "3", "--rehearse-cpu", "--fail-rank", "1", "--rendezvous-timeout", "170"], {"EPI_TEST_MARK": mark})
    took = time.monotonic() - t0
    assert p.returncode == 7, (p.returncode, p.stderr)
    assert took < 20.0, "launcher took %.1f s to give up" % took
    assert "rank 1 of 3 failed (exit status 7)" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")], "no result line from a failed run"
    assert _stragglers(mark) == []


def test_a_rank_that_dies_inside_the_exchange_ends_the_run_in_seconds():
    mark = uuid.uuid4().hex
    t0 = time.monotonic()
    p = _run(["--gpus", "3", "--rehearse-cpu", "--fail-rank", "2", "--fail-at", "collective", "--rendezvous-timeout", "170"],
             {"EPI_TEST_MARK": mark})
    took = time.monotonic() - t0
    assert p.returncode != 0 and took < 20.0, (p.returncode, took, p.stderr)
    assert "failed (" in p.stderr and "stopping the other" in p.stderr
    assert _stragglers(mark) == []


def test_the_launcher_process_never_imports_torch():
    """The parent starts the ranks, so it must never load the HIP runtime: it imports neither torch nor the engine.
    `-X importtime` lists every module the PARENT imports on stderr (plain child processes do not inherit -X)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["EPI_BENCH_LAUNCHER_REPORT"] = "1"
    p = subprocess.run([sys.executable, "-X", "importtime", BENCH, "--gpus", "2", "--rehearse-cpu"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    imported = [l.rsplit("|", 1)[1].strip() for l in p.stderr.splitlines() if l.startswith("import time:") and "|" in l]
    assert "argparse" in imported, "importtime output missing"
    bad = [m for m in imported if m.split(".")[0] in ("torch", "numpy", "epidemicsimulator_amd", "ctypes")]
    assert bad == [], bad
    assert "modules loaded in the parent: none of torch/numpy/engine" in p.stderr
    # the refusal path (too few devices) is decided from sysfs, again without torch
    p = subprocess.run([sys.executable, "-X", "importtime", BENCH, "--gpus", "64", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "needs 64 HIP devices" in p.stderr
    assert not [l for l in p.stderr.splitlines() if l.startswith("import time:") and l.rsplit("|", 1)[1].strip() == "torch"]


def test_refuses_more_ranks_than_devices():
    """No GPU here: `--gpus 2` must fail loudly instead of running one rank and printing n_gpus: 1."""
    sys.path.insert(0, ROOT)
    import bench
    if bench.visible_gpu_count() >= 2:
        import pytest
        pytest.skip("this machine has two devices")
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert p.returncode != 0
    assert "needs 2 HIP devices" in p.stderr and "{" not in p.stdout


def test_refuses_world_size_mismatch():
    p = _run(["--gpus", "2", "--rehearse-cpu"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert p.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in p.stderr
    p = _run(["--gpus", "1", "--rehearse-cpu"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert p.returncode != 0 and "WORLD_SIZE=2 but --gpus 1" in p.stderr
