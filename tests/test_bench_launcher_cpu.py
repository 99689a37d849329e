"""bench.py's own rank launcher on the CPU: `--gpus N` must start N ranks (or refuse), never run fewer and
report fewer.  The engine needs a GPU, so the ranks run the exchange rehearsal (gloo, no engine)."""
import json
import os
import subprocess
import sys

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


def test_refuses_more_ranks_than_devices():
    """No GPU here: `--gpus 2` must fail loudly instead of running one rank and printing n_gpus: 1."""
    import torch
    if torch.cuda.device_count() >= 2:
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
