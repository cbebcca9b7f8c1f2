"""bench.py's own launcher (`python bench.py --gpus N` without torchrun): N ranks are started before anything touches
the GPU, rendezvous on 127.0.0.1, rank 0 prints the one JSON line with n_gpus = N.  The GPU workload is replaced by
bench.py's stub (VOFOD_BENCH_STUB=1: gloo group + all-gather of the rank ids), so this runs on the CPU."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _run(extra_env, *argv):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=300)


def test_gpus_flag_starts_that_many_ranks():
    r = _run({"VOFOD_BENCH_STUB": "1"}, "--gpus", "2", "--steps", "2", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # one JSON line, from rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["gpus_arg"] == 2
    assert d["ranks_seen"] == [0, 1]
    # every rank that owns its GPU asks the runtime for 16 hardware queues before anything initialises it (bench.py main())
    assert d["hw_queues_seen"] == [16, 16]


def test_the_one_gpu_rehearsal_keeps_the_default_queues():
    r = _run({"VOFOD_BENCH_STUB": "1"}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--rehearse-one-gpu", "--backend", "gloo")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["hw_queues_seen"] == [0, 0]


def test_launcher_env_wins_over_the_flag():
    # under torch.distributed.run the launcher's WORLD_SIZE is authoritative: no second level of children
    r = _run({"VOFOD_BENCH_STUB": "1", "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29631"}, "--gpus", "8")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["ranks_seen"] == [0]


def test_failed_rank_fails_the_launch():
    # no GPU here: every real rank exits with bench.py's "needs an MI355X" message; the launcher reports the failure
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("needs a machine without a GPU")
    r = _run({}, "--gpus", "2", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0
    assert "needs an MI355X" in r.stderr
