"""bench.py --gpus N starts N rank processes itself ("--gpus is dead").  Here: the launcher / sharding /
collective plumbing with 2 ranks on gloo (CPU; `--device cpu` marks the line as a plumbing test, not a measurement).
On the 8-GPU node the same code path runs `nccl` (= RCCL)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--device", "cpu", "--dim", "16", "--blocks", "2", "--hidden", "16", "--steps", "2", "--warmup", "1"]


def _run(extra):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                       # exactly ONE JSON line (rank 0)
    return json.loads(lines[0])


def test_gpus_flag_launches_that_many_ranks_weak_scaling():
    one = _run(["--gpus", "1", "--batch", "64"] + SMALL)
    two = _run(["--gpus", "2", "--batch", "64"] + SMALL)
    assert one["n_gpus"] == 1 and one["world_size"] == 1 and one["backend"] is None
    assert "no collective" in one["config"]["parallelism"]
    assert two["n_gpus"] == 2 and two["world_size"] == 2 and two["backend"] == "gloo"
    assert two["scaling"] == "weak" and two["config"]["rows_per_gpu"] == 64 and two["config"]["global_rows"] == 128
    assert "all-reduce of 2 fp64 scalars" in two["config"]["parallelism"]
    assert "PLUMBING TEST" in two["metric"]
    # mean over both ranks' shards differs from rank 0's own mean (the all-reduce happened): ranks draw different rows
    assert two["mean_log_prob"] != one["mean_log_prob"]


def test_cfg3_strong_sharding_and_cfg5_sampling_modes():
    c3 = _run(["--gpus", "2", "--config", "cfg3"] + SMALL)
    assert c3["scaling"] == "strong" and c3["config"]["global_rows"] == 262144 and c3["config"]["rows_per_gpu"] == 131072
    c5 = _run(["--gpus", "2", "--config", "cfg5", "--no-kernel-timing"] + SMALL)
    assert c5["scaling"] == "strong" and c5["config"]["global_rows"] == 1000000 and c5["config"]["rows_per_gpu"] == 500000
    assert "no collective" in c5["config"]["parallelism"] and c5["metric"].count("sample()") == 1


def test_eight_ranks_shard_cfg3_and_cfg5_as_the_node_will():
    """the 8-GPU node's launch (--gpus 8; here 8 gloo ranks on the CPU): the process group sees 8 ranks, cfg3 / cfg5 shard
    their totals into 32768 / 125000 rows per rank, the weak-scaling default keeps the per-rank batch"""
    env_threads = os.environ.get("OMP_NUM_THREADS")
    os.environ["OMP_NUM_THREADS"] = "1"
    try:
        w = _run(["--gpus", "8", "--batch", "32"] + SMALL)
        assert w["n_gpus"] == 8 and w["world_size"] == 8 and w["backend"] == "gloo" and w["scaling"] == "weak"
        assert w["config"]["rows_per_gpu"] == 32 and w["config"]["global_rows"] == 256
        assert "dp8" in w["config"]["parallelism"]
        c3 = _run(["--gpus", "8", "--config", "cfg3"] + SMALL)
        assert c3["world_size"] == 8 and c3["scaling"] == "strong"
        assert c3["config"]["global_rows"] == 262144 and c3["config"]["rows_per_gpu"] == 32768
        c5 = _run(["--gpus", "8", "--config", "cfg5", "--no-kernel-timing"] + SMALL)
        assert c5["world_size"] == 8 and c5["config"]["global_rows"] == 1000000 and c5["config"]["rows_per_gpu"] == 125000
    finally:
        if env_threads is None:
            os.environ.pop("OMP_NUM_THREADS", None)
        else:
            os.environ["OMP_NUM_THREADS"] = env_threads
