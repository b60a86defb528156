"""The RCCL path executed on hardware: a fresh child process (tests/rccl_child.py) forms a one-rank "nccl"
process group on the test box's GPU and drives ``parallel.mean_log_prob(force_collective=True)``, a data-parallel flat
training step and a data-parallel image training step through it.  This file sorts FIRST in the suite on purpose: the child
must be started before this pytest process touches the GPU (a process that has initialised the GPU must not start another
program on this pool); when something already did, the test says so and skips."""
import json
import os
import subprocess
import sys
import tempfile

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_rccl_world1_mean_log_prob_and_gradient_allreduce():
    if torch.cuda.is_initialized():
        pytest.skip("this process has already initialised the GPU: the RCCL child must be started first (run the whole suite, "
                    "or this file alone)")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "rccl.json")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_child.py"), out], env=env, capture_output=True, text=True,
                           timeout=420)
        assert os.path.isfile(out), (p.returncode, p.stderr[-2000:])
        res = json.load(open(out))
    assert res.get("ok"), res.get("trace") or res
    assert res["backend"] == "nccl" and res["native_lib"]
    # mean_log_prob: exactly one collective per call when forced (none by default in a one-rank group), results identical
    assert res["mean_collectives_per_call"] == 1 and res["mean_collectives_default_world1"] == 0
    assert res["mean_equal"] and res["mean_golden_rel"] < 1e-5
    # flat training: one all-reduce of the gradient arena; same gradients as the non-distributed step (x B / B)
    assert res["flat_train_collectives"] == 1 and res["flat_train_grad_max_rel"] < 1e-6, res
    # image training: one all-reduce per step over the persistent flat buffer the gradients are views of
    assert res["image_train_collectives"] == 2 and res["image_grads_bound"], res
    assert res["image_train_grad_max_rel_vs_reference"] < 5e-5, res
