"""Multi-process data-parallel path on CPU (gloo, world_size 2): batch sharding + the single scalar
all-reduce reproduce the single-process mean log_prob; sharded sampling geometry."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        from golden_util import load_case
        from model_util import build_flow
        from usflows_amd.parallel import mean_log_prob, shard_rows
        spec, sd, a = load_case("synth_d16_k4_hh2_conj_laplace")
        flow = build_flow(spec, sd)
        x = a["x"]                                   # 48 rows -> uneven-friendly split below
        lo, hi = shard_rows(x.shape[0] - 1, rank, world)      # 47 rows: 24 + 23
        mean, lp = mean_log_prob(flow, x[lo:hi])
        q.put((rank, float(mean), lp.double().sum().item(), hi - lo))
    finally:
        dist.destroy_process_group()


def test_mean_log_prob_allreduce_world2():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from golden_util import load_case
    _, _, a = load_case("synth_d16_k4_hh2_conj_laplace")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = sum(r[3] for r in res)
    assert n == 47 and [r[3] for r in res] == [24, 23]
    expect = a["log_prob64"][:47].mean().item()
    for _, mean, _, _ in res:                        # every rank holds the global mean
        assert abs(mean - expect) < 1e-5 * abs(expect)
    assert abs(sum(r[2] for r in res) / n - expect) < 1e-5 * abs(expect)


def _worker8(rank, world, port, q, n_rows):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        from golden_util import load_case
        from model_util import build_flow
        from usflows_amd.parallel import mean_log_prob, shard_rows
        spec, sd, _a = load_case("synth_d16_k4_hh2_conj_laplace")
        flow = build_flow(spec, sd)
        x = torch.rand(n_rows, 16, generator=torch.Generator().manual_seed(9))      # the same global batch on every rank
        lo, hi = shard_rows(n_rows, rank, world)
        mean, lp = mean_log_prob(flow, x[lo:hi])
        q.put((rank, float(mean), lp.double().sum().item(), hi - lo))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_rows", [1003, 5])
def test_mean_log_prob_allreduce_world8_uneven_total(n_rows):
    """8 ranks (the node's world size), a total that does not divide (1003 = 3 x 126 + 5 x 125) and one smaller than the world
    (5 rows: three ranks hold an empty shard and still join the collective): every rank ends with the same global mean"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from golden_util import load_case
    from model_util import build_flow
    from usflows_amd.parallel import shard_rows
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q, n_rows)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(8))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[3] for r in res] == [b - a for a, b in (shard_rows(n_rows, r, 8) for r in range(8))] and sum(r[3] for r in res) == n_rows
    spec, sd, _a = load_case("synth_d16_k4_hh2_conj_laplace")
    with torch.no_grad():
        expect = build_flow(spec, sd).log_prob(torch.rand(n_rows, 16, generator=torch.Generator().manual_seed(9))).double().mean().item()
    means = [r[1] for r in res]
    assert max(means) - min(means) <= 1e-9 * abs(expect)                   # every rank's line agrees
    assert abs(means[0] - expect) < 1e-5 * abs(expect)
    # the shares the 8-GPU configurations get
    assert [b - a for a, b in (shard_rows(262144, r, 8) for r in range(8))] == [32768] * 8
    assert [b - a for a, b in (shard_rows(10 ** 6, r, 8) for r in range(8))] == [125000] * 8


def test_shard_rows_partition():
    from usflows_amd.parallel import shard_rows
    for n in (0, 1, 7, 8, 65536, 1000003):
        for w in (1, 2, 3, 8):
            parts = [shard_rows(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def _train_worker(rank, world, port, q, n_rows=48):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(2)
        import emulator
        from golden_util import load_case
        from model_util import build_flow
        from usflows_amd import training
        from usflows_amd.parallel import data_parallel_training, shard_rows
        emulator.install_training_emulation(pytest.MonkeyPatch())       # HIP entry points -> documented semantics on CPU
        spec, sd, a = load_case("synth_d16_k3_densenn_relu")
        flow = build_flow(spec, sd)
        data_parallel_training(flow)
        lo, hi = shard_rows(n_rows, rank, world)                        # 48 rows: 24 + 24; 47: 24 + 23; 1: 1 + 0
        if hi > lo:
            loss = -training.log_prob_with_grad(flow._train_obj, a["x"][lo:hi], None).mean()
        else:                                                           # what Flow.log_prob does for an empty shard
            loss = -training.log_prob_empty_shard(flow._train_obj, a["x"][lo:hi]).sum()
        loss.backward()
        # (numpy: pickled by value -- torch tensors would travel as shared-memory handles that die with the worker)
        q.put((rank, {n: p.grad.numpy().copy() for n, p in flow.named_parameters() if p.grad is not None}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_rows", [48, 47, 1])
def test_data_parallel_training_gradients_world2(n_rows):
    """two ranks, half the batch each, ONE all-reduce of the flat gradient arena: every rank ends with the
    gradient of the full-batch mean loss -- also for shards of unequal size (47 = 24 + 23: weighted by row counts) and
    when one rank's shard is EMPTY (1 = 1 + 0: it joins the collective with weight 0 instead of hanging the other)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emulator
    from golden_util import load_case
    from model_util import build_flow
    from usflows_amd import training
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q, n_rows)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    mpatch = pytest.MonkeyPatch()
    try:
        emulator.install_training_emulation(mpatch)
        spec, sd, a = load_case("synth_d16_k3_densenn_relu")
        flow = build_flow(spec, sd)
        path = training.TrainPath(flow)
        (-training.log_prob_with_grad(path, a["x"][:n_rows], None).mean()).backward()
        ref = {n: p.grad for n, p in flow.named_parameters() if p.grad is not None}
    finally:
        mpatch.undo()
    assert set(ref) <= set(res[0]) and set(ref) <= set(res[1]) and len(ref) >= 10
    for n, g in ref.items():
        big = max(g.abs().max().item(), 1e-12)
        for r in (0, 1):
            assert (torch.from_numpy(res[r][n]) - g).abs().max().item() <= 1e-5 * big, (r, n)


# ---- image-shaped flows: gradients live in p.grad, one all-reduce of the flattened gradients per step -------------------
def _image_train_worker(rank, world, port, q, n_rows):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np
        from golden_util import load_image_case
        from usflows_amd.parallel import data_parallel_training, shard_rows
        flow, a = load_image_case("image_c4_6x6_k3_gated_ln_hh1_conj")
        x = a["x"][:n_rows]
        data_parallel_training(flow)
        assert flow.__dict__.get("_grad_allreduce") is not None
        lo, hi = shard_rows(n_rows, rank, world)                        # 12 rows: 6 + 6; 11: 6 + 5
        ds = torch.utils.data.TensorDataset(x[lo:hi], torch.zeros(hi - lo))
        np.random.seed(3)
        flow.fit(ds, optim=torch.optim.SGD, optim_params=dict(lr=1e-3), batch_size=hi - lo, shuffle=False, device=torch.device("cpu"),
                 epochs=1)
        # (numpy: pickled by value -- torch tensors would travel as shared-memory handles that die with the worker)
        q.put((rank, {k: v.detach().numpy().copy() for k, v in flow.state_dict().items()}))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_rows", [12, 11])
def test_image_flow_data_parallel_fit_world2(n_rows):
    """two ranks, each fitting its shard with the gradient all-reduce between backward and the optimiser step, end with the
    parameters of ONE process fitting the whole batch (equal and unequal shards)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from golden_util import load_image_case
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_image_train_worker, args=(r, 2, port, q, n_rows)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    flow, a = load_image_case("image_c4_6x6_k3_gated_ln_hh1_conj")
    ds = torch.utils.data.TensorDataset(a["x"][:n_rows], torch.zeros(n_rows))
    flow.fit(ds, optim=torch.optim.SGD, optim_params=dict(lr=1e-3), batch_size=n_rows, shuffle=False, device=torch.device("cpu"), epochs=1)
    ref = flow.state_dict()
    for rank in (0, 1):
        for k, v in ref.items():
            s = max(v.abs().max().item(), 1e-6)
            assert (torch.from_numpy(got[rank][k]) - v).abs().max().item() <= 2e-6 * s + 1e-9, (rank, k)
