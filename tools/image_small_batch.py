#!/usr/bin/env python3
"""log_prob of small batches of the reference's MNIST / CIFAR image models: the eager layer loop, the loop replayed as a
hipGraph, the loop as ONE recorded usf_run_ops list -- wall time per call (python3 tools/image_small_batch.py [cfg] [rows ...])"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from usflows_amd.flows import USFlow  # noqa: E402
from usflows_amd.networks import ConvNet2D  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "mnist_image"
rows = [int(v) for v in sys.argv[2:]] or [32, 100, 256, 1024, 4096]
cfg = bench.IMAGE_CONFIGS[name]
dims = list(cfg["in_dims"])
dev = torch.device("cuda:0")
torch.manual_seed(100)
host = USFlow(torch.distributions.Laplace(torch.zeros(dims), torch.ones(dims)), dims, cfg["blocks"], ConvNet2D, dict(cfg["cond"]),
              householder=cfg["householder"], affine_conjugation=True)
bench._condition_image_flow(host, seed=100)
flow = USFlow(torch.distributions.Laplace(torch.zeros(dims, device=dev), torch.ones(dims, device=dev)), dims, cfg["blocks"],
              ConvNet2D, dict(cfg["cond"]), householder=cfg["householder"], affine_conjugation=True)
flow.load_state_dict(host.state_dict(), strict=True)
flow = flow.to(dev)


def timed(x, n=200):
    with torch.no_grad():
        for _ in range(4):
            flow.log_prob(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            flow.log_prob(x)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for B in rows:
    x = torch.rand(B, *dims, device=dev)
    flow.graph_max_rows, flow.list_max_rows = 0, 0
    t_e = timed(x)
    flow.graph_max_rows, flow.list_max_rows = 1 << 30, 0
    t_g = timed(x)
    flow.graph_max_rows, flow.list_max_rows = 1 << 30, 1 << 30
    t_l = timed(x)
    plan = flow.__dict__.get("_loop_lists", {}).get((tuple(x.shape), str(dev)), (None, None))[1]
    print(f"{name} {B} rows: eager loop {t_e:.3f} ms, hipGraph replay {t_g:.3f} ms, op list {t_l:.3f} ms "
          f"({'%d calls, %.1f MB kept' % (plan['n'], plan['bytes'] / 1e6) if plan else 'no list'})", flush=True)
