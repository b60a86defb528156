#!/usr/bin/env python3
"""Flow.fit of the reference's MNIST / CIFAR image models (tests/explib/mnist.yaml:44-77, experiments/cifar/cifar.yaml:56-77):
GPU time of one optimiser step (zero grads, log_prob, backward, SophiaG update) as Flow.fit replays it (the captured
hipGraph, timed over N replays) -- the device training path (usflows_amd/image_training.py) against torch autograd + MIOpen
(usflows_amd.config.image_train = False) on the same flow, same data; the two runs' epoch losses side by side.

    python3 tools/fit_image.py [mnist_image|cifar_image] [batch ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from usflows_amd.flows import USFlow  # noqa: E402
from usflows_amd.networks import ConvNet2D  # noqa: E402
from usflows_amd.sophia import SophiaG  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "mnist_image"
batches = [int(v) for v in sys.argv[2:]] or [32, 256, 4096]
cfg = bench.IMAGE_CONFIGS[name]
dims = list(cfg["in_dims"])
dev = torch.device("cuda:0")


def build():
    torch.manual_seed(100)
    host = USFlow(torch.distributions.Laplace(torch.zeros(dims), torch.ones(dims)), dims, cfg["blocks"], ConvNet2D, dict(cfg["cond"]),
                  householder=cfg["householder"], affine_conjugation=True)
    bench._condition_image_flow(host, seed=100)
    flow = USFlow(torch.distributions.Laplace(torch.zeros(dims, device=dev), torch.ones(dims, device=dev)), dims, cfg["blocks"],
                  ConvNet2D, dict(cfg["cond"]), householder=cfg["householder"], affine_conjugation=True)
    flow.load_state_dict(host.state_dict(), strict=True)
    return flow.to(dev)


def run(B, device_path):
    from usflows_amd.config import config
    config.image_train = bool(device_path)
    flow = build()
    x = torch.rand(B * 6, *dims, generator=torch.Generator().manual_seed(5))
    ds = torch.utils.data.TensorDataset(x, torch.zeros(x.shape[0]))
    losses = flow.fit(ds, SophiaG, dict(lr=1e-6), batch_size=B, shuffle=False, device=dev, epochs=2)
    st = flow.__dict__.get("_train_graph_state") or {}
    g = st.get("graph")
    if g is None:
        return None, losses, 0
    n = 20
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, losses, st.get("replays", 0)


for B in batches:
    ms_d, l_d, rep = run(B, True)
    ms_t, l_t, _ = run(B, False) if os.environ.get("FIT_IMAGE_ONLY") != "device" else (None, l_d, 0)
    fmt = lambda v: "no graph" if v is None else f"{v:.3f} ms"
    print(f"Flow.fit {name} batch {B}: replayed step {fmt(ms_d)} on the device path ({rep} replays in fit) vs {fmt(ms_t)} with torch "
          f"autograd; epoch losses {l_d[-1]:.6f} vs {l_t[-1]:.6f}", flush=True)
