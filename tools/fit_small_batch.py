#!/usr/bin/env python3
"""Flow.fit of the cfg2 flat flow at the reference's default batch size (32 rows; tests/explib/mnist.yaml:34): ms per
optimiser step with the training step replayed as a hipGraph (default) and with eager steps (USFLOWS_AMD_TRAIN_GRAPH=0),
and the two runs' losses side by side (same data, same order: they must agree)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usflows_amd.synth import ModelSpec, synth_state_dict, build_usflow  # noqa: E402
from usflows_amd.sophia import SophiaG  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = "cuda:0"


class DS:
    def __init__(self, x):
        self.x = x

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        return (self.x[i],)


def run(graph: bool):
    os.environ["USFLOWS_AMD_TRAIN_GRAPH"] = "1" if graph else "0"
    spec = ModelSpec(784, 32, [256, 256], householder=0, affine_conjugation=False, negative_slope=0.01,
                     conditioner="ConditionalDenseNN", base="laplace")
    flow = build_usflow(spec, synth_state_dict(spec, seed=100, alpha=0.1), device=dev)
    x = torch.rand(B * STEPS, 784, generator=torch.Generator().manual_seed(5))
    ds = DS(x)
    # one warm epoch (caches, eager steps, capture), then the timed one
    flow.fit(ds, SophiaG, dict(lr=1e-6), batch_size=B, shuffle=False, device=torch.device(dev), epochs=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = flow.fit(ds, SophiaG, dict(lr=1e-6), batch_size=B, shuffle=False, device=torch.device(dev), epochs=1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = flow.__dict__.get("_train_graph_state") or {}
    return dt / STEPS * 1e3, losses, st.get("replays", 0), getattr(flow, "_train_graph_failed", False)


ms_g, l_g, rep, failed = run(True)
if os.environ.get("FIT_ONLY") == "graph":  # for kernel traces: the replayed step alone
    print(f"Flow.fit cfg2 flat flow, batch {B}: graph replay {ms_g:.2f} ms/step ({rep} replays, capture failed: {failed})")
    sys.exit(0)
ms_e, l_e, _, _ = run(False)
print(f"Flow.fit cfg2 flat flow, batch {B}, {STEPS} steps per epoch: graph replay {ms_g:.2f} ms/step ({rep} replays, capture failed: {failed}); "
      f"eager {ms_e:.2f} ms/step; epoch loss graph {l_g[-1]:.6f} vs eager {l_e[-1]:.6f}")
