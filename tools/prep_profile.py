#!/usr/bin/env python3
"""Run the parameter prep of the cfg2 model a few times (for `rocprofv3 --kernel-trace --stats -- python3 tools/prep_profile.py`)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usflows_amd.synth import ModelSpec, synth_state_dict, build_usflow  # noqa: E402

dev = torch.device("cuda:0")
spec = ModelSpec(784, 32, [256, 256], householder=0, affine_conjugation=False, negative_slope=0.01,
                 conditioner="ConditionalDenseNN", base="laplace")
flow = build_usflow(spec, synth_state_dict(spec, seed=100, alpha=0.1), device=str(dev))
eng = flow.engine()
x = torch.rand(65536, 784, device=dev)
for it in range(6):
    eng.refresh()
    torch.cuda.synchronize()
    t = time.perf_counter()
    eng.pack(dev)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    with torch.no_grad():
        flow.log_prob(x[:64] if it % 2 else x)
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f"pack host {1e3*(t1-t):.2f} ms, +sync {1e3*(t2-t):.2f} ms; plan+first call host {1e3*(t3-t2):.2f} ms, +sync {1e3*(t4-t2):.2f} ms")
