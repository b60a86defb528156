#!/usr/bin/env python3
"""Where a replayed training step spends its host time: taped C calls vs host_op closures (per tape)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usflows_amd import _ext  # noqa: E402
from usflows_amd.synth import ModelSpec, synth_state_dict, build_usflow  # noqa: E402

stats = []


def replay(tape):
    t_c = t_h = 0.0
    n_c = n_h = 0
    with _ext.record(None):
        for e in tape.entries:
            t0 = time.perf_counter()
            if e.__class__ is tuple:
                rc = e[0](*e[1])
                if rc != 0:
                    _ext.check(rc, e[2])
                t_c += time.perf_counter() - t0
                n_c += 1
            else:
                e()
                t_h += time.perf_counter() - t0
                n_h += 1
    stats.append((n_c, t_c * 1e3, n_h, t_h * 1e3))


_ext.replay = replay
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
spec = ModelSpec(784, 32, [256, 256], householder=0, affine_conjugation=False, negative_slope=0.01,
                 conditioner="ConditionalDenseNN", base="laplace")
flow = build_usflow(spec, synth_state_dict(spec, seed=100, alpha=0.1), device=str(dev))
x = torch.rand(B, 784, device=dev)
opt = torch.optim.Adam(flow.parameters(), lr=1e-6)
for it in range(5):
    stats.clear()
    opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = -flow.log_prob(x).mean()
    t1 = time.perf_counter()
    loss.backward()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    opt.step()
    print(f"it {it}: fwd host {1e3*(t1-t0):.1f} bwd host {1e3*(t2-t1):.1f} (+sync {1e3*(t3-t2):.1f}) ms; tapes (n_c, ms_c, n_host, ms_host): "
          + "; ".join(f"({a}, {b:.2f}, {c}, {d:.2f})" for a, b, c, d in stats))
