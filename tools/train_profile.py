#!/usr/bin/env python3
"""Wall-clock split of the device training step (forward / backward / optimiser) at the cfg2 model.
`rocprofv3 --kernel-trace --stats -- python3 tools/train_profile.py` adds the per-kernel GPU time."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usflows_amd.synth import ModelSpec, synth_state_dict, build_usflow  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--dim", type=int, default=784)
ap.add_argument("--blocks", type=int, default=32)
ap.add_argument("--hidden", type=int, nargs="+", default=[256, 256])
ap.add_argument("--householder", type=int, default=0)
ap.add_argument("--conj", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")
spec = ModelSpec(args.dim, args.blocks, list(args.hidden), householder=args.householder, affine_conjugation=args.conj,
                 negative_slope=0.01, conditioner="ConditionalDenseNN", base="laplace")
flow = build_usflow(spec, synth_state_dict(spec, seed=100, alpha=0.1), device=str(dev))
x = torch.rand(args.batch, args.dim, device=dev)
opt = torch.optim.Adam(flow.parameters(), lr=1e-6)


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


for it in range(args.steps + 2):
    opt.zero_grad(set_to_none=True)
    t0 = sync()
    loss = -flow.log_prob(x).mean()
    t1h = time.perf_counter()
    t1 = sync()
    loss.backward()
    t2h = time.perf_counter()
    t2 = sync()
    opt.step()
    t3 = sync()
    print(f"step {it}: forward {1e3*(t1-t0):.1f} ms (host {1e3*(t1h-t0):.1f}), backward {1e3*(t2-t1):.1f} ms "
          f"(host {1e3*(t2h-t1):.1f}), optimiser {1e3*(t3-t2):.1f} ms, total {1e3*(t3-t0):.1f} ms, loss {loss.item():.3f}")
