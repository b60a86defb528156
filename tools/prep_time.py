#!/usr/bin/env python3
"""How long the batch-independent parameter prep takes (cold / warm), and what one training step
(log_prob under autograd + backward) costs at the cfg2 model.  Baseline numbers for SURVEY rows N1 / N2.

    python tools/prep_time.py [--dim 784 --blocks 32 --hidden 256 256 --batch 4096]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=784)
    ap.add_argument("--blocks", type=int, default=32)
    ap.add_argument("--hidden", type=int, nargs="+", default=[256, 256])
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--householder", type=int, default=0)
    ap.add_argument("--conj", action="store_true")
    ap.add_argument("--train-steps", type=int, default=3)
    args = ap.parse_args()
    from usflows_amd.synth import ModelSpec, synth_state_dict, build_usflow
    dev = torch.device("cuda:0")
    spec = ModelSpec(args.dim, args.blocks, list(args.hidden), householder=args.householder,
                     affine_conjugation=args.conj, negative_slope=0.01, conditioner="ConditionalDenseNN",
                     base="laplace")
    flow = build_usflow(spec, synth_state_dict(spec, seed=100, alpha=0.1), device=str(dev))
    eng = flow.engine()
    x = torch.rand(args.batch, args.dim, device=dev)
    out = {}

    def timed(f):
        torch.cuda.synchronize()
        t = time.perf_counter()
        f()
        torch.cuda.synchronize()
        return time.perf_counter() - t

    def prep():
        eng.refresh()
        eng.pack(dev)
        with torch.no_grad():
            flow.log_prob(x[:64])          # builds the plan -> forces every cached matrix / split plane

    out["prep_cold_s"] = timed(prep)
    out["prep_warm_s"] = [round(timed(prep), 4) for _ in range(3)]
    with torch.no_grad():
        out["log_prob_nograd_s"] = [round(timed(lambda: flow.log_prob(x)), 4) for _ in range(3)]

    opt = torch.optim.Adam(flow.parameters(), lr=1e-6)

    def train_step():
        opt.zero_grad()
        loss = -flow.log_prob(x).mean()
        loss.backward()
        opt.step()

    out["train_step_s"] = [round(timed(train_step), 4) for _ in range(args.train_steps)]
    out["config"] = vars(args)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
