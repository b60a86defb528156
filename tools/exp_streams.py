"""Experiment: one 65536-row log_prob vs two 32768-row halves on two HIP streams (two engines, own workspaces)."""
import os, sys, time, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from usflows_amd.synth import ModelSpec, synth_state_dict, build_usflow

dev = torch.device("cuda:0")
spec = ModelSpec(784, 32, [256, 256], householder=0, affine_conjugation=False, negative_slope=0.01,
                 conditioner="ConditionalDenseNN", base="laplace")
sd = synth_state_dict(spec, seed=100, alpha=0.1)
flows = [build_usflow(spec, sd, device="cuda:0") for _ in range(4)]
B = 65536
x = torch.rand(B, 784, generator=torch.Generator().manual_seed(1234)).to(dev)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    ref = flows[0].log_prob(x)
    print("1 stream  :", round(timeit(lambda: flows[0].log_prob(x)), 3), "ms")
    for ns in (2, 4):
        streams = [torch.cuda.Stream() for _ in range(ns)]
        h = B // ns
        xs = [x[i * h:(i + 1) * h].contiguous() for i in range(ns)]

        def multi():
            outs = []
            cur = torch.cuda.current_stream()
            for i in range(ns):
                streams[i].wait_stream(cur)
                with torch.cuda.stream(streams[i]):
                    outs.append(flows[i].log_prob(xs[i]))
            for s in streams:
                cur.wait_stream(s)
            return outs

        outs = multi()
        torch.cuda.synchronize()
        err = (torch.cat(outs) - ref).abs().max().item()
        print(f"{ns} streams :", round(timeit(multi), 3), "ms   max diff vs single", err)
