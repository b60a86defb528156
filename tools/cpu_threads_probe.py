"""How many host threads serve the CPU oracle best on this box? (dev tool for bench.py's cpu_baseline)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import usflows_oracle as orc
spec = orc.FlowSpec(784, 32, [256, 256], householder=0)
sd = orc.synth_state_dict(spec, seed=100)
x = torch.rand(4096, 784, generator=torch.Generator().manual_seed(1234))
print("cpu_count", os.cpu_count(), "default threads", torch.get_num_threads(), flush=True)
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    with torch.no_grad():
        orc.flow_log_prob(sd, spec, x[:64])
        t0 = time.perf_counter(); orc.flow_log_prob(sd, spec, x); dt = time.perf_counter() - t0
    print(f"threads {nt:4d}: {dt:.2f} s -> {4096/dt:.0f} samples/s", flush=True)
