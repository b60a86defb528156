#!/usr/bin/env python3
"""log_prob of an image-shaped flow (the reference's MNIST configuration: in_dims [16, 7, 7], ConvNet2D(c_hidden 32, one
layer, gated, layer-normalised), 2 coupling blocks, householder 1, affine conjugation) on the GPU: ms per call and the
kernels it spends them in (tuning aid for SURVEY row N4)."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usflows_amd.flows import USFlow
from usflows_amd.networks import ConvNet2D

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
CFG = sys.argv[2] if len(sys.argv) > 2 else "mnist"      # "cifar": in_dims [48, 8, 8], 3 conditioner layers, 10 blocks
dev = "cuda:0"
dims = [16, 7, 7] if CFG == "mnist" else [48, 8, 8]
torch.manual_seed(0)
base = torch.distributions.Laplace(torch.zeros(dims).to(dev), torch.ones(dims).to(dev))
flow = USFlow(base, dims, 2 if CFG == "mnist" else 10, ConvNet2D, dict(c_in=dims[0], c_hidden=32, num_layers=1 if CFG == "mnist" else 3, padding="same", kernel_size=3,
                                            normalize_layers=True, gating=True, nonlinearity=torch.nn.ReLU()),
              householder=1 if CFG == "mnist" else 0, affine_conjugation=True).to(dev)
x = torch.rand(B, *dims, device=dev)
import warnings
warnings.simplefilter("ignore")
with torch.no_grad():
    for _ in range(3):
        lp = flow.log_prob(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        lp = flow.log_prob(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
print(f"image flow log_prob B={B}: {ms:.2f} ms per call, {B / ms * 1e3:.0f} samples/s")
if os.environ.get("IMAGE_PROFILE_PLAIN") == "1":      # under rocprofv3: no second profiler in the process
    sys.exit(0)
from torch.profiler import profile, ProfilerActivity
with torch.no_grad(), profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(3):
        flow.log_prob(x)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=70))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=12, max_name_column_width=60))
