#!/usr/bin/env python3
"""usf_conv2d_same_f32 at the conditioner shapes of the MNIST image configuration: register-weight kernel vs first kernel
(USFLOWS_AMD_TUNE=conv_wreg=0) -- run once per setting of USFLOWS_AMD_TUNE=convw_dbg=... (phase ablations; tuning aid)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usflows_amd import _ext
_ext.load()
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
tag = f"WREG={os.environ.get('USFLOWS_AMD_TUNE', '')} DBG={'see USFLOWS_AMD_TUNE (conv_wreg, convw_dbg)'}"
out = []
for cin, cout, H, W in [(32, 32, 7, 7), (16, 32, 7, 7), (32, 16, 7, 7), (32, 32, 8, 8), (48, 32, 8, 8), (32, 48, 8, 8)]:
    b_ = B if H * W < 60 else B // 4
    x = torch.randn(b_, cin, H, W, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
    planes = _ext.conv2d_weight_planes(w)
    for _ in range(3): _ext.conv2d_same(x, planes, cout, 3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): _ext.conv2d_same(x, planes, cout, 3)
    e1.record(); torch.cuda.synchronize()
    out.append(f"{cin}->{cout}@{H}x{W}: {e0.elapsed_time(e1) / 10:.3f} ms")
print(tag, " | ".join(out), flush=True)
