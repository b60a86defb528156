#!/usr/bin/env python3
"""usf_conv2d_same_f32 against torch's Conv2d (MIOpen) at the conditioner shapes of the reference's MNIST / CIFAR
configurations (tuning aid)."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usflows_amd import _ext
_ext.load()
dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for cin, cout, H, W, ks in [(16, 32, 7, 7, 3), (32, 32, 7, 7, 3), (32, 64, 7, 7, 1), (32, 16, 7, 7, 3), (48, 32, 8, 8, 3), (4, 32, 14, 14, 3)]:
    b_ = B if H * W < 100 else B // 4
    x = torch.randn(b_, cin, H, W, device=dev)
    w = torch.randn(cout, cin, ks, ks, device=dev) / (cin * ks * ks) ** 0.5
    bias = torch.randn(cout, device=dev)
    planes = _ext.conv2d_weight_planes(w)
    def t(fn, n=10):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    t_hip = t(lambda: _ext.conv2d_same(x, planes, cout, ks, bias=bias))
    t_ref = t(lambda: F.conv2d(x, w, bias, padding=ks // 2))
    flops = 2.0 * b_ * H * W * cout * cin * ks * ks
    print(f"conv {cin:2d}->{cout:2d} {ks}x{ks} on {H}x{W}, B={b_}: HIP {t_hip:.3f} ms ({flops / t_hip / 1e9:.1f} TF/s), torch/MIOpen {t_ref:.3f} ms; "
          f"S = {_ext.load().usf_conv2d_same_fits(cin, cout, H, W, ks)}", flush=True)
