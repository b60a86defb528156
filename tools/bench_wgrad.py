"""usf_conv_wgrad_f32 / usf_layernorm_channels_bwd_f32 alone at the image models' shapes (python3 tools/bench_wgrad.py)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usflows_amd import _ext  # noqa: E402

dev = "cuda:0"
SHAPES = [(65536, 32, 32, 7, 7, 3), (65536, 16, 32, 7, 7, 3), (65536, 32, 16, 7, 7, 3), (65536, 32, 64, 7, 7, 1), (65536, 16, 16, 7, 7, 1),
          (16384, 32, 32, 8, 8, 3), (16384, 48, 32, 8, 8, 3), (16384, 32, 48, 8, 8, 3), (16384, 48, 48, 8, 8, 1)]


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


out = []
for B, cin, cout, H, W, ks in SHAPES:
    x = torch.randn(B, cin, H, W, device=dev)
    dy = torch.randn(B, cout, H, W, device=dev)
    pre = torch.randn(cin, device=dev) if ks == 1 and cin == cout else None
    ms = timeit(lambda: _ext.conv_wgrad(x, dy, ks, pre_sub=pre))
    fl = 2.0 * B * H * W * cin * cout * ks * ks
    by = 4.0 * B * H * W * (cin + cout)
    out.append(f"{cin}->{cout} k{ks} {H}x{W}: {ms:.3f} ms ({fl / ms / 1e9:.1f} TF/s, {by / ms / 1e6:.0f} GB/s)")
for B, C, P in [(65536, 32, 49), (16384, 32, 64)]:
    x = torch.randn(B, C, P, device=dev)
    dy = torch.randn(B, C, P, device=dev)
    g = torch.ones(C, device=dev)
    ms = timeit(lambda: _ext.layernorm_channels_bwd(x, dy, g, 1e-5, _ext.ACT_LEAKY_RELU, 0.0))
    out.append(f"ln_bwd C{C} P{P}: {ms:.3f} ms ({12.0 * B * C * P / ms / 1e6:.0f} GB/s)")
print(f"DBG={os.environ.get('USF_WGRAD_DBG', '0')} " + " | ".join(out))
