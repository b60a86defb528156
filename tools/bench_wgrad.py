#!/usr/bin/env python3
"""usf_wgrad_f32 at the training shapes of the cfg2 model (tuning aid): ms per launch and fp32-equivalent TFLOP/s.
USF_WGRAD_OLD=1 selects the round-1 bf16x3 kernel, USF_WGRAD_BLOCKS=n the block target of the loader-wave kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usflows_amd import _ext  # noqa: E402

_ext.load()
dev = "cuda:0"
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for N, K in [(784, 784), (256, 392), (256, 256), (392, 256)]:
    ld_y, ld_a = (N + 3) // 4 * 4, (K + 3) // 4 * 4
    Y = torch.randn(M, ld_y, device=dev)
    A = torch.randn(M, ld_a, device=dev)
    G = torch.zeros(N, K, device=dev)
    ref = (Y[:4096, :N].double().t() @ A[:4096, :K].double())
    _ext.wgrad(Y[:4096], A[:4096], G, M=4096, N=N, K=K, ldy=ld_y, lda=ld_a, ldg=K, mode=1)
    err = (G.double() - ref).abs().max().item() / ref.abs().max().item()
    for _ in range(3):
        _ext.wgrad(Y, A, G, M=M, N=N, K=K, ldy=ld_y, lda=ld_a, ldg=K, mode=1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 20
    e0.record()
    for _ in range(it):
        _ext.wgrad(Y, A, G, M=M, N=N, K=K, ldy=ld_y, lda=ld_a, ldg=K, mode=1)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    print(f"wgrad M={M} N={N} K={K}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s fp32-equivalent  (4096-row check: rel err {err:.1e})", flush=True)
