#!/usr/bin/env python3
"""usf_linear_f32 at small batches as the flow runs it: a chain of LAYERS dependent launches, every layer with weights of its
own (cold in the L2), replayed as a hipGraph -- microseconds per layer.  tools/bench_skinny.py [rows] [layers]
Variants by environment (read once per process): USF_SKINNY_KS_SMALL=4|8 (K ranges per block up to 32 rows),
USFLOWS_AMD_TUNE=skinny_g=4|8 (k-steps fetched together)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usflows_amd import _ext  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32
LAYERS = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
_ext.load()
for N, K in ((784, 784), (256, 392), (256, 256), (392, 256)):
    Ws = [torch.randn(N, K, device=dev) / math.sqrt(K) for _ in range(LAYERS)]
    bias = torch.randn(N, device=dev)
    A = torch.randn(M, K, device=dev)
    outs = [torch.empty(M, N, device=dev) for _ in range(2)]

    def chain():
        for i, W in enumerate(Ws):
            _ext.linear(A, W, outs[i & 1], M=M, N=N, K=K, lda=K, ldw=K, ldc=N, bias=bias, act=1, slope=0.01)

    chain()
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            chain()
        for _ in range(3):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        e1.synchronize()
    print(f"rows {M}  {N} x {K}: {e0.elapsed_time(e1) / 20 / LAYERS * 1e3:.2f} us per layer "
          f"(KS_SMALL={os.environ.get('USFLOWS_AMD_TUNE', '')} G={'(skinny_ks_small / skinny_g)'})", flush=True)
