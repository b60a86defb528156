"""Micro-benchmark of usf_linear_f32 at the shapes of BASELINE cfg2 (dev tool, GPU box only)."""
import sys, os, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from usflows_amd import _ext

dev = torch.device("cuda:0")
_ext.load()

def run(M, N, K, iters=20, **kw):
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) / math.sqrt(K)
    C = torch.empty(M, N, device=dev)
    bias = torch.randn(N, device=dev)
    f = lambda: _ext.linear(A, W, C, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, bias=bias, **kw)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    tf = 2.0 * M * N * K / ms / 1e9
    # reference: torch (rocBLAS/hipBLASLt) sgemm
    g = lambda: torch.addmm(bias, A, W.t())
    for _ in range(3): g()
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): g()
    e1.record(); torch.cuda.synchronize()
    ms2 = e0.elapsed_time(e1) / iters
    print(f"M={M} N={N} K={K}: usf {ms:.3f} ms {tf:.1f} TF/s ({tf/157.3*100:.0f}% of f32 MFMA peak) | torch addmm {ms2:.3f} ms {2.0*M*N*K/ms2/1e9:.1f} TF/s", flush=True)

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    run(B, 784, 784)
    run(B, 256, 392, act=1, slope=0.01)
    run(B, 256, 256, act=1, slope=0.01)
    run(B, 392, 256)
    run(B, 3072, 3072, iters=5)
    run(B, 1024, 1536, iters=5)
