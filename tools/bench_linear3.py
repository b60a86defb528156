"""usf_linear_f32 on the bf16x3 kernel (W_split given) at the GEMM shapes of the cfg2 training step (dev tool, GPU box only):
   python tools/bench_linear3.py [rows]        USFLOWS_AMD_TUNE=bf16x3_wm=4 / 8 forces the 4- / 8-wave tile"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from usflows_amd import _ext

dev = torch.device("cuda:0")
lib = _ext.load()


def planes(W):
    N, K = W.shape
    P = torch.zeros(3, N, -(-K // 32) * 32, dtype=torch.bfloat16, device=dev)
    hi = W.bfloat16(); r = W - hi.float(); mid = r.bfloat16(); lo = (r - mid.float()).bfloat16()
    P[0, :, :K], P[1, :, :K], P[2, :, :K] = hi, mid, lo
    return P


def run(M, N, K, residual=False, iters=30):
    A = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) / math.sqrt(K)
    C = torch.randn(M, N, device=dev)
    Wp = planes(W)
    kw = dict(residual=C, ldr=N, res_sign=-1.0) if residual else {}
    f = lambda: _ext.linear(A, W, C, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, W_split=Wp, **kw)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    d = _ext.LinearDesc(); d.M, d.N, d.K = M, N, K
    print(f"M={M} N={N} K={K} residual={residual}: {ms*1e3:7.1f} us  {2.0*M*N*K/ms/1e9:6.1f} TF/s", flush=True)


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    run(B, 784, 784)
    run(B, 256, 392)
    run(B, 256, 256)
    run(B, 392, 256)
    run(B, 392, 256, residual=True)
