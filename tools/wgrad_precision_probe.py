import sys, torch
sys.path.insert(0, '/root/repo')
from usflows_amd import _ext
_ext.load()
DEV = "cuda:0"
def pack(X, nkb):
    M, Cn = X.shape
    idx = torch.full((32 * nkb,), -1, dtype=torch.int32); idx[:Cn] = torch.arange(Cn, dtype=torch.int32)
    buf = torch.zeros(_ext.planes_bytes(M, nkb), dtype=torch.uint8, device=DEV)
    _ext.pack_planes(X.to(DEV).contiguous(), buf, M=M, nkb=nkb, idx=idx.to(DEV))
    return buf
for M in (8192,):
  for kind in ("pos", "laplace"):
    g = torch.Generator().manual_seed(1)
    if kind == "pos":
        Y = torch.rand(M, 784, generator=g) + 0.5; A = torch.rand(M, 784, generator=g) + 0.5
    else:
        Y = torch.sign(torch.randn(M, 784, generator=g)) * (0.5 + torch.rand(M, 1, generator=g)) / M
        A = torch.randn(M, 784, generator=g) * 2 + 1
    ref = Y.double().t() @ A.double()
    G1 = torch.zeros(784, 784, device=DEV)
    _ext.wgrad_blocked(pack(Y, 25), 25, 0, pack(A, 25), 25, 0, G1, M=M, N=784, K=784, ldg=784)
    Yd, Ad = Y.to(DEV), A.to(DEV)
    G2 = torch.zeros(784, 784, device=DEV)
    _ext.wgrad(Yd, Ad, G2, M=M, N=784, K=784, ldy=784, lda=784, ldg=784, mode=1, defer=False)
    Yp, Ap = _ext.row_planes(M, 784, DEV), _ext.row_planes(M, 784, DEV)
    _ext.split_planes(Yd, Yp, M=M, N=784, ldx=784); _ext.split_planes(Ad, Ap, M=M, N=784, ldx=784)
    G3 = torch.zeros(784, 784, device=DEV)
    _ext.wgrad_planes(Yp, Ap, G3, M=M, N=784, K=784, ldg=784)
    torch.cuda.synchronize()
    big = ref.abs().max().item()
    for nm, G in (("blocked", G1), ("fp32 rows", G2), ("row planes", G3)):
        d = (G.cpu().double() - ref)
        print(kind, nm, "max/big %.2e fro %.2e" % (d.abs().max().item() / big, d.norm().item() / ref.norm().item()),
              "cols>=768: %.2e  cols<768: %.2e" % (d[:, 768:].abs().max().item() / big, d[:, :768].abs().max().item() / big),
              "rows>=768: %.2e" % (d[768:].abs().max().item() / big))
