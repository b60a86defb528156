"""Backward-pass kernels (usf_train.hip) through the C ABI against plain torch fp64 on the CPU:
wgrad (weight gradient of F.linear = reduction over the batch), colsum (bias gradient), act_grad (LeakyReLU
backward as ATen's leaky_relu_backward) and the gradient of the base log-density.  fp32 tolerance: 2e-6 x sqrt(M)
relative to the magnitude of the sum (exact-f32 MFMA, fp32 accumulation)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ext():
    from usflows_amd import _ext
    _ext.load()
    return _ext


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("M,N,K", [(1, 4, 4), (17, 20, 36), (256, 128, 128), (1000, 392, 256), (4096, 784, 784),
                                   (5000, 130, 260), (2048, 392, 256), (0, 8, 8), (8200, 784, 784), (33001, 130, 260),
                                   (21000, 392, 256)])
def test_wgrad_matches_fp64(M, N, K, mode):
    """mode 1 = bf16x3 split on the bf16 MFMA (taken from M >= 2048; the loader-wave kernel from 8192 rows with
    enough tiles: the last three shapes -- ragged row counts, edge tiles in both directions): same tolerance as the
    exact-f32 kernel"""
    ext = _ext()
    if M > 8000:
        lib = ext.load()
        assert lib.usf_wgrad_variant(M, N, K, (N + 3) // 4 * 4 + 4, (K + 3) // 4 * 4 + 8, mode) == (2 if mode == 1 else 0)
    g = torch.Generator().manual_seed(M + N + K)
    ldy, lda, ldg = (N + 3) // 4 * 4 + 4, (K + 3) // 4 * 4 + 8, K + 4     # rows of Y / A 16-byte aligned (contract)
    Y = torch.randn(M, ldy, generator=g)
    A = torch.randn(M, lda, generator=g)
    G0 = torch.randn(N, ldg, generator=g)
    ref = 0.5 * (Y[:, :N].double().t() @ A[:, :K].double()) - 1.5 * G0[:, :K].double()
    Gd = G0.to(DEV)
    ext.wgrad(Y.to(DEV), A.to(DEV), Gd, M=M, N=N, K=K, ldy=ldy, lda=lda, ldg=ldg, alpha=0.5, beta=-1.5, mode=mode)
    torch.cuda.synchronize()
    got = Gd.cpu()
    tol = 2e-6 * math.sqrt(max(M, 1)) * max(1.0, ref.abs().max().item())
    assert (got[:, :K].double() - ref).abs().max().item() <= tol
    assert torch.equal(got[:, K:], G0[:, K:])                 # padding columns of G untouched


def test_wgrad_is_the_autograd_weight_gradient_and_reproducible():
    ext = _ext()
    g = torch.Generator().manual_seed(1)
    M, N, K = 3000, 96, 200
    x = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g, requires_grad=True)
    gy = torch.randn(M, N, generator=g)
    F.linear(x.double(), W.double()).backward(gy.double())
    outs = []
    for _ in range(2):
        G = torch.empty(N, K, device=DEV)
        ext.wgrad(gy.to(DEV), x.to(DEV), G, M=M, N=N, K=K, ldy=N, lda=K, ldg=K)
        outs.append(G.cpu())
    assert torch.equal(outs[0], outs[1])                      # fixed summation order: bitwise reproducible
    assert (outs[0].double() - W.grad).abs().max().item() <= 2e-6 * math.sqrt(M) * W.grad.abs().max().item()


@pytest.mark.parametrize("M,N", [(1, 3), (777, 130), (65536, 64), (0, 5)])
def test_colsum(M, N):
    ext = _ext()
    g = torch.Generator().manual_seed(M + N)
    Y = torch.randn(M, N + 4, generator=g)
    out0 = torch.randn(N, generator=g)
    od = out0.to(DEV)
    ext.colsum(Y.to(DEV), od, M=M, N=N, ldy=N + 4, alpha=-1.0, beta=1.0)
    ref = out0.double() - Y[:, :N].double().sum(0)
    assert (od.cpu().double() - ref).abs().max().item() <= 2e-6 * math.sqrt(max(M, 1)) * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("slope", [0.0, 0.01, 0.3])
def test_act_grad_is_leaky_relu_backward(slope):
    ext = _ext()
    g = torch.Generator().manual_seed(2)
    M, H = 333, 260
    pre = torch.randn(M, H, generator=g)
    pre[0, :8] = 0.0                                          # the kink: ATen takes the negative branch at 0
    pre.requires_grad_(True)
    h = F.leaky_relu(pre, slope)
    d = torch.randn(M, H, generator=g)
    h.backward(d)
    dd = torch.zeros(M, H + 4)
    dd[:, :H] = d
    dd = dd.to(DEV)
    hh = torch.zeros(M, H + 8)
    hh[:, :H] = h.detach()
    ext.act_grad(dd, hh.to(DEV), M=M, H=H, ldd=H + 4, ldh=H + 8, act=ext.ACT_LEAKY_RELU, slope=slope)
    assert torch.equal(dd.cpu()[:, :H], pre.grad)


@pytest.mark.parametrize("base", ["laplace", "normal"])
def test_base_logprob_grad(base):
    ext = _ext()
    g = torch.Generator().manual_seed(3)
    M, D, ldz, ldg = 129, 37, 40, 44
    z = torch.randn(M, ldz, generator=g)
    loc = 0.1 * torch.randn(D, generator=g)
    scale = 0.5 + torch.rand(D, generator=g)
    g_lp = torch.randn(M, generator=g)
    zz = z[:, :D].clone().double().requires_grad_(True)
    dist = (torch.distributions.Laplace if base == "laplace" else torch.distributions.Normal)(loc.double(), scale.double())
    dist.log_prob(zz).sum(-1).backward(g_lp.double())
    out = torch.full((M, ldg), 9.0, device=DEV)
    ext.base_logprob_grad(z.to(DEV), ldz, g_lp.to(DEV), M, D, ext.BASE_LAPLACE if base == "laplace" else ext.BASE_NORMAL,
                          loc.to(DEV), scale.to(DEV), out, ldg)
    got = out.cpu()
    assert (got[:, :D].double() - zz.grad).abs().max().item() <= 1e-6 * zz.grad.abs().max().item()
    assert got[:, D:].abs().max().item() == 0.0


@pytest.mark.parametrize("M", [1, 32, 48, 256])
def test_grad_jobs_one_launch_equals_the_single_calls(M):
    """usf_grad_jobs_f32: the weight and bias gradients of many layers as ONE launch (queued inside
    ``batch_jobs(defer_grads=True)``) -- the weight gradients bit for bit what usf_wgrad_f32 writes for the same operands
    (edge tiles, row / column offsets, alpha / beta), the column sums within fp32 rounding of fp64"""
    ext = _ext()
    g = torch.Generator().manual_seed(M)
    shapes = [(784, 784), (392, 256), (256, 256), (130, 36), (16, 4), (256, 392)]
    jobs = []
    for i, (N, K) in enumerate(shapes):
        ldy, lda = ((N + 3) // 4) * 4 + 8, ((K + 3) // 4) * 4 + 4
        Y = torch.randn(M, ldy, generator=g).to(DEV)
        A = torch.randn(M, lda, generator=g).to(DEV)
        G0 = torch.randn(N, K + 3, generator=g).to(DEV)
        gs0 = torch.randn(N, generator=g).to(DEV)
        jobs.append(dict(Y=Y, A=A, G0=G0, gs0=gs0, N=N, K=K, ldy=ldy, lda=lda, alpha=(-1.0 if i & 1 else 1.0),
                         beta=(1.0 if i == 2 else 0.0), y_off=(4 if i == 3 else 0), a_off=(4 if i == 1 else 0)))
    single, queued = [], []
    for j in jobs:
        Gs, ss = j["G0"].clone(), j["gs0"].clone()
        n_eff, k_eff = j["N"] - j["y_off"], j["K"] - j["a_off"]
        ext.wgrad(j["Y"], j["A"], Gs, M=M, N=n_eff, K=k_eff, ldy=j["ldy"], lda=j["lda"], ldg=Gs.shape[1], y_off=j["y_off"],
                  a_off=j["a_off"], alpha=j["alpha"], beta=j["beta"])
        ext.colsum(j["Y"], ss, M=M, N=n_eff, ldy=j["ldy"], y_off=j["y_off"], alpha=j["alpha"], beta=j["beta"])
        single.append((Gs, ss))
    with ext.batch_jobs(torch.device(DEV), defer_grads=True) as bj:
        for j in jobs:
            Gq, sq = j["G0"].clone(), j["gs0"].clone()
            n_eff, k_eff = j["N"] - j["y_off"], j["K"] - j["a_off"]
            ext.wgrad(j["Y"], j["A"], Gq, M=M, N=n_eff, K=k_eff, ldy=j["ldy"], lda=j["lda"], ldg=Gq.shape[1],
                      y_off=j["y_off"], a_off=j["a_off"], alpha=j["alpha"], beta=j["beta"], mode=1)
            ext.colsum(j["Y"], sq, M=M, N=n_eff, ldy=j["ldy"], y_off=j["y_off"], alpha=j["alpha"], beta=j["beta"])
            queued.append((Gq, sq))
        assert len(bj.grad_jobs) == 2 * len(jobs)
        assert all(torch.equal(q[0], j["G0"]) for q, j in zip(queued, jobs))      # nothing ran yet
    torch.cuda.synchronize()
    for (Gs, ss), (Gq, sq), j in zip(single, queued, jobs):
        assert torch.equal(Gs, Gq), (j["N"], j["K"])
        n_eff = j["N"] - j["y_off"]
        ref = j["alpha"] * j["Y"][:, j["y_off"]: j["y_off"] + n_eff].double().sum(0) + j["beta"] * j["gs0"][:n_eff].double()
        scale = j["Y"].abs().max().item() * max(M, 1)
        assert (sq[:n_eff].double() - ref).abs().max().item() <= 2e-6 * math.sqrt(max(M, 1)) * scale
        assert torch.equal(sq[n_eff:], j["gs0"][n_eff:])                           # nothing written past N


def test_grad_jobs_are_not_queued_above_the_row_limit():
    ext = _ext()
    M = ext.GRAD_JOB_MAX_ROWS + 1
    Y, A = torch.randn(M, 16, device=DEV), torch.randn(M, 8, device=DEV)
    G = torch.zeros(16, 8, device=DEV)
    with ext.batch_jobs(torch.device(DEV), defer_grads=True) as bj:
        ext.wgrad(Y, A, G, M=M, N=16, K=8, ldy=16, lda=8, ldg=8)
        assert not bj.grad_jobs
    assert torch.allclose(G, Y.t() @ A, rtol=1e-4, atol=1e-3)
