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


@pytest.mark.parametrize("M,N,K", [(8200, 784, 784), (33001, 130, 260), (21000, 392, 256), (65536, 256, 392), (4096, 784, 784), (2500, 392, 256)])
def test_wgrad_bias_adds_the_column_sums_without_changing_the_gradient(M, N, K):
    """usf_wgrad_bias_f32: the weight gradient has the bits of usf_wgrad_f32, the column sums are those of usf_colsum_f32
    within fp32 summation-order noise, both reproducible"""
    ext = _ext()
    g = torch.Generator().manual_seed(M + N)
    ldy, lda = (N + 3) // 4 * 4 + 4, (K + 3) // 4 * 4 + 8
    Y, A = torch.randn(M, ldy, generator=g).to(DEV), torch.randn(M, lda, generator=g).to(DEV)
    assert ext.wgrad_bias_ok(M, N, K, ldy, lda, 1) and not ext.wgrad_bias_ok(M, N, K, ldy, lda, 0)
    assert not ext.wgrad_bias_ok(1000, N, K, ldy, lda, 1)             # below 2048 rows mode 1 runs the exact-f32 kernel
    G0, G1, G2 = (torch.empty(N, K, device=DEV) for _ in range(3))
    ext.wgrad(Y, A, G0, M=M, N=N, K=K, ldy=ldy, lda=lda, ldg=K, alpha=0.5, mode=1)
    cs1, cs2 = torch.full((N,), 2.0, device=DEV), torch.full((N,), 2.0, device=DEV)
    ext.wgrad(Y, A, G1, M=M, N=N, K=K, ldy=ldy, lda=lda, ldg=K, alpha=0.5, mode=1, colsum=cs1, cs_alpha=-1.0, cs_beta=3.0)
    ext.wgrad(Y, A, G2, M=M, N=N, K=K, ldy=ldy, lda=lda, ldg=K, alpha=0.5, mode=1, colsum=cs2, cs_alpha=-1.0, cs_beta=3.0)
    torch.cuda.synchronize()
    assert torch.equal(G0, G1) and torch.equal(G1, G2) and torch.equal(cs1, cs2)
    ref = 6.0 - Y[:, :N].double().sum(0)
    assert (cs1.double() - ref).abs().max().item() <= 2e-6 * math.sqrt(M) * max(1.0, ref.abs().max().item())


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


# ---- weight gradient from pre-split operand planes (usf_wgrad_planes_f32, usf_split_planes_f32, A_planes_out) ----
def _planes_sum(P):
    """fp32 value of [3, rows, cols] bf16 planes (the sum is exact in fp32: three 8-bit pieces of a 24-bit significand)"""
    return (P[0].float() + P[1].float()) + P[2].float()


@pytest.mark.parametrize("M,N,ld", [(1, 8, 8), (37, 20, 24), (1000, 392, 400), (8200, 784, 800)])
def test_split_planes_is_an_exact_three_way_split(M, N, ld):
    ext = _ext()
    g = torch.Generator().manual_seed(M + N)
    X = (torch.randn(M, ld, generator=g) * torch.exp(3 * torch.randn(M, ld, generator=g))).to(DEV)
    P = ext.row_planes(M, N, DEV)
    P.fill_(7.0)                                               # every element of the buffer is written
    ext.split_planes(X, P, M=M, N=N, ldx=ld)
    torch.cuda.synchronize()
    S = _planes_sum(P)
    assert torch.equal(S[:M, :N], X[:, :N])
    assert not S[M:].any() and not S[:, N:].any()              # rows up to ceil32(M) and padding columns: zeros
    hi = X[:, :N].bfloat16()
    assert torch.equal(P[0, :M, :N], hi)                       # round-to-nearest residual split
    assert torch.equal(P[1, :M, :N], (X[:, :N] - hi.float()).bfloat16())


@pytest.mark.parametrize("M,N,K,y_off,a_off", [(8192, 784, 784, 0, 0), (8200, 784, 784, 0, 0), (33001, 130, 260, 8, 16),
                                               (21000, 392, 256, 0, 8), (16384, 256, 392, 0, 0), (9000, 144, 128, 0, 0),
                                               (40, 20, 36, 0, 0)])
def test_wgrad_planes_matches_fp64_and_is_reproducible(M, N, K, y_off, a_off):
    """every tile class of the balanced schedule (full, edge in k, edge in n, corner), ragged row counts, column offsets into
    wider planes; the small shape runs where usf_wgrad_planes_ok says the kernel does not pay (it is still correct)"""
    ext = _ext()
    g = torch.Generator().manual_seed(M + N + K)
    Y = torch.randn(M, y_off + N + 8, generator=g)
    A = torch.randn(M, a_off + K + 16, generator=g)
    G0 = torch.randn(N, K + 4, generator=g)
    ref = 0.5 * (Y[:, y_off:y_off + N].double().t() @ A[:, a_off:a_off + K].double()) - 1.5 * G0[:, :K].double()
    Yp, Ap = ext.row_planes(M, Y.shape[1], DEV), ext.row_planes(M, A.shape[1], DEV)
    ext.split_planes(Y.to(DEV), Yp, M=M, N=Y.shape[1], ldx=Y.shape[1])
    ext.split_planes(A.to(DEV), Ap, M=M, N=A.shape[1], ldx=A.shape[1])
    outs, sums = [], []
    cs_ok = bool(ext.load().usf_wgrad_planes_colsum_ok(M, N, K))
    assert cs_ok == (K >= 64)
    for _ in range(2):
        Gd = G0.to(DEV)
        cs = torch.full((N,), 2.0, device=DEV) if cs_ok else None
        ext.wgrad_planes(Yp, Ap, Gd, M=M, N=N, K=K, ldg=K + 4, y_off=y_off, a_off=a_off, alpha=0.5, beta=-1.5, colsum=cs,
                         cs_alpha=-1.0, cs_beta=3.0)
        torch.cuda.synchronize()
        outs.append(Gd.cpu())
        sums.append(cs.cpu() if cs_ok else None)
    assert torch.equal(outs[0], outs[1])
    if cs_ok:                                                   # the bias gradient from the same pass: -colsum(Y) + 3 * 2
        assert torch.equal(sums[0], sums[1])
        ref_cs = 6.0 - Y[:, y_off:y_off + N].double().sum(0)
        assert (sums[0].double() - ref_cs).abs().max().item() <= 2e-6 * math.sqrt(M) * max(1.0, ref_cs.abs().max().item())
    tol = 2e-6 * math.sqrt(M) * max(1.0, ref.abs().max().item())
    assert (outs[0][:, :K].double() - ref).abs().max().item() <= tol
    assert torch.equal(outs[0][:, K:], G0[:, K:])
    assert ext.wgrad_planes_ok(M, N, K) == (M >= 8192 and M * (-(-N // 128)) * (-(-K // 128)) >= 160000)


def test_wgrad_planes_equals_the_loader_wave_kernel_on_the_plain_grid(monkeypatch):
    """same products in the same order: with the balanced schedule switched off the gradient has the bits of usf_wgrad_f32"""
    import subprocess, sys, textwrap
    code = textwrap.dedent('''
        import torch
        from usflows_amd import _ext as ext
        ext.load()
        g = torch.Generator().manual_seed(5)
        M, N, K = 32768, 392, 256
        assert ext.load().usf_wgrad_variant(M, N, K, N, K, 1) == 2
        Y, A = torch.randn(M, N, generator=g).cuda(), torch.randn(M, K, generator=g).cuda()
        G1, G2 = torch.empty(N, K, device="cuda"), torch.empty(N, K, device="cuda")
        ext.wgrad(Y, A, G1, M=M, N=N, K=K, ldy=N, lda=K, ldg=K, mode=1)
        Yp, Ap = ext.row_planes(M, N, "cuda"), ext.row_planes(M, K, "cuda")
        ext.split_planes(Y, Yp, M=M, N=N, ldx=N); ext.split_planes(A, Ap, M=M, N=K, ldx=K)
        ext.wgrad_planes(Yp, Ap, G2, M=M, N=N, K=K, ldg=K)
        torch.cuda.synchronize()
        assert torch.equal(G1, G2), (G1 - G2).abs().max().item()
        print("same bits")
    ''')
    import os
    env = dict(os.environ, USFLOWS_AMD_TUNE="wgrad_sched=0")      # (the library's tuning table: the plain grid of row ranges)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "same bits" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("M,N,K", [(8192, 784, 784), (8200, 784, 800), (300, 256, 392), (70, 96, 40), (4096, 160, 256)])
def test_linear_writes_the_planes_of_its_input(M, N, K):
    """usf_linear_desc::A_planes_out: the bf16x3 instantiations write the planes they split (column block 0's blocks), the
    kernels that do not split get them from a pass of their own; the GEMM's result does not change by a bit"""
    ext = _ext()
    g = torch.Generator().manual_seed(M + N + K)
    lda = K + 8
    A = torch.randn(M, lda, generator=g).to(DEV)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    Kp = -(-K // 32) * 32
    Wp = torch.zeros(3, N, Kp, dtype=torch.bfloat16, device=DEV)
    ext.split_planes(W, Wp, M=N, N=K, ldx=K) if N % 32 == 0 else None
    if N % 32:                                                  # weight planes [3, N, ceil32(K)]: rows as they are
        hi = W.bfloat16(); r = W - hi.float(); mid = r.bfloat16(); lo = (r - mid.float()).bfloat16()
        Wp[0, :, :K], Wp[1, :, :K], Wp[2, :, :K] = hi, mid, lo
    C0, C1 = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    ext.linear(A, W, C0, M=M, N=N, K=K, lda=lda, ldw=K, ldc=N, bias=bias, W_split=Wp)
    P = ext.row_planes(M, K, DEV)                               # zeros; rows [M, ceil32(M)) stay as the caller left them
    P[:, :M].fill_(3.0)
    ext.linear(A, W, C1, M=M, N=N, K=K, lda=lda, ldw=K, ldc=N, bias=bias, W_split=Wp, planes_out=P)
    torch.cuda.synchronize()
    assert torch.equal(C0, C1)
    S = _planes_sum(P)
    assert torch.equal(S[:M, :K], A[:, :K])
    assert not S[M:].any()                                      # rows [M, ceil32(M)) zero: the weight gradient sums over them
    Q = ext.row_planes(M, K, DEV)
    ext.split_planes(A, Q, M=M, N=K, ldx=lda)
    assert torch.equal(P[:, :M, :K], Q[:, :M, :K])             # the same split as usf_split_planes_f32


def test_linear_planes_out_rejects_a_prologue():
    ext = _ext()
    A, W, Cc = torch.randn(128, 64, device=DEV), torch.randn(96, 64, device=DEV), torch.empty(128, 96, device=DEV)
    with pytest.raises(RuntimeError, match="pre_div"):
        ext.linear(A, W, Cc, M=128, N=96, K=64, lda=64, ldw=64, ldc=96, pre_sub=torch.zeros(64, device=DEV),
                   planes_out=ext.row_planes(128, 64, DEV))


def test_planes_entry_points_reject_what_they_cannot_serve():
    """loud errors, no silent fall-backs: misaligned planes, a workspace that is too small, column sums where no
    instantiation carries them, the bias-fused call where no bf16x3 kernel is chosen"""
    ext = _ext()
    lib = ext.load()
    M, N, K = 8192, 256, 256
    Yp, Ap = ext.row_planes(M, N, DEV), ext.row_planes(M, K, DEV)
    G = torch.empty(N, K, device=DEV)
    ws = torch.empty(16, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    args = lambda yp=Yp, y_off=0, wsf=ws.numel(), cs=None, k=K: (
        yp.data_ptr(), yp.shape[2], yp.shape[1] * yp.shape[2], y_off, Ap.data_ptr(), Ap.shape[2], Ap.shape[1] * Ap.shape[2], 0,
        M, N, k, G.data_ptr(), K, 1.0, 0.0, cs, 1.0, 0.0, ws.data_ptr(), wsf, st)
    assert lib.usf_wgrad_planes_f32(*args()) == -4 and b"workspace" in lib.usf_last_error()
    assert lib.usf_wgrad_planes_f32(*args(yp=ext.row_planes(M, N + 32, DEV), y_off=4, wsf=1 << 30)) == -3   # column offsets in units of 8
    assert lib.usf_wgrad_planes_f32(*args(cs=G.data_ptr(), k=40)) == -2 and b"colsum_out" in lib.usf_last_error()
    A32, Y32 = torch.empty(4096, K, device=DEV), torch.empty(4096, N, device=DEV)
    with pytest.raises(RuntimeError, match="usf_wgrad_bias_ok"):            # 1024 rows: the exact-f32 kernel carries no column sums
        ext.wgrad(Y32, A32, G, M=1024, N=N, K=K, ldy=N, lda=K, ldg=K, mode=1, colsum=torch.empty(N, device=DEV))
    assert lib.usf_split_planes_f32(Y32.data_ptr(), N, 4096, N, Yp.data_ptr() + 2, Yp.shape[2], Yp.shape[1] * Yp.shape[2], st) == -1


@pytest.mark.parametrize("C,gated,ln,M", [(40, True, True, 37), (130, True, False, 5), (33, False, True, 300), (256, True, True, 2049),
                                          (7, False, True, 1)])
def test_gated_norm_rows_backward_vs_fp64_autograd(C, gated, ln, M):
    """usf_gated_norm_rows_bwd_f32 (the backward twin of the vector ConvNet's row pass: GatedMLP's gate + LayerNormVector,
    reference networks.py:206-245) against fp64 autograd of the same formulas; padding columns written as zeros"""
    from usflows_amd import _ext
    g = torch.Generator().manual_seed(C + M)
    cp = (C + 3) // 4 * 4
    ld, ldv = cp + 8, 2 * cp + 12
    skip = torch.randn(M, ld, generator=g).to(DEV)
    vg = torch.randn(M, ldv, generator=g).to(DEV) if gated else None
    dy = torch.randn(M, ld, generator=g).to(DEV)
    gamma = (1 + 0.3 * torch.randn(C, generator=g)).to(DEV) if ln else None
    d_skip = torch.full((M, ld), 7.0, device=DEV)
    d_vg = torch.full((M, ldv), 7.0, device=DEV) if gated else None
    dy_xh = torch.full((M, ld), 7.0, device=DEV) if ln else None
    _ext.gated_norm_rows_bwd(skip, dy, d_skip, M=M, C_cols=C, c_pad=cp, ld_skip=ld, ld_dy=ld, ld_d_skip=ld, vg=vg, ld_vg=ldv, gate_off=cp,
                             d_vg=d_vg, ld_d_vg=ldv, gamma=gamma, eps=1e-5, dy_xh=dy_xh, ld_dy_xh=ld)
    s6 = skip[:, :C].double().requires_grad_(True)
    r = s6
    if gated:
        v6 = vg.double().requires_grad_(True)
        r = r + v6[:, :C] * torch.sigmoid(v6[:, cp: cp + C])
    if ln:
        g6 = gamma.double().requires_grad_(True)
        mean = r.mean(dim=1, keepdim=True)
        var = ((r - mean) ** 2).mean(dim=1, keepdim=True)
        xh = (r - mean) / torch.sqrt(var + 1e-5)
        r = xh * g6
    r.backward(dy[:, :C].double())
    close = lambda a_, b_, what: _assert_close(a_, b_, what)      # noqa: E731
    close(d_skip[:, :C], s6.grad, "d_skip")
    assert (d_skip[:, C:cp] == 0).all() and (d_skip[:, cp:] == 7.0).all()
    if gated:
        close(d_vg[:, :C], v6.grad[:, :C], "d_val")
        close(d_vg[:, cp: cp + C], v6.grad[:, cp: cp + C], "d_gate")
        assert (d_vg[:, C:cp] == 0).all() and (d_vg[:, cp + C: 2 * cp] == 0).all() and (d_vg[:, 2 * cp:] == 7.0).all()
    if ln:
        close(dy_xh[:, :C].sum(0), g6.grad, "dgamma = colsum(dy * xh)")
        assert (dy_xh[:, C:cp] == 0).all()


def _assert_close(got, want, what, tol=2e-5):
    got, want = got.double().cpu(), want.double().cpu()
    s_ = max(want.abs().max().item(), 1e-30)
    err = (got - want).abs().max().item()
    assert err <= tol * s_, f"{what}: max abs err {err:.3e} vs scale {s_:.3e}"
