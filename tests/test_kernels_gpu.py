"""Per-kernel parity on a real MI355X, through the C ABI (usflows_amd._ext -> libusflows_hip.so).
Reference for each op = the torch-CPU expression the oracle uses (fp32), plus fp64 for error bars."""
import math

import pytest
import torch

import shapes

pytestmark = pytest.mark.gpu


def _ext():
    from usflows_amd import _ext
    _ext.load()
    return _ext


def _dev():
    assert torch.cuda.is_available(), "GPU test needs a GPU"
    return torch.device("cuda:0")


def _linear_ref(A, W, bias, pre_div, pre_sub, residual, post_mul, res_sign, slope, act, dtype):
    a = A.to(dtype)
    if pre_div is not None:
        a = a / pre_div.to(dtype)
    if pre_sub is not None:
        a = a - pre_sub.to(dtype)
    v = a @ W.to(dtype).t()
    if bias is not None:
        v = v + bias.to(dtype)
    if act:
        v = torch.nn.functional.leaky_relu(v, slope)
    if residual is not None:
        v = residual.to(dtype) + res_sign * v
    if post_mul is not None:
        v = v * post_mul.to(dtype)
    return v


CASES = [
    # M, N, K, flags
    (1, 1, 4, {}),
    (5, 3, 8, dict(bias=True)),
    (48, 7, 8, dict(bias=True, pre_sub=True)),
    (65, 33, 36, dict(bias=True, act=True)),
    (64, 64, 64, dict(bias=True, residual=True, res_sign=-1.0)),
    (130, 160, 52, dict(pre_div=True, pre_sub=True)),
    (257, 161, 100, dict(bias=True, post_mul=True)),
    (300, 784, 784, dict(pre_sub=True)),
    (513, 256, 392, dict(bias=True, act=True)),
    (256, 392, 256, dict(bias=True, residual=True, res_sign=1.0)),
    (1000, 129, 260, dict(bias=True, act=True, slope=0.0)),
    (2048, 800, 784, dict(bias=True)),
]


@pytest.mark.parametrize("M,N,K,flags", CASES)
def test_linear_parity(M, N, K, flags):
    ext, dev = _ext(), _dev()
    g = torch.Generator().manual_seed(M * 1000003 + N * 1009 + K)
    lda, ldw, ldc = K + 4, K, N + 3
    A = torch.randn(M, lda, generator=g)
    W = torch.randn(N, ldw, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g) if flags.get("bias") else None
    pre_div = (torch.rand(K, generator=g) + 0.5) * torch.where(torch.rand(K, generator=g) < 0.5, -1.0, 1.0) \
        if flags.get("pre_div") else None
    pre_sub = torch.randn(K, generator=g) if flags.get("pre_sub") else None
    residual = torch.randn(M, N + 5, generator=g) if flags.get("residual") else None
    post_mul = torch.randn(N, generator=g) if flags.get("post_mul") else None
    act = 1 if flags.get("act") else 0
    slope = flags.get("slope", 0.01)
    res_sign = flags.get("res_sign", 1.0)

    d = lambda t: None if t is None else t.to(dev)
    C = torch.full((M, ldc), float("nan"), device=dev)
    ext.linear(d(A), d(W), C, M=M, N=N, K=K, lda=lda, ldw=ldw, ldc=ldc, bias=d(bias), pre_div=d(pre_div),
               pre_sub=d(pre_sub), residual=d(residual), ldr=(N + 5), post_mul=d(post_mul), res_sign=res_sign,
               act=act, slope=slope)
    torch.cuda.synchronize()
    out = C.cpu()
    res = None if residual is None else residual[:, :N]
    ref64 = _linear_ref(A[:, :K], W, bias, pre_div, pre_sub, res, post_mul, res_sign, slope, act, torch.float64)
    ref32 = _linear_ref(A[:, :K], W, bias, pre_div, pre_sub, res, post_mul, res_sign, slope, act, torch.float32)
    assert torch.isnan(out[:, N:]).all(), "kernel wrote outside its N columns"
    got = out[:, :N].double()
    scale = ref64.abs().max().item() + 1e-30
    err = (got - ref64).abs().max().item() / scale
    err_ref = (ref32.double() - ref64).abs().max().item() / scale
    # fp32 fma-chain GEMM: same error class as the CPU sgemm (tolerance: 1e-5 relative, north_star)
    assert err < max(4 * err_ref, 2e-6), (err, err_ref)
    assert err < 1e-5


@pytest.mark.parametrize("M,N,K", shapes.BF16X3_SMALL)
def test_linear_bf16x3_split_precision(M, N, K):
    """bf16x3 path: three-way residual split of both operands on the bf16 MFMA; must carry fp32-class error"""
    ext, dev = _ext(), _dev()
    from usflows_amd.engine import FlowEngine
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g) * 3
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    Wd = W.to(dev)
    planes = FlowEngine._split_planes({"mats": {}}, Wd)
    assert torch.equal(planes.float().sum(0)[:, :K], Wd) or (planes.float().sum(0)[:, :K] - Wd).abs().max() < 1e-9
    C = torch.full((M, N), float("nan"), device=dev)
    ext.linear(A.to(dev), Wd, C, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, bias=bias.to(dev), W_split=planes)
    torch.cuda.synchronize()
    ref64 = A.double() @ W.double().t() + bias.double()
    ref32 = A @ W.t() + bias
    scale = ref64.abs().max().item()
    err = (C.cpu().double() - ref64).abs().max().item() / scale
    err32 = (ref32.double() - ref64).abs().max().item() / scale
    assert err < max(4 * err32, 1e-6), (err, err32)


@pytest.mark.parametrize("M,N,K", shapes.BF16X3_BIG)
def test_linear_bf16x3_baseline_shapes_full_size(M, N, K):
    """every usf_linear_f32 shape of BASELINE cfg2 / cfg3 / cfg4 / cfg5 at FULL size: rows at the head, in the middle
    and at the tail of the batch (first / last row panels, every column block) against fp64 and fp32 torch-CPU"""
    ext, dev = _ext(), _dev()
    from usflows_amd.engine import FlowEngine
    gd = torch.Generator(device=dev).manual_seed(M + 7 * N + 13 * K)
    A = torch.randn(M, K, generator=gd, device=dev) * 3
    g = torch.Generator().manual_seed(N + K)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    Wd = W.to(dev)
    planes = FlowEngine._split_planes({"mats": {}}, Wd)
    C = torch.full((M, N), float("nan"), device=dev)
    ext.linear(A, Wd, C, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, bias=bias.to(dev), W_split=planes)
    torch.cuda.synchronize()
    assert not torch.isnan(C).any()
    idx = torch.cat([torch.arange(0, 96), torch.arange(M // 2 - 160, M // 2 + 160), torch.arange(M - 300, M),
                     torch.randint(0, M, (256,), generator=g)])
    a = A[idx.to(dev)].cpu()
    ref64 = a.double() @ W.double().t() + bias.double()
    ref32 = a @ W.t() + bias
    scale = ref64.abs().max().item()
    err = (C[idx.to(dev)].cpu().double() - ref64).abs().max().item() / scale
    err32 = (ref32.double() - ref64).abs().max().item() / scale
    # fp32 error class: a sequential fp32 FMA chain over K terms of this data measures 7.4e-7 (K = 784) ... 1.9e-6
    # (K = 3072) against the largest entry (numpy emulation; the CPU sgemm's blocked summation is 3-4x tighter) --
    # the bound is 6e-8 sqrt(K), far inside north_star's 1e-5
    assert err < max(4 * err32, 6e-8 * math.sqrt(K)), (err, err32)
    assert err < 1e-5
    # whole-matrix linearity check: C(A) + C(-A) == 2 bias for every row (catches a wrong row anywhere in the batch)
    C2 = torch.empty(M, N, device=dev)
    ext.linear(-A, Wd, C2, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, bias=bias.to(dev), W_split=planes)
    assert ((C + C2) - 2 * bias.to(dev)).abs().max().item() < 1e-4 * scale


@pytest.mark.parametrize("flags", [dict(act=True), dict(residual=True, res_sign=-1.0), dict(addend=True, act=True),
                                   dict(pre=True), dict(post_mul=True)])
def test_linear_bf16x3_epilogues(flags):
    ext, dev = _ext(), _dev()
    from usflows_amd.engine import FlowEngine
    M, N, K = 777, 392, 256
    g = torch.Generator().manual_seed(11)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g) if flags.get("residual") else None
    add = torch.randn(M, N, generator=g) if flags.get("addend") else None
    pdiv = (torch.rand(K, generator=g) + 0.5) if flags.get("pre") else None
    psub = torch.randn(K, generator=g) if flags.get("pre") else None
    pm = torch.randn(N, generator=g) if flags.get("post_mul") else None
    d = lambda t: None if t is None else t.to(dev)
    planes = FlowEngine._split_planes({"mats": {}}, W.to(dev))
    C = torch.full((M, N), float("nan"), device=dev)
    ext.linear(d(A), d(W), C, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, bias=d(bias), residual=d(res), ldr=N, addend=d(add),
               ldadd=N, pre_div=d(pdiv), pre_sub=d(psub), post_mul=d(pm), res_sign=flags.get("res_sign", 1.0),
               act=1 if flags.get("act") else 0, slope=0.01, W_split=planes)
    torch.cuda.synchronize()
    ref = _linear_ref(A, W, bias, pdiv, psub, res, pm, flags.get("res_sign", 1.0), 0.01, 1 if flags.get("act") else 0,
                      torch.float64) if add is None else None
    if add is not None:
        v = A.double() @ W.double().t() + bias.double() + add.double()
        ref = torch.nn.functional.leaky_relu(v, 0.01)
    err = (C.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-6, err


def test_linear_rejects_bad_args():
    ext, dev = _ext(), _dev()
    A = torch.zeros(4, 6, device=dev)
    W = torch.zeros(4, 6, device=dev)
    C = torch.zeros(4, 4, device=dev)
    with pytest.raises(RuntimeError):
        ext.linear(A, W, C, M=4, N=4, K=6, lda=6, ldw=6, ldc=4)   # K % 4 != 0


@pytest.mark.parametrize("base", ["laplace", "normal", "l1", "l2", "linf"])
@pytest.mark.parametrize("M,D", [(1, 1), (7, 5), (100, 784), (1000, 33)])
def test_base_logprob(base, M, D):
    ext, dev = _ext(), _dev()
    g = torch.Generator().manual_seed(M * 31 + D)
    ld = D + (4 - D % 4) % 4
    z = torch.randn(M, ld, generator=g) * 3
    loc = torch.randn(D, generator=g)
    sc = torch.rand(D, generator=g) + 0.5
    const = -12.5
    out = torch.empty(M, device=dev)
    acc = torch.zeros(2, dtype=torch.float64, device=dev)
    ids = dict(laplace=ext.BASE_LAPLACE, normal=ext.BASE_NORMAL, l1=ext.BASE_LPNORM1, l2=ext.BASE_LPNORM2,
               linf=ext.BASE_LPNORMINF)
    ext.base_logprob(z.to(dev), ld, M, D, ids[base], loc.to(dev), sc.to(dev), const, out, acc)
    torch.cuda.synchronize()
    if base in ("laplace", "normal"):
        # the same constant as a device fp64 scalar (what the engine passes: no host read-back): identical bits
        out2 = torch.empty(M, device=dev)
        ext.base_logprob(z.to(dev), ld, M, D, ids[base], loc.to(dev), sc.to(dev), 0.0, out2, None,
                         logdet_dev=torch.tensor(const, dtype=torch.float64, device=dev))
        assert torch.equal(out, out2)
    zz = z[:, :D].double()
    l64, s64 = loc.double(), sc.double()
    if base == "laplace":
        ref = torch.distributions.Laplace(l64, s64).log_prob(zz).sum(-1) + const
    elif base == "normal":
        ref = torch.distributions.Normal(l64, s64).log_prob(zz).sum(-1) + const
    elif base == "l1":
        ref = (zz - l64).norm(p=1, dim=-1)
    elif base == "l2":
        ref = (zz - l64).norm(p=2, dim=-1)
    else:
        ref = (zz - l64).norm(p=math.inf, dim=-1)
    got = out.cpu().double()
    assert ((got - ref).abs() / ref.abs().clamp_min(1e-3)).max().item() < 2e-6
    a = acc.cpu()
    assert abs(a[0].item() - got.sum().item()) < 1e-6 * max(1.0, abs(got.sum().item()))
    assert a[1].item() == M


def test_scale_and_gather():
    ext, dev = _ext(), _dev()
    M, D = 37, 21
    x = torch.randn(M, D)
    s = torch.randn(D) + 2
    y = torch.empty(M, D, device=dev)
    ext.scale(x.to(dev), D, y, D, M, D, s.to(dev), 1)
    assert torch.equal(y.cpu(), x / s)
    ext.scale(x.to(dev), D, y, D, M, D, s.to(dev), 0)
    assert torch.equal(y.cpu(), x * s)
    idx = torch.tensor([3, 0, -1, 20, 5], dtype=torch.int32)
    out = torch.empty(M, 8, device=dev).fill_(7.0)
    ext.gather_cols(x.to(dev), D, out, 8, M, 5, idx.to(dev))
    o = out.cpu()
    assert torch.equal(o[:, 0], x[:, 3]) and torch.equal(o[:, 1], x[:, 0]) and (o[:, 2] == 0).all()
    assert torch.equal(o[:, 3], x[:, 20]) and (o[:, 5:] == 7).all()


@pytest.mark.parametrize("base", ["laplace", "normal"])
def test_base_sample_statistics_and_determinism(base):
    ext, dev = _ext(), _dev()
    M, D = 20000, 20
    loc = torch.linspace(-1, 1, D)
    sc = torch.linspace(0.5, 2.0, D)
    bid = ext.BASE_LAPLACE if base == "laplace" else ext.BASE_NORMAL
    z1 = torch.empty(M, D, device=dev)
    z2 = torch.empty(M, D, device=dev)
    ext.base_sample(z1, D, M, D, bid, loc.to(dev), sc.to(dev), 1234, 0)
    ext.base_sample(z2, D, M, D, bid, loc.to(dev), sc.to(dev), 1234, 0)
    assert torch.equal(z1, z2)
    # rank substreams: rows [M/2, M) drawn with row_offset reproduce the tail of the full draw
    z3 = torch.empty(M // 2, D, device=dev)
    ext.base_sample(z3, D, M // 2, D, bid, loc.to(dev), sc.to(dev), 1234, 0, row_offset=M // 2)
    assert torch.equal(z3, z1[M // 2:])
    ext.base_sample(z2, D, M, D, bid, loc.to(dev), sc.to(dev), 1235, 0)
    assert not torch.equal(z1, z2)
    z = z1.cpu().double()
    assert torch.isfinite(z).all()
    mean, std = z.mean(0), z.std(0)
    true_std = sc.double() * (math.sqrt(2.0) if base == "laplace" else 1.0)
    assert ((mean - loc.double()).abs() < 5 * true_std / math.sqrt(M)).all()
    assert ((std - true_std).abs() / true_std < 0.05).all()
    # distribution check (KS against the analytic CDF) on one column
    col = ((z[:, 3] - loc[3].double()) / sc[3].double()).sort().values
    if base == "laplace":
        cdf = torch.where(col < 0, 0.5 * torch.exp(col), 1 - 0.5 * torch.exp(-col))
    else:
        cdf = 0.5 * (1 + torch.erf(col / math.sqrt(2.0)))
    emp = (torch.arange(1, M + 1, dtype=torch.float64)) / M
    assert (cdf - emp).abs().max().item() < 1.7 / math.sqrt(M)


# ---- RadialDistribution.sample on the device (distributions.py:283-319, 474-499) -----------------
@pytest.mark.parametrize("p,base_id", [(1.0, 2), (2.0, 3), (float("inf"), 4)])
@pytest.mark.parametrize("D", [7, 64, 784])
def test_radial_sample_kernel(p, base_id, D):
    from usflows_amd import _ext
    _ext.load()
    M = 4096
    g = torch.Generator().manual_seed(D)
    loc = torch.randn(D, generator=g).to("cuda:0")
    r = (0.5 + torch.rand(M, generator=g)).to("cuda:0")
    z = torch.full((M, D + 3), 7.0, device="cuda:0")
    _ext.radial_sample(z, D + 3, M, D, base_id, loc, r, 1234, 0)
    z2 = torch.full((M, D + 3), 7.0, device="cuda:0")
    _ext.radial_sample(z2[: M // 2], D + 3, M // 2, D, base_id, loc, r, 1234, 0)
    _ext.radial_sample(z2[M // 2:], D + 3, M // 2, D, base_id, loc, r[M // 2:].contiguous(), 1234, 0, row_offset=M // 2)
    assert torch.equal(z, z2)                                   # substreams: rows do not depend on the launch split
    assert (z[:, D:] == 7.0).all()                              # padding columns untouched
    u = (z[:, :D] - loc).double().cpu()
    rr = r.double().cpu()
    # the Lp radius of every row is the radius handed in (the property RadialDistribution.log_prob relies on)
    norm = u.abs().sum(1) if p == 1.0 else (u.pow(2).sum(1).sqrt() if p == 2.0 else u.abs().max(1).values)
    assert ((norm - rr).abs() / rr).max().item() < 1e-5
    un = u / rr[:, None]
    if p == float("inf"):
        # exactly one coordinate at +1 (the reference sets 1.0, never -1), uniformly placed; the rest U(-1,1)
        at_one = (un - 1.0).abs() < 1e-6
        assert (at_one.sum(1) >= 1).all()
        idx = at_one.float().argmax(1).double()
        assert abs(idx.mean().item() - (D - 1) / 2) < 4 * D / (12 * M) ** 0.5 + 0.5
        rest = un[~at_one]
        assert abs(rest.mean().item()) < 0.02 and abs(rest.var().item() - 1 / 3) < 0.02
    else:
        assert abs(un.mean().item()) < 4.0 / (M * D) ** 0.5 + 1e-3          # sign-symmetric
        assert abs((un > 0).double().mean().item() - 0.5) < 0.01
        if p == 1.0:
            # Dirichlet(1,..,1) marginals: E|u_i| = 1/D, Var|u_i| = (D-1) / (D^2 (D+1))
            a = un.abs()
            assert abs(a.mean().item() * D - 1.0) < 1e-6
            assert abs(a.var().item() / ((D - 1) / (D * D * (D + 1.0))) - 1.0) < 0.1
        else:
            assert abs(un.pow(2).mean().item() * D - 1.0) < 1e-5
            assert abs(un[:, 0].var().item() * D - 1.0) < 0.15


def test_flow_sample_with_radial_base_runs_on_device():
    from golden_util import load_case
    from model_util import build_flow
    spec, sd, a = load_case("synth_d16_k3_hh1_radial2")
    flow = build_flow(spec, sd, device="cuda:0")
    before = flow.engine().launch_count
    with torch.no_grad():
        xs = flow.sample([2000], seed=11)
        xs2 = flow.sample([2000], seed=11)
        assert flow.engine().launch_count > before
        assert xs.shape == (2000, 16) and torch.isfinite(xs).all()
        # round trip: the latent radius distribution is the norm distribution's (mean of log r for a LogNormal)
        z = flow.backward(xs)
    r = (z - flow.base_distribution.loc).norm(p=2, dim=1)
    nd = flow.base_distribution.norm_distribution
    ref = nd.sample((20000,)).reshape(-1).log()
    assert abs(r.log().mean().item() - ref.mean().item()) < 0.05 * max(1.0, ref.std().item())
    del xs2


@pytest.mark.parametrize("mode", ["f32", "bf16x3"])
@pytest.mark.parametrize("M,N,K", [(5, 12, 20), (300, 256, 392), (4096, 256, 256), (20000, 392, 256)])
def test_linear_gate_epilogue_is_act_grad_of_the_plain_product(M, N, K, mode):
    """USF_ACT_GATE (`addend` read as the saved layer output h): C = (A W^T) * (h > 0 ? 1 : slope) -- the same values as
    usf_linear_f32 followed by usf_act_grad_f32, for the small-batch, exact-f32 and bf16x3 kernels (the shapes of the
    conditioner's data gradients)"""
    ext, dev = _ext(), _dev()
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    ldh = N + 4
    h = torch.randn(M, ldh, generator=g).to(dev)
    planes = None
    if mode == "bf16x3":
        planes = torch.empty(3, N, (K + 31) // 32 * 32, dtype=torch.bfloat16, device=dev)
        ext.pack_weight(W, None, N, None, K, planes=planes)
    ref = torch.empty(M, N, device=dev)
    ext.linear(A, W, ref, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, W_split=planes)
    ext.act_grad(ref, h, M=M, H=N, ldd=N, ldh=ldh, act=ext.ACT_LEAKY_RELU, slope=0.01)
    got = torch.empty(M, N, device=dev)
    ext.linear(A, W, got, M=M, N=N, K=K, lda=K, ldw=K, ldc=N, W_split=planes, act=ext.ACT_GATE, slope=0.01, addend=h, ldadd=ldh)
    torch.cuda.synchronize()
    assert torch.equal(got, ref)


@pytest.mark.parametrize("gated,norm,act", [(True, True, True), (True, False, True), (False, True, False), (False, False, True),
                                            (True, True, False)])
@pytest.mark.parametrize("M,C", [(1, 1), (7, 24), (130, 50), (1000, 256), (4099, 300), (257, 1024), (33, 1500)])
def test_gated_norm_rows(M, C, gated, norm, act):
    """usf_gated_norm_rows_f32 -- the row pass of the vector ConvNet conditioner's blocks (networks.py:206-245): gate,
    torch.nn.LayerNorm and the activation in front of the next Linear -- against the torch fp64 formulation; padded
    strides, zero-filled operand padding, in place on the skip buffer"""
    ext, dev = _ext(), _dev()
    g = torch.Generator().manual_seed(1000 * M + C)
    cp = (C + 3) // 4 * 4
    ld = cp + 4
    skip = torch.randn(M, ld, generator=g).to(dev)
    vg = torch.randn(M, 2 * cp + 8, generator=g).to(dev)
    gamma, beta = (0.5 + torch.rand(C, generator=g)).to(dev), torch.randn(C, generator=g).to(dev)
    out = torch.full((M, ld), 7.0, device=dev)
    out_act = torch.full((M, ld), 7.0, device=dev)
    r = skip[:, :C].double()
    if gated:
        r = r + vg[:, :C].double() * torch.sigmoid(vg[:, cp: cp + C].double())
    if norm:
        r = torch.nn.functional.layer_norm(r, (C,), gamma.double(), beta.double(), 1e-5)
    ext.gated_norm_rows(skip, M=M, C_cols=C, c_pad=cp, ld_skip=ld, vg=vg if gated else None, ld_vg=vg.shape[1], gate_off=cp,
                        gamma=gamma if norm else None, beta=beta if norm else None, eps=1e-5, out=out, ld_out=ld,
                        out_act=out_act if act else None, ld_act=ld, act=ext.ACT_LEAKY_RELU if act else ext.ACT_NONE, slope=0.01)
    torch.cuda.synchronize()
    tol = 2e-6 * max(1.0, r.abs().max().item()) * (4 if norm else 1)
    assert (out[:, :C].double() - r).abs().max().item() < tol
    assert torch.equal(out[:, C:cp], torch.zeros(M, cp - C, device=dev)) and torch.equal(out[:, cp:], torch.full((M, ld - cp), 7.0, device=dev))
    if act:
        assert torch.equal(out_act[:, :cp], torch.nn.functional.leaky_relu(out[:, :cp], 0.01))
    else:
        assert torch.equal(out_act, torch.full((M, ld), 7.0, device=dev))
    # in place on the skip buffer (what the engine does when the block needs no projection)
    skip2 = skip.clone()
    ext.gated_norm_rows(skip2, M=M, C_cols=C, c_pad=cp, ld_skip=ld, vg=vg if gated else None, ld_vg=vg.shape[1], gate_off=cp,
                        gamma=gamma if norm else None, beta=beta if norm else None, eps=1e-5, out=skip2, ld_out=ld)
    torch.cuda.synchronize()
    assert torch.equal(skip2[:, :cp], out[:, :cp])


def test_gated_norm_rows_rejects_bad_args():
    ext, dev = _ext(), _dev()
    x = torch.zeros(4, 8, device=dev)
    with pytest.raises(RuntimeError):
        ext.gated_norm_rows(x, M=4, C_cols=8, ld_skip=8)                                    # no output
    with pytest.raises(RuntimeError):
        ext.gated_norm_rows(x, M=4, C_cols=8, ld_skip=4, out=x, ld_out=8)                   # stride shorter than the row
    with pytest.raises(RuntimeError):
        ext.gated_norm_rows(x, M=4, C_cols=8, ld_skip=8, out=x, ld_out=8, gamma=x)          # gamma without beta
    with pytest.raises(RuntimeError):
        ext.gated_norm_rows(x, M=4, C_cols=5000, c_pad=5000, ld_skip=5000, out=x, ld_out=5000)


def test_coupling_hidden_out_and_gate_need_the_bf16x3_kernel():
    DEV = "cuda:0"
    """usf_coupling_desc::hidden_out / USF_ACT_GATE: served by the split-precision kernel only -- a descriptor that another
    kernel would serve is rejected, not silently run without the side output"""
    from usflows_amd import _ext
    lib = _ext.load()
    d = _ext.CouplingDesc()
    M, n = 2048, 64
    z = torch.zeros(M, 2 * n, device=DEV)
    W = torch.zeros(256, 256, device=DEV)
    b = torch.zeros(256, device=DEV)
    H = torch.zeros(M, 256, device=DEV)
    d.z = d.out = z.data_ptr(); d.ldz = d.ldo = 2 * n; d.M = M
    d.off_pass, d.n_pass, d.off_trans, d.n_trans = 0, n, n, n
    d.n_hidden = 1; d.hidden[0] = 256
    d.W_in, d.ldw_in, d.b_in = W.data_ptr(), 256, b.data_ptr()
    d.W_out, d.ldw_out, d.b_out = W.data_ptr(), 256, b.data_ptr()
    d.sign, d.slope, d.act = 1.0, 0.01, _ext.ACT_LEAKY_RELU
    d.hidden_out[0], d.ld_hidden_out = H.data_ptr(), 256          # no split planes given: the exact-f32 kernel would serve it
    import ctypes as C
    st = torch.cuda.current_stream().cuda_stream
    assert lib.usf_coupling_additive_f32(C.byref(d), st) == -2 and b"hidden_out / USF_ACT_GATE are served by" in lib.usf_last_error()
    d.hidden_out[0] = None
    d.act = _ext.ACT_GATE
    assert lib.usf_coupling_additive_f32(C.byref(d), st) == -2 and b"hidden_out / USF_ACT_GATE are served by" in lib.usf_last_error()


@pytest.mark.parametrize("mode", ["plain", "hidden_out", "gate", "ctx"])
@pytest.mark.parametrize("M,n_pass,n_trans,hidden", [(1, 4, 4, [32]), (32, 8, 12, [32, 32]), (33, 52, 48, [32, 20]), (200, 24, 20, [32, 32, 24]), (70, 32, 32, [64]),
                                                      (256, 4, 8, [7, 5])])
def test_tiny_layer_coupling_kernel(M, n_pass, n_trans, hidden, mode):
    """usf_coupling_additive_f32 at launch-bound batches with tiny layers (usf_coupling_tiny.hip: the reference's live flat
    configuration, gaussian_mixture.yaml): forward, the saved hidden activations, the backward chain in gate mode and the context
    branch against the fp64 formulation (transforms.py:277-306, networks.py:739-751)"""
    import ctypes as C
    DEV = "cuda:0"
    from usflows_amd import _ext
    lib = _ext.load()
    g = torch.Generator().manual_seed(M + n_pass)
    nh = len(hidden)
    ldz = n_pass + (n_trans + 3) // 4 * 4 + 4
    off_trans = n_pass
    z = torch.randn(M, ldz, generator=g)
    dims = [n_pass] + hidden + [n_trans]
    Ws = [torch.randn(dims[i + 1], dims[i], generator=g) / dims[i] ** 0.5 for i in range(nh + 1)]
    bs = [torch.randn(dims[i + 1], generator=g) * 0.1 for i in range(nh + 1)]
    lds_ = [(dims[i] + 3) // 4 * 4 + 4 for i in range(nh + 1)]                       # row strides with slack, multiples of 4
    Wp = [torch.zeros(dims[i + 1], lds_[i]) for i in range(nh + 1)]
    for i in range(nh + 1):
        Wp[i][:, : dims[i]] = Ws[i]
    hmax = (max(hidden) + 3) // 4 * 4
    gates = [torch.randn(M, hmax, generator=g) for _ in range(nh)]
    ctx, Wc, bc = torch.randn(M, generator=g), torch.randn(hidden[0], generator=g), torch.randn(hidden[0], generator=g)
    slope, sign = 0.01, -1.0
    dev = lambda t: t.to(DEV).contiguous()
    zd, Wd, bd, gd = dev(z), [dev(w) for w in Wp], [dev(b) for b in bs], [dev(t) for t in gates]
    houts = [torch.full((M, hmax), 9.0, device=DEV) for _ in range(nh)]
    ctxd, Wcd, bcd = dev(ctx), dev(Wc), dev(bc)
    d = _ext.CouplingDesc()
    d.z = d.out = zd.data_ptr(); d.ldz = d.ldo = ldz; d.M = M
    d.off_pass, d.n_pass, d.off_trans, d.n_trans = 0, n_pass, off_trans, n_trans
    d.n_hidden = nh
    for i, h in enumerate(hidden):
        d.hidden[i] = h
    d.W_in, d.ldw_in, d.b_in = Wd[0].data_ptr(), lds_[0], bd[0].data_ptr()
    for i in range(1, nh):
        d.W_hid[i - 1], d.ldw_hid[i - 1], d.b_hid[i - 1] = Wd[i].data_ptr(), lds_[i], bd[i].data_ptr()
    d.W_out, d.ldw_out, d.b_out = Wd[nh].data_ptr(), lds_[nh], bd[nh].data_ptr()
    d.sign, d.slope, d.act = sign, slope, _ext.ACT_GATE if mode == "gate" else _ext.ACT_LEAKY_RELU
    if mode in ("hidden_out", "gate"):
        for i in range(nh):
            d.hidden_out[i] = houts[i].data_ptr()
        d.ld_hidden_out = hmax
    if mode == "gate":
        for i in range(nh):
            d.gate[i] = gd[i].data_ptr()
        d.ld_gate = hmax
    if mode == "ctx":
        d.context, d.W_ctx, d.b_ctx = ctxd.data_ptr(), Wcd.data_ptr(), bcd.data_ptr()
    rc = lib.usf_coupling_additive_f32(C.byref(d), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.usf_last_error()
    torch.cuda.synchronize()
    # fp64 formulation
    h = z[:, :n_pass].double()
    hs = []
    for i in range(nh):
        v = h @ Ws[i].double().t() + bs[i].double()
        if mode == "gate":
            v = v * torch.where(gates[i][:, : hidden[i]].double() > 0, 1.0, slope)
        else:
            if i == 0 and mode == "ctx":
                v = v + (ctx.double()[:, None] * Wc.double()[None, :] + bc.double()[None, :])
            v = torch.where(v > 0, v, v * slope)
        hs.append(v)
        h = v
    t = h @ Ws[nh].double().t() + bs[nh].double()
    ref = z.double().clone()
    ref[:, off_trans: off_trans + n_trans] += sign * t
    got = zd.cpu().double()
    s = max(1.0, ref.abs().max().item())
    assert (got - ref).abs().max().item() < 2e-6 * s                                   # everything else of z untouched, too
    if mode in ("hidden_out", "gate"):
        for i in range(nh):
            hv = houts[i].cpu().double()
            assert (hv[:, : hidden[i]] - hs[i]).abs().max().item() < 2e-6 * max(1.0, hs[i].abs().max().item())
            assert (hv[:, hidden[i]:] == 9.0).all()                                    # columns beyond the layer's width are not written
