"""Planes pipeline on a real MI355X, through the C ABI: usf_pack_planes_f32 / usf_gemm_planes_bf16x3 against torch
reference arithmetic (kernel level), and whole flows through the planes launch plan against the reference's outputs."""
import math

import pytest
import torch

import emulator
import shapes
from golden_util import case_names, load_case
from model_util import build_flow

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ext():
    from usflows_amd import _ext
    _ext.load()
    return _ext


FMT = {"bf16x3": 0, "f16x2": 1}


def _view(buf, M, nkb, fmt=0):
    npl, dt = (2, torch.float16) if fmt == 1 else (3, torch.bfloat16)
    return buf.cpu().view(dt).view(-(-M // 16), nkb, npl, 64, 8)


def _weight_planes(W, n_rows_pad, fmt=0):
    """[3, rows, K] bf16 / [2, rows, K] fp16 planes of logical W [n, K] with the slot permutation on K (K % 32 == 0)"""
    n, K = W.shape
    Wp = torch.zeros(n_rows_pad, K)
    Wp[:n] = W
    phys = torch.tensor([32 * (c // 32) + emulator._slot_feature(c % 32) for c in range(K)])
    Wp = Wp[:, phys]
    dt = torch.float16 if fmt == 1 else torch.bfloat16
    p1 = Wp.to(dt)
    r = Wp - p1.float()
    p2 = r.to(dt)
    if fmt == 1:
        return torch.stack([p1, p2]).contiguous()
    p3 = (r - p2.float()).to(dt)
    return torch.stack([p1, p2, p3]).contiguous()


@pytest.mark.parametrize("rows", [False, True])
@pytest.mark.parametrize("fmt", ["bf16x3", "f16x2"])
@pytest.mark.parametrize("M,D,nkb", [(1, 5, 1), (37, 50, 2), (1000, 784, 25), (4099, 96, 3)])
def test_pack_planes_is_the_exact_three_way_split(M, D, nkb, fmt, rows):
    """rows: with the promise src_cols (every index below it) the kernel stages whole rows in LDS -- the same planes, bit for bit"""
    ext = _ext()
    fmt = FMT[fmt]
    g = torch.Generator().manual_seed(M + D)
    x = torch.randn(M, D + 3, generator=g) * 5
    perm = torch.randperm(D, generator=g)
    idx = torch.full((32 * nkb,), -1, dtype=torch.int32)
    pos = torch.randperm(32 * nkb, generator=g)[:D]
    idx[pos] = perm.to(torch.int32)
    pdiv = torch.rand(32 * nkb, generator=g) + 0.5
    psub = torch.randn(32 * nkb, generator=g)
    buf = torch.zeros(ext.planes_bytes(M, nkb, fmt), dtype=torch.uint8, device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    ext.pack_planes(x.to(DEV), buf, M=M, nkb=nkb, idx=idx.to(DEV), pre_div=pdiv.to(DEV), pre_sub=psub.to(DEV), fmt=fmt,
                    range_flag=flag, src_cols=(D + 3) if rows else 0)
    torch.cuda.synchronize()
    Mp = -(-M // 16) * 16
    got = emulator.planes_decode(_view(buf, M, nkb, fmt), Mp)
    ref = torch.zeros(M, 32 * nkb)
    ok = idx >= 0
    ref[:, ok] = x[:, idx[ok].long()] / pdiv[ok] - psub[ok]
    if fmt == 0:
        assert torch.equal(got[:M], ref)    # p1 + p2 + p3 == x exactly (24 significant bits in three bf16)
    else:                                   # two fp16 planes: 22 significant bits (11 + 11), absolute floor 2^-25
        assert ((got[:M] - ref).abs() <= ref.abs() * 2.0 ** -22 + 2.0 ** -25).all()
        # bitwise what the documented split gives
        hi = ref.to(torch.float16)
        assert torch.equal(got[:M], hi.float() + (ref - hi.float()).to(torch.float16).float())
        assert int(flag.item()) == 0
    assert (got[M:] == 0).all()             # padding rows of the last panel: zeros


@pytest.mark.parametrize("base", ["laplace", "normal"])
@pytest.mark.parametrize("M,D", [(33, 40), (5000, 784)])
def test_pack_planes_with_the_base_gradient_on_the_way_in(M, D, base):
    """grad mode (the head of the training backward): planes of g = row_weight * d/dz base(z), bit for bit what
    usf_base_logprob_grad_f32 followed by a plain pack gives"""
    ext = _ext()
    g = torch.Generator().manual_seed(M)
    z = (torch.randn(M, D + 4, generator=g) * 3).to(DEV)
    z[3, 5] = 0.25                                                      # on the Laplace kink: loc[5] = 0.25 below
    w = (torch.randn(M, generator=g)).to(DEV)
    loc = (torch.randn(D, generator=g)).to(DEV)
    loc[5] = 0.25
    scale = (torch.rand(D, generator=g) + 0.5).to(DEV)
    bid = ext.BASE_LAPLACE if base == "laplace" else ext.BASE_NORMAL
    nkb = -(-D // 32)
    idx = torch.full((32 * nkb,), -1, dtype=torch.int32)
    idx[:D] = torch.arange(D, dtype=torch.int32)
    idx = idx.to(DEV)
    b1 = torch.zeros(ext.planes_bytes(M, nkb), dtype=torch.uint8, device=DEV)
    b2 = torch.zeros_like(b1)
    ext.pack_planes(z, b1, M=M, nkb=nkb, idx=idx, ld=D + 4, src_cols=D, grad=(bid, w, loc, scale))
    gbuf = torch.zeros(M, D, device=DEV)
    ext.base_logprob_grad(z, D + 4, w, M, D, bid, loc, scale, gbuf, D)
    ext.pack_planes(gbuf, b2, M=M, nkb=nkb, idx=idx)
    torch.cuda.synchronize()
    assert torch.equal(b1, b2)
    assert emulator.planes_decode(_view(b1, M, nkb), M)[3, 5].item() == 0.0 or base == "normal"


CASES = [
    # M, a_nkb, a_kb0, nk, out blocks (c_nkb, c_kb0, c_kbn) or fp32 N, flags
    (100, 3, 0, 3, (4, 1, 2), dict(bias=True)),
    (1000, 25, 0, 25, (25, 0, 25), dict(bias=True)),                              # affine, TN = 5
    (3000, 25, 12, 13, (8, 0, 8), dict(bias=True, act=True)),                     # conditioner first layer (K range), TN = 4
    (3000, 8, 0, 8, (25, 0, 13), dict(bias=True, residual=True, sign=-1.0)),      # conditioner last layer, in place
    (3000, 8, 0, 8, (25, 12, 13), dict(bias=True, residual=True, sign=1.0)),
    (8205, 25, 0, 25, 784, dict(bias=True, post_mul=True)),                       # fp32 output, ragged rows / columns
    (513, 4, 1, 2, 77, dict()),                                                   # fp32 output, TN = 4
    (65536, 25, 0, 25, (25, 0, 25), dict(bias=True)),                             # BASELINE cfg2 affine at full size
    (32768, 96, 0, 96, (96, 0, 96), dict(bias=True)),                             # BASELINE cfg4 affine at full size
]


def test_fp16_range_guard_flags_values_fp16_cannot_carry():
    ext = _ext()
    x = torch.ones(40, 32)
    idx = torch.arange(32, dtype=torch.int32).to(DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    buf = torch.zeros(ext.planes_bytes(40, 1, 1), dtype=torch.uint8, device=DEV)
    for bad, expect in ((None, 0), (64999.0, 0), (65000.0, 1), (-1e9, 1), (float("nan"), 1), (float("inf"), 1)):
        xx = x.clone()
        if bad is not None:
            xx[37, 5] = bad
        flag.zero_()
        ext.pack_planes(xx.to(DEV), buf, M=40, nkb=1, idx=idx, fmt=1, range_flag=flag)
        assert int(flag.item()) == expect, bad
    # GEMM output that leaves fp16's range: flagged by the planes epilogue; fp32 output: only non-finite values are
    W = _weight_planes(torch.full((32, 32), 100.0), 32, 1).to(DEV)
    A = torch.zeros(ext.planes_bytes(40, 1, 1), dtype=torch.uint8)
    emulator.planes_encode(_view(A, 40, 1, 1), torch.full((40, 32), 30.0), 0)        # 32 * 30 * 100 = 96000 > 65000
    flag.zero_()
    ext.gemm_planes(A.to(DEV), W, M=40, a_nkb=1, nk=1, C_planes=buf, c_nkb=1, c_kbn=1, fmt=1, range_flag=flag)
    assert int(flag.item()) == 1
    flag.zero_()
    C = torch.empty(40, 32, device=DEV)
    ext.gemm_planes(A.to(DEV), W, M=40, a_nkb=1, nk=1, C_f32=C, ldc=32, N=32, fmt=1, range_flag=flag)
    assert int(flag.item()) == 0 and torch.allclose(C.cpu(), torch.full((40, 32), 96000.0))


@pytest.mark.parametrize("fmt", ["bf16x3", "f16x2"])
@pytest.mark.parametrize("M,a_nkb,a_kb0,nk,out,flags", CASES)
def test_gemm_planes_parity(M, a_nkb, a_kb0, nk, out, flags, fmt):
    ext = _ext()
    fmt = FMT[fmt]
    g = torch.Generator().manual_seed(M + 31 * nk)
    K = 32 * nk
    f32out = isinstance(out, int)
    n_out = out if f32out else 32 * out[2]
    w_rows = -(-n_out // 32) * 32
    X = torch.randn(M, 32 * a_nkb, generator=g) * 3
    W = torch.randn(n_out, K, generator=g) / math.sqrt(K)
    bias = torch.randn(w_rows, generator=g) if flags.get("bias") else None
    if bias is not None:
        bias[n_out:] = 0
    Abuf = torch.zeros(ext.planes_bytes(M, a_nkb, fmt), dtype=torch.uint8)
    emulator.planes_encode(_view(Abuf, M, a_nkb, fmt), X, 0)
    Wp = _weight_planes(W, w_rows, fmt).to(DEV)
    d = lambda t: None if t is None else t.to(DEV)
    # reference on sampled rows (head / middle / tail) in fp64 and fp32
    idx = torch.unique(torch.cat([torch.arange(0, min(M, 40)), torch.arange(max(M // 2 - 20, 0), min(M // 2 + 20, M)),
                                  torch.arange(max(M - 40, 0), M)]))
    Xk = X[idx][:, 32 * a_kb0: 32 * (a_kb0 + nk)]
    # (the reference multiplies the fp32 values; the kernel's operands are those values as planes: exact in bf16x3, rounded
    # to 22 significant bits in fp16x2 -- the error that format is allowed, inside the same gate)

    def ref(dt):
        v = Xk.to(dt) @ W.to(dt).t()
        if bias is not None:
            v = v + bias[:n_out].to(dt)
        if flags.get("act"):
            v = torch.nn.functional.leaky_relu(v, 0.01)
        return v

    if f32out:
        pm = torch.randn(w_rows, generator=g) if flags.get("post_mul") else None
        C = torch.full((M, n_out + 3), float("nan"), device=DEV)
        ext.gemm_planes(Abuf.to(DEV), Wp, M=M, a_nkb=a_nkb, a_kb0=a_kb0, nk=nk, bias=d(bias), post_mul=d(pm), C_f32=C,
                        ldc=n_out + 3, N=n_out, fmt=fmt)
        torch.cuda.synchronize()
        assert torch.isnan(C[:, n_out:]).all() and not torch.isnan(C[:, :n_out]).any()
        got = C[:, :n_out].cpu()[idx].double()
        r64, r32 = ref(torch.float64), ref(torch.float32)
        if pm is not None:
            r64, r32 = r64 * pm[:n_out].double(), r32 * pm[:n_out]
    else:
        c_nkb, c_kb0, c_kbn = out
        R = torch.randn(M, 32 * c_nkb, generator=g) * 2
        Cbuf = torch.zeros(ext.planes_bytes(M, c_nkb, fmt), dtype=torch.uint8)
        emulator.planes_encode(_view(Cbuf, M, c_nkb, fmt), R, 0)
        R = emulator.planes_decode(_view(Cbuf, M, c_nkb, fmt), M)          # what the buffer holds (fp16x2: R rounded to 22 bits)
        Cd = Cbuf.to(DEV)
        ext.gemm_planes(Abuf.to(DEV), Wp, M=M, a_nkb=a_nkb, a_kb0=a_kb0, nk=nk, bias=d(bias), C_planes=Cd, c_nkb=c_nkb,
                        c_kb0=c_kb0, c_kbn=c_kbn, residual=Cd if flags.get("residual") else None,
                        res_sign=flags.get("sign", 1.0), act=1 if flags.get("act") else 0, slope=0.01, fmt=fmt)
        torch.cuda.synchronize()
        full = emulator.planes_decode(_view(Cd, M, c_nkb, fmt), M)
        lo, hi = 32 * c_kb0, 32 * (c_kb0 + c_kbn)
        # blocks outside the output range are untouched
        assert torch.equal(full[:, :lo], R[:, :lo]) and torch.equal(full[:, hi:], R[:, hi:])
        got = full[idx][:, lo:hi].double()
        r64, r32 = ref(torch.float64), ref(torch.float32)
        if flags.get("residual"):
            s = flags.get("sign", 1.0)
            r64, r32 = R[idx][:, lo:hi].double() + s * r64, R[idx][:, lo:hi] + s * r32
    scale = r64.abs().max().item()
    err = (got - r64).abs().max().item() / scale
    err32 = (r32.double() - r64).abs().max().item() / scale
    assert err < max(4 * err32, 6e-8 * math.sqrt(K)), (err, err32)
    assert err < 1e-5


def test_gemm_planes_rejects_bad_args():
    ext = _ext()
    A = torch.zeros(ext.planes_bytes(64, 2), dtype=torch.uint8, device=DEV)
    W = torch.zeros(3, 32, 64, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(RuntimeError):
        ext.gemm_planes(A, W, M=64, a_nkb=2, a_kb0=1, nk=2, C_planes=A, c_nkb=2, c_kbn=1)      # K range past the buffer
    with pytest.raises(RuntimeError):
        ext.gemm_planes(A, W, M=64, a_nkb=2, nk=2)                                             # no output


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("fmt", ["bf16x3", "f16x2"])
@pytest.mark.parametrize("name", case_names())
def test_golden_parity_through_the_planes_plan(name, fmt, fused):
    """every golden case of the reference through the planes launch plan (forced: planes_min_rows = 0), activations as
    bf16x3 or as fp16x2 planes, couplings as ONE fused launch (usf_coupling_planes) or as a chain of GEMMs on planes"""
    spec, sd, a = load_case(name)
    if a.get("context") is not None or spec.soft_training:
        pytest.skip("context: served by the fp32-activation path")
    if spec.conditioner == "ConvNet" and (spec.extra.get("gating") or spec.extra.get("normalize_layers")):
        pytest.skip("gate / layer-norm blocks: chain of fp32 ops (FlowEngine._general_coupling_ops), no planes plan")
    flow = build_flow(spec, sd, device=DEV)
    eng = flow.engine()
    eng.use_planes, eng.planes_min_rows, eng.gemm_mode = True, 0, fmt
    eng.use_fused_coupling, eng.fused_min_rows = fused, 0
    with torch.no_grad():
        lp = flow.log_prob(a["x"].to(DEV))
        z = flow.backward(a["x"].to(DEV))
        xf = flow._forward(a["zin"].to(DEV))
    assert any(p.get("planes") and p["planes_fmt"] == FMT[fmt] for p in eng._plans.values()), "planes plan was not built"
    if spec.base in ("laplace", "normal"):
        assert any(p.get("n_part", 0) >= 1 for p in eng._plans.values()), "base density was not reduced in the last GEMM's epilogue"
    from usflows_amd import _ext as E
    has_fused = any(p["arr"][j].kind == E.OP_COUPLING_PLANES for p in eng._plans.values() if p.get("planes")
                    for j in range(p["n"]))
    assert has_fused == (fused and not (fmt == "bf16x3" and len(spec.hidden_dims) == 3))
    assert eng.f16_fallbacks == 0
    rel = lambda u, v: ((u.double().cpu() - v.double()).abs() / v.double().abs().clamp_min(1e-30)).max().item()
    tol = 1e-5
    if fmt == "f16x2":
        # 22 significant bits per operand: on the ill-conditioned default-initialised cases, where the reference's OWN
        # fp32 run is 2e-6 .. 7e-6 away from its fp64 run, the gate is 3x that gap (1e-5 everywhere else)
        tol = max(1e-5, 3 * rel(a["log_prob32"], a["log_prob64"]))
    assert rel(lp, a["log_prob64"]) < tol and rel(lp, a["log_prob32"]) < tol, name
    s = max(1.0, a["backward64"].abs().max().item())
    assert (z.cpu().double() - a["backward64"]).abs().max().item() < 2e-5 * s
    s = max(1.0, a["forward64"].abs().max().item())
    assert (xf.cpu().double() - a["forward64"]).abs().max().item() < 2e-5 * s


@pytest.mark.parametrize("B", [1, 17, 8191, 8200])
def test_planes_plan_ragged_rows_vs_fp32_plan(B):
    """row counts that are not multiples of the 16-row panels / 256-row blocks: planes plan == fp32-activation plan"""
    from oracle import usflows_oracle as orc
    spec = orc.FlowSpec(72, 3, [40, 24], householder=1, affine_conjugation=True)
    sd = orc.synth_state_dict(spec, seed=31)
    flow = build_flow(spec, sd, device=DEV)
    eng = flow.engine()
    x = torch.rand(B, 72, generator=torch.Generator().manual_seed(B)).to(DEV)
    with torch.no_grad():
        eng.use_planes, eng.planes_min_rows = True, 0
        lp1, z1 = flow.log_prob(x), flow.backward(x)
        eng.use_planes = False
        lp2, z2 = flow.log_prob(x), flow.backward(x)
    assert ((lp1 - lp2).abs() / lp2.abs()).max().item() < 5e-6
    assert (z1 - z2).abs().max().item() < 1e-4 * max(1.0, z2.abs().max().item())
    ref = orc.flow_log_prob(orc.to_dtype(sd, torch.float64), spec, x[:64].cpu().double())
    assert ((lp1[:64].cpu().double() - ref).abs() / ref.abs()).max().item() < 1e-5


@pytest.mark.parametrize("base", ["laplace", "normal"])
@pytest.mark.parametrize("D,B", [(72, 1), (70, 17), (72, 8200), (784, 4099), (1000, 300)])
def test_base_density_in_the_last_gemms_epilogue(D, B, base, monkeypatch):
    """Flow.log_prob on a planes plan: the Laplace / Normal base density reduced by the last GEMM's epilogue (one partial sum
    per row and column block + the row-sum tail, z never stored) against the fp64 oracle and against the plan that stores z
    and runs usf_base_logprob_f32 over it; per-feature loc / scale, ragged row counts, D not a multiple of 4, 1 .. 8 column
    blocks, and the data-parallel sums"""
    from oracle import usflows_oracle as orc
    from usflows_amd.config import config
    g = torch.Generator().manual_seed(D + B)
    spec = orc.FlowSpec(D, 2, [40, 24], householder=0, affine_conjugation=True, base=base,
                        base_loc=torch.randn(D, generator=g) * 0.3, base_scale=0.5 + torch.rand(D, generator=g))
    sd = orc.synth_state_dict(spec, seed=5)
    flow = build_flow(spec, sd, device=DEV)
    eng = flow.engine()
    eng.use_planes, eng.planes_min_rows = True, 0
    x = torch.rand(B, D, generator=g).to(DEV)
    sums = torch.zeros(2, dtype=torch.float64, device=DEV)
    with torch.no_grad():
        lp1 = flow._log_prob_device(x, None, sums)
        assert any(p.get("n_part", 0) >= 1 for p in eng._plans.values()), "the fused tail was not taken"
        n_part = max(p.get("n_part", 0) for p in eng._plans.values())
        assert n_part == {72: 1, 70: 1, 784: 5, 1000: 8}[D]
        monkeypatch.setattr(config, "base_in_epilogue", False)
        eng._plans.clear()
        lp2 = flow.log_prob(x)
        assert not any(p.get("n_part", 0) >= 1 for p in eng._plans.values())
    assert torch.isfinite(lp1).all()
    assert ((lp1 - lp2).abs() / lp2.abs()).max().item() < 2e-6
    assert abs(sums[0].item() - lp1.double().sum().item()) < 1e-9 * abs(sums[0].item()) + 1e-9 and sums[1].item() == B
    n = min(B, 64)
    ref = orc.flow_log_prob(orc.to_dtype(sd, torch.float64), spec, x[:n].cpu().double())
    assert ((lp1[:n].cpu().double() - ref).abs() / ref.abs()).max().item() < 1e-5


def test_clock_meter_counts_the_gemm_blocks_lifetimes():
    """usf_set_clock_buffer (bench.py: roofline.clock_mhz): set -> the planes GEMM and the MFMA probe add shader cycles and
    100 MHz ticks of their blocks; unset -> the counters stay untouched"""
    ext = _ext()
    M, nkb = 8192, 25
    A = torch.zeros(ext.planes_bytes(M, nkb), dtype=torch.uint8, device=DEV)
    W = torch.zeros(3, 800, 800, dtype=torch.bfloat16, device=DEV)
    C_ = torch.zeros_like(A)
    with ext.clock_meter(DEV) as cm:
        for _ in range(3):
            ext.gemm_planes(A, W, M=M, a_nkb=nkb, nk=nkb, C_planes=C_, c_nkb=nkb, c_kbn=nkb)
        torch.cuda.synchronize()
    mhz = cm.mhz()
    assert mhz is not None and 300.0 < mhz < 2600.0, mhz
    before = cm.buf.clone()
    ext.gemm_planes(A, W, M=M, a_nkb=nkb, nk=nkb, C_planes=C_, c_nkb=nkb, c_kbn=nkb)
    torch.cuda.synchronize()
    assert torch.equal(cm.buf, before)
    with ext.clock_meter(DEV) as cm2:
        ext.mfma_probe(DEV, iters=50, repeats=1)
    assert 300.0 < cm2.mhz() < 2600.0


def test_fp16_overflow_falls_back_to_bf16x3_planes():
    """a flow whose activations leave fp16's range (inputs ~1e6): the fp16x2 pass raises its range flag, the engine
    redoes it with bf16x3 planes -- same result as the bf16x3 mode, never an inf / NaN from the plane format"""
    from oracle import usflows_oracle as orc
    spec = orc.FlowSpec(64, 3, [48, 48], householder=0)
    sd = orc.synth_state_dict(spec, seed=77)
    flow = build_flow(spec, sd, device=DEV)
    eng = flow.engine()
    eng.use_planes, eng.planes_min_rows = True, 0
    x = (torch.rand(300, 64, generator=torch.Generator().manual_seed(1)) * 2e6).to(DEV)
    with torch.no_grad():
        eng.gemm_mode = "f16x2"
        lp1 = flow.log_prob(x)
        assert eng.f16_fallbacks == 1
        small = flow.log_prob(x * 1e-6)          # in range again: fp16x2 planes, no fallback
        assert eng.f16_fallbacks == 1
        eng.gemm_mode = "bf16x3"
        lp2 = flow.log_prob(x)
    assert torch.isfinite(lp1).all() and torch.equal(lp1, lp2) and torch.isfinite(small).all()
    ref = orc.flow_log_prob(orc.to_dtype(sd, torch.float64), spec, x.cpu().double())
    assert ((lp1.cpu().double() - ref).abs() / ref.abs()).max().item() < 1e-5


@pytest.mark.parametrize("fmt", ["bf16x3", "f16x2"])
@pytest.mark.parametrize("M,nh", [(40, 1), (3000, 2), (65536, 2), (2051, 3)])
def test_coupling_planes_kernel_vs_reference_arithmetic(M, nh, fmt):
    """usf_coupling_planes at the cfg2 layer shape (25 blocks; conditioning blocks 12..24, transformed blocks 0..12 sharing
    the straddling block 12; hidden widths 256 / 200 / 136 padded to 256) against torch fp64 / fp32 of the same layer"""
    ext = _ext()
    if fmt == "bf16x3" and nh == 3:
        pytest.skip("three hidden layers in bf16x3 take the GEMM chain (register budget)")
    f = FMT[fmt]
    g = torch.Generator().manual_seed(M + nh)
    nkb, kb_p0, nk_p, kb_t0, nk_t = 25, 12, 13, 0, 13
    widths = [256, 200, 136][:nh]
    X = torch.randn(M, 32 * nkb, generator=g) * 2
    is_p = torch.zeros(800, dtype=torch.bool); is_p[392:784] = True       # conditioning features
    is_t = torch.zeros(800, dtype=torch.bool); is_t[:392] = True          # transformed features
    Ws, bs = [], []
    k_in = 32 * nk_p
    W0 = torch.randn(widths[0], k_in, generator=g) / math.sqrt(392)
    W0[:, ~is_p[32 * kb_p0: 32 * (kb_p0 + nk_p)]] = 0                      # zero weights on the other set's features
    Ws.append(W0); bs.append(torch.randn(widths[0], generator=g) * 0.1)
    for j in range(1, nh):
        Ws.append(torch.randn(widths[j], widths[j - 1], generator=g) / math.sqrt(widths[j - 1]))
        bs.append(torch.randn(widths[j], generator=g) * 0.1)
    Wo = torch.randn(32 * nk_t, widths[-1], generator=g) / math.sqrt(widths[-1])
    bo = torch.randn(32 * nk_t, generator=g) * 0.1
    dead = ~is_t[32 * kb_t0: 32 * (kb_t0 + nk_t)]
    Wo[dead] = 0; bo[dead] = 0

    def pad(W, rows, cols):
        o = torch.zeros(rows, cols); o[: W.shape[0], : W.shape[1]] = W; return o

    def padv(b, n):
        o = torch.zeros(n); o[: b.numel()] = b; return o

    zbuf = torch.zeros(ext.planes_bytes(M, nkb, f), dtype=torch.uint8)
    emulator.planes_encode(_view(zbuf, M, nkb, f), X, 0)
    X = emulator.planes_decode(_view(zbuf, M, nkb, f), M)                  # what the buffer holds
    zd = zbuf.to(DEV)
    d = ext.CouplingPlanesDesc()
    keep = []
    d.z, d.z_nkb, d.M = zd.data_ptr(), nkb, M
    d.kb_p0, d.nk_p, d.kb_t0, d.nk_t = kb_p0, nk_p, kb_t0, nk_t
    d.n_hidden, d.hidden_padded = nh, 256
    Wi = _weight_planes(pad(Ws[0], 256, k_in), 256, f).to(DEV); keep.append(Wi)
    d.W_in, d.ldw_in, d.w_in_plane = Wi.data_ptr(), Wi.shape[2], Wi.shape[1] * Wi.shape[2]
    b0 = padv(bs[0], 256).to(DEV); keep.append(b0); d.b_in = b0.data_ptr()
    for j in range(1, nh):
        Wh = _weight_planes(pad(Ws[j], 256, 256), 256, f).to(DEV); keep.append(Wh)
        bh = padv(bs[j], 256).to(DEV); keep.append(bh)
        d.W_hid[j - 1], d.b_hid[j - 1] = Wh.data_ptr(), bh.data_ptr()
        d.ldw_hid, d.w_hid_plane = Wh.shape[2], Wh.shape[1] * Wh.shape[2]
    Wod = _weight_planes(pad(Wo, 32 * nk_t, 256), 32 * nk_t, f).to(DEV); keep.append(Wod)
    d.W_out, d.ldw_out, d.w_out_plane = Wod.data_ptr(), Wod.shape[2], Wod.shape[1] * Wod.shape[2]
    bod = bo.to(DEV); keep.append(bod); d.b_out = bod.data_ptr()
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    d.sign, d.slope, d.act, d.format, d.range_flag = -1.0, 0.01, 1, f, flag.data_ptr()
    import ctypes
    ext.check(ext.load().usf_coupling_planes(ctypes.byref(d), ext.current_stream(zd.device)), "usf_coupling_planes")
    torch.cuda.synchronize()
    got = emulator.planes_decode(_view(zd, M, nkb, f), M)
    idx = torch.unique(torch.cat([torch.arange(0, min(M, 40)), torch.arange(max(M // 2 - 20, 0), min(M // 2 + 20, M)),
                                  torch.arange(max(M - 40, 0), M)]))

    def ref(dt):
        h = X[idx][:, 32 * kb_p0: 32 * (kb_p0 + nk_p)].to(dt)
        for W, b in zip(Ws, bs):
            h = torch.nn.functional.leaky_relu(h @ W.to(dt).t() + b.to(dt), 0.01)
        return X[idx][:, 32 * kb_t0: 32 * (kb_t0 + nk_t)].to(dt) - (h @ Wo.to(dt).t() + bo.to(dt))

    r64, r32 = ref(torch.float64), ref(torch.float32)
    out = got[idx][:, 32 * kb_t0: 32 * (kb_t0 + nk_t)].double()
    scale = r64.abs().max().item()
    err = (out - r64).abs().max().item() / scale
    err32 = (r32.double() - r64).abs().max().item() / scale
    assert err < max(4 * err32, 2e-6), (err, err32)
    # conditioning-only blocks untouched, the other set's features of the shared block rewritten unchanged
    assert torch.equal(got[:, 32 * (kb_t0 + nk_t):], X[:, 32 * (kb_t0 + nk_t):])
    assert torch.equal(got[:, 392:416], X[:, 392:416])
    assert int(flag.item()) == 0
