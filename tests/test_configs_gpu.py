"""BASELINE.json's configurations at FULL size on a real MI355X ("configs_untested").

cfg2  D=784, K=32, h=[256,256], B=65536            -> also at B = 32768 (the per-rank shape of cfg3) and 16384
cfg3  same model, 262144 rows over 8 GPUs          -> 32768 rows per rank (different tile instantiation: wave quantisation)
cfg4  D=3072, K=48, h=[1024,1024], B=32768
cfg5  sample() of 10^6 draws over 8 GPUs           -> 125000 draws per rank with that rank's Philox substream

Each case compares rows at the HEAD, in the MIDDLE and at the TAIL of the batch (first / last panels, other XCD slots)
with outputs of the real reference (committed golden vectors, fp32 and fp64 runs) or with the fp64 oracle, and checks
the size-independent properties on the whole batch: round trip, the constant-Jacobian (UDL) property
log_prob(x) - base.log_prob(f^-1(x)) == -sum ladj for every row, and rank order.  Every test also records which
usf_linear_f32 instantiations its launch plans selected; test_variants_cpu.py proves on the host that the BASELINE
shapes select only instantiations the kernel parity tests cover."""
import ctypes as C

import pytest
import torch

from golden_util import load_case
from model_util import build_flow
from oracle import usflows_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RTOL = 1e-5


def _rel(a, b):
    return ((a.double().cpu() - b.double()).abs() / b.double().abs().clamp_min(1e-30)).max().item()


def plan_variants(eng):
    """set of usf_linear_variant codes of every linear op in the engine's built plans"""
    from usflows_amd import _ext
    lib = _ext.load()
    out = set()
    for plan in eng._plans.values():
        arr = plan["arr"]
        for j in range(plan["n"]):
            if arr[j].kind == _ext.OP_LINEAR:
                out.add(lib.usf_linear_variant(C.byref(arr[j].u.linear)))
            elif arr[j].kind == _ext.OP_GEMM_PLANES:
                out.add(lib.usf_gemm_planes_variant(C.byref(arr[j].u.gemm_planes)))
            elif arr[j].kind == _ext.OP_COUPLING_PLANES:       # fused coupling on planes: 6000 + planes per operand
                out.add(6002 if arr[j].u.coupling_planes.format == _ext.PLANES_F16X2 else 6003)
    return out


def place_probes(x, probes):
    """write the probe rows into the head, the middle (straddling B/2) and the tail of x; returns their row indices"""
    B, n = x.shape[0], probes.shape[0]
    a, b = n // 3, 2 * (n // 3)
    idx = torch.cat([torch.arange(0, a), torch.arange(B // 2 - (b - a) // 2, B // 2 - (b - a) // 2 + (b - a)),
                     torch.arange(B - (n - b), B)])
    x[idx] = probes
    return idx


@pytest.fixture(scope="module")
def cfg2():
    spec, sd, a = load_case("synth_d784_k32_cfg2")        # BASELINE cfg2 model; 64 rows + outputs of the real reference
    flow = build_flow(spec, sd, device=DEV)
    # (the log-dets read the U factors and the scale only: no fp64 copy of the other ~10^9 parameters)
    ladj = float(orc.total_ladj({k: v.double() for k, v in sd.items() if k.endswith("U_raw") or k.endswith("scale")}, spec))
    return spec, sd, a, flow, ladj


def _properties(flow, x, lp, ladj, base=None, rt_tol=2e-4):
    with torch.no_grad():
        z = flow.backward(x)
        xr = flow._forward(z)
    assert torch.isfinite(lp).all() and torch.isfinite(z).all()
    assert (xr - x).abs().max().item() < rt_tol                                   # round trip
    base = base or torch.distributions.Laplace(0.0, 1.0)
    base_lp = base.log_prob(z.double()).sum(-1)
    const = lp.double() - base_lp
    assert (const + ladj).abs().max().item() < 1e-5 * base_lp.abs().max().item()  # ONE constant: -sum ladj (UDL)
    for lo in (0, x.shape[0] // 2 - 2048, x.shape[0] - 4096):                     # rank order, head / middle / tail
        sl = slice(lo, lo + 4096)
        order = torch.argsort(lp[sl])
        assert (base_lp[sl][order].diff() >= -1e-2).all()
    return z


@pytest.mark.parametrize("planes", [True, False])
@pytest.mark.parametrize("B", [16384, 32768, 65536])
def test_cfg2_model_at_cfg3_rank_shapes(cfg2, B, planes):
    """B = 32768 is cfg3's per-rank batch (262144 / 8).  planes=True: the default plan at these sizes (activations as
    bf16 planes between layers, usf_gemm_planes_bf16x3); planes=False: the fp32-activation plan (what training, context
    and small batches use) -- there 16384 / 32768 select the 128-row bf16x3 tile with the 2-buffer ring, 65536 the
    256-row tile with the 4-buffer ring."""
    spec, sd, a, flow, ladj = cfg2
    flow.engine().use_planes = planes            # (None = automatic: planes only in "f16x2" mode)
    flow.engine()._plans.clear()
    x = torch.rand(B, 784, generator=torch.Generator().manual_seed(1234 + B))
    idx = place_probes(x, a["x"])
    xd = x.to(DEV)
    eng = flow.engine()
    n0 = eng.launch_count
    with torch.no_grad():
        lp = flow.log_prob(xd)
    torch.cuda.synchronize()
    assert eng.launch_count > n0, "HIP path did not run"
    got = lp[idx.to(DEV)]
    assert _rel(got, a["log_prob64"]) < RTOL            # the reference's fp64 run
    assert _rel(got, a["log_prob32"]) < RTOL            # the reference's own fp32 run
    z = _properties(flow, xd, lp, ladj)
    s = max(1.0, a["backward64"].abs().max().item())
    assert (z[idx.to(DEV)].cpu().double() - a["backward64"]).abs().max().item() < 2e-5 * s
    v = plan_variants(eng)
    if planes:
        assert {5050, 6003, 5051} <= v, (B, v)      # affine GEMMs on planes, fused couplings on planes, fp32 output of the last layer
    else:
        assert {16384: 3542, 32768: 3542, 65536: 3584}[B] in v, (B, v)
    flow.engine().use_planes = None


_CFG5_REF = {}


@pytest.mark.parametrize("mode", ["bf16x3", "f16x2"])
def test_cfg5_per_rank_sample(cfg2, mode):
    """cfg5 per rank: rank 3's 125000 of the 10^6 draws.  (1) sample() == _forward of the head kernel's own noise;
    (2) head / middle / tail rows of x vs the fp64 oracle's Flow._forward of that same noise; (3) the UDL check of
    BASELINE cfg5: log_prob(x) - base.log_prob(backward(x)) is one constant; (4) the draw is the matching slice of
    the single-process draw (Philox substreams)."""
    from usflows_amd import _ext
    spec, sd, a, flow, ladj = cfg2
    n, rank, seed = 125000, 3, 2026
    eng = flow.engine()
    eng.gemm_mode, eng.use_planes = mode, None      # "f16x2": the planes pipeline with fp16x2 planes (opt-in fast mode)
    eng._plans.clear()
    with torch.no_grad():
        xs = flow.sample([n], seed=seed, row_offset=rank * n)
        z = torch.empty(n, 784, device=DEV)
        zero, one = torch.zeros(784, device=DEV), torch.ones(784, device=DEV)
        _ext.base_sample(z, 784, n, 784, _ext.BASE_LAPLACE, zero, one, seed, 0, rank * n)
        xf = flow._forward(z)
        lp = flow.log_prob(xs)
        zb = flow.backward(xs)
    assert xs.shape == (n, 784) and torch.isfinite(xs).all() and torch.isfinite(z).all()
    assert torch.equal(xs, xf)
    idx = torch.cat([torch.arange(0, 24), torch.arange(n // 2 - 12, n // 2 + 12), torch.arange(n - 24, n)])
    zi = z[idx.to(DEV)].cpu().double()
    if _CFG5_REF.get("z") is None or not torch.equal(_CFG5_REF["z"], zi):       # (the noise does not depend on the GEMM mode)
        _CFG5_REF.update(z=zi, x=orc.flow_forward(orc.to_dtype(sd, torch.float64), spec, zi))
    ref = _CFG5_REF["x"]
    s = max(1.0, ref.abs().max().item())
    assert (xs[idx.to(DEV)].cpu().double() - ref).abs().max().item() < 2e-5 * s
    base_lp = torch.distributions.Laplace(0.0, 1.0).log_prob(zb.double()).sum(-1)
    const = lp.double() - base_lp
    assert (const + ladj).abs().max().item() < 1e-5 * base_lp.abs().max().item()
    # f^-1(f(z)) == z through 2 x 65 layers (fp16x2 planes carry 22 instead of 24 significant bits per operand: 4x the slack)
    assert (zb - z).abs().max().item() < (2e-3 if mode == "f16x2" else 5e-4)
    # noise statistics of the rank's substream: Laplace(0,1) -> E|z| = 1, no +-inf (u01 reached 1.0)
    assert abs(z.abs().mean().item() - 1.0) < 5e-3 and z.abs().max().item() < 40.0
    # the rank's rows are rows [rank*n, (rank+1)*n) of a single-process draw
    with torch.no_grad():
        part = flow.sample([1000], seed=seed, row_offset=rank * n + 5000)
    # (same Philox noise row for row; a 1000-row pass takes other tiles / the unfused conditioner: equal to fp32 rounding
    # through the 65 layers, not bitwise)
    dev_ = (part - xs[5000:6000]).abs().max().item()
    assert dev_ < 2e-5 * max(1.0, xs.abs().max().item()), dev_
    v = plan_variants(eng)
    # 125000 rows: the planes pipeline in both modes (automatic from 49152 rows in bf16x3), fused couplings on planes
    assert {5050, 5051, 6002 if mode == "f16x2" else 6003} <= v, v
    assert eng.f16_fallbacks == 0
    eng.gemm_mode = "bf16x3"


@pytest.fixture(scope="module")
def cfg4():
    spec, sd, a = load_case("synth_d3072_k48_cfg4")       # 16 rows + outputs of the real reference (fp32 and fp64 runs)
    flow = build_flow(spec, sd, device=DEV)
    ladj = float(a["total_ladj64"])
    del sd
    return spec, a, flow, ladj


@pytest.mark.parametrize("mode", ["bf16x3", "f16x2"])
def test_cfg4_full_size(cfg4, mode):
    """BASELINE cfg4: D = 3072, 48 blocks, hidden [1024, 1024] (1.28 G parameters), B = 32768.  The 3072 x 3072 affine
    layers and the 1536 <-> 1024 conditioner layers run on the bf16x3 tiles (hidden width > 256: chain of linears)."""
    spec, a, flow, ladj = cfg4
    B = 32768
    x = torch.rand(B, 3072, generator=torch.Generator().manual_seed(77))
    idx = place_probes(x, a["x"])
    xd = x.to(DEV)
    eng = flow.engine()
    eng.gemm_mode, eng.use_planes = mode, None
    eng._plans.clear()
    n0 = eng.launch_count
    with torch.no_grad():
        lp = flow.log_prob(xd)
    torch.cuda.synchronize()
    assert eng.launch_count > n0
    got = lp[idx.to(DEV)]
    assert _rel(got, a["log_prob64"]) < RTOL
    assert _rel(got, a["log_prob32"]) < 2 * RTOL        # (the reference's own fp32-vs-fp64 gap at this depth: see fixture log)
    # (round trip through 97 fp32 layers of 3072 columns: 8.8e-4 in the runs this was written on, 1.1e-3 once on another
    # box of the pool -- a property of the fp32 chain, not a parity bound: parity is the golden rows above and below)
    z = _properties(flow, xd, lp, ladj, rt_tol=4e-3 if mode == "f16x2" else 2e-3)
    s = max(1.0, a["backward64"].abs().max().item())
    assert (z[idx.to(DEV)].cpu().double() - a["backward64"]).abs().max().item() < 2e-5 * s
    with torch.no_grad():
        xf = flow._forward(a["zin"].to(DEV))
    s = max(1.0, a["forward64"].abs().max().item())
    assert (xf.cpu().double() - a["forward64"]).abs().max().item() < 2e-5 * s
    v = plan_variants(eng)                                      # (the 16-row _forward above takes the small-batch kernel)
    assert {5040, 5041} <= v, v            # hidden 1024: no fused coupling kernel -> the planes pipeline in both modes
    assert eng.f16_fallbacks == 0
    eng.gemm_mode = "bf16x3"


def test_laplace_head_extreme_words_are_finite():
    """the random word 0xFFFFFFFF used to map to u = 1.0 (fp32 round-to-even) -> log1p(-1) = -inf.
    The word -> variate maps of the head kernels on the extreme and neighbouring words."""
    from usflows_amd import _ext
    import numpy as np
    words = torch.tensor([0, 1, 0x1FF, 0x200, 0x7FFFFFFF, 0x80000000, 0xFFFFFE00, 0xFFFFFFFE, 0xFFFFFFFF],
                         dtype=torch.int64)
    bits = torch.from_numpy(words.numpy().astype(np.uint32).view(np.int32)).to(DEV)      # same 32 bits
    n = words.numel()
    u, lap, ex = (torch.empty(n, device=DEV) for _ in range(3))
    _ext.variates_from_bits(bits, u, lap, ex)
    torch.cuda.synchronize()
    u, lap, ex = u.cpu(), lap.cpu(), ex.cpu()
    assert (u > 0).all() and (u < 1).all()
    assert torch.isfinite(lap).all() and torch.isfinite(ex).all()
    expect = ((words >> 9).double() + 0.5) / 2 ** 23
    assert torch.equal(u.double(), expect)                       # exact in fp32
    # symmetric: word w and its complement give -v
    assert abs(lap[0].item() + lap[-1].item()) < 1e-6 * abs(lap[0].item())
    assert abs(lap[0].item()) == pytest.approx(-torch.log1p(torch.tensor(-(1 - 2.0 ** -23))).item(), rel=1e-6)
