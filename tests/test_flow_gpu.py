"""Parity of the HIP path (Flow API -> FlowEngine -> C ABI -> gfx950 kernels) on a real MI355X
against (a) the golden vectors from the reference, (b) the CPU oracle on fresh seeded inputs, and
(c) size-independent properties at BASELINE.json's full batch size.

Tolerance (north_star): log_prob within 1e-5 relative (fp32) of the reference CPU path."""
import math

import pytest
import torch

from golden_util import case_names, load_case
from model_util import build_flow
from oracle import usflows_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RTOL = 1e-5


def _rel(a, b):
    return ((a.double().cpu() - b.double()).abs() / b.double().abs().clamp_min(1e-30)).max().item()


def _engine_used(flow, before):
    eng = flow.engine()
    assert eng is not None, "flow has no device engine"
    assert eng.launch_count > before, "HIP path did not run"


@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("name", case_names())
def test_golden_parity(name, fused):
    spec, sd, a = load_case(name)
    flow = build_flow(spec, sd, device=DEV)
    eng = flow.engine()
    assert eng is not None
    eng.use_fused_coupling = fused
    eng.fused_min_rows = 0             # small fixtures: force the fused kernel when asked for
    ctx = a.get("context")
    x, zin = a["x"].to(DEV), a["zin"].to(DEV)
    n0 = eng.launch_count
    with torch.no_grad():
        lp = flow.log_prob(x, context=ctx.to(DEV)) if ctx is not None else flow.log_prob(x)
    torch.cuda.synchronize()
    _engine_used(flow, n0)
    # vs the reference's fp64 run and vs the reference's own fp32 run
    assert _rel(lp, a["log_prob64"]) < RTOL, name
    assert _rel(lp, a["log_prob32"]) < RTOL, name
    if ctx is None:
        with torch.no_grad():
            z = flow.backward(x)
            xf = flow._forward(zin)
        s = max(1.0, a["backward64"].abs().max().item())
        assert (z.cpu().double() - a["backward64"]).abs().max().item() < 2e-5 * s
        s = max(1.0, a["forward64"].abs().max().item())
        assert (xf.cpu().double() - a["forward64"]).abs().max().item() < 2e-5 * s


@pytest.mark.parametrize("mode", ["bf16x3", "f32"])
@pytest.mark.parametrize("name", [n for n in case_names() if "d64" in n or "d784" in n or "d33" in n])
def test_golden_parity_gemm_modes(name, mode):
    """affine GEMMs on the bf16 matrix cores with the 3-way split (default) or on exact-f32 MFMA: same 1e-5 gate"""
    spec, sd, a = load_case(name)
    flow = build_flow(spec, sd, device=DEV)
    flow.engine().gemm_mode = mode
    flow.engine().fused_min_rows = 0
    with torch.no_grad():
        lp = flow.log_prob(a["x"].to(DEV))
        xf = flow._forward(a["zin"].to(DEV))
    assert _rel(lp, a["log_prob64"]) < RTOL, name
    assert _rel(lp, a["log_prob32"]) < RTOL, name
    s = max(1.0, a["forward64"].abs().max().item())
    assert (xf.cpu().double() - a["forward64"]).abs().max().item() < 2e-5 * s


@pytest.mark.parametrize("B", [0, 1, 3, 64, 65, 257, 1000])
def test_ragged_batch_sizes_vs_oracle(B):
    spec = orc.FlowSpec(20, 3, [24, 12], householder=1, affine_conjugation=True)
    sd = orc.synth_state_dict(spec, seed=77)
    flow = build_flow(spec, sd, device=DEV)
    x = torch.rand(B, 20, generator=torch.Generator().manual_seed(B))
    with torch.no_grad():
        lp = flow.log_prob(x.to(DEV))
    assert lp.shape == (B,)
    if B:
        ref = orc.flow_log_prob(orc.to_dtype(sd, torch.float64), spec, x.double())
        assert _rel(lp, ref) < RTOL


def test_noncontiguous_and_unaligned_inputs():
    spec = orc.FlowSpec(10, 2, [16], householder=0)      # D % 4 != 0 -> staged through the padded copy
    sd = orc.synth_state_dict(spec, seed=3)
    flow = build_flow(spec, sd, device=DEV)
    big = torch.rand(50, 23, generator=torch.Generator().manual_seed(1))
    x = big[:, 3:13]                                      # non-contiguous, unaligned view
    with torch.no_grad():
        lp = flow.log_prob(x.to(DEV)[:, :])
        lp2 = flow.log_prob(big.to(DEV)[:, 3:13])
    ref = orc.flow_log_prob(orc.to_dtype(sd, torch.float64), spec, x.double())
    assert _rel(lp, ref) < RTOL and _rel(lp2, ref) < RTOL


# (BASELINE cfg2 at full size -- round trip, the UDL constant, rank order and the reference's golden rows at the head, middle
# and tail of 65 536 rows, in both plans -- is tests/test_configs_gpu.py::test_cfg2_model_at_cfg3_rank_shapes[65536-*].)


def test_wide_conditioner_unfused_path_cfg4_like():
    """BASELINE cfg4 shape class (D = 3072, hidden 1024 > the fused kernel's 256): the coupling runs as a
    chain of usf_linear_f32 launches; parity vs the fp64 oracle."""
    spec = orc.FlowSpec(3072, 2, [1024, 1024], householder=0)
    sd = orc.synth_state_dict(spec, seed=4, alpha=0.05)
    flow = build_flow(spec, sd, device=DEV)
    x = torch.rand(96, 3072, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        lp = flow.log_prob(x.to(DEV))
        z = flow.backward(x.to(DEV))
        xr = flow._forward(z)
    ref = orc.flow_log_prob(orc.to_dtype(sd, torch.float64), spec, x.double())
    assert _rel(lp, ref) < RTOL
    assert (xr.cpu() - x).abs().max().item() < 2e-4


@pytest.mark.parametrize("extra,hidden,mode", [({"gating": True, "normalize_layers": True}, [256, 192], "bf16x3"),
                                               ({"gating": True, "normalize_layers": True}, [256, 192], "f32"),
                                               ({"gating": True}, [128, 128, 128], "bf16x3"),
                                               ({"normalize_layers": True}, [300, 256], "bf16x3")])
def test_vector_convnet_gated_layernorm_conditioner_on_the_engine(extra, hidden, mode):
    """the reference's default vector ConvNet (GatedMLP blocks + LayerNormVector, networks.py:206-245, 287-308) at the
    headline width: the engine runs it as usf_linear_f32 launches plus one usf_gated_norm_rows_f32 pass per block (no torch
    composite, no warning); rows from the head, middle and tail of an 8192-row batch against the fp64 oracle, round trip"""
    import warnings
    spec = orc.FlowSpec(784, 3, hidden, householder=0, conditioner="ConvNet", extra=extra)
    sd = orc.synth_state_dict(spec, seed=31)
    flow = build_flow(spec, sd, device=DEV)
    B = 8192
    x = torch.rand(B, 784, generator=torch.Generator().manual_seed(5))
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        eng = flow.engine()
        assert eng is not None and eng._general_cond
        eng.gemm_mode = mode
        with torch.no_grad():
            lp = flow.log_prob(x.to(DEV))
            z = flow.backward(x.to(DEV))
            xr = flow._forward(z)
    from usflows_amd import _ext
    plan = next(iter(eng._plans.values()))
    kinds = [plan["arr"][j].kind for j in range(plan["n"])]
    assert kinds.count(_ext.OP_GATED_NORM) == 3 * (len(hidden) + 1) and _ext.OP_COUPLING not in kinds
    rows = torch.cat([torch.arange(0, 32), torch.arange(B // 2 - 16, B // 2 + 16), torch.arange(B - 32, B)])
    ref = orc.flow_log_prob(orc.to_dtype(sd, torch.float64), spec, x[rows].double())
    assert _rel(lp[rows.to(DEV)], ref) < RTOL
    assert (xr.cpu() - x).abs().max().item() < 2e-4
    # the composite torch formulation of the same modules (what a gradient-carrying call runs) agrees
    lp_t = flow._layer_loop_log_prob(x[rows].to(DEV))
    assert _rel(lp_t.detach(), ref) < RTOL


@pytest.mark.parametrize("B", [0, 1, 3, 257])
def test_vector_convnet_gated_conditioner_ragged_batches_and_sampling(B):
    """the op chain of the gated / layer-normalised vector ConvNet at empty, single-row and ragged batches (partial row
    blocks of usf_gated_norm_rows_f32, the skinny linear kernel), a non-contiguous input view, and Flow.sample through it"""
    spec = orc.FlowSpec(20, 3, [24, 16], householder=1, conditioner="ConvNet", extra={"gating": True, "normalize_layers": True})
    sd = orc.synth_state_dict(spec, seed=5)
    flow = build_flow(spec, sd, device=DEV)
    big = torch.rand(B, 45, generator=torch.Generator().manual_seed(B))
    x = big[:, 5:25]
    with torch.no_grad():
        lp = flow.log_prob(big.to(DEV)[:, 5:25])
        xs = flow.sample([7], seed=1)
    assert lp.shape == (B,) and xs.shape == (7, 20) and torch.isfinite(xs).all()
    if B:
        ref = orc.flow_log_prob(orc.to_dtype(sd, torch.float64), spec, x.double())
        assert _rel(lp, ref) < RTOL


def test_three_hidden_layers_and_narrow_widths_fused():
    spec = orc.FlowSpec(40, 3, [48, 20, 136], householder=1, affine_conjugation=True, negative_slope=0.0)
    sd = orc.synth_state_dict(spec, seed=8)
    flow = build_flow(spec, sd, device=DEV)
    flow.engine().fused_min_rows = 0
    x = torch.rand(333, 40, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        lp = flow.log_prob(x.to(DEV))
    ref = orc.flow_log_prob(orc.to_dtype(sd, torch.float64), spec, x.double())
    assert _rel(lp, ref) < RTOL


@pytest.mark.parametrize("hidden,soft", [([256], False), ([256, 256, 256], False), ([200, 256], True), ([256, 160], False)])
@pytest.mark.parametrize("mode", ["bf16x3", "f32"])
def test_wide_hidden_variants_both_gemm_modes(hidden, soft, mode):
    """hidden widths in (128, 256] with M >= 1024 take the bf16x3 fused coupling kernel in bf16x3 mode:
    1 / 2 / 3 hidden layers, ragged widths, soft-training context"""
    spec = orc.FlowSpec(72, 2, hidden, householder=0, soft_training=soft)
    sd = orc.synth_state_dict(spec, seed=21)
    flow = build_flow(spec, sd, device=DEV)
    flow.engine().gemm_mode = mode
    flow.engine().fused_min_rows = 0
    g = torch.Generator().manual_seed(5)
    x = torch.rand(1500, 72, generator=g)
    ctx = torch.rand(1500, 1, generator=g) if soft else None
    with torch.no_grad():
        lp = flow.log_prob(x.to(DEV), context=ctx.to(DEV)) if soft else flow.log_prob(x.to(DEV))
    ref = orc.flow_log_prob(orc.to_dtype(sd, torch.float64), spec, x.double(), ctx.double() if soft else None)
    assert _rel(lp, ref) < RTOL, (hidden, soft, mode)


def test_linearity_of_affine_layer():
    """BlockAffineTransform.backward is affine: f(a x1 + (1-a) x2) = a f(x1) + (1-a) f(x2)."""
    spec = orc.FlowSpec(64, 1, [8], householder=1)
    sd = orc.synth_state_dict(spec, seed=9)
    flow = build_flow(spec, sd, device=DEV)
    layer = flow.layers[0]
    g = torch.Generator().manual_seed(0)
    x1, x2 = torch.randn(33, 64, generator=g).to(DEV), torch.randn(33, 64, generator=g).to(DEV)
    with torch.no_grad():
        f = layer.backward
        lhs = f(0.25 * x1 + 0.75 * x2)
        rhs = 0.25 * f(x1) + 0.75 * f(x2)
    assert (lhs - rhs).abs().max().item() < 1e-4


def test_single_layer_dispatch_on_device():
    spec, sd, a = load_case("synth_d16_k4_hh2_conj_laplace")
    flow_gpu = build_flow(spec, sd, device=DEV)
    flow_cpu = build_flow(spec, sd)
    x = a["x"]
    with torch.no_grad():
        for lg, lc in zip(flow_gpu.layers, flow_cpu.layers):
            for fn in ("forward", "backward"):
                got = getattr(lg, fn)(x.to(DEV)).cpu()
                ref = getattr(lc, fn)(x)
                assert torch.allclose(got, ref, rtol=2e-5, atol=2e-5), (type(lg).__name__, fn)


def test_sampling_device_path():
    spec, sd, _ = load_case("synth_d16_k4_hh2_conj_laplace")
    flow = build_flow(spec, sd, device=DEV)
    with torch.no_grad():
        s1 = flow.sample([1000], seed=42)
        s2 = flow.sample([1000], seed=42)
        s3 = flow.sample([10, 100], seed=42)
        # rank-sharded draw reproduces the single-process draw
        from usflows_amd.parallel import sample_sharded
        parts = [sample_sharded(flow, 1000, 42, r, 4) for r in range(4)]
    assert s1.shape == (1000, 16) and s3.shape == (10, 100, 16)
    assert torch.equal(s1, s2) and torch.equal(s1, s3.reshape(1000, 16))
    # same Philox noise row for row; the forward pass of a 250-row shard takes the small-batch kernel, whose
    # summation order differs from the tiled kernel's: equal to fp32 rounding, not bitwise
    assert torch.allclose(torch.cat(parts), s1, rtol=2e-5, atol=2e-5)
    assert torch.isfinite(s1).all()
    # samples pushed back through the flow are Laplace(loc, scale) noise: z = f^-1(x)
    with torch.no_grad():
        z = flow.backward(s1).cpu().double()
    loc, sc = spec.base_loc.double(), spec.base_scale.double()
    u = (z - loc) / sc
    assert abs(u.mean().item()) < 0.1 and abs(u.abs().mean().item() - 1.0) < 0.1


def test_autograd_path_still_differentiable_on_device():
    """Flow.fit's use of the path (loss.backward) goes through the composite formulation."""
    spec, sd, a = load_case("synth_d7_k3_hh0_laplace")
    flow = build_flow(spec, sd, device=DEV)
    x = a["x"].to(DEV)
    loss = -flow.log_prob(x).mean()
    loss.backward()
    named = dict(flow.named_parameters())
    # (the context branch layers.1.* of ConditionalDenseNN is unused without a context -> no grad)
    missing = [k for k, p in named.items() if p.requires_grad and p.grad is None and ".layers.1." not in k]
    assert not missing, missing
    with torch.no_grad():
        lp = flow.log_prob(x)
    assert abs(lp.mean().item() + loss.item()) < 1e-4 * abs(loss.item())


def test_missing_library_fails_loudly(monkeypatch):
    from usflows_amd import _ext
    monkeypatch.setattr(_ext, "_lib", None)
    monkeypatch.setattr(_ext, "LIB_PATH", "/nonexistent/libusflows_hip.so")
    spec, sd, a = load_case("synth_d7_k3_hh0_laplace")
    flow = build_flow(spec, sd, device=DEV)
    with pytest.raises(RuntimeError, match="HIP extension not built"):
        with torch.no_grad():
            flow.log_prob(a["x"].to(DEV))


def test_small_batches_replay_a_hip_graph():
    """B <= 1024: call 1 launches plainly, call 2 captures the launch list into a hipGraph, later calls replay it
    (engine.py:_execute_graph); results must not depend on which of the three a call was, must follow new inputs,
    and must follow an in-place parameter update (the pack is refreshed at the same addresses)."""
    spec, sd, a = load_case("synth_d64_k6_hh0_laplace")
    flow = build_flow(spec, sd, device=DEV)
    eng = flow.engine()
    eng.use_graphs = True                  # opt-in (USFLOWS_AMD_GRAPH=1): no faster than the launch loop on this stack
    g = torch.Generator().manual_seed(7)
    xs = [torch.rand(100, 64, generator=g) for _ in range(4)]
    with torch.no_grad():
        for i, x in enumerate(xs):
            lp = flow.log_prob(x.to(DEV)).cpu().double()
            ref = orc.flow_log_prob(orc.to_dtype(sd, torch.float64), spec, x.double())
            assert ((lp - ref).abs() / ref.abs()).max().item() < 1e-5, i
            z = flow.backward(x.to(DEV))
            assert torch.allclose(flow._forward(z).cpu(), x, rtol=1e-4, atol=1e-4)
        plans = [p for p in eng._plans.values() if p.get("graph") is not None]
        assert plans, "no launch list was captured"
        # optimiser-style in-place update of every parameter
        for p in flow.parameters():
            p.mul_(1.0 + 1e-3)
        sd2 = {k: v.detach().cpu().clone() for k, v in flow.state_dict().items()}
        lp = flow.log_prob(xs[0].to(DEV)).cpu().double()
        ref = orc.flow_log_prob(orc.to_dtype(sd2, torch.float64), spec, xs[0].double())
        assert ((lp - ref).abs() / ref.abs()).max().item() < 1e-5
        # the sampling direction reads the refreshed M / bias / scale images too
        zin = torch.randn(100, 64, generator=g)
        xf = flow._forward(zin.to(DEV)).cpu().double()
        xref = orc.flow_forward(orc.to_dtype(sd2, torch.float64), spec, zin.double())
        assert (xf - xref).abs().max().item() < 2e-5 * max(1.0, xref.abs().max().item())


@pytest.mark.parametrize("hh,conj", [(0, True), (1, True)])
def test_cfg2_scale_variants_of_the_live_configs(hh, conj):
    """the variants every live config of the reference uses (affine_conjugation=True, with and without a
    Householder factor) at the cfg2 width (D = 784; 8 blocks = 17 affine applications, which keeps the fp64 oracle's
    per-call matrix inversions at ~10 s): log_prob of 64 rows vs the fp64 oracle, round trip and the constant-Jacobian
    (UDL) property on 4096 rows."""
    from usflows_amd.synth import ModelSpec, synth_state_dict
    spec = ModelSpec(784, 8, [256, 256], householder=hh, affine_conjugation=conj, negative_slope=0.01,
                     conditioner="ConditionalDenseNN", base="laplace")
    sd = synth_state_dict(spec, seed=100, alpha=0.1)
    flow = build_flow(spec, sd, device=DEV)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(4096, 784, generator=g)
    with torch.no_grad():
        lp = flow.log_prob(x.to(DEV))
        z = flow.backward(x.to(DEV))
        xr = flow._forward(z)
    ospec = orc.FlowSpec(784, 8, [256, 256], householder=hh, affine_conjugation=conj)
    ref = orc.flow_log_prob(orc.to_dtype(sd, torch.float64), ospec, x[:64].double())
    assert ((lp[:64].cpu().double() - ref).abs() / ref.abs()).max().item() < 1e-5
    assert (xr.cpu() - x).abs().max().item() < 5e-4
    # log_prob(x) - base.log_prob(z) is one constant for all samples
    base = torch.distributions.Laplace(torch.zeros(784), torch.ones(784)).log_prob(z.cpu()).sum(-1)
    c = (lp.cpu() - base).double()
    assert (c - c.mean()).abs().max().item() < 1e-5 * abs(c.mean().item()) + 2e-2


def test_udl_profile_on_device_matches_reference():
    """calibrated_latent_radial_udl_profile (flows.py:294-378) with the latents from the device backward pass against
    the profiles the REAL reference computed (tests/golden/udl_*.npz)"""
    from golden_util import load_udl, udl_case_names
    from test_modules_cpu import _check_udl
    for name in udl_case_names():
        spec, sd, a = load_case(name)
        flow = build_flow(spec, sd, device=DEV)
        before = flow.engine().launch_count
        _check_udl(flow, a["x"].to(DEV), load_udl(name))
        assert flow.engine().launch_count > before


def test_composite_fallback_on_device_warns_once():
    """a layer list without a fused device form (a conditioner whose nonlinearity the kernels do not have: Tanh)
    still works on the device through the torch composite loop -- but says so"""
    import warnings
    from usflows_amd.flows import USFlow
    from usflows_amd.networks import ConvNet
    flow = USFlow(torch.distributions.Laplace(torch.zeros(12, device=DEV), torch.ones(12, device=DEV)), [12], 2, ConvNet,
                  dict(in_dims=[12], c_hidden=[16], nonlinearity=torch.nn.Tanh()), householder=0).to(DEV)
    x = torch.rand(8, 12, device=DEV)
    with torch.no_grad():
        with pytest.warns(RuntimeWarning, match="composite formulation"):
            lp = flow.log_prob(x)
        with warnings.catch_warnings():
            warnings.simplefilter("error")              # second call: no further warning
            lp2 = flow.log_prob(x)
    assert torch.isfinite(lp).all() and torch.equal(lp, lp2)


def test_radial_base_with_gammamm_norm_on_device():
    """the base distribution of the reference's live configs (gaussian_mixture.yaml:74-93: RadialDistribution, p = 1, over
    a GammaMM norm distribution): log_prob against the golden of the real reference comes through test_golden_parity;
    here Flow.sample on the device -- the latent radii of the drawn samples follow the Gamma mixture"""
    spec, sd, a = load_case("synth_d16_k3_hh0_conj_radial1_gammamm")
    flow = build_flow(spec, sd, device=DEV)
    before = flow.engine().launch_count
    with torch.no_grad():
        xs = flow.sample([4000], seed=5)
        assert flow.engine().launch_count > before and xs.shape == (4000, 16) and torch.isfinite(xs).all()
        z = flow.backward(xs)
        lp = flow.log_prob(xs)
    r = (z - flow.base_distribution.loc).abs().sum(-1)
    ref = flow.base_distribution.norm_distribution.sample((40000,)).reshape(-1)
    assert abs(r.mean().item() - ref.mean().item()) < 0.05 * ref.mean().item()
    assert abs(r.std().item() - ref.std().item()) < 0.08 * ref.std().item()
    # UDL: log_prob(x) - base.log_prob(z) is one constant
    c = lp.double() - flow.base_distribution.log_prob(z).double()
    assert (c - c.mean()).abs().max().item() < 1e-4 * max(1.0, abs(c.mean().item()))


@pytest.mark.parametrize("planes", [False, True])
@pytest.mark.parametrize("name", [n for n in case_names() if "conj" in n and "cfg4" not in n])
def test_merged_affine_runs_on_device(name, planes):
    """opt-in FlowEngine.merge_affine (consecutive affine maps of a conjugated flow composed at pack time): golden parity
    on the device in both activation formats -- 1e-5 on the conditioned cases, 3e-5 on the default-initialised ones (the
    reason the switch is opt-in) -- and fewer launches than the one-by-one plan"""
    spec, sd, a = load_case(name)
    if a.get("context") is not None or spec.soft_training:
        pytest.skip("context flows: training / soft-training plans are never merged")
    counts = {}
    for merge in (False, True):
        flow = build_flow(spec, sd, device=DEV)
        eng = flow.engine()
        eng.merge_affine = merge
        eng.fused_min_rows = 0
        if planes:
            eng.use_planes, eng.planes_min_rows = True, 0
        x, zin = a["x"].to(DEV), a["zin"].to(DEV)
        with torch.no_grad():
            lp = flow.log_prob(x)
            z = flow.backward(x)
            xf = flow._forward(zin)
        tol = 3e-5 if name.startswith("init_") else RTOL
        assert _rel(lp, a["log_prob64"]) < tol, (name, merge)
        s = max(1.0, a["backward64"].abs().max().item())
        assert (z.cpu().double() - a["backward64"]).abs().max().item() < 2 * tol * s
        s = max(1.0, a["forward64"].abs().max().item())
        assert (xf.cpu().double() - a["forward64"]).abs().max().item() < 2 * tol * s
        counts[merge] = min(p["n"] for p in eng._plans.values())
        # an in-place parameter update (optimiser step): the pack refresh recomposes the merged maps in place
        with torch.no_grad():
            for p in flow.parameters():
                if p.dim() >= 1 and p.numel() > 1:
                    p.mul_(1.0 + 1e-3)
            counts[("lp2", merge)] = flow.log_prob(x).cpu()
    assert counts[True] < counts[False], counts
    a2, b2 = counts[("lp2", False)], counts[("lp2", True)]
    assert not torch.equal(a2, lp.cpu())                                    # the update changed the result ...
    assert ((a2 - b2).abs() / a2.abs()).max().item() < (3e-5 if name.startswith("init_") else 2e-6)   # ... in both plans alike
