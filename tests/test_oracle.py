"""Pin the CPU oracle (oracle/usflows_oracle.py) against
 (a) golden vectors generated from the real reference (tests/golden/*.npz), fp32 and fp64,
 (b) the reference's own known-answer tests (tests/veriflow/transforms_test.py:5-19, 35-51),
 (c) -- only where /root/reference exists -- a live re-run of the reference."""
import math
import os

import pytest
import torch

from oracle import usflows_oracle as orc
from golden_util import case_names, load_case

SMALL = case_names(small_only=True)
ALL = case_names()


def _rel(a, b):
    return ((a.double() - b.double()).abs() / b.double().abs().clamp_min(1e-30)).max().item()


@pytest.mark.parametrize("name", SMALL)
def test_oracle_fp64_matches_reference_fp64(name):
    spec, sd, a = load_case(name)
    sd64 = orc.to_dtype(sd, torch.float64)
    ctx = a.get("context")
    ctx64 = ctx.double() if ctx is not None else None
    lp = orc.flow_log_prob(sd64, spec, a["x"].double(), ctx64)
    assert _rel(lp, a["log_prob64"]) < 1e-11
    z = orc.flow_backward(sd64, spec, a["x"].double(), ctx64)
    assert torch.allclose(z, a["backward64"], rtol=1e-10, atol=1e-10)
    xf = orc.flow_forward(sd64, spec, a["zin"].double(), ctx64)
    assert torch.allclose(xf, a["forward64"], rtol=1e-10, atol=1e-10)
    assert abs(float(orc.total_ladj(sd64, spec)) - float(a["total_ladj64"])) < 1e-9 * max(1.0, abs(float(a["total_ladj64"])))


@pytest.mark.parametrize("name", SMALL)
def test_oracle_fp32_matches_reference_fp32(name):
    """Same ops in the same order -> the fp32 oracle reproduces the fp32 reference to rounding
    noise (bitwise on most cases; threaded BLAS reductions may differ in the last ulp)."""
    spec, sd, a = load_case(name)
    ctx = a.get("context")
    lp = orc.flow_log_prob(sd, spec, a["x"], ctx)
    assert _rel(lp, a["log_prob32"]) < 2e-6
    # and both sit within the reference's own fp32 noise of the fp64 truth
    assert _rel(lp, a["log_prob64"]) < 2e-5
    z = orc.flow_backward(sd, spec, a["x"], ctx)
    scale = a["backward64"].abs().max().item()
    assert (z.double() - a["backward64"]).abs().max().item() < 2e-5 * scale
    xf = orc.flow_forward(sd, spec, a["zin"], ctx)
    scale = a["forward64"].abs().max().item()
    assert (xf.double() - a["forward64"]).abs().max().item() < 2e-5 * scale


@pytest.mark.slow
@pytest.mark.parametrize("name", [n for n in ALL if "d784" in n])
def test_oracle_cfg2_model(name):
    """BASELINE cfg2 model (D=784, K=32, h=[256,256]): state dict regenerated from the seed."""
    spec, sd, a = load_case(name)
    lp = orc.flow_log_prob(sd, spec, a["x"])
    assert _rel(lp, a["log_prob32"]) < 5e-6
    assert _rel(lp, a["log_prob64"]) < 1e-5


def test_udl_property_constant_jacobian():
    """log_prob(x) - base.log_prob(f^-1(x)) is ONE constant (= -sum ladj): the uniformly-scaling
    property the reference exists for (README.md:7-12)."""
    spec, sd, a = load_case("synth_d16_k4_hh2_conj_laplace")
    sd64 = orc.to_dtype(sd, torch.float64)
    x = a["x"].double()
    z, ld = orc.flow_backward(sd64, spec, x, return_logdet=True)
    assert (ld - ld[0]).abs().max() == 0
    assert abs(ld[0].item() + float(orc.total_ladj(sd64, spec))) < 1e-10


# ---- the reference's own known-answer tests, restated on the oracle's primitives -----------
def test_kat_scale_transform():
    """tests/veriflow/transforms_test.py:5-19: scale==2 -> y=2x, exact inverse, ladj = 10 log 2."""
    dim = 10
    s = torch.ones(dim) * 2
    x = torch.ones(dim)
    y = x * s
    assert (y == 2 * x).all()
    assert (y / s == x).all()
    assert s.abs().log().sum() == dim * torch.log(torch.tensor(2.0))


def test_kat_lu_transform():
    """tests/veriflow/transforms_test.py:35-51: L=tril(ones), U=I, b=0 -> y = arange+1, exact
    inverse, ladj 0."""
    dim = 10
    L_raw, U_raw, b = torch.tril(torch.ones(dim, dim)), torch.eye(dim), torch.zeros(dim)
    x = torch.ones(dim)
    y = torch.nn.functional.linear(x, orc.lu_matrix(L_raw, U_raw), b)
    assert (y == torch.arange(dim) + 1.0).all()
    xb = torch.nn.functional.linear(y - b, orc.lu_inverse_matrix(L_raw, U_raw))
    assert (xb == x).all()
    assert orc.lu_ladj(U_raw) == 0


def test_checkerboard_mask():
    m = orc.checkerboard_mask(7)
    assert m.shape == (1, 7) and m.flatten().tolist() == [0, 1, 0, 1, 0, 1, 0]


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/usflows"), reason="reference not present")
def test_oracle_vs_live_reference():
    """Build container only: rerun the real reference now and compare with the oracle."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import ref_shim
    flows, transforms, networks, distributions = ref_shim.install()
    torch.manual_seed(7)
    D = 12
    flow = flows.USFlow(torch.distributions.Laplace(torch.zeros(D), torch.ones(D)), [D], 3,
                        networks.ConditionalDenseNN,
                        dict(input_dim=D, context_dim=1, hidden_dims=[20, 20], out_dim=D,
                             nonlinearity=torch.nn.LeakyReLU(0.01)),
                        householder=1, affine_conjugation=True)
    spec = orc.FlowSpec(D, 3, [20, 20], householder=1, affine_conjugation=True)
    sd = {k: v.detach() for k, v in flow.state_dict().items()}
    x = torch.rand(9, D)
    with torch.no_grad():
        ref = flow.log_prob(x)
    assert _rel(orc.flow_log_prob(sd, spec, x), ref) < 2e-6


# ---- the oracle's GRADIENTS against the real reference's (tests/golden/grads_*.npz) ---------------------------
from golden_util import grad_case_names, load_grads  # noqa: E402


@pytest.mark.parametrize("name", grad_case_names())
def test_oracle_autograd_matches_reference_gradients(name):
    """what Flow.fit differentiates (flows.py:196-199): -log_prob(x, context).mean(); fp64 on both sides"""
    import copy
    spec, sd, a = load_case(name)
    loss_ref, g_ref = load_grads(name)
    sd64 = {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    if spec.base == "radial":
        spec = copy.copy(spec)
        spec.base_loc = sd64["base_distribution.loc"]
    ctx = a["context"].double() if "context" in a else None
    if ctx is None and spec.soft_training:
        ctx = torch.zeros(a["x"].shape[0], 1, dtype=torch.float64)
    loss = -orc.flow_log_prob(sd64, spec, a["x"].double(), ctx).mean()
    loss.backward()
    assert abs(loss.item() - loss_ref) <= 1e-10 * abs(loss_ref)
    checked = 0
    for k, ref in g_ref.items():
        if "norm_distribution" in k:
            continue                       # constants in the oracle's FlowSpec
        got = sd64[k].grad
        assert got is not None, k
        assert (got - ref.reshape(got.shape)).abs().max().item() <= 1e-9 * max(1.0, ref.abs().max().item()), k
        checked += 1
    assert checked >= 20


@pytest.mark.slow
@pytest.mark.skipif(os.environ.get("USFLOWS_SLOW") != "1", reason="minutes of CPU (98 triangular 3072x3072 inversions per "
                    "call, 10 GB of fp64 parameters): run with USFLOWS_SLOW=1")
def test_oracle_matches_reference_at_cfg4():
    """BASELINE cfg4 (D=3072, K=48, h=[1024,1024]): the oracle's fp32 log_prob of the fixture's 16 rows against the
    real reference's fp32 and fp64 runs"""
    spec, sd, a = load_case("synth_d3072_k48_cfg4")
    with torch.no_grad():
        lp = orc.flow_log_prob(sd, spec, a["x"])
    rel64 = ((lp.double() - a["log_prob64"]).abs() / a["log_prob64"].abs()).max().item()
    rel32 = ((lp.double() - a["log_prob32"].double()).abs() / a["log_prob64"].abs()).max().item()
    assert rel64 < 1e-5 and rel32 < 1e-5, (rel64, rel32)


def test_oracle_layer_plan_is_its_own_restatement_and_agrees_with_the_product():
    """the oracle restates the layer list of USFlow.__init__ (flows.py:434-482) itself; the product's copy
    (usflows_amd.synth.layer_plan, which the parameter generator and the engine tests walk) must agree with it, and both
    with the state-dict keys of a model the REAL reference built (the fixtures)"""
    from usflows_amd import synth
    assert orc.layer_plan is not synth.layer_plan and orc.layer_plan.__module__ == "oracle.usflows_oracle"
    S = orc.FlowSpec
    specs = [S(6, 1, [8], householder=0), S(6, 2, [8], householder=1), S(6, 3, [8], householder=2, affine_conjugation=True),
             S(6, 2, [8], lu_transform=2, householder=0, affine_conjugation=True), S(6, 2, [8], lu_transform=0, householder=0),
             S(6, 3, [8], lu_transform=0, householder=0, affine_conjugation=True), S(6, 4, [8], lu_transform=0, householder=1)]
    for spec in specs:
        assert orc.layer_plan(spec) == synth.layer_plan(spec), spec
    for name in case_names(small_only=True):
        spec, sd, _ = load_case(name)
        prefixes = {p for _, p, _, _ in orc.layer_plan(spec)}
        keys = [k for k in sd if k.startswith("trainable_layers.")]
        # every parameter of the reference's model lives under one of the plan's prefixes (InverseTransform re-registers its
        # block under `<idx>.transform.`: those aliases are the same tensors)
        for k in keys:
            if ".transform.block_transform." in k:
                continue
            assert any(k.startswith(p) for p in prefixes), (name, k)
        for p in prefixes:
            assert any(k.startswith(p) for k in keys), (name, p)


def test_oracle_owns_its_case_description_and_parameter_generator():
    """Nothing under oracle/ imports the product package (its FlowSpec and synth_state_dict live in oracle/synth.py); the
    product-side copies that bench.py / smoke() build their flows from (usflows_amd/synth.py) produce the same
    parameters bit for bit, so oracle, fixtures and device path still see one model."""
    import ast
    import glob
    import os
    from usflows_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for path in glob.glob(os.path.join(root, "oracle", "*.py")):
        for node in ast.walk(ast.parse(open(path).read())):
            mods = [a.name for a in node.names] if isinstance(node, ast.Import) else \
                [node.module or ""] if isinstance(node, ast.ImportFrom) and node.level == 0 else []
            assert not any(m.split(".")[0] == "usflows_amd" for m in mods), (path, mods)
    assert orc.FlowSpec is not synth.ModelSpec and orc.synth_state_dict is not synth.synth_state_dict
    kw = [dict(dim=6, coupling_blocks=2, hidden_dims=[8, 8]),
          dict(dim=7, coupling_blocks=3, hidden_dims=[16, 8], householder=2, affine_conjugation=True, conditioner="DenseNN"),
          dict(dim=9, coupling_blocks=2, hidden_dims=[12], lu_transform=2, householder=0, affine_conjugation=True,
               conditioner="ConvNet", extra={"gating": True, "normalize_layers": True})]
    for k in kw:
        a = orc.synth_state_dict(orc.FlowSpec(**k), seed=17, alpha=0.2)
        b = synth.synth_state_dict(synth.ModelSpec(**k), seed=17, alpha=0.2)
        assert list(a) == list(b)
        assert all(torch.equal(a[n], b[n]) for n in a), k
