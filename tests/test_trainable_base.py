"""Flows over the reference's TRAINABLE base modules -- distributions.Laplace / distributions.Normal (src/usflows/distributions.py:
199-238; loc and a softplus-constrained scale as nn.Parameters) -- under Flow.fit's loss: every gradient, the base's included,
against goldens of the REAL reference (tests/golden/make_golden_trainable_base.py).  Round 4 sent such flows to the torch
composite formulation; now the device training path serves them (usf_base_param_grad_f32)."""
import warnings

import pytest
import torch

import emulator
from golden_util import load_trainable_base_case, trainable_base_case_names
from model_util import build_flow

DEV = "cuda:0"


def _check(flow, g_ref, tol):
    named = dict(flow.named_parameters())
    assert "base_distribution.loc" in g_ref and "base_distribution.scale_unconstrained" in g_ref
    gmax = max(g.abs().max().item() for g in g_ref.values())
    for k, g in g_ref.items():
        assert named[k].grad is not None, k
        d = (named[k].grad.detach().cpu().double().reshape(g.shape) - g.double()).abs().max().item()
        assert d <= tol * g.abs().max().item() + 0.2 * tol * gmax, (k, d, g.abs().max().item())


@pytest.mark.parametrize("name", trainable_base_case_names())
def test_mirror_gradients_cpu(name):
    """the mirror's torch formulation in fp64 (what a CPU run of Flow.fit differentiates)"""
    spec, sd, x, lp_ref, loss_ref, g_ref = load_trainable_base_case(name)
    torch.set_default_dtype(torch.float64)
    try:
        flow = build_flow(spec, sd).double()
        for l in flow.layers:
            if hasattr(l, "mask") and torch.is_tensor(l.mask):
                l.mask = l.mask.double()
        lp = flow.log_prob(x.double())
        assert ((lp.detach() - lp_ref).abs() / lp_ref.abs()).max().item() < 1e-11
        (-lp.mean()).backward()
    finally:
        torch.set_default_dtype(torch.float32)
    _check(flow, g_ref, 1e-8)


@pytest.mark.parametrize("name", trainable_base_case_names())
def test_training_path_with_emulated_entry_points_cpu(name, monkeypatch):
    """host logic of the device training path for a trainable base (persistent loc / scale buffers, the softplus chain rule,
    broadcast parameters), every entry point emulated from its documented semantics"""
    emulator.install_training_emulation(monkeypatch)
    from usflows_amd import training
    from usflows_amd.training import TrainPath
    spec, sd, x, lp_ref, loss_ref, g_ref = load_trainable_base_case(name)
    flow = build_flow(spec, sd)
    if spec.dim >= 128:
        eng = flow.engine()
        eng.use_planes, eng.planes_min_rows, eng.fused_min_rows, eng.train_planes_min_rows = True, 0, 0, 0
    path = TrainPath(flow)
    assert path.supported(x, None)
    lp = training.log_prob_with_grad(path, x, None)
    assert ((lp.detach().double() - lp_ref).abs() / lp_ref.abs()).max().item() < 2e-5
    (-lp.mean()).backward()
    _check(flow, g_ref, 2e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("planes", [False, True])
@pytest.mark.parametrize("name", trainable_base_case_names())
def test_device_gradients_match_the_reference(name, planes):
    spec, sd, x, lp_ref, loss_ref, g_ref = load_trainable_base_case(name)
    if planes and spec.dim < 128:
        pytest.skip("the planes training plan needs segments of at least 64 features")
    flow = build_flow(spec, sd, device=DEV)
    eng = flow.engine()
    if planes:
        eng.use_planes, eng.planes_min_rows, eng.fused_min_rows, eng.train_planes_min_rows = True, 0, 0, 0
    xd = x.to(DEV)
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)           # no composite fallback (flows._warn_composite)
        before = eng.launch_count
        lp = flow.log_prob(xd)
        assert lp.requires_grad and eng.launch_count > before, "the device training path did not run"
        assert bool(eng._plan("backward", x.shape[0], torch.device(DEV), False, "nat", train=True).get("planes_train")) == planes
        assert ((lp.detach().cpu().double() - lp_ref).abs() / lp_ref.abs()).max().item() < 1e-5
        (-lp.mean()).backward()
        _check(flow, g_ref, 5e-5)
        # a second step after the parameters moved (the recorded backward launches read the base's refreshed loc / scale buffers)
        with torch.no_grad():
            flow.base_distribution.loc.add_(0.05)
            flow.base_distribution.scale_unconstrained.add_(0.1)
        for p in flow.parameters():
            p.grad = None
        lp2 = flow.log_prob(xd)
        (-lp2.mean()).backward()
    ref = build_flow(spec, {k: v.detach().cpu() for k, v in flow.state_dict().items()})          # torch autograd on the CPU
    (-ref.log_prob(x).mean()).backward()
    g2 = {k: p.grad for k, p in ref.named_parameters() if p.grad is not None}
    named = dict(flow.named_parameters())
    for k in ("base_distribution.loc", "base_distribution.scale_unconstrained"):
        d = (named[k].grad.cpu() - g2[k]).abs().max().item()
        assert d <= 2e-4 * g2[k].abs().max().item() + 1e-6, (k, d)
