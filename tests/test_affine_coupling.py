"""Affine (scale-and-shift) coupling EXTENSION (usflows_amd.transforms.AffineMaskedCoupling; BASELINE.json north_star,
SURVEY 7-H6 / 8a-C1).  The reference has additive coupling only: PARITY UNPINNED.  What is tested is self-consistency:
round trip, per-sample log-det against the autograd Jacobian, a change-of-variables check of log_prob, and -- on the
GPU -- the device path (usf_linear_f32 launches + usf_affine_coupling_apply_f32) against the torch formulation.
Flows with this layer are not uniformly scaling and take no part in any UDL check."""
import pytest
import torch

from usflows_amd.flows import Flow, USFlow
from usflows_amd.networks import ConditionalDenseNN, DenseNN
from usflows_amd.transforms import (AffineMaskedCoupling, BlockAffineTransform, LUTransform, MaskedCoupling,
                                    ScaleTransform)


def _coupling(D, seed, cond="dense", bound=2.0):
    torch.manual_seed(seed)
    mask = USFlow.create_checkerboard_mask([D])
    act = torch.nn.LeakyReLU(0.01)
    if cond == "dense":
        net = DenseNN(D, [16, 12], param_dims=[D, D], nonlinearity=act)
    else:
        net = ConditionalDenseNN(D, 1, [16, 16], 2 * D, nonlinearity=act)
    return AffineMaskedCoupling(mask, net, scale_bound=bound)


@pytest.mark.parametrize("cond,bound", [("dense", 2.0), ("conditional", None)])
def test_round_trip_and_logdet_vs_autograd_jacobian(cond, bound):
    D = 6
    c = _coupling(D, 3, cond, bound)
    x = torch.randn(9, D, generator=torch.Generator().manual_seed(1))
    y = c.forward(x)
    assert torch.allclose(c.backward(y), x, rtol=1e-5, atol=1e-6)
    assert torch.equal((y * c.mask), (x * c.mask))                       # pass-through features untouched
    ld = c.log_abs_det_jacobian(x, y)
    assert ld.shape == (9,)
    for i in range(9):
        J = torch.autograd.functional.jacobian(lambda v: c.forward(v[None])[0], x[i])
        assert abs(torch.linalg.slogdet(J)[1].item() - ld[i].item()) < 1e-5
    assert ld.std().item() > 1e-3                                         # data-dependent: NOT uniformly scaling


def _flow(D, seed, device="cpu"):
    torch.manual_seed(seed)
    base = torch.distributions.Laplace(torch.zeros(D, device=device), torch.ones(D, device=device))
    mask = USFlow.create_checkerboard_mask([D])
    layers = []
    for k in range(3):
        layers.append(BlockAffineTransform([D], LUTransform(D)))
        net = DenseNN(D, [16, 16], param_dims=[D, D], nonlinearity=torch.nn.LeakyReLU(0.01))
        layers.append(AffineMaskedCoupling(mask if k % 2 == 0 else 1 - mask, net))
    layers.append(ScaleTransform([D]))
    with torch.no_grad():
        for l in layers:
            if isinstance(l, ScaleTransform):
                l.scale.copy_(torch.linspace(0.7, 1.4, D))
            if isinstance(l, BlockAffineTransform):
                lu = l.block_transform
                lu.L_raw.copy_(torch.eye(D) + 0.2 * lu.L_raw.tril(-1))
                lu.U_raw.copy_(0.2 * lu.U_raw.triu(1) + torch.eye(D))
    return Flow(base, layers)


def test_flow_log_prob_is_the_change_of_variables_density():
    """log_prob(x) == base.log_prob(f^-1(x)) + log|det d f^-1 / dx| with the Jacobian of the WHOLE flow from autograd"""
    D = 4
    flow = _flow(D, 7)
    x = torch.rand(5, D, generator=torch.Generator().manual_seed(2))
    lp = flow.log_prob(x)
    for i in range(5):
        J = torch.autograd.functional.jacobian(lambda v: flow.backward(v[None])[0], x[i])
        z = flow.backward(x[i][None])
        ref = flow.base_distribution.log_prob(z)[0] + torch.linalg.slogdet(J)[1]
        assert abs(ref.item() - lp[i].item()) < 1e-4 * max(1.0, abs(ref.item()))
    assert torch.allclose(flow._forward(flow.backward(x)), x, rtol=1e-4, atol=1e-5)


def test_cached_device_logdet_is_keyed_on_identity_and_never_served_under_autograd():
    """The log-det a device call leaves behind belongs to ONE (input, output) pair: it is matched by object identity and
    version (a freed tensor's address can come back), dropped by the next forward / backward, and never returned where
    a gradient could be asked of it (it is detached)."""
    D = 6
    c = _coupling(D, 7)
    x = torch.randn(5, D, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        y = c.forward(x)
        true_ld = c.log_abs_det_jacobian(x, y)
    fake = torch.full((5,), 123.0)
    c._last = (x, y, (x._version, y._version), fake)                 # what a device call would have left
    with torch.no_grad():
        assert c.log_abs_det_jacobian(x, y) is fake                   # same pair, unmodified, no autograd: served
        assert torch.allclose(c.log_abs_det_jacobian(x.clone(), y), true_ld)      # equal values, other object: recomputed
    ld = c.log_abs_det_jacobian(x, y)                                 # grad mode on, trainable conditioner: recomputed
    assert ld.requires_grad and torch.allclose(ld, true_ld)
    x.add_(0.0)                                                       # in-place edit bumps the version
    with torch.no_grad():
        assert c.log_abs_det_jacobian(x, y) is not fake
        c._last = (x, y, (x._version, y._version), fake)
        c.forward(x)                                                  # any later call drops the cache (torch path too)
        assert c._last is None
        c._last = (x, y, (x._version, y._version), fake)
        c.backward(y)
        assert c._last is None


@pytest.mark.gpu
@pytest.mark.parametrize("cond,bound,B", [("dense", 2.0, 1), ("dense", 2.0, 777), ("conditional", None, 4100)])
def test_device_path_matches_torch_formulation(cond, bound, B):
    D = 16
    c_cpu = _coupling(D, 5, cond, bound)
    c_dev = _coupling(D, 5, cond, bound).to("cuda:0")
    x = torch.randn(B, D, generator=torch.Generator().manual_seed(B))
    with torch.no_grad():
        y_ref, ld_ref = c_cpu.forward(x), c_cpu.log_abs_det_jacobian(x, c_cpu.forward(x))
        xd = x.to("cuda:0")
        y = c_dev.forward(xd)
        ld = c_dev.log_abs_det_jacobian(xd, y)                     # served by the kernel's wave reduction (cached pair)
        xb = c_dev.backward(y)
        ld_b = c_dev.log_abs_det_jacobian(xb, y)
    assert c_dev.device_calls == 2, "the HIP path did not run"
    assert torch.allclose(y.cpu(), y_ref, rtol=2e-5, atol=2e-5)
    assert torch.allclose(ld.cpu(), ld_ref, rtol=2e-5, atol=2e-5)
    assert torch.allclose(xb.cpu(), x, rtol=1e-4, atol=1e-4) and torch.allclose(ld_b.cpu(), ld_ref, rtol=2e-5, atol=2e-5)
    assert torch.equal(y * c_dev.mask, xd * c_dev.mask)


@pytest.mark.gpu
def test_flow_with_affine_couplings_on_device():
    import warnings
    D = 16
    flow_cpu, flow_dev = _flow(D, 11), _flow(D, 11, "cuda:0")
    flow_dev = flow_dev.to("cuda:0")
    x = torch.rand(300, D, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        ref = flow_cpu.log_prob(x)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                        # (the one-time "layer loop, not the fused launch list" note)
            got = flow_dev.log_prob(x.to("cuda:0"))
    assert ((got.cpu() - ref).abs() / ref.abs()).max().item() < 2e-5
    assert all(l.device_calls > 0 for l in flow_dev.layers if isinstance(l, AffineMaskedCoupling))
