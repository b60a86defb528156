"""Host-side mirror of the reference API on CPU: constructor surface, state-dict layout, composite
(autograd) path vs the golden vectors, feasibility helpers, fit()."""
import math

import pytest
import torch

from golden_util import case_names, load_case
from model_util import build_flow
from oracle import usflows_oracle as orc
from usflows_amd.flows import Flow, USFlow
from usflows_amd.networks import ConditionalDenseNN, DenseNN
from usflows_amd import transforms as T
from usflows_amd import distributions as D

SMALL = case_names(small_only=True)


def _rel(a, b):
    return ((a.double() - b.double()).abs() / b.double().abs().clamp_min(1e-30)).max().item()


@pytest.mark.parametrize("name", SMALL)
def test_composite_path_matches_golden(name):
    spec, sd, a = load_case(name)
    flow = build_flow(spec, sd)
    ctx = a.get("context")
    with torch.no_grad():
        lp = flow.log_prob(a["x"], context=ctx) if ctx is not None else flow.log_prob(a["x"])
    assert _rel(lp, a["log_prob64"]) < 2e-5
    assert _rel(lp, a["log_prob32"]) < 5e-6


@pytest.mark.parametrize("name", SMALL)
def test_state_dict_keys_match_reference_layout(name):
    """strict key-for-key equality with the state dict the REFERENCE model produced (fixture)"""
    spec, sd, _ = load_case(name)
    flow = build_flow(spec)
    ours = set(flow.state_dict().keys())
    theirs = set(sd.keys())
    assert ours == theirs, (sorted(ours - theirs)[:5], sorted(theirs - ours)[:5])
    for k, v in flow.state_dict().items():
        assert tuple(v.shape) == tuple(sd[k].shape), k


def test_layer_order_matches_reference():
    f = USFlow(torch.distributions.Laplace(torch.zeros(6), torch.ones(6)), [6], 2, ConditionalDenseNN,
               dict(input_dim=6, context_dim=1, hidden_dims=[8], out_dim=6), affine_conjugation=True, householder=1)
    names = [type(l).__name__ for l in f.layers]
    assert names == ["BlockAffineTransform", "MaskedCoupling", "InverseTransform"] * 2 + ["BlockAffineTransform", "ScaleTransform"]
    assert f.layers[2].transform is f.layers[0]                 # InverseTransform shares the block
    assert f.layers[1].mask.flatten().tolist() == [0, 1, 0, 1, 0, 1]
    assert f.layers[4].mask.flatten().tolist() == [1, 0, 1, 0, 1, 0]
    f2 = USFlow(torch.distributions.Laplace(torch.zeros(6), torch.ones(6)), [6], 2, ConditionalDenseNN,
                dict(input_dim=6, context_dim=1, hidden_dims=[8], out_dim=6), householder=0)
    assert [type(l).__name__ for l in f2.layers] == ["BlockAffineTransform", "MaskedCoupling"] * 2 + \
        ["BlockAffineTransform", "ScaleTransform"]


def test_constructor_errors():
    base = torch.distributions.Laplace(torch.zeros(4), torch.ones(4))
    args = dict(input_dim=4, context_dim=1, hidden_dims=[8], out_dim=4)
    with pytest.raises(ValueError):
        USFlow(base, [4], 1, ConditionalDenseNN, args, masktype="diagonal")
    with pytest.raises(ValueError):
        USFlow(base, [4], 1, ConditionalDenseNN, args, lu_transform=-1)
    with pytest.raises(ValueError):
        USFlow(base, [4], 1, ConditionalDenseNN, args, householder=-1)
    with pytest.raises(ValueError):
        T.BlockAffineTransform([5], T.LUTransform(4))
    with pytest.raises(ValueError):
        T.SequentialAffineTransform([T.LUTransform(4), T.LUTransform(5)])
    # image-shaped in_dims: the 1x1-conv form, n_blocks = prod(spatial) (transforms.py:904-911)
    blk = T.BlockAffineTransform([4, 2, 3], T.LUTransform(4))
    assert blk.n_blocks == 6 and blk.input_rank == 2 and blk.global_transform is torch.nn.functional.conv2d
    with pytest.raises(KeyError):
        T.BlockAffineTransform([4, 2, 2, 2, 2], T.LUTransform(4))        # rank 5: no conv form, as in the reference


def test_reference_known_answer_tests_on_product_layers():
    """tests/veriflow/transforms_test.py:5-19 and :35-51 restated on usflows_amd's layers"""
    dim = 10
    st = T.ScaleTransform([dim])
    with torch.no_grad():
        st.scale.copy_(torch.ones(dim) * 2)
    x = torch.ones(dim)
    y = st(x)
    assert (y == 2 * x).all() and (st.backward(y) == x).all()
    assert st.log_abs_det_jacobian(x, y) == dim * torch.log(torch.tensor(2.0))
    lu = T.LUTransform(dim)
    with torch.no_grad():
        lu.L_raw.copy_(torch.tril(torch.ones(dim, dim)))
        lu.U_raw.copy_(torch.eye(dim))
        lu.bias_vector.copy_(torch.zeros(dim))
    y = lu(x)
    assert (y == torch.arange(dim) + 1.0).all()
    assert (lu.backward(y) == x).all()
    assert lu.log_abs_det_jacobian(x, y) == 0


def test_feasibility_and_jitter():
    spec, sd, _ = load_case("synth_d7_k3_hh0_laplace")
    flow = build_flow(spec, sd)
    assert flow.is_feasible()
    lu = flow.layers[0].block_transform.transforms[0]
    with torch.no_grad():
        lu.U_raw[2, 2] = 0.0
    # as in the reference, the wrappers (BlockAffineTransform / SequentialAffineTransform) do not forward
    # is_feasible / add_jitter (transforms.py:32-34): only the LU factor itself reports the zero pivot
    assert flow.is_feasible() and not lu.is_feasible()
    lu.add_jitter(1e-3)
    assert lu.is_feasible()
    with torch.no_grad():
        flow.layers[-1].scale[1] = 0.0          # ScaleTransform is a direct layer: this one Flow.is_feasible sees
    assert not flow.is_feasible()
    flow.add_jitter(1e-3)
    assert flow.is_feasible()


def test_fit_runs_and_reduces_loss_on_cpu():
    torch.manual_seed(0)
    D_ = 4
    flow = USFlow(torch.distributions.Laplace(torch.zeros(D_), torch.ones(D_)), [D_], 2, ConditionalDenseNN,
                  dict(input_dim=D_, context_dim=1, hidden_dims=[16], out_dim=D_, nonlinearity=torch.nn.LeakyReLU(0.01)),
                  householder=0)
    with torch.no_grad():        # tame the reference's default init
        flow.layers[-1].scale.copy_(torch.ones(D_))
    data = torch.randn(512, D_) * 0.5 + 1.0
    ds = torch.utils.data.TensorDataset(data)
    losses = flow.fit(ds, optim=torch.optim.Adam, optim_params=dict(lr=1e-2), batch_size=64, epochs=6,
                      device=torch.device("cpu"))
    assert len(losses) == 6 and losses[-1] < losses[0]


def test_soft_training_fit_step():
    torch.manual_seed(0)
    D_ = 4
    flow = USFlow(torch.distributions.Laplace(torch.zeros(D_), torch.ones(D_)), [D_], 1, ConditionalDenseNN,
                  dict(input_dim=D_, context_dim=1, hidden_dims=[8], out_dim=D_), householder=0, soft_training=True,
                  training_noise_prior=torch.distributions.Uniform(1e-20, 0.01))
    with torch.no_grad():
        flow.layers[-1].scale.copy_(torch.ones(D_))
    ds = torch.utils.data.TensorDataset(torch.randn(64, D_))
    losses = flow.fit(ds, optim=torch.optim.SGD, optim_params=dict(lr=1e-4), batch_size=32, epochs=1,
                      device=torch.device("cpu"))
    assert math.isfinite(losses[0])
    assert flow.log_prob(torch.randn(5, D_)).shape == (5,)      # implicit zero context


def test_forward_export_modes_and_sample_shape():
    spec, sd, a = load_case("synth_d7_k3_hh0_laplace")
    flow = build_flow(spec, sd)
    with torch.no_grad():
        flow.export = "log_prob"
        assert torch.equal(flow(a["x"]), flow.log_prob(a["x"]))
        flow.export = "backward"
        assert torch.equal(flow(a["x"]), flow.backward(a["x"]))
        flow.export = "forward"
        assert torch.equal(flow(a["x"]), flow._forward(a["x"]))
        flow.export = "nope"
        with pytest.raises(ValueError):
            flow(a["x"])
        assert flow.sample([5]).shape == (5, 7) and flow.sample().shape == (1, 7)


def test_densenn_matches_conditional_densenn_without_context():
    torch.manual_seed(1)
    a = ConditionalDenseNN(6, 1, [8, 8], 6, torch.nn.LeakyReLU(0.01))
    b = DenseNN(6, [8, 8], [6], torch.nn.LeakyReLU(0.01))
    with torch.no_grad():
        b.layers[0].load_state_dict(a.layers[0].state_dict())
        b.layers[1].load_state_dict(a.layers[2].state_dict())
        b.layers[2].load_state_dict(a.layers[3].state_dict())
    x = torch.randn(9, 6)
    assert torch.equal(a(x), b(x))


def test_radial_distribution_matches_oracle():
    spec, sd, a = load_case("synth_d16_k4_hh0_conj_radial1")
    flow = build_flow(spec, sd)
    z = a["backward32"]
    assert _rel(flow.base_distribution.log_prob(z), orc.base_log_prob(spec, z)) < 1e-6
    s = flow.base_distribution.sample([100])
    assert s.shape == (100, 16) and torch.isfinite(s).all()


def test_simplify_is_equivalent():
    spec, sd, a = load_case("synth_d7_k3_hh1_conj_normal")
    flow = build_flow(spec, sd)
    with torch.no_grad():
        simple = flow.simplify()
        assert torch.allclose(simple.log_prob(a["x"]), flow.log_prob(a["x"]), rtol=1e-4, atol=1e-4)


def test_convnet_vector_path_layout_and_engine_view():
    """vector path of the reference's generic ConvNet (networks.py:287-308): module order / state-dict keys, the
    default gated + layer-normalised variant and its block view, and the plain variant's folded MLP view"""
    from usflows_amd.networks import ConvNet, GatedMLP, LayerNormVector
    from usflows_amd.engine import conditioner_supported
    torch.manual_seed(0)
    plain = ConvNet([6], [8, 5], nonlinearity=torch.nn.LeakyReLU(0.01), normalize_layers=False, gating=False)
    assert list(plain.state_dict()) == ["nn.0.weight", "nn.0.bias", "nn.1.1.weight", "nn.1.1.bias", "nn.2.1.weight",
                                        "nn.2.1.bias", "nn.3.weight", "nn.3.bias"]
    assert plain.nn[1][1].weight.shape == (8, 8) and plain.nn[2][1].weight.shape == (5, 8) and plain.nn[3].weight.shape == (6, 5)
    x = torch.randn(9, 6)
    first, hidden, (W_out, b_out), widths = plain.mlp_view()
    h = torch.nn.functional.leaky_relu(first(x), 0.01)
    for l in hidden:
        h = torch.nn.functional.leaky_relu(l(h), 0.01)
    folded = h.double() @ W_out.t() + b_out
    assert widths == [8, 8] and torch.allclose(folded.float(), plain(x), atol=1e-6)
    assert conditioner_supported(plain)
    # accepted input layouts (networks.py:379-387)
    assert torch.equal(plain(x.unsqueeze(-1)), plain(x))
    default = ConvNet([6], [8, 5])
    assert isinstance(default.nn[1], GatedMLP) and isinstance(default.nn[2], LayerNormVector)
    assert default.nn[3].proj is not None and default.nn[1].proj is None      # 8 -> 5 needs the residual projection
    assert default(x).shape == (9, 6) and conditioner_supported(default)      # device form: chain of ops (engine.py)
    first, blocks, final = default.block_view()
    assert first is default.nn[0] and final is default.nn[-1] and [b["w_in"] for b in blocks] == [8, 8]
    assert blocks[0]["l1"] is default.nn[1].net1[1] and blocks[1]["proj"] is default.nn[3].proj
    assert blocks[1]["ln"] is default.nn[4].layernorm and "lin" not in blocks[0]
    assert not conditioner_supported(ConvNet([6], [8, 5], nonlinearity=torch.nn.Tanh()))
    spatial = ConvNet([3, 8, 8], [4])        # the spatial path: image-shaped flows (layer loop), not the flat engine
    assert not spatial.is_vector and not conditioner_supported(spatial) and spatial(torch.randn(2, 3, 8, 8)).shape == (2, 3, 8, 8)


# ---- Flow.fit against a golden run of the real reference (tests/golden/fit_*.npz) ---------------------------------
def _run_fit(flow, data, device):
    import numpy as np
    ds = torch.utils.data.TensorDataset(data, torch.zeros(data.shape[0]))
    np.random.seed(5)
    return flow.fit(ds, optim=torch.optim.SGD, optim_params=dict(lr=1e-3), batch_size=32, shuffle=True,
                    device=torch.device(device), epochs=2)


def _check_fit(flow, losses, losses_ref, sd_ref, tol):
    for a, b in zip(losses, losses_ref):
        assert abs(a - b) <= 1e-4 * abs(b), (losses, losses_ref)
    sd = flow.state_dict()
    n = 0
    for k, ref in sd_ref.items():
        if not ref.is_floating_point() or k not in sd:
            continue
        got = sd[k].detach().cpu()
        assert (got - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item()), k
        n += 1
    assert n >= 20


def test_fit_matches_reference_run_cpu():
    """the mirror's training loop (shuffling, batching, loss, SGD steps) on CPU = the composite formulation"""
    from golden_util import fit_case_names, load_case, load_fit
    from model_util import build_flow
    for name in fit_case_names():
        spec, sd, _ = load_case(name)
        data, losses_ref, sd_ref, prior_scale = load_fit(name)
        if prior_scale is not None:
            spec.extra["prior_scale"] = prior_scale      # the loss then carries -log_prior() (transforms.py:1371-1379)
        flow = build_flow(spec, sd)
        losses = _run_fit(flow, data, "cpu")
        _check_fit(flow, losses, losses_ref, sd_ref, 2e-5)


# ---- the UDL machinery (flows.py:294-378, distributions.py:390-456) against the real reference -------------------
def _check_udl(flow, x, g):
    step = g["r_max"] / g["n_samples"]
    for cut in (True, False):
        prof = flow.calibrated_latent_radial_udl_profile(g["q"], x, r_max=g["r_max"], n_samples=g["n_samples"],
                                                         cut_to_data_tail=cut).cpu().double()
        ref = g["cut" if cut else "full"]
        assert prof.shape == ref.shape, (prof, ref)
        assert (prof - ref).abs().max().item() <= 1.5 * step, (prof, ref)      # interval ends sit on the r grid


def test_udl_profile_matches_reference_cpu():
    from golden_util import load_case, load_udl, udl_case_names
    from model_util import build_flow
    for name in udl_case_names():
        spec, sd, a = load_case(name)
        _check_udl(build_flow(spec, sd), a["x"], load_udl(name))


def test_round2_host_switches_are_inert_on_the_cpu():
    """the device-only shortcuts added to the host code leave CPU calls on the reference's formulation: no graphed training
    step, `optim.zero_grad()` semantics, no cached log-determinant total, LUTransform's inverse through torch.inverse"""
    spec, sd, a = load_case("synth_d7_k3_hh0_laplace")
    flow = build_flow(spec, sd)
    x = a["x"]
    opt = torch.optim.SGD(flow.parameters(), lr=1e-3)
    assert flow._train_graph_step(opt, x, None) is None and "_train_graph_state" not in flow.__dict__
    loss = -flow.log_prob(x).mean()
    loss.backward()
    assert any(p.grad is not None for p in flow.parameters())
    flow._zero_grad_for_step(opt)                                   # no captured graph: plain zero_grad()
    assert all(p.grad is None or float(p.grad.abs().sum()) == 0.0 for p in flow.parameters())
    with torch.no_grad():
        assert flow._parameter_only_ladj_total(x) is None
    lu = flow.layers[0].block_transform.transforms[0]
    assert torch.equal(lu._tri_inverse(lu.L, False), torch.inverse(lu.L))
    from usflows_amd.flows import _ladj_is_parameter_only
    assert all(_ladj_is_parameter_only(l) for l in flow.layers)


def test_lazy_log_det_sum_equals_the_layer_by_layer_subtraction():
    """Flow._layer_loop_log_prob in training collects the layers' log-dets lazily (numbers on the host, scalars stacked once,
    per-sample terms as tensors): same value and same gradients as the reference's `log_det = log_det - term` chain
    (flows.py:236-245)"""
    from usflows_amd.flows import _LogDetSum
    g = torch.Generator().manual_seed(0)
    B = 7
    lp = torch.randn(B, generator=g)
    a = torch.randn((), generator=g, requires_grad=True)
    b = torch.randn((), generator=g, requires_grad=True)
    v = torch.randn(B, generator=g, requires_grad=True)
    terms = [0.0, a * 3.0, 1.25, v * 2.0, b.exp(), 0, v.sin()]
    ld = _LogDetSum()
    for t in terms:
        ld.sub(t)
    out = ld.add_to(lp)
    ref = torch.zeros(B)
    for t in terms:
        ref = ref - t
    ref = lp + ref
    assert torch.allclose(out, ref, rtol=1e-6, atol=1e-6)
    w = torch.randn(B, generator=g)
    g1 = torch.autograd.grad((out * w).sum(), [a, b, v], retain_graph=True)
    g2 = torch.autograd.grad((ref * w).sum(), [a, b, v])
    for x, y in zip(g1, g2):
        assert torch.allclose(x, y, rtol=1e-5, atol=1e-6)
    # nothing collected: the density passes through unchanged
    assert _LogDetSum().add_to(lp) is lp
