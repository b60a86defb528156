"""Host logic of the device training path (usflows_amd/training.py) on CPU: forward + hand-derived backward with every
HIP entry point replaced by its documented semantics (tests/emulator.py), against autograd through the oracle's
fp64 restatement of Flow.log_prob (what Flow.fit differentiates, flows.py:196-199)."""
import pytest
import torch

from golden_util import load_case
from model_util import build_flow
from oracle import usflows_oracle as orc
from usflows_amd.training import TrainPath
import emulator

CASES = ["synth_d7_k3_hh0_laplace", "synth_d16_k3_densenn_relu", "synth_d7_k3_hh1_conj_normal",
         "synth_d16_k4_hh2_conj_laplace", "synth_d33_k3_lu2_hh1", "synth_d7_k3_soft_ctx", "synth_d64_k6_hh0_laplace",
         "init_d2_k4_hh0_laplace", "synth_d16_k3_hh0_radialinf", "synth_d16_k3_hh1_radial2",
         "synth_d16_k4_hh0_conj_radial1",
         # the vector ConvNet conditioner with GatedMLP / LayerNormVector blocks (training.py: _coupling_backward_general)
         "synth_d16_k3_convnet_gated_ln", "synth_d33_k2_convnet_gated_conj", "synth_d64_k3_convnet_ln", "init_d4_k2_convnet_default"]


@pytest.fixture(autouse=True)
def _emulated(monkeypatch):
    emulator.install_training_emulation(monkeypatch)


def oracle_grads(spec, sd, x, g_lp, context=None):
    sd64 = {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    if spec.base == "radial" and "base_distribution.loc" in sd64:
        import copy
        spec = copy.copy(spec)
        spec.base_loc = sd64["base_distribution.loc"]          # trainable loc of RadialDistribution
    lp = orc.flow_log_prob(sd64, spec, x.double(), context.double() if context is not None else None)
    (lp * g_lp.double()).sum().backward()
    return lp.detach(), {k: v.grad for k, v in sd64.items() if torch.is_tensor(v) and v.is_floating_point()}


@pytest.mark.parametrize("name", CASES)
def test_training_path_gradients_match_oracle_autograd(name):
    spec, sd, a = load_case(name)
    flow = build_flow(spec, sd)
    x = a["x"]
    ctx = a.get("context")
    if ctx is None and spec.soft_training:
        ctx = torch.zeros(x.shape[0], 1)
    g_lp = torch.randn(x.shape[0], generator=torch.Generator().manual_seed(1))
    path = TrainPath(flow)
    assert path.supported(x, ctx)
    from usflows_amd import training
    lp = training.log_prob_with_grad(path, x, ctx)              # one autograd node (+ the radial finishing formula)
    (lp * g_lp).sum().backward()
    lp = lp.detach()
    lp_ref, g_ref = oracle_grads(spec, sd, x, g_lp, ctx)
    assert ((lp.double() - lp_ref).abs() / lp_ref.abs()).max().item() < 2e-5
    checked = 0
    for pname, p in flow.named_parameters():
        if not p.requires_grad or pname not in g_ref or "norm_distribution" in pname:
            continue            # (the radial norm distribution's parameters are constants in the oracle; their gradient
            #                      comes from torch autograd through the finishing formula, outside the node)
        ref = g_ref[pname]          # None: the oracle never touched it (context layer without context)
        got = p.grad
        if ref is None or ref.abs().max().item() == 0.0:
            assert got is None or got.abs().max().item() < 1e-6, pname
            continue
        assert got is not None, f"no gradient for {pname}"
        err = (got.double() - ref.reshape(got.shape)).abs().max().item()
        assert err <= 2e-4 * ref.abs().max().item(), (pname, err, ref.abs().max().item())
        checked += 1
    assert checked >= 5


def test_autograd_node_and_optimizer_step(monkeypatch):
    """Flow-level wiring: loss.backward() through the single autograd node fills .grad of every trainable parameter"""
    from usflows_amd import training
    spec, sd, a = load_case("synth_d16_k3_densenn_relu")
    flow = build_flow(spec, sd)
    path = TrainPath(flow)
    x = a["x"]
    loss = -training.log_prob_with_grad(path, x, None).mean()
    loss.backward()
    lp_ref, g_ref = oracle_grads(spec, sd, x, torch.full((x.shape[0],), -1.0 / x.shape[0]))
    n = 0
    for pname, p in flow.named_parameters():
        if p.requires_grad and p.grad is not None and pname in g_ref and g_ref[pname] is not None:
            ref = g_ref[pname].reshape(p.shape)
            assert (p.grad.double() - ref).abs().max().item() <= 2e-4 * max(ref.abs().max().item(), 1e-12), pname
            n += 1
    assert n >= 5
    assert abs(loss.item() + lp_ref.mean().item()) < 2e-5 * abs(lp_ref.mean().item())


from golden_util import grad_case_names, load_grads  # noqa: E402


@pytest.mark.parametrize("name", grad_case_names())
def test_training_path_matches_reference_gradients(name):
    """the device training path's host logic (entry points emulated) against gradients computed by the REAL
    reference (tests/golden/grads_*.npz, fp64): loss = -log_prob(x, context).mean() as in Flow.fit"""
    from usflows_amd import training
    spec, sd, a = load_case(name)
    loss_ref, g_ref = load_grads(name)
    flow = build_flow(spec, sd)
    x = a["x"]
    ctx = a.get("context")
    if ctx is None and spec.soft_training:
        ctx = torch.zeros(x.shape[0], 1)
    loss = -training.log_prob_with_grad(TrainPath(flow), x, ctx).mean()
    loss.backward()
    assert abs(loss.item() - loss_ref) <= 2e-5 * abs(loss_ref)
    checked = 0
    for pname, p in flow.named_parameters():
        if pname not in g_ref:
            assert p.grad is None or not p.requires_grad or p.grad.abs().max().item() < 1e-6, pname
            continue
        ref = g_ref[pname].reshape(p.shape)
        assert p.grad is not None, pname
        tol = 2e-4 * max(ref.abs().max().item(), 1e-9)
        assert (p.grad.double() - ref).abs().max().item() <= tol, pname
        checked += 1
    assert checked >= 20


@pytest.mark.parametrize("name", ["synth_d7_k3_soft_ctx", "synth_d7_k3_hh0_laplace"])
def test_eval_pass_between_forward_and_backward_does_not_corrupt_gradients(name):
    """train and eval plans share the (B, device) workspace.  A no_grad pass with the same batch size
    between loss = -log_prob(x).mean() and loss.backward() overwrites the saved latent, the staged input (D % 4 != 0)
    and the context column; the backward must notice (workspace pass counter) and re-run its forward."""
    from usflows_amd import training
    spec, sd, a = load_case(name)
    x, ctx = a["x"], a.get("context")
    if ctx is None and spec.soft_training:
        ctx = torch.zeros(x.shape[0], 1)

    def grads(disturb):
        flow = build_flow(spec, sd)
        path = TrainPath(flow)
        loss = -training.log_prob_with_grad(path, x, ctx).mean()
        if disturb:
            other = torch.rand(x.shape, generator=torch.Generator().manual_seed(9)) * 5
            octx = None if ctx is None else torch.full_like(ctx, 1.7)
            with torch.no_grad():
                flow.engine().latent(other, octx)                   # Flow.log_prob's no_grad path, same batch size
                flow.engine().transform(other, "backward", octx)    # Flow.backward
        loss.backward()
        return {n: p.grad.clone() for n, p in flow.named_parameters() if p.grad is not None}

    clean, dirty = grads(False), grads(True)
    assert set(clean) == set(dirty) and len(clean) >= 5
    for n in clean:
        assert torch.allclose(clean[n], dirty[n], rtol=1e-6, atol=1e-9), n


def test_parameter_update_between_forward_and_backward_raises():
    """an optimiser step between log_prob and its backward -> a clear error instead of a gradient
    evaluated at other parameters than the returned log_prob"""
    from usflows_amd import training
    spec, sd, a = load_case("synth_d7_k3_hh0_laplace")
    flow = build_flow(spec, sd)
    loss = -training.log_prob_with_grad(TrainPath(flow), a["x"], None).mean()
    with torch.no_grad():
        for p in flow.parameters():
            p.mul_(1.001)
    with pytest.raises(RuntimeError, match="parameters were modified"):
        loss.backward()


def _grads_of(flow, x, g_lp, ctx=None, defer=True):
    from usflows_amd import training
    path = TrainPath(flow)
    path.defer_small_grads = defer
    assert path.supported(x, ctx)
    for p in flow.parameters():
        p.grad = None
    lp = training.log_prob_with_grad(path, x, ctx)
    (lp * g_lp).sum().backward()
    return lp.detach(), {n: p.grad.clone() for n, p in flow.named_parameters() if p.grad is not None}, path


@pytest.mark.parametrize("name", ["synth_d16_k3_densenn_relu", "synth_d16_k4_hh2_conj_laplace", "synth_d7_k3_soft_ctx"])
def test_queued_gradient_jobs_equal_launching_each_in_place(name):
    """the small-batch weight / bias gradients wait until the layer loop is over and leave as one launch
    (usf_grad_jobs_f32; the emulation runs the queue at the flush, ahead of the queued scatter jobs, as the library does):
    every layer's operands must still hold what they held when the job was queued -- same gradients, bit for bit, as with
    each gradient launched at its place"""
    spec, sd, a = load_case(name)
    flow = build_flow(spec, sd)
    x, ctx = a["x"], a.get("context")
    if ctx is None and spec.soft_training:
        ctx = torch.zeros(x.shape[0], 1)
    g_lp = torch.randn(x.shape[0], generator=torch.Generator().manual_seed(1))
    lp_d, g_d, _ = _grads_of(flow, x, g_lp, ctx, defer=True)
    lp_n, g_n, _ = _grads_of(flow, x, g_lp, ctx, defer=False)
    assert torch.equal(lp_d, lp_n) and g_d.keys() == g_n.keys() and len(g_d) >= 6
    for k in g_d:
        # (the queued form also reads the hidden activations the forward pass saved instead of recomputing them: on the
        # device the same kernel on the same operands; the emulation's forward runs in fp64 and its recomputation in fp32)
        scale = float(g_n[k].abs().max())
        assert torch.allclose(g_d[k], g_n[k], rtol=1e-5, atol=2e-6 * max(scale, 1e-12)), k


@pytest.mark.parametrize("name", ["synth_d16_k3_densenn_relu", "synth_d7_k3_hh1_conj_normal"])
def test_bound_flat_gradients_accumulate_like_autograd(name):
    """TrainPath.bind_flat_grads: the parameters' .grad become views of one buffer and backward adds the whole arena at once
    -- same values as autograd's per-parameter accumulation, accumulating over two backward passes, undone by grad = None"""
    from usflows_amd import training
    spec, sd, a = load_case(name)
    flow = build_flow(spec, sd)
    x, ctx = a["x"], a.get("context")
    g_lp = torch.randn(x.shape[0], generator=torch.Generator().manual_seed(2))
    _lp, g1, path = _grads_of(flow, x, g_lp, ctx)
    assert path.bind_flat_grads()
    named = dict(flow.named_parameters())
    ptrs = {n: named[n].grad.data_ptr() for n in g1}
    for n in g1:
        assert torch.equal(named[n].grad, g1[n])            # current values kept
    path._gflat.zero_()
    # outside the scope that asks for the bound node, log_prob is an ordinary autograd node over the parameters
    lp = training.log_prob_with_grad(path, x, ctx)
    gs = torch.autograd.grad((lp * g_lp).sum(), [named[n] for n in g1])
    for n, g in zip(g1, gs):
        assert torch.allclose(g, g1[n], rtol=1e-6, atol=1e-7 * float(g1[n].abs().max())), n
        assert named[n].grad.data_ptr() == ptrs[n] and not named[n].grad.any()      # (autograd.grad has no side effect on .grad)
    path.use_bound_node = True                               # what Flow.fit's captured step sets
    for rep in (1, 2):
        lp = training.log_prob_with_grad(path, x, ctx)
        (lp * g_lp).sum().backward()
        for n in g1:
            assert named[n].grad.data_ptr() == ptrs[n]       # still the views: nothing was handed to AccumulateGrad
            assert torch.allclose(named[n].grad, rep * g1[n], rtol=1e-6, atol=1e-7 * float(g1[n].abs().max())), (n, rep)
    some = next(iter(g1))
    named[some].grad = None                                  # the binding is gone: autograd's own accumulation again
    for p in flow.parameters():
        if p.grad is not None:
            p.grad.zero_()
    lp = training.log_prob_with_grad(path, x, ctx)
    (lp * g_lp).sum().backward()
    for n in g1:
        assert torch.allclose(named[n].grad, g1[n], rtol=1e-6, atol=1e-7 * float(g1[n].abs().max())), n


# ---- the training step on the planes pipeline (round 5): engine._build_plan_planes(train=True) + TrainPath._backward_body_planes ----
def _planes_flow(D, K, hidden, conj=False, hh=0, base="laplace", seed=3):
    from oracle.synth import ModelSpec, synth_state_dict
    spec = ModelSpec(dim=D, coupling_blocks=K, hidden_dims=hidden, householder=hh, affine_conjugation=conj, base=base)
    sd = synth_state_dict(spec, seed=seed)
    return spec, sd


@pytest.mark.parametrize("D,K,hidden,conj,hh", [(160, 2, [96, 64], False, 0), (136, 3, [72], False, 0), (160, 2, [64, 64], True, 1)])
def test_planes_training_path_gradients_match_oracle_autograd(D, K, hidden, conj, hh):
    """host logic of the planes training path (layouts, block ranges, transposed images, head unfolding, scatter maps)
    with every entry point emulated from its documented semantics, against fp64 autograd through the oracle"""
    spec, sd = _planes_flow(D, K, hidden, conj, hh)
    flow = build_flow(spec, sd)
    eng = flow.engine()
    eng.use_planes, eng.planes_min_rows, eng.fused_min_rows, eng.train_planes_min_rows = True, 0, 0, 0
    g = torch.Generator().manual_seed(7)
    x = torch.rand(37, D, generator=g)
    g_lp = torch.randn(37, generator=g)
    path = TrainPath(flow)
    assert path.supported(x, None)
    from usflows_amd import training
    lp = training.log_prob_with_grad(path, x, None)
    plan = eng._plan("backward", 37, x.device, False, "nat", train=True)
    assert plan.get("planes_train"), "the planes training plan was not chosen"
    (lp * g_lp).sum().backward()
    lp_ref, g_ref = oracle_grads(spec, sd, x, g_lp)
    assert ((lp.detach().double() - lp_ref).abs() / lp_ref.abs()).max().item() < 2e-5
    checked = 0
    for pname, p in flow.named_parameters():
        if not p.requires_grad or pname not in g_ref:
            continue
        ref, got = g_ref[pname], p.grad
        if ref is None or ref.abs().max().item() == 0.0:
            assert got is None or got.abs().max().item() < 1e-6, pname
            continue
        assert got is not None, f"no gradient for {pname}"
        err = (got.double() - ref.reshape(got.shape)).abs().max().item()
        assert err <= 2e-4 * ref.abs().max().item(), (pname, err, ref.abs().max().item())
        checked += 1
    assert checked >= 5


@pytest.mark.parametrize("name", ["synth_d7_k3_hh0_laplace", "synth_d16_k4_hh2_conj_laplace", "synth_d16_k4_hh0_conj_radial1",
                                  "synth_d16_k3_convnet_gated_ln"])
def test_input_gradient_from_the_training_path(name):
    """an input that requires grad (round 5): the node also returns d log_prob / dx -- the first layer's data gradient, divided
    by the ScaleTransform's scale folded in front of it -- against autograd through the oracle; parameter gradients unchanged"""
    spec, sd, a = load_case(name)
    flow = build_flow(spec, sd)
    x = a["x"].clone().requires_grad_(True)
    g_lp = torch.randn(x.shape[0], generator=torch.Generator().manual_seed(2))
    path = TrainPath(flow)
    assert path.supported(x, None)
    from usflows_amd import training
    lp = training.log_prob_with_grad(path, x, None)
    (lp * g_lp).sum().backward()
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    x6 = a["x"].double().requires_grad_(True)
    lp6 = orc.flow_log_prob(sd64, spec, x6, None)
    (lp6 * g_lp.double()).sum().backward()
    assert x.grad is not None and x.grad.shape == x.shape
    s_ = x6.grad.abs().max().item()
    assert (x.grad.double() - x6.grad).abs().max().item() <= 2e-5 * s_, name
    # ... and once more (the recorded tape replays): same values
    x2 = a["x"].clone().requires_grad_(True)
    (training.log_prob_with_grad(path, x2, None) * g_lp).sum().backward()
    assert torch.allclose(x2.grad, x.grad, rtol=1e-6, atol=1e-7 * s_)
