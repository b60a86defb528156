"""Device training path on MI355X: Flow.log_prob under autograd = HIP forward + HIP backward
(usflows_amd/training.py through the C ABI), against autograd through the oracle's fp64 restatement of the
reference's log_prob (what Flow.fit differentiates, flows.py:196-199).  Gradient tolerance: 2e-4 of the largest
entry of each parameter's gradient (fp32 arithmetic through 2K+2 layers, forward and back)."""
import pytest
import torch

from golden_util import load_case
from model_util import build_flow
from oracle import usflows_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = ["synth_d7_k3_hh0_laplace", "synth_d16_k3_densenn_relu", "synth_d7_k3_hh1_conj_normal",
         "synth_d16_k4_hh2_conj_laplace", "synth_d33_k3_lu2_hh1", "synth_d7_k3_soft_ctx", "synth_d64_k6_hh0_laplace",
         "synth_d64_k4_hh1_conj_laplace", "init_d2_k4_hh0_laplace", "synth_d16_k3_hh0_radialinf",
         "synth_d16_k3_hh1_radial2", "synth_d16_k4_hh0_conj_radial1"]


def oracle_grads(spec, sd, x, g_lp, context=None):
    sd64 = {k: (v.double().clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    if spec.base == "radial" and "base_distribution.loc" in sd64:
        import copy
        spec = copy.copy(spec)
        spec.base_loc = sd64["base_distribution.loc"]          # trainable loc of RadialDistribution
    lp = orc.flow_log_prob(sd64, spec, x.double(), context.double() if context is not None else None)
    (lp * g_lp.double()).sum().backward()
    return lp.detach(), {k: v.grad for k, v in sd64.items() if torch.is_tensor(v) and v.is_floating_point()}


def _compare(flow, g_ref, tol=2e-4, kink_frac=0.0):
    """kink_frac > 0: the conditioners are piece-wise linear, so a hidden unit whose pre-activation is within fp32
    noise of zero takes the other LeakyReLU branch than in the fp64 oracle and changes one row of a gradient by
    O(1) of that sample's contribution -- measure-zero events that show up at 10^5..10^6 hidden units.  Then at most
    that fraction of a tensor's entries may miss `tol` (and none by more than 5 % of the tensor's largest entry)."""
    n = 0
    for pname, p in flow.named_parameters():
        if not p.requires_grad or "norm_distribution" in pname:
            continue            # (radial norm-distribution parameters: constants in the oracle, torch autograd here)
        ref = g_ref.get(pname)
        if ref is None or ref.abs().max().item() == 0.0:
            assert p.grad is None or p.grad.abs().max().item() < 1e-6, pname
            continue
        assert p.grad is not None, f"no gradient for {pname}"
        diff = (p.grad.cpu().double() - ref.reshape(p.shape)).abs()
        big = ref.abs().max().item()
        if kink_frac > 0.0:
            n_bad = int((diff > tol * big).sum().item())
            assert n_bad <= max(2, int(kink_frac * diff.numel())), (pname, n_bad, diff.numel())
            assert diff.max().item() <= 0.05 * big, (pname, diff.max().item(), big)
        else:
            assert diff.max().item() <= tol * big, (pname, diff.max().item(), big)
        n += 1
    return n


@pytest.mark.parametrize("mode", ["bf16x3", "f32"])
@pytest.mark.parametrize("name", CASES)
def test_log_prob_backward_matches_oracle_autograd(name, mode):
    spec, sd, a = load_case(name)
    flow = build_flow(spec, sd, device=DEV)
    flow.engine().gemm_mode = mode
    x = a["x"].to(DEV)
    ctx = a.get("context")
    g_lp = torch.randn(x.shape[0], generator=torch.Generator().manual_seed(1))
    before = flow.engine().launch_count
    lp = flow.log_prob(x, ctx.to(DEV) if ctx is not None else None)
    assert lp.requires_grad and flow.engine().launch_count > before, "device training path did not run"
    (lp * g_lp.to(DEV)).sum().backward()
    ctx_ref = ctx
    if ctx_ref is None and spec.soft_training:
        ctx_ref = torch.zeros(x.shape[0], 1)
    lp_ref, g_ref = oracle_grads(spec, sd, a["x"], g_lp, ctx_ref)
    assert ((lp.detach().cpu().double() - lp_ref).abs() / lp_ref.abs()).max().item() < 1e-5
    assert _compare(flow, g_ref) >= 5


def test_same_gradients_as_the_composite_formulation():
    """device backward == torch autograd through the composite layer loop on the same device (USFLOWS_AMD_TRAIN=composite)"""
    spec, sd, a = load_case("synth_d64_k4_hh1_conj_laplace")
    x = a["x"].to(DEV)
    grads = []
    for device_training in (True, False):
        flow = build_flow(spec, sd, device=DEV)
        flow.use_device_training = device_training
        (-flow.log_prob(x).mean()).backward()
        grads.append({n: p.grad.clone() for n, p in flow.named_parameters() if p.grad is not None})
    assert set(grads[0]) == set(grads[1])
    for n in grads[0]:
        ref = grads[1][n]
        assert (grads[0][n] - ref).abs().max().item() <= 5e-4 * max(ref.abs().max().item(), 1e-9), n


@pytest.mark.parametrize("conj", [False, True])
def test_cfg2_training_step_ragged_batch(conj):
    """BASELINE cfg2 model (and its affine_conjugation=True variant, what the reference's live configs use), ragged
    batch: gradients of -mean log_prob vs the oracle on a 48-row batch; a second forward before the backward
    (activation buffers overwritten) still gives the right gradients"""
    spec, sd, a = load_case("synth_d784_k32_cfg2")
    if conj:
        import copy
        from usflows_amd.synth import synth_state_dict
        spec = copy.copy(spec)
        spec.affine_conjugation = True           # (the layer numbering of the state dict changes with the variant)
        spec.coupling_blocks = 8                 # 17 affine applications; keeps the fp64 oracle's autograd ~15 s
        sd = synth_state_dict(spec, seed=100, alpha=0.1)
    flow = build_flow(spec, sd, device=DEV)
    x = a["x"][:48]
    if conj:
        # a LeakyReLU kink flip (fp32 vs fp64 pre-activation of opposite sign) changes that sample's gradient row in every
        # layer behind it; with 65 affine layers that shows in thousands of entries.  This batch has none
        # (tools/grad_noise_probe.py --conj: max deviation 7e-6 of the largest entry on the device path).
        x = torch.rand(48, 784, generator=torch.Generator().manual_seed(1))
    xd = x.to(DEV)
    lp1 = flow.log_prob(xd)
    _ = flow.log_prob(x.flip(0).to(DEV))                 # same batch size: reuses (overwrites) the saved activations
    (-lp1.mean()).backward()
    lp_ref, g_ref = oracle_grads(spec, sd, x, torch.full((48,), -1.0 / 48))
    assert ((lp1.detach().cpu().double() - lp_ref).abs() / lp_ref.abs()).max().item() < 1e-5
    assert _compare(flow, g_ref, tol=5e-4, kink_frac=5e-3) >= (25 if conj else 100)


def test_fit_runs_on_device_and_reduces_the_loss():
    spec, sd, a = load_case("synth_d16_k3_densenn_relu")
    flow = build_flow(spec, sd, device=DEV)
    g = torch.Generator().manual_seed(0)
    data = torch.rand(512, 16, generator=g)

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return data.shape[0]

        def __getitem__(self, i):
            return data[i], 0

    before = flow.engine().launch_count
    losses = flow.fit(DS(), optim=torch.optim.Adam, optim_params=dict(lr=1e-3), batch_size=128, shuffle=False,
                      device=torch.device(DEV), epochs=4)
    assert flow.engine().launch_count > before
    assert losses[-1] < losses[0]


@pytest.mark.parametrize("name", ["synth_d64_k6_hh0_laplace", "synth_d64_k4_hh1_conj_laplace", "synth_d7_k3_soft_ctx"])
def test_replayed_tapes_track_parameter_updates(name):
    """steps 2.. of a training loop replay the recorded launch tapes (pack refresh in place + backward): gradients
    must keep matching the composite formulation while the optimiser moves the parameters"""
    spec, sd, a = load_case(name)
    flow = build_flow(spec, sd, device=DEV)
    ref = build_flow(spec, sd, device=DEV)
    ref.use_device_training = False
    opts = [torch.optim.SGD(f.parameters(), lr=1e-4) for f in (flow, ref)]
    g = torch.Generator().manual_seed(3)
    packs = []
    for step in range(4):
        x = torch.rand(96, spec.dim, generator=g).to(DEV)
        ctx = (torch.rand(96, 1, generator=g) * 2).to(DEV) if spec.soft_training else None
        for f, o in zip((flow, ref), opts):
            o.zero_grad(set_to_none=True)
            (-f.log_prob(x, ctx).mean()).backward()
        for (n1, p1), (n2, p2) in zip(flow.named_parameters(), ref.named_parameters()):
            if p2.grad is None:
                assert p1.grad is None or p1.grad.abs().max().item() < 1e-6, n1
                continue
            big = max(p2.grad.abs().max().item(), 1e-9)
            assert (p1.grad - p2.grad).abs().max().item() <= 5e-4 * big, (step, n1)
        packs.append(flow.engine()._pack)
        for o in opts:
            o.step()
    assert all(p is packs[0] for p in packs), "the pack was rebuilt instead of refreshed in place"
    plan = next(p for p in flow.engine()._plans.values() if p.get("bwd_tape") is not None)      # the training plan
    assert len(plan["bwd_tape"].entries) > 10


def test_single_row_batch_under_autograd():
    """(an empty batch under autograd takes the composite loop, which fails inside torch's Independent.log_prob exactly
    as the reference does; without grad it returns an empty tensor, tests/test_flow_gpu.py)"""
    spec, sd, a = load_case("synth_d16_k3_densenn_relu")
    flow = build_flow(spec, sd, device=DEV)
    lp1 = flow.log_prob(a["x"][:1].to(DEV))
    lp1.sum().backward()
    lp_ref, g_ref = oracle_grads(spec, sd, a["x"][:1], torch.ones(1))
    assert abs(lp1.item() - lp_ref.item()) < 1e-5 * abs(lp_ref.item())
    assert _compare(flow, g_ref) >= 5


from golden_util import grad_case_names, load_grads  # noqa: E402


@pytest.mark.parametrize("name", grad_case_names())
def test_device_gradients_match_the_real_reference(name):
    """Flow.fit's loss differentiated on the MI355X (HIP forward + HIP backward) against the gradients the REAL
    reference computed in fp64 (tests/golden/grads_*.npz, made by tests/golden/make_golden_grads.py)"""
    spec, sd, a = load_case(name)
    loss_ref, g_ref = load_grads(name)
    flow = build_flow(spec, sd, device=DEV)
    ctx = a.get("context")
    before = flow.engine().launch_count
    loss = -flow.log_prob(a["x"].to(DEV), ctx.to(DEV) if ctx is not None else None).mean()
    loss.backward()
    assert flow.engine().launch_count > before
    # (the device training path took the call -- not the torch composite formulation: vector ConvNet conditioners included)
    assert flow._train_obj is not None and not getattr(flow, "_train_failed", False)
    assert abs(loss.item() - loss_ref) <= 1e-5 * abs(loss_ref)
    checked = 0
    for pname, p in flow.named_parameters():
        if pname not in g_ref:
            assert p.grad is None or p.grad.abs().max().item() < 1e-6, pname
            continue
        ref = g_ref[pname].reshape(p.shape)
        assert p.grad is not None, pname
        assert (p.grad.cpu().double() - ref).abs().max().item() <= 2e-4 * max(ref.abs().max().item(), 1e-9), pname
        checked += 1
    assert checked >= 20


@pytest.mark.parametrize("name", ["synth_d7_k3_hh1_conj_normal", "synth_d16_k4_hh0_conj_radial1", "synth_d64_k6_hh0_laplace",
                                  "synth_d16_k3_convnet_gated_ln"])
def test_input_gradient_on_the_device_path(name):
    """``Flow.log_prob`` of an input that requires grad (it used to take the torch composite formulation):
    the device training path returns d log_prob / dx as well -- against autograd through the fp64 oracle -- and the parameters'
    gradients are the ones of a call whose input does not require grad"""
    from oracle import usflows_oracle as orc_
    spec, sd, a = load_case(name)
    flow = build_flow(spec, sd, device=DEV)
    g = torch.Generator().manual_seed(3)
    w = torch.randn(a["x"].shape[0], generator=g)
    x = a["x"].to(DEV).requires_grad_(True)
    before = flow.engine().launch_count
    lp = flow.log_prob(x)
    (lp * w.to(DEV)).sum().backward()
    assert flow.engine().launch_count > before and flow._train_obj is not None and not getattr(flow, "_train_failed", False)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    x6 = a["x"].double().requires_grad_(True)
    lp6 = orc_.flow_log_prob(sd64, spec, x6, None)
    (lp6 * w.double()).sum().backward()
    assert ((lp.detach().cpu().double() - lp6.detach()).abs() / lp6.detach().abs()).max().item() < 2e-5
    s_ = x6.grad.abs().max().item()
    assert (x.grad.cpu().double() - x6.grad).abs().max().item() <= 5e-5 * s_
    grads_x = {k: p.grad.clone() for k, p in flow.named_parameters() if p.grad is not None}
    for p in flow.parameters():
        p.grad = None
    (flow.log_prob(a["x"].to(DEV)) * w.to(DEV)).sum().backward()
    for k, p in flow.named_parameters():
        if p.grad is not None:
            assert torch.allclose(grads_x[k], p.grad, rtol=1e-4, atol=1e-5 * float(p.grad.abs().max()) + 1e-12), k


def test_fit_on_device_matches_reference_run():
    """Flow.fit on the MI355X (HIP forward + backward, launch tapes replayed from step 2, pack refreshed in place
    after every SGD step) against the golden run of the REAL reference's Flow.fit: per-epoch losses and every
    parameter after the 6 steps"""
    from golden_util import fit_case_names, load_fit
    from test_modules_cpu import _check_fit, _run_fit
    for name in fit_case_names():
        spec, sd, _ = load_case(name)
        data, losses_ref, sd_ref, prior_scale = load_fit(name)
        if prior_scale is not None:
            spec.extra["prior_scale"] = prior_scale      # the loss then carries -log_prior() (transforms.py:1371-1379)
        flow = build_flow(spec, sd, device=DEV)
        before = flow.engine().launch_count
        losses = _run_fit(flow, data, DEV)
        assert flow.engine().launch_count > before and not getattr(flow, "_train_failed", False)
        _check_fit(flow, losses, losses_ref, sd_ref, 5e-5)


def test_failed_capture_of_a_training_step_is_recoverable(monkeypatch):
    """A training step whose hipGraph capture breaks off (an op that synchronises the host) must leave the flow usable:
    the capture ran no GPU work but its Python side moved version counters and the engine's pack key -- Flow.fit drops
    every cache keyed on them and continues with eager steps whose losses equal those of a run that never tried to
    capture (the recovery path was untested)."""
    import numpy as np
    from usflows_amd.flows import Flow
    from oracle import usflows_oracle as orc
    spec = orc.FlowSpec(12, 3, [16, 16], householder=1, affine_conjugation=True)
    sd = orc.synth_state_dict(spec, seed=21)
    x = torch.rand(32 * 8, 12, generator=torch.Generator().manual_seed(3))

    class DS:
        def __len__(self):
            return x.shape[0]

        def __getitem__(self, i):
            return (x[i],)

    def run(break_capture):
        flow = build_flow(spec, sd, device=DEV)
        if break_capture:
            cls = type(flow)                            # (USFlow overrides Flow.log_prior)
            real = cls.log_prior

            def syncing_prior(self):
                if torch.cuda.is_current_stream_capturing():
                    torch.cuda.synchronize()            # not permitted while capturing: the capture fails here
                return real(self)
            monkeypatch.setattr(cls, "log_prior", syncing_prior)
        else:
            monkeypatch.setenv("USFLOWS_AMD_TRAIN_GRAPH", "0")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            losses = flow.fit(DS(), torch.optim.SGD, dict(lr=1e-3), batch_size=32, shuffle=False, device=torch.device(DEV), epochs=2)
        monkeypatch.undo()
        with torch.no_grad():
            lp = flow.log_prob(x[:64].to(DEV)).cpu()
        return losses, lp, getattr(flow, "_train_graph_failed", False)

    import warnings
    l_ref, lp_ref, _ = run(False)
    l_brk, lp_brk, failed = run(True)
    assert failed, "the capture was expected to fail"
    # torch's generator of the device is usable again (a capture that breaks off leaves it in capture mode)
    assert torch.isfinite(torch.normal(torch.zeros(8, device=DEV), torch.ones(8, device=DEV))).all()
    assert np.allclose(l_brk, l_ref, rtol=1e-6, atol=0), (l_brk, l_ref)
    assert torch.allclose(lp_brk, lp_ref, rtol=1e-6, atol=1e-5)


@pytest.mark.parametrize("variant", ["plain", "conj8", "ctx", "hh8"])
def test_large_batch_training_paths_equal_the_round_3_path(monkeypatch, variant):
    """From 8192 rows the affine layers' GEMMs leave the bf16 planes of their operands (usf_linear_desc::A_planes_out) and the
    weight gradients multiply those (usf_wgrad_planes_f32); bias gradients ride in the weight-gradient passes; from 1024 rows
    the fused coupling kernel stores its hidden activations (hidden_out) and the conditioner's data-gradient chain is ONE
    launch of the same kernel run backwards (USF_ACT_GATE).  Same products, other orders of summation: every parameter
    gradient within 1e-4 of its largest entry of the path with all four switched off, which the tests above pin against the oracle and the
    reference's goldens -- the operand planes and the saved activations also on their own."""
    from usflows_amd import _ext
    monkeypatch.setenv("USFLOWS_AMD_TRAIN_PLANES", "0")       # (this test: the fp32-row path, which still serves context / f32 mode / wide conditioners)
    spec, sd, _a = load_case("synth_d784_k32_cfg2")
    n_cpl = 32
    if variant in ("conj8", "hh8"):                             # affine_conjugation: every block also in its M form (what the live configs
        import copy                                             # use); hh8: Sequential([LU, Householder]) blocks, the constructor's default
        from usflows_amd.synth import synth_state_dict
        spec = copy.copy(spec)
        spec.coupling_blocks, n_cpl = 8, 8
        if variant == "conj8":
            spec.affine_conjugation = True
        else:
            spec.householder = 1
        sd = synth_state_dict(spec, seed=100, alpha=0.1)
    x = torch.rand(16400, 784, generator=torch.Generator().manual_seed(3)).to(DEV)       # ragged (16400 = 512 x 32 + 16), above the fused coupling kernel's cross-over
    ctx = torch.rand(16400, 1, generator=torch.Generator().manual_seed(4)).to(DEV) if variant == "ctx" else None   # soft-training context branch
    calls = []
    real = _ext.wgrad_planes
    monkeypatch.setattr(_ext, "wgrad_planes", lambda *a, **k: (calls.append(k["N"]), real(*a, **k))[1])
    real_c = _ext.coupling_op
    monkeypatch.setattr(_ext, "coupling_op", lambda *a, **k: (calls.append("cbwd"), real_c(*a, **k))[1])
    from usflows_amd.config import config
    switches = ("wgrad_planes", "fused_bias", "save_hidden", "fused_cbwd")          # (usflows_amd.config's knobs of the fp32-row path)

    def grads_with(on):
        for sw in switches:
            monkeypatch.setattr(config, sw, sw in on)
        flow = build_flow(spec, sd, device=DEV)
        n0 = len(calls)
        for _ in range(2):                                      # the second pass replays the recorded launches
            for p in flow.parameters():
                p.grad = None
            lp = flow.log_prob(x, ctx)
            (-lp.mean()).backward()
        torch.cuda.synchronize()
        return {n: p.grad.clone() for n, p in flow.named_parameters() if p.grad is not None}, calls[n0:], lp.detach()

    base, c0, lp0 = grads_with(())
    assert not c0 and len(base) > (100 if variant in ("plain", "ctx") else 25)
    for on in ((switches[:1], switches[2:3], switches) if variant == "plain" else (switches,)):      # (alone: the planes, the saved activations)
        got, c, lp = grads_with(on)
        assert (c.count(784) >= (31 if variant in ("plain", "ctx") else 7)) == (switches[0] in on)   # every affine layer behind the first one
        assert (c.count("cbwd") == n_cpl) == (switches[3] in on)                              # one backward launch per coupling layer
        assert torch.equal(lp, lp0)                             # the forward values do not depend on any of them
        assert got.keys() == base.keys()
        for n in base:
            # (the saved activations are the fused forward kernel's own, the recomputed ones come from the unfused GEMMs: a
            # hidden unit within fp32 noise of zero can take the other LeakyReLU branch -- of 2.7e8 units a handful do, each
            # changing one sample's contribution to a few entries; see _compare's kink_frac)
            big = base[n].abs().max().item()
            diff = (base[n] - got[n]).abs()
            n_bad = int((diff > 1e-4 * big + 1e-12).sum().item())       # (other orders of summation over 16 400 rows: fp32 noise of cancelling sums)
            assert n_bad <= max(2, int(1e-3 * diff.numel())), (on, n, n_bad, diff.numel())
            assert diff.max().item() <= 1e-3 * big + 1e-12, (on, n, diff.max().item(), big)


def test_fit_hands_large_host_batches_over_under_the_running_step(monkeypatch):
    """Flow.fit on a HOST data set with batches of >= 16 MB: the next batch is staged in pinned memory and uploaded on a copy
    stream while the current step runs (flows._BatchFeed).  Same batches in the same order: the loss curve equals, value for
    value, the one of the reference's hand-over (a pageable .to(device) in front of every step), ragged last batch included."""
    import numpy as np
    from usflows_amd import flows as F_
    from usflows_amd.sophia import SophiaG
    spec, sd, _a = load_case("synth_d64_k6_hh0_laplace")
    B, N = 70000, 3 * 70000 + 1234                              # 70000 x 64 x 4 = 17.9 MB per batch
    data = torch.rand(N, 64, generator=torch.Generator().manual_seed(11))

    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return N

        def __getitem__(self, i):
            return data[i], 0

    made = []
    real = F_._BatchFeed.make
    monkeypatch.setattr(F_._BatchFeed, "make", staticmethod(lambda *a: (made.append(real(*a)), made[-1])[1]))
    curves = []
    for prefetch in ("0", "1"):
        from usflows_amd.config import config
        monkeypatch.setattr(config, "fit_prefetch", prefetch == "1")
        flow = build_flow(spec, sd, device=DEV)
        np.random.seed(5)
        curves.append(flow.fit(DS(), optim=SophiaG, optim_params=dict(lr=1e-4), batch_size=B, shuffle=True,
                               device=torch.device(DEV), epochs=2))
        torch.cuda.synchronize()
    assert made[0] is None and made[1] is None and made[2] is not None and made[3] is not None     # off twice, on twice (one per epoch)
    assert made[2] is made[3]                                   # ... ONE feed (buffers, copy stream, events) for both epochs
    assert curves[0] == curves[1] and len(curves[0]) == 2


@pytest.mark.parametrize("hidden,soft", [([256], False), ([256, 192, 256], False), ([200, 256], True)])
def test_fused_coupling_backward_for_one_to_three_hidden_layers(monkeypatch, hidden, soft):
    """the conditioner's backward pass on the fused kernel (USF_ACT_GATE) for 1 / 2 / 3 hidden layers, ragged widths and the
    soft-training context branch: gradients of -mean log_prob against the fp64 oracle's autograd, and against the path with
    the saved activations and the fused backward switched off"""
    from oracle import usflows_oracle as orc_
    from usflows_amd import _ext
    spec = orc_.FlowSpec(72, 2, hidden, householder=0, soft_training=soft)
    sd = orc_.synth_state_dict(spec, seed=21)
    B = 2048
    x = torch.rand(B, 72, generator=torch.Generator().manual_seed(6))
    ctx = torch.rand(B, 1, generator=torch.Generator().manual_seed(7)) if soft else None
    n_launch = []
    real_c = _ext.coupling_op
    monkeypatch.setattr(_ext, "coupling_op", lambda *a, **k: (n_launch.append(1), real_c(*a, **k))[1])
    grads = []
    for on in ("1", "0"):
        from usflows_amd.config import config
        monkeypatch.setattr(config, "save_hidden", on == "1")
        monkeypatch.setattr(config, "fused_cbwd", on == "1")
        flow = build_flow(spec, sd, device=DEV)
        flow.engine().fused_min_rows = 0                        # the fused kernel from 1024 rows on
        n0 = len(n_launch)
        lp = flow.log_prob(x.to(DEV), ctx.to(DEV) if soft else None)
        (-lp.mean()).backward()
        torch.cuda.synchronize()
        assert (len(n_launch) - n0 == 2) == (on == "1")         # one backward launch per coupling layer
        grads.append({n: p.grad.clone() for n, p in flow.named_parameters() if p.grad is not None})
        if on == "1":
            lp_ref, g_ref = oracle_grads(spec, sd, x, torch.full((B,), -1.0 / B), ctx)
            assert ((lp.detach().cpu().double() - lp_ref).abs() / lp_ref.abs()).max().item() < 1e-5
            assert _compare(flow, g_ref, tol=5e-4, kink_frac=5e-3) >= 5
    assert grads[0].keys() == grads[1].keys()
    for n in grads[0]:
        big = grads[1][n].abs().max().item()
        diff = (grads[0][n] - grads[1][n]).abs()
        assert int((diff > 1e-4 * big + 1e-12).sum().item()) <= max(2, int(1e-3 * diff.numel())), n


# ---- round 5: the training step on the planes pipeline -----------------------------------------------------------------------
def _cfg2_like(blocks, seed=5, **kw):
    import copy
    from usflows_amd.synth import synth_state_dict
    spec, _sd, _a = load_case("synth_d784_k32_cfg2")
    spec = copy.copy(spec)
    spec.coupling_blocks = blocks
    for k_, v_ in kw.items():
        setattr(spec, k_, v_)
    return spec, synth_state_dict(spec, seed=seed, alpha=0.1)


def _compare_large(flow, g_ref, linear):
    """every parameter gradient against the oracle's.  linear (negative_slope = 1 AND a Normal base: no kink anywhere in
    log_prob): 2e-4 of each tensor's largest entry, no exception.  Otherwise a hidden unit whose pre-activation -- or a latent
    coordinate under the Laplace base, d|z|/dz = sign z -- is within fp32 noise of zero sits on the other branch than in fp64:
    ~1e-6 of the units, i.e. a few per layer at these batches (tools/train_planes_debug.py names them: at 8192 rows ONE latent
    coordinate, sample 3859 / feature 614, carries the whole deviation of that run), and moves that sample's contribution
    (~1 / sqrt(B) of an entry's sum) in a whole row of the layer's weight gradient and, diluted, in the layers in front:
    entries may miss 5e-4 in up to 2 % of a tensor, none by more than 5 % of its largest entry, and the tensor as a whole
    agrees to 3e-3 in the Frobenius norm (a wrong index map / sign / missing term gives O(1) there)."""
    n = 0
    for pname, p in flow.named_parameters():
        ref = g_ref.get(pname)
        if not p.requires_grad or ref is None or ref.abs().max().item() == 0.0:
            continue
        assert p.grad is not None, f"no gradient for {pname}"
        got = p.grad.cpu().double()
        diff = (got - ref.reshape(p.shape)).abs()
        big = ref.abs().max().item()
        if linear:
            assert diff.max().item() <= 2e-4 * big, (pname, diff.max().item(), big)
        else:
            n_bad = int((diff > 5e-4 * big).sum().item())
            assert n_bad <= max(2, int(2e-2 * diff.numel())), (pname, n_bad, diff.numel())
            assert diff.max().item() <= 0.05 * big, (pname, diff.max().item(), big)
            assert (diff.norm() / ref.norm()).item() <= 3e-3, (pname, (diff.norm() / ref.norm()).item())
        n += 1
    return n


_SMOOTH = dict(negative_slope=1.0, base="normal")


@pytest.mark.parametrize("B,force,kw", [(8192, True, {}), (8192, True, _SMOOTH), (16400, False, _SMOOTH),
                                        (8200, True, dict(affine_conjugation=True, householder=1, **_SMOOTH)),
                                        (8192, True, dict(hidden_dims=[200]))])
def test_planes_training_backward_matches_oracle_autograd_at_d784(B, force, kw):
    """D = 784, K = 4 (the BASELINE cfg2 layer shapes), >= 8192 rows: ONE backward pass of the planes training path
    (usf_gemm_planes_bf16x3 / usf_coupling_planes gate mode / usf_wgrad_blocked_f32) held DIRECTLY against fp64 autograd through
    the oracle's restatement of Flow.log_prob (flows.py:196-203) -- no link through another device path.  Also with
    affine_conjugation + Householder blocks (the constructor's defaults / the live configurations) and a one-layer conditioner;
    with negative_slope = 1 and a Normal base (no kink anywhere: every launch's linear algebra, index map and the chain rule at
    the tight tolerance) and with the BASELINE's LeakyReLU(0.01) + Laplace base (the gates; tolerance for branch flips at fp32
    noise, see _compare_large)."""
    spec, sd = _cfg2_like(4, **kw)
    flow = build_flow(spec, sd, device=DEV)
    eng = flow.engine()
    if force:
        eng.train_planes_min_rows = eng.fused_min_rows = 0
    g = torch.Generator().manual_seed(B)
    x = torch.rand(B, 784, generator=g)
    # weights of one sign, as in Flow.fit's loss (-mean log_prob), but different from row to row: random signs would make every
    # gradient entry a cancelling sum whose fp32 noise is sqrt(B) times larger relative to the entry
    g_lp = -(0.5 + torch.rand(B, generator=g)) / B
    before = eng.launch_count
    lp = flow.log_prob(x.to(DEV))
    assert lp.requires_grad and eng.launch_count > before
    plan = eng._plan("backward", B, torch.device(DEV), False, "nat", train=True)
    assert plan.get("planes_train"), "the planes training plan was not chosen"
    (lp * g_lp.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    lp_ref, g_ref = oracle_grads(spec, sd, x, g_lp)
    assert ((lp.detach().cpu().double() - lp_ref).abs() / lp_ref.abs()).max().item() < 1e-5
    assert _compare_large(flow, g_ref, linear=kw.get("negative_slope") == 1.0 and kw.get("base") == "normal") >= 20
    # the replayed pass (recorded launches) gives the same bits
    first = {n: p.grad.clone() for n, p in flow.named_parameters() if p.grad is not None}
    for p in flow.parameters():
        p.grad = None
    lp2 = flow.log_prob(x.to(DEV))
    (lp2 * g_lp.to(DEV)).sum().backward()
    torch.cuda.synchronize()
    assert torch.equal(lp2.detach(), lp.detach())
    for n, p in flow.named_parameters():
        if n in first:
            assert torch.equal(p.grad, first[n]), n


def test_planes_training_path_equals_the_fp32_row_path(monkeypatch):
    """the planes training path against the round-4 path (pinned above and in the tests before) on the full 32-block cfg2
    model at 16400 rows: the same gradients up to summation order / a handful of LeakyReLU kink flips"""
    spec, sd, _a = load_case("synth_d784_k32_cfg2")
    x = torch.rand(16400, 784, generator=torch.Generator().manual_seed(3)).to(DEV)
    res = []
    for on in ("0", "1"):
        monkeypatch.setenv("USFLOWS_AMD_TRAIN_PLANES", on)
        flow = build_flow(spec, sd, device=DEV)
        lp = flow.log_prob(x)
        plan = flow.engine()._plan("backward", 16400, torch.device(DEV), False, "nat", train=True)
        assert bool(plan.get("planes_train")) == (on == "1")
        (-lp.mean()).backward()
        torch.cuda.synchronize()
        res.append(({n: p.grad.clone() for n, p in flow.named_parameters() if p.grad is not None}, lp.detach()))
    (g0, lp0), (g1, lp1) = res
    assert ((lp0 - lp1).abs() / lp0.abs()).max().item() < 2e-6
    assert g0.keys() == g1.keys() and len(g0) > 100
    for n in g0:
        big = g0[n].abs().max().item()
        diff = (g0[n] - g1[n]).abs()
        n_bad = int((diff > 1e-4 * big + 1e-12).sum().item())
        assert n_bad <= max(2, int(1e-3 * diff.numel())), (n, n_bad, diff.numel())
        assert diff.max().item() <= 1e-3 * big + 1e-12, (n, diff.max().item(), big)
