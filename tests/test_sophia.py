"""SophiaG (usflows_amd/sophia.py; SURVEY row N2: the optimiser Flow.fit defaults to, flows.py:116) against golden
vectors of the REAL reference's class (tests/golden/make_golden_sophia.py): parameters, exp_avg and hessian after every
step, with and without maximize; on the CPU (torch-op arithmetic of the mirror) and on the MI355X (one
usf_sophiag_step_f32 / usf_sophiag_hessian_f32 launch per group).  Tolerance: 2e-6 relative plus 2.5e-7 of the tensor's
largest magnitude -- the reference's ATen kernels may or may not contract a*b+c, which moves a sum by an ulp of its
ADDENDS (under cancellation that is many ulps of the result); where the clamp min(|m| / (rho bs h + 1e-15), 1) switches,
both sides of the switch give the same value to that accuracy."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
GOLD = os.path.join(HERE, "golden", "sophia_steps.npz")
SHAPES = [(5, 7), (13,), (1,), (64, 33)]
STEPS, BS, HESS_AT = 6, 7, (0, 3)
HYPER = dict(lr=3e-3, betas=(0.9, 0.95), rho=0.05, weight_decay=0.2)


def _close(got, ref, what):
    got, ref = got.detach().cpu().double(), torch.from_numpy(ref).double()
    err = (got - ref).abs()
    assert bool((err <= 2e-6 * ref.abs() + 2.5e-7 * ref.abs().max() + 1e-9).all()), (what, err.max().item())


def _run(device, maximize):
    from usflows_amd.sophia import SophiaG
    z = np.load(GOLD)
    pre = f"max{int(maximize)}/"
    ps = [torch.nn.Parameter(torch.from_numpy(z[pre + f"p0/{i}"]).to(device)) for i in range(len(SHAPES))]
    opt = SophiaG(ps, maximize=maximize, **HYPER)
    for t in range(STEPS):
        for i, p in enumerate(ps):
            p.grad = torch.from_numpy(z[pre + f"g/{t}/{i}"]).to(device)
        if t in HESS_AT:
            opt.update_hessian()
        opt.step(bs=BS)
        for i, p in enumerate(ps):
            st = opt.state[p]
            _close(p, z[pre + f"p/{t}/{i}"], f"p step {t} tensor {i}")
            _close(st["exp_avg"], z[pre + f"m/{t}/{i}"], f"exp_avg step {t} tensor {i}")
            _close(st["hessian"], z[pre + f"h/{t}/{i}"], f"hessian step {t} tensor {i}")
            assert float(st["step"]) == t + 1
    return opt


@pytest.mark.parametrize("maximize", [False, True])
def test_sophiag_matches_reference_golden_cpu(maximize):
    _run("cpu", maximize)


@pytest.mark.gpu
@pytest.mark.parametrize("maximize", [False, True])
def test_sophiag_matches_reference_golden_gpu(maximize):
    from usflows_amd import _ext
    _ext.load()
    opt = _run("cuda:0", maximize)
    assert opt._tables, "the HIP multi-tensor kernels were not used"


@pytest.mark.gpu
def test_sophiag_table_follows_reallocated_gradients_and_mixed_tensors():
    """new gradient tensors every step (zero_grad(set_to_none=True)), one non-contiguous parameter (torch-op path) beside
    the HIP ones: the same numbers as the all-torch arithmetic on the CPU"""
    from usflows_amd.sophia import SophiaG
    g = torch.Generator().manual_seed(3)
    init = [torch.randn(40000, generator=g), torch.randn(6, 8, generator=g), torch.randn(9, 4, generator=g)]

    def make(device):
        ps = [torch.nn.Parameter(init[0].clone().to(device)), torch.nn.Parameter(init[1].clone().to(device)),
              torch.nn.Parameter(init[2].clone().to(device).t())]                     # a transposed (non-contiguous) leaf
        return ps, SophiaG(ps, lr=1e-2, weight_decay=0.0)

    (pc, oc), (pd, od) = make("cpu"), make("cuda:0")
    for t in range(4):
        grads = [torch.randn(p.shape, generator=g) for p in pc]
        for ps, dev in ((pc, "cpu"), (pd, "cuda:0")):
            for p, gr in zip(ps, grads):
                p.grad = gr.clone().to(dev)
        if t % 2 == 0:
            oc.update_hessian(); od.update_hessian()
        oc.step(bs=3); od.step(bs=3)
        for a, b in zip(pc, pd):
            assert torch.allclose(a.detach(), b.detach().cpu(), rtol=2e-6, atol=1e-6)
        oc.zero_grad(set_to_none=True); od.zero_grad(set_to_none=True)


def test_constructor_checks_and_state_dict_round_trip():
    from usflows_amd.sophia import SophiaG
    p = torch.nn.Parameter(torch.ones(3))
    for bad in (dict(lr=-1.0), dict(betas=(1.0, 0.9)), dict(betas=(0.9, 1.0)), dict(rho=-0.1), dict(weight_decay=-1.0)):
        with pytest.raises(ValueError):
            SophiaG([p], **bad)
    opt = SophiaG([p])
    assert opt.defaults["lr"] == 1e-4 and opt.defaults["betas"] == (0.965, 0.99) and opt.defaults["rho"] == 0.04 \
        and opt.defaults["weight_decay"] == 1e-1                                     # the reference's defaults (sophia.py:9-11)
    p.grad = torch.ones(3)
    opt.update_hessian()
    opt.step()
    sd = opt.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "hessian"}                     # the reference's state keys
    opt2 = SophiaG([p])
    opt2.load_state_dict(sd)
    assert torch.equal(opt2.state[p]["hessian"], opt.state[p]["hessian"])


@pytest.mark.skipif(not os.path.isdir("/root/reference/src/usflows"), reason="reference not present")
def test_sophiag_side_by_side_with_live_reference():
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import ref_shim
    ref_shim.install()
    from src.usflows.sophia import SophiaG as RefSophia
    from usflows_amd.sophia import SophiaG
    g = torch.Generator().manual_seed(9)
    init = [torch.randn(17, 5, generator=g), torch.randn(3, generator=g)]
    pr = [torch.nn.Parameter(t.clone()) for t in init]
    pm = [torch.nn.Parameter(t.clone()) for t in init]
    orf, om = RefSophia(pr, lr=1e-2), SophiaG(pm, lr=1e-2)
    for t in range(5):
        for a, b in zip(pr, pm):
            a.grad = torch.randn(a.shape, generator=g)
            b.grad = a.grad.clone()
        if t in (1, 2):
            orf.update_hessian(); om.update_hessian()
        orf.step(bs=2); om.step(bs=2)
        for a, b in zip(pr, pm):
            assert torch.equal(a, b)                                                # same ops in the same order on the CPU
            assert torch.equal(orf.state[a]["exp_avg"], om.state[b]["exp_avg"])
            assert torch.equal(orf.state[a]["hessian"], om.state[b]["hessian"])


def _fit_default(device):
    from golden_util import load_case
    from model_util import build_flow
    name = "synth_d7_k3_hh0_laplace"
    z = np.load(os.path.join(HERE, "golden", "fitsophia_" + name + ".npz"))
    spec, sd, _ = load_case(name)
    flow = build_flow(spec, sd)
    data = torch.from_numpy(z["data"])
    ds = torch.utils.data.TensorDataset(data, torch.zeros(data.shape[0]))
    np.random.seed(5)
    losses = flow.fit(ds, batch_size=32, shuffle=True, device=torch.device(device), epochs=2)       # default optimiser
    return flow, losses, z


def test_fit_default_optimiser_is_sophiag_and_matches_reference_run_cpu():
    """Flow.fit with NO optimiser argument = the reference's default (SophiaG, lr 1e-4, weight decay 0.1, Hessian estimate
    never updated: sign-momentum steps): per-epoch losses and every parameter of the real reference's run"""
    flow, losses, z = _fit_default("cpu")
    for a, b in zip(losses, z["losses"]):
        assert abs(a - b) <= 1e-5 * abs(b)
    sdm = flow.state_dict()
    n = 0
    for k in z.files:
        if k.startswith("sd/") and k[3:] in sdm and sdm[k[3:]].is_floating_point():
            assert torch.allclose(sdm[k[3:]].cpu(), torch.from_numpy(z[k]), rtol=1e-6, atol=1e-7), k
            n += 1
    assert n >= 20


@pytest.mark.gpu
def test_fit_default_optimiser_on_device_matches_reference_run():
    """the same run on the MI355X (HIP forward / backward / SophiaG step).  A step is lr * sign(momentum): an element whose
    momentum is within rounding of zero may step the other way (2 lr apart after that step), so: losses to 1e-4, at
    least 99.5 % of all parameter elements within 1e-5, none further than 6 steps' worth (2 lr each) away"""
    flow, losses, z = _fit_default("cuda:0")
    for a, b in zip(losses, z["losses"]):
        assert abs(a - b) <= 1e-4 * abs(b)
    sdm = flow.state_dict()
    tot = bad = 0
    for k in z.files:
        if k.startswith("sd/") and k[3:] in sdm and sdm[k[3:]].is_floating_point():
            d = (sdm[k[3:]].cpu() - torch.from_numpy(z[k])).abs()
            assert d.max().item() <= 6 * 2 * 1e-4 + 1e-5, k
            tot += d.numel()
            bad += int((d > 1e-5).sum())
    assert tot > 500 and bad <= 0.005 * tot, (bad, tot)
