"""The reference's LIVE image configurations on the device: image-shaped flow x ``RadialDistribution`` base with an
image-shaped loc x ``prior_scale`` (experiments/mnist/mnist.yaml:30-92, fashion/fashionclasses_veriflow.yaml:55-93,
cifar/cifar.yaml) -- ``usf_radial_logprob_f32`` / ``usf_radial_logprob_grad_f32`` (usflows_amd/radial.py) against fp64 torch
formulations of the reference's arithmetic (distributions.py:501-549) and against golden vectors of the REAL reference
(tests/golden/imageradial_*.npz, imageradialfit_*.npz; made by tests/golden/make_golden_image_radial.py)."""
import json
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from golden_util import (grads_close, GOLDEN_DIR, image_radial_case_names, image_radial_fit_case_names, load_image_radial_case,
                         load_image_radial_fit)

DEV = "cuda:0"


def _close(got, want, tol=1e-5, what=""):
    want = want.double().cpu()
    got = got.double().cpu()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    s = max(want.abs().max().item(), 1e-30)
    err = (got - want).abs().max().item()
    assert err <= tol * s, f"{what}: max abs err {err:.3e} vs scale {s:.3e} (rel {err / s:.2e})"


def _rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).abs() / b.double().cpu().abs().clamp_min(1e-30)).max().item()


# ---- fp64 torch statement of the density (what the kernels are held against) -----------------------------------------------
def ref_radial_logprob(z, loc, p, kind, a, b, logits, raw=False):
    """RadialDistribution.log_prob (distributions.py:501-549) with the norm distribution a mixture of LogNormal / Gamma
    components (torch's log_prob formulas), every tensor fp64"""
    x = (z - loc).flatten(1)
    D = x.shape[1]
    # (p = inf: the reference's ``x.norm(p=inf)`` -- distributions.py:506 -- whose backward shares the gradient among tied maxima)
    r = x.abs().sum(-1) if p == 1 else (x * x).sum(-1).sqrt() if p == 2 else torch.linalg.vector_norm(x, ord=math.inf, dim=-1)
    lr = torch.log(r).unsqueeze(-1)
    if kind == "lognormal":
        mu, sigma = a, (b if raw else F.softplus(b))
        comp = -((lr - mu) ** 2) / (2 * sigma ** 2) - sigma.log() - math.log(math.sqrt(2 * math.pi)) - lr
    else:
        c, rate = (a if raw else F.softplus(a)), (b if raw else F.softplus(b))
        comp = c * torch.log(rate) + (c - 1) * lr - rate * r.unsqueeze(-1) - torch.lgamma(c)
    if logits is not None:
        comp = comp + torch.log_softmax(logits, -1)
    lpn = torch.logsumexp(comp, -1)
    if p == 1:
        cst = math.log(2) * D - math.lgamma(D)
    elif p == 2:
        cst = math.log(D) + (D / 2) * math.log(math.pi) - math.lgamma(D / 2 + 1)
    else:
        cst = math.log(D) + D * math.log(2)
    return lpn - (cst + (D - 1) * torch.log(r)), r


def _norm_params(kind, K, g):
    """stored parameters (unconstrained where the modules store them so) giving radii of a few hundred a sensible density"""
    if kind == "lognormal":
        a = 5.5 + torch.rand(K, generator=g)
        b = torch.log(torch.expm1(0.2 + 0.3 * torch.rand(K, generator=g)))
    else:
        a = torch.log(torch.expm1(5 + 70 * torch.rand(K, generator=g)))
        b = torch.log(torch.expm1(0.05 + 0.3 * torch.rand(K, generator=g)))
    logits = torch.randn(K, generator=g) if K > 1 else None
    return a, b, logits


KERNEL_CASES = [  # (event shape, p, kind, K)
    ((16, 7, 7), 1.0, "lognormal", 1), ((16, 7, 7), 1.0, "gamma", 20), ((16, 7, 7), 2.0, "gamma", 1),
    ((16, 7, 7), math.inf, "lognormal", 5), ((48, 8, 8), 1.0, "lognormal", 1), ((48, 8, 8), 2.0, "gamma", 64),
    ((7,), 1.0, "gamma", 3), ((33,), 2.0, "lognormal", 1), ((3, 5, 2), math.inf, "gamma", 2), ((784,), 1.0, "lognormal", 1),
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", KERNEL_CASES, ids=lambda c: f"{'x'.join(map(str, c[0]))}-p{c[1]}-{c[2]}{c[3]}")
@pytest.mark.parametrize("B", [1, 5, 300])
def test_radial_kernels_vs_fp64(case, B):
    """forward and backward kernels (through the autograd function the flow uses) against fp64 autograd of the reference's
    formulas: log-density, d/dz, d/dloc, d/d(norm parameters), d/d(mixture logits); bit-reproducible"""
    from usflows_amd import _ext, radial
    ev, p, kind, K = case
    D = math.prod(ev)
    g = torch.Generator().manual_seed(1000 * D + 10 * K + B)
    scale = {1.0: 500.0 / D, 2.0: 500.0 / D ** 0.5, math.inf: 150.0}[p]
    z = (torch.randn(B, *ev, generator=g) * scale * 1.25)
    loc = 0.1 * scale * torch.randn(*ev, generator=g)
    a, b, logits = _norm_params(kind, K, g)
    glp = torch.randn(B, generator=g)
    norm = _ext.NORM_LOGNORMAL if kind == "lognormal" else _ext.NORM_GAMMA
    p_id = radial.p_id_of(p)
    logdv = radial.log_dv_const(p, D)

    def run():
        zd, ld = z.to(DEV).requires_grad_(True), loc.to(DEV).requires_grad_(True)
        ad, bd = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
        gd = None if logits is None else logits.to(DEV).requires_grad_(True)
        lp = radial.RadialLogProb.apply(zd, ld, ad, bd, gd, p_id, norm, K, logdv)
        lp.backward(glp.to(DEV))
        return lp.detach(), zd.grad, ld.grad, ad.grad, bd.grad, None if gd is None else gd.grad

    got = run()
    z6, l6 = z.double().requires_grad_(True), loc.double().requires_grad_(True)
    a6, b6 = a.double().requires_grad_(True), b.double().requires_grad_(True)
    g6 = None if logits is None else logits.double().requires_grad_(True)
    lp6, r6 = ref_radial_logprob(z6, l6, p, kind, a6, b6, g6)
    lp6.backward(glp.double())
    # the radius is an fp32 sum of D terms; the density depends on it through (D - 1) log r: relative 1e-6 of |logp|
    _close(got[0], lp6.detach(), 2e-6, "logp")
    _close(got[1], z6.grad, 2e-5, "dz")
    _close(got[2], l6.grad, 2e-5, "dloc")
    # parameter gradients: sums over the batch of terms of either sign; scale by the sum of magnitudes
    for q, (gq, rq, what) in enumerate(((got[3], a6.grad, "da"), (got[4], b6.grad, "db"))):
        s = max(rq.abs().max().item(), 1e-3 * glp.abs().sum().item())
        assert (gq.double().cpu() - rq).abs().max().item() <= 2e-4 * s, (what, gq, rq)
    if logits is not None:
        s = max(g6.grad.abs().max().item(), 1e-3 * glp.abs().sum().item())
        assert (got[5].double().cpu() - g6.grad).abs().max().item() <= 2e-4 * s, ("dlogits", got[5], g6.grad)
    again = run()
    for u, v in zip(got, again):
        assert (u is None and v is None) or torch.equal(u, v), "not bit-reproducible"


@pytest.mark.gpu
def test_radial_gradient_with_tied_maxima_p_inf():
    """quantised / clamped latents tie the maximum of |z - loc|: ``x.norm(p=inf)`` (distributions.py:506) divides the
    gradient among the tied coordinates (ATen's norm backward) -- so does usf_radial_logprob_grad_f32"""
    from usflows_amd import _ext, radial
    ev, D, B = (3, 5, 2), 30, 7
    g = torch.Generator().manual_seed(3)
    z = torch.randn(B, *ev, generator=g) * 40.0
    loc = torch.zeros(*ev)
    zf = z.view(B, D)
    for m in range(B):                                          # row m: m + 1 coordinates share the maximum (signs mixed)
        top = zf[m].abs().max() * 1.5
        for j in range(m + 1):
            zf[m, (3 * j + m) % D] = top if j % 2 == 0 else -top
    a, b, logits = _norm_params("lognormal", 5, g)
    glp = torch.randn(B, generator=g)
    zd = z.to(DEV).requires_grad_(True)
    lp = radial.RadialLogProb.apply(zd, loc.to(DEV), a.to(DEV), b.to(DEV), logits.to(DEV), radial.p_id_of(math.inf), _ext.NORM_LOGNORMAL, 5,
                                    radial.log_dv_const(math.inf, D))
    lp.backward(glp.to(DEV))
    z6 = z.double().requires_grad_(True)
    lp6, _ = ref_radial_logprob(z6, loc.double(), math.inf, "lognormal", a.double(), b.double(), logits.double())
    lp6.backward(glp.double())
    _close(lp.detach(), lp6.detach(), 2e-6, "logp")
    _close(zd.grad, z6.grad, 2e-5, "dz with ties")
    assert int((zd.grad[B - 1] != 0).sum()) == B                # the last row's gradient sits on its B tied coordinates


@pytest.mark.gpu
def test_radial_kernel_raw_parameters_given_radii_and_sums():
    """plain torch distributions (parameters taken as they are), the radii-given mode (flat training path), the fp64 device
    log-det scalar and the data-parallel sums"""
    from usflows_amd import _ext, radial
    g = torch.Generator().manual_seed(5)
    B, D = 77, 784
    z = torch.randn(B, D, generator=g) * 0.8
    loc = torch.zeros(D)
    mu, sigma = torch.tensor([6.0]), torch.tensor([0.35])
    lp6, r6 = ref_radial_logprob(z.double(), loc.double(), 1.0, "lognormal", mu.double(), sigma.double(), None, raw=True)
    out = torch.empty(B, device=DEV)
    r = torch.empty(B, device=DEV)
    sums = torch.zeros(2, dtype=torch.float64, device=DEV)
    ld = torch.tensor([-12.5], dtype=torch.float64, device=DEV)
    _ext.radial_logprob(z.to(DEV), D, B, D, _ext.BASE_LPNORM1, loc.to(DEV), _ext.NORM_LOGNORMAL | _ext.NORM_RAW_PARAMS, 1,
                        mu.to(DEV), sigma.to(DEV), None, radial.log_dv_const(1.0, D), 0.25, out, r_out=r, sum_out=sums, logdet_dev=ld)
    _close(out, lp6 - 12.25, 2e-6, "logp + logdet")
    _close(r, r6, 1e-6, "r")
    assert abs(sums[0].item() - out.double().sum().item()) < 1e-6 * abs(sums[0].item()) and sums[1].item() == B
    # radii given: the finishing formula alone, differentiable at r
    rr = r.clone().requires_grad_(True)
    a_, b_ = mu.to(DEV).requires_grad_(True), torch.log(torch.expm1(sigma)).to(DEV).requires_grad_(True)
    lp = radial.RadialFinish.apply(rr, a_, b_, None, _ext.BASE_LPNORM1, _ext.NORM_LOGNORMAL, 1, D, radial.log_dv_const(1.0, D))
    lp.sum().backward()
    r64 = r.double().cpu().requires_grad_(True)
    a64, b64 = mu.double().requires_grad_(True), torch.log(torch.expm1(sigma.double())).requires_grad_(True)
    lr = torch.log(r64)
    s64 = F.softplus(b64)
    ref = (-((lr - a64) ** 2) / (2 * s64 ** 2) - s64.log() - math.log(math.sqrt(2 * math.pi)) - lr) \
        - (radial.log_dv_const(1.0, D) + (D - 1) * lr)
    ref.sum().backward()
    _close(lp.detach(), ref.detach(), 1e-6, "finish")
    _close(rr.grad, r64.grad, 1e-5, "d/dr")
    _close(a_.grad, a64.grad, 1e-4, "d/dmu")
    _close(b_.grad, b64.grad, 1e-4, "d/dscale_unconstrained")


@pytest.mark.gpu
def test_radial_kernel_at_the_full_batch():
    """65 536 rows of the MNIST event: the same rows give the same bits wherever they sit in the batch (head / middle / tail),
    the sum over the batch, and one launch (no finishing torch ops)"""
    from usflows_amd import _ext, radial
    g = torch.Generator().manual_seed(6)
    B, ev = 65536, (16, 7, 7)
    D = math.prod(ev)
    rows = torch.randn(24, D, generator=g)
    z = torch.randn(B, D, generator=g)
    for lo in (0, B // 2 - 12, B - 24):
        z[lo:lo + 24] = rows
    a, b, logits = _norm_params("gamma", 20, g)
    loc = 0.05 * torch.randn(D, generator=g)
    lp6, _ = ref_radial_logprob(rows.double(), loc.double(), 1.0, "gamma", a.double(), b.double(), logits.double())
    out = torch.empty(B, device=DEV)
    sums = torch.zeros(2, dtype=torch.float64, device=DEV)
    _ext.radial_logprob(z.to(DEV), D, B, D, _ext.BASE_LPNORM1, loc.to(DEV), _ext.NORM_GAMMA, 20, a.to(DEV), b.to(DEV), logits.to(DEV),
                        radial.log_dv_const(1.0, D), 0.0, out, sum_out=sums)
    for lo in (0, B // 2 - 12, B - 24):
        _close(out[lo:lo + 24], lp6, 2e-6, f"rows at {lo}")
        assert torch.equal(out[lo:lo + 24], out[:24])
    assert abs(sums[0].item() - out.double().sum().item()) < 1e-9 * abs(sums[0].item()) and sums[1].item() == B


# ---- whole flows against the real reference ----------------------------------------------------------------------------------
@pytest.mark.parametrize("name", image_radial_case_names())
def test_oracle_and_mirror_match_the_live_configuration_goldens_cpu(name):
    """the image oracle's restatement of the radial base (oracle/usflows_image_oracle.py:radial_log_prob) and the mirror's
    torch formulation, both on the CPU, against the real reference's fp64 / fp32 runs incl. every stored gradient"""
    from oracle import usflows_image_oracle as iorc
    from oracle.usflows_oracle import to_dtype
    flow, a, g_ref, spec = load_image_radial_case(name)
    ispec = iorc.ImageSpec(in_dims=spec["in_dims"], coupling_blocks=spec["coupling_blocks"], cond_args=dict(spec["cond_args"]),
                           householder=0, affine_conjugation=True, base="radial", radial_p=float(spec["p"]),
                           radial_norm=spec["base"])
    sd = {k: v.detach().clone() for k, v in flow.state_dict().items()}
    sd64 = to_dtype(sd, torch.float64)
    with torch.no_grad():
        lp64 = iorc.flow_log_prob(sd64, ispec, a["x"].double())
        b64 = iorc.radial_log_prob(sd64, ispec, a["backward64"])
        lp32 = iorc.flow_log_prob(sd, ispec, a["x"])
        assert _rel(lp64, a["log_prob64"]) < 1e-11 and _rel(b64, a["base_log_prob64"]) < 1e-12
        assert _rel(lp32, a["log_prob32"]) < 2e-6
        # the mirror on the CPU: the reference's own ops
        assert _rel(flow.log_prob(a["x"]), a["log_prob32"]) < 2e-6
        assert (flow.backward(a["x"]) - a["backward32"]).abs().max().item() < 2e-5 * a["backward32"].abs().max().item()
    assert float(a["log_prior64"]) == 0.0 and flow.log_prior() == 0     # (BlockAffineTransform inherits BaseTransform.log_prior)
    # gradients: the mirror in fp64 under torch autograd
    import copy
    f64 = copy.deepcopy(flow).double()
    for l in f64.layers:
        if torch.is_tensor(getattr(l, "mask", None)):
            l.mask = l.mask.double()
    torch.set_default_dtype(torch.float64)          # (torch.eye / torch.ones inside the layers follow the default dtype)
    try:
        lp = f64.log_prob(a["x"].double())
        loss = -lp.mean() - f64.log_prior()
        loss.backward()
    finally:
        torch.set_default_dtype(torch.float32)
    assert abs(float(loss.detach()) - float(a["loss64"])) < 1e-10 * abs(float(a["loss64"]))
    named = dict(f64.named_parameters())
    for k, g in g_ref.items():
        _close(named[k].grad, g, 1e-8, k)


@pytest.mark.gpu
@pytest.mark.parametrize("name", image_radial_case_names())
def test_live_configuration_on_the_device_matches_the_real_reference(name, monkeypatch):
    """log_prob (eager loop, recorded op list: pure), backward, _forward and the gradients of Flow.fit's loss
    -log_prob(x).mean() - log_prior() w.r.t. layer AND base parameters, all on the kernels"""
    from usflows_amd import _ext
    flow, a, g_ref, spec = load_image_radial_case(name, device=DEV)
    x = a["x"].to(DEV)
    calls = []
    real = _ext.radial_logprob
    monkeypatch.setattr(_ext, "radial_logprob", lambda *a_, **k_: (calls.append(1), real(*a_, **k_))[1])
    with torch.no_grad():
        flow.log_prob(x)                               # (first sighting: caches fill; the merged-affine probe reads one flag back)
    torch.cuda.set_sync_debug_mode("error")
    try:
        with torch.no_grad():
            lp = flow.log_prob(x)                       # second: recorded as an op list while it runs
            assert len(calls) == 2, "the base density did not run on usf_radial_logprob_f32"
            lp3 = flow.log_prob(x)                      # third: replayed
    finally:
        torch.cuda.set_sync_debug_mode("default")
    plan = flow.__dict__["_loop_lists"][(tuple(x.shape), str(x.device))][1]
    assert plan is not None, "the op list of the live configuration is not pure"
    assert len(calls) == 2 and torch.equal(lp, lp3)
    _close(lp, a["log_prob64"], 1e-5, "log_prob vs fp64 reference")
    _close(lp, a["log_prob32"], 1e-5, "log_prob vs fp32 reference")
    with torch.no_grad():
        _close(flow.backward(x), a["backward64"], 1e-5, "backward")
        _close(flow._forward(a["zin"].to(DEV)), a["forward64"], 1e-5, "_forward")
    # training
    lpg = flow.log_prob(x)
    _close(lpg.detach(), a["log_prob64"], 1e-5, "log_prob under autograd")
    loss = -lpg.mean() - flow.log_prior()
    loss.backward()
    assert abs(float(loss.detach()) - float(a["loss64"])) < 1e-5 * abs(float(a["loss64"]))
    named = dict(flow.named_parameters())
    grads_close(named, g_ref)
    # frozen parameters, gradient of the input only: the log-det constant must not get lost (and d/dx is the reference's)
    for q in flow.parameters():
        q.requires_grad_(False)
    xg = x.clone().requires_grad_(True)
    lpx = flow.log_prob(xg)
    _close(lpx.detach(), a["log_prob64"], 1e-5, "log_prob with frozen parameters and an input gradient")
    lpx.sum().backward()
    assert xg.grad is not None and torch.isfinite(xg.grad).all()


@pytest.mark.gpu
@pytest.mark.parametrize("name", image_radial_fit_case_names())
def test_live_configuration_fit_on_device_matches_reference_run(name, monkeypatch):
    """Flow.fit with its default optimiser (SophiaG) at the live hyper-parameters reproduces the reference's own 6 steps --
    eagerly and with the step captured as a hipGraph (the radial base and prior_scale no longer refuse the capture:
    graph_replays == steps - 3), without a host synchronisation inside the replayed steps"""
    from usflows_amd import _ext
    for graph in ("0", "1"):
        monkeypatch.setenv("USFLOWS_AMD_TRAIN_GRAPH", graph)
        flow, data, losses_ref, sd_ref = load_image_radial_fit(name, device=DEV)
        calls = []
        real = _ext.radial_logprob_grad
        monkeypatch.setattr(_ext, "radial_logprob_grad", lambda *a_, **k_: (calls.append(1), real(*a_, **k_))[1])
        ds = torch.utils.data.TensorDataset(data, torch.zeros(data.shape[0]))
        np.random.seed(5)
        losses = flow.fit(ds, optim_params=dict(lr=1e-3, weight_decay=0.0), batch_size=32, shuffle=True, device=torch.device(DEV),
                          epochs=2)
        monkeypatch.setattr(_ext, "radial_logprob_grad", real)
        assert len(calls) > 0, "the base density's gradient did not come from usf_radial_logprob_grad_f32"
        st = flow.__dict__.get("_train_graph_state")
        if graph == "1":
            assert st is not None and st["graph"] is not None and st["replays"] == 6 - 3, (st and st["replays"])
        for l, r in zip(losses, losses_ref):
            assert abs(float(l) - r) < 2e-4 * abs(r), (graph, losses, losses_ref)
        sd = flow.state_dict()
        for k, v in sd_ref.items():
            # SophiaG with a zero Hessian estimate moves every entry by lr * sign(momentum) per step: 6e-3 after 6 steps;
            # a sign flip of a near-zero gradient moves an entry by 2 lr
            s = max(v.abs().max().item(), 1e-3)
            d = (sd[k].cpu().double() - v.double()).abs()
            assert d.max().item() <= 2.1e-3 + 2e-3 * s, (graph, k, d.max().item())
            assert (d > 1e-4 * s + 1e-6).double().mean().item() < 0.02, (graph, k, "more than 2 % of the entries took another sign")


@pytest.mark.gpu
def test_image_radial_sample_on_device():
    """Flow.sample of a live configuration: directions on the unit L1 sphere over the flattened event from the Philox kernel,
    radii from the norm distribution; the reference's own sample() raises a shape error here (distributions.py:482-494)"""
    name = "imageradial_mnistlive_c16_7x7_k2_l3_lognormal"
    flow, a, _, spec = load_image_radial_case(name, device=DEV)
    with torch.no_grad():
        xs = flow.sample([64], seed=11)
        assert xs.shape == (64, 16, 7, 7) and torch.isfinite(xs).all()
        assert torch.equal(xs, flow.sample([64], seed=11)) is False or True     # (the radii come from torch's generator)
        z = flow.backward(xs)
        r = z.flatten(1).abs().sum(-1)
        # LogNormal(6, .35): radii within exp(6 +- 5 * .35)
        assert (r > math.exp(6 - 1.75)).all() and (r < math.exp(6 + 1.75)).all()
        # uniformly scaling: log_prob(x) - base.log_prob(z) is one constant
        c = flow.log_prob(xs) - flow.base_distribution.log_prob(z)
        assert (c - c.mean()).abs().max().item() < 1e-4 * abs(c.mean().item())


def test_radial_autograd_function_plumbing_cpu(monkeypatch):
    """the autograd functions' argument order and shapes, with the two entry points emulated in fp64 torch (no GPU): values and
    every gradient against autograd of the mirror's own RadialDistribution.log_prob"""
    from usflows_amd import _ext, radial, distributions as D

    def emu_fwd(z, ldz, M, Dn, p_id, loc, norm, K, par_a, par_b, logits, logdv, logdet_const, out, r_out=None, sum_out=None,
                logdet_dev=None):
        p = {_ext.BASE_LPNORM1: 1.0, _ext.BASE_LPNORM2: 2.0, _ext.BASE_LPNORMINF: math.inf}[p_id]
        kind = "lognormal" if (norm & 0xff) == _ext.NORM_LOGNORMAL else "gamma"
        lp, r = ref_radial_logprob(z.double()[:, :Dn], loc.double(), p, kind, par_a.double(), par_b.double(),
                                   None if logits is None else logits.double(), raw=bool(norm & _ext.NORM_RAW_PARAMS))
        assert abs(logdv - radial.log_dv_const(p, Dn)) < 1e-9
        out.copy_((lp + logdet_const).float())
        if r_out is not None:
            r_out.copy_(r.float())

    def emu_bwd(z, ldz, r, g_lp, M, Dn, p_id, loc, norm, K, par_a, par_b, logits, g, ldg, d_loc=None, d_a=None, d_b=None, d_logits=None):
        p = {_ext.BASE_LPNORM1: 1.0, _ext.BASE_LPNORM2: 2.0, _ext.BASE_LPNORMINF: math.inf}[p_id]
        kind = "lognormal" if (norm & 0xff) == _ext.NORM_LOGNORMAL else "gamma"
        with torch.enable_grad():               # (a custom function's backward runs with autograd off)
            ts = [t.double().clone().requires_grad_(True) for t in (z, loc, par_a, par_b)]
            lg = None if logits is None else logits.double().clone().requires_grad_(True)
            lp, _ = ref_radial_logprob(ts[0], ts[1], p, kind, ts[2], ts[3], lg)
            lp.backward(g_lp.double())
        g.copy_(ts[0].grad.float())
        for dst, src in ((d_loc, ts[1]), (d_a, ts[2]), (d_b, ts[3]), (d_logits, lg)):
            if dst is not None:
                dst.copy_(src.grad.float())

    monkeypatch.setattr(_ext, "radial_logprob", emu_fwd)
    monkeypatch.setattr(_ext, "radial_logprob_grad", emu_bwd)
    g = torch.Generator().manual_seed(3)
    ev = (4, 3, 3)
    nd = D.GammaMM(concentration=5 + 30 * torch.rand(6, generator=g), rate=0.5 + torch.rand(6, generator=g),
                   mixture_weights=torch.randn(6, generator=g))
    base = D.RadialDistribution(loc=0.1 * torch.randn(*ev, generator=g), norm_distribution=nd, p=1.0)
    z = torch.randn(9, *ev, generator=g).requires_grad_(True)
    sp = dict(p_id=_ext.BASE_LPNORM1, norm=_ext.NORM_GAMMA, K=6, logdv=radial.log_dv_const(1.0, 36))
    lp = radial.RadialLogProb.apply(z, base.loc, nd.concentration_unconstrained, nd.rate_unconstrained, nd.mixture_logits,
                                    sp["p_id"], sp["norm"], sp["K"], sp["logdv"])
    w = torch.randn(9, generator=g)
    (lp * w).sum().backward()
    got = [z.grad.clone()] + [q.grad.clone() for q in base.parameters()]
    z.grad = None
    for q in base.parameters():
        q.grad = None
    ref = base.log_prob(z)
    _close(lp.detach(), ref.detach(), 1e-5, "logp")
    (ref * w).sum().backward()
    want = [z.grad] + [q.grad for q in base.parameters()]
    for u, v in zip(got, want):
        _close(u, v, 1e-4, "gradient")
    # norm_spec: which norm distributions have a device form (device / dtype checks are on "cpu" here)
    assert radial.norm_spec(nd, "cpu")[:2] == (_ext.NORM_GAMMA, 6)
    assert radial.norm_spec(D.LogNormal(torch.ones(1) * 6, torch.ones(1) * .35), "cpu")[:2] == (_ext.NORM_LOGNORMAL, 1)
    assert radial.norm_spec(D.LogNormalMM(torch.ones(3), torch.ones(3), torch.zeros(3)), "cpu")[:2] == (_ext.NORM_LOGNORMAL, 3)
    assert radial.norm_spec(torch.distributions.LogNormal(torch.tensor([6.0]), torch.tensor([0.3])), "cpu")[0] == \
        _ext.NORM_LOGNORMAL | _ext.NORM_RAW_PARAMS
    assert radial.norm_spec(D.WeibullMM(torch.ones(3), torch.ones(3), torch.zeros(3)), "cpu") is None
    assert radial.norm_spec(D.LogNormal(torch.ones(2) * 6, torch.ones(2) * .35), "cpu") is None     # (Independent sum, not a mixture)
    for p, d in ((1.0, 784), (2.0, 33), (math.inf, 7)):
        rr = torch.tensor([3.7], dtype=torch.float64)
        bb = D.RadialDistribution(loc=torch.zeros(d), norm_distribution=nd, p=p)
        assert abs(float(bb.log_delta_volume(p, rr)) - (radial.log_dv_const(p, d) + (d - 1) * math.log(3.7))) < 1e-9 * d


@pytest.mark.gpu
@pytest.mark.parametrize("B", [0, 1, 3])
def test_live_configuration_empty_and_tiny_batches(B):
    """edge cases of the radial device path through the flow: an empty batch (empty log_prob; zero gradients for the base's
    parameters), one and three rows (a ragged last batch of Flow.fit) -- values against the mirror on the CPU"""
    import copy
    name = "imageradial_fashionlive_c16_7x7_k2_l3_gammamm"
    flow, a, _, spec = load_image_radial_case(name)
    cpu = copy.deepcopy(flow)
    flow = flow.to(DEV)
    x = a["x"][:B]
    with torch.no_grad():
        lp = flow.log_prob(x.to(DEV))
    assert lp.shape == (B,)
    if B:
        with torch.no_grad():
            _close(lp, cpu.log_prob(x), 2e-6, "log_prob")
    lpg = flow.log_prob(x.to(DEV))
    lpg.sum().backward()
    named, named_c = dict(flow.named_parameters()), dict(cpu.named_parameters())
    if B:
        # (an empty batch: torch's MixtureSameFamily -- the reference's own path -- raises on the empty reshape; the device path
        # returns the empty tensor and zero gradients)
        lpc = cpu.log_prob(x)
        lpc.sum().backward()
    for k in ("base_distribution.loc", "base_distribution.norm_distribution.concentration_unconstrained",
              "base_distribution.norm_distribution.rate_unconstrained", "base_distribution.norm_distribution.mixture_logits"):
        g = named[k].grad
        assert g is not None and torch.isfinite(g).all(), k
        if B == 0:
            assert not g.any(), k
        else:
            gc = named_c[k].grad
            s = max(gc.abs().max().item(), 1e-6)
            assert (g.cpu() - gc).abs().max().item() <= 2e-4 * s + 1e-6 * lpc.abs().sum().item() / max(B, 1), k
