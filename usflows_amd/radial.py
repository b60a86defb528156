"""``RadialDistribution.log_prob`` on the device, forward and backward (the base distribution of every live
image configuration of the reference: experiments/mnist/mnist.yaml:79-92 ``LogNormal(6, .35)``,
experiments/fashion/fashionclasses_veriflow.yaml:79-93 ``GammaMM`` x 20, experiments/cifar/cifar.yaml).

Reference arithmetic (src/usflows/distributions.py:501-549)::

    r    = (x - loc).norm(p, dim=event_dims)
    logp = norm_distribution.log_prob(r.unsqueeze(-1)).squeeze(-1) - log_delta_volume(p, r)

The torch formulation is ~25 small launches plus the construction of a fresh, argument-validating distribution object per
call (``DistributionModule.distribution``, distributions.py:127-139) whose validation reads a flag back to the host: no
stream capture, one host round trip per training step.  Here the whole density is ``usf_radial_logprob_f32`` (radius
reduction over the flattened event + the norm distribution's mixture density + the volume term, one launch) and its
gradient ``usf_radial_logprob_grad_f32`` (d/dz, d/dloc, d/d(norm parameters), d/d(mixture logits)): no host
synchronisation, capturable, for flat AND image-shaped events.

Served norm distributions (``norm_spec``): the ``LogNormal`` / ``Gamma`` modules with one-element parameters, ``GammaMM`` and
``MixtureModel`` over ``torch.distributions.LogNormal`` (``LogNormalMM``) with up to 64 components along one axis, and plain
``torch.distributions.LogNormal`` / ``Gamma`` objects with one-element parameters on the device.  Anything else keeps the
distribution object's op chain.
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import _ext
from . import distributions as D_


def p_id_of(p: float) -> Optional[int]:
    return {1.0: _ext.BASE_LPNORM1, 2.0: _ext.BASE_LPNORM2, float("inf"): _ext.BASE_LPNORMINF}.get(float(p))


def log_dv_const(p: float, d: int) -> float:
    """the r-independent part of ``RadialDistribution.log_delta_volume`` (distributions.py:513-549), in fp64 on the host:
    log dV_p(r) = log_dv_const + (d - 1) log r"""
    if p == 1:
        return math.log(2) * d - math.lgamma(d)                      # sum_{i<d} log i = log (d-1)!
    if p == 2:
        return math.log(d) + (d / 2) * math.log(math.pi) - math.lgamma(d / 2 + 1)
    if p == math.inf:
        return math.log(d) + d * math.log(2)
    raise ValueError(f"p={p} not implemented. Use p=1,2, or infinity")


def _dev_f32(t, device) -> bool:
    return torch.is_tensor(t) and t.dtype == torch.float32 and t.device == device and t.is_contiguous()


def norm_spec(nd, device):
    """(norm id, K, par_a, par_b, logits | None) -- the STORED tensors of the norm distribution, as the kernels read them --
    or None when this norm distribution has no device form"""
    device = torch.device(device)
    if isinstance(nd, D_.LogNormal) and type(nd) is D_.LogNormal:
        a, b = nd.loc, nd.scale_unconstrained
        if a.numel() == 1 and b.numel() == 1 and _dev_f32(a, device) and _dev_f32(b, device):
            return _ext.NORM_LOGNORMAL, 1, a, b, None
        return None
    if isinstance(nd, D_.Gamma) and type(nd) is D_.Gamma:
        a, b = nd.concentration_unconstrained, nd.rate_unconstrained
        if a.numel() == 1 and b.numel() == 1 and _dev_f32(a, device) and _dev_f32(b, device):
            return _ext.NORM_GAMMA, 1, a, b, None
        return None
    if isinstance(nd, D_.GammaMM) and type(nd) is D_.GammaMM:
        a, b, l = nd.concentration_unconstrained, nd.rate_unconstrained, nd.mixture_logits
        K = a.numel()
        if a.dim() == 1 and b.shape == a.shape and l.shape == a.shape and 1 <= K <= _ext.RADIAL_MAX_K \
                and all(_dev_f32(t, device) for t in (a, b, l)):
            return _ext.NORM_GAMMA, K, a, b, l
        return None
    if isinstance(nd, D_.MixtureModel) and nd.component_distribution_class is torch.distributions.LogNormal \
            and list(nd.param_names) == ["loc", "scale"]:
        a, b = nd.unconstrained_params[0], nd.unconstrained_params[1]
        l = nd.mixture_logits
        K = a.numel()
        c = nd.param_constraints.get("scale")
        positive = isinstance(c, type(torch.distributions.constraints.positive)) and getattr(c, "lower_bound", None) == 0.0
        if a.dim() == 1 and b.shape == a.shape and l.shape == a.shape and 1 <= K <= _ext.RADIAL_MAX_K \
                and all(_dev_f32(t, device) for t in (a, b, l)):
            return _ext.NORM_LOGNORMAL | (0 if positive else _ext.NORM_RAW_PARAMS), K, a, b, l
        return None
    if type(nd) is torch.distributions.LogNormal:
        a, b = nd.loc, nd.scale
        if a.numel() == 1 and b.numel() == 1 and len(nd.batch_shape) <= 1 and _dev_f32(a, device) and _dev_f32(b, device):
            return _ext.NORM_LOGNORMAL | _ext.NORM_RAW_PARAMS, 1, a, b, None
        return None
    if type(nd) is torch.distributions.Gamma:
        a, b = nd.concentration, nd.rate
        if a.numel() == 1 and b.numel() == 1 and len(nd.batch_shape) <= 1 and _dev_f32(a, device) and _dev_f32(b, device):
            return _ext.NORM_GAMMA | _ext.NORM_RAW_PARAMS, 1, a, b, None
        return None
    return None


def radial_spec(base, device):
    """everything the kernels need of a ``RadialDistribution``: dict(p_id, D, loc, norm, K, a, b, logits, logdv) or None"""
    if not isinstance(base, D_.RadialDistribution) or base.n_batch_dims != 0:
        return None
    p_id = p_id_of(base.p)
    if p_id is None or not _dev_f32(base.loc, torch.device(device)):
        return None
    ns = norm_spec(base.norm_distribution, device)
    if ns is None:
        return None
    d = int(base.loc.numel())
    return dict(p_id=p_id, D=d, loc=base.loc, norm=ns[0], K=ns[1], a=ns[2], b=ns[3], logits=ns[4],
                logdv=log_dv_const(base.p, d))


class RadialLogProb(torch.autograd.Function):
    """logp [B] of rows z [B, *event] under the radial density; gradients of z, loc, the norm parameters and the logits"""

    @staticmethod
    def forward(ctx, z, loc, par_a, par_b, logits, p_id, norm, K, logdv):
        B = z.shape[0]
        d = loc.numel()
        zf = z.detach().reshape(B, d).contiguous()
        locf = loc.detach().reshape(-1)
        out = torch.empty(B, dtype=torch.float32, device=z.device)
        r = torch.empty(B, dtype=torch.float32, device=z.device)
        a, b = par_a.detach(), par_b.detach()
        lg = None if logits is None else logits.detach()
        if B > 0:
            _ext.radial_logprob(zf, d, B, d, p_id, locf, norm, K, a, b, lg, logdv, 0.0, out, r_out=r)
        ctx.save_for_backward(zf, r, locf, a, b, lg)
        ctx.cfg = (p_id, norm, K, tuple(z.shape), tuple(loc.shape), tuple(par_a.shape), tuple(par_b.shape),
                   None if logits is None else tuple(logits.shape))
        return out

    @staticmethod
    def backward(ctx, g_lp):
        zf, r, locf, a, b, lg = ctx.saved_tensors
        p_id, norm, K, zshape, lshape, ashape, bshape, gshape = ctx.cfg
        B, d = zf.shape
        need = ctx.needs_input_grad
        dev = zf.device
        g = torch.empty_like(zf)
        d_loc = torch.empty(d, dtype=torch.float32, device=dev) if need[1] else None
        d_a = torch.empty(K, dtype=torch.float32, device=dev) if need[2] else None
        d_b = torch.empty(K, dtype=torch.float32, device=dev) if need[3] else None
        d_l = torch.empty(K, dtype=torch.float32, device=dev) if (lg is not None and need[4]) else None
        _ext.radial_logprob_grad(zf, d, r, g_lp.contiguous(), B, d, p_id, locf, norm, K, a, b, lg, g, d, d_loc=d_loc, d_a=d_a,
                                 d_b=d_b, d_logits=d_l)
        return (g.reshape(zshape) if need[0] else None,
                None if d_loc is None else d_loc.reshape(lshape),
                None if d_a is None else d_a.reshape(ashape),
                None if d_b is None else d_b.reshape(bshape),
                None if d_l is None else d_l.reshape(gshape), None, None, None, None)


class RadialFinish(torch.autograd.Function):
    """logp [B] from GIVEN radii r [B] (the flat training path's tail kernel reduces them): the finishing formula
    norm_dist.log_prob(r) - log dV_p(r) and its gradients at r, the norm parameters and the logits"""

    @staticmethod
    def forward(ctx, r, par_a, par_b, logits, p_id, norm, K, d, logdv):
        B = r.shape[0]
        rc = r.detach().contiguous()
        out = torch.empty(B, dtype=torch.float32, device=r.device)
        a, b = par_a.detach(), par_b.detach()
        lg = None if logits is None else logits.detach()
        if B > 0:
            _ext.radial_logprob(None, 0, B, d, p_id, None, norm, K, a, b, lg, logdv, 0.0, out, r_out=rc)
        ctx.save_for_backward(rc, a, b, lg)
        ctx.cfg = (p_id, norm, K, d, tuple(par_a.shape), tuple(par_b.shape), None if logits is None else tuple(logits.shape))
        return out

    @staticmethod
    def backward(ctx, g_lp):
        rc, a, b, lg = ctx.saved_tensors
        p_id, norm, K, d, ashape, bshape, gshape = ctx.cfg
        B = rc.shape[0]
        need = ctx.needs_input_grad
        dev = rc.device
        g = torch.empty(B, dtype=torch.float32, device=dev)
        d_a = torch.empty(K, dtype=torch.float32, device=dev) if need[1] else None
        d_b = torch.empty(K, dtype=torch.float32, device=dev) if need[2] else None
        d_l = torch.empty(K, dtype=torch.float32, device=dev) if (lg is not None and need[3]) else None
        _ext.radial_logprob_grad(None, 0, rc, g_lp.contiguous(), B, d, p_id, None, norm, K, a, b, lg, g, 0, d_a=d_a, d_b=d_b,
                                 d_logits=d_l)
        return (g if need[0] else None,
                None if d_a is None else d_a.reshape(ashape),
                None if d_b is None else d_b.reshape(bshape),
                None if d_l is None else d_l.reshape(gshape), None, None, None, None, None)


def log_prob(base, z: torch.Tensor, logdet_dev: Optional[torch.Tensor] = None, sum_out: Optional[torch.Tensor] = None,
             ldz: Optional[int] = None):
    """``base.log_prob(z)`` (+ the fp64 device scalar ``logdet_dev``) on the kernels, or None when this base / input has no
    device form.  Differentiable when autograd is on and anything involved requires a gradient.  ``ldz``: z is a raw
    [B, ldz] fp32 buffer whose first D columns are the event (the flat engine's latent buffer); inference only."""
    if not (torch.is_tensor(z) and z.is_cuda and z.dtype == torch.float32 and z.dim() >= 2):
        return None
    sp = radial_spec(base, z.device)
    if sp is None:
        return None
    d = sp["D"]
    if ldz is None and (math.prod(z.shape[1:]) != d or tuple(z.shape[1:]) != tuple(base.loc.shape)):
        return None
    _ext.load()
    train = torch.is_grad_enabled() and (z.requires_grad or any(
        torch.is_tensor(t) and t.requires_grad for t in (sp["loc"], sp["a"], sp["b"], sp["logits"])))
    if train:
        if ldz is not None or logdet_dev is not None or sum_out is not None:
            return None
        return RadialLogProb.apply(z, sp["loc"], sp["a"], sp["b"], sp["logits"], sp["p_id"], sp["norm"], sp["K"], sp["logdv"])
    B = z.shape[0]
    out = torch.empty(B, dtype=torch.float32, device=z.device)
    if B == 0:
        return out
    if ldz is None:
        zf, ldz = z.reshape(B, d), d
        if not zf.is_contiguous():
            zf = zf.contiguous()
    else:
        zf = z
    lg = sp["logits"]
    _ext.radial_logprob(zf, ldz, B, d, sp["p_id"], sp["loc"].detach().reshape(-1), sp["norm"], sp["K"], sp["a"].detach(),
                        sp["b"].detach(), None if lg is None else lg.detach(), sp["logdv"], 0.0, out, sum_out=sum_out,
                        logdet_dev=logdet_dev)
    return out


def log_prob_from_radius(base, r: torch.Tensor):
    """the finishing formula on given radii (differentiable), or None when the norm distribution has no device form"""
    if not (torch.is_tensor(r) and r.is_cuda and r.dtype == torch.float32 and r.dim() == 1):
        return None
    sp = radial_spec(base, r.device)
    if sp is None:
        return None
    _ext.load()
    return RadialFinish.apply(r, sp["a"], sp["b"], sp["logits"], sp["p_id"], sp["norm"], sp["K"], sp["D"], sp["logdv"])
