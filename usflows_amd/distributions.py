"""Base distributions of the hot path -- mirror of the in-scope part of the reference's
``src/usflows/distributions.py``: the ``Independent`` wrapper
``Flow.__init__`` applies, the ``DistributionModule`` family (LogNormal / Laplace / Normal / Gamma)
and the Lp-``RadialDistribution`` with its unit-ball sampler and UDL profile helpers, and the mixture families
used as radial norm distributions / data generators by the live configs (``GammaMM``, ``MixtureModel`` with
``GMM`` / ``LogNormalMM`` / ``WeibullMM``).  ``RotatedLaplace`` / ``Chi`` are out of scope.

On the device fast path the per-sample reduction over the feature axis (the only per-sample
reduction on the whole path) runs in ``usf_base_logprob_f32``; the code here is the host-side
definition (CPU / autograd) and the O(B) finishing math of the radial density.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Optional, Union

import torch
from torch import nn
from torch.distributions import Distribution, constraints
from torch.distributions import Independent as DIndependent
from torch.nn import Module, Parameter
from torch.nn.functional import softplus


def inv_softplus(x: torch.Tensor) -> torch.Tensor:
    """utils.py:3-9"""
    return torch.log(torch.exp(x) - 1)


class Independent(nn.Module, torch.distributions.Independent):
    """``torch.distributions.Independent`` that is also an nn.Module (distributions.py:709-728)."""

    def __init__(self, base_distribution, reinterpreted_batch_ndims: int = 0, *args, **kwargs):
        nn.Module.__init__(self)
        self._base_distribution = base_distribution
        torch.distributions.Independent.__init__(
            self, self._base_distribution, reinterpreted_batch_ndims=reinterpreted_batch_ndims, *args, **kwargs)


class DistributionModule(Module):
    """A torch distribution whose parameters are nn.Parameters (distributions.py:117-160)."""

    def __init__(self, distribution_class: type, n_batch_dims: int = 0):
        super().__init__()
        self.distribution_class = distribution_class
        self.n_batch_dims = n_batch_dims

    @property
    def distribution(self) -> Distribution:
        d = self.distribution_class(**self._get_distribution_params())
        extra = len(d.batch_shape) - self.n_batch_dims
        if extra > 0:
            d = DIndependent(d, extra)
        return d

    def _get_distribution_params(self) -> Dict[str, torch.Tensor]:
        raise NotImplementedError

    def forward(self, x):
        return self.log_prob(x)

    def sample(self, sample_shape: Iterable[int] = None):
        return self.distribution.sample(sample_shape)

    def log_prob(self, x):
        return self.distribution.log_prob(x)

    @property
    def event_shape(self):
        return self.distribution.event_shape

    @property
    def batch_shape(self):
        return self.distribution.batch_shape


class Gamma(DistributionModule):
    def __init__(self, concentration, rate, device: str = "cpu"):
        super().__init__(torch.distributions.Gamma)
        self.concentration_unconstrained = Parameter(inv_softplus(concentration))
        self.rate_unconstrained = Parameter(inv_softplus(rate))
        self.to(device)

    def _get_distribution_params(self):
        return {"concentration": softplus(self.concentration_unconstrained),
                "rate": softplus(self.rate_unconstrained)}


class _LocScale(DistributionModule):
    def __init__(self, cls, loc, scale, device):
        super().__init__(cls)
        self.loc = Parameter(loc)
        self.scale_unconstrained = Parameter(inv_softplus(scale))
        self.to(device)

    def _get_distribution_params(self):
        return {"loc": self.loc, "scale": softplus(self.scale_unconstrained)}


class LogNormal(_LocScale):
    def __init__(self, loc, scale, device: str = "cpu"):
        super().__init__(torch.distributions.LogNormal, loc, scale, device)


class Laplace(_LocScale):
    def __init__(self, loc, scale, device: str = "cpu"):
        super().__init__(torch.distributions.Laplace, loc, scale, device)


class Normal(_LocScale):
    def __init__(self, loc, scale, device: str = "cpu"):
        super().__init__(torch.distributions.Normal, loc, scale, device)

    def _get_distribution_params(self):
        sc = softplus(self.scale_unconstrained)
        if self.scale_unconstrained.dim() == 0:
            sc = sc.expand_as(self.loc)
        return {"loc": self.loc, "scale": sc}


class GammaMM(DistributionModule):
    """Mixture of Gamma distributions (distributions.py:674-707): the radial norm distribution of the reference's
    live configs (experiments/synthetic/gaussian_mixture.yaml:84, tests/explib/mnist.yaml:84).  Parameters: softplus-
    constrained concentration / rate with the COMPONENT axis first, mixture logits."""

    def __init__(self, concentration: torch.Tensor, rate: torch.Tensor, mixture_weights: torch.Tensor, device: str = "cpu"):
        super().__init__(torch.distributions.MixtureSameFamily)
        self.concentration_unconstrained = Parameter(inv_softplus(concentration))
        self.rate_unconstrained = Parameter(inv_softplus(rate))
        self.mixture_logits = Parameter(mixture_weights)
        self.to(device)

    def _get_distribution_params(self):
        concentration = softplus(self.concentration_unconstrained)
        rate = softplus(self.rate_unconstrained)
        order = list(range(1, concentration.dim())) + [0]          # component axis last (:690-694)
        comp = torch.distributions.Gamma(concentration.permute(*order), rate.permute(*order))
        mix = torch.distributions.Categorical(logits=self.mixture_logits)
        return {"mixture_distribution": mix, "component_distribution": comp}


class MixtureModel(DistributionModule):
    """Mixture of ``distribution_class`` components (distributions.py:730-795): parameters named ``param_names`` with
    ``constraints.positive`` ones stored through inv_softplus, mixture logits."""

    def __init__(self, distribution_class, param_names, param_constraints, *params, mixture_weights, device="cpu"):
        super().__init__(distribution_class=torch.distributions.MixtureSameFamily)
        self.component_distribution_class = distribution_class
        self.param_names = param_names
        self.param_constraints = param_constraints
        self.unconstrained_params = nn.ParameterList()
        for name, param in zip(param_names, params):
            if param_constraints.get(name) == constraints.positive:
                self.unconstrained_params.append(nn.Parameter(inv_softplus(param)))
            else:
                self.unconstrained_params.append(nn.Parameter(param))
        self.mixture_logits = nn.Parameter(mixture_weights)
        self.to(device)

    def _get_constrained_params(self):
        out = []
        for i, name in enumerate(self.param_names):
            c = self.param_constraints.get(name)
            p = self.unconstrained_params[i]
            positive = isinstance(c, type(constraints.positive)) and c.lower_bound == 0.0       # (:762)
            out.append(softplus(p) if positive else p)
        return out

    def _get_distribution_params(self):
        comp = self.component_distribution_class(**dict(zip(self.param_names, self._get_constrained_params())))
        return {"mixture_distribution": torch.distributions.Categorical(logits=self.mixture_logits),
                "component_distribution": comp}


class GMM(MixtureModel):
    """Gaussian mixture with full covariances (distributions.py:798-820); the covariance is stored unconstrained."""

    def __init__(self, loc, covariance_matrix, mixture_weights, device: str = "cpu"):
        super().__init__(torch.distributions.MultivariateNormal, ["loc", "covariance_matrix"],
                         {"loc": constraints.real, "covariance_matrix": constraints.positive_definite},
                         loc, covariance_matrix, mixture_weights=mixture_weights, device=device)


class LogNormalMM(MixtureModel):
    """Mixture of log-normals (distributions.py:822-834)."""

    def __init__(self, loc, scale, mixture_weights, device="cpu"):
        super().__init__(torch.distributions.LogNormal, ["loc", "scale"], {"loc": None, "scale": constraints.positive},
                         loc, scale, mixture_weights=mixture_weights, device=device)


class WeibullMM(MixtureModel):
    """Mixture of Weibulls (distributions.py:836-850)."""

    def __init__(self, scale, concentration, mixture_weights, device="cpu"):
        super().__init__(torch.distributions.Weibull, ["scale", "concentration"],
                         {"scale": constraints.positive, "concentration": constraints.positive},
                         scale, concentration, mixture_weights=mixture_weights, device=device)


class UniformUnitLpBall(torch.distributions.Distribution):
    """Uniform distribution on the unit Lp sphere, p in {1, 2, inf} (distributions.py:254-324)."""

    support = constraints.real
    has_enumerate_support = False

    def __init__(self, dim, p: float):
        self.p = p
        self.dim = int(dim)
        d = self.dim
        if p == 1:
            self.log_surface_area_unit_ball = (1.5 * math.log(d) + math.log(2) * d
                                               - torch.log(torch.arange(1, d + 1)).sum())
        elif p == 2:
            self.log_surface_area_unit_ball = math.log(2) + (d / 2) * math.log(math.pi) - math.lgamma(d / 2)
        elif p == math.inf:
            self.log_surface_area_unit_ball = math.log(2) * d + math.log(d)
        else:
            raise ValueError("p must be 1, 2, or inf.")
        super().__init__(event_shape=(d,), validate_args=False)

    def sample(self, sample_shape: Iterable[int] = None):
        sample_shape = () if sample_shape is None else tuple(sample_shape)
        d = self.dim
        if self.p == 1:
            x = torch.distributions.Dirichlet(torch.ones(d)).sample(sample_shape)
            signs = torch.distributions.Categorical(probs=torch.ones(2) / 2).sample(sample_shape + (d,)) * 2 - 1
            return x * signs
        if self.p == 2:
            x = torch.distributions.Normal(0.0, 1.0).sample(sample_shape + (d,))
            return x / x.norm(dim=-1, keepdim=True)
        extremal = torch.distributions.Categorical(torch.ones(d) / d).sample(sample_shape + (1,))
        mask = torch.ones(sample_shape + (d,)).cumsum(dim=-1) - 1 == extremal
        boundary = torch.ones(sample_shape + (d,))
        x = torch.distributions.Uniform(-boundary, boundary).sample()
        x[mask] = 1.0
        return x

    def log_prob(self, x):
        return -self.log_surface_area_unit_ball


class RadialDistribution(nn.Module):
    """Lp-radial distribution: density depends on x only through r = ||x - loc||_p
    (distributions.py:327-549).  log p(x) = norm_dist.log_prob(r) - log dV_p(r)."""

    arg_constraints = {"loc": constraints.real}
    support = constraints.real
    has_enumerate_support = False

    def __init__(self, loc: torch.Tensor, norm_distribution, p: float, n_batch_dims: int = 0, device: str = "cpu"):
        nn.Module.__init__(self)
        self.norm_distribution = norm_distribution
        self.event_shape = loc.shape[n_batch_dims:]
        self.batch_shape = loc.shape[:n_batch_dims]
        if not isinstance(p, float):
            raise ValueError("p must be a float.")
        if p <= 0:
            raise ValueError("p must be positive.")
        self.device = device
        self.loc = nn.Parameter(loc.to(device))
        self.p = p
        self.n_batch_dims = n_batch_dims
        self.dim = torch.prod(torch.tensor(loc.shape[self.n_batch_dims:]))
        self.shape = loc.shape
        self.unit_ball_distribution = UniformUnitLpBall(self.dim, p)
        self.to(self.device)

    # ---- density --------------------------------------------------------------------------
    def log_delta_volume(self, p, r):
        d = int(self.dim)
        if p == 1:      # d/dr of (2r)^d / d!
            return math.log(2) * d + torch.log(r) * (d - 1) - sum(math.log(i) for i in range(1, d))
        if p == 2:
            return (math.log(d) + (d / 2) * math.log(math.pi) + (d - 1) * torch.log(r)) - math.lgamma(d / 2 + 1)
        if p == math.inf:
            return math.log(d) + d * math.log(2) + (d - 1) * torch.log(r)
        raise ValueError(f"p={p} not implemented. Use p=1,2, or infinity")

    def log_prob_from_radius(self, r: torch.Tensor) -> torch.Tensor:
        """finishing math on the [B] radius vector (distributions.py:506-511)"""
        return self.norm_distribution.log_prob(r.unsqueeze(-1)).squeeze(-1) - self.log_delta_volume(self.p, r)

    def log_prob(self, x):
        x = x - self.loc
        event_dims = tuple(range(x.dim() - len(self.event_shape), x.dim()))
        return self.log_prob_from_radius(x.norm(p=self.p, dim=event_dims))

    def r_profile(self, r):
        r = r.to(self.device) if isinstance(r, torch.Tensor) else torch.tensor(r, device=self.device)
        return self.log_prob_from_radius(r)

    # ---- sampling -------------------------------------------------------------------------
    def sample(self, sample_shape: Iterable[int] = None):
        peel = sample_shape is None
        sample_shape = (1,) if peel else tuple(sample_shape)
        r = self.norm_distribution.sample(sample_shape).to(self.device)
        r = r.repeat(*[1 for _ in sample_shape], *[1 for _ in range(self.n_batch_dims)], *tuple(self.event_shape))
        u = self.unit_ball_distribution.sample(sample_shape + tuple(self.batch_shape)).to(self.device)
        x = r * u.reshape(*sample_shape, *self.shape)
        if peel:
            x = x.squeeze(0)
        return x + self.loc

    # ---- UDL profiles (distributions.py:390-456) ---------------------------------------------
    def _merge_intervals(self, idx: torch.Tensor) -> torch.Tensor:
        if len(idx) == 1:
            return torch.tensor([[idx[0], idx[0]]])
        idx = idx.sort().values
        merged, start, end = [], idx[0], idx[0]
        for i in range(1, len(idx)):
            if idx[i] == end + 1:
                end = idx[i]
            else:
                merged.append([start, end])
                start = end = idx[i]
        merged.append([start, end])
        return torch.tensor(merged, device=self.device)

    def _profile(self, q, threshold, r_max, n_samples, upper: bool):
        if q is not None and threshold is not None:
            raise ValueError("Only one of 'q' or 'threshold' can be provided.")
        if q is None and threshold is None:
            raise ValueError("Either 'q' or 'threshold' must be provided.")
        rs = torch.linspace(1e-20, r_max, n_samples, device=self.device).reshape(-1, 1)
        profile = self.norm_distribution.log_prob(rs) - self.log_delta_volume(self.p, rs).flatten()
        if q is not None:
            s = self.norm_distribution.sample((n_samples,)).to(self.device)
            lp = self.norm_distribution.log_prob(s) - self.log_delta_volume(self.p, s).flatten()
            threshold = torch.sort(lp, descending=upper).values[int(n_samples * q)]
        idx = torch.arange(n_samples, device=self.device)
        idx = idx[profile > threshold] if upper else idx[profile <= threshold]
        return rs.flatten()[self._merge_intervals(idx)]

    def radial_udl_profile(self, q=None, threshold=None, r_max: float = 100000, n_samples: int = 10000):
        return self._profile(q, threshold, r_max, n_samples, upper=True)

    def radial_ldl_profile(self, q=None, threshold=None, r_max: float = 100000, n_samples: int = 10000):
        return self._profile(q, threshold, r_max, n_samples, upper=False)
