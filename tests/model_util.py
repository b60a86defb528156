"""Build product (usflows_amd) models from an oracle FlowSpec + reference-layout state dict."""
import math

import torch

from usflows_amd.flows import USFlow
from usflows_amd.networks import ConditionalDenseNN, DenseNN
from usflows_amd import distributions as D


def make_base(spec, device="cpu"):
    n = spec.dim
    if spec.base in ("laplace", "normal"):
        loc = (spec.base_loc if spec.base_loc is not None else torch.zeros(n)).to(device)
        sc = (spec.base_scale if spec.base_scale is not None else torch.ones(n)).to(device)
        cls = torch.distributions.Laplace if spec.base == "laplace" else torch.distributions.Normal
        return cls(loc, sc)
    nd = D.LogNormal(torch.tensor([spec.radial_norm_loc]), torch.tensor([spec.radial_norm_scale]), device=device)
    loc = spec.base_loc if spec.base_loc is not None else torch.zeros(n)
    return D.RadialDistribution(loc.clone(), nd, float(spec.radial_p), device=device)


def build_flow(spec, sd=None, device="cpu"):
    act = torch.nn.LeakyReLU(spec.negative_slope) if spec.negative_slope != 0 else torch.nn.ReLU()
    if spec.conditioner == "ConditionalDenseNN":
        cls, args = ConditionalDenseNN, dict(input_dim=spec.dim, context_dim=1, hidden_dims=list(spec.hidden_dims),
                                             out_dim=spec.dim, nonlinearity=act)
    else:
        cls, args = DenseNN, dict(input_dim=spec.dim, hidden_dims=list(spec.hidden_dims), param_dims=[spec.dim],
                                  nonlinearity=act)
    prior = torch.distributions.Uniform(1e-20, 0.01) if spec.soft_training else None
    flow = USFlow(make_base(spec, device), [spec.dim], spec.coupling_blocks, cls, args, soft_training=spec.soft_training,
                  training_noise_prior=prior, affine_conjugation=spec.affine_conjugation,
                  lu_transform=spec.lu_transform, householder=spec.householder)
    if sd is not None:
        res = flow.load_state_dict(sd, strict=False)
        assert not res.unexpected_keys, res.unexpected_keys
        assert all(k.startswith("base_distribution.") for k in res.missing_keys), res.missing_keys
    if device != "cpu":
        flow = flow.to(device)
    return flow
