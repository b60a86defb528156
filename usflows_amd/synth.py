"""Deterministic synthetic USFlow models for benchmarks, smoke tests and parity fixtures.

Pure data generation -- no flow arithmetic: (1) ``ModelSpec``: the constructor arguments of a reference
``USFlow`` that are not in its state dict; (2) ``synth_state_dict``: a reference-layout state dict drawn
from the reference's init *distributions* followed by the documented conditioning transform of
SURVEY.md section 7-H2 (the reference's default init explodes at depth: |z| ~ 7e22 at D=784, K=32);
(3) ``build_usflow``: a ``usflows_amd.flows.USFlow`` built from a spec (+ state dict).
The CPU oracle has an independent twin of (1) and (2) (``oracle/synth.py``; ``tests/test_oracle.py`` holds the two
against each other), so that oracle, reference fixtures and the device path all see the same parameters without the
oracle importing the product."""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence

import torch


@dataclass
class ModelSpec:
    """Everything about a reference ``USFlow`` that is not in its state dict."""

    dim: int                              # in_dims=[dim] (flat inputs only: SURVEY section 8a)
    coupling_blocks: int
    hidden_dims: Sequence[int]
    lu_transform: int = 1                 # flows.py:401
    householder: int = 1                  # flows.py:402 (ctor default)
    affine_conjugation: bool = False      # flows.py:399
    negative_slope: float = 0.01          # LeakyReLU slope; 0.0 == ReLU
    conditioner: str = "ConditionalDenseNN"   # or "DenseNN" (pyro layout: no context layer) or "ConvNet" (vector path, plain)
    base: str = "laplace"                 # "laplace" | "normal" | "radial"
    base_loc: Optional[torch.Tensor] = None
    base_scale: Optional[torch.Tensor] = None
    radial_p: float = 1.0                 # RadialDistribution p (1, 2, inf)
    radial_norm: str = "lognormal"        # norm_distribution family: "lognormal" | "gammamm" (GammaMM: parameters in the
    #                                       state dict under base_distribution.norm_distribution.*; extra["gammamm_k"] components)
    radial_norm_loc: float = 0.0
    radial_norm_scale: float = 1.0        # (already soft-plussed) sigma
    soft_training: bool = False
    extra: dict = field(default_factory=dict)


def layer_plan(spec: ModelSpec):
    """[(kind, trainable_layers prefix, mask flip, sequential?)] in ``Flow.layers`` order (USFlow.__init__,
    flows.py:434-482).  kinds: 'affine' (BlockAffineTransform), 'coupling', 'inv_affine' (InverseTransform
    sharing the block's parameters), 'scale'."""
    has_affine = spec.lu_transform > 0 or spec.householder > 0
    plan = []
    idx = 0
    for i in range(spec.coupling_blocks):
        a_idx = None
        if has_affine:
            a_idx = idx
            plan.append(("affine", f"trainable_layers.{idx}.block_transform.", None, True))
            idx += 1
        plan.append(("coupling", f"trainable_layers.{idx}.", i % 2, None))
        idx += 1
        if spec.affine_conjugation and has_affine:
            plan.append(("inv_affine", f"trainable_layers.{a_idx}.block_transform.", None, True))
            idx += 1
    plan.append(("affine", f"trainable_layers.{idx}.block_transform.", None, False))
    idx += 1
    plan.append(("scale", f"trainable_layers.{idx}.", None, None))
    return plan


def synth_state_dict(spec: ModelSpec, seed: int = 0, alpha: float = 0.1) -> Dict[str, torch.Tensor]:
    """Reference-layout state dict with the reference's init *distributions* followed by the
    documented conditioning transform (L <- I + alpha*tril(L,-1); U <- alpha*triu(U,1) +
    diag(sign*U[0.75,1.25]); scale <- sign*U[0.5,1.5]).  The default init of the reference
    explodes at depth (|z| ~ 7e22 at D=784,K=32); these parameters keep |z| = O(10)."""
    g = torch.Generator().manual_seed(seed)
    D = spec.dim
    sd: Dict[str, torch.Tensor] = {}

    def ku(shape, fan_in, gain=math.sqrt(2.0)):      # kaiming_uniform_(nonlinearity="relu")
        bound = gain * math.sqrt(3.0 / fan_in)
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    def lu_params(prefix):
        L = torch.eye(D) + alpha * ku((D, D), D).tril(-1)
        sign = torch.where(torch.rand(D, generator=g) < 0.5, -1.0, 1.0)
        diag = sign * (0.75 + 0.5 * torch.rand(D, generator=g))
        U = alpha * ku((D, D), D).triu(1) + torch.diag(diag)
        sd[prefix + "L_raw"] = L
        sd[prefix + "U_raw"] = U
        sd[prefix + "bias_vector"] = (torch.rand(D, generator=g) * 2 - 1) / math.sqrt(D)

    def linear(prefix, out_f, in_f):                  # nn.Linear default init
        bound = 1.0 / math.sqrt(in_f)
        sd[prefix + "weight"] = (torch.rand(out_f, in_f, generator=g) * 2 - 1) * bound
        sd[prefix + "bias"] = (torch.rand(out_f, generator=g) * 2 - 1) * bound

    done = set()
    for kind, prefix, flip, seq in layer_plan(spec):
        if prefix in done:
            continue
        done.add(prefix)
        if kind in ("affine", "inv_affine"):
            if not seq:
                lu_params(prefix)
                continue
            j = 0
            for _ in range(spec.lu_transform):
                lu_params(f"{prefix}transforms.{j}.")
                j += 1
            if spec.householder > 0:
                q = f"{prefix}transforms.{j}."
                sd[q + "vk_householder"] = 0.2 * torch.randn(spec.householder, D, generator=g)
                w = torch.zeros(D, D)
                w[torch.arange(D), torch.randperm(D, generator=g)] = 1.0
                sd[q + "w_0"] = w
        elif kind == "coupling":
            c = prefix + "conditioner."
            hs = list(spec.hidden_dims)
            if spec.conditioner == "ConditionalDenseNN":
                linear(c + "layers.0.", hs[0], D)
                linear(c + "layers.1.", hs[0], 1)
                idx = 2
                for i in range(1, len(hs)):
                    linear(c + f"layers.{idx}.", hs[i], hs[i - 1])
                    idx += 1
                linear(c + f"layers.{idx}.", D, hs[-1])
            elif spec.conditioner == "ConvNet":     # vector path; extra["gating"] / extra["normalize_layers"] (default off)
                gating, norm = bool(spec.extra.get("gating", False)), bool(spec.extra.get("normalize_layers", False))
                linear(c + "nn.0.", hs[0], D)
                width, m = hs[0], 1
                for hdim in hs:
                    if gating:          # GatedMLP (networks.py:222-245): net1 = [f, Linear, f, Linear], proj if widths differ
                        linear(c + f"nn.{m}.net1.1.", hdim, width)
                        linear(c + f"nn.{m}.net1.3.", 2 * hdim, hdim)
                        if width != hdim:
                            linear(c + f"nn.{m}.proj.", hdim, width)
                    else:
                        linear(c + f"nn.{m}.1.", hdim, width)
                    m += 1
                    if norm:            # LayerNormVector (networks.py:206-219): gain around 1, small offset
                        sd[c + f"nn.{m}.layernorm.weight"] = 0.75 + 0.5 * torch.rand(hdim, generator=g)
                        sd[c + f"nn.{m}.layernorm.bias"] = 0.2 * (torch.rand(hdim, generator=g) * 2 - 1)
                        m += 1
                    width = hdim
                linear(c + f"nn.{m}.", D, width)
            else:
                linear(c + "layers.0.", hs[0], D)
                for i in range(1, len(hs)):
                    linear(c + f"layers.{i}.", hs[i], hs[i - 1])
                linear(c + f"layers.{len(hs)}.", D, hs[-1])
        elif kind == "scale":
            sign = torch.where(torch.rand(D, generator=g) < 0.5, -1.0, 1.0)
            sd[prefix + "scale"] = sign * (0.5 + torch.rand(D, generator=g))
    # InverseTransform aliases (same tensors under '<idx>.transform.block_transform.')
    if spec.affine_conjugation:
        idx = 0
        for i in range(spec.coupling_blocks):
            a = idx
            inv = idx + 2
            for k in [k for k in sd if k.startswith(f"trainable_layers.{a}.block_transform.")]:
                sd[k.replace(f"trainable_layers.{a}.", f"trainable_layers.{inv}.transform.")] = sd[k]
            idx += 3
    return sd


def make_base(spec: ModelSpec, device="cpu"):
    from . import distributions as D
    n = spec.dim
    if spec.base in ("laplace", "normal"):
        loc = (spec.base_loc if spec.base_loc is not None else torch.zeros(n)).to(device)
        sc = (spec.base_scale if spec.base_scale is not None else torch.ones(n)).to(device)
        if spec.extra.get("trainable_base"):
            # the reference's TRAINABLE base modules (distributions.py:199-238): loc / softplus-constrained scale as nn.Parameters
            # (their values travel in the state dict); "scalar_scale": a 0-dim scale (Normal's expand path, distributions.py:231-234)
            if spec.extra.get("trainable_base") == "scalar_scale":
                sc = sc.flatten()[0].clone()
            return (D.Laplace if spec.base == "laplace" else D.Normal)(loc.clone(), sc.clone(), device=device)
        cls = torch.distributions.Laplace if spec.base == "laplace" else torch.distributions.Normal
        return cls(loc, sc)
    if spec.radial_norm == "gammamm":
        # the norm distribution of the live configs (gaussian_mixture.yaml:84-93): its parameters travel in the state dict
        k = int(spec.extra.get("gammamm_k", 4))
        nd = D.GammaMM(torch.linspace(2.0, 6.0, k), torch.ones(k), torch.ones(k) / k, device=device)
    else:
        nd = D.LogNormal(torch.tensor([spec.radial_norm_loc]), torch.tensor([spec.radial_norm_scale]), device=device)
    loc = spec.base_loc if spec.base_loc is not None else torch.zeros(n)
    return D.RadialDistribution(loc.clone(), nd, float(spec.radial_p), device=device)


def build_usflow(spec: ModelSpec, sd: Optional[Dict[str, torch.Tensor]] = None, device="cpu"):
    """usflows_amd.flows.USFlow for ``spec`` (optionally loaded with a reference-layout state dict)."""
    from .flows import USFlow
    from .networks import ConditionalDenseNN, DenseNN
    act = torch.nn.LeakyReLU(spec.negative_slope) if spec.negative_slope != 0 else torch.nn.ReLU()
    if spec.conditioner == "ConditionalDenseNN":
        cls, args = ConditionalDenseNN, dict(input_dim=spec.dim, context_dim=1, hidden_dims=list(spec.hidden_dims),
                                             out_dim=spec.dim, nonlinearity=act)
    elif spec.conditioner == "ConvNet":
        from .networks import ConvNet
        cls, args = ConvNet, dict(in_dims=[spec.dim], c_hidden=list(spec.hidden_dims), nonlinearity=act,
                                  normalize_layers=bool(spec.extra.get("normalize_layers", False)),
                                  gating=bool(spec.extra.get("gating", False)))
    else:
        cls, args = DenseNN, dict(input_dim=spec.dim, hidden_dims=list(spec.hidden_dims), param_dims=[spec.dim],
                                  nonlinearity=act)
    prior = torch.distributions.Uniform(1e-20, 0.01) if spec.soft_training else None
    flow = USFlow(make_base(spec, device), [spec.dim], spec.coupling_blocks, cls, args, soft_training=spec.soft_training,
                  training_noise_prior=prior, affine_conjugation=spec.affine_conjugation,
                  lu_transform=spec.lu_transform, householder=spec.householder,
                  prior_scale=spec.extra.get("prior_scale"))
    if sd is not None:
        res = flow.load_state_dict(sd, strict=False)
        assert not res.unexpected_keys, res.unexpected_keys
        assert all(k.startswith("base_distribution.") for k in res.missing_keys), res.missing_keys
    if device != "cpu":
        flow = flow.to(device)
    return flow
