"""Deterministic synthetic parameters for the oracle's cases (TEST INFRASTRUCTURE, like everything under oracle/).

The oracle's OWN copy of the case description (``ModelSpec``) and of the synthetic-parameter generator -- the product
package carries one for ``bench.py`` / ``smoke()`` (``usflows_amd/synth.py``); nothing under ``oracle/`` imports the
product, and ``tests/test_oracle.py`` holds the two generators against each other bit for bit.  Pure data generation, no
flow arithmetic: a reference-layout state dict drawn from the reference's init *distributions* followed by the
documented conditioning transform of SURVEY.md section 7-H2 (the reference's default init explodes at depth)."""
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
    from .usflows_oracle import layer_plan          # the oracle's own restatement of USFlow.__init__'s layer list
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


