"""Deterministic, well-conditioned parameters for image-shaped flows (test infrastructure).

``synth_image_params_(flow, seed)`` fills every parameter of a flow IN PLACE from (seed, position in
``named_parameters()``, name suffix).  It only touches ``nn.Module`` plumbing, so the same call conditions the real
reference's ``USFlow`` (tests/golden/make_golden_image.py, this container only) and the mirror (the tests): a golden
case made this way stores inputs and outputs but no state dict -- which is what lets the full 10-block CIFAR
configuration (experiments/cifar/cifar.yaml:56-77; 5.5 MB of parameters) be a fixture of a few KB.  The rules follow
SURVEY 7-H2 (the default initialisation explodes): unit-diagonal-ish LU factors, scales in +-[0.5, 1.5]."""
import math

import torch


def synth_image_params_(flow, seed: int, alpha: float = 0.3):
    with torch.no_grad():
        for i, (name, p) in enumerate(flow.named_parameters()):
            if name.startswith("base_distribution."):
                continue        # (a trainable base -- RadialDistribution and its norm distribution -- keeps its constructor values)
            g = torch.Generator().manual_seed(1_000_003 * seed + i)
            u = lambda *shape: torch.rand(*shape, generator=g, dtype=torch.float64) * 2 - 1           # noqa: E731  U(-1, 1)
            leaf = name.rsplit(".", 1)[-1]
            if leaf == "L_raw":
                D = p.shape[0]
                v = torch.eye(D, dtype=torch.float64) + alpha * (u(D, D) / math.sqrt(D)).tril(-1)
            elif leaf == "U_raw":
                D = p.shape[0]
                sign = torch.where(u(D) < 0, -1.0, 1.0).double()
                v = alpha * (u(D, D) / math.sqrt(D)).triu(1) + torch.diag(sign * (1.0 + 0.25 * u(D)))
            elif leaf == "bias_vector":
                v = 0.1 * u(*p.shape)
            elif leaf == "vk_householder":
                v = torch.randn(*p.shape, generator=g, dtype=torch.float64)
            elif leaf == "w_0":
                D = p.shape[0]
                v = torch.eye(D, dtype=torch.float64)[torch.randperm(D, generator=g)]
            elif leaf == "scale":
                sign = torch.where(u(*p.shape) < 0, -1.0, 1.0).double()
                v = sign * (1.0 + 0.5 * u(*p.shape))
            elif leaf == "gamma":
                v = 1.0 + 0.2 * u(*p.shape)
            elif leaf == "beta":
                v = 0.1 * u(*p.shape)
            elif leaf == "weight":
                fan_in = max(int(p[0].numel()), 1)
                v = u(*p.shape) / math.sqrt(fan_in)
            elif leaf == "bias":
                v = 0.1 * u(*p.shape)
            else:
                raise KeyError(f"synth_image_params_: no rule for parameter {name!r} {tuple(p.shape)}")
            p.copy_(v.to(p.dtype))
    return flow
