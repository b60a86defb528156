"""CPU oracle for IMAGE-SHAPED flows (``in_dims = [C, H, W]``).

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this file; ``usflows_amd`` never does.

A functional restatement, in plain torch-CPU ops, of what the reference computes for ``USFlow(in_dims=[C, H, W],
conditioner_cls=ConvNet2D, ...)``: it consumes a reference-layout state dict plus an ``ImageSpec`` and never instantiates
reference or product modules.  The parameter-only pieces (LU / Householder factors, their inverses and log-dets) and
the layer list are the flat oracle's (``usflows_oracle._AffineParams`` / ``layer_plan``: the reference shares that code
between flat and image inputs too); what is restated here is what differs for rank-3 inputs:

  * ``BlockAffineTransform`` as a 1 x 1 convolution over the channel axis, log-det counted once per pixel
    (transforms.py:904-962, 964-980);
  * the image checkerboard / channel masks (flows.py:494-536);
  * ``MaskedCoupling`` with the CNN conditioner ``ConvNet2D`` = Conv2d, num_layers x [GatedConv | Conv2d, nonlinearity,
    LayerNormChannels], Conv2d (networks.py:405-510; GatedConv :61-122 -- its own nonlinearity is always its default
    ReLU, ConvNet2D does not pass one on; LayerNormChannels :40-58);
  * ``ScaleTransform`` with a [C, H, W] scale (transforms.py:105-144) and the base density summed over all three axes
    (flows.py:97-101 ``Independent``).

Pinning: tests/test_image_oracle.py holds it against the golden vectors of the REAL reference for image flows
(tests/golden/image_*.npz: fp64 and fp32 runs, five hand-picked configurations incl. the MNIST and CIFAR ones, plus the
full 10-block CIFAR configuration)."""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List

import torch
import torch.nn.functional as F

from .usflows_oracle import _AffineParams, gammamm_log_prob, layer_plan


@dataclass
class ImageSpec:
    in_dims: List[int]
    coupling_blocks: int
    cond_args: Dict = field(default_factory=dict)     # ConvNet2D's constructor arguments (c_hidden, num_layers, kernel_size, normalize_layers, gating)
    householder: int = 1
    lu_transform: int = 1
    affine_conjugation: bool = True
    masktype: str = "checkerboard"
    negative_slope: float = 0.0                       # ConvNet2D's own nonlinearity: ReLU (0.0) or LeakyReLU(slope)
    base: str = "laplace"                             # "laplace": Laplace(0, 1) | "radial": RadialDistribution, parameters in the state dict
    radial_p: float = 1.0
    radial_norm: str = "lognormal"                    # "lognormal" (LogNormal module, distributions.py:181-197) | "gammamm" (:674-707)

    @property
    def dim(self) -> int:                             # the affine blocks act on the channel axis (transforms.py:899-902)
        return int(self.in_dims[0])


def image_mask(spec: ImageSpec, flip: int, dtype=torch.float32) -> torch.Tensor:
    """flows.py:494-514 (checkerboard: parity of the index sum over ALL axes) / :516-536 (channel: parity of the channel index),
    viewed (1, C, H, W); flows.py:472 alternates it after every block."""
    axes = [torch.arange(d, dtype=torch.int32) for d in spec.in_dims]
    idx = torch.stack(torch.meshgrid(*axes, indexing="ij"))
    par = idx.sum(dim=0) if spec.masktype == "checkerboard" else idx[0]
    m = torch.fmod(par, 2).to(dtype).view(1, *spec.in_dims)
    return 1 - m if flip else m


def _act(h, slope):
    return F.leaky_relu(h, slope) if slope != 0.0 else F.relu(h)


def convnet2d_forward(sd, prefix: str, spec: ImageSpec, x: torch.Tensor) -> torch.Tensor:
    """ConvNet2D.forward (networks.py:497-510) over its nn.Sequential (built at :448-495)."""
    a = spec.cond_args
    ks = int(a.get("kernel_size", 3))
    pad = a.get("padding", 0)
    gating, norm = bool(a.get("gating", True)), bool(a.get("normalize_layers", True))
    conv = lambda h, q, k, p: F.conv2d(h, sd[f"{prefix}{q}.weight"], sd[f"{prefix}{q}.bias"], padding=p)     # noqa: E731
    h = conv(x, "nn.0", ks, pad)
    m = 1
    for _ in range(int(a.get("num_layers", 3))):
        if gating:
            # GatedConv (networks.py:61-122): net = [ReLU, Conv2d(k), ReLU, Conv2d(1 x 1, same padding argument)]
            out = conv(F.relu(conv(F.relu(h), f"nn.{m}.net.1", ks, pad)), f"nn.{m}.net.3", 1, pad)
            val, gate = out.chunk(2, dim=1)
            h = h + val * torch.sigmoid(gate)
        else:
            h = conv(h, f"nn.{m}", ks, pad)
        h = _act(h, spec.negative_slope)               # the Sequential's `nonlinearity` entry behind either block
        m += 2
        if norm:
            # LayerNormChannels.forward (networks.py:53-58)
            mean = h.mean(dim=1, keepdim=True)
            var = h.var(dim=1, unbiased=False, keepdim=True)
            h = (h - mean) / torch.sqrt(var + 1e-5)
            h = h * sd[f"{prefix}nn.{m}.gamma"] + sd[f"{prefix}nn.{m}.beta"]
            m += 1
    return conv(h, f"nn.{m}", ks, pad)


def _affine(sd, prefix, spec, seq):
    return _AffineParams(sd, prefix, spec, seq)


def _affine_forward(ap: _AffineParams, x):
    """BlockAffineTransform.forward for rank-3 in_dims (transforms.py:913-934): F.conv2d with the [C, C, 1, 1] matrix"""
    C = x.shape[1]
    return F.conv2d(x, ap.matrix().view(C, C, 1, 1), ap.bias())


def _affine_backward(ap: _AffineParams, y):
    """transforms.py:936-962: subtract the bias, then the 1 x 1 convolution with the inverse matrix"""
    C = y.shape[1]
    return F.conv2d(y - ap.bias().view(C, 1, 1), ap.inverse_matrix().view(C, C, 1, 1))


def flow_backward(sd, spec: ImageSpec, x, return_logdet=False):
    """Flow.backward (flows.py:57-67) / the loop of Flow.log_prob (flows.py:234-243) for image inputs"""
    n_pix = math.prod(spec.in_dims[1:])
    log_det = torch.zeros(x.shape[0], dtype=x.dtype)
    for kind, prefix, flip, seq in reversed(layer_plan(spec)):
        if kind == "scale":
            s = sd[prefix + "scale"]
            y, ladj = x / s, s.abs().log().sum()
        elif kind == "affine":
            ap = _affine(sd, prefix, spec, seq)
            y, ladj = _affine_backward(ap, x), ap.ladj() * n_pix            # transforms.py:980
        elif kind == "inv_affine":
            ap = _affine(sd, prefix, spec, seq)
            y, ladj = _affine_forward(ap, x), -ap.ladj() * n_pix
        else:                                                                # MaskedCoupling.backward (transforms.py:292-306)
            mask = image_mask(spec, flip, x.dtype)
            y, ladj = x - (1 - mask) * convnet2d_forward(sd, prefix + "conditioner.", spec, x * mask), 0.0
        log_det = log_det - ladj
        x = y
    return (x, log_det) if return_logdet else x


def flow_forward(sd, spec: ImageSpec, z):
    """Flow._forward (flows.py:45-55)"""
    y = z
    for kind, prefix, flip, seq in layer_plan(spec):
        if kind == "scale":
            y = y * sd[prefix + "scale"]
        elif kind == "affine":
            y = _affine_forward(_affine(sd, prefix, spec, seq), y)
        elif kind == "inv_affine":
            y = _affine_backward(_affine(sd, prefix, spec, seq), y)
        else:                                                                # MaskedCoupling.forward (transforms.py:277-290)
            mask = image_mask(spec, flip, y.dtype)
            y = y + (1 - mask) * convnet2d_forward(sd, prefix + "conditioner.", spec, y * mask)
    return y


def laplace_log_prob(z):
    """Independent(Laplace(0, 1), 3).log_prob (flows.py:97-101; torch Laplace.log_prob)"""
    return (-math.log(2.0) - z.abs()).flatten(1).sum(-1)


def radial_log_prob(sd, spec: ImageSpec, z, prefix="base_distribution."):
    """RadialDistribution.log_prob for an image-shaped loc (distributions.py:501-511): the p-norm runs over ALL event axes
    (``event_dims = tuple(range(x.dim() - len(event_shape), x.dim()))``), then
    ``norm_distribution.log_prob(r.unsqueeze(-1)).squeeze(-1) - log_delta_volume(p, r)`` (:513-549) with dim = C*H*W.
    The norm distribution's parameters live in the state dict (they are trained: mnist.yaml:79-92)."""
    dt = z.dtype
    x = z - sd[prefix + "loc"].to(dt)
    p = spec.radial_p
    flat = x.flatten(1).abs()
    if p == 1:
        r = flat.sum(-1)
    elif p == 2:
        r = (flat * flat).sum(-1).sqrt()
    elif p == math.inf:
        r = flat.max(-1).values
    else:
        raise ValueError(p)
    if spec.radial_norm == "lognormal":
        # LogNormal module: torch LogNormal(loc, softplus(scale_unconstrained)) wrapped Independent(.., 1) (:127-139, 181-197);
        # torch LogNormal.log_prob(r) = Normal(mu, sigma).log_prob(log r) - log r  (TransformedDistribution, ExpTransform)
        mu = sd[prefix + "norm_distribution.loc"].to(dt)
        sigma = F.softplus(sd[prefix + "norm_distribution.scale_unconstrained"].to(dt))
        lr = torch.log(r).unsqueeze(-1)
        lpn = (-((lr - mu) ** 2) / (2 * sigma ** 2) - sigma.log() - math.log(math.sqrt(2 * math.pi)) - lr).sum(-1)
    elif spec.radial_norm == "gammamm":
        lpn = gammamm_log_prob(sd, r, prefix + "norm_distribution.")
    else:
        raise ValueError(spec.radial_norm)
    D = math.prod(spec.in_dims)
    if p == 1:
        log_dv = math.log(2) * D + torch.log(r) * (D - 1) - sum(math.log(i) for i in range(1, D))
    elif p == 2:
        log_dv = (math.log(D) + (D / 2) * math.log(math.pi) + (D - 1) * torch.log(r)) - math.lgamma(D / 2 + 1)
    else:
        log_dv = math.log(D) + D * math.log(2) + (D - 1) * torch.log(r)
    return lpn - log_dv


def flow_log_prob(sd, spec: ImageSpec, x):
    """Flow.log_prob (flows.py:225-245): Laplace(0, 1) base of the first golden cases, or the live configurations' radial base"""
    z, log_det = flow_backward(sd, spec, x, return_logdet=True)
    return (radial_log_prob(sd, spec, z) if spec.base == "radial" else laplace_log_prob(z)) + log_det


def total_ladj(sd, spec: ImageSpec):
    n_pix = math.prod(spec.in_dims[1:])
    t = 0.0
    for kind, prefix, flip, seq in layer_plan(spec):
        if kind == "scale":
            t = t + sd[prefix + "scale"].abs().log().sum()
        elif kind == "affine":
            t = t + _affine(sd, prefix, spec, seq).ladj() * n_pix
        elif kind == "inv_affine":
            t = t - _affine(sd, prefix, spec, seq).ladj() * n_pix
    return t
