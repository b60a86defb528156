"""CPU oracle for the USFlows coupling-flow hot path (log_prob / backward / _forward / sample).

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this file.  ``usflows_amd`` never does; the
product path fails loudly if its HIP extension is missing instead of routing through here.

What it is: an op-for-op restatement, in plain torch-CPU tensor ops, of what the reference
computes on the path named by BASELINE.json (reference = /root/reference, aai-institute/USFlows
@2025-09-12).  It is *functional*: it consumes a reference-layout ``state_dict`` plus a small
``FlowSpec`` and never instantiates reference or product modules, so it cannot accidentally
share arithmetic with either.  Nothing under ``oracle/`` imports the product package: the case
description (``FlowSpec``) and the synthetic-parameter generator are the oracle's own
(``oracle/synth.py``; tests/test_oracle.py holds them against the product-side copies bench.py uses).
Each function cites the reference lines it follows.

Pinning: checked (a) against the reference's own known-answer tests
(tests/veriflow/transforms_test.py:5-19 Scale, :35-51 LU) restated in tests/test_oracle.py,
and (b) against golden vectors produced by importing the real reference in the build container
(tests/golden/make_golden.py -> tests/golden/*.npz; fp32 and fp64 runs of the reference).
``pyro.nn.DenseNN`` (pyro-ppl 1.8.6, third-party, not vendored) has no reference test pinning
it: DenseNN parity is *unpinned*; the canonical conditioner for golden vectors is the
reference's in-repo ``ConditionalDenseNN`` (networks.py:681-751), whose no-context arithmetic
is identical.

The oracle deliberately re-derives M, M^-1 and the log-dets from the raw parameters on every
call exactly like the reference does (transforms.py:1289-1293, 795-809, 1457-1476) -- that is
part of what the CPU baseline measures.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# spec
# --------------------------------------------------------------------------------------
from .synth import ModelSpec as FlowSpec  # noqa: E402  (the plain dataclass of a case's hyper-parameters: data, no logic)


def layer_plan(spec: FlowSpec):
    """The oracle's OWN restatement of the layer list USFlow.__init__ builds (flows.py:434-482) -- independent of the
    product's `usflows_amd.synth.layer_plan` (tests/test_oracle.py holds the two against each other).  One entry per
    element of ``Flow.layers``: (kind, state-dict prefix of its parameters, mask flip, wrapped in SequentialAffineTransform?).

    per coupling block i (flows.py:435-472):
      * lu_transform x LUTransform + (householder > 0: one HouseholderTransform) collected in ``affine_layers``; if there
        are any: BlockAffineTransform(SequentialAffineTransform(affine_layers))                     -> 'affine' (sequential)
      * MaskedCoupling(mask, conditioner)                                                           -> 'coupling'
      * affine_conjugation and a block exists: InverseTransform(block) -- shares the block's parameters, registered a
        second time under its own trainable_layers index                                            -> 'inv_affine'
      * mask = 1 - mask                                                                             (flip for the next block)
    after the blocks (flows.py:475-482): BlockAffineTransform(LUTransform) -- a bare LU, no Sequential wrapper -- and
    ScaleTransform."""
    entries = []
    n = 0                                     # running index into trainable_layers (every layer is registered there)
    block_has_affine = (spec.lu_transform + (1 if spec.householder > 0 else 0)) > 0
    for i in range(spec.coupling_blocks):
        block_idx = None
        if block_has_affine:
            block_idx = n
            entries.append(("affine", "trainable_layers.%d.block_transform." % n, None, True))
            n += 1
        entries.append(("coupling", "trainable_layers.%d." % n, i % 2, None))
        n += 1
        if spec.affine_conjugation and block_idx is not None:
            entries.append(("inv_affine", "trainable_layers.%d.block_transform." % block_idx, None, True))
            n += 1
    entries.append(("affine", "trainable_layers.%d.block_transform." % n, None, False))
    entries.append(("scale", "trainable_layers.%d." % (n + 1), None, None))
    return entries


def checkerboard_mask(dim: int, dtype=torch.float32) -> torch.Tensor:
    """flows.py:494-514 for in_dims=[dim]: fmod(index, 2) viewed (1, dim)."""
    return torch.fmod(torch.arange(dim, dtype=torch.int32), 2).to(dtype).view(1, dim)


# --------------------------------------------------------------------------------------
# parameter-only pieces (the reference recomputes these on every call)
# --------------------------------------------------------------------------------------
def lu_L(L_raw: torch.Tensor) -> torch.Tensor:
    """transforms.py:1271-1274"""
    return L_raw.tril(-1) + torch.eye(L_raw.shape[0], dtype=L_raw.dtype)


def lu_U(U_raw: torch.Tensor) -> torch.Tensor:
    """transforms.py:1276-1279"""
    return U_raw.triu()


def lu_matrix(L_raw, U_raw):
    """transforms.py:1281-1283"""
    return torch.matmul(lu_L(L_raw), lu_U(U_raw))


def lu_inverse_matrix(L_raw, U_raw):
    """transforms.py:1289-1293: inverse(L), inverse(U) separately, then U^-1 @ L^-1."""
    L_inv = torch.inverse(lu_L(L_raw))
    U_inv = torch.inverse(lu_U(U_raw))
    return torch.matmul(U_inv, L_inv)


def lu_ladj(U_raw):
    """transforms.py:1303-1320 (the diag()-free 'dU' formulation)."""
    U = lu_U(U_raw)
    d = U.shape[0]
    dU = U - U.triu(1) + (torch.ones_like(U) - torch.eye(d, dtype=U.dtype))
    return dU.abs().log().sum()


def householder_matrix(vk: torch.Tensor, w_0: torch.Tensor) -> torch.Tensor:
    """transforms.py:795-809: w_0 @ prod_k (I - 2 v v^T / v.v)."""
    d = w_0.shape[0]
    w = w_0
    for v in vk:
        w = torch.mm(w, torch.eye(d, dtype=w.dtype) - 2 * torch.ger(v, v) / torch.dot(v, v))
    return w


class _AffineParams:
    """matrix / inverse_matrix / bias / ladj of one ``BlockAffineTransform`` block."""

    def __init__(self, sd: Dict[str, torch.Tensor], prefix: str, spec: FlowSpec, sequential: bool):
        self.sd, self.prefix, self.spec, self.sequential = sd, prefix, spec, sequential

    def _parts(self):
        """list of (matrix, inverse_matrix, bias, ladj) per sub-transform, in order."""
        sd, p, spec = self.sd, self.prefix, self.spec
        out = []
        if not self.sequential:  # tail block: bare LUTransform (flows.py:475-480)
            L, U, b = sd[p + "L_raw"], sd[p + "U_raw"], sd[p + "bias_vector"]
            return [(lu_matrix(L, U), lu_inverse_matrix(L, U), b, lu_ladj(U))]
        j = 0
        for _ in range(spec.lu_transform):
            q = f"{p}transforms.{j}."
            L, U, b = sd[q + "L_raw"], sd[q + "U_raw"], sd[q + "bias_vector"]
            out.append((lu_matrix(L, U), lu_inverse_matrix(L, U), b, lu_ladj(U)))
            j += 1
        if spec.householder > 0:
            q = f"{p}transforms.{j}."
            w = householder_matrix(sd[q + "vk_householder"], sd[q + "w_0"])
            winv = w.transpose(0, 1).contiguous()           # transforms.py:864-868
            out.append((w, winv, torch.zeros(spec.dim, dtype=w.dtype), 0.0))  # :870-872, ladj :760
        return out

    def matrix(self):
        parts = self._parts()
        if not self.sequential:
            return parts[0][0]
        M = torch.eye(self.spec.dim, dtype=parts[0][0].dtype)  # transforms.py:1457-1462
        for m, _, _, _ in parts:
            M = torch.matmul(M, m)
        return M

    def inverse_matrix(self):
        parts = self._parts()
        if not self.sequential:
            return parts[0][1]
        M = torch.eye(self.spec.dim, dtype=parts[0][0].dtype)  # transforms.py:1464-1469
        for _, mi, _, _ in parts[::-1]:
            M = torch.matmul(M, mi)
        return M

    def bias(self):
        parts = self._parts()
        if not self.sequential:
            return parts[0][2]
        b = torch.zeros(self.spec.dim, dtype=parts[0][0].dtype)  # transforms.py:1471-1476
        for m, _, bi, _ in parts:
            b = torch.matmul(b, m) + bi
        return b

    def ladj(self):
        # transforms.py:1444-1446: the sum of the sub-transforms' log-dets -- LUTransform: lu_ladj(U) (:1303-1320), Householder: 0
        # (:760).  (Not through _parts(): the matrices and their inverses are not needed for it -- 49 fp64 inverses of 3072 x 3072
        # at cfg4 size.)
        sd, p, spec = self.sd, self.prefix, self.spec
        if not self.sequential:
            return lu_ladj(sd[p + "U_raw"])
        return sum(lu_ladj(sd[f"{p}transforms.{j}.U_raw"]) for j in range(spec.lu_transform))

    # BlockAffineTransform for rank-1 in_dims: F.linear (transforms.py:904-934, 936-962)
    def forward(self, x):
        return F.linear(x, self.matrix(), self.bias())

    def backward(self, y):
        w = self.inverse_matrix()
        b = self.bias()
        return F.linear(y - b, w)


# --------------------------------------------------------------------------------------
# conditioner + coupling
# --------------------------------------------------------------------------------------
def _act(h, slope):
    return F.leaky_relu(h, slope) if slope != 0.0 else F.relu(h)


def conditioner_forward(sd, prefix, spec: FlowSpec, x, context=None):
    """ConditionalDenseNN.forward (networks.py:739-751) / pyro DenseNN (see ref_shim.py)."""
    n_hidden = len(spec.hidden_dims)
    if spec.conditioner == "ConditionalDenseNN":
        # layers[0]: input, layers[1]: context, layers[2:-1]: hidden, layers[-1]: output
        h = F.linear(x, sd[f"{prefix}layers.0.weight"], sd[f"{prefix}layers.0.bias"])
        if context is not None:
            h = h + F.linear(context, sd[f"{prefix}layers.1.weight"], sd[f"{prefix}layers.1.bias"])
        h = _act(h, spec.negative_slope)
        idx = 2
        for _ in range(n_hidden - 1):
            h = _act(F.linear(h, sd[f"{prefix}layers.{idx}.weight"], sd[f"{prefix}layers.{idx}.bias"]),
                     spec.negative_slope)
            idx += 1
        return F.linear(h, sd[f"{prefix}layers.{idx}.weight"], sd[f"{prefix}layers.{idx}.bias"])
    elif spec.conditioner == "DenseNN":
        h = x
        for idx in range(n_hidden):
            h = _act(F.linear(h, sd[f"{prefix}layers.{idx}.weight"], sd[f"{prefix}layers.{idx}.bias"]),
                     spec.negative_slope)
        return F.linear(h, sd[f"{prefix}layers.{n_hidden}.weight"], sd[f"{prefix}layers.{n_hidden}.bias"])
    elif spec.conditioner == "ConvNet":
        # vector path of ConvNet (networks.py:287-308, forward 379-389); plain form (gating=False, normalize_layers=False):
        # nn.0 = Linear(D, h0); nn.{i+1} = Sequential(f, Linear(prev, c_hidden[i])); nn.{n+1} = Linear(c_hidden[-1], D)
        # -- the activation sits IN FRONT of every block's Linear, none in front of the final Linear
        # with gating (spec.extra["gating"]): block = GatedMLP (networks.py:222-245): [val, gate] = Linear(f(Linear(f(h)))),
        # h <- (h or proj(h)) + val * sigmoid(gate); with spec.extra["normalize_layers"]: LayerNormVector (networks.py:206-219)
        # after every block
        gating, norm = bool(spec.extra.get("gating", False)), bool(spec.extra.get("normalize_layers", False))
        lin = lambda h_, q: F.linear(h_, sd[f"{prefix}{q}.weight"], sd[f"{prefix}{q}.bias"])      # noqa: E731
        h = lin(x, "nn.0")
        m = 1
        for i in range(n_hidden):
            if gating:
                vg = lin(_act(lin(_act(h, spec.negative_slope), f"nn.{m}.net1.1"), spec.negative_slope), f"nn.{m}.net1.3")
                val, gate = vg.chunk(2, dim=1)
                skip = lin(h, f"nn.{m}.proj") if f"{prefix}nn.{m}.proj.weight" in sd else h
                h = skip + val * torch.sigmoid(gate)
            else:
                h = lin(_act(h, spec.negative_slope), f"nn.{m}.1")
            m += 1
            if norm:
                h = F.layer_norm(h, (h.shape[1],), sd[f"{prefix}nn.{m}.layernorm.weight"], sd[f"{prefix}nn.{m}.layernorm.bias"], 1e-5)
                m += 1
        return lin(h, f"nn.{m}")
    raise ValueError(spec.conditioner)


def coupling_forward(sd, prefix, spec, mask, x, context=None):
    """MaskedCoupling.forward, transforms.py:277-290 (additive; ladj == 0.0, :316-326)."""
    x_masked = x * mask
    return x + (1 - mask) * conditioner_forward(sd, prefix + "conditioner.", spec, x_masked, context)


def coupling_backward(sd, prefix, spec, mask, y, context=None):
    """MaskedCoupling.backward, transforms.py:292-306."""
    y_masked = y * mask
    return y - (1 - mask) * conditioner_forward(sd, prefix + "conditioner.", spec, y_masked, context)


# --------------------------------------------------------------------------------------
# layer list (USFlow.__init__, flows.py:434-482): layer_plan above
# --------------------------------------------------------------------------------------
def _mask_for(spec, flip, dtype):
    m = checkerboard_mask(spec.dim, dtype)
    return 1 - m if flip else m            # flows.py:472 alternates after every block


# --------------------------------------------------------------------------------------
# base distributions (distributions.py:709-728 wrapper; torch Laplace/Normal; Radial 501-549)
# --------------------------------------------------------------------------------------
def gammamm_log_prob(sd, r: torch.Tensor, prefix="base_distribution.norm_distribution.") -> torch.Tensor:
    """GammaMM.log_prob (distributions.py:674-707) = torch MixtureSameFamily over Gamma components:
    logsumexp_k( log_softmax(logits)_k + Gamma(softplus(c_k), softplus(b_k)).log_prob(r) )"""
    dt = r.dtype
    c = F.softplus(sd[prefix + "concentration_unconstrained"].to(dt))
    b = F.softplus(sd[prefix + "rate_unconstrained"].to(dt))
    logw = torch.log_softmax(sd[prefix + "mixture_logits"].to(dt), dim=-1)
    rr = r.unsqueeze(-1)
    comp = c * torch.log(b) + (c - 1) * torch.log(rr) - b * rr - torch.lgamma(c)      # torch Gamma.log_prob
    return torch.logsumexp(comp + logw, dim=-1)


def base_log_prob(spec: FlowSpec, z: torch.Tensor, sd=None) -> torch.Tensor:
    dt = z.dtype
    D = spec.dim
    if spec.base in ("laplace", "normal"):
        loc = (spec.base_loc if spec.base_loc is not None else torch.zeros(D)).to(dt)
        scale = (spec.base_scale if spec.base_scale is not None else torch.ones(D)).to(dt)
        if spec.base == "laplace":          # torch Laplace.log_prob
            lp = -torch.log(2 * scale) - torch.abs(z - loc) / scale
        else:                                # torch Normal.log_prob
            var = scale ** 2
            lp = -((z - loc) ** 2) / (2 * var) - scale.log() - math.log(math.sqrt(2 * math.pi))
        return lp.sum(-1)                    # Independent(..., 1)
    if spec.base == "radial":               # distributions.py:501-549
        loc = (spec.base_loc if spec.base_loc is not None else torch.zeros(D)).to(dt)
        x = z - loc
        r = x.norm(p=spec.radial_p, dim=(-1,))
        if spec.radial_norm == "gammamm":
            if sd is None:
                raise ValueError("a GammaMM norm distribution needs the state dict (its parameters live there)")
            log_prob_norm = gammamm_log_prob(sd, r)
        elif spec.radial_norm == "lognormal":
            nd = torch.distributions.LogNormal(torch.tensor([spec.radial_norm_loc], dtype=dt),
                                               torch.tensor([spec.radial_norm_scale], dtype=dt))
            nd = torch.distributions.Independent(nd, 1)     # DistributionModule.distribution :131-139
            log_prob_norm = nd.log_prob(r.unsqueeze(-1)).squeeze(-1)
        else:
            raise ValueError(spec.radial_norm)
        p = spec.radial_p
        if p == 1:
            log_den = sum(math.log(i) for i in range(1, D))
            log_dv = math.log(2) * D + torch.log(r) * (D - 1) - log_den
        elif p == 2:
            log_dv = (math.log(D) + (D / 2) * math.log(math.pi) + (D - 1) * torch.log(r)) - math.lgamma(D / 2 + 1)
        elif p == math.inf:
            log_dv = math.log(D) + D * math.log(2) + (D - 1) * torch.log(r)
        else:
            raise ValueError(p)
        return log_prob_norm - log_dv
    raise ValueError(spec.base)


# --------------------------------------------------------------------------------------
# the three entry points of the path
# --------------------------------------------------------------------------------------
def flow_backward(sd, spec: FlowSpec, x, context=None, return_logdet=False):
    """Flow.backward (flows.py:57-67) / the loop of Flow.log_prob (flows.py:234-243)."""
    dt = x.dtype
    log_det = torch.zeros(x.shape[0], dtype=dt)
    for kind, prefix, flip, seq in reversed(layer_plan(spec)):
        if kind == "scale":
            s = sd[prefix + "scale"]
            y = x / s                                     # transforms.py:116-125
            ladj = s.abs().log().sum()                    # :135-144
        elif kind == "affine":
            ap = _AffineParams(sd, prefix, spec, seq)
            y = ap.backward(x)
            ladj = ap.ladj()                              # x n_blocks == 1 (transforms.py:980)
        elif kind == "inv_affine":                        # InverseTransform: :370-376, :386-396
            ap = _AffineParams(sd, prefix, spec, seq)
            y = ap.forward(x)
            ladj = -ap.ladj()
        elif kind == "coupling":
            y = coupling_backward(sd, prefix, spec, _mask_for(spec, flip, dt), x, context)
            ladj = 0.0
        log_det = log_det - ladj                          # flows.py:242
        x = y
    return (x, log_det) if return_logdet else x


def flow_log_prob(sd, spec: FlowSpec, x, context=None):
    """Flow.log_prob (flows.py:225-245) incl. USFlow.log_prob's implicit zero context
    for soft-trained models (flows.py:559-565)."""
    if spec.soft_training and context is None:
        context = torch.zeros(x.shape[0], dtype=x.dtype).unsqueeze(-1)
    z, log_det = flow_backward(sd, spec, x, context, return_logdet=True)
    return base_log_prob(spec, z, sd) + log_det


def flow_forward(sd, spec: FlowSpec, z, context=None):
    """Flow._forward (flows.py:45-55) == the layer loop of Flow.sample (flows.py:259-263)."""
    y = z
    for kind, prefix, flip, seq in layer_plan(spec):
        if kind == "scale":
            y = y * sd[prefix + "scale"]                  # transforms.py:105-114
        elif kind == "affine":
            y = _AffineParams(sd, prefix, spec, seq).forward(y)
        elif kind == "inv_affine":
            y = _AffineParams(sd, prefix, spec, seq).backward(y)
        elif kind == "coupling":
            y = coupling_forward(sd, prefix, spec, _mask_for(spec, flip, y.dtype), y, context)
    return y


def total_ladj(sd, spec: FlowSpec):
    """sum of the (parameter-only) layer log-dets; log_prob = base(z) - total_ladj."""
    t = 0.0
    for kind, prefix, flip, seq in layer_plan(spec):
        if kind == "scale":
            t = t + sd[prefix + "scale"].abs().log().sum()
        elif kind == "affine":
            t = t + _AffineParams(sd, prefix, spec, seq).ladj()
        elif kind == "inv_affine":
            t = t - _AffineParams(sd, prefix, spec, seq).ladj()
    return t


def laplace_icdf_sample(u: torch.Tensor, loc, scale):
    """torch Laplace.rsample given its uniform draw u in (eps-1, 1):
    loc - scale * sign(u) * log1p(-|u|)."""
    return loc - scale * u.sign() * torch.log1p(-u.abs())


def to_dtype(sd: Dict[str, torch.Tensor], dtype) -> Dict[str, torch.Tensor]:
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}


# --------------------------------------------------------------------------------------
# deterministic synthetic parameters: pure data generation (no flow arithmetic), the oracle's own
# copy (oracle/synth.py), re-exported here for the tests
# --------------------------------------------------------------------------------------
from .synth import synth_state_dict  # noqa: E402,F401
