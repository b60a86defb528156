"""Bijective layers of the USFlows hot path -- host-side mirror of the reference's
``src/usflows/transforms.py`` (same class names, constructor signatures, parameter names and
state-dict keys) with the device arithmetic routed to hand-written gfx950 kernels.

Dispatch rule (every layer, every call):
  * tensors on a ROCm device and no autograd graph needed (``torch.no_grad()`` or nothing
    requires grad)  ->  HIP kernels through ``usflows_amd.engine`` / the C ABI.  If the HIP
    library is missing this raises; there is no eager fallback on that branch.
  * autograd needed (``Flow.fit``) or CPU tensors  ->  the differentiable *composite*
    formulation below, written with torch ops exactly as the math in the reference reads.
    (HIP backward kernels are SURVEY row N2, "next".)

Flat inputs (``in_dims=[D]``) are the accelerated hot path.  Image-shaped inputs (row N4, first
slice): ``BlockAffineTransform`` for ``in_dims=[C, *spatial]`` -- the reference's 1x1 convolution over the channel axis
-- runs on ``usf_channel_affine_f32`` for rank-3 ``in_dims`` on a ROCm device; masks, scale layer and CNN conditioners
of such flows run as torch ops.
"""
from __future__ import annotations

import math
import os
from typing import Any, Iterable, List, Optional

import torch

from .config import config  # noqa: E402
from torch import nn
from torch.distributions import constraints
from torch.nn import functional as F
from torch.nn import init


class TransformModule(torch.distributions.Transform, nn.Module):
    """``torch.distributions.Transform`` that is also an ``nn.Module`` (what the reference takes
    from ``pyro.distributions.TransformModule``; re-stated here so pyro is not a dependency)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)

    def __hash__(self):
        return nn.Module.__hash__(self)


def _needs_grad(module: nn.Module, *tensors) -> bool:
    if not torch.is_grad_enabled():
        return False
    if any(t is not None and torch.is_tensor(t) and t.requires_grad for t in tensors):
        return True
    return any(p.requires_grad for p in module.parameters())


def use_hip(module: nn.Module, x: torch.Tensor, *more) -> bool:
    """True when this call must run on the HIP kernels."""
    return x.is_cuda and x.dtype == torch.float32 and not _needs_grad(module, x, *more)


class BaseTransform(TransformModule):
    """Layer protocol of the reference (transforms.py:23-69): forward / backward /
    log_abs_det_jacobian (+ feasibility helpers)."""

    bijective = True
    domain = constraints.real_vector
    codomain = constraints.real_vector

    def __init__(self, *args, **kwargs):
        # the reference skips TransformModule.__init__ and initialises nn.Module + Transform
        nn.Module.__init__(self)
        self._cache_size = 0
        self._inv = None
        self._engine = None

    def is_feasible(self) -> bool:
        return True

    def jitter(self, jitter: float = 1e-6) -> None:
        pass

    def add_jitter(self, jitter: float = 1e-6) -> None:
        pass

    def forward(self, x: torch.Tensor, context: Optional[torch.Tensor] = None) -> torch.Tensor:
        raise NotImplementedError()

    def backward(self, y: torch.Tensor, context: Optional[torch.Tensor] = None) -> torch.Tensor:
        raise NotImplementedError()

    def log_abs_det_jacobian(self, x, y, context=None):
        raise NotImplementedError()

    def _call(self, x):
        return self.forward(x)

    def _inverse(self, y):
        return self.backward(y)

    def log_prior(self):
        return 0.0

    def simplify(self):
        return self

    def sign(self):
        return 1

    # -- HIP dispatch for a stand-alone layer call: a one-layer engine -------------------------
    def _hip(self, direction: str, x: torch.Tensor, context=None) -> torch.Tensor:
        from .engine import FlowEngine
        if self._engine is None:
            self._engine = FlowEngine([self])
        return self._engine.transform(x, direction, context)


class ScaleTransform(BaseTransform):
    """y = scale * x  (transforms.py:73-171)."""

    def __init__(self, in_dims, prior_scale: float = 1.0, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.in_dims = in_dims
        self.prior_scale = prior_scale
        self.dim = math.prod(in_dims) if isinstance(in_dims, Iterable) else in_dims
        self.scale = nn.Parameter(torch.empty(in_dims))
        self.init_params()

    def init_params(self):
        bound = 1 / math.sqrt(self.dim) if self.dim > 0 else 0
        init.uniform_(self.scale, -bound, bound)

    def _image_hip(self, x, divide: bool):
        """image-shaped inputs on the device: usf_scale_f32 over the flattened rows (one HBM-bound pass; the layer loop of an
        image flow is then HIP calls only and can run as one op list, flows.py); None: not applicable"""
        if not (x.dim() == self.scale.dim() + 1 and x.dim() >= 3 and tuple(x.shape[1:]) == tuple(self.scale.shape)
                and x.shape[0] > 0 and use_hip(self, x) and x.is_contiguous()):
            return None
        from . import _ext
        key = (self.scale.data_ptr(), self.scale._version, str(x.device))
        cache = getattr(self, "_scale_flat", None)
        if cache is None or cache[0] != key:
            cache = self._scale_flat = (key, self.scale.detach().to(device=x.device, dtype=torch.float32).reshape(-1).contiguous())
        B, D = x.shape[0], cache[1].numel()
        y = torch.empty_like(x)
        _ext.scale(x, D, y, D, B, D, cache[1], divide)
        return y

    def forward(self, x, context=None):
        if x.dim() == 2 and self.scale.dim() == 1 and use_hip(self, x):
            return self._hip("forward", x)
        y = self._image_hip(x, False)
        return y if y is not None else x * self.scale

    def backward(self, x, context=None):
        if x.dim() == 2 and self.scale.dim() == 1 and use_hip(self, x):
            return self._hip("backward", x)
        y = self._image_hip(x, True)
        return y if y is not None else x / self.scale

    def log_abs_det_jacobian(self, x, y, context=None):
        return self.scale.abs().log().sum()

    def sign(self) -> int:
        return 1 if (self.scale < 0).int().sum() % 2 == 0 else -1

    def is_feasible(self) -> bool:
        return (self.scale != 0).all()

    def add_jitter(self, jitter: float = 1e-6) -> None:
        # (the reference's version refers to a non-existent U_raw, transforms.py:154-157)
        with torch.no_grad():
            self.scale.add_(torch.randn(self.scale.shape, device=self.scale.device) * jitter)

    def log_prior(self):
        return 0


class MaskedCoupling(BaseTransform):
    """Additive coupling  y = x + (1-mask) * conditioner(x*mask)  (transforms.py:254-347).
    ``log_abs_det_jacobian`` is the python float 0.0: the Jacobian is unit-triangular."""

    def __init__(self, mask: torch.Tensor, conditioner: nn.Module, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.mask = mask
        self.conditioner = conditioner
        self.input_shape = mask.shape

    def _hip_ok(self, x, context) -> bool:
        from .engine import conditioner_supported
        return x.dim() == 2 and self.mask.dim() == 2 and conditioner_supported(self.conditioner) \
            and use_hip(self, x, context)

    def _image_residual(self, x, t, sign):
        """x + sign * (1 - mask) * t for image-shaped inputs on the device: one ``usf_masked_residual_f32`` pass"""
        if not (x.dim() >= 3 and torch.is_tensor(t) and t.shape == x.shape and t.dtype == torch.float32
                and self.mask.numel() == math.prod(x.shape[1:]) and use_hip(self, x, t)):
            return None
        from . import _ext
        return _ext.masked_residual(x.contiguous(), t.contiguous(), self._one_minus_mask(x), sign)

    def _one_minus_mask(self, x):
        key = (self.mask.data_ptr(), self.mask._version, str(x.device))
        cache = getattr(self, "_om_cache", None)
        if cache is None or cache[0] != key:
            om = (1 - self.mask).to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
            cache = self._om_cache = (key, om)
        return cache[1]

    def _mask_flat(self, x):
        key = (self.mask.data_ptr(), self.mask._version, str(x.device))
        cache = getattr(self, "_m_cache", None)
        if cache is None or cache[0] != key:
            cache = self._m_cache = (key, self.mask.to(device=x.device, dtype=torch.float32).reshape(-1).contiguous())
        return cache[1]

    def _image_train(self, x, context, sign):
        """the layer as differentiable device passes when this call trains an image-shaped flow (image_training.py: the
        conditioner's convolutions, their weight gradients and this residual on the HIP kernels); None: not applicable"""
        cond = self.conditioner
        if context is not None or not (torch.is_tensor(x) and x.dim() == 4 and x.is_cuda and x.dtype == torch.float32
                                       and torch.is_grad_enabled() and hasattr(cond, "train_on_device")
                                       and self.mask.numel() == math.prod(x.shape[1:])):
            return None
        if not cond.train_on_device(x):
            return None
        from .image_training import MaskedResidual
        # (x forks into the conditioner and the residual: the conditioner's first convolution hands x back through its own
        # autograd node, so the two gradients are summed inside its data-gradient pass)
        # (... and the residual leaves the last convolution's launch where its kernel has that form)
        t, xs, done = cond._forward_train_device(x, self._mask_flat(x), fork=True, residual=(self._one_minus_mask(x), sign))
        if done:
            return t
        return MaskedResidual.apply(xs if xs is not None else x, t, self._one_minus_mask(x), sign)

    def _conditioner_masked(self, x, context, sign=None):
        """conditioner(x * mask); a ConvNet2D on the device takes x and the mask and multiplies inside its first
        convolution's staging pass.  sign (+-1.0, device image path): the conditioner may also write the coupling's output
        x + sign * (1 - mask) * t from its last convolution -- returns (tensor, done)"""
        cond = self.conditioner
        if context is None and hasattr(cond, "first_conv_on_device") and x.dim() == 4 \
                and self.mask.numel() == math.prod(x.shape[1:]) and use_hip(self, x) and cond.first_conv_on_device(x):
            if sign is not None:
                return cond(x, in_mul=self._mask_flat(x), residual=(x, self._one_minus_mask(x), sign))
            return cond(x, in_mul=self._mask_flat(x))
        x_masked = x * self.mask
        t = cond(x_masked) if context is None else cond(x_masked, context)
        return (t, False) if sign is not None else t

    def forward(self, x, context=None):
        if self._hip_ok(x, context):
            return self._hip("forward", x, context)
        y = self._image_train(x, context, 1.0)
        if y is not None:
            return y
        t, done = self._conditioner_masked(x, context, 1.0)
        if done:
            return t
        y = self._image_residual(x, t, 1.0)
        return y if y is not None else x + (1 - self.mask) * t

    def backward(self, y, context=None):
        if self._hip_ok(y, context):
            return self._hip("backward", y, context)
        x = self._image_train(y, context, -1.0)
        if x is not None:
            return x
        t, done = self._conditioner_masked(y, context, -1.0)
        if done:
            return t
        x = self._image_residual(y, t, -1.0)
        return x if x is not None else y - (1 - self.mask) * t

    def log_abs_det_jacobian(self, x, y, context=None) -> float:
        return 0.0

    def sign(self):
        return 1.0

    def to(self, device):
        self.mask = self.mask.to(device)
        return super().to(device)


class AffineMaskedCoupling(BaseTransform):
    """Affine (scale-and-shift) coupling -- an EXTENSION, not part of the reference (BASELINE.json's north_star names
    it; the reference's ``MaskedCoupling`` is additive only, transforms.py:254-347, and its vestige ``AdditiveAffineNN``
    fixes the log-scale to 0, networks.py:14-37).  PARITY UNPINNED: there is no reference arithmetic to match; the tests
    are self-consistency (round trip, log-det against the autograd Jacobian, device against the torch formulation).
    A flow containing it is NOT uniformly scaling: its log-det depends on x, upper-density-level sets are not
    preserved (README.md:7-12), so it is excluded from every UDL check and never built by ``USFlow``.

        forward :  y = x * m + (1 - m) * (x * exp(s) + t)        (t, s_raw) = conditioner(x * m),  mask m == 1: pass-through
        backward:  x = y * m + (1 - m) * ((y - t) * exp(-s))     s = scale_bound * tanh(s_raw / scale_bound)  (None: s_raw)
        log|det J| per sample = sum_d (1 - m_d) * s_d            -> a [B] tensor

    ``conditioner`` returns ``[..., 2 D]`` (shift | raw log-scale) or a ``(shift, raw log-scale)`` pair (``DenseNN`` with
    ``param_dims=[D, D]``).  On a ROCm device (flat inputs, no autograd, D and the hidden widths multiples of 4, a
    (Conditional)DenseNN with (Leaky)ReLU) the conditioner runs as ``usf_linear_f32`` launches with the masks folded
    into its first / last weights, and the masked scale / shift apply with the per-sample log-det wave reduction is
    ``usf_affine_coupling_apply_f32``."""

    def __init__(self, mask: torch.Tensor, conditioner: nn.Module, scale_bound: Optional[float] = 2.0, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.mask = mask
        self.conditioner = conditioner
        self.scale_bound = scale_bound
        self.input_shape = mask.shape
        self._dev_cache = None
        self._last = None              # (forward input, forward output, versions, log-det [B]) of the last device call; holds
                                       # the tensors themselves: an address can be reused by a later allocation
        self.device_calls = 0

    # ---- torch formulation (CPU, autograd) ----------------------------------------------------------------------
    def _shift_logscale(self, x_masked, context=None):
        out = self.conditioner(x_masked) if context is None else self.conditioner(x_masked, context)
        if isinstance(out, (tuple, list)):
            t, s = out
        else:
            D = out.shape[-1] // 2
            t, s = out[..., :D], out[..., D:]
        if self.scale_bound is not None:
            s = self.scale_bound * torch.tanh(s / self.scale_bound)
        return t, s

    def forward(self, x, context=None):
        self._last = None
        if self._device_ok(x, context):
            return self._device(x, inverse=False)
        t, s = self._shift_logscale(x * self.mask, context)
        return x * self.mask + (1 - self.mask) * (x * torch.exp(s) + t)

    def backward(self, y, context=None):
        self._last = None
        if self._device_ok(y, context):
            return self._device(y, inverse=True)
        t, s = self._shift_logscale(y * self.mask, context)
        return y * self.mask + (1 - self.mask) * ((y - t) * torch.exp(-s))

    def log_abs_det_jacobian(self, x, y, context=None):
        """[B]: sum over the transformed features of s(x * m); x is the forward-direction INPUT (the conditioning features
        are the same on both sides, so either side's tensor gives the same s)"""
        c = self._last
        if c is not None and c[0] is x and c[1] is y and c[2] == (x._version, y._version):
            # the very pair (by identity, unmodified since) the last device call produced: its kernel already reduced the
            # log-det.  The value is detached, so it is never served where a gradient could be asked of it.
            needs_grad = torch.is_grad_enabled() and (x.requires_grad or y.requires_grad or
                                                      any(q.requires_grad for q in self.conditioner.parameters()))
            if not needs_grad:
                return c[3]
        _, s = self._shift_logscale(x * self.mask, context)
        return ((1 - self.mask) * s).flatten(1).sum(-1)

    def sign(self):
        return 1.0

    def to(self, device):
        self.mask = self.mask.to(device)
        return super().to(device)

    # ---- device path ---------------------------------------------------------------------------------------------
    def _device_ok(self, x, context) -> bool:
        from .networks import ConditionalDenseNN, DenseNN
        from .engine import _activation_of
        cond = self.conditioner
        if context is not None or x.dim() != 2 or self.mask.dim() != 2 or not use_hip(self, x):
            return False
        if not isinstance(cond, (ConditionalDenseNN, DenseNN)) or _activation_of(cond.f) is None:
            return False
        D = x.shape[1]
        widths = [int(h) for h in cond.hidden_dims]
        lin = list(cond.layers)
        return D % 4 == 0 and all(h % 4 == 0 for h in widths) and lin[-1].weight.shape[0] == 2 * D

    def _device_weights(self, device):
        from .networks import ConditionalDenseNN
        cond = self.conditioner
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in cond.parameters())
        if self._dev_cache is None or self._dev_cache[0] != key:
            lin = list(cond.layers)
            if isinstance(cond, ConditionalDenseNN):
                lin = [lin[0]] + lin[2:]                        # (layers[1] is the context branch: unused without context)
            m = self.mask.to(device).flatten().float()
            with torch.no_grad():
                Ws = [l.weight.detach().float().contiguous() for l in lin]
                bs = [l.bias.detach().float().contiguous() for l in lin]
                Ws[0] = (Ws[0] * m[None, :]).contiguous()                       # conditioner(x * m) == (W0 diag(m)) x
                keep = torch.cat([1 - m, 1 - m])                                 # zero shift / log-scale on pass-through features
                Ws[-1] = (Ws[-1] * keep[:, None]).contiguous()
                bs[-1] = (bs[-1] * keep).contiguous()
            self._dev_cache = (key, Ws, bs)
        return self._dev_cache[1], self._dev_cache[2]

    def _device(self, x, inverse: bool):
        from . import _ext
        from .engine import _activation_of
        x = x.contiguous().float()
        B, D = x.shape
        out = x.clone()
        if B == 0:
            return out
        Ws, bs = self._device_weights(x.device)
        act, slope = _activation_of(self.conditioner.f)
        h = x
        for j, (W, b) in enumerate(zip(Ws, bs)):
            last = j == len(Ws) - 1
            nxt = torch.empty(B, W.shape[0], dtype=torch.float32, device=x.device)
            _ext.linear(h, W, nxt, M=B, N=W.shape[0], K=W.shape[1], lda=h.shape[1], ldw=W.shape[1], ldc=W.shape[0],
                        bias=b, act=_ext.ACT_NONE if last else act, slope=slope)
            h = nxt
        logdet = torch.zeros(B, dtype=torch.float32, device=x.device)
        bound = float(self.scale_bound) if self.scale_bound is not None else 0.0
        _ext.affine_coupling_apply(out, D, h, 2 * D, h, 2 * D, B, D, bound, inverse, logdet, s_off=D)
        self.device_calls += 1
        # forward-direction log-det of the pair (forward input, forward output)
        fwd_ld = -logdet if inverse else logdet
        a, b_ = (out, x) if inverse else (x, out)
        self._last = (a, b_, (a._version, b_._version), fwd_ld)
        return out


class InverseTransform(BaseTransform):
    """Swaps forward/backward of the wrapped (shared) transform (transforms.py:349-414)."""

    def __init__(self, transform, *args, **kwargs):
        super().__init__()
        self.transform = transform
        self.bijective = transform.bijective
        self.args = args
        self.kwargs = kwargs

    def forward(self, x, context=None):
        return self.transform.backward(x, context)

    def backward(self, y, context=None):
        return self.transform.forward(y, context)

    def log_abs_det_jacobian(self, x, y, context=None):
        return -self.transform.log_abs_det_jacobian(x, y, context)

    def sign(self):
        return self.transform.sign()

    # (is_feasible / add_jitter: inherited from BaseTransform as in the reference, transforms.py:349-414)

    def simplify(self):
        return InverseTransform(self.transform.simplify(), *self.args, **self.kwargs)


class AffineTransform(BaseTransform):
    """Interface: matrix() / inverse_matrix() / bias() of y = A x + b (transforms.py:697-750)."""

    def __init__(self, dim: int, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.dim = dim
        self.input_shape = dim

    def matrix(self) -> torch.Tensor:
        raise NotImplementedError()

    def bias(self) -> torch.Tensor:
        raise NotImplementedError()

    def inverse_matrix(self) -> torch.Tensor:
        raise NotImplementedError()

    def _to_plane_linear(self):
        return PlaneBijectiveLinearTransform(self.dim, self.matrix(), self.bias(), self.inverse_matrix())

    def simplify(self):
        return self._to_plane_linear()


class PlaneBijectiveLinearTransform(BaseTransform):
    """Frozen dense form of an affine layer, used by ``simplify()`` (transforms.py:618-695)."""

    def __init__(self, dim, m, bias, m_inv=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.dim = dim
        self.bias_vector = bias
        self.forth = nn.Linear(dim, dim, bias=True)
        self.forth.weight = nn.Parameter(m)
        self.forth.bias = nn.Parameter(bias)
        self.back = nn.Linear(dim, dim, bias=True)
        self.back.weight = nn.Parameter(m_inv)
        self.back.bias = nn.Parameter(-torch.matmul(m_inv, bias))
        self.m_inv = m_inv
        with torch.no_grad():
            self.ladj = torch.linalg.slogdet(m)[1]

    def log_abs_det_jacobian(self, x, y, context=None):
        return self.ladj

    def forward(self, x, context=None):
        return self.forth(x)

    def backward(self, y, context=None):
        return self.back(y)

    def matrix(self):
        return self.forth.weight

    def inverse_matrix(self):
        return self.back.weight

    def bias(self):
        return self.forth.bias


class HouseholderTransform(AffineTransform):
    """w_0 @ prod_k (I - 2 v_k v_k^T / v_k.v_k), w_0 a fixed random permutation
    (transforms.py:752-872)."""

    sign = 1
    ladj = 0

    def __init__(self, dim: int, nvs: int = 1, device="cpu", *args, **kwargs) -> None:
        super().__init__(dim, *args, **kwargs)
        self.nvs = nvs
        self.dim = dim
        indices = torch.randperm(dim)
        w = torch.zeros((dim, dim))
        w[torch.arange(dim), indices] = 1.0
        self.vk_householder = nn.Parameter(0.2 * torch.randn(nvs, dim), requires_grad=True)
        self.w_0 = nn.Parameter(w.to(torch.float32), requires_grad=False)
        self.to(device)

    def _construct_householder_permutation(self) -> torch.Tensor:
        w = self.w_0
        eye = torch.eye(self.dim, dtype=w.dtype, device=w.device)
        for vk in self.vk_householder:
            w = torch.mm(w, eye - 2 * torch.outer(vk, vk) / torch.dot(vk, vk))
        return w

    def forward(self, x, context=None):
        return torch.matmul(x, self._construct_householder_permutation().transpose(0, 1).contiguous())

    def backward(self, y, context=None):
        return torch.matmul(y, self._construct_householder_permutation().T)

    def log_abs_det_jacobian(self, x, y, context=None):
        return self.ladj

    def matrix(self):
        return self._construct_householder_permutation()

    def inverse_matrix(self):
        return self._construct_householder_permutation().transpose(0, 1).contiguous()

    def bias(self):
        return torch.zeros(self.dim, device=self.vk_householder.device)


def _grad_is_masked(grad) -> bool:
    from .image_training import take_masked
    return grad.is_cuda and take_masked(grad)


class LUTransform(AffineTransform):
    """y = (L U) x + b with unit-lower L and upper U (transforms.py:1178-1379)."""

    volume_preserving = False

    def __init__(self, dim: int, prior_scale: float = 1.0, *args, **kwargs):
        super().__init__(dim, *args, **kwargs)
        self.L_raw = nn.Parameter(torch.empty(dim, dim))
        self.U_raw = nn.Parameter(torch.empty(dim, dim))
        self.bias_vector = nn.Parameter(torch.empty(dim))
        self.dim = dim
        self.prior_scale = prior_scale
        self.init_params()
        self.input_shape = dim
        self.L_mask = torch.tril(torch.ones(dim, dim), diagonal=-1)
        self.U_mask = torch.triu(torch.ones(dim, dim), diagonal=0)
        # keep the structural zeros: gradients outside the triangles are masked
        # (a gradient that comes from usf_affine_prep_bwd_f32 already holds exact zeros there: image_training.take_masked)
        self.L_raw.register_hook(lambda grad: None if _grad_is_masked(grad) else grad * self.L_mask)
        self.U_raw.register_hook(lambda grad: None if _grad_is_masked(grad) else grad * self.U_mask)

    def init_params(self):
        d = self.dim
        init.kaiming_uniform_(self.L_raw, nonlinearity="relu")
        with torch.no_grad():
            self.L_raw.copy_(self.L_raw.tril(diagonal=-1).fill_diagonal_(1))
        init.kaiming_uniform_(self.U_raw, nonlinearity="relu")
        with torch.no_grad():
            self.U_raw.fill_diagonal_(0)
            sign = -torch.ones(d) + 2 * torch.bernoulli(0.5 * torch.ones(d))
            scale = self.prior_scale * torch.ones(d) * 1 / d if self.prior_scale is not None else torch.ones(d)
            self.U_raw += sign * torch.normal(torch.zeros(d), scale).exp().diag()
            self.U_raw.copy_(self.U_raw.triu())
        bound = 1 / math.sqrt(d) if d > 0 else 0
        init.uniform_(self.bias_vector, -bound, bound)

    @property
    def L(self):
        return self.L_raw.tril(-1) + torch.eye(self.dim, dtype=self.L_raw.dtype, device=self.L_raw.device)

    @property
    def U(self):
        return self.U_raw.triu()

    def matrix(self):
        return torch.matmul(self.L, self.U)

    def bias(self):
        return self.bias_vector

    @staticmethod
    def _tri_inverse(t: torch.Tensor, upper: bool) -> torch.Tensor:
        """``torch.inverse`` of a triangular factor (transforms.py:1289-1293).  On a ROCm device: a triangular solve against
        the identity -- the same matrix without a pivoted LU behind it, differentiable, no host synchronisation, so a
        training step of the composite formulation can be captured in a hipGraph (``Flow.fit``)"""
        if t.is_cuda:
            return torch.linalg.solve_triangular(t, torch.eye(t.shape[-1], dtype=t.dtype, device=t.device), upper=upper)
        return torch.inverse(t)

    def inverse_matrix(self):
        return torch.matmul(self._tri_inverse(self.U, True), self._tri_inverse(self.L, False))

    def forward(self, x, context=None):
        return F.linear(x, self.matrix(), self.bias())

    def backward(self, y, context=None):
        x = y - self.bias_vector
        x = F.linear(x, self._tri_inverse(self.L, False))
        return F.linear(x, self._tri_inverse(self.U, True))

    def log_abs_det_jacobian(self, x, y, context=None):
        # sum log|diag U| in the reference's diag()-free (ONNX-friendly) form, transforms.py:1313-1320
        U = self.U
        dU = U - U.triu(1) + (torch.ones_like(U) - torch.eye(self.dim, dtype=U.dtype, device=U.device))
        return dU.abs().log().sum()

    def sign(self):
        return self.L.diag().prod().sign() * self.U.diag().prod().sign()

    def to(self, device):
        self.L_mask = self.L_mask.to(device)
        self.U_mask = self.U_mask.to(device)
        self.device = device
        return super().to(device)

    def is_feasible(self) -> bool:
        return (self.U_raw.diag() != 0).all()

    def add_jitter(self, jitter: float = 1e-6) -> None:
        perturbation = torch.randn(self.dim, device=self.U_raw.device) * jitter
        with torch.no_grad():
            self.U_raw.copy_(self.U_raw + perturbation * torch.eye(self.dim, device=self.U_raw.device))

    def log_prior(self):
        x = self.U.diag().abs().log()
        return -(x * x).sum() / (2 * self.prior_scale ** 2) - x.sum()


class SequentialAffineTransform(AffineTransform):
    """Composition of affine transforms, row-vector convention of the reference
    (transforms.py:1381-1486): matrix = M_1 M_2 ..., bias = (..(0 M_1 + b_1) M_2 + b_2 ..)."""

    def __init__(self, transforms: Iterable[AffineTransform], *args, **kwargs) -> None:
        transforms = list(transforms)
        dim = transforms[0].dim
        if any(t.dim != dim for t in transforms):
            raise ValueError("All transforms must have the same dimension")
        super().__init__(dim, *args, **kwargs)
        self.transforms = nn.ModuleList(transforms)
        self.device = "cpu"

    def _dev(self):
        for p in self.parameters():
            return p.device
        return torch.device(self.device)

    def forward(self, x, context=None):
        for t in self.transforms:
            x = t(x, context)
        return x

    def backward(self, y, context=None):
        for t in self.transforms[::-1]:
            y = t.backward(y, context)
        return y

    def log_abs_det_jacobian(self, x, y, context=None):
        return sum(t.log_abs_det_jacobian(x, y, context) for t in self.transforms)

    def sign(self):
        return math.prod(t.sign() if callable(t.sign) else t.sign for t in self.transforms)

    def matrix(self):
        M = torch.eye(self.dim, device=self._dev())
        for t in self.transforms:
            M = torch.matmul(M, t.matrix())
        return M

    def inverse_matrix(self):
        M = torch.eye(self.dim, device=self._dev())
        for t in self.transforms[::-1]:
            M = torch.matmul(M, t.inverse_matrix())
        return M

    def bias(self):
        b = torch.zeros(self.dim, device=self._dev())
        for t in self.transforms:
            b = torch.matmul(b, t.matrix()) + t.bias()
        return b

    # (no log_prior override: the reference's SequentialAffineTransform, transforms.py:1381-1486, inherits
    # BaseTransform.log_prior == 0.0 -- see BlockAffineTransform below)

    def to(self, device):
        for t in self.transforms:
            t.to(device)
        self.device = device
        return super().to(device)


class BlockAffineTransform(BaseTransform):
    """y = A x + b with the SAME block matrix applied at every position of the trailing axes (transforms.py:874-1029):
    ``F.linear`` for flat ``in_dims=[D]`` (``n_blocks`` = 1), a 1x1 convolution over the channel axis for
    ``in_dims=[C, *spatial]`` (``n_blocks`` = prod(spatial): the log-det counts once per position)."""

    def __init__(self, in_dims: Iterable[int], block_transform: AffineTransform, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.in_dims = in_dims
        if block_transform.dim != in_dims[0]:
            raise ValueError("block_transform dim must match input dim")
        self.block_size = in_dims[0]
        self.input_rank = len(in_dims) - 1
        self.n_blocks = math.prod(in_dims[1:])
        self.global_transform = {1: F.linear, 2: F.conv1d, 3: F.conv2d, 4: F.conv3d}[len(in_dims)]   # (:904-910)
        self.block_transform = block_transform
        self._chan_cache = None

    def _channel_prep(self, device):
        """(M, M^-1, bias) fp32 on the device for the image-shaped path, cached per parameter version; derived by the
        same batched fp64 prep kernels as the flat path (engine.prepare_affine_blocks)"""
        from .engine import prepare_affine_blocks
        key = (str(device),) + tuple((p.data_ptr(), p._version) for p in self.block_transform.parameters())
        if self._chan_cache is None or self._chan_cache[0] != key:
            with torch.no_grad():
                r = prepare_affine_blocks([self.block_transform], device)[id(self.block_transform)]
                M, Minv, b = r["M"], r["Minv"], r["b"]
                C = M.shape[0]
                conv = None
                if C > 16 and C not in (24, 32, 48, 64):      # (those widths have a full-width instance of usf_channel_affine_f32)
                    # wide channel counts: the 1 x 1 convolution goes to the matrix cores (usf_conv2d_same_f32, kernel 1):
                    # forward y = M x + b, backward x = Minv y + c with c = -(Minv b) folded in fp64
                    from . import _ext
                    c = -(Minv @ b)
                    conv = (_ext.conv2d_weight_planes(M.float().reshape(C, C, 1, 1)), b.float().contiguous(),
                            _ext.conv2d_weight_planes(Minv.float().reshape(C, C, 1, 1)), c.float().contiguous())
                self._chan_cache = (key, M.float().contiguous(), Minv.float().contiguous(), b.float().contiguous(), conv)
        return self._chan_cache[1:]

    def _channel_hip(self, x, forward: bool):
        """the 1x1 convolution on NCHW data (row N4): usf_channel_affine_f32 (one thread per pixel, C^2 scalar FMAs, HBM-bound)
        up to 16 channels and at 24 / 32 / 48 / 64 (widths with an instance of their own: the CIFAR configuration's 48);
        other widths above 16 go to usf_conv2d_same_f32 with kernel 1 on the matrix cores when the shape fits it"""
        from . import _ext
        M, Minv, b, conv = self._channel_prep(x.device)
        x = x.contiguous()
        if conv is not None and x.dim() == 4 and \
                _ext.load().usf_conv2d_same_fits(x.shape[1], x.shape[1], x.shape[2], x.shape[3], 1) >= 2:
            return _ext.conv2d_same(x, conv[0] if forward else conv[2], x.shape[1], 1, bias=conv[1] if forward else conv[3])
        y = torch.empty_like(x)
        if forward:
            _ext.channel_affine(x, y, M, bias=b)
        else:
            _ext.channel_affine(x, y, Minv, pre_sub=b)
        return y

    def _channel_train(self, x) -> bool:
        """a call that needs gradients on NCHW data: the 1 x 1 convolution, its data gradient and the C x C weight gradient
        on the HIP kernels (image_training.ChannelAffine); the parameter maps stay torch ops on C x C tensors"""
        if not (self.input_rank == 2 and torch.is_tensor(x) and x.dim() == 4 and x.shape[0] > 0 and torch.is_grad_enabled()
                and config.image_train):
            return False
        from .image_training import channel_affine_train_ok
        return channel_affine_train_ok(x, self.block_size) and _needs_grad(self, x)

    def _use_channel_hip(self, x) -> bool:
        return (self.input_rank >= 1 and x.dim() == self.input_rank + 2 and x.shape[1] == self.block_size
                and self.block_size <= 64 and use_hip(self, x))

    def forward(self, x, context=None):
        if self.input_rank == 0:
            if x.dim() == 2 and use_hip(self, x):
                return self._hip("forward", x)
            return F.linear(x, self.block_transform.matrix().to(x.device), self.block_transform.bias().to(x.device))
        if self._use_channel_hip(x):
            return self._channel_hip(x, True)
        if self._channel_train(x):
            from .image_training import ChannelAffine, current_prep
            prep = current_prep(self.block_transform)
            if prep is not None:
                return ChannelAffine.apply(x, prep[0], prep[2], False, None, len(prep) > 5)      # (device prep: its backward issues queued sums)
            return ChannelAffine.apply(x, self.block_transform.matrix().to(x.device), self.block_transform.bias().to(x.device), False)
        w = self.block_transform.matrix().view(self.block_size, self.block_size, *([1] * self.input_rank)).to(x.device)
        return self.global_transform(x, w, self.block_transform.bias().to(x.device))

    def backward(self, y, context=None):
        if self.input_rank == 0:
            if y.dim() == 2 and use_hip(self, y):
                return self._hip("backward", y)
            w = self.block_transform.inverse_matrix().to(y.device)
            b = self.block_transform.bias().to(y.device)
            return F.linear(y - b, w)
        if self._use_channel_hip(y):
            return self._channel_hip(y, False)
        if self._channel_train(y):
            from .image_training import ChannelAffine, current_prep
            prep = current_prep(self.block_transform)
            if prep is not None:
                return ChannelAffine.apply(y, prep[1], prep[2], True)
            return ChannelAffine.apply(y, self.block_transform.inverse_matrix().to(y.device),
                                       self.block_transform.bias().to(y.device), True)
        w = self.block_transform.inverse_matrix().view(self.block_size, self.block_size,
                                                       *([1] * self.input_rank)).to(y.device)
        b = self.block_transform.bias().view(self.block_size, *([1] * self.input_rank)).to(y.device)
        return self.global_transform(y - b, w)

    def log_abs_det_jacobian(self, x, y, context=None):
        if self.input_rank >= 1 and torch.is_grad_enabled():
            from .image_training import current_prep
            prep = current_prep(self.block_transform)           # (inside Flow.log_prob of an image flow in training)
            if prep is not None:
                return prep[3] * self.n_blocks
        return self.block_transform.log_abs_det_jacobian(x, y, context) * self.n_blocks

    def sign(self):
        s = self.block_transform.sign
        return (s() if callable(s) else s) ** self.n_blocks

    # No is_feasible / add_jitter / log_prior override, exactly as in the reference (transforms.py:874-1029): BlockAffineTransform inherits
    # BaseTransform.log_prior == 0.0 (transforms.py:62-64), so LUTransform.log_prior (transforms.py:1371-1379) is never
    # reached through USFlow.log_prior (flows.py:538-549) and prior_scale does not change the training loss (pinned by
    # the golden Flow.fit run with prior_scale = 0.5, tests/golden/fit_synth_d7_k3_hh1_conj_normal.npz); likewise
    # Flow.is_feasible (flows.py:278-282) sees BaseTransform.is_feasible == True for every wrapped block
    # (transforms.py:32-34), i.e. the "Model is not invertible" check of Flow.fit only watches ScaleTransform.
    # tests/test_mirror_vs_live_reference.py holds both next to the real reference.

    def simplify(self):
        return BlockAffineTransform(self.in_dims, self.block_transform._to_plane_linear())

    def to(self, device):
        self.block_transform.to(device)
        return super().to(device)
