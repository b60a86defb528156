"""``Flow`` / ``USFlow`` -- drop-in mirror of the reference's model API (src/usflows/flows.py:22-605)
with ``log_prob`` / ``sample`` / ``backward`` / ``_forward`` running on MI355X HIP kernels.

Same constructor signatures, attributes (``layers``, ``trainable_layers``, ``base_distribution``,
``transform``, ``device``), methods and state-dict keys as the reference, so
``HyperoptExperiment._trial`` (explib/hyperopt.py:101-122), ``from_checkpoint``
(explib/config_parser.py:233-249) and the evaluators keep working unchanged.

Device path: ``log_prob`` = one fused launch list (``FlowEngine``) + the base-density tail kernel;
used whenever the input is on a ROCm device and no autograd graph is required.  Under autograd
(``fit``) or on CPU the layer loop below runs the differentiable composite formulation.
"""
from __future__ import annotations

import warnings
from typing import Any, Dict, Iterable, List, Literal, Optional, Type

import numpy as np
import torch
from torch import distributions as tdist

from . import _ext
from .config import config
from .distributions import Independent, RadialDistribution, DistributionModule
from .engine import EngineUnsupported, FlowEngine
from .fit_loop import FitMixin, _BatchFeed, _unvalidated  # noqa: F401  (re-exported: tests and tools reach them through this module)
from .layer_loop import LayerLoopMixin, _LogDetSum, _ladj_is_parameter_only, _pure_pass_mode  # noqa: F401
from .transforms import (BaseTransform, BlockAffineTransform, HouseholderTransform, InverseTransform,
                         LUTransform, MaskedCoupling, ScaleTransform, SequentialAffineTransform, _needs_grad)


class _NoCache:
    """stands in for ``TransformedDistribution`` when it cannot be built (only clear_cache is used)."""

    def clear_cache(self):
        pass


class TransformedDistribution(tdist.TransformedDistribution):
    """torch's TransformedDistribution plus the ``clear_cache()`` that ``Flow.fit`` calls after every
    optimiser step (flows.py:207; a pyro extension of the torch class)."""

    def clear_cache(self):
        for t in self.transforms:
            if getattr(t, "_cache_size", 0) == 1:
                t._cached_x_y = None, None



class Flow(LayerLoopMixin, FitMixin, torch.nn.Module):
    """Base flow: a list of bijective layers over a base distribution (flows.py:22-378)."""

    export_modes = Literal["log_prob", "sample"]
    export: export_modes = "log_prob"
    device = "cpu"

    def __init__(self, base_distribution, layers, soft_training: bool = False, training_noise_prior=None,
                 device: str = "cpu", *args, **kwargs) -> None:
        if training_noise_prior is None:
            training_noise_prior = tdist.Uniform(0, 1e-6)
        super().__init__(*args, **kwargs)
        self.soft_training = soft_training
        self.training_noise_prior = training_noise_prior
        self.layers = layers
        self.trainable_layers = torch.nn.ModuleList([l for l in layers if isinstance(l, torch.nn.Module)])
        self.base_distribution = base_distribution
        self._engine_obj = None
        self._engine_failed = False
        self._train_obj = None
        # True: log_prob under autograd runs forward AND backward on the HIP kernels (training.py); False: the
        # differentiable composite formulation in torch ops (USFLOWS_AMD_TRAIN=composite)
        self.use_device_training = config.train_on_device
        self.to(device)
        self.device = device
        # batch dims of the base become event dims (flows.py:94-101)
        batch_shape = self.base_distribution.batch_shape
        if len(batch_shape) > 0:
            self.base_distribution = Independent(self.base_distribution, len(batch_shape))
        try:
            self.transform = TransformedDistribution(self.base_distribution, layers)
        except Exception:
            self.transform = _NoCache()

    # ---- nn.Module plumbing ----------------------------------------------------------------
    def forward(self, x: torch.Tensor):
        if self.export == "log_prob":
            return self.log_prob(x)
        elif self.export == "sample":
            return self.sample()
        elif self.export == "forward":
            return self._forward(x)
        elif self.export == "backward":
            return self.backward(x)
        raise ValueError(f"Unknown export mode {self.export}")

    def to(self, device):
        self.device = device
        self.trainable_layers = torch.nn.ModuleList([l.to(device) for l in self.trainable_layers])
        self._distribution_to(device)
        self._base_cache = None
        return super().to(device)

    def _distribution_to(self, device) -> None:
        pass

    # ---- device engine ---------------------------------------------------------------------
    def engine(self) -> Optional[FlowEngine]:
        """The compiled device form of ``self.layers`` (None if a layer has no fused form)."""
        if self._engine_obj is None and not self._engine_failed:
            try:
                self._engine_obj = FlowEngine(self.layers)
            except EngineUnsupported as e:
                self._engine_failed = True
                self._engine_reason = str(e)
        return self._engine_obj

    def _warn_composite(self, what: str) -> None:
        """one warning per flow when a call on a ROCm device runs the torch composite loop because the layer list has no
        fused device form (never silent: DESIGN.md section 1)"""
        if not getattr(self, "_warned_composite", False):
            self._warned_composite = True
            warnings.warn(f"usflows_amd: {what} on a ROCm device runs the torch composite formulation (the layer loop; "
                          f"layers with a device form of their own still dispatch to their kernels), not the fused "
                          f"HIP launch list: {getattr(self, '_engine_reason', 'layer list has no fused device form')}",
                          RuntimeWarning, stacklevel=3)

    def _on_device_fast_path(self, x: torch.Tensor, context=None) -> bool:
        if not (torch.is_tensor(x) and x.is_cuda and x.dim() == 2):
            return False
        if _needs_grad(self, x, context):
            return False
        if self.engine() is None:
            self._warn_composite("Flow.log_prob / backward / _forward")
            return False
        return True

    def _base_info(self, device):
        """('laplace'|'normal', loc, scale) / ('radial', loc, p) / None for the tail kernel."""
        cache = getattr(self, "_base_cache", None)
        b = self.base_distribution
        d = b
        if isinstance(d, DistributionModule):
            d = d.distribution
        while isinstance(d, tdist.Independent):
            d = d.base_dist
        key = None
        if isinstance(d, (tdist.Laplace, tdist.Normal)):
            key = (id(d), d.loc.data_ptr(), d.loc._version, d.scale.data_ptr(), d.scale._version, str(device))
        elif isinstance(b, RadialDistribution):
            key = (id(b), b.loc.data_ptr(), b.loc._version, str(device))
        if key is None:
            return None
        if cache is not None and cache[0] == key:
            return cache[1]
        D = self.engine().D
        if isinstance(d, (tdist.Laplace, tdist.Normal)):
            loc = d.loc.detach().to(device=device, dtype=torch.float32).expand(D).contiguous()
            scale = d.scale.detach().to(device=device, dtype=torch.float32).expand(D).contiguous()
            info = ("laplace" if isinstance(d, tdist.Laplace) else "normal", loc, scale)
        else:
            if b.p not in (1.0, 2.0, float("inf")) or b.loc.dim() != 1:
                return None
            info = ("radial", b.loc.detach().to(device=device, dtype=torch.float32).contiguous(), b.p)
        if not isinstance(self.base_distribution, DistributionModule):   # parametrised bases change every step
            self._base_cache = (key, info)
        return info

    # ---- the hot path ----------------------------------------------------------------------
    def _forward(self, x: torch.Tensor):
        if self._on_device_fast_path(x):
            return self.engine().transform(x, "forward")
        for layer in self.layers:
            x = layer.forward(x)
        return x

    def backward(self, x: torch.Tensor):
        if self._on_device_fast_path(x):
            return self.engine().transform(x, "backward")
        for layer in reversed(self.layers):
            x = layer.backward(x)
        return x

    def _train_path(self, x: torch.Tensor, context=None):
        """the device training path (training.py) when this call needs gradients of the parameters only"""
        if not (torch.is_tensor(x) and x.is_cuda and x.dim() == 2 and x.shape[0] > 0 and torch.is_grad_enabled()):
            return None
        if getattr(self, "_train_failed", False) or not self.use_device_training or self.engine() is None:
            return None
        from .training import TrainPath
        if self._train_obj is None:
            self._train_obj = TrainPath(self)
        return self._train_obj if self._train_obj.supported(x, context) else None

    def log_prob(self, x: torch.Tensor, context: Optional[torch.Tensor] = None) -> torch.Tensor:
        """log p(x) = base.log_prob(f^-1(x)) - sum_layers log|det J|   (flows.py:225-245)"""
        if self._on_device_fast_path(x, context):
            return self._log_prob_device(x, context)
        path = self._train_path(x, context)
        dp = self._train_obj is not None and self._train_obj.grad_allreduce is not None and torch.is_grad_enabled() \
            and _needs_grad(self, x, context)
        if dp and path is None:
            # data-parallel training: EVERY rank must reach the step's one collective (TrainPath.backward)
            if torch.is_tensor(x) and x.dim() == 2 and x.shape[0] == 0 and not getattr(self, "_train_failed", False):
                from .training import log_prob_empty_shard
                return log_prob_empty_shard(self._train_obj, x)
            raise RuntimeError("usflows_amd: data_parallel_training is enabled but this call cannot take the device "
                               "training path (CPU tensor, input requiring grad, unsupported layer or "
                               "USFLOWS_AMD_TRAIN=composite): the rank would skip the gradient all-reduce and the "
                               "other ranks would hang")
        if path is not None:
            from .training import TrainUnsupported, log_prob_with_grad
            try:
                return log_prob_with_grad(path, x, context)
            except TrainUnsupported as e:
                if dp:
                    raise RuntimeError("usflows_amd: data_parallel_training: this layer list has no device backward")
                if not getattr(e, "input_grad_only", False):
                    self._train_failed = True      # this layer list has no device backward: composite from now on
        if self._layer_loop_list_ok(x, context):
            out = self._layer_loop_listed(x)
            if out is not None:
                return out
        if self._layer_loop_graph_ok(x, context):
            out = self._layer_loop_graphed(x)
            if out is not None:
                return out
        return self._layer_loop_log_prob(x, context)

    def _log_prob_device(self, x, context=None, sum_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        eng = self.engine()
        B = x.shape[0]
        out = torch.empty(B, dtype=torch.float32, device=x.device)
        if B == 0:
            return out
        info = self._base_info(x.device)
        if info is not None and info[0] in ("laplace", "normal") and context is None:
            # large batches on the planes pipeline: the base density is reduced in the last layer's epilogue (z is never stored)
            base = _ext.BASE_LAPLACE if info[0] == "laplace" else _ext.BASE_NORMAL
            fused = eng.latent_base_sums(x, base, info[1], info[2])
            if fused is not None:
                part, n_part, logdet = fused
                _ext.base_logprob(part, 8, B, n_part, _ext.BASE_ROWSUM, None, None, 0.0, out, sum_out, logdet_dev=logdet.neg_dev)
                return out
        zbuf, ldz, logdet = eng.latent(x, context)
        if info is None:
            # arbitrary torch base distribution: density evaluated by the distribution object itself
            z = zbuf[:, : eng.D]
            res = self.base_distribution.log_prob(z) + logdet.neg32(z.device)
            if sum_out is not None:
                sum_out[0] += res.double().sum()
                sum_out[1] += B
            return res
        if info[0] in ("laplace", "normal"):
            base = _ext.BASE_LAPLACE if info[0] == "laplace" else _ext.BASE_NORMAL
            _ext.base_logprob(zbuf, ldz, B, eng.D, base, info[1], info[2], 0.0, out, sum_out, logdet_dev=logdet.neg_dev)
            return out
        if config.radial:
            # radius + norm density + volume term + log-det constant (+ the data-parallel sums) in one launch
            from . import radial
            res = radial.log_prob(self.base_distribution, zbuf, logdet_dev=logdet.neg_dev, sum_out=sum_out, ldz=ldz)
            if res is not None:
                return res
        p = info[2]
        base = {1.0: _ext.BASE_LPNORM1, 2.0: _ext.BASE_LPNORM2}.get(p, _ext.BASE_LPNORMINF)
        _ext.base_logprob(zbuf, ldz, B, eng.D, base, info[1], None, 0.0, out, None)
        res = self.base_distribution.log_prob_from_radius(out) + logdet.neg32(out.device)     # O(B) finishing math
        if sum_out is not None:
            sum_out[0] += res.double().sum()
            sum_out[1] += B
        return res

    def sample(self, sample_shape: Iterable[int] = None, context: Optional[torch.Tensor] = None,
               seed: Optional[int] = None, row_offset: int = 0) -> torch.Tensor:
        """x = f(z), z ~ base (flows.py:247-265).  On a ROCm device with a Laplace/Normal base the
        noise comes from the Philox head kernel (``seed``/``row_offset`` select the substream; the
        default draws a fresh seed from torch's generator), then one fused forward pass."""
        if sample_shape is None:
            sample_shape = [1]
        dev = self._param_device()
        if dev.type == "cuda" and context is None and not _needs_grad(self) and self.engine() is not None:
            info = self._base_info(dev)
            shape = tuple(sample_shape)
            n = int(np.prod(shape)) if len(shape) else 1
            eng = self.engine()
            if info is not None and info[0] in ("laplace", "normal"):
                if seed is None:
                    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
                z = torch.empty(n, eng.D, dtype=torch.float32, device=dev)
                base = _ext.BASE_LAPLACE if info[0] == "laplace" else _ext.BASE_NORMAL
                _ext.base_sample(z, eng.D, n, eng.D, base, info[1], info[2], seed, 0, row_offset)
            elif info is not None and info[0] == "radial" and self.base_distribution.n_batch_dims == 0:
                # RadialDistribution.sample (distributions.py:474-499): radii from the (arbitrary, 1-D) norm
                # distribution -- O(n) torch work -- directions on the unit Lp sphere from the Philox kernel
                if seed is None:
                    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
                r = self.base_distribution.norm_distribution.sample((n,)).reshape(n).to(device=dev, dtype=torch.float32)
                z = torch.empty(n, eng.D, dtype=torch.float32, device=dev)
                base = {1.0: _ext.BASE_LPNORM1, 2.0: _ext.BASE_LPNORM2}.get(info[2], _ext.BASE_LPNORMINF)
                _ext.radial_sample(z, eng.D, n, eng.D, base, info[1], r.contiguous(), seed, 0, row_offset)
            else:
                z = self.base_distribution.sample(shape).to(dev).reshape(n, eng.D).float()
            x = eng.transform(z, "forward")
            return x.reshape(*shape, eng.D)
        y = None
        if dev.type == "cuda" and context is None and not _needs_grad(self) and self.engine() is None:
            y = self._radial_sample_image(tuple(sample_shape), dev, seed, row_offset)
        if y is None:
            if dev.type == "cuda" and not _needs_grad(self) and self.engine() is None:
                self._warn_composite("Flow.sample")
            y = self.base_distribution.sample(sample_shape)
        for layer in self.layers:
            y = layer.forward(y, context=context) if context is not None else layer.forward(y)
        return y

    def _radial_sample_image(self, shape, dev, seed, row_offset):
        """z ~ RadialDistribution with an image-shaped loc [C, H, W] (the live configurations' base: mnist.yaml:79-92) on
        the Philox kernel: radii from the norm distribution (O(n) torch work), directions on the unit Lp sphere over the
        FLATTENED event (distributions.py:474-499: u.reshape(*sample_shape, *loc.shape)) from usf_radial_sample_f32.
        None: not this kind of base"""
        b = self.base_distribution
        if not (isinstance(b, RadialDistribution) and b.n_batch_dims == 0 and b.loc.is_cuda and b.loc.dtype == torch.float32
                and float(b.p) in (1.0, 2.0, float("inf")) and config.radial):
            return None
        n = int(np.prod(shape)) if len(shape) else 1
        D = int(b.loc.numel())
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        with torch.no_grad():
            r = b.norm_distribution.sample((n,)).reshape(n).to(device=dev, dtype=torch.float32).contiguous()
            z = torch.empty(n, D, dtype=torch.float32, device=dev)
            if n > 0:
                base = {1.0: _ext.BASE_LPNORM1, 2.0: _ext.BASE_LPNORM2}.get(float(b.p), _ext.BASE_LPNORMINF)
                _ext.radial_sample(z, D, n, D, base, b.loc.detach().reshape(-1).contiguous(), r, seed, 0, row_offset)
        return z.reshape(*shape, *b.loc.shape)

    def _param_device(self) -> torch.device:
        for p in self.parameters():
            return p.device
        return torch.device("cpu")

    # ---- training (caller of the hot path under autograd) -------------------------------------
    def log_prior(self):
        return 0

    def is_feasible(self) -> bool:
        return all(bool(l.is_feasible()) for l in self.layers if isinstance(l, BaseTransform))

    def add_jitter(self, jitter: float = 1e-6) -> None:
        for l in self.layers:
            if isinstance(l, BaseTransform) and not l.is_feasible():
                l.add_jitter(jitter)

    def calibrated_latent_radial_udl_profile(self, q: float, calibration_dataset: torch.Tensor, r_max: float = 10000,
                                             n_samples: int = 10000, cut_to_data_tail: bool = True) -> torch.Tensor:
        """Radial UDL profile of the base that holds a q-fraction of the calibration set's latents
        (flows.py:294-378); the latents come from the device ``backward`` pass."""
        if not isinstance(self.base_distribution, RadialDistribution):
            raise TypeError("The base distribution of the flow must be of type RadialDistribution.")
        with torch.no_grad():
            latent = self.backward(calibration_dataset)
            lp = None
            if torch.is_tensor(latent) and latent.is_cuda and config.radial:
                from . import radial
                lp = radial.log_prob(self.base_distribution, latent.float().contiguous())      # one launch (usf_radial_logprob_f32)
            if lp is None:
                lp = self.base_distribution.log_prob(latent)
        lp, _ = torch.sort(lp, descending=True)
        threshold = lp[int(len(lp) * q)]
        profile = self.base_distribution.radial_udl_profile(threshold=threshold, r_max=r_max, n_samples=n_samples)
        if not cut_to_data_tail:
            return profile
        tail = self.base_distribution.radial_ldl_profile(threshold=lp[0], r_max=r_max, n_samples=n_samples)
        return _intersect_intervals(profile, tail)


def _intersect_intervals(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Intersection of two unions of disjoint intervals given as [n,2] tensors (sweep)."""
    a = a[a[:, 0].argsort()]
    b = b[b[:, 0].argsort()]
    out, i, j = [], 0, 0
    while i < len(a) and j < len(b):
        lo, hi = torch.maximum(a[i, 0], b[j, 0]), torch.minimum(a[i, 1], b[j, 1])
        if lo <= hi:
            out.append(torch.stack([lo, hi]))
        if a[i, 1] < b[j, 1]:
            i += 1
        else:
            j += 1
    return torch.stack(out) if out else a.new_zeros((0, 2))


class USFlow(Flow):
    """Uniformly scaling flow: [LU(/Householder) affine, additive coupling(, inverse affine)] x K,
    then an LU affine and a scale layer (flows.py:380-605)."""

    MASKTYPE = Literal["checkerboard", "channel"]

    def __init__(self, base_distribution, in_dims: List[int], coupling_blocks: int,
                 conditioner_cls: Type[torch.nn.Module], conditioner_args: Dict[str, Any], soft_training=False,
                 prior_scale: Optional[float] = None, training_noise_prior=None, affine_conjugation: bool = False,
                 nonlinearity: Optional[torch.nn.Module] = None, lu_transform: int = 1, householder: int = 1,
                 masktype: MASKTYPE = "checkerboard", *args, **kwargs):
        self.coupling_blocks = coupling_blocks
        self.in_dims = in_dims
        self.soft_training = soft_training
        self.training_noise_prior = training_noise_prior
        self.conditioner_cls = conditioner_cls
        self.conditioner_args = conditioner_args
        self.prior_scale = prior_scale
        if masktype == "checkerboard":
            self.mask_Generator = USFlow.create_checkerboard_mask
        elif masktype == "channel":
            self.mask_Generator = USFlow.create_channel_mask
        else:
            raise ValueError(f"Unknown mask type {masktype}")
        if lu_transform < 0:
            raise ValueError("Number of LU transforms must be non-negative")
        self.lu_transform = lu_transform
        if householder < 0:
            raise ValueError("Number of Householder vectors transforms must be non-negative")
        self.householder = householder

        layers = []
        mask = self.mask_Generator(in_dims)
        for _ in range(coupling_blocks):
            affine_layers = [LUTransform(in_dims[0], prior_scale) for _ in range(lu_transform)]
            if householder > 0:
                affine_layers.append(HouseholderTransform(dim=in_dims[0], nvs=householder, device=self.device))
            block = None
            if affine_layers:
                block = BlockAffineTransform(in_dims, SequentialAffineTransform(affine_layers))
                layers.append(block)
            layers.append(MaskedCoupling(mask, conditioner_cls(**conditioner_args)))
            if affine_conjugation and block is not None:
                layers.append(InverseTransform(block))
            mask = 1 - mask
        layers.append(BlockAffineTransform(in_dims, LUTransform(in_dims[0], prior_scale)))
        layers.append(ScaleTransform(in_dims))
        super().__init__(base_distribution, layers, soft_training=soft_training,
                         training_noise_prior=training_noise_prior, *args, **kwargs)

    @classmethod
    def create_checkerboard_mask(cls, in_dims, invert: bool = False) -> torch.Tensor:
        """fmod(sum of indices, 2) viewed (1, *in_dims): mask==1 passes through (flows.py:494-514)."""
        axes = [torch.arange(d, dtype=torch.int32) for d in in_dims]
        grid = torch.stack(torch.meshgrid(*axes, indexing="ij"))
        mask = torch.fmod(grid.sum(dim=0), 2).to(torch.float32).view(1, *in_dims)
        return 1 - mask if invert else mask

    @classmethod
    def create_channel_mask(cls, in_dims, invert: bool = False) -> torch.Tensor:
        """fmod(first-axis index, 2) (flows.py:516-536)."""
        axes = [torch.arange(d, dtype=torch.int32) for d in in_dims]
        grid = torch.stack(torch.meshgrid(*axes, indexing="ij"))
        mask = torch.fmod(grid[0], 2).to(torch.float32).view(1, *in_dims)
        return 1 - mask if invert else mask

    def log_prior(self):
        if self.prior_scale is None:
            return 0
        return sum(p.log_prior() for p in self.layers)

    def log_prob(self, x: torch.Tensor, context: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self.soft_training and context is None:
            # implicit conditioning with noise scale 0 (flows.py:559-565)
            context = torch.zeros(x.shape[0], 1, device=x.device)
        return super().log_prob(x, context)

    def sample(self, sample_shape: Iterable[int] = None, context: Optional[torch.Tensor] = None, **kw) -> torch.Tensor:
        return super().sample(sample_shape, context, **kw)

    def simplify(self) -> Flow:
        return Flow(self.base_distribution, [l.simplify() for l in self.layers])
