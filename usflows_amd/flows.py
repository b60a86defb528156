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

import contextlib
import os
import warnings
from typing import Any, Dict, Iterable, List, Literal, Optional, Type

import numpy as np
import torch
from torch import distributions as tdist

from . import _ext
from .config import config
from .distributions import Independent, RadialDistribution, DistributionModule
from .engine import EngineUnsupported, FlowEngine
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



class _BatchFeed:
    """Hand-over of large batches of a HOST data set to ``Flow.fit``'s step (the reference slices the permuted data set and
    the model's ``log_prob`` pulls each slice to the device, flows.py:157-166: a pageable, synchronous copy in front of every
    step -- 205 MB at 65 536 x 784).  Two pinned staging buffers and two device buffers: while step i runs, batch i + 1 is
    copied into pinned memory and uploaded on a copy stream; step i + 1 waits for that upload's event only.  Same batches,
    same order, same values.  Used from 16 MB per batch on a CUDA device with a float32 CPU tensor; USFLOWS_AMD_FIT_PREFETCH=0:
    off.  One feed serves all epochs of a ``fit`` (``make(..., reuse=feed)`` re-points it at the epoch's permuted data).  A device
    buffer is overwritten two batches after its use: the upload waits for the event ``done`` recorded on the compute stream behind
    the step that consumed it -- the order does not rest on the host reading every step's loss back."""

    MIN_BYTES = 16 << 20

    @staticmethod
    def make(data, N, batch_size, device, reuse=None):
        device = torch.device(device)
        if (device.type != "cuda" or not torch.is_tensor(data) or data.is_cuda or data.dtype != torch.float32 or data.dim() < 2
                or not config.fit_prefetch or N <= batch_size):
            return None
        if min(batch_size, N) * data[0].numel() * 4 < _BatchFeed.MIN_BYTES:
            return None
        if reuse is not None and reuse.fits(data, N, batch_size, device):
            reuse.rebind(data)
            return reuse
        return _BatchFeed(data, N, batch_size, device)

    def __init__(self, data, N, batch_size, device):
        self.data, self.N, self.bs, self.device = data, N, batch_size, device
        shape = (min(batch_size, N),) + tuple(data.shape[1:])
        self.pin = [torch.empty(shape, dtype=torch.float32, pin_memory=True) for _ in range(2)]
        self.dev = [torch.empty(shape, dtype=torch.float32, device=device) for _ in range(2)]
        self.up = [torch.cuda.Event() for _ in range(2)]
        self.used = [None, None]                                  # recorded behind the last step that read dev[j]
        self.copy_stream = torch.cuda.Stream(device=device)
        self.staged = -1
        self.stage(0)

    def fits(self, data, N, batch_size, device) -> bool:
        return (self.N == N and self.bs == batch_size and self.device == device
                and tuple(self.pin[0].shape[1:]) == tuple(data.shape[1:]))

    def rebind(self, data) -> None:
        """the next epoch's (permuted) data set through the same buffers, stream and events"""
        self.data, self.staged = data, -1
        self.stage(0)

    def done(self, idx) -> None:
        """the step on the batch that begins at row idx has been issued: its device buffer may be overwritten once the compute
        stream gets here"""
        j = (idx // self.bs) & 1
        if self.used[j] is None:
            self.used[j] = torch.cuda.Event()
        self.used[j].record(torch.cuda.current_stream(self.device))

    def stage(self, idx):
        """start the hand-over of the batch that begins at row idx (no-op beyond the data set or when already staged)"""
        if idx >= self.N or idx <= self.staged:
            return
        j = (idx // self.bs) & 1
        n = min(self.bs, self.N - idx)
        self.up[j].synchronize()                                  # (the upload that last read this pinned buffer: two batches ago)
        self.pin[j][:n].copy_(self.data[idx: idx + n])
        with torch.cuda.stream(self.copy_stream):
            if self.used[j] is not None:
                self.copy_stream.wait_event(self.used[j])        # (the last step that read dev[j])
            self.dev[j][:n].copy_(self.pin[j][:n], non_blocking=True)
            self.up[j].record(self.copy_stream)
        self.staged = idx

    def take(self, idx):
        self.stage(idx)                                           # (normally staged during the previous step)
        j = (idx // self.bs) & 1
        torch.cuda.current_stream(self.device).wait_event(self.up[j])
        return self.dev[j][: min(self.bs, self.N - idx)]


class Flow(torch.nn.Module):
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

    def _parameter_only_ladj_total(self, x):
        """Sum of the layers' log|det J| when every one of them depends on the parameters only (what makes the flow uniformly
        scaling: transforms.py:316-326, 1303-1320, ScaleTransform; not the affine-coupling extension) and nothing is to be
        differentiated -- computed once per parameter version on the device (a 0-dim tensor, no host copy) instead of ~10
        small torch launches per affine layer on every call, as the reference's loop does.  None when it does not apply."""
        if not (torch.is_tensor(x) and x.is_cuda) or (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())):
            return None
        if not all(_ladj_is_parameter_only(l) for l in self.layers):
            return None
        key = (str(x.device),) + tuple((p.data_ptr(), p._version) for p in self.parameters())
        c = self.__dict__.get("_ladj_total_cache")
        if c is None or c[0] != key:
            if torch.cuda.is_current_stream_capturing():
                return None           # (never fill a cache inside a capture: its values would only exist after a replay)
            total = None
            with torch.no_grad():
                for layer in reversed(self.layers):
                    v = layer.log_abs_det_jacobian(None, None)
                    if not torch.is_tensor(v):
                        if float(v) == 0.0:
                            continue                                   # (MaskedCoupling: ladj == 0.0, transforms.py:316-326)
                        v = torch.full((), float(v), dtype=torch.float32, device=x.device)
                    total = v.to(x.device) if total is None else total + v.to(x.device)
                if total is None:
                    total = torch.zeros((), dtype=torch.float32, device=x.device)
            c = self.__dict__["_ladj_total_cache"] = (key, total, (-total.detach().double()).reshape(1).contiguous())
        return c[1]

    def _layer_loop_log_prob(self, x, context=None):
        """the reference's loop (flows.py:236-245), layer by layer"""
        ladj_total = self._parameter_only_ladj_total(x)
        if ladj_total is not None:
            steps = self._image_loop_steps(x) if context is None else None
            if steps is not None:
                for fn in steps:
                    x = fn(x)
            else:
                for layer in reversed(self.layers):
                    x = layer.backward(x, context=context) if context is not None else layer.backward(x)
            y = x
            # the log-det constant joins the base density's pass (an fp64 device scalar, no torch op)
            lp = self._base_log_prob_layer_loop(y, logdet_dev=self.__dict__["_ladj_total_cache"][2])
            if lp is not None:
                return lp
            lp = self._base_log_prob_layer_loop(y)
            return (self.base_distribution.log_prob(y) if lp is None else lp) - ladj_total
        prep = wpl = contextlib.nullcontext()
        if torch.is_tensor(x) and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and torch.is_grad_enabled() \
                and config.image_train:
            # an image-shaped flow in training: the affine blocks' parameter maps once per pass, batched over the blocks
            from .image_training import batched_affine_prep, batched_weight_planes
            prep = batched_affine_prep(self.layers, x.device)
            if x.shape[0] <= _ext.PSUM_DEFER_MAX_ROWS and config.batch_wplanes:
                # (a launch-bound batch: the convolutions' weight planes from ONE launch per pass)
                wpl = batched_weight_planes(self, self.layers, x.device)
        with prep, wpl:
            ld = _LogDetSum()
            seq = list(reversed(self.layers))
            batched = not isinstance(prep, contextlib.nullcontext)
            k = 0
            while k < len(seq):
                layer = seq[k]
                run = self._train_affine_run(seq, k, x) if batched else None
                if run is not None:
                    # a run of consecutive 1 x 1-convolution affine layers in training: ONE differentiable channel-affine
                    # pass on the composed map (the C x C compositions are torch ops on the batched prep's tensors)
                    k1, A, cvec, At = run
                    from .image_training import ChannelAffine
                    y = ChannelAffine.apply(x, A, cvec, False, At, At is not None)
                    for l2 in seq[k:k1]:
                        if not ld.take_affine(l2):
                            ld.sub(l2.log_abs_det_jacobian(None, None))
                    x, k = y, k1
                    continue
                if context is not None:
                    y = layer.backward(x, context=context)
                    ld.sub(layer.log_abs_det_jacobian(y, x, context=context))
                else:
                    y = layer.backward(x)
                    if not (batched and ld.take_affine(layer)):
                        ld.sub(layer.log_abs_det_jacobian(y, x))
                x = y
                k += 1
            lp = self._base_log_prob_layer_loop(y)
            return ld.add_to(self.base_distribution.log_prob(y) if lp is None else lp)

    def _train_affine_run(self, seq, k, x):
        """(end index, A, c, A^T | None) when seq[k:] starts with >= 2 affine layers whose backward is a device channel-affine
        pass in training and whose parameter maps come from the batched prep: y = A x + c for the whole run; else None.  The runs
        of the whole sequence are composed together on first use in a pass (image_training.compose_runs)."""
        from . import image_training as it
        if config.merge_affine is False or self.merge_image_affine is False:
            return None
        runs = it._STATE.runs
        if runs is None:
            runs = it._STATE.runs = self._compose_affine_runs(seq, x)
        return runs.get(k)

    def _compose_affine_runs(self, seq, x) -> dict:
        from .transforms import BlockAffineTransform, InverseTransform
        from . import image_training as it
        found, j = [], 0
        while j < len(seq):
            k, rows, group = j, [], None
            while j < len(seq):
                layer = seq[j]
                inv = isinstance(layer, InverseTransform)
                blk = layer.transform if inv else layer
                if not (isinstance(blk, BlockAffineTransform) and blk._channel_train(x)):
                    break
                pr = it.current_prep(blk.block_transform)
                if pr is None:
                    break
                g = pr[5] if len(pr) > 5 else None                     # (group, row) in the prep kernel's stacks; None: torch prep
                kind = g[0] if g is not None else "torch"
                group = kind if not rows or kind == group else "torch"   # (a run over two stacks: composed on its own, below)
                rows.append((g[1] if g is not None else None, inv, pr))
                j += 1
            if j - k >= 2:
                found.append((k, j, group, rows))
            j = max(j, k + 1)
        out = {}
        dev_runs = [f for f in found if f[2] != "torch"]
        if dev_runs:
            specs = [(g, [(row, inv) for row, inv, _ in rows]) for _, _, g, rows in dev_runs]
            for (k, j, _, _), (A, cvec, At) in zip(dev_runs, it.compose_runs(specs)):
                out[k] = (j, A, cvec, At)
        for k, j, group, rows in found:
            if group != "torch":
                continue
            A = cvec = None                                          # (the torch formulation of the prep: composed run by run)
            for _, inv, pr in rows:
                M, Minv, b = pr[0], pr[1], pr[2]
                c = pr[4] if len(pr) > 4 else None                     # -Minv b, from the prep kernel
                Ak, ck = (M, b) if inv else (Minv, c if c is not None else -(Minv @ b))   # InverseTransform(block).backward == block.forward
                A, cvec = (Ak, ck) if A is None else (Ak @ A, Ak @ cvec + ck)
            out[k] = (j, A, cvec, None)
        return out

    # ---- runs of consecutive 1 x 1-convolution affine layers composed (image-shaped flows, inference) -------------------
    # With ``affine_conjugation=True`` a coupling is followed by ``block_i^-1`` and ``block_(i+1)`` (flows.py:452-470): two
    # C x C maps per pixel with nothing in between -- two HBM-bound passes where one does.  As FlowEngine.merge_affine does
    # for flat flows: every run is composed in fp64 once per parameter version (y = A2 (A1 x + c1) + c2) and applied by ONE
    # usf_channel_affine_f32 launch -- when an end-to-end probe (up to 64 rows of the caller's batch through the loop with
    # composed and with separate layers) agrees to 1e-5 of the largest log-density and 1e-6 in relative L1: a flow that
    # amplifies a change of rounding pattern beyond that (default-initialised, exploding) keeps the reference's layer list.
    merge_image_affine = "auto"   # True / False force it; USFLOWS_AMD_MERGE_AFFINE=0/1 likewise

    def _image_loop_steps(self, x):
        """the reversed layer loop of an image-shaped flow as a list of callables, runs of channel-affine layers composed;
        None: use the plain loop"""
        from .transforms import BlockAffineTransform, InverseTransform
        mode = self.merge_image_affine if config.merge_affine == "auto" else config.merge_affine
        if mode is False or not (torch.is_tensor(x) and x.is_cuda and x.dim() == 4 and x.dtype == torch.float32 and x.shape[0] > 0) \
                or (torch.is_grad_enabled() and _needs_grad(self, x)):
            return None
        ver = tuple((p.data_ptr(), p._version) for p in self.parameters()) + (str(x.device), tuple(x.shape[1:]))
        c = self.__dict__.get("_image_steps_cache")
        if c is not None and c[0] == ver:
            return c[1]
        if torch.cuda.is_current_stream_capturing():
            return None

        def affine_of(layer):
            """(block, forward?) when the layer's backward is a channel-affine launch on this input"""
            inv = isinstance(layer, InverseTransform)
            blk = layer.transform if inv else layer
            if isinstance(blk, BlockAffineTransform) and blk._use_channel_hip(x) and blk._channel_prep(x.device)[3] is None:
                return blk, inv                          # InverseTransform(block).backward == block.forward
            return None

        seq, runs = list(reversed(self.layers)), []
        i = 0
        while i < len(seq):
            j = i
            while j < len(seq) and affine_of(seq[j]) is not None:
                j += 1
            if j - i >= 2:
                runs.append((i, j))
            i = max(j, i + 1)
        steps = None
        if runs:
            from .engine import prepare_affine_blocks
            merged = {}
            with torch.no_grad():
                for (i0, i1) in runs:
                    A = cvec = None
                    for layer in seq[i0:i1]:
                        blk, fwd = affine_of(layer)
                        r = prepare_affine_blocks([blk.block_transform], x.device)[id(blk.block_transform)]
                        Ak = r["M"] if fwd else r["Minv"]                       # fp64
                        ck = r["b"] if fwd else -(r["Minv"] @ r["b"])
                        A, cvec = (Ak, ck) if A is None else (Ak @ A, Ak @ cvec + ck)
                    merged[i0] = (i1, A.float().contiguous(), cvec.float().contiguous())

            def make(Wm, cm):
                def run(t):
                    t = t.contiguous()
                    y = torch.empty_like(t)
                    _ext.channel_affine(t, y, Wm, bias=cm)
                    return y
                return run

            steps, k = [], 0
            while k < len(seq):
                if k in merged:
                    i1, Wm, cm = merged[k]
                    steps.append(make(Wm, cm))
                    k = i1
                else:
                    steps.append(seq[k].backward)
                    k += 1
            if mode == "auto":                           # the end-to-end probe
                n = min(64, x.shape[0])
                xs = x[:n].contiguous()
                with torch.no_grad():
                    a_ = xs
                    for layer in seq:
                        a_ = layer.backward(a_)
                    b_ = xs
                    for fn in steps:
                        b_ = fn(b_)
                    a_, b_ = a_.double().flatten(1), b_.double().flatten(1)
                    d = ((b_ - a_).abs().max() / a_.abs().max().clamp_min(1e-30)).item()
                    l1 = a_.abs().sum(-1)
                    d1 = ((b_.abs().sum(-1) - l1).abs() / l1.clamp_min(1e-30)).max().item()
                ok = bool(d <= 1e-5 and d1 <= 1e-6)
                log = self.__dict__.setdefault("merge_guard_log", [])
                log.append((ok, d, d1))
                del log[:-64]
                if not ok:
                    steps = None
        self.__dict__["_image_steps_cache"] = (ver, steps)
        return steps

    # ---- the layer loop of an image-shaped flow as ONE op list (usf_run_ops / USF_OP_CALL) ------------------------------
    # On the device the loop of an image-shaped flow is HIP calls only (scale, channel affine, convolutions, pointwise /
    # elementwise passes, base density with the log-det constant).  The second time a (shape, parameter version) pair is
    # seen the loop runs once more while its calls are RECORDED (argument words as they are; a torch dispatch mode keeps
    # every tensor the pass allocates alive and checks that nothing but allocations and views ran beside the HIP calls);
    # from then on the call is one C-side list with the input / output pointers patched in: no per-layer Python, no
    # per-layer ctypes call, no stream capture and none of its restrictions.  The list keeps the pass's intermediates
    # alive, so it serves batches whose intermediates stay under ``list_max_bytes``; a pass that is not pure (a shape one of
    # the kernels does not serve -> torch fallback inside a layer) is remembered as such and keeps the eager loop / graph.
    # MEMORY: a list pins its pass's intermediates (that is what makes it replayable): at most ``list_max_bytes`` per list,
    # 8 lists / 2 GB per flow, oldest out first; ``flow.list_max_bytes = 0`` (or USFLOWS_AMD_LOOP_LIST=0) keeps nothing.
    list_max_rows = 4096          # USFLOWS_AMD_LOOP_LIST=0: off (and graph_max_rows = 0 switches every replay form off)
    list_max_bytes = 1 << 30

    def _layer_loop_list_ok(self, x, context) -> bool:
        return (context is None and torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() >= 3
                and 0 < x.shape[0] <= self.list_max_rows and self.graph_max_rows > 0 and x.is_contiguous()
                and config.loop_list and not _needs_grad(self, x)
                and not torch.cuda.is_current_stream_capturing())

    def _loop_versions(self):
        ver = tuple((p.data_ptr(), p._version) for p in self.parameters()) + \
            tuple((b.data_ptr(), b._version) for b in self.buffers())
        d = self.base_distribution                               # (a torch distribution's tensors are not module state)
        while isinstance(d, tdist.Independent):
            d = d.base_dist
        ver += tuple((t.data_ptr(), t._version) for t in (getattr(d, "loc", None), getattr(d, "scale", None)) if torch.is_tensor(t))
        ver += tuple((l.mask.data_ptr(), l.mask._version) for l in self.layers if torch.is_tensor(getattr(l, "mask", None)))
        return ver

    def _layer_loop_listed(self, x):
        """log_prob of an image-shaped batch through the recorded op list; None when it did not run (first sighting, impure
        pass, too large)"""
        ver = self._loop_versions()
        cache = self.__dict__.setdefault("_loop_lists", {})
        key = (tuple(x.shape), str(x.device))
        hit = cache.get(key)
        if hit is not None and hit[0] == ver:
            plan = hit[1]
            if plan is None:
                return None
            out = torch.empty(plan["out_shape"], dtype=torch.float32, device=x.device)
            ops = plan["ops"]
            for i, j in plan["in_pos"]:
                ops[i].u.call.a[j] = x.data_ptr()
            for i, j in plan["out_pos"]:
                ops[i].u.call.a[j] = out.data_ptr()
            _ext.run_ops(ops, plan["n"], x.device)
            return out
        seen = self.__dict__.setdefault("_loop_list_seen", {})
        if seen.get(key) != ver:                                 # hysteresis: record on the second sighting (caches are warm)
            seen[key] = ver
            if len(seen) > 16:
                seen.pop(next(iter(seen)))
            return None
        cl = _ext.CallList()
        mode = _pure_pass_mode()
        with torch.no_grad(), _ext.recording_calls(cl), mode:
            out = self._layer_loop_log_prob(x)
        plan = None
        kept = {t.untyped_storage().data_ptr(): t.untyped_storage().nbytes() for t in mode.kept}
        if cl.bad is None and not mode.impure and cl.calls and sum(kept.values()) <= self.list_max_bytes \
                and torch.is_tensor(out) and out.dtype == torch.float32 and out.is_contiguous():
            xin, xout = x.data_ptr(), out.data_ptr()
            in_pos = [(i, j) for i, (_, words, isp) in enumerate(cl.calls) for j, w in enumerate(words) if isp[j] and w == xin]
            out_pos = [(i, j) for i, (_, words, isp) in enumerate(cl.calls) for j, w in enumerate(words) if isp[j] and w == xout]
            if in_pos and out_pos:
                plan = dict(ops=cl.ops(), n=len(cl.calls), in_pos=in_pos, out_pos=out_pos, out_shape=tuple(out.shape),
                            keep=mode.kept, bytes=sum(kept.values()))
        cache[key] = (ver, plan)
        # at most 8 lists and 2 GB of kept intermediates over all of them (oldest first out)
        while len(cache) > 8 or (len(cache) > 1 and sum(v[1]["bytes"] for v in cache.values() if v[1] is not None) > (2 << 30)):
            cache.pop(next(iter(cache)))
        return out

    # ---- small batches of the layer loop (image-shaped flows): one hipGraph replay instead of ~50 launches ------------
    graph_max_rows = 256          # the reference evaluates in chunks of 100 (hyperopt.py:273-278); USFLOWS_AMD_LOOP_GRAPH=0: off

    def _layer_loop_graph_ok(self, x, context) -> bool:
        return (context is None and torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() >= 3
                and 0 < x.shape[0] <= self.graph_max_rows and not getattr(self, "_loop_graph_off", False)
                and config.loop_graph and not _needs_grad(self, x)
                and not torch.cuda.is_current_stream_capturing())

    def _layer_loop_graphed(self, x):
        """The sync-free layer loop captured once per (input shape, parameter versions) and replayed: at 32 .. 256 rows the
        loop is ~50 dependent launches whose host side (module calls, ctypes, allocations) costs twice their GPU time --
        MNIST image configuration, 100 rows: 0.90 -> 0.43 ms.  Any failure to capture switches this off for the flow (the
        eager loop serves the call).  Returns None when it did not run."""
        ver = self._loop_versions()
        cache = self.__dict__.setdefault("_loop_graphs", {})
        key = (tuple(x.shape), str(x.device))
        hit = cache.get(key)
        if hit is None or hit[0] != ver:
            # hysteresis: a capture costs two warm-up passes and a capture pass -- several eager calls' worth.  A (shape,
            # parameter version) pair is captured the SECOND time it is seen; a caller that alternates one optimiser step with
            # one small evaluation (new versions every call) keeps the eager loop and pays nothing.
            seen = self.__dict__.setdefault("_loop_graph_seen", {})
            if seen.get(key) != ver:
                seen[key] = ver
                if len(seen) > 16:
                    seen.pop(next(iter(seen)))
                return None
            try:
                with torch.no_grad():
                    static_x = x.detach().clone()
                    side = torch.cuda.Stream(device=x.device)
                    side.wait_stream(torch.cuda.current_stream(x.device))
                    with torch.cuda.stream(side):
                        for _ in range(2):                       # caches (prep, weight planes, masks) fill outside the capture
                            self._layer_loop_log_prob(static_x)
                    torch.cuda.current_stream(x.device).wait_stream(side)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        static_out = self._layer_loop_log_prob(static_x)
                hit = cache[key] = (ver, g, static_x, static_out)
                if len(cache) > 8:
                    cache.pop(next(iter(cache)))
            except Exception as e:                               # noqa: BLE001 -- capture is an optimisation, never a requirement
                import warnings
                warnings.warn(f"usflows_amd: hipGraph capture of the layer loop failed ({type(e).__name__}: {e}); "
                              "small batches keep the eager loop", RuntimeWarning)
                self._loop_graph_off = True
                return None
        _, g, static_x, static_out = hit
        static_x.copy_(x)
        g.replay()
        return static_out.clone()

    def _base_log_prob_layer_loop(self, y: torch.Tensor, logdet_dev: Optional[torch.Tensor] = None):
        """Laplace / Normal base density of the layer loop's result through ``usf_base_logprob_f32`` (rows flattened) when
        nothing needs a gradient: one launch instead of the distribution object's op chain, whose argument validation
        (``_validate_sample``) synchronises the host with the device on every call.  None: not applicable."""
        if not (torch.is_tensor(y) and y.is_cuda and y.dtype == torch.float32 and y.dim() >= 2):
            return None
        if y.shape[0] == 0 and not isinstance(self.base_distribution, RadialDistribution):
            return None                                  # (the radial path serves an empty batch itself: empty result, zero gradients)
        train = torch.is_grad_enabled() and (y.requires_grad or _needs_grad(self, y, None))
        if train and (y.dim() < 3 or not config.image_train):
            return None                                  # (flat flows train through training.py; image flows: below)
        if train and logdet_dev is not None:
            return None                                  # (the differentiable forms below do not add the constant: the caller subtracts it)
        d, n_ind = self.base_distribution, 0
        if isinstance(d, RadialDistribution):
            # the Lp-radial base of the live image configurations (mnist.yaml:79-92, fashionclasses_veriflow.yaml:79-93):
            # radius, norm density, volume term -- and in training their gradients -- on usf_radial_logprob(_grad)_f32
            if not config.radial:
                return None
            from . import radial
            return radial.log_prob(d, y, logdet_dev=logdet_dev)
        if isinstance(d, DistributionModule):
            return None
        while isinstance(d, tdist.Independent):
            n_ind += d.reinterpreted_batch_ndims
            d = d.base_dist
        ev = tuple(y.shape[1:])
        if not isinstance(d, (tdist.Laplace, tdist.Normal)) or n_ind != len(ev) or tuple(d.batch_shape) != ev:
            return None
        key = (id(d), d.loc.data_ptr(), d.loc._version, d.scale.data_ptr(), d.scale._version, str(y.device), ev)
        cache = getattr(self, "_base_loop_cache", None)
        if cache is None or cache[0] != key:
            loc = d.loc.detach().to(device=y.device, dtype=torch.float32).expand(ev).reshape(-1).contiguous()
            scale = d.scale.detach().to(device=y.device, dtype=torch.float32).expand(ev).reshape(-1).contiguous()
            cache = self._base_loop_cache = (key, loc, scale)
        _ext.load()
        if train:
            # an image-shaped flow in training: the density and its gradient on the device (image_training.BaseLogProb);
            # a base with trainable parameters keeps the distribution object's op chain
            if d.loc.requires_grad or d.scale.requires_grad:
                return None
            from .image_training import BaseLogProb
            return BaseLogProb.apply(y, cache[1], cache[2], _ext.BASE_LAPLACE if isinstance(d, tdist.Laplace) else _ext.BASE_NORMAL)
        B, D = y.shape[0], cache[1].numel()
        yf = y.reshape(B, D).contiguous()
        out = torch.empty(B, dtype=torch.float32, device=y.device)
        _ext.base_logprob(yf, D, B, D, _ext.BASE_LAPLACE if isinstance(d, tdist.Laplace) else _ext.BASE_NORMAL, cache[1], cache[2],
                          0.0, out, logdet_dev=logdet_dev)
        return out

    def _log_prob_device(self, x, context=None, sum_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        eng = self.engine()
        B = x.shape[0]
        out = torch.empty(B, dtype=torch.float32, device=x.device)
        if B == 0:
            return out
        zbuf, ldz, logdet = eng.latent(x, context)
        info = self._base_info(x.device)
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

    def fit(self, data_train, optim=None, optim_params: Dict[str, Any] = None, batch_size: int = 32,
            shuffle: bool = True, gradient_clip: float = None, device: torch.device = None, epochs: int = 1):
        """Maximum-posterior fitting loop with the reference's semantics (flows.py:113-210):
        loss = -log_prob(batch).mean() - log_prior(); feasibility check after every step.
        ``optim`` defaults to SophiaG as in the reference (flows.py:116; usflows_amd/sophia.py)."""
        if optim is None:
            from .sophia import SophiaG
            optim = SophiaG
        if device is None:
            device = torch.device("cuda:0") if torch.cuda.is_available() else torch.device("cpu")
        model = self.to(device)
        optim = optim(model.parameters(), **optim_params) if optim_params is not None else optim(model.parameters())
        N = len(data_train)
        epoch_losses = []
        with self.fit_stream(device):
            self._fit_epochs(model, optim, data_train, N, epochs, batch_size, shuffle, gradient_clip, device, epoch_losses)
        return epoch_losses

    @contextlib.contextmanager
    def fit_stream(self, device):
        """On a GPU the whole loop of ``fit`` runs on a stream of the flow's own (created once): the launch tapes of the device
        training path record the stream they were made on, and a training step can only be captured into a hipGraph on that
        very stream (a capture does not reach over to another one) -- never torch's legacy default stream.  Entering makes
        that stream current (ordered behind the caller's), leaving orders the caller's stream behind it.  A caller that drives
        ``_train_graph_step`` itself (bench.py) runs its steps inside this context."""
        side = None
        device = torch.device(device)
        if device.type == "cuda" and self.use_train_graph and config.train_graph:
            side = self.__dict__.get("_fit_stream")
            if side is None or side.device != (device if device.index is not None else torch.device("cuda", torch.cuda.current_device())):
                side = self.__dict__["_fit_stream"] = torch.cuda.Stream(device=device)
            side.wait_stream(torch.cuda.current_stream(device))
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            yield side
        if side is not None:
            torch.cuda.current_stream(device).wait_stream(self.__dict__.get("_fit_stream", side))

    def _fit_epochs(self, model, optim, data_train, N, epochs, batch_size, shuffle, gradient_clip, device, epoch_losses):
        feed = None
        for _ in range(epochs):
            losses = []
            if shuffle:
                perm = np.random.choice(N, N, replace=False)
                data = data_train[perm][0]
            else:
                data = data_train[np.arange(N)][0]
            # large host batches: the next one crosses PCIe under this step (one feed for all epochs)
            feed = _BatchFeed.make(data, N, batch_size, device, feed)
            for idx in range(0, N, batch_size):
                if feed is not None:
                    sample = feed.take(idx)
                else:
                    sample = data[idx: min(idx + batch_size, N)]
                    if not isinstance(sample, torch.Tensor):
                        sample = torch.Tensor(sample)
                    sample = sample.to(device)
                noise = None
                if self.soft_training:
                    noise = self.training_noise_prior.sample([sample.shape[0]]).to(device)
                    sigma = noise.reshape(-1, *([1] * (sample.dim() - 1))).expand_as(sample)
                    sample = sample + torch.normal(torch.zeros_like(sigma), sigma)
                    # conditioning scale recommended by SoftFlow (flows.py:188-191)
                    noise = noise.unsqueeze(-1).detach() * 2 / self.training_noise_prior.high
                dp = self.__dict__.get("_grad_allreduce")      # parallel.data_parallel_training on a flow without the flat arena
                graphed = model._train_graph_step(optim, sample, noise) if (gradient_clip is None and dp is None) else None
                if graphed is not None:
                    losses.append(graphed)
                else:
                    model._zero_grad_for_step(optim)
                    loss = -model.log_prob(sample, context=noise).mean() - model.log_prior()
                    with model._fit_backward_scope():
                        loss.backward()
                    if feed is not None:
                        feed.stage(idx + batch_size)       # host copy + asynchronous upload of the next batch, before the read-back below waits
                    losses.append(float(loss.detach()))
                    if dp is not None:
                        from .parallel import allreduce_gradients
                        allreduce_gradients(model, sample.shape[0], group=dp[0], average=dp[1])
                    if gradient_clip is not None:
                        torch.nn.utils.clip_grad_norm_(model.parameters(), gradient_clip)
                    optim.step()
                    # (drop the step's autograd graph now: it keeps the parameters' AccumulateGrad nodes alive, and those are
                    # bound to the stream they were created on -- a later capture of the step must create its own)
                    del loss
                if feed is not None:
                    feed.done(idx)
                if not self.is_feasible():
                    raise RuntimeError("Model is not invertible")
                model.transform.clear_cache()
            epoch_losses.append(np.mean(losses))

    # ---- Flow.fit: steps of the composite formulation replayed as ONE hipGraph -------------------------------------
    # A step of a flow without a device training path (image-shaped inputs, conditioners with no HIP backward) is some
    # hundreds of small torch ops forward and as many backward: ~10 ms of host time per step whatever the batch (MNIST image
    # configuration, batch 32 .. 4096).  After three eager steps the whole step -- zeroing the gradients, log_prob, backward,
    # the optimiser's update -- is captured once per (batch shape, optimiser) and replayed; a ragged last batch runs eagerly.
    use_train_graph = True        # USFLOWS_AMD_TRAIN_GRAPH=0: off
    train_graph_max_rows = 4096   # flat flows with a device backward: above this the step is not launch-bound any more
    _TRAIN_GRAPH_EAGER_STEPS = 3

    def _train_graph_step(self, optim, sample: torch.Tensor, noise) -> Optional[float]:
        """one optimiser step as a graph replay; the loss as a float, or None when the step has to run eagerly.  The caller
        must not hold the loss tensor (or anything else with a grad_fn over the parameters) of an earlier eager step: the
        parameters' gradient-accumulation nodes stay bound to the eager stream through it, and a capture that reaches over
        to that stream does not survive hipStreamEndCapture."""
        if not (self.use_train_graph and config.train_graph
                and torch.is_tensor(sample) and sample.is_cuda and sample.dtype == torch.float32 and sample.shape[0] > 0
                and not getattr(self, "_train_graph_failed", False) and not torch.cuda.is_current_stream_capturing()):
            return None
        from .sophia import SophiaG
        if not (isinstance(optim, SophiaG) or type(optim) is torch.optim.SGD):
            return None               # (optimisers whose step is known to be free of host synchronisation)
        if any(g_.get("capturable") for g_ in optim.param_groups if isinstance(optim, SophiaG)):
            return None
        # bases that build a fresh, argument-validating torch distribution on every log_prob (DistributionModule) read a
        # flag back to the host inside the step: no stream capture -- unless the density runs on the radial kernels
        # (radial.py: RadialDistribution over LogNormal / Gamma / GammaMM / LogNormalMM, every live configuration's base),
        # which never build the distribution object.  (prior_scale: USFlow.log_prior() sums the LAYERS' priors and
        # BlockAffineTransform inherits BaseTransform.log_prior == 0.0 -- transforms.py:62-64, 874-1029 -- so the term is
        # the number 0.0 for every flow USFlow builds; a layer list with a tensor-valued prior is torch ops on parameters,
        # which a capture records like any other.)
        base = self.base_distribution
        if isinstance(base, DistributionModule) or \
                (isinstance(base, torch.nn.Module) and any(isinstance(m_, DistributionModule) for m_ in base.modules())):
            from . import radial
            if not config.radial or radial.radial_spec(base, sample.device) is None:
                return None
        with torch.enable_grad():
            if self._train_path(sample, noise) is not None:
                # flat flows with a device backward (training.py): the step is ~850 dependent launches of a few microseconds
                # at the reference's batch of 32 -- launch-bound.  Capturable when the loop runs on the flow's own stream
                # (Flow.fit): the tapes replay on the stream they were recorded on.
                if torch.cuda.current_stream(sample.device) != self.__dict__.get("_fit_stream") or \
                        sample.shape[0] > self.train_graph_max_rows:
                    return None
        st = self.__dict__.get("_train_graph_state")
        key = (tuple(sample.shape), None if noise is None else tuple(noise.shape))
        if st is None or st["optim"] is not optim:
            st = self.__dict__["_train_graph_state"] = dict(optim=optim, key=key, seen=0, graph=None, replays=0)
        if st["key"] != key:
            if st["graph"] is not None:
                return None           # ragged last batch of an epoch: eagerly; the captured graph serves the next epoch
            st.update(key=key, seen=0)
        if st["graph"] is None:
            st["seen"] += 1
            if st["seen"] <= self._TRAIN_GRAPH_EAGER_STEPS:
                return None           # allocations, MIOpen searches, the optimiser's state and pointer tables
            params = [p for g_ in optim.param_groups for p in g_["params"]]
            if any(p.grad is not None and not p.grad.is_contiguous() for p in params):
                return None
            sx = sample.detach().clone()
            sc = noise.detach().clone() if noise is not None else None
            gflat = None
            with torch.enable_grad():
                tp = self._train_path(sample, noise)
            if tp is not None and tp.bind_flat_grads():
                # flat flows: the gradients become views of one buffer -- zeroed and accumulated by one launch each
                gflat = tp._gflat
                if hasattr(optim, "prepare_tables"):
                    optim.prepare_tables()
            bound = set() if gflat is None else {id(e[0]) for e in tp._gflat_views.values()}

            def body():
                # flat flows: the bound gradient buffer is zeroed in place (one launch).  Every other gradient is dropped:
                # autograd then TAKES the tensors the backward pass produces as the new .grad (no zeroing launch, no
                # per-parameter add); they are allocated inside the capture, i.e. at the same addresses in every replay,
                # and the optimiser's pointer table is built for exactly those (uploaded after the capture).
                if gflat is not None:
                    gflat.zero_()
                for p in params:
                    if id(p) not in bound:
                        p.grad = None
                if gflat is not None:
                    tp.use_bound_node = True        # (this scope only: training.log_prob_with_grad)
                try:
                    with _unvalidated(self.base_distribution):
                        loss = -self.log_prob(sx, context=sc).mean() - self.log_prior()
                    with self._fit_backward_scope():
                        loss.backward()
                finally:
                    if gflat is not None:
                        tp.use_bound_node = False
                optim.step()
                return loss.detach()

            try:
                torch.cuda.synchronize(sample.device)
                graph = torch.cuda.CUDAGraph()
                cur = torch.cuda.current_stream(sample.device)
                on_own = cur == self.__dict__.get("_fit_stream")
                if hasattr(optim, "defer_uploads"):
                    optim.defer_uploads(True)
                tables = _ext.capture_tables(sample.device)          # (job tables of launches inside the capture: _ext.conv_wgrad)
                try:
                    with tables, (torch.cuda.graph(graph, stream=cur) if on_own else torch.cuda.graph(graph)):
                        sl = body()
                finally:
                    if hasattr(optim, "defer_uploads"):
                        optim.defer_uploads(False)
                if hasattr(optim, "flush_uploads"):
                    optim.flush_uploads()
                tables.upload()
            except Exception as e:      # noqa: BLE001  (an op that cannot be captured: eager steps from now on)
                self._train_graph_failed = True
                self._recover_from_failed_capture(optim, params)
                import traceback
                where = " <- ".join(f"{f.name} ({os.path.basename(f.filename)}:{f.lineno})"
                                    for f in reversed(traceback.extract_tb(e.__traceback__)[-4:]))
                warnings.warn(f"usflows_amd: hipGraph capture of the training step failed ({type(e).__name__}: "
                              f"{str(e).splitlines()[0] if str(e) else ''}; at {where}); Flow.fit runs eager steps",
                              RuntimeWarning)
                return None
            # the graph holds raw addresses: keep what it writes to and reads from alive whatever happens to `p.grad` or to
            # the optimiser's pointer tables afterwards (an eager step in between -- the ragged last batch of an epoch --
            # must not free them: a replay into freed gradient buffers is a GPU memory fault waiting for the allocator)
            keep = ([p.grad for p in params], dict(getattr(optim, "_tables", {}) or {}), tables.keep)
            st.update(graph=graph, x=sx, ctx=sc, loss=sl, params=params, keep=keep)
        st["x"].copy_(sample)
        if st["ctx"] is not None:
            st["ctx"].copy_(noise)
        st["graph"].replay()
        st["replays"] += 1
        if hasattr(optim, "note_graph_replays"):
            optim.note_graph_replays(1)             # (SophiaG's per-parameter step counters live on the host)
        for p in st["params"]:
            torch.autograd.graph.increment_version(p)      # a replay runs no Python: tell the version-keyed caches
        return float(st["loss"])

    def _recover_from_failed_capture(self, optim, params) -> None:
        """A capture that broke off ran no GPU work, but its Python side ran: version counters moved, the engine took its
        parameter pack for refreshed (the refreshing launches were only recorded, then discarded) and the training path
        its tapes for current.  Drop every cache keyed on them -- the next (eager) step rebuilds from the parameters'
        actual values -- and make sure the device is out of capture mode."""
        dev = params[0].device if params else None
        for _ in range(2):              # (the first call may report -- and thereby clear -- the capture's sticky error)
            try:
                torch.cuda.synchronize(dev)
            except Exception:           # noqa: BLE001
                pass
        if dev is not None and dev.type == "cuda" and torch.cuda.current_stream(dev) == self.__dict__.get("_fit_stream"):
            # the capture ran on the loop's own stream and leaves it invalidated: the rest of the loop moves to a fresh one
            # (Flow.fit's stream context restores the caller's stream on exit whatever the current one is by then)
            fresh = torch.cuda.Stream(device=dev)
            torch.cuda.set_stream(fresh)
            self.__dict__["_fit_stream"] = fresh
        if dev is not None and dev.type == "cuda":
            # the broken capture never reached its epilogue: torch's default generator of the device still believes it is
            # being captured ("Offset increment outside graph capture" at the next random draw).  A clone of its state
            # (same seed and offset) is a fresh state object that is not marked as capturing: the generator moves to it.
            try:
                gen = torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]
                gen.graphsafe_set_state(gen.clone_state())
            except Exception:           # noqa: BLE001
                pass
        eng = getattr(self, "_engine_obj", None)
        if eng is not None:
            eng._pack, eng._pack_key = None, None
            eng._plans.clear()
            eng._ws.clear()
        self._train_obj = None
        self.__dict__.pop("_train_graph_state", None)
        if hasattr(optim, "_tables"):
            # (a pointer table built during the broken capture was never uploaded, and a later allocation may land on the
            # addresses it is keyed on)
            optim._tables = {}
            optim._pending_uploads = []
        for p in params:
            if p.grad is not None:
                p.grad = None

    def _fit_backward_scope(self):
        """the scope of a backward pass Flow.fit itself drives: the last sums of its convolution weight gradients may be queued
        until the pass ends (_ext.deferred_sums_scope explains what the opener vouches for).  Not when a process group is up
        without this flow's own data-parallel path in charge of it: a DistributedDataParallel wrapper would hook the
        parameters' gradient accumulators and read the gradients mid-pass."""
        import torch.distributed as dist
        foreign_dp = (dist.is_available() and dist.is_initialized() and self.__dict__.get("_grad_allreduce") is None
                      and getattr(self.__dict__.get("_train_obj"), "grad_allreduce", None) is None)
        return contextlib.nullcontext() if foreign_dp else _ext.deferred_sums_scope()

    def _zero_grad_for_step(self, optim) -> None:
        """``optim.zero_grad()`` of an eager step -- but once a training step of this optimiser has been captured, the
        gradients are zeroed IN PLACE: the captured graph (and the optimiser's pointer table inside it) address exactly these
        buffers, and autograd accumulates into an existing ``.grad`` in place, so eager steps and replays keep sharing them"""
        st = self.__dict__.get("_train_graph_state")
        if st is not None and st.get("graph") is not None and st["optim"] is optim:
            for p, g in zip(st["params"], st["keep"][0]):
                if g is not None:
                    if p.grad is not g:
                        p.grad = g              # (someone set it to None or replaced it: back to the graph's buffer)
                    g.zero_()
            return
        dpg = self.__dict__.get("_dp_grads")
        if dpg is not None and self.__dict__.get("_grad_allreduce") is not None:
            # data-parallel steps of a flow without the flat arena (parallel.bind_dp_grads): the gradients stay views of the
            # one buffer the collective runs over -- zeroed in place by one launch
            params = [p for p in self.parameters() if p.requires_grad]
            if len(params) == len(dpg["views"]) and all(p.grad is v or (p.grad is not None and p.grad.data_ptr() == v.data_ptr())
                                                        for p, v in zip(params, dpg["views"])):
                dpg["flat"].zero_()
                return
        optim.zero_grad()

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


def _pure_pass_mode():
    """a torch dispatch mode for recording a layer loop: keeps every device tensor the pass creates alive (``kept``) and
    notes every torch op on device tensors that is not an allocation or a view (``impure``): such a pass cannot be
    replayed from its recorded HIP calls alone"""
    from torch.utils._python_dispatch import TorchDispatchMode
    from torch.utils._pytree import tree_flatten
    aten = torch.ops.aten
    allowed = set()
    for name in ("empty.memory_format", "empty_like.default", "empty_strided.default", "view.default", "_unsafe_view.default",
                 "detach.default", "alias.default", "expand.default", "as_strided.default", "reshape.default", "t.default",
                 "transpose.int", "select.int", "slice.Tensor", "unsqueeze.default", "squeeze.dim", "_reshape_alias.default",
                 "permute.default", "lift_fresh.default", "squeeze.default", "flatten.using_ints", "unflatten.int"):
        pkt, _, ov = name.partition(".")
        op = getattr(getattr(aten, pkt, None), ov, None)
        if op is not None:
            allowed.add(op)

    class _Mode(TorchDispatchMode):
        def __init__(self):
            super().__init__()
            self.kept, self.impure = [], []

        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            res = func(*args, **(kwargs or {}))
            outs = [t for t in tree_flatten(res)[0] if torch.is_tensor(t) and t.is_cuda]
            if func not in allowed:
                ins = [t for t in tree_flatten((args, kwargs or {}))[0] if torch.is_tensor(t) and t.is_cuda]
                if ins or outs:
                    self.impure.append(str(func))
            self.kept.extend(outs)
            return res

    return _Mode()


def _ladj_is_parameter_only(layer) -> bool:
    """True for the layers whose log|det J| does not depend on the sample (the reference's own layer set)"""
    if isinstance(layer, InverseTransform):
        return _ladj_is_parameter_only(layer.transform)
    if isinstance(layer, BlockAffineTransform):
        return isinstance(layer.block_transform, (LUTransform, HouseholderTransform, SequentialAffineTransform))
    return type(layer) in (ScaleTransform, MaskedCoupling)


class _LogDetSum:
    """log_det = - sum over the layers of log|det J| (flows.py:236-245), collected lazily in training.  The reference subtracts
    every layer's term from a [B] tensor -- three launches per layer and as many in the backward pass, although most terms
    are parameter-only scalars (additive couplings contribute the number 0.0).  Here numbers are summed on the host,
    scalars (0-dim tensors) are stacked and reduced once, and the affine blocks covered by the batched prep kernel enter as ONE
    weighted sum over its stacked log-determinants; only per-sample terms are added as tensors."""

    def __init__(self):
        self.const = 0.0
        self.scalars = []        # (0-dim tensor, weight)
        self.groups = {}         # prep group -> weights per row
        self.vec = None

    def sub(self, t) -> None:
        if isinstance(t, (int, float)):
            self.const -= float(t)
        elif torch.is_tensor(t) and t.dim() == 0:
            self.scalars.append((t, -1.0))
        else:
            self.vec = -t if self.vec is None else self.vec - t

    def take_affine(self, layer) -> bool:
        """a BlockAffineTransform (or its InverseTransform) whose maps come from the prep kernel: weight -/+ n_blocks on its
        row of the stacked log-determinants (transforms.py:1017-1029: one C x C block per position)"""
        from .transforms import BlockAffineTransform, InverseTransform
        from .image_training import current_prep
        inv = isinstance(layer, InverseTransform)
        blk = layer.transform if inv else layer
        if not isinstance(blk, BlockAffineTransform):
            return False
        pr = current_prep(blk.block_transform)
        if pr is None or len(pr) < 6 or pr[5] is None:
            return False
        group, row = pr[5]
        w = self.groups.setdefault(group, [0.0] * len(group[1]))
        w[row] += float(blk.n_blocks) if inv else -float(blk.n_blocks)
        return True

    def add_to(self, lp: torch.Tensor) -> torch.Tensor:
        from .image_training import coef_tensor, prep_stack
        total = None
        for group, w in self.groups.items():
            term = (prep_stack(group) * coef_tensor(w, lp.device)).sum()
            total = term if total is None else total + term
        if self.scalars:
            term = (torch.stack([t for t, _ in self.scalars]) * coef_tensor([w for _, w in self.scalars], lp.device)).sum()
            total = term if total is None else total + term
        if total is not None:
            lp = lp + total
        if self.vec is not None:
            lp = lp + self.vec
        if self.const != 0.0:
            lp = lp + self.const
        return lp


class _unvalidated:
    """context: argument validation of a (nested) torch distribution switched off -- ``_validate_sample`` reads a flag back
    to the host, which a stream capture does not allow (NaN inputs then propagate instead of raising)"""

    def __init__(self, dist):
        self.saved = []
        seen, stack = set(), [dist]
        while stack:
            d = stack.pop()
            if d is None or id(d) in seen:
                continue
            seen.add(id(d))
            if isinstance(d, tdist.Distribution):
                self.saved.append((d, d.__dict__.get("_validate_args", None)))
            for name in ("base_dist", "distribution", "norm_distribution"):
                if name == "distribution" and isinstance(d, DistributionModule):
                    continue           # (a property that BUILDS a validating distribution object -- a host read-back -- per access)
                try:
                    stack.append(getattr(d, name, None))
                except Exception:      # noqa: BLE001  (a property that needs arguments)
                    pass

    def __enter__(self):
        for d, _ in self.saved:
            d._validate_args = False
        return self

    def __exit__(self, *exc):
        for d, v in self.saved:
            if v is None:
                d.__dict__.pop("_validate_args", None)
            else:
                d._validate_args = v
        return False


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
