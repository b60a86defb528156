"""``Flow.fit`` (reference flows.py:113-210) on the device: the epoch loop with the reference's batching and shuffling, large host
batches handed over under the running step (``_BatchFeed``), and the optimiser step of small batches captured once as a hipGraph and
replayed (``_train_graph_step``).  A mixin of ``usflows_amd.flows.Flow``."""
from __future__ import annotations

import contextlib
import os
import warnings
from typing import Any, Dict, Optional

import numpy as np
import torch
from torch import distributions as tdist

from . import _ext
from .config import config
from .distributions import DistributionModule


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


class FitMixin:
    """see the module docstring"""

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
