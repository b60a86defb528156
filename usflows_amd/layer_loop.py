"""The layer loop of ``Flow.log_prob`` (reference flows.py:225-245) on the device -- what ``usflows_amd.flows.Flow`` runs when the
fused launch list of ``FlowEngine`` does not cover a flow (image-shaped flows; flat flows fall back to it under autograd only when the
training path declines): per-layer device passes, the affine runs of conjugated flows composed, the calls of a pass recorded as ONE op
list (``usf_run_ops``) or replayed as a hipGraph at small batches, the log-det terms collected lazily.  A mixin of ``Flow``: the methods
use ``self.layers`` / ``self.base_distribution`` / ``self.parameters()`` only."""
from __future__ import annotations

import contextlib
import warnings
from typing import Optional

import torch
from torch import distributions as tdist

from . import _ext
from .config import config
from .distributions import RadialDistribution, DistributionModule
from .transforms import (BlockAffineTransform, HouseholderTransform, InverseTransform, LUTransform, MaskedCoupling, ScaleTransform,
                         SequentialAffineTransform, _needs_grad)


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


class LayerLoopMixin:
    """see the module docstring"""

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
