"""Training step on the device: ``Flow.log_prob`` under autograd as ONE autograd node whose
forward is the engine's launch list and whose backward is hand-derived -- what ``Flow.fit`` (flows.py:196-199:
``loss = -log_prob(batch).mean() - log_prior()``) asks autograd to differentiate.

forward   the log_prob launch list (engine.py) with every affine output kept in a buffer of its own (the saved
          activations; couplings update their transformed half in place, and their conditioning half -- all the
          backward needs of them -- is untouched), then the base-density tail.
backward  usf_base_logprob_grad_f32 -> per layer, last to first:
            affine    usf_wgrad_f32 (g^T a), usf_colsum_f32, data gradient = usf_linear_f32 on the transposed image
            coupling  hidden activations recomputed by two usf_linear_f32 launches, then per conditioner layer
                      usf_wgrad_f32 / usf_colsum_f32 / usf_linear_f32 (transposed image) / usf_act_grad_f32;
                      the conditioning half of the gradient is updated in place
          (every layer keeps gradient images of its own; the scatter / un-permute copies into the parameter layout are
          queued and leave as one usf_pack_weights_f32 launch per size class behind the last layer)
          then the parameter-sized chain rule, batched over all LU blocks on the f64 MFMA (usf_gemm_f64):
            M^-1 = U^-1 L^-1:   dU = -triu(U^-T G M^-T),   dL = -tril(M^-T G L^-T, -1)
            M    = L U      :   dL += tril(G U^T, -1),     dU += triu(L^T G)            (affine_conjugation)
          Householder factors / longer Sequential chains go through a small torch graph over the prepared fp64
          matrices (parameter-sized library GEMMs).

Supported here: flat inputs, Laplace / Normal base without trainable parameters or a RadialDistribution base (the
node then returns the Lp radius + log-det and the finishing formula stays in torch), inputs that do not require grad,
ConditionalDenseNN / DenseNN conditioners; anything else keeps using the differentiable composite formulation
(flows.py / transforms.py mirrors) -- same results, torch ops.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import os

import torch

from . import _ext
from . import transforms as T
from .config import config
from .engine import FlowEngine, _round_up
from .networks import ConditionalDenseNN, ConvNet, DenseNN


class TrainUnsupported(Exception):
    pass


def _inverse_index(idx: torch.Tensor, D: int, device) -> torch.Tensor:
    """layout index (column -> feature, -1 = padding) -> int32 [D]: feature -> column"""
    inv = torch.full((D,), -1, dtype=torch.int32)
    for c, f in enumerate(idx.tolist()):
        if f >= 0:
            inv[f] = c
    return inv.to(device)


class TrainPath:
    """forward / backward of ``Flow.log_prob`` for one flow (owned by the Flow; shares the FlowEngine)."""

    defer_small_grads = True      # batches <= _ext.GRAD_JOB_MAX_ROWS: weight / bias gradients as one usf_grad_jobs_f32 launch

    def __init__(self, flow):
        self.flow = flow
        self.eng: FlowEngine = flow.engine()
        self._inv: Dict[tuple, torch.Tensor] = {}
        # data-parallel training (parallel.data_parallel_training): (process group | None, average?) -- the whole flat
        # gradient arena goes through ONE all-reduce per step (RCCL over xGMI with the "nccl" backend)
        self.grad_allreduce = None

    # ---- eligibility ----------------------------------------------------------------------------------
    def supported(self, x: torch.Tensor, context) -> bool:
        eng = self.eng
        if eng is None or (context is not None and context.requires_grad):
            return False
        # (an input that requires grad -- round 5: the backward also returns d log_prob / dx, see ``_input_grad``)
        info = self.flow._base_info(x.device)
        if info is None:
            return False
        if info[0] == "radial":
            # RadialDistribution (distributions.py:327-549): trainable loc / norm distribution are fine -- the node
            # returns the Lp radius and the finishing formula stays in torch (O(B), differentiable)
            if getattr(self.flow.base_distribution, "n_batch_dims", 1) != 0:
                return False
        elif any(p.requires_grad for p in self._base_params()) and self._locscale_base() is None:
            return False
        for s in eng.steps:
            if s.kind == "coupling" and not isinstance(s.module.conditioner, (ConditionalDenseNN, DenseNN)):
                # the vector ConvNet with GatedMLP / LayerNormVector blocks (networks.py:206-245, 287-308): chain of linear launches
                # + row passes, backward in _coupling_backward_general
                cond = s.module.conditioner
                if not (isinstance(cond, ConvNet) and cond.is_vector and not cond.is_plain_mlp() and context is None):
                    return False
            if s.kind == "scale" and s.inverted:
                return False
        return True

    def _base_params(self):
        b = self.flow.base_distribution
        return list(b.parameters()) if isinstance(b, torch.nn.Module) else []

    def _locscale_base(self):
        """the base distribution when it is one of the reference's TRAINABLE Laplace / Normal modules (distributions.py:199-238:
        ``loc`` and a softplus-constrained ``scale_unconstrained`` as nn.Parameters, [D] or broadcast over the features): their
        gradients come from usf_base_param_grad_f32"""
        from .distributions import Laplace, Normal
        b = self.flow.base_distribution
        if not isinstance(b, (Laplace, Normal)) or getattr(b, "n_batch_dims", 0) != 0:
            return None
        D = self.eng.D
        for p in (b.loc, b.scale_unconstrained):
            if p.dim() > 1 or (p.dim() == 1 and p.shape[0] not in (1, D)) or p.dtype != torch.float32:
                return None
        return b

    def base_extra_params(self, device=None) -> List[torch.nn.Parameter]:
        """trainable parameters of the base distribution whose gradients the node itself produces (besides the layers')"""
        b = self._locscale_base()
        if b is None:
            return []
        return [p for p in (b.loc, b.scale_unconstrained) if p.requires_grad]

    def _base_tensors(self, ws, info):
        """(base id, loc, scale) for the tail / gradient kernels.  A trainable Laplace / Normal module: persistent [D] buffers
        of the workspace, refilled by every forward pass (the recorded backward launches keep reading the same addresses)"""
        base, loc, scale = self._base_ids(info)
        if info[0] != "radial" and self.base_extra_params():
            D = self.eng.D
            lb, sb = self._buf(ws, "base_loc", 1, D)[0, :D], self._buf(ws, "base_scale", 1, D)[0, :D]
            return base, lb, sb
        return base, loc, scale

    def params(self) -> List[torch.nn.Parameter]:
        return self.eng._params()

    # ---- forward --------------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor, context, want_dx: bool = False):
        eng = self.eng
        x = eng._check_input(x)
        B = x.shape[0]
        dev = x.device
        eng.keep_factors = True
        if want_dx:
            # the input's gradient leaves the fp32-row backward (the planes backward ends in planes of the first layer's weight
            # gradient operand, not in rows)
            keep, eng.use_train_planes = eng.use_train_planes, False
            try:
                plan = eng._plan("backward", B, dev, context is not None, "nat", train=True)
            finally:
                eng.use_train_planes = keep
        else:
            plan = eng._plan("backward", B, dev, context is not None, "nat", train=True)
        self._check_plan(plan)
        if want_dx:
            self._check_input_grad(plan)
        eng._run(plan, x, None, context)
        zname, _, ldn = plan["out_buf"]
        info = self.flow._base_info(dev)
        base, loc, scale = self._base_ids(info)
        if info[0] != "radial" and self.base_extra_params():
            _b, lb, sb = self._base_tensors(plan["ws"], info)
            lb.copy_(loc)
            sb.copy_(scale)
            loc, scale = lb, sb
        # what the backward must find unchanged: the workspace's pass counter (ANY later pass over the same (B, device)
        # workspace -- a training forward or a no_grad log_prob / backward / sample -- overwrites the saved
        # activations, the staged input and the context columns) and the parameter versions
        gen = (plan["ws"]["_gen"], eng._version_key(dev))
        if info[0] == "radial":
            # the radius stays in a plan-owned buffer (the backward's d r / d z needs it); the caller gets a copy
            rbuf = self._buf(plan["ws"], "radius", 1, B)[0, :B]
            _ext.base_logprob(plan["ws"][zname], ldn, B, eng.D, base, loc, None, 0.0, rbuf, None)
            return rbuf.clone(), plan, x, gen
        lp = torch.empty(B, dtype=torch.float32, device=dev)
        _ext.base_logprob(plan["ws"][zname], ldn, B, eng.D, base, loc, scale, 0.0, lp, None,
                          logdet_dev=plan["pk"]["ladj_total"].neg_dev)
        return lp, plan, x, gen

    def revalidate(self, ctx):
        """called by the autograd nodes' backward: (1) parameters changed since the forward (optimiser step /
        load_state_dict in between): the gradient would be evaluated at other parameters than the returned log_prob
        -> error, as torch does for saved tensors modified in place; (2) another pass used the workspace: run this
        node's forward again (same parameters, same input -> same activations)"""
        ws_gen, vkey = ctx.gen
        if self.eng._version_key(ctx.x.device) != vkey:
            raise RuntimeError("usflows_amd: parameters were modified between Flow.log_prob and its backward pass "
                               "(optimiser step or load_state_dict in between); the device training path keeps "
                               "activations, not parameter copies -- call backward() before changing parameters")
        if ctx.plan["ws"].get("_gen") != ws_gen:
            _, ctx.plan, ctx.x, ctx.gen = self.forward(ctx.x, ctx.context, want_dx=getattr(ctx, "want_dx", False))

    @staticmethod
    def _base_ids(info):
        if info[0] == "radial":
            return {1.0: _ext.BASE_LPNORM1, 2.0: _ext.BASE_LPNORM2}.get(info[2], _ext.BASE_LPNORMINF), info[1], None
        return (_ext.BASE_LAPLACE if info[0] == "laplace" else _ext.BASE_NORMAL), info[1], info[2]

    def _check_plan(self, plan):
        if plan["final_gather"] is not None:
            raise TrainUnsupported("layer list does not end in an affine block")
        for g in plan["side"]:
            if g[0] != "gather" or g[1] != 0 or g[2][0] != "user_in":
                raise TrainUnsupported("stand-alone scale layer / mid-flow layout change")
        for m in plan["meta"]:
            if m["kind"] == "affine" and m["post_scale"] is not None:
                raise TrainUnsupported("forward-direction scale fusion")
        # everything the backward relies on is checked HERE, before the autograd node exists, so that an unsupported
        # layer list falls back to the composite formulation in Flow.log_prob instead of failing inside loss.backward()
        if plan.get("train_checked"):
            return
        D = self.eng.D
        for m in plan["meta"]:
            if m["kind"] == "affine":
                blk = m["blk"]
                parts = list(blk.transforms) if isinstance(blk, T.SequentialAffineTransform) else [blk]
                if not all(isinstance(t, (T.LUTransform, T.HouseholderTransform)) for t in parts):
                    raise TrainUnsupported(f"affine part {[type(t).__name__ for t in parts]}")
            elif m["kind"] == "coupling":
                cond = self.eng.steps[m["step"]].module.conditioner
                if isinstance(cond, ConvNet):
                    if not plan["pk"]["coupling"][m["step"]].get("general"):
                        raise TrainUnsupported("ConvNet conditioner without the general (block-wise) pack")
                    continue
                lin = [l for l in cond.layers]
                has_ctx = isinstance(cond, ConditionalDenseNN)
                h = [int(v) for v in cond.hidden_dims]
                want = [(h[0], D)] + ([(h[0], 1)] if has_ctx else []) + \
                       [(h[i + 1], h[i]) for i in range(len(h) - 1)] + [(D, h[-1])]
                got = [tuple(l.weight.shape) for l in lin]
                if got != want:
                    raise TrainUnsupported(f"conditioner weight shapes {got} != {want}")
        plan["train_checked"] = True

    # ---- helpers --------------------------------------------------------------------------------------
    def _inv_idx(self, layout: str, device) -> torch.Tensor:
        key = (layout, str(device))
        if key not in self._inv:
            self._inv[key] = _inverse_index(self.eng._idx(layout), self.eng.D, device)
        return self._inv[key]

    def _buf(self, ws, name, rows, cols, dtype=torch.float32):
        t = ws.get(name)
        if t is None or t.shape[0] != rows or t.shape[1] < cols:
            t = torch.zeros(rows, cols, dtype=dtype, device=ws["zA"].device)
            ws[name] = t
        return t

    def _mat_t(self, pk, blk, which, out_layout, in_layout):
        """transposed image [n_in, n_out] of the affine weight (+ planes): the data-gradient operand"""
        eng = self.eng
        key = (id(blk), which, out_layout, in_layout, "T")
        if key not in pk["mats"]:
            src = pk["affine"][id(blk)][which]
            dev = src.device
            oi, ii = eng._idx_dev(out_layout, dev), eng._idx_dev(in_layout, dev)
            n_out, n_in = int(oi.numel()), int(ii.numel())
            Wt = torch.empty(n_in, n_out, dtype=torch.float32, device=dev)
            planes = None
            if eng._wants_planes(n_in, n_out):
                planes = torch.empty(3, n_in, _round_up(n_out, 32), dtype=torch.bfloat16, device=dev)
            with eng._pk_record(pk):
                _ext.pack_weight(src, ii, n_in, oi, n_out, W=Wt, ldw=n_out, planes=planes, transpose=True)
            pk["mats"][key] = Wt
            if planes is not None:
                pk["mats"][("planes", Wt.data_ptr())] = planes
        return pk["mats"][key]

    def _linear(self, pk, A, a_off, lda, W, Cbuf, c_off, ldc, M, N, K, **kw):
        """one usf_linear_f32 launch (bf16x3 planes attached when the engine's mode wants them)"""
        planes = None
        if self.eng._wants_planes(W.shape[0], K) and W.shape[1] == K:
            planes = self.eng._split_planes(pk, W)
        _ext.linear(A, W, Cbuf, M=M, N=N, K=K, lda=lda, ldw=W.shape[1], ldc=ldc, a_off=a_off, c_off=c_off,
                    W_split=planes, **kw)

    # ---- backward -------------------------------------------------------------------------------------
    def _check_input_grad(self, plan) -> None:
        """d log_prob / dx is served when the first layer of the plan is an affine block that reads the caller's tensor in its
        natural order (what every USFlow layer list gives: [ScaleTransform folded,] BlockAffineTransform first)"""
        first = plan["meta"][0] if plan["meta"] else None
        # ("nat": a zero-padded copy of the caller's rows when D is not a multiple of 4 -- same order)
        if first is None or first["kind"] != "affine" or first["in_buf"] not in ("user_in", "nat") or first["in_layout"] != "nat" \
                or plan.get("planes_train"):
            e = TrainUnsupported("input gradient: the layer list does not start with an affine block on the caller's tensor")
            e.input_grad_only = True            # (parameter-only calls of this flow keep the device path: flows.Flow.log_prob)
            raise e

    def _input_grad(self, plan) -> torch.Tensor:
        """d sum_m g_lp[m] log_prob[m] / dx [B, D] from the gradient the last backward pass left at the first layer's input"""
        g, ld = self._dx
        D = self.eng.D
        dx = g[:, :D]
        sm = plan["meta"][0]["pre_scale"]
        if sm is not None:                                             # x' = x / scale in front of the block (ScaleTransform.backward)
            dx = dx / sm.scale.detach().to(dx.device, torch.float32).reshape(1, D)
        return dx.contiguous() if sm is None else dx

    def backward(self, plan, x, g_lp: torch.Tensor, gsum: Optional[torch.Tensor] = None,
                 into_bound: bool = False, want_dx: bool = False) -> Dict[int, torch.Tensor]:
        """gradients of sum_m g_lp[m] * log_prob(x)[m] w.r.t. every trainable parameter: id(param) -> tensor.

        The launch sequence depends only on the plan: it is recorded on the first call (``_ext.Tape``) and replayed
        afterwards; the per-call inputs (x, g_lp) enter through ``host_op`` closures reading ``self._cur``."""
        dev = x.device
        # gsum: gradient at the log-det output (radial bases: log_prob = f(radius) + logdet, the node returns both);
        # None: log_prob = base(z) + logdet came out of the node itself, so it is sum(g_lp)
        self._cur = dict(x=x, g_lp=g_lp.detach().to(torch.float32).contiguous(),
                         gsum=None if gsum is None else gsum.detach())
        pk = plan["pk"]
        arena = self._arena(plan)
        self._want_dx = bool(want_dx)
        tkey = "bwd_tape_dx" if want_dx else "bwd_tape"              # (the launch sequence differs: the first layer's data gradient)
        tape = plan.get(tkey)
        usable = _ext.TAPES_ENABLED and pk.get("replayable", False)
        with torch.no_grad():
            if usable and tape is not None and tape.stream == _ext.current_stream(dev) and plan.get(tkey + "_pk") is pk:
                _ext.replay(tape)
            else:
                tape = _ext.Tape() if usable else None
                with _ext.record(tape):
                    self._backward_body(plan, arena)
                if tape is not None:
                    tape.stream = _ext.current_stream(dev)
                    tape.dx = getattr(self, "_dx", None)
                plan[tkey], plan[tkey + "_pk"] = tape, pk
            if want_dx and tape is not None:
                self._dx = tape.dx                                     # (a replay writes the same buffer)
            self._last_arena = arena
            if into_bound:
                # the node was built over bound gradients (bind_flat_grads; _LogProbFn took no parameter inputs): what
                # autograd's AccumulateGrad would do per parameter (~290 small adds) is one add over the flat buffer
                gf = self._bound_grads(arena)
                if gf is not None:
                    gf.add_(arena["flat"])
                else:               # the binding went away between forward and backward: per parameter, by hand
                    params = self._arena_params()
                    for pid in arena["touched"]:
                        p, g = params[pid], arena["views"][pid]
                        if p.grad is None:
                            p.grad = g.clone()
                        else:
                            p.grad.add_(g)
                return {}
            flat = arena["flat"].clone()          # autograd may keep what we return: never hand out the arena itself
            self.allreduce_gradients(flat, x.shape[0])
        # parameters the path never reaches (a context layer without context) get no gradient, as under autograd
        out = {pid: flat[o: o + n].view(shape) for pid, (o, n, shape) in arena["slots"].items() if pid in arena["touched"]}
        # the LU factors' gradients leave the chain rule with exact zeros outside their triangles (triu / tril in fp64): LUTransform's
        # mask hooks (transforms.py: one product per factor and pass -- 66 launches of the cfg2 model) skip exactly these tensors
        lu_ids = [id(getattr(lu, nm)) for ch in pk["affine_parts"]["__chunks__"] for lu in ch["lus"] for nm in ("L_raw", "U_raw")]
        masked = [out[i] for i in lu_ids if i in out]
        if masked:
            from .image_training import mark_masked
            mark_masked(masked)
        return out

    # ---- gradients accumulated by one launch ----------------------------------------------------------------
    def _arena_params(self) -> Dict[int, torch.nn.Parameter]:
        ps = list(self.params())
        b = self.flow.base_distribution
        if hasattr(b, "loc") and isinstance(getattr(b, "loc", None), torch.nn.Parameter):
            ps.append(b.loc)
        ps += self.base_extra_params()
        return {id(p): p for p in ps}

    def bind_flat_grads(self) -> bool:
        """Make the ``.grad`` of every parameter this path produces a gradient for a VIEW of one flat fp32 buffer laid
        out like the gradient arena (current values kept).  While the views stay in place (``optimizer.zero_grad(
        set_to_none=True)`` or any reassignment undoes it), ``backward`` adds the whole arena with one launch instead of
        handing ~9 tensors per block to autograd's per-parameter accumulation -- at the reference's batch of 32 rows those
        adds are a tenth of the step.  Flow.fit calls this before it captures the step as a hipGraph and sets
        ``use_bound_node`` for the duration of the captured step only: outside that scope ``log_prob`` stays an ordinary
        autograd node over the parameters (which accumulates into the views in place).  Data-parallel runs stay unbound."""
        ar = self.__dict__.get("_last_arena")
        if ar is None or self.grad_allreduce is not None:
            return False
        params = self._arena_params()
        gf = torch.zeros_like(ar["flat"])
        views = {}
        for pid in ar["touched"]:
            p = params.get(pid)
            if p is None or p.dtype != torch.float32 or p.device != gf.device:
                return False
            o, n, shape = ar["slots"][pid]
            views[pid] = (p, gf[o: o + n].view(shape))
        with torch.no_grad():
            for p, v in views.values():
                if p.grad is not None:
                    v.copy_(p.grad)
                p.grad = v
        self._gflat = gf
        # what the autograd node hangs on while the gradients are bound (its own gradient is a constant zero)
        self._anchor = torch.zeros((), dtype=torch.float32, device=gf.device, requires_grad=True)
        self._anchor_zero = torch.zeros((), dtype=torch.float32, device=gf.device)
        self._gflat_views = {pid: (p, v.data_ptr(), ar["slots"][pid][0]) for pid, (p, v) in views.items()}
        return True

    def grads_bound(self) -> bool:
        """are the parameters' .grad still the views bind_flat_grads made?  (checked when the autograd node is built)"""
        views = self.__dict__.get("_gflat_views")
        if not views or self.grad_allreduce is not None:
            return False
        return all(e[0].grad is not None and e[0].grad.data_ptr() == e[1] for e in views.values())

    def _bound_grads(self, arena) -> Optional[torch.Tensor]:
        gf = self.__dict__.get("_gflat")
        if gf is None or self.grad_allreduce is not None or gf.shape != arena["flat"].shape:
            return None
        views = self._gflat_views
        if len(views) != len(arena["touched"]):
            return None
        for pid in arena["touched"]:
            e = views.get(pid)
            if e is None or e[0].grad is None or e[0].grad.data_ptr() != e[1] or arena["slots"][pid][0] != e[2]:
                return None
        return gf

    def allreduce_gradients(self, flat: torch.Tensor, local_rows: int) -> None:
        """data-parallel training: ONE all-reduce over the flat gradient arena (its last element carries this rank's
        row count).  ``average``: every rank's loss is the MEAN over its own shard, so the global-mean gradient is
        sum_r B_r grad_r / sum_r B_r -- shards of unequal size (a short last batch) are weighted by their row counts and
        an empty shard contributes zeros with weight 0.  Otherwise (losses are sums): the plain sum."""
        if self.grad_allreduce is None:
            return
        import torch.distributed as dist
        group, average = self.grad_allreduce
        if not (dist.is_available() and dist.is_initialized() and
                (dist.get_world_size(group) > 1 or getattr(self, "force_collective", False))):
            return
        if average:
            flat[:-1] *= float(local_rows)
        flat[-1] = float(local_rows)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat[:-1] /= flat[-1].clamp_min(1.0)

    def empty_shard_gradients(self, device) -> Dict[int, torch.Tensor]:
        """a rank whose shard of the batch is empty still has to join the step's collective (the others would hang):
        it contributes a zero arena with weight 0 and receives the same global gradient as every other rank"""
        pk = self.eng.pack(device)
        arena = self._arena(dict(pk=pk, ws=None), device)
        with torch.no_grad():
            flat = torch.zeros_like(arena["flat"])
            self.allreduce_gradients(flat, 0)
        return {pid: flat[o: o + n].view(shape) for pid, (o, n, shape) in arena["slots"].items()}

    def _arena(self, plan, dev=None) -> dict:
        """one flat fp32 buffer for all parameter gradients of this plan; the LU factors' gradients lie batched
        ([n,D,D] L_raw | [n,D,D] U_raw | [n,D] bias, in prepare order) so the chain rule writes each with one copy.
        Its layout depends on the parameter pack only (the same on every rank); one extra element at the end carries
        the rank's row count through the data-parallel all-reduce."""
        ar = plan.get("grad_arena")
        if ar is not None and ar["pk"] is plan["pk"]:
            return ar
        pk = plan["pk"]
        if dev is None:
            dev = plan["ws"]["zA"].device
        slots, off = {}, 0
        lu_views = []
        for ch in pk["affine_parts"]["__chunks__"]:
            n, D = len(ch["lus"]), self.eng.D
            base = off
            for name, per in (("L_raw", D * D), ("U_raw", D * D), ("bias_vector", D)):
                for j, lu in enumerate(ch["lus"]):
                    p = getattr(lu, name)
                    slots[id(p)] = (off + j * per, per, tuple(p.shape))
                off += n * per
            lu_views.append((base, n))
        extra = []
        info = self.flow._base_info(dev)
        if info is not None and info[0] == "radial":
            extra.append(self.flow.base_distribution.loc)
        extra += self.base_extra_params()
        for p in list(self.params()) + extra:
            if id(p) not in slots and p.requires_grad:
                slots[id(p)] = (off, p.numel(), tuple(p.shape))
                off += p.numel()
        flat = torch.zeros(max(off, 1) + 1, dtype=torch.float32, device=dev)
        ar = dict(flat=flat, slots=slots, lu_views=lu_views, pk=pk, touched=set(),
                  views={pid: flat[o: o + n].view(shape) for pid, (o, n, shape) in slots.items()})
        plan["grad_arena"] = ar
        return ar

    def _backward_body(self, plan, arena):
        if plan.get("planes_train"):
            return self._backward_body_planes(plan, arena)
        eng = self.eng
        ws, pk = plan["ws"], plan["pk"]
        dev = ws["zA"].device
        B = ws["zA"].shape[0]
        D = eng.D
        wid = max(eng.LD, eng.LDn)
        gA, gB = self._buf(ws, "gA", B, wid), self._buf(ws, "gB", B, wid)
        glp = self._buf(ws, "g_lp", 1, B)[0, :B]
        _ext.host_op(lambda: (arena["flat"].zero_(), glp.copy_(self._cur["g_lp"])))
        info = self.flow._base_info(dev)
        base, loc, scale = self._base_tensors(ws, info)
        zname, _, ldn = plan["out_buf"]
        grads = arena["views"]
        self._touched = arena["touched"]
        if info[0] == "radial":
            scale = self._buf(ws, "radius", 1, B)[0, :B]              # the forward left r = ||z - loc||_p here
        self._base_param_grads(ws, ws[zname], ldn, glp, B, base, loc, scale, grads)
        _ext.base_logprob_grad(ws[zname], ldn, glp, B, D, base, loc, scale, gA, ldn)
        if info[0] == "radial":
            g_loc = self._grad_slot(grads, self.flow.base_distribution.loc)
            if g_loc is not None:                                     # r depends on z - loc: d/dloc = -sum_m d/dz
                _ext.colsum(gA, g_loc, M=B, N=D, ldy=ldn, alpha=-1.0)
        g_cur, g_other, g_ld = gA, gB, ldn
        aff: Dict[int, dict] = {}            # id(block) -> natural-layout gradients of its usages
        n_aff = sum(1 for m in plan["meta"] if m["kind"] == "affine")
        stacks = dict(G=torch.zeros(max(n_aff, 1), D, D, dtype=torch.float32, device=dev),
                      gs=torch.zeros(max(n_aff, 1), D, dtype=torch.float32, device=dev), next=0)
        self._lu_slot = self._lu_slots(plan)
        if self._lu_slot is not None and any(m["kind"] == "affine" and m["prim"] == "affine_fwd" for m in plan["meta"]):
            # affine_conjugation: a block is also used in its M form (InverseTransform.backward): stacks of their own
            stacks["GM"] = torch.zeros_like(stacks["G"])
            stacks["gsM"] = torch.zeros_like(stacks["gs"])
        self._wmode = 1 if eng.gemm_mode == "bf16x3" else 0      # weight gradients on the same arithmetic as the GEMMs
        first_meta = plan["meta"][0] if plan["meta"] else None
        self._prepare_images(plan, first_meta)
        # gradient images -> parameter layout (scatter / un-permute): every layer has image buffers of its own, so all
        # of these copies wait in one batch and go out as one launch per size class behind the last layer
        # Small batches (the reference trains on 32 rows: tests/explib/mnist.yaml:34): the ~260 weight / bias gradients of
        # a step wait as well and leave as ONE usf_grad_jobs_f32 launch.  Their operands must then outlive the layer loop:
        # every layer gets gradient / activation buffers of its own (a few hundred KB each at these batches) instead of
        # the ping-pong pairs, and a coupling layer that would update, in place, columns a queued job of the coupling layer
        # before it still has to read flushes the queue first (flows with an affine layer between couplings never do).
        self._defer = bool(self.defer_small_grads) and 0 < B <= _ext.GRAD_JOB_MAX_ROWS
        self._g_pending = False
        with _ext.batch_jobs(dev, defer_grads=self._defer):
            for m in reversed(plan["meta"]):
                if m["kind"] == "affine":
                    g_cur, g_other, g_ld = self._affine_backward(plan, m, g_cur, g_other, g_ld, aff, stacks,
                                                                 need_dgrad=(m is not first_meta) or self._want_dx)
                    self._g_pending = False
                else:
                    self._coupling_backward(plan, m, g_cur, g_ld, grads)
                    self._g_pending = self._defer
        self._dx = (g_cur, g_ld) if self._want_dx else None          # the gradient at the first layer's input (natural order)
        _ext.host_op(lambda: self._affine_param_grads(plan, aff, stacks, glp, grads, arena))

    def _base_param_grads(self, ws, z, ldz, glp, B, base, loc, scale, grads):
        """a trainable Laplace / Normal base: d/dloc, d/dscale_unconstrained of sum_m g_lp[m] log_prob[m] -- one pass over the
        latent (usf_base_param_grad_f32), then the softplus chain rule / the reduction of broadcast parameters on [D] tensors"""
        extra = self.base_extra_params()
        if not extra:
            return
        bm = self._locscale_base()
        D = self.eng.D
        tmp = self._buf(ws, "base_pg", 2, D)
        _ext.base_param_grad(z, ldz, glp, B, D, base, loc, scale, tmp)

        def finish():
            def to_shape(v, p):
                return v.reshape(p.shape) if p.numel() == D and p.dim() == 1 else v.sum().reshape(p.shape)
            g = self._grad_slot(grads, bm.loc)
            if g is not None:
                g.copy_(to_shape(tmp[0, :D], bm.loc))
            g = self._grad_slot(grads, bm.scale_unconstrained)
            if g is not None:
                raw = bm.scale_unconstrained.detach()
                g.copy_(to_shape(tmp[1, :D] * torch.sigmoid(raw).expand(D), bm.scale_unconstrained))
        _ext.host_op(finish)

    # ---- the backward pass on the planes pipeline (round 5) -----------------------------------------------------------
    # The forward ran as the inference pipeline does -- usf_pack_planes_f32, usf_gemm_planes_bf16x3, usf_coupling_planes --
    # with every affine output in a planes buffer of its own and the conditioners' hidden activations saved as planes
    # (engine._build_plan_planes(train=True)).  Backwards, layer by layer on a pair of gradient planes buffers:
    #   affine    usf_wgrad_blocked_f32 (gradient planes x saved input planes; the bias gradient from the same pass), then the
    #             data gradient = usf_gemm_planes_bf16x3 on the transposed weight planes
    #   coupling  ONE usf_coupling_planes launch in gate mode (the conditioner's transposed chain with leaky_relu_backward
    #             from the saved activations; it leaves the gradients at the pre-activations as planes), then the three
    #             weight gradients on usf_wgrad_blocked_f32
    # No fp32 row buffer and no operand split anywhere in the loop; the parameter-sized chain rule is the one above.
    def _planes_buf(self, ws, name, B, blocks):
        n = (-(-B // 16)) * blocks * 3072
        t = ws.get(name)
        if t is None or t.numel() < n:
            t = ws[name] = torch.zeros(n, dtype=torch.uint8, device=ws["zA"].device)
        return t

    def _backward_body_planes(self, plan, arena):
        eng = self.eng
        ws, pk = plan["ws"], plan["pk"]
        dev = ws["zA"].device
        B = ws["zA"].shape[0]
        D = eng.D
        nkb = eng.LDp // 32
        nkb_g = max(eng.LDp, eng.LDnp) // 32
        wid = max(eng.LD, eng.LDn)
        gA = self._buf(ws, "gA", B, wid)
        glp = self._buf(ws, "g_lp", 1, B)[0, :B]
        _ext.host_op(lambda: (arena["flat"].zero_(), glp.copy_(self._cur["g_lp"])))
        info = self.flow._base_info(dev)
        base, loc, scale = self._base_tensors(ws, info)
        zname, _, ldn = plan["out_buf"]
        grads = arena["views"]
        self._touched = arena["touched"]
        self._base_param_grads(ws, ws[zname], ldn, glp, B, base, loc, scale, grads)
        gp = [self._planes_buf(ws, "pgA", B, nkb_g), self._planes_buf(ws, "pgB", B, nkb_g)]
        key = ("natp_g", nkb_g, str(dev))
        if key not in self._inv:
            t = torch.full((32 * nkb_g,), -1, dtype=torch.int32)
            t[:D] = torch.arange(D, dtype=torch.int32)
            self._inv[key] = t.to(dev)
        if info[0] != "radial" and (16 * (D + 1) + 96 * nkb_g) * 4 <= 65536:      # (the rows kernel's LDS: 16 rows + the layout tables)
            # Laplace / Normal base: the gradient at the latent goes straight into the planes (one pass over z instead of
            # usf_base_logprob_grad_f32's fp32 rows + their repack)
            _ext.pack_planes(ws[zname], gp[0], M=B, nkb=nkb_g, idx=self._inv[key], ld=ldn, src_cols=D, grad=(base, glp, loc, scale))
        else:
            if info[0] == "radial":
                scale = self._buf(ws, "radius", 1, B)[0, :B]
            _ext.base_logprob_grad(ws[zname], ldn, glp, B, D, base, loc, scale, gA, ldn)
            if info[0] == "radial":
                g_loc = self._grad_slot(grads, self.flow.base_distribution.loc)
                if g_loc is not None:
                    _ext.colsum(gA, g_loc, M=B, N=D, ldy=ldn, alpha=-1.0)
            _ext.pack_planes(gA, gp[0], M=B, nkb=nkb_g, idx=self._inv[key], ld=ldn, src_cols=D)
        cur = 0
        aff: Dict[int, dict] = {}
        n_aff = sum(1 for m in plan["meta"] if m["kind"] == "affine")
        stacks = dict(G=torch.zeros(max(n_aff, 1), D, D, dtype=torch.float32, device=dev),
                      gs=torch.zeros(max(n_aff, 1), D, dtype=torch.float32, device=dev), next=0)
        self._lu_slot = self._lu_slots(plan)
        if self._lu_slot is not None and any(m["kind"] == "affine" and m["prim"] == "affine_fwd" for m in plan["meta"]):
            stacks["GM"] = torch.zeros_like(stacks["G"])
            stacks["gsM"] = torch.zeros_like(stacks["gs"])
        first_meta = plan["meta"][0] if plan["meta"] else None
        with eng._pk_record(pk), _ext.batch_jobs(dev):      # every weight image of the backward, one batched launch
            for m in plan["meta"]:
                if m["kind"] == "coupling":
                    eng.planes_coupling_bwd(pk, m)
                elif m is not first_meta:
                    eng.planes_dgrad_image(pk, m)
        self._defer = False
        # the reductions that end the weight gradients are queued (each with a workspace of its own) and leave as ONE launch behind
        # the layer loop, in front of the batched un-permute jobs that read their outputs (config.wreduce_jobs)
        self._wred = [] if config.wreduce_jobs else None
        with _ext.batch_jobs(dev):
            for m in reversed(plan["meta"]):
                if m["kind"] == "affine":
                    cur = self._affine_backward_planes(plan, m, gp, cur, nkb, nkb_g, aff, stacks, need_dgrad=(m is not first_meta))
                else:
                    self._coupling_backward_planes(plan, m, gp[cur], nkb, nkb_g, grads)
            if self._wred:
                _ext.wgrad_reduce_flush(self._wred, dev)
        _ext.host_op(lambda: self._affine_param_grads(plan, aff, stacks, glp, grads, arena))

    def _wq(self, ws, tag, B, N, K) -> dict:
        """queue + a workspace of this call's own for usf_wgrad_blocked_plan_f32 (empty: the undeferred call)"""
        if getattr(self, "_wred", None) is None:
            return {}
        return dict(queue=self._wred, ws=self._buf(ws, f"WGws{tag}", 1, _ext.wgrad_blocked_workspace(B, N, K)))

    def _affine_backward_planes(self, plan, m, gp, cur, nkb, nkb_g, aff, stacks, need_dgrad):
        eng = self.eng
        ws, pk = plan["ws"], plan["pk"]
        dev = ws["zA"].device
        B = ws["zA"].shape[0]
        D = eng.D
        blk = m["blk"]
        which = "Minv" if m["prim"] == "affine_bwd" else "M"
        n_out, n_in = m["N"], m["K"]
        wid = max(eng.LD, eng.LDn)
        Gp = self._buf(ws, f"Gp{m['op']}", wid, wid)
        gs = self._buf(ws, f"gs{m['op']}", 1, wid)
        _ext.wgrad_blocked(gp[cur], nkb_g, 0, ws[m["in_buf"]], nkb, 0, Gp, M=B, N=n_out, K=n_in, ldg=Gp.shape[1], colsum=gs,
                           **self._wq(ws, f"a{m['op']}", B, n_out, n_in))
        k = self._lu_slot[id(blk)] if self._lu_slot is not None else stacks["next"]
        stacks["next"] += 1
        if self._lu_slot is not None and which == "M":
            G_nat, gs_nat = stacks["GM"][k], stacks["gsM"][k]
        else:
            G_nat, gs_nat = stacks["G"][k], stacks["gs"][k]
        _ext.pack_weight(Gp, self._inv_idx(m["out_layout"], dev), D, self._inv_idx(m["in_layout"], dev), D,
                         W=G_nat, ldw=D, ld_src=Gp.shape[1])
        _ext.pack_weight(gs, None, 1, self._inv_idx(m["out_layout"], dev), D, W=gs_nat, ldw=D, ld_src=gs.shape[1])
        rec = aff.setdefault(id(blk), dict(blk=blk, uses=[]))
        rec["uses"].append(dict(which=which, G=G_nat, gsum=gs_nat, pre_scale=m["pre_scale"], row=k,
                                pre_sub_folded=bool(m.get("pre_sub_folded"))))
        if not need_dgrad:
            return cur
        Wt = eng.planes_dgrad_image(pk, m)
        nk = (eng.LDnp if m["out_layout"] == "natp" else eng.LDp) // 32
        _ext.gemm_planes(gp[cur], Wt, M=B, a_nkb=nkb_g, nk=nk, C_planes=gp[1 - cur], c_nkb=nkb_g, c_kb0=0, c_kbn=nkb)
        return 1 - cur

    def _seg_sel(self, m, which: str, device) -> torch.Tensor:
        """int32 [D]: feature -> position relative to the first block of the coupling's conditioning ('p') / transformed ('t')
        block range (-1: the feature is not in that set)"""
        key = ("segsel", m["step"], which, str(device))
        if key not in self._inv:
            feat = m["feat_p"] if which == "p" else m["feat_t"]
            kb0 = m["kb_p0"] if which == "p" else m["kb_t0"]
            sel = torch.full((self.eng.D,), -1, dtype=torch.int32)
            for pos, f in enumerate(feat.tolist()):
                if f >= 0:
                    sel[f] = pos - 32 * kb0
            self._inv[key] = sel.to(device)
        return self._inv[key]

    def _coupling_backward_planes(self, plan, m, g, nkb, nkb_g, grads):
        eng = self.eng
        ws, pk = plan["ws"], plan["pk"]
        cp = pk["coupling"][m["step"]]
        cond = eng.steps[m["step"]].module.conditioner
        raw = cp["raw"]
        dev = raw["device"]
        B = ws["zA"].shape[0]
        h = list(raw["h"])
        nl = len(h)
        sign = m["sign"]
        hs = [ws[n_] for n_ in m["hidden_planes"]]
        dh = [self._planes_buf(ws, f"pD{j}", B, 8) for j in range(nl)]
        _ext.coupling_planes_op(eng.planes_coupling_bwd_op(pk, m, g, nkb_g, B, hs, dh), dev)
        lin = list(cond.layers)
        has_ctx = isinstance(cond, ConditionalDenseNN)
        first_l, last_l = lin[0], lin[-1]
        hidden_l = lin[2:-1] if has_ctx else lin[1:-1]
        wmax = max(eng.LDp, 256)
        gimg = lambda tag: self._buf(ws, f"gW{m['step']}_{tag}", wmax, wmax)      # noqa: E731
        gvec = lambda tag: self._buf(ws, f"gb{m['step']}_{tag}", 1, wmax)          # noqa: E731
        # output layer: Y = the gradient at the transformed blocks (unchanged by the launch above), A = the last hidden activations
        n_t = int((m["feat_t"] >= 0).nonzero().max().item()) + 1 - 32 * m["kb_t0"]
        gW, gb = gimg("out"), gvec("out")
        _ext.wgrad_blocked(g, nkb_g, m["kb_t0"], hs[nl - 1], 8, 0, gW, M=B, N=n_t, K=256, ldg=gW.shape[1], alpha=sign,
                           colsum=gb, cs_alpha=sign, **self._wq(ws, f"c{m['step']}o", B, n_t, 256))
        tsel = self._seg_sel(m, "t", dev)
        self._scatter_weight(grads, last_l.weight, gW, rows_sel=tsel, n_rows=eng.D, cols_sel=None, n_cols=h[-1])
        self._scatter_vec(grads, last_l.bias, gb, tsel, eng.D)
        # hidden layers, last to first: Y = the gradient at layer j's pre-activation, A = layer j - 1's activations
        for j in range(nl - 1, 0, -1):
            l = hidden_l[j - 1]
            gW, gb = gimg(f"h{j}"), gvec(f"h{j}")
            _ext.wgrad_blocked(dh[j], 8, 0, hs[j - 1], 8, 0, gW, M=B, N=h[j], K=256, ldg=gW.shape[1], alpha=sign,
                               colsum=gb, cs_alpha=sign, **self._wq(ws, f"c{m['step']}h{j}", B, h[j], 256))
            self._scatter_weight(grads, l.weight, gW, None, h[j], None, h[j - 1])
            self._scatter_vec(grads, l.bias, gb, None, h[j])
        # input layer: A = the conditioning blocks of the layer's own z buffer (what the forward's conditioner saw)
        n_p = int((m["feat_p"] >= 0).nonzero().max().item()) + 1 - 32 * m["kb_p0"]
        gW, gb = gimg("in"), gvec("in")
        _ext.wgrad_blocked(dh[0], 8, 0, ws[m["buf"]], nkb, m["kb_p0"], gW, M=B, N=h[0], K=n_p, ldg=gW.shape[1], alpha=sign,
                           colsum=gb, cs_alpha=sign, **self._wq(ws, f"c{m['step']}i", B, h[0], n_p))
        self._scatter_weight(grads, first_l.weight, gW, None, h[0], self._seg_sel(m, "p", dev), eng.D)
        self._scatter_vec(grads, first_l.bias, gb, None, h[0])

    def _prepare_images(self, plan, first_meta):
        """every weight image the backward reads (unfused conditioner layers, transposed images of the data-gradient
        GEMMs), built up front in two batched launches that join the pack's tape"""
        eng = self.eng
        pk = plan["pk"]
        dev = plan["ws"]["zA"].device
        with eng._pk_record(pk):
            with _ext.batch_jobs(dev):
                for m in plan["meta"]:
                    if m["kind"] == "coupling":
                        eng._unfused_pack(pk, pk["coupling"][m["step"]])
            with _ext.batch_jobs(dev):
                for m in plan["meta"]:
                    if m["kind"] == "coupling":
                        if self._fused_cbwd(m, plan["ws"]["zA"].shape[0]):
                            eng._fused_pack_bwd(pk, pk["coupling"][m["step"]])      # the transposed set of the fused kernel
                            continue
                        un = eng._unfused_pack(pk, pk["coupling"][m["step"]])
                        if un.get("general"):
                            for W in self._general_weights(un):
                                self._transposed(pk, W)
                            continue
                        for W, _b in un["layers"]:
                            self._transposed(pk, W)
                        self._transposed(pk, un["W_out"])
                    elif m is not first_meta or getattr(self, "_want_dx", False):
                        which = "Minv" if m["prim"] == "affine_bwd" else "M"
                        self._mat_t(pk, m["blk"], which, m["out_layout"], m["in_layout"])

    def _fused_cbwd(self, m, B) -> bool:
        """the data-gradient chain of this coupling layer's conditioner as ONE launch of the fused kernel (engine.
        coupling_backward_op): where the forward ran fused and left its hidden activations, unless USFLOWS_AMD_FUSED_CBWD=0"""
        # (queued gradient jobs read the d_h buffers after the layer loop: the tiny-layer form gives every layer buffers of its own)
        return (bool(m.get("hidden_saved_fused")) and config.fused_cbwd
                and (bool(m.get("tiny")) or not (bool(self.defer_small_grads) and 0 < B <= _ext.GRAD_JOB_MAX_ROWS)))

    def _lu_slots(self, plan) -> Optional[Dict[int, int]]:
        """id(affine block) -> row of the batched gradient stacks, when the batched chain rule applies: every affine
        block is one LU factor -- bare, Sequential([LU]) or Sequential([LU, Householder with one vector]) (the USFlow
        constructor's default) --, used once in its M^-1 form and at most once in its M form (affine_conjugation),
        all LU factors prepared in one chunk.  Also fills self._hh: row -> Householder module (or absent)."""
        pk = plan["pk"]
        chunks = pk["affine_parts"]["__chunks__"]
        self._hh = {}
        if len(chunks) != 1:
            return None
        idx = {id(lu): j for j, lu in enumerate(chunks[0]["lus"])}
        slots, seen = {}, set()
        for m in plan["meta"]:
            if m["kind"] != "affine":
                continue
            blk = m["blk"]
            leaf, hh = blk, None
            if isinstance(blk, T.SequentialAffineTransform):
                parts = list(blk.transforms)
                if len(parts) == 1:
                    leaf = parts[0]
                elif (len(parts) == 2 and isinstance(parts[1], T.HouseholderTransform) and parts[1].nvs == 1):
                    leaf, hh = parts
            use = (id(leaf), m["prim"])
            if not isinstance(leaf, T.LUTransform) or use in seen or (m["prim"] == "affine_fwd" and m["pre_scale"] is not None):
                return None
            seen.add(use)
            slots[id(blk)] = idx[id(leaf)]
            if hh is not None:
                self._hh[idx[id(leaf)]] = hh
        if {u[0] for u in seen if u[1] == "affine_bwd"} != set(idx):
            return None
        return slots

    # ---- affine layers --------------------------------------------------------------------------------
    def _affine_backward(self, plan, m, g_cur, g_other, g_ld, aff, stacks, need_dgrad):
        eng = self.eng
        ws, pk = plan["ws"], plan["pk"]
        dev = ws["zA"].device
        B = ws["zA"].shape[0]
        blk = m["blk"]
        which = "Minv" if m["prim"] == "affine_bwd" else "M"
        oi = eng._idx_dev(m["out_layout"], dev)
        ii = eng._idx_dev(m["in_layout"], dev)
        n_out, n_in = int(oi.numel()), int(ii.numel())
        # weight gradient in the image layout, then back to the natural [D, D] layout of the block's matrix
        wid = max(eng.LD, eng.LDn)
        Gp = self._buf(ws, f"Gp{m['op']}", wid, wid)           # own image per layer: its un-permute job runs later
        # large batches: the data-gradient GEMM runs first and leaves the planes it makes of g; the weight gradient then
        # multiplies them with the planes the forward GEMM left of the layer input (no operand split in its loaders)
        from_planes = (need_dgrad and not self._defer and self._wmode == 1 and m.get("in_planes") is not None
                       and m["in_planes"] in ws and eng.wgrad_from_planes(B, n_out, n_in))
        if from_planes:
            gpl = ws.get("gpl")
            if gpl is None or gpl.shape[1] != -(-B // 32) * 32 or gpl.shape[2] < -(-n_out // 32) * 32:
                gpl = ws["gpl"] = _ext.row_planes(B, max(n_out, eng.LD, eng.LDn), dev)
            Wt = self._mat_t(pk, blk, which, m["out_layout"], m["in_layout"])
            self._linear(pk, g_cur, 0, g_ld, Wt, g_other, 0, n_in, B, n_in, n_out, planes_out=gpl)
            gs = self._buf(ws, f"gs{m['op']}", 1, wid)
            cs_fused = (config.fused_bias                                                  # the bias gradient from the same pass
                        and bool(_ext.load().usf_wgrad_planes_colsum_ok(B, n_out, n_in)))
            _ext.wgrad_planes(gpl, ws[m["in_planes"]], Gp, M=B, N=n_out, K=n_in, ldg=Gp.shape[1], colsum=gs if cs_fused else None)
            if not cs_fused:
                _ext.colsum(g_cur, gs, M=B, N=n_out, ldy=g_ld)
        elif m["in_buf"] == "user_in":
            # the caller's tensor changes from call to call: issued through the wrapper on every replay
            _ext.host_op(lambda g=g_cur, ld=g_ld: _ext.wgrad(g, self._cur["x"], Gp, M=B, N=n_out, K=n_in, ldy=ld,
                                                            lda=self._cur["x"].shape[1], ldg=Gp.shape[1], mode=self._wmode,
                                                            defer=False))
        else:
            _ext.wgrad(g_cur, ws[m["in_buf"]], Gp, M=B, N=n_out, K=n_in, ldy=g_ld, lda=m["in_ld"], ldg=Gp.shape[1],
                       mode=self._wmode)
        if not from_planes:
            gs = self._buf(ws, f"gs{m['op']}", 1, wid)
            _ext.colsum(g_cur, gs, M=B, N=n_out, ldy=g_ld)
        D = eng.D
        k = self._lu_slot[id(blk)] if self._lu_slot is not None else stacks["next"]
        stacks["next"] += 1
        if self._lu_slot is not None and which == "M":
            G_nat, gs_nat = stacks["GM"][k], stacks["gsM"][k]
        else:
            G_nat, gs_nat = stacks["G"][k], stacks["gs"][k]
        _ext.pack_weight(Gp, self._inv_idx(m["out_layout"], dev), D, self._inv_idx(m["in_layout"], dev), D,
                         W=G_nat, ldw=D, ld_src=Gp.shape[1])
        _ext.pack_weight(gs, None, 1, self._inv_idx(m["out_layout"], dev), D, W=gs_nat, ldw=D, ld_src=gs.shape[1])
        rec = aff.setdefault(id(blk), dict(blk=blk, uses=[]))
        rec["uses"].append(dict(which=which, G=G_nat, gsum=gs_nat, pre_scale=m["pre_scale"], row=k))
        if from_planes:
            return g_other, g_cur, n_in
        if need_dgrad:
            Wt = self._mat_t(pk, blk, which, m["out_layout"], m["in_layout"])
            if self._defer:                       # g_cur waits for this layer's queued gradient jobs: never written again
                dst = self._buf(ws, f"gD{m['op']}", B, wid)
                self._linear(pk, g_cur, 0, g_ld, Wt, dst, 0, n_in, B, n_in, n_out)
                return dst, g_other, n_in
            self._linear(pk, g_cur, 0, g_ld, Wt, g_other, 0, n_in, B, n_in, n_out)
            return g_other, g_cur, n_in
        return g_cur, g_other, g_ld

    # ---- coupling layers ------------------------------------------------------------------------------
    @staticmethod
    def _general_weights(un):
        ws_ = [un["first"][0], un["W_out"]]
        for e in un["blocks"]:
            ws_ += [e[k][0] for k in ("lin", "l1", "l2", "proj") if k in e]
        return ws_

    def _coupling_backward_general(self, plan, m, g_cur, g_ld, grads):
        """backward of a coupling layer whose conditioner is the vector ConvNet with GatedMLP / LayerNormVector blocks
        (reference networks.py:206-245, 287-308; forward: engine._general_coupling_ops).  The conditioner runs once more from the
        layer's saved input with every intermediate kept (x_j, f(x_j), the hidden activations, [val, gate], the projected skip),
        then block by block backwards: usf_gated_norm_rows_bwd_f32 (layer norm + gate), usf_wgrad_f32 / usf_colsum_f32 for the
        parameters, usf_linear_f32 on the transposed images for the data gradients (the (Leaky)ReLU derivative in its epilogue:
        USF_ACT_GATE) -- what torch.autograd derives from the module under Flow.fit."""
        eng = self.eng
        ws, pk = plan["ws"], plan["pk"]
        cp = pk["coupling"][m["step"]]
        cond = eng.steps[m["step"]].module.conditioner
        un = eng._unfused_pack(pk, cp)
        raw = cp["raw"]
        dev = raw["device"]
        zbuf = ws[m["buf"]]
        B = zbuf.shape[0]
        LD, hm = eng.LD, eng.hmax
        sign = m["sign"]
        act, slope = cp["act"], cp["slope"]
        r4 = lambda n: _round_up(n, 4)                                 # noqa: E731
        own = f"_{m['step']}" if self._defer else ""                   # (queued gradient jobs read these after the layer loop)
        buf = lambda tag, j, w=hm: self._buf(ws, f"GC{tag}{j}{own}", B, w)   # noqa: E731
        first_m, blocks_m, final_m = cond.block_view()
        nb = len(un["blocks"])
        h0 = raw["h"][0]
        pass_n, pass_off, tr_n, tr_off = cp["pass_n"], cp["pass_off"], cp["tr_n"], cp["tr_off"]
        gate = lambda hbuf: dict(act=_ext.ACT_GATE, slope=slope, addend=hbuf, ldadd=hm) if act != _ext.ACT_NONE else {}   # noqa: E731
        # ---- 1. forward again, everything kept ----
        X = [buf("X", j) for j in range(nb + 1)]
        A = [buf("A", j) for j in range(nb)]
        Wf, bf = un["first"]
        self._linear(pk, zbuf, pass_off, LD, Wf, X[0], 0, hm, B, Wf.shape[0], pass_n, bias=bf)
        _ext.gated_norm_rows(X[0], M=B, C_cols=h0, c_pad=r4(h0), ld_skip=hm, out_act=A[0], ld_act=hm, act=act, slope=slope)
        Tb, VG, S = {}, {}, {}
        for j, e in enumerate(un["blocks"]):
            wi, wo = e["w_in"], e["w_out"]
            nxt = A[j + 1] if j + 1 < nb else None
            ln = e.get("ln")
            kw_ln = dict(gamma=ln[0], beta=ln[1], eps=e["eps"]) if ln is not None else {}
            kw_act = dict(out_act=nxt, ld_act=hm, act=act, slope=slope) if nxt is not None else {}
            Tb[j] = buf("T", j)
            if "lin" in e:
                W, b = e["lin"]
                self._linear(pk, A[j], 0, hm, W, Tb[j], 0, hm, B, W.shape[0], W.shape[1], bias=b)
                _ext.gated_norm_rows(Tb[j], M=B, C_cols=wo, c_pad=r4(wo), ld_skip=hm, out=X[j + 1], ld_out=hm, **kw_ln, **kw_act)
            else:
                W1, b1 = e["l1"]
                W2, b2 = e["l2"]
                self._linear(pk, A[j], 0, hm, W1, Tb[j], 0, hm, B, W1.shape[0], W1.shape[1], bias=b1, act=act, slope=slope)
                VG[j] = buf("VG", j, 2 * hm)
                self._linear(pk, Tb[j], 0, hm, W2, VG[j], 0, 2 * hm, B, W2.shape[0], W2.shape[1], bias=b2)
                skip = X[j]
                if "proj" in e:
                    Wp, bp = e["proj"]
                    S[j] = buf("S", j)
                    self._linear(pk, X[j], 0, hm, Wp, S[j], 0, hm, B, Wp.shape[0], Wp.shape[1], bias=bp)
                    skip = S[j]
                _ext.gated_norm_rows(skip, M=B, C_cols=wo, c_pad=r4(wo), ld_skip=hm, vg=VG[j], ld_vg=2 * hm, gate_off=r4(wo),
                                     out=X[j + 1], ld_out=hm, **kw_ln, **kw_act)
        # ---- 2. the output Linear: Y = the gradient at the transformed half, A = x_nb ----
        gimg = lambda tag: self._buf(ws, f"gWG{m['step']}_{tag}", max(2 * hm, LD), max(hm, LD))     # noqa: E731
        gvec = lambda tag: self._buf(ws, f"gbG{m['step']}_{tag}", 1, max(2 * hm, LD))              # noqa: E731
        W_out = un["W_out"]
        w_last = un["blocks"][-1]["w_out"]
        gW, gb = gimg("out"), gvec("out")
        _ext.wgrad(g_cur, X[nb], gW, M=B, N=tr_n, K=W_out.shape[1], ldy=g_ld, lda=hm, ldg=gW.shape[1], y_off=tr_off, alpha=sign,
                   mode=self._wmode)
        tsel = self._sel_inv(raw["tr_idx"], dev)
        self._scatter_weight(grads, final_m.weight, gW, rows_sel=tsel, n_rows=eng.D, cols_sel=None, n_cols=w_last)
        _ext.colsum(g_cur, gb, M=B, N=tr_n, ldy=g_ld, y_off=tr_off, alpha=sign)
        self._scatter_vec(grads, final_m.bias, gb, tsel, eng.D)
        DX = [buf("DX", j) for j in range(nb + 1)]
        Wt = self._transposed(pk, W_out)
        self._linear(pk, g_cur, tr_off, g_ld, Wt, DX[nb], 0, hm, B, Wt.shape[0], Wt.shape[1])
        # ---- 3. the blocks, last to first ----
        for j in range(nb - 1, -1, -1):
            e, bm = un["blocks"][j], blocks_m[j]
            wi, wo = e["w_in"], e["w_out"]
            ln = e.get("ln")
            gated = "lin" not in e
            skip = Tb[j] if not gated else (S[j] if "proj" in e else X[j])
            DR = buf("DR", j)
            DVG = buf("DVG", j, 2 * hm) if gated else None
            DYX = buf("DYX", j) if ln is not None else None
            _ext.gated_norm_rows_bwd(skip, DX[j + 1], DR, M=B, C_cols=wo, c_pad=r4(wo), ld_skip=hm, ld_dy=hm, ld_d_skip=hm,
                                     vg=VG.get(j), ld_vg=2 * hm, gate_off=r4(wo), d_vg=DVG, ld_d_vg=2 * hm,
                                     gamma=None if ln is None else ln[0], eps=e["eps"], dy_xh=DYX, ld_dy_xh=hm)
            if ln is not None:
                self._colsum_to(grads, bm["ln"].weight, DYX, B, wo, hm, sign)
                self._colsum_to(grads, bm["ln"].bias, DX[j + 1], B, wo, hm, sign)
            if not gated:
                W, _b = e["lin"]
                gW = gimg(f"l{j}")
                _ext.wgrad(DR, A[j], gW, M=B, N=W.shape[0], K=W.shape[1], ldy=hm, lda=hm, ldg=gW.shape[1], alpha=sign, mode=self._wmode)
                self._scatter_weight(grads, bm["lin"].weight, gW, None, wo, None, wi)
                self._colsum_to(grads, bm["lin"].bias, DR, B, wo, hm, sign)
                Wt = self._transposed(pk, W)
                self._linear(pk, DR, 0, hm, Wt, DX[j], 0, hm, B, Wt.shape[0], Wt.shape[1], **gate(X[j]))
                continue
            W1, _b1 = e["l1"]
            W2, _b2 = e["l2"]
            # second Linear: Y = d[val, gate] (value rows [0, wp), gate rows [wp, 2 wp)), A = the hidden activations
            gW = gimg(f"b{j}")
            _ext.wgrad(DVG, Tb[j], gW, M=B, N=W2.shape[0], K=W2.shape[1], ldy=2 * hm, lda=hm, ldg=gW.shape[1], alpha=sign, mode=self._wmode)
            self._scatter_weight(grads, bm["l2"].weight, gW, self._two_sel(wo, dev), 2 * wo, None, wo)
            g2 = self._grad_slot(grads, bm["l2"].bias)
            if g2 is not None:
                _ext.colsum(DVG, g2[:wo], M=B, N=wo, ldy=2 * hm, alpha=sign)
                _ext.colsum(DVG, g2[wo:], M=B, N=wo, ldy=2 * hm, y_off=r4(wo), alpha=sign)
            DT = buf("DT", j)
            Wt2 = self._transposed(pk, W2)
            self._linear(pk, DVG, 0, 2 * hm, Wt2, DT, 0, hm, B, Wt2.shape[0], Wt2.shape[1], **gate(Tb[j]))
            gW = gimg(f"a{j}")
            _ext.wgrad(DT, A[j], gW, M=B, N=W1.shape[0], K=W1.shape[1], ldy=hm, lda=hm, ldg=gW.shape[1], alpha=sign, mode=self._wmode)
            self._scatter_weight(grads, bm["l1"].weight, gW, None, wo, None, wi)
            self._colsum_to(grads, bm["l1"].bias, DT, B, wo, hm, sign)
            Wt1 = self._transposed(pk, W1)
            if "proj" in e:
                Wp, _bp = e["proj"]
                gW = gimg(f"p{j}")
                _ext.wgrad(DR, X[j], gW, M=B, N=Wp.shape[0], K=Wp.shape[1], ldy=hm, lda=hm, ldg=gW.shape[1], alpha=sign, mode=self._wmode)
                self._scatter_weight(grads, bm["proj"].weight, gW, None, wo, None, wi)
                self._colsum_to(grads, bm["proj"].bias, DR, B, wo, hm, sign)
                # d x_j = f'(x_j) (dT W1) + dr Wp
                self._linear(pk, DT, 0, hm, Wt1, DX[j], 0, hm, B, Wt1.shape[0], Wt1.shape[1], **gate(X[j]))
                Wtp = self._transposed(pk, Wp)
                self._linear(pk, DR, 0, hm, Wtp, DX[j], 0, hm, B, Wtp.shape[0], Wtp.shape[1], residual=DX[j], ldr=hm)
            else:
                # d x_j = f'(x_j) (dT W1) + dr (the skip connection; USF_ACT_GATE takes no residual: one elementwise launch more)
                self._linear(pk, DT, 0, hm, Wt1, DX[j], 0, hm, B, Wt1.shape[0], Wt1.shape[1], **gate(X[j]))
                _ext.add_rows(DX[j], DR, self._ones(hm, dev))
        # ---- 4. the first Linear and the conditioning half of the gradient ----
        gW = gimg("in")
        _ext.wgrad(DX[0], zbuf, gW, M=B, N=Wf.shape[0], K=pass_n, ldy=hm, lda=LD, ldg=gW.shape[1], a_off=pass_off, alpha=sign,
                   mode=self._wmode)
        self._scatter_weight(grads, first_m.weight, gW, None, h0, self._sel_inv(raw["pass_idx"], dev), eng.D)
        self._colsum_to(grads, first_m.bias, DX[0], B, h0, hm, sign)
        if self._g_pending:
            _ext.flush_jobs()          # the coupling layer before this one queued reads of columns this update rewrites
        Wtf = self._transposed(pk, Wf)
        self._linear(pk, DX[0], 0, hm, Wtf, g_cur, pass_off, g_ld, B, pass_n, Wtf.shape[1],
                     residual=g_cur, r_off=pass_off, ldr=g_ld, res_sign=sign)

    def _ones(self, n: int, device) -> torch.Tensor:
        key = ("ones", n, str(device))
        if key not in self._inv:
            self._inv[key] = torch.ones(n, dtype=torch.float32, device=device)
        return self._inv[key]

    def _two_sel(self, wo: int, device) -> torch.Tensor:
        """row selector of a GatedMLP's second Linear: parameter row i -> image row (value rows [0, wo) stay, gate rows
        [wo, 2 wo) sit at [wp, wp + wo), wp = wo rounded up to 4)"""
        key = ("twosel", wo, str(device))
        if key not in self._inv:
            t = torch.arange(2 * wo, dtype=torch.int32)
            t[wo:] += _round_up(wo, 4) - wo
            self._inv[key] = t.to(device)
        return self._inv[key]

    def _coupling_backward(self, plan, m, g_cur, g_ld, grads):
        eng = self.eng
        ws, pk = plan["ws"], plan["pk"]
        cp = pk["coupling"][m["step"]]
        if cp.get("general"):
            return self._coupling_backward_general(plan, m, g_cur, g_ld, grads)
        layer = eng.steps[m["step"]].module
        cond = layer.conditioner
        un = eng._unfused_pack(pk, cp)
        raw = cp["raw"]
        dev = raw["device"]
        zbuf = ws[m["buf"]]
        B = zbuf.shape[0]
        LD, hmax = eng.LD, eng.hmax
        sign = m["sign"]
        h, hp = raw["h"], cp["hidden"]
        nl = len(un["layers"])
        act, slope = cp["act"], cp["slope"]
        # 1. hidden activations again (the conditioning half of the saved buffer is what the forward saw)
        saved_fused = bool(m.get("hidden_saved_fused"))       # large batches: the fused forward kernel left them (engine.py)
        saved = saved_fused or bool(m.get("hidden_saved"))    # ... or the unfused forward's GEMMs wrote them into the layer's own buffers
        own = f"_{m['step']}" if (self._defer or saved) else ""         # deferred gradient jobs read these after the layer loop
        hbufs = [self._buf(ws, f"Hs{j}{own}", B, hmax) for j in range(nl)]
        src, src_off, src_ld, src_K = zbuf, cp["pass_off"], LD, cp["pass_n"]
        # (small batches: the forward plan left the hidden activations in exactly these buffers -- engine.py, `hidden_saved`)
        recompute = not saved
        for j, (W, b) in enumerate(un["layers"] if recompute else []):
            kw = {}
            if j == 0 and m["use_ctx"]:
                self._linear(pk, ws["ctx4"], 0, 4, un["W_ctx4"], ws["P"], 0, hmax, B, hp[0], 4, bias=un["b_ctx"])
                kw = dict(addend=ws["P"], ldadd=hmax)
            self._linear(pk, src, src_off, src_ld, W, hbufs[j], 0, hmax, B, W.shape[0], src_K, bias=b, act=act,
                         slope=slope, **kw)
            src, src_off, src_ld, src_K = hbufs[j], 0, hmax, W.shape[0]
        lin = list(cond.layers)
        has_ctx = isinstance(cond, ConditionalDenseNN)
        first_l, last_l = lin[0], lin[-1]
        hidden_l = lin[2:-1] if has_ctx else lin[1:-1]
        fused_bwd = self._fused_cbwd(m, B)
        if fused_bwd:
            # ONE launch: g_P += s * MLP^T(g_T) with the (Leaky)ReLU backward from the saved activations; the gradients at the
            # hidden activations land in Dh{j} for the weight gradients below
            dh = [self._buf(ws, f"DhF{j}{own if self._defer else ''}", B, hmax) for j in range(nl)]
            op = eng.coupling_backward_op(pk, cp, g_cur.data_ptr(), g_ld, B, sign, hbufs, dh,
                                          act=_ext.ACT_GATE if act != _ext.ACT_NONE else _ext.ACT_NONE)
            if self._g_pending:
                _ext.flush_jobs()      # the coupling layer before this one queued reads of columns this launch rewrites
            _ext.coupling_op(op, dev)
        # 2. output layer: d_out = gradient at the transformed half (unchanged by the layer: out_T = z_T + s MLP)
        tr_n, tr_off = cp["tr_n"], cp["tr_off"]
        W_out = un["W_out"]                                   # [tr_n, hp_last]
        gimg = lambda tag: self._buf(ws, f"gW{m['step']}_{tag}", max(hmax, LD), max(hmax, LD))
        gW = gimg("out")
        gb = self._buf(ws, f"gb{m['step']}", 1, max(hmax, LD))
        # large batches: the bias gradient (column sums of the same Y) rides in the weight-gradient pass (usf_wgrad_bias_f32)
        fuse = lambda n, k, ldy, lda: (not self._defer and config.fused_bias  # noqa: E731
                                       and _ext.wgrad_bias_ok(B, n, k, ldy, lda, self._wmode))
        fused_out = fuse(tr_n, hp[-1], g_ld, hmax)
        _ext.wgrad(g_cur, hbufs[-1], gW, M=B, N=tr_n, K=hp[-1], ldy=g_ld, lda=hmax, ldg=gW.shape[1], y_off=tr_off,
                   alpha=sign, mode=self._wmode, **(dict(colsum=gb, cs_alpha=sign) if fused_out else {}))
        self._scatter_weight(grads, last_l.weight, gW, rows_sel=self._sel_inv(raw["tr_idx"], dev), n_rows=eng.D,
                             cols_sel=None, n_cols=h[-1])
        if not fused_out:
            _ext.colsum(g_cur, gb, M=B, N=tr_n, ldy=g_ld, y_off=tr_off, alpha=sign)
        self._scatter_vec(grads, last_l.bias, gb, self._sel_inv(raw["tr_idx"], dev), eng.D)
        # d_h = d_out W_out  (sign is applied where the result leaves the MLP)
        if self._defer:
            d_bufs = [self._buf(ws, f"Dh{j}{own}", B, hmax) for j in range(nl)]
            d_of = lambda j: d_bufs[j]                            # noqa: E731  (gradient at hidden layer j's output)
        else:
            d_bufs = [self._buf(ws, "Dh0", B, hmax), self._buf(ws, "Dh1", B, hmax)]
            d_of = lambda j: d_bufs[(nl - 1 - j) & 1]             # noqa: E731
        if fused_bwd:
            d_of = lambda j: dh[j]                                # noqa: E731
        d = d_of(nl - 1)
        # (the (Leaky)ReLU backward from the saved layer output rides in the GEMM's epilogue: USF_ACT_GATE)
        gate = lambda hbuf: dict(act=_ext.ACT_GATE, slope=slope, addend=hbuf, ldadd=hmax) if act != _ext.ACT_NONE else {}
        if not fused_bwd:
            Wt = self._transposed(pk, W_out)                  # [hp_last, tr_n4]
            self._linear(pk, g_cur, tr_off, g_ld, Wt, d, 0, hmax, B, hp[-1], Wt.shape[1], **gate(hbufs[-1]))
        # 3. hidden layers, last to first
        for j in range(nl - 1, 0, -1):
            W, _b = un["layers"][j]                           # [hp_j, hp_{j-1}]
            l = hidden_l[j - 1]
            gW = gimg(f"h{j}")
            gbias = self._grad_slot(grads, l.bias) if (hp[j] == h[j] and fuse(hp[j], hp[j - 1], hmax, hmax)) else None
            _ext.wgrad(d, hbufs[j - 1], gW, M=B, N=hp[j], K=hp[j - 1], ldy=hmax, lda=hmax, ldg=gW.shape[1], alpha=sign,
                       mode=self._wmode, **(dict(colsum=gbias, cs_alpha=sign) if gbias is not None else {}))
            self._scatter_weight(grads, l.weight, gW, None, h[j], None, h[j - 1])
            if gbias is None:
                self._colsum_to(grads, l.bias, d, B, h[j], hmax, sign)
            d_next = d_of(j - 1)
            if not fused_bwd:
                self._linear(pk, d, 0, hmax, self._transposed(pk, W), d_next, 0, hmax, B, hp[j - 1], hp[j], **gate(hbufs[j - 1]))
            d = d_next
        # 4. input layer
        W_in, _b = un["layers"][0]                            # [hp0, pass_n]
        pass_n, pass_off = cp["pass_n"], cp["pass_off"]
        gW = gimg("in")
        gbias = self._grad_slot(grads, first_l.bias) if (hp[0] == h[0] and fuse(hp[0], pass_n, hmax, LD)) else None
        _ext.wgrad(d, zbuf, gW, M=B, N=hp[0], K=pass_n, ldy=hmax, lda=LD, ldg=gW.shape[1], a_off=pass_off, alpha=sign,
                   mode=self._wmode, **(dict(colsum=gbias, cs_alpha=sign) if gbias is not None else {}))
        self._scatter_weight(grads, first_l.weight, gW, None, h[0], self._sel_inv(raw["pass_idx"], dev), eng.D)
        if gbias is None:
            self._colsum_to(grads, first_l.bias, d, B, h[0], hmax, sign)
        if has_ctx:
            ctx_l = lin[1]
            if m["use_ctx"]:
                gW = gimg("ctx")
                _ext.wgrad(d, ws["ctx4"], gW, M=B, N=hp[0], K=4, ldy=hmax, lda=4, ldg=gW.shape[1], alpha=sign)
                self._scatter_weight(grads, ctx_l.weight, gW, None, h[0], None, 1)
                self._colsum_to(grads, ctx_l.bias, d, B, h[0], hmax, sign)
        # conditioning half of the gradient: g_P += s * d W_in   (in place)
        if fused_bwd:
            return                     # (the fused launch above already updated the conditioning half)
        if self._g_pending:
            _ext.flush_jobs()          # the coupling layer before this one queued reads of columns this update rewrites
        self._linear(pk, d, 0, hmax, self._transposed(pk, W_in), g_cur, pass_off, g_ld, B, pass_n, hp[0],
                     residual=g_cur, r_off=pass_off, ldr=g_ld, res_sign=sign)

    def _transposed(self, pk, W: torch.Tensor) -> torch.Tensor:
        """[K, N4] image of W^T for a conditioner weight image W [N, K] (N padded to 4: it is the linear kernel's K),
        built from the raw parameter the image came from (no dependency on the image: both can sit in one batch)"""
        key = ("T", W.data_ptr())
        if key not in pk["mats"]:
            N, K = W.shape
            N4 = _round_up(N, 4)
            Wt = torch.empty(K, N4, dtype=torch.float32, device=W.device)
            planes = None
            if self.eng._wants_planes(K, N4):
                planes = torch.empty(3, K, _round_up(N4, 32), dtype=torch.bfloat16, device=W.device)
            with self.eng._pk_record(pk):
                origin = pk["mats"].get(("imgsrc", W.data_ptr()))
                if origin is not None:
                    src, out_sel, n_out, in_sel, n_in = origin           # W[o, c] = src[out_sel[o], in_sel[c]]
                    rows = self.eng._sel(out_sel[:n_out].cpu(), N4, W.device) if N4 != n_out else out_sel
                    _ext.pack_weight(src, in_sel, n_in, rows, N4, W=Wt, ldw=N4, planes=planes, transpose=True)
                else:
                    _ext.flush_jobs()
                    sel = self.eng._iarange(N, N4, W.device)
                    _ext.pack_weight(W, None, K, sel, N4, W=Wt, ldw=N4, planes=planes, transpose=True)
            pk["mats"][key] = Wt
            if planes is not None:
                pk["mats"][("planes", Wt.data_ptr())] = planes
        return pk["mats"][key]

    def _sel_inv(self, idx: torch.Tensor, device) -> torch.Tensor:
        """segment selector (position -> feature) -> int32 [D]: feature -> position in the segment (-1: not in it)"""
        key = ("selinv", idx.data_ptr(), str(device))
        if key not in self._inv:
            self._inv[key] = _inverse_index(idx, self.eng.D, device)
            self._inv[("keep", idx.data_ptr())] = idx
        return self._inv[key]

    def _grad_slot(self, grads, p: torch.Tensor) -> Optional[torch.Tensor]:
        """the parameter's view in the gradient arena (zeroed at the start of every backward)"""
        if not p.requires_grad:
            return None
        self._touched.add(id(p))
        return grads.get(id(p))

    def _scatter_weight(self, grads, p, img, rows_sel, n_rows, cols_sel, n_cols):
        g = self._grad_slot(grads, p)
        if g is None:
            return
        if g.shape != (n_rows, n_cols):
            raise RuntimeError(f"usflows_amd internal: gradient image {n_rows}x{n_cols} vs parameter {tuple(g.shape)} "
                               "(shapes are validated in TrainPath._check_plan)")
        _ext.pack_weight(img, rows_sel, n_rows, cols_sel, n_cols, W=g, ldw=n_cols, ld_src=img.shape[1])

    def _scatter_vec(self, grads, p, vec_img, sel, n):
        g = self._grad_slot(grads, p)
        if g is None:
            return
        _ext.pack_weight(vec_img, None, 1, sel, n, W=g, ldw=n, ld_src=vec_img.shape[1])

    def _colsum_to(self, grads, p, d, B, n, ld, sign):
        g = self._grad_slot(grads, p)
        if g is not None:
            _ext.colsum(d, g, M=B, N=n, ldy=ld, alpha=sign)

    # ---- parameter-sized chain rule -------------------------------------------------------------------
    def _affine_param_grads(self, plan, aff, stacks, g_lp, grads, arena):
        eng = self.eng
        pk = plan["pk"]
        dev = g_lp.device
        Gsum = g_lp.double().sum() if self._cur["gsum"] is None else self._cur["gsum"].double().reshape(())
        # log-det: log_prob = base(z) - ladj_total, ladj_total = sum_steps (+/-) ladj(step)  (flows.py:236-245)
        coef: Dict[int, float] = {}
        for s in eng.steps:
            if s.kind == "affine":
                coef[id(s.module)] = coef.get(id(s.module), 0.0) + (1.0 if s.inverted else -1.0)
        if self._lu_slot is not None:
            return self._lu_chain_rule_batched(plan, aff, stacks, Gsum, coef, grads, arena)
        lu_acc: Dict[int, dict] = {}         # id(LUTransform) -> dict(dMinv, dM, db, c) fp64
        for rec in aff.values():
            blk = rec["blk"]
            prep = pk["affine"][id(blk)]
            dMinv = dM = None
            db = torch.zeros(eng.D, dtype=torch.float64, device=dev)
            for u in rec["uses"]:
                G = u["G"].double()
                gs = u["gsum"].double()
                if u["which"] == "Minv":
                    if u["pre_scale"] is not None:
                        if u.get("pre_sub_folded"):
                            G = self._unfold_head(u["pre_scale"], G, gs, prep["b"])
                        G = self._scale_grad(u["pre_scale"], prep["Minv"], G, Gsum, grads)
                    # y = (a - b) Minv^T:  dMinv = g^T a - gsum (x) b ,  db = -Minv^T gsum
                    G = G - torch.outer(gs, prep["b"])
                    dMinv = G if dMinv is None else dMinv + G
                    db = db - prep["Minv"].t() @ gs
                else:
                    # y = a M^T + b (InverseTransform.backward = block forward, transforms.py:370-376)
                    dM = G if dM is None else dM + G
                    db = db + gs
            self._to_leaves(blk, prep, dMinv, dM, db, coef.get(id(blk), 0.0) * Gsum, lu_acc, grads, pk)
        self._lu_param_grads(pk, lu_acc, grads)

    @staticmethod
    def _unfold_head(scale_mod, G3, gs, b):
        """planes pipeline: the head packs a' = x / s - b, so the first layer's weight gradient arrives as G3 = g^T a';
        g^T x = (G3 + gsum (x) b) diag(s) is what _scale_grad takes (fp64, parameter-sized)"""
        s64 = scale_mod.scale.detach().double()
        return (G3 + torch.outer(gs, b)) * s64[None, :]

    def _scale_grad(self, scale_mod, Minv, G2, Gsum, grads):
        """first layer: a = x / s.  G2 = g^T x; returns g^T a = G2 diag(1/s) and writes
        ds_d = -(1/s_d^2) sum_n Minv[n,d] G2[n,d] - Gsum / s_d   (transforms.py:116-144)"""
        s64 = scale_mod.scale.detach().double()
        gsc = self._grad_slot(grads, scale_mod.scale)
        if gsc is not None:
            ds = -(Minv * G2).sum(0) / (s64 * s64) - Gsum / s64
            gsc.copy_(ds.reshape(gsc.shape))
        return G2 / s64[None, :]

    def _lu_chain_rule_batched(self, plan, aff, stacks, Gsum, coef, grads, arena):
        """One pass over [n, D, D] stacks for all affine blocks (block matrix M_t = M_lu H, H = the block's Householder
        factor or I; transforms.py:1457-1476):
            usage M_t^-1 (y = (a - b_t) M_t^-T):  dMinv_t = g^T a - gsum (x) b_t,   db_t = -M_t^-T gsum
            usage M_t    (y = a M_t^T + b_t)    :  dM_t = g^T a,                     db_t += gsum
            M_t^-1 = H^T M_lu^-1, M_t = M_lu H, b_t = b_lu H:
               dMinv_lu = H dMinv_t,  dM_lu = dM_t H^T,  db_lu = H db_t,
               dH = M_lu^-1 dMinv_t^T + M_lu^T dM_t + b_lu (x) db_t
            M_lu^-1 = U^-1 L^-1, M_lu = L U:
               dU = -triu(U^-T (dMinv_lu M_lu^-T)) + triu(L^T dM_lu) + c diag(1/U_jj)
               dL = -tril((M_lu^-T dMinv_lu) L^-T, -1) + tril(dM_lu U^T, -1)
            H = w_0 (I - 2 v v^T / v.v),  A = w_0^T dH:   dv = -2 (A + A^T) v / s + 4 (v^T A v) v / s^2,  s = v.v
        everything batched on the f64 MFMA (usf_gemm_f64) / batched torch ops."""
        pk = plan["pk"]
        ch = pk["affine_parts"]["__chunks__"][0]
        out = ch["out"]
        n, D = len(ch["lus"]), self.eng.D
        DD = D * D
        dev = out["Minv"].device
        bat = dict(batch=n, M=D, N=D, K=D, lda=D, ldb=D, ldc=D, strideC=DD)
        f64 = lambda: torch.zeros(n, D, D, dtype=torch.float64, device=dev)
        Minv_lu, b_lu = out["Minv"], ch["b"]
        has_hh = bool(self._hh)
        blocks = self._lu_blocks(plan)
        if has_hh:
            # block-level matrices M_t^-1 / b_t and H per row (H = I where the block has no Householder factor)
            eye = stacks.setdefault("eye", torch.eye(D, dtype=torch.float64, device=dev))
            H = torch.stack([pk["affine_parts"][id(self._hh[j])]["M"] if j in self._hh else eye for j in range(n)])
            Minv_t = torch.stack([pk["affine"][id(b_)]["Minv"] for b_ in blocks])
            b_t = torch.stack([pk["affine"][id(b_)]["b"] for b_ in blocks])
        else:
            H, Minv_t, b_t = None, Minv_lu, b_lu
        G = stacks["G"][:n].double()
        gs = stacks["gs"][:n].double()
        for rec in aff.values():
            for u in rec["uses"]:
                if u["pre_scale"] is not None:
                    if u.get("pre_sub_folded"):
                        G[u["row"]] = self._unfold_head(u["pre_scale"], G[u["row"]], gs[u["row"]], b_t[u["row"]])
                    G[u["row"]] = self._scale_grad(u["pre_scale"], Minv_t[u["row"]], G[u["row"]], Gsum, grads)
        dMinv_t = G - gs[:, :, None] * b_t[:, None, :]
        # db_t = -M_t^-T gsum, one row-vector product per block on the library's batched fp64 kernel (no BLAS call on the path)
        db_t = torch.empty(n, D, dtype=torch.float64, device=dev)
        _ext.gemm_f64(gs.contiguous(), Minv_t.contiguous(), db_t, batch=n, M=1, N=D, K=D, lda=D, ldb=D, ldc=D, strideA=D, strideB=DD,
                      strideC=D, alpha=-1.0)
        dM_t = None
        if "GM" in stacks:
            dM_t = stacks["GM"][:n].double()
            db_t = db_t + stacks["gsM"][:n].double()
        if has_hh:
            dMinv_lu, dH = stacks.setdefault("dMinv_lu", f64()), stacks.setdefault("dH", f64())
            _ext.gemm_f64(H, dMinv_t, dMinv_lu, strideA=DD, strideB=DD, **bat)                    # H dMinv_t
            _ext.gemm_f64(Minv_lu, dMinv_t, dH, transB=True, strideA=DD, strideB=DD, **bat)       # M_lu^-1 dMinv_t^T
            db_lu = torch.bmm(H, db_t.unsqueeze(2)).squeeze(2)
            dH += b_lu[:, :, None] * db_t[:, None, :]
            dM_lu = None
            if dM_t is not None:
                dM_lu = stacks.setdefault("dM_lu", f64())
                _ext.gemm_f64(dM_t, H, dM_lu, transB=True, strideA=DD, strideB=DD, **bat)         # dM_t H^T
                _ext.gemm_f64(out["M"], dM_t, dH, transA=True, strideA=DD, strideB=DD, beta=1.0, **bat)   # + M_lu^T dM_t
            self._householder_grads(dH, stacks, grads, n, D, dev)
        else:
            dMinv_lu, dM_lu, db_lu = dMinv_t, dM_t, db_t
        # ---- LU factors
        tmp = stacks.get("T")
        if tmp is None:      # zero-filled once: tiles the masked products never write must stay finite for triu / tril
            tmp = stacks["T"] = torch.zeros(3, n, D, D, dtype=torch.float64, device=dev)
        T1, dU, dL = tmp[0], tmp[1], tmp[2]
        tinv = out["tri_inv"]                                   # [2n, D, D]: L^-1 at even, (U^-1)^T at odd rows
        # only one triangle of each result survives (triu / tril below) and the inverses are triangular: the tile masks
        # and k-range hints of usf_gemm_f64 cut the four products to ~3/8 of their dense cost
        _ext.gemm_f64(dMinv_lu, Minv_lu, T1, transB=True, strideA=DD, strideB=DD, tri=8, **bat)   # triu(G M^-T)
        _ext.gemm_f64(tinv, T1, dU, alpha=-1.0, strideA=2 * DD, strideB=DD, a_off=DD, tri=8 + 3, **bat)   # -U^-T (.)
        _ext.gemm_f64(Minv_lu, dMinv_lu, T1, transA=True, strideA=DD, strideB=DD, tri=16, **bat)  # tril(M^-T G)
        _ext.gemm_f64(T1, tinv, dL, transB=True, alpha=-1.0, strideA=DD, strideB=2 * DD, tri=16 + 4, **bat)  # -(.) L^-T
        if "coef" not in stacks:        # built once: a host-to-device copy here would drain the stream every step
            stacks["coef"] = torch.tensor([coef.get(id(b_), 0.0) for b_ in blocks], dtype=torch.float64, device=dev)
        c = stacks["coef"] * Gsum
        TL = TU = None
        if dM_lu is not None:
            # the M = L U usages:  dL += tril(G U^T, -1),  dU += triu(L^T G)
            tri = out["tri"]                                    # [2n, D, D]: L at even, U^T at odd rows
            TL, TU = tmp[0], stacks.setdefault("T2", f64())
            _ext.gemm_f64(dM_lu, tri, TL, strideA=DD, strideB=2 * DD, b_off=DD, tri=16, **bat)    # tril(G U^T)
            _ext.gemm_f64(tri, dM_lu, TU, transA=True, strideA=2 * DD, strideB=DD, tri=8, **bat)  # triu(L^T G)
        # tril(dL + TL, -1), triu(dU + TU) + diag(c / U_jj) (transforms.py:1303-1320) and the converting copies into the fp32
        # gradient arena: one pass (usf_lu_grad_finish_f64) instead of eight over [n, D, D] tensors
        base, _n = arena["lu_views"][0]
        flat = arena["flat"]
        _ext.lu_grad_finish(dL, dU, TL, TU, c.contiguous(), out["tri"], n, D, flat[base: base + n * DD],
                            flat[base + n * DD: base + 2 * n * DD])
        flat[base + 2 * n * DD: base + 2 * n * DD + n * D].view(n, D).copy_(db_lu)
        for lu in ch["lus"]:
            for name in ("L_raw", "U_raw", "bias_vector"):
                if getattr(lu, name).requires_grad:
                    arena["touched"].add(id(getattr(lu, name)))

    def _householder_grads(self, dH, stacks, grads, n, D, dev):
        """dLoss/dv of H = w_0 (I - 2 v v^T / v.v) for the rows that have a Householder factor (one vector each)"""
        rows = sorted(self._hh)
        hhs = [self._hh[j] for j in rows]
        if "w0T" not in stacks:     # w_0 is a fixed permutation (requires_grad = False, transforms.py:789)
            stacks["w0T"] = torch.stack([h.w_0.detach().double().t() for h in hhs]).contiguous()
            stacks["hh_rows"] = torch.tensor(rows, dtype=torch.long, device=dev)
        A = torch.bmm(stacks["w0T"], dH[stacks["hh_rows"]])                          # w_0^T dH
        v = torch.stack([h.vk_householder.detach()[0].double() for h in hhs])        # [m, D]
        s_ = (v * v).sum(1, keepdim=True)
        Av = torch.bmm(A, v.unsqueeze(2)).squeeze(2)
        ATv = torch.bmm(A.transpose(1, 2), v.unsqueeze(2)).squeeze(2)
        vAv = (v * Av).sum(1, keepdim=True)
        dv = -2.0 * (Av + ATv) / s_ + 4.0 * vAv * v / (s_ * s_)
        for i, h in enumerate(hhs):
            g = self._grad_slot(grads, h.vk_householder)
            if g is not None:
                g.copy_(dv[i].reshape(g.shape))

    def _lu_blocks(self, plan):
        """affine blocks in the row order of the batched stacks"""
        order = sorted(((row, bid) for bid, row in self._lu_slot.items()))
        by_id = {id(m["blk"]): m["blk"] for m in plan["meta"] if m["kind"] == "affine"}
        return [by_id[bid] for _row, bid in order]

    def _to_leaves(self, blk, prep, dMinv, dM, db, ladj_coef, lu_acc, grads, pk):
        """gradients w.r.t. a block's (M^-1, M, bias) -> its LU factors' accumulators / Householder parameters"""
        if isinstance(blk, T.SequentialAffineTransform) and len(blk.transforms) == 1:
            blk = blk.transforms[0]              # eye @ M_1 == M_1: the same matrices
        if isinstance(blk, T.LUTransform):
            a = lu_acc.setdefault(id(blk), dict(lu=blk, dMinv=None, dM=None, db=None, c=0.0))
            for k_, v in (("dMinv", dMinv), ("dM", dM), ("db", db)):
                if v is not None:
                    a[k_] = v if a[k_] is None else a[k_] + v
            a["c"] = a["c"] + ladj_coef
            return
        # general composition: small torch graph over the prepared fp64 matrices of the parts
        parts = list(blk.transforms) if isinstance(blk, T.SequentialAffineTransform) else [blk]
        with torch.enable_grad():
            leaves = []
            mats = []
            for t in parts:
                if isinstance(t, T.LUTransform):
                    pr = pk["affine_parts"][id(t)]
                    Mi = pr["Minv"].detach().clone().requires_grad_(True)
                    Mm = pr["M"].detach().clone().requires_grad_(True)
                    bb = pr["b"].detach().clone().requires_grad_(True)
                    leaves.append(("lu", t, Mi, Mm, bb))
                    mats.append((Mm, Mi, bb))
                elif isinstance(t, T.HouseholderTransform):
                    Hw = t._construct_householder_permutation().double()        # differentiable w.r.t. vk
                    leaves.append(("hh", t, Hw))
                    mats.append((Hw, Hw.t(), torch.zeros(t.dim, dtype=torch.float64, device=Hw.device)))
                else:
                    raise RuntimeError(f"usflows_amd internal: affine part {type(t).__name__} "
                                       "(validated in TrainPath._check_plan)")
            M, Minv, b = mats[0]
            for m_, mi_, b_ in mats[1:]:                     # transforms.py:1457-1476
                M = M @ m_
                b = b @ m_ + b_
                Minv = mi_ @ Minv
            proxy = (b * db).sum()
            if dMinv is not None:
                proxy = proxy + (Minv * dMinv).sum()
            if dM is not None:
                proxy = proxy + (M * dM).sum()
            wanted, owners = [], []
            for lf in leaves:
                if lf[0] == "lu":
                    wanted += [lf[2], lf[3], lf[4]]
                    owners += [(lf[1], "dMinv"), (lf[1], "dM"), (lf[1], "db")]
                else:
                    for p in lf[1].parameters():
                        if p.requires_grad:
                            wanted.append(p)
                            owners.append((p, "param"))
            got = torch.autograd.grad(proxy, wanted, allow_unused=True)
        for (owner, kind), g in zip(owners, got):
            if kind == "param":
                if g is not None:
                    slot = self._grad_slot(grads, owner)
                    slot.add_(g.to(slot.dtype))
                continue
            a = lu_acc.setdefault(id(owner), dict(lu=owner, dMinv=None, dM=None, db=None, c=0.0))
            if g is not None:
                a[kind] = g if a[kind] is None else a[kind] + g
        for lf in leaves:
            if lf[0] == "lu":                                # the block's log-det is the sum of its LU parts'
                a = lu_acc.setdefault(id(lf[1]), dict(lu=lf[1], dMinv=None, dM=None, db=None, c=0.0))
                a["c"] = a["c"] + ladj_coef

    def _lu_param_grads(self, pk, lu_acc, grads):
        """batched over all LU blocks: (dMinv, dM, db) -> (L_raw, U_raw, bias_vector).grad on the f64 MFMA"""
        if not lu_acc:
            return
        accs = list(lu_acc.values())
        n = len(accs)
        D = self.eng.D
        dev = pk["affine_parts"][id(accs[0]["lu"])]["Minv"].device
        DD = D * D
        f64 = lambda *shape: torch.empty(*shape, dtype=torch.float64, device=dev)
        stack = lambda key: torch.stack([pk["affine_parts"][id(a["lu"])][key] for a in accs]).contiguous()
        dL = torch.zeros(n, D, D, dtype=torch.float64, device=dev)
        dU = torch.zeros(n, D, D, dtype=torch.float64, device=dev)
        bat = dict(batch=n, strideA=DD, strideB=DD, strideC=DD, M=D, N=D, K=D, lda=D, ldb=D, ldc=D)
        if any(a["dMinv"] is not None for a in accs):
            G = torch.stack([a["dMinv"] if a["dMinv"] is not None else torch.zeros(D, D, dtype=torch.float64, device=dev)
                             for a in accs]).contiguous()
            Minv, Linv, UinvT = stack("Minv"), stack("Linv"), stack("Uinv_t")
            T1, T2 = f64(n, D, D), f64(n, D, D)
            _ext.gemm_f64(G, Minv, T1, transB=True, **bat)                       # G Minv^T
            _ext.gemm_f64(UinvT, T1, dU, alpha=-1.0, **bat)                      # -U^-T (G Minv^T)
            _ext.gemm_f64(Minv, G, T2, transA=True, **bat)                       # Minv^T G
            _ext.gemm_f64(T2, Linv, dL, transB=True, alpha=-1.0, **bat)          # -(Minv^T G) L^-T
        if any(a["dM"] is not None for a in accs):
            G = torch.stack([a["dM"] if a["dM"] is not None else torch.zeros(D, D, dtype=torch.float64, device=dev)
                             for a in accs]).contiguous()
            L, Ut = stack("L"), stack("Ut")
            _ext.gemm_f64(G, Ut, dL, beta=1.0, **bat)                            # + G U^T
            _ext.gemm_f64(L, G, dU, transA=True, beta=1.0, **bat)                # + L^T G
        for i, a in enumerate(accs):
            lu = a["lu"]
            gU = self._grad_slot(grads, lu.U_raw)
            if gU is not None:
                du = dU[i].triu()
                if a["c"] is not None and not (isinstance(a["c"], float) and a["c"] == 0.0):
                    # d/dU_jj of c * sum_j log|U_jj|  (transforms.py:1303-1320)
                    du = du + torch.diag(a["c"] / lu.U_raw.detach().double().diagonal())
                gU.add_(du.to(gU.dtype))
            gL = self._grad_slot(grads, lu.L_raw)
            if gL is not None:
                gL.add_(dL[i].tril(-1).to(gL.dtype))
            gb = self._grad_slot(grads, lu.bias_vector)
            if gb is not None and a["db"] is not None:
                gb.add_(a["db"].to(gb.dtype))


class _LogProbFn(torch.autograd.Function):
    """``Flow.log_prob`` as one autograd node (forward: launch list; backward: TrainPath.backward)."""

    @staticmethod
    def forward(ctx, path: TrainPath, x, context, *params):
        ctx.want_dx = bool(ctx.needs_input_grad[1])
        lp, plan, xc, gen = path.forward(x, context, want_dx=ctx.want_dx)
        ctx.path, ctx.plan, ctx.x, ctx.context, ctx.gen = path, plan, xc, context, gen
        ctx.params = params
        # bound gradients (TrainPath.bind_flat_grads): the only differentiable input is the path's anchor scalar; the
        # parameters' gradients are added into their flat buffer by backward itself
        ctx.bound = len(params) == 1 and params[0] is path.__dict__.get("_anchor")
        return lp

    @staticmethod
    def backward(ctx, g_lp):
        path: TrainPath = ctx.path
        path.revalidate(ctx)
        if ctx.bound:
            path.backward(ctx.plan, ctx.x, g_lp, into_bound=True, want_dx=ctx.want_dx)
            return (None, path._input_grad(ctx.plan).reshape(ctx.x.shape) if ctx.want_dx else None, None, path._anchor_zero)
        grads = path.backward(ctx.plan, ctx.x, g_lp, want_dx=ctx.want_dx)
        out = tuple(grads.get(id(p)) if p.requires_grad else None for p in ctx.params)
        return (None, path._input_grad(ctx.plan).reshape(ctx.x.shape) if ctx.want_dx else None, None) + out


class _RadiusFn(torch.autograd.Function):
    """Radial bases: the node returns (r = ||f^-1(x) - loc||_p, log-det); ``RadialDistribution.log_prob``'s finishing
    formula norm_dist.log_prob(r) - log dV_p(r) (distributions.py:506-511) is applied outside, in torch, so a trainable
    norm distribution gets its gradients from autograd and this node's backward starts from d/dr."""

    @staticmethod
    def forward(ctx, path: TrainPath, x, context, *params):
        ctx.want_dx = bool(ctx.needs_input_grad[1])
        r, plan, xc, gen = path.forward(x, context, want_dx=ctx.want_dx)
        ctx.path, ctx.plan, ctx.x, ctx.context, ctx.gen = path, plan, xc, context, gen
        ctx.params = params
        # bound gradients (as _LogProbFn): the only differentiable input is the path's anchor; backward adds the whole arena -- the
        # base's loc included -- into the bound .grad views with one launch
        ctx.bound = len(params) == 1 and params[0] is path.__dict__.get("_anchor")
        logdet = plan["pk"]["ladj_total"].neg32(r.device)
        return r, logdet

    @staticmethod
    def backward(ctx, g_r, g_logdet):
        path: TrainPath = ctx.path
        path.revalidate(ctx)
        if g_r is None:
            g_r = torch.zeros(ctx.x.shape[0], dtype=torch.float32, device=ctx.x.device)
        if g_logdet is None:
            g_logdet = torch.zeros((), dtype=torch.float32, device=ctx.x.device)
        if ctx.bound:
            path.backward(ctx.plan, ctx.x, g_r, gsum=g_logdet, into_bound=True, want_dx=ctx.want_dx)
            return (None, path._input_grad(ctx.plan).reshape(ctx.x.shape) if ctx.want_dx else None, None, path._anchor_zero)
        grads = path.backward(ctx.plan, ctx.x, g_r, gsum=g_logdet, want_dx=ctx.want_dx)
        out = tuple(grads.get(id(p)) if p.requires_grad else None for p in ctx.params)
        return (None, path._input_grad(ctx.plan).reshape(ctx.x.shape) if ctx.want_dx else None, None) + out


class _EmptyShardFn(torch.autograd.Function):
    """data-parallel training, this rank's shard is empty: log_prob is an empty tensor, but the backward joins the
    gradient all-reduce (weight 0) so the other ranks do not hang and this replica receives the same gradients"""

    @staticmethod
    def forward(ctx, path: TrainPath, x, *params):
        ctx.path, ctx.params, ctx.dev = path, params, x.device
        return torch.empty(0, dtype=torch.float32, device=x.device)

    @staticmethod
    def backward(ctx, g_lp):
        grads = ctx.path.empty_shard_gradients(ctx.dev)
        return (None, None) + tuple(grads.get(id(p)) if p.requires_grad else None for p in ctx.params)


def log_prob_empty_shard(path: TrainPath, x):
    extra = []
    info = path.flow._base_info(x.device)
    if info is not None and info[0] == "radial":
        extra.append(path.flow.base_distribution.loc)
    return _EmptyShardFn.apply(path, x, *(extra + list(path.params())))


def log_prob_with_grad(path: TrainPath, x, context):
    params = list(path.params())
    info = path.flow._base_info(x.device)
    if info[0] == "radial":
        base = path.flow.base_distribution
        if path.__dict__.get("use_bound_node", False) and path.grads_bound() and torch.is_grad_enabled():
            r, logdet = _RadiusFn.apply(path, x, context, path._anchor)     # (Flow.fit's captured step: see below)
        else:
            r, logdet = _RadiusFn.apply(path, x, context, base.loc, *params)
        lp = None
        if config.radial:
            from . import radial
            lp = radial.log_prob_from_radius(base, r)      # (one launch each way, no validating distribution object)
        return (base.log_prob_from_radius(r) if lp is None else lp) + logdet
    # (the bound node -- no parameter inputs, the whole arena added into the bound .grad views by one launch -- only inside
    # the scope that asked for it: Flow.fit's captured step sets ``use_bound_node``; anywhere else a grad-enabled log_prob
    # is an ordinary autograd node over the parameters, so torch.autograd.grad / hooks / backward(inputs=...) behave as usual
    # even while the .grad views of an earlier fit are still in place)
    if path.__dict__.get("use_bound_node", False) and path.grads_bound() and torch.is_grad_enabled():
        return _LogProbFn.apply(path, x, context, path._anchor)
    return _LogProbFn.apply(path, x, context, *(params + path.base_extra_params()))
