"""SophiaG, the optimiser ``Flow.fit`` defaults to in the reference (flows.py:116; sophia.py:8-199) -- SURVEY row N2.

Same constructor, state (``step`` / ``exp_avg`` / ``hessian`` per parameter), ``update_hessian()`` and
``step(closure=None, bs=5120)`` as the reference class, so optimiser state dicts are interchangeable.  The update

    p *= 1 - lr * weight_decay;  m = beta1 * m + (1 - beta1) * g;
    p -= lr * sign(m) * min(|m| / (rho * bs * h + 1e-15), 1)          (sophia.py:175-199)

is elementwise and HBM-bound.  The reference walks the parameter list with seven ATen ops per tensor; on a ROCm device
all fp32 tensors of a parameter group go through ONE launch of ``usf_sophiag_step_f32`` (a device table of chunks, one
block each; 24 bytes per parameter and step), likewise ``update_hessian`` (``usf_sophiag_hessian_f32``).  CPU tensors
(the mirror's CPU tests) and anything that is not contiguous fp32 take the same arithmetic as torch ops.
"""
from typing import List

import numpy as np
import torch
from torch.optim.optimizer import Optimizer

_CHUNK = 16384          # elements per block of the multi-tensor kernels


class SophiaG(Optimizer):
    def __init__(self, params, lr=1e-4, betas=(0.965, 0.99), rho=0.04, weight_decay=1e-1, *, maximize: bool = False,
                 capturable: bool = False):
        # argument checks and messages of sophia.py:12-21
        if not 0.0 <= lr:
            raise ValueError("Invalid learning rate: {}".format(lr))
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError("Invalid beta parameter at index 0: {}".format(betas[0]))
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError("Invalid beta parameter at index 1: {}".format(betas[1]))
        if not 0.0 <= rho:
            raise ValueError("Invalid rho parameter at index 1: {}".format(rho))
        if not 0.0 <= weight_decay:
            raise ValueError("Invalid weight_decay value: {}".format(weight_decay))
        defaults = dict(lr=lr, betas=betas, rho=rho, weight_decay=weight_decay, maximize=maximize, capturable=capturable)
        super().__init__(params, defaults)
        self._tables = {}           # per group: (key of data pointers, device chunk table, number of chunks)

    def __setstate__(self, state):
        super().__setstate__(state)
        for group in self.param_groups:
            group.setdefault("maximize", False)
            group.setdefault("capturable", False)
        values = list(self.state.values())
        if values and not torch.is_tensor(values[0]["step"]):
            for s in values:
                s["step"] = torch.tensor(float(s["step"]))
        self._tables = {}

    # ---- state (sophia.py:46-55, 86-95) ----
    def _state_of(self, p):
        state = self.state[p]
        if len(state) == 0:
            state["step"] = (torch.zeros((1,), dtype=torch.float, device=p.device) if self.defaults["capturable"]
                             else torch.tensor(0.))
            state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            state["hessian"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        if "hessian" not in state:
            state["hessian"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return state

    @staticmethod
    def _on_hip(p) -> bool:
        return (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad is not None
                and p.grad.dtype == torch.float32 and p.grad.is_contiguous() and not p.grad.is_sparse)

    def _table(self, gi: int, ps: List[torch.Tensor]):
        """device table of usf_mt_chunk for the tensors ``ps`` of group ``gi`` (rebuilt when a pointer moved)"""
        from . import _ext
        key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["hessian"].data_ptr())
                    for p in ps)
        hit = self._tables.get(gi)
        if hit is not None and hit[0] == key:
            return hit[1], hit[2]
        rows = []
        for p in ps:
            st = self.state[p]
            base = (p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["hessian"].data_ptr())
            n = p.numel()
            for off in range(0, n, _CHUNK):
                rows.append((base[0] + 4 * off, base[1] + 4 * off, base[2] + 4 * off, base[3] + 4 * off,
                             min(_CHUNK, n - off), 0))
        dt = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("h", "<u8"), ("n", "<i4"), ("r", "<i4")])
        assert dt.itemsize == _ext.C.sizeof(_ext.MtChunk)
        host = torch.from_numpy(np.array(rows, dtype=dt).view(np.uint8).reshape(-1).copy())
        if getattr(self, "_defer_uploads", False) and torch.cuda.is_current_stream_capturing():
            # Flow.fit captures a step whose gradients are allocated inside the capture (their addresses are known only
            # now): a host-to-device copy is not capturable, and not needed -- nothing runs during a capture.  The table
            # goes into a buffer allocated BEFORE the capture (memory allocated inside one is recycled between the graph's
            # own kernels on every replay: a table uploaded once would be overwritten by whatever shared its block); its
            # contents are uploaded by flush_uploads() before the first replay.
            buf = self._capture_buffers.get(gi)
            if buf is None or buf.numel() < host.numel():
                raise RuntimeError("SophiaG: no pointer-table buffer prepared for this capture (defer_uploads)")
            dev = buf[: host.numel()]
            self._pending_uploads.append((dev, host))
        else:
            dev = host.to(ps[0].device)
        self._tables[gi] = (key, dev, len(rows))
        return dev, len(rows)

    def defer_uploads(self, on: bool) -> None:
        """Flow.fit, around the capture of a training step: table uploads wait for ``flush_uploads``; ``on`` allocates
        one table buffer per group, large enough for all of the group's device parameters"""
        from . import _ext
        self._defer_uploads = bool(on)
        if on:
            self._pending_uploads = []
            self._capture_buffers = {}
            for gi, group in enumerate(self.param_groups):
                ps = [p for p in group["params"] if p.is_cuda and p.dtype == torch.float32]
                rows = sum((p.numel() + _CHUNK - 1) // _CHUNK for p in ps)
                if rows:
                    self._capture_buffers[gi] = torch.empty(rows * _ext.C.sizeof(_ext.MtChunk), dtype=torch.uint8,
                                                            device=ps[0].device)

    def flush_uploads(self) -> None:
        for dev, host in getattr(self, "_pending_uploads", []):
            dev.copy_(host)
        self._pending_uploads = []

    def prepare_tables(self) -> None:
        """(re)build the device chunk tables for the parameters' current gradient buffers now -- Flow.fit calls this
        before it captures a step as a hipGraph: the upload of a table is a host-to-device copy, which a capture refuses"""
        for gi, group in enumerate(self.param_groups):
            if group["capturable"]:
                continue
            hip = [p for p in group["params"]
                   if p.grad is not None and not p.grad.is_sparse and self._on_hip(p) and not torch.is_complex(p)]
            if hip:
                for p in hip:
                    self._state_of(p)
                self._table(gi, hip)

    @torch.no_grad()
    def update_hessian(self):
        """h = beta2 * h + (1 - beta2) * g * g   (sophia.py:39-58)"""
        for gi, group in enumerate(self.param_groups):
            _, beta2 = group["betas"]
            hip = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                state = self._state_of(p)
                if self._on_hip(p) and state["hessian"].is_contiguous():
                    hip.append(p)
                else:
                    state["hessian"].mul_(beta2).addcmul_(p.grad, p.grad, value=1 - beta2)
            if hip:
                from . import _ext
                _ext.load()                                  # no silent fallback on a GPU box
                dev, n = self._table(gi, hip)
                _ext.sophiag_hessian(dev, n, beta2=beta2)

    def note_graph_replays(self, n: int = 1) -> None:
        """A hipGraph replay of a captured training step (Flow.fit) runs the update kernels but no Python: the per-parameter
        ``state['step']`` counters (CPU tensors, bumped by ``step()``) do not move.  Flow.fit reports every replay here so
        that ``state_dict()`` stays interchangeable with the reference's (sophia.py:151-163)."""
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p)
                if st is not None and "step" in st and p.grad is not None:
                    st["step"] += n

    @torch.no_grad()
    def step(self, closure=None, bs=5120):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            beta1, _ = group["betas"]
            lr, rho, wd, maximize = group["lr"], group["rho"], group["weight_decay"], group["maximize"]
            hip = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("Hero does not support sparse gradients")     # (the reference's message)
                state = self._state_of(p)
                state["step"] += 1
                if self._on_hip(p) and not torch.is_complex(p) and not group["capturable"]:
                    hip.append(p)
                    continue
                # the reference's per-tensor arithmetic (sophia.py:164-199)
                grad = p.grad if not maximize else -p.grad
                exp_avg, hess, param = state["exp_avg"], state["hessian"], p
                if torch.is_complex(param):
                    grad, exp_avg, hess, param = (torch.view_as_real(t) for t in (grad, exp_avg, hess, param))
                param.mul_(1 - lr * wd)
                exp_avg.mul_(beta1).add_(grad, alpha=1 - beta1)
                ratio = (exp_avg.abs() / (rho * bs * hess + 1e-15)).clamp(None, 1)
                param.addcmul_(exp_avg.sign(), ratio, value=-lr)
            if hip:
                from . import _ext
                _ext.load()
                dev, n = self._table(gi, hip)
                _ext.sophiag_step(dev, n, decay=1 - lr * wd, beta1=beta1, rho_bs=rho * float(bs), lr=lr, maximize=maximize)
                for p in hip:                                # the kernel wrote through raw pointers: tell autograd / the
                    torch.autograd.graph.increment_version(p)       # engine's parameter-version cache that p changed
        return loss
