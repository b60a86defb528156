"""Data-parallel use of the hot path: one process per GPU, the sample batch sharded row-wise,
parameters replicated, and exactly ONE collective per batch -- an all-reduce (RCCL over xGMI when
the backend is "nccl") of two fp64 scalars [sum log_prob, count] for the mean log-likelihood.

The reference has no distributed code; this is the multi-GPU row of the scope
table (section 8e).  Sampling needs no collective: ranks draw disjoint Philox substreams
(``row_offset``)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_rows(n_rows: int, rank: int, world_size: int) -> Tuple[int, int]:
    """contiguous row range [lo, hi) of ``rank`` (first ``n_rows % world_size`` ranks get one extra)"""
    base, rem = divmod(n_rows, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def mean_log_prob(flow, x_shard: torch.Tensor, context: Optional[torch.Tensor] = None, group=None,
                  acc: Optional[torch.Tensor] = None, force_collective: bool = False):
    """(mean log_prob over ALL ranks' rows as a 0-dim fp64 tensor, this rank's per-sample log_prob).

    No host synchronisation: the sums are accumulated on the device by the tail kernel and the
    all-reduce is enqueued behind it.  ``force_collective``: issue the all-reduce also in a group of ONE rank (it is the
    identity there and skipped by default) -- how the RCCL path is exercised on a single GPU."""
    dev = x_shard.device
    if acc is None:
        acc = torch.zeros(2, dtype=torch.float64, device=dev)
    else:
        acc.zero_()
    with torch.no_grad():
        if x_shard.shape[0] == 0:
            # an empty shard (fewer rows than ranks): nothing to evaluate -- torch's Independent cannot even reshape an empty
            # event batch --, but the rank still joins the collective below with [0, 0]: the others would hang without it
            lp = torch.empty(0, dtype=torch.float32, device=dev)
        elif x_shard.is_cuda and flow._on_device_fast_path(x_shard, context):
            lp = flow._log_prob_device(x_shard, context, sum_out=acc)
        else:
            lp = flow.log_prob(x_shard, context) if context is not None else flow.log_prob(x_shard)
            acc[0] = lp.double().sum()
            acc[1] = float(lp.numel())
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force_collective):
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    return acc[0] / acc[1], lp


def data_parallel_training(flow, group=None, average: bool = True, force_collective: bool = False) -> None:
    """Replicated parameters, batch sharded over the ranks: after this call every backward pass of the device training
    path (training.py) all-reduces the flow's gradients across ``group`` -- ONE collective per step over the flat
    gradient arena (cfg2: 195 MB), so an optimiser step on every rank keeps the replicas identical.
    ``average=True`` (each rank's loss is the MEAN over its own shard, as in ``Flow.fit``): shards are weighted by
    their row counts, sum_r B_r grad_r / sum_r B_r -- the gradient of the global mean also for unequal shards (a short
    last batch); a rank with an EMPTY shard still joins the collective (weight 0) and receives the global gradient.
    ``average=False``: plain sum (losses that are sums).  While enabled, a ``log_prob`` call under autograd that cannot
    take the device training path raises instead of silently skipping the collective (the other ranks would hang).
    (Gradients produced outside the node -- a trainable radial norm distribution -- are not included.)
    ``force_collective``: issue the all-reduce also in a group of one rank (the identity; skipped by default) -- how the RCCL
    path is exercised on a single GPU.
    The reference has no distributed training; this is the data-parallel row of the scope table for ``Flow.fit``."""
    from .training import TrainPath
    if _is_image_flow(flow) or flow.engine() is None:
        # flows that train through per-layer autograd functions (image-shaped inputs: image_training.py) or torch autograd:
        # their gradients land in ``p.grad``; ``Flow.fit`` calls ``allreduce_gradients`` between backward and the optimiser
        # step (eager steps: a collective is not captured into the step's hipGraph), a caller's own loop does the same
        flow.__dict__["_grad_allreduce"] = (group, average)
        return
    if flow._train_obj is None:
        flow._train_obj = TrainPath(flow)
    flow._train_obj.grad_allreduce = (group, average)
    flow._train_obj.force_collective = bool(force_collective)


def _is_image_flow(flow) -> bool:
    dims = getattr(flow, "in_dims", None)
    return dims is not None and len(dims) > 1


def bind_dp_grads(flow):
    """(flat, views): every trainable parameter's ``.grad`` becomes a view of ONE persistent fp32 buffer (its last element
    carries the rank's row count through the collective) -- bound once, so that a data-parallel step is one scale, one
    all-reduce and one divide over the buffer instead of a ``torch.cat`` of every gradient and a scatter back per step.
    Existing gradients are copied in.  Rebinds when the parameter set or a ``.grad`` changed identity (``zero_grad(
    set_to_none=True)``; ``Flow.fit`` zeroes the buffer in place instead: ``Flow._zero_grad_for_step``)."""
    params = [p for p in flow.parameters() if p.requires_grad]
    st = flow.__dict__.get("_dp_grads")
    key = tuple((id(p), p.numel()) for p in params)
    if st is not None and st["key"] == key and all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() and p.grad.shape == v.shape
                                                   for p, v in zip(params, st["views"])):
        return st["flat"], st["views"]
    if not params:
        return None, []
    dev = params[0].device
    if any(p.dtype != torch.float32 or p.device != dev for p in params):
        return None, []
    n = sum(p.numel() for p in params)
    flat = st["flat"] if (st is not None and st["key"] == key and st["flat"].device == dev) else torch.zeros(n + 1, dtype=torch.float32, device=dev)
    views, o = [], 0
    for p in params:
        v = flat[o: o + p.numel()].view(p.shape)
        if p.grad is None:
            v.zero_()
        elif p.grad.data_ptr() != v.data_ptr():
            v.copy_(p.grad)
        p.grad = v
        views.append(v)
        o += p.numel()
    flow.__dict__["_dp_grads"] = dict(key=key, flat=flat, views=views)
    return flat, views


def allreduce_gradients(flow, local_rows: int, group=None, average: bool = True) -> None:
    """ONE all-reduce of every parameter gradient of ``flow`` (one persistent flat buffer the gradients are views of --
    ``bind_dp_grads`` -- with the rank's row count as its last element): ``average=True`` leaves
    sum_r B_r grad_r / sum_r B_r in every ``p.grad`` -- the gradient of the global mean when each rank's loss is the mean over
    its own ``local_rows`` rows (``Flow.fit``); a rank with no rows passes 0 and receives the global gradient.
    ``average=False``: plain sum.  Call between ``loss.backward()`` and ``optimizer.step()``; every rank must call it once
    per step.  No host synchronisation."""
    import torch.distributed as dist
    flat, views = bind_dp_grads(flow)
    if flat is None:
        if not [p for p in flow.parameters() if p.requires_grad]:
            return
        return _allreduce_gradients_cat(flow, local_rows, group, average)
    if average:
        flat[:-1].mul_(float(local_rows))
    flat[-1] = float(local_rows)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat[:-1].div_(flat[-1].clamp_min(1.0))


def _allreduce_gradients_cat(flow, local_rows: int, group=None, average: bool = True) -> None:
    """mixed dtypes / devices: gather, reduce, scatter"""
    import torch.distributed as dist
    params = [p for p in flow.parameters() if p.requires_grad]
    dev, dt = params[0].device, torch.float32
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    w = float(local_rows) if average else 1.0
    flat = torch.cat([p.grad.reshape(-1).to(dt) * w for p in params] + [torch.tensor([float(local_rows)], dtype=dt, device=dev)])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat = flat / flat[-1].clamp_min(1.0)
    o = 0
    for p in params:
        n = p.numel()
        p.grad.copy_(flat[o: o + n].reshape(p.shape).to(p.grad.dtype))
        o += n


def sample_sharded(flow, n_total: int, seed: int, rank: int, world_size: int) -> torch.Tensor:
    """this rank's rows of a global draw of ``n_total`` samples (same result for any world size)"""
    lo, hi = shard_rows(n_total, rank, world_size)
    return flow.sample([hi - lo], seed=seed, row_offset=lo)
