"""Device engine: compiles a list of USFlows layers into a chain of HIP kernel launches.

What it does, per flow (see DESIGN.md for the rationale):

* **parameter pack** (cached per parameter version): for every affine block the dense ``M``,
  ``M^-1`` (triangular solves in fp64 on the device, rounded once to fp32), bias and
  ``sum log|diag U|``; for every coupling the mask-aware slices of the conditioner weights.  The
  reference re-derives all of this on every ``log_prob`` call (transforms.py:1289-1293, 795-809).
* **segment layout**: between layers the activation matrix is stored ``[mask==0 features |
  mask==1 features]`` (each segment padded to 4 floats), so a coupling layer's conditioning half
  and transformed half are *contiguous* column ranges -- no strided gathers, and the ``x*mask`` /
  ``(1-mask)*NN`` zeros of the reference are never multiplied.  The permutation is folded into
  the rows/columns of the neighbouring affine matrices at pack time, i.e. it is free.
* **op list**: fused linear ops (``usf_linear_f32``: scale / bias prologue + GEMM + bias /
  LeakyReLU / residual / scale epilogue), optionally the fused coupling kernel, then the base
  density tail (``usf_base_logprob_f32``).  The whole list is launched by ONE ``usf_run_ops``
  call on torch's current stream.

No arithmetic of the device path is done by torch ops, except the (cached, batch-independent)
parameter prep and the O(B) finishing formula of the radial density.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import nn

from . import _ext
from . import transforms as T
from .config import config
from .networks import ConditionalDenseNN, ConvNet, DenseNN
from .engine_planes import PlanesPlanMixin


class EngineUnsupported(Exception):
    """The layer list contains something the fused device path cannot express."""


def _round_up(n: int, m: int) -> int:
    return (n + m - 1) // m * m


def _activation_of(f) -> Optional[Tuple[int, float]]:
    if isinstance(f, nn.LeakyReLU):
        return _ext.ACT_LEAKY_RELU, float(f.negative_slope)
    if isinstance(f, nn.ReLU):
        return _ext.ACT_LEAKY_RELU, 0.0
    return None


def conditioner_supported(cond: nn.Module) -> bool:
    if isinstance(cond, ConditionalDenseNN):
        return cond.context_dim == 1 and _activation_of(cond.f) is not None
    if isinstance(cond, DenseNN):
        return cond.count_params == 1 and _activation_of(cond.f) is not None
    if isinstance(cond, ConvNet):        # vector path: plain chain of Linears, or GatedMLP / LayerNormVector blocks
        return cond.is_vector and _activation_of(cond.f) is not None     # (spatial path: image-shaped flows, the layer loop)
    return False


@dataclass
class _Step:
    kind: str            # 'scale' | 'affine' | 'coupling'
    module: nn.Module    # ScaleTransform | AffineTransform block | MaskedCoupling
    inverted: bool       # wrapped in InverseTransform


class _MergedAffine:
    """Stand-in block for a run of consecutive affine steps of one direction (inference plans only): with
    ``affine_conjugation`` every coupling is followed by block_i^-1 and block_(i+1), two dense D x D maps with nothing
    in between -- applied as ONE map y = (W2 W1) x + (W2 c1 + c2) composed in fp64 at pack time (half the affine GEMMs
    of such a flow; the reference applies them one by one).  ``pk["affine"][id(self)]`` holds the composite as the M / b
    of an ``affine_fwd`` step."""

    def __init__(self, parts):
        self.parts = parts          # [(prim, block)] in application order


def _analyze(layers: Sequence[nn.Module]) -> List[_Step]:
    steps = []
    for l in layers:
        inv = False
        while isinstance(l, T.InverseTransform):
            inv = not inv
            l = l.transform
        if isinstance(l, T.ScaleTransform) and l.scale.dim() == 1:
            steps.append(_Step("scale", l, inv))
        elif isinstance(l, T.BlockAffineTransform):
            steps.append(_Step("affine", l.block_transform, inv))
        elif isinstance(l, (T.LUTransform, T.HouseholderTransform, T.SequentialAffineTransform)):
            steps.append(_Step("affine", l, inv))
        elif isinstance(l, T.MaskedCoupling) and l.mask.dim() == 2 and conditioner_supported(l.conditioner):
            steps.append(_Step("coupling", l, inv))
        else:
            raise EngineUnsupported(f"layer {type(l).__name__} has no fused device form")
    if not steps:
        raise EngineUnsupported("empty layer list")
    return steps


# ---------------------------------------------------------------------------------------------
# fp64 parameter prep: batched HIP kernels (usf_prep.hip), cached per parameter version
# ---------------------------------------------------------------------------------------------
_copied_params = [0]      # bumped whenever _param had to COPY a parameter (a pack built from copies cannot be replayed)


def _param(t: torch.Tensor, device=None) -> torch.Tensor:
    """fp32 contiguous device view of a parameter (its own storage when it already is one)"""
    src = t
    t = t.detach()
    if device is not None and t.device != torch.device(device):
        t = t.to(device)
    if t.dtype != torch.float32:
        t = t.float()
    t = t if t.is_contiguous() else t.contiguous()
    if t.data_ptr() != src.data_ptr():
        _copied_params[0] += 1
    return t


class LogDet:
    """Sum of the layers' log|det J| (parameter-only for these flows: flows.py:236-245).  ``neg_dev`` is a persistent
    fp64 device scalar holding MINUS the total (what log_prob adds); kernels take its pointer
    (``usf_base_logprob_f32(logdet_dev=...)``), torch expressions the tensor.  ``float(obj)`` is the total as a Python
    float -- the one host read-back, done on demand and cached until the next refresh."""
    __slots__ = ("neg_dev", "_val")

    def __init__(self, neg_dev):
        self.neg_dev, self._val = neg_dev, (0.0 if neg_dev is None else None)

    def refresh(self, total: torch.Tensor) -> None:
        torch.neg(total, out=self.neg_dev)
        self._val = None

    def __float__(self) -> float:
        if self._val is None:
            self._val = -float(self.neg_dev.item())
        return self._val

    def neg32(self, device) -> torch.Tensor:
        """-total as a 0-dim fp32 tensor on ``device`` (for torch-side sums)"""
        if self.neg_dev is None:
            return torch.zeros((), dtype=torch.float32, device=device)
        return self.neg_dev.to(torch.float32)


_PERM_CONST: dict = {}      # FlowEngine._perm_vec: constants of a (layout index, padding) pair


def _refreshed(shape, dtype, device, fn) -> torch.Tensor:
    """persistent tensor filled by fn(out) now and again on every replay of the pack tape"""
    out = torch.empty(shape, dtype=dtype, device=device)
    _ext.host_op(lambda: fn(out))
    return out


def prepare_affine_blocks(blocks: Sequence[nn.Module], device=None, keep_factors: bool = False) -> Dict[int, dict]:
    """``id(block) -> dict(M, Minv, b, ladj)`` (fp64 device tensors) for LUTransform / HouseholderTransform /
    SequentialAffineTransform blocks, following the reference's definitions (transforms.py:1271-1320, 795-809,
    1457-1476).  All LU blocks of the flow go through ONE batched ``usf_lu_prepare_f64`` call (chunked only to
    bound the fp64 scratch at large D).  Every result lives in a tensor of its own that a replay of the recording
    tape (``_ext.Tape``) refills in place: pointers handed to plans stay valid across optimiser steps."""
    lus: Dict[int, nn.Module] = {}
    hhs: Dict[int, nn.Module] = {}

    def visit(b):
        if isinstance(b, T.LUTransform):
            lus.setdefault(id(b), b)
        elif isinstance(b, T.HouseholderTransform):
            hhs.setdefault(id(b), b)
        elif isinstance(b, T.SequentialAffineTransform):
            for t in b.transforms:
                visit(t)
        else:
            raise EngineUnsupported(type(b).__name__)

    for b in blocks:
        visit(b)
    res: Dict[int, dict] = {}
    lu_list = list(lus.values())
    chunks = []
    if lu_list:
        D = int(lu_list[0].dim)
        chunk = max(1, min(len(lu_list), int(8e9 // (64 * D * D))))       # 8 fp64 [D,D] arrays per block in flight
        for c0 in range(0, len(lu_list), chunk):
            part = lu_list[c0: c0 + chunk]
            out = _ext.lu_prepare([_param(l.L_raw, device) for l in part], [_param(l.U_raw, device) for l in part],
                                  keep_factors=keep_factors)
            dev = out["ladj"].device
            bias = [_param(l.bias_vector, device) for l in part]
            b64 = _refreshed((len(part), D), torch.float64, dev,
                             lambda o, bias=bias: o.copy_(torch.stack(bias)))       # (one gather + one converting copy, not one per block)
            chunks.append(dict(lus=part, out=out, b=b64))
            # bias folding, (y - b) Minv^T == y Minv^T + c: c = -(Minv b) of ALL blocks in one batched launch (it was one
            # usf_matvec_f64 per block on the pack's tape: 32 launches of every small-batch training step)
            c64 = torch.empty(len(part), D, dtype=torch.float64, device=dev)
            _ext.gemm_f64(b64, out["Minv"], c64, batch=len(part), M=1, N=D, K=D, lda=D, ldb=D, ldc=D, strideA=D, strideB=D * D,
                          strideC=D, transB=True, alpha=-1.0)
            for j, l in enumerate(part):
                r = dict(M=out["M"][j], Minv=out["Minv"][j], b=b64[j], c=c64[j], ladj=out["ladj"][j])
                if keep_factors:
                    r.update(L=out["tri"][2 * j], Ut=out["tri"][2 * j + 1], Linv=out["tri_inv"][2 * j],
                             Uinv_t=out["tri_inv"][2 * j + 1])
                res[id(l)] = r
    hh_list = list(hhs.values())
    if hh_list:
        D = int(hh_list[0].dim)
        dev = torch.device(device) if device is not None else hh_list[0].w_0.device
        Hs = torch.empty(len(hh_list), D, D, dtype=torch.float64, device=dev)     # one stack: batched composition below
        HsT = torch.empty(len(hh_list), D, D, dtype=torch.float64, device=dev)
        zero = torch.zeros(D, dtype=torch.float64, device=dev)
        for i, h in enumerate(hh_list):
            _ext.householder(_param(h.w_0, device), _param(h.vk_householder, device), out=Hs[i])
        _ext.host_op(lambda: HsT.copy_(Hs.transpose(1, 2)))                       # (after the launches above)
        for i, h in enumerate(hh_list):
            res[id(h)] = dict(M=Hs[i], Minv=HsT[i], b=zero, ladj=zero.sum(), row=i)
    seq = [b for b in blocks if isinstance(b, T.SequentialAffineTransform) and id(b) not in res]
    # the USFlow constructor's block: Sequential([LU, Householder]) for every coupling block.  When the LU factors are
    # rows first .. first+m-1 of one prepare chunk and the Householder factors rows 0 .. m-1 of their stack, the
    # composition (transforms.py:1457-1476) is three batched launches instead of three per block
    batched = False
    if (len(seq) > 1 and len(chunks) == 1 and all(len(b.transforms) == 2 and isinstance(b.transforms[0], T.LUTransform)
                                                  and isinstance(b.transforms[1], T.HouseholderTransform) for b in seq)):
        lu_row = {id(l): j for j, l in enumerate(chunks[0]["lus"])}
        rows = [lu_row[id(b.transforms[0])] for b in seq]
        hrows = [res[id(b.transforms[1])]["row"] for b in seq]
        m = len(seq)
        if rows == list(range(rows[0], rows[0] + m)) and hrows == list(range(m)):
            out, D = chunks[0]["out"], int(seq[0].dim)
            DD, r0 = D * D, rows[0]
            f64 = lambda *shape: torch.empty(*shape, dtype=torch.float64, device=Hs.device)
            Mt, Mit, bt = f64(m, D, D), f64(m, D, D), f64(m, D)
            bat = dict(batch=m, M=D, N=D, K=D, lda=D, ldb=D, ldc=D, strideA=DD, strideB=DD, strideC=DD)
            _ext.gemm_f64(out["M"], Hs, Mt, a_off=r0 * DD, **bat)                          # M_lu H
            _ext.gemm_f64(Hs, out["Minv"], Mit, transA=True, b_off=r0 * DD, **bat)         # H^T M_lu^-1
            _ext.gemm_f64(chunks[0]["b"], Hs, bt, batch=m, M=1, N=D, K=D, lda=D, ldb=D, ldc=D, strideA=D, strideB=DD,
                          strideC=D, a_off=r0 * D)                                         # b_lu H
            ct = f64(m, D)
            _ext.gemm_f64(bt, Mit, ct, batch=m, M=1, N=D, K=D, lda=D, ldb=D, ldc=D, strideA=D, strideB=DD, strideC=D, transB=True,
                          alpha=-1.0)                                                      # c = -(Minv b), as for the LU blocks
            for i, b in enumerate(seq):
                res[id(b)] = dict(M=Mt[i], Minv=Mit[i], b=bt[i], c=ct[i], ladj=out["ladj"][r0 + i])  # (Householder: ladj 0)
            batched = True
    for b in (seq if not batched else []):
        if True:                                                                  # general composition, block by block
            parts = [res[id(t)] for t in b.transforms]
            M, Minv, bias = parts[0]["M"], parts[0]["Minv"], parts[0]["b"]      # eye @ M_1 == M_1 exactly
            ladj = parts[0]["ladj"]
            for p_ in parts[1:]:                                                  # :1457-1462, :1471-1476
                M = _ext.matmul_f64(M, p_["M"])
                bm = _ext.matmul_f64(bias.reshape(1, -1), p_["M"])
                bias = _refreshed(tuple(p_["b"].shape), torch.float64, bm.device,
                                  lambda o, bm=bm, pb=p_["b"]: torch.add(bm.reshape(-1), pb, out=o))
                Minv = _ext.matmul_f64(p_["Minv"], Minv)                          # :1464-1469 (reverse order)
                ladj = _refreshed((), torch.float64, bm.device,
                                  lambda o, a=ladj, c=p_["ladj"]: torch.add(a, c, out=o))
            res[id(b)] = dict(M=M, Minv=Minv, b=bias, ladj=ladj)
            if len(parts) == 1 and "c" in parts[0]:
                res[id(b)]["c"] = parts[0]["c"]          # Sequential([LU]): the block IS its LU factor, folded bias included
    res["__chunks__"] = chunks
    return res


class FlowEngine(PlanesPlanMixin):
    """Compiled device form of ``layers`` (a whole ``Flow`` or a single layer)."""

    def __init__(self, layers: Sequence[nn.Module]):
        self.steps = _analyze(layers)
        self.D = self._infer_dim()
        self._pack = None
        self._pack_key = None
        self._plans: Dict[tuple, dict] = {}
        self._ws: Dict[tuple, Dict[str, torch.Tensor]] = {}
        # Consecutive affine steps of an inference plan run as ONE composed map (_MergedAffine) -- with affine_conjugation
        # (every live configuration of the reference) that halves the D x D GEMMs.  The reference applies the maps one
        # by one in fp32, and on badly conditioned (default-initialised) blocks the composite rounds differently -- golden
        # init_d7_k2_hh1_conj_laplace: relative log_prob error 4e-6 one by one, 1.0e-5 composed (well-conditioned
        # flows: unchanged, 5e-8 .. 1.5e-7).  merge_affine = "auto" (default): every run of layers is composed only if a
        # pack-time probe says the composite is as accurate as the run applied layer by layer (_merge_guard); True
        # (USFLOWS_AMD_MERGE_AFFINE=1): always; False (=0): never.
        self.merge_affine = config.merge_affine
        self.merge_guard_log: List[tuple] = []   # (accepted, element-wise deviation, L1-norm deviation) of composed vs layer-by-layer, per probe
        self._virtual: List[_Step] = []          # merged steps, addressed as step index len(self.steps) + n
        self._virtual_ix: Dict[tuple, int] = {}
        self.launch_count = 0          # number of usf_run_ops calls (tests assert the HIP path ran)
        self.op_timing = None          # set to a list to record (tag, start_event, end_event) per op (bench.py)
        self.use_fused_coupling = True
        self.fused_min_rows = 14336    # measured cross-over on MI355X at D=784, hidden 256 (bench.py --fused-min-rows sweep:
        #                                fused / unfused ms per step: 8192 4.79 / 3.98, 12288 5.06 / 5.05, 16384 6.45 / 6.66, 20480 7.54 / 8.43)
        # "bf16x3": the GEMMs run on the bf16 matrix cores with a 3-way residual split of both operands (24 significant
        # bits, six MFMAs per product); "f16x2": the planes pipeline uses two fp16 planes per operand
        # (22 significant bits, three MFMAs per product, range-guarded with a bf16x3 redo; everything outside the planes
        # pipeline as "bf16x3"); "f32": exact-f32 MFMA everywhere (USFLOWS_AMD_GEMM=... or engine.gemm_mode = ...)
        self.gemm_mode = config.gemm_mode
        # True: the pack also keeps L, U^T, L^-1, U^-T of every LU block (fp64) -- the training backward's operands
        self.keep_factors = False
        # USFLOWS_AMD_GRAPH=1: batches up to graph_max_rows replay their launch list as one hipGraph.  Off by default:
        # measured on MI355X / ROCm 7.2 the replay is no faster than the C-side launch loop of usf_run_ops (cfg2,
        # B = 100: 1.22 ms either way) -- the ~7 us between two dependent dispatches is not host time
        # planes pipeline: from planes_min_rows rows, inference plans in bf16x3 mode keep the activations
        # between layers as pre-split bf16 planes in MFMA-operand order (usf_planes.hip); USFLOWS_AMD_PLANES=0 disables
        self._f16_overflow = False      # set while a pass is being redone in bf16x3 because fp16 planes overflowed
        self.f16_fallbacks = 0          # number of such passes (tests / diagnostics)
        # use_planes: None = automatic, True / False = forced (USFLOWS_AMD_PLANES=1 / 0).  Automatic: "f16x2" mode from
        # planes_min_rows rows; "bf16x3" mode from planes_min_rows_bf16x3 rows (measured on MI355X, cfg2, ms per step
        # planes / fp32-activation kernels, round 3, same box: 16384 6.77 / 6.55-6.78, 24576 7.48 / 7.75, 32768 9.71 / 9.91
        # -- the per-rank shard of the 8-GPU configuration cfg3 --, 40960 14.05 / 14.30, 65536 18.8 / 19.2),
        # or from planes_min_rows when a conditioner is too wide / deep for the fused coupling kernels (cfg4, hidden 1024:
        # 176.6 vs 215.8 ms at 32768 rows)
        self.use_planes = config.planes
        self.planes_min_rows = int(config.planes_min_rows)
        self.planes_min_rows_bf16x3 = 24576
        # training on the planes pipeline (round 5): the forward keeps every layer's planes buffer and the conditioners' hidden
        # activations as planes, the backward runs on usf_gemm_planes_bf16x3 / usf_coupling_planes (gate mode) /
        # usf_wgrad_blocked_f32.  USFLOWS_AMD_TRAIN_PLANES=0 keeps the fp32-row path of rounds 3 / 4.
        self.use_train_planes = config.train_planes
        self.train_planes_min_rows = 16384
        self.use_graphs = bool(config.engine_graph)
        self.graph_max_rows = 1024
        self._layout_from_masks()

    # ---- static structure ---------------------------------------------------------------------
    def _infer_dim(self) -> int:
        for s in self.steps:
            if s.kind == "scale":
                return int(s.module.scale.shape[0])
            if s.kind == "affine":
                return int(s.module.dim)
            if s.kind == "coupling":
                return int(s.module.mask.shape[-1])
        raise EngineUnsupported("cannot infer dimension")

    def _layout_from_masks(self):
        D = self.D
        base = None
        self._flip = {}
        for i, s in enumerate(self.steps):
            if s.kind != "coupling":
                continue
            m = s.module.mask.detach().flatten().cpu()
            if m.numel() != D or not bool(((m == 0) | (m == 1)).all()):
                raise EngineUnsupported("coupling mask must be a 0/1 vector of length D")
            if base is None:
                base = m
            if torch.equal(m, base):
                self._flip[i] = False
            elif torch.equal(m, 1 - base):
                self._flip[i] = True
            else:
                raise EngineUnsupported("coupling masks must alternate between m and 1-m")
        if base is None:
            idx0 = torch.arange(D)
            idx1 = torch.zeros(0, dtype=torch.long)
        else:
            idx0 = torch.nonzero(base == 0).flatten()
            idx1 = torch.nonzero(base == 1).flatten()
        self.n0, self.n1 = int(idx0.numel()), int(idx1.numel())
        self.n0a, self.n1a = _round_up(self.n0, 4), _round_up(self.n1, 4)
        self.LD = max(self.n0a + self.n1a, 4)
        seg = torch.full((self.LD,), -1, dtype=torch.long)
        seg[: self.n0] = idx0
        seg[self.n0a: self.n0a + self.n1] = idx1
        self.seg_idx = seg
        self.LDn = _round_up(D, 4)
        nat = torch.full((self.LDn,), -1, dtype=torch.long)
        nat[:D] = torch.arange(D)
        self.nat_idx = nat
        # the same two layouts padded to whole 32-feature blocks (planes pipeline, usf_planes.hip)
        self.LDp, self.LDnp = _round_up(self.LD, 32), _round_up(D, 32)
        self.segp_idx = torch.full((self.LDp,), -1, dtype=torch.long)
        self.segp_idx[: self.LD] = seg
        self.natp_idx = torch.full((self.LDnp,), -1, dtype=torch.long)
        self.natp_idx[:D] = torch.arange(D)
        self.hmax = 4
        self._general_cond = False       # a conditioner with gate / layer-norm blocks: chain of ops, no fused kernel
        for s in self.steps:
            if s.kind == "coupling":
                cond = s.module.conditioner
                widths = cond.c_hidden if isinstance(cond, ConvNet) else cond.hidden_dims
                self.hmax = max(self.hmax, max(_round_up(int(h), 4) for h in widths))
                self._general_cond = self._general_cond or (isinstance(cond, ConvNet) and not cond.is_plain_mlp())

    def _params(self):
        seen, out = set(), []
        for s in self.steps:
            for p in s.module.parameters():
                if id(p) not in seen:
                    seen.add(id(p))
                    out.append(p)
        return out

    # ---- parameter pack -----------------------------------------------------------------------
    def _version_key(self, device):
        return (str(device),) + tuple((p.data_ptr(), p._version) for p in self._params())

    def refresh(self):
        """Drop the cached parameter pack (it is also rebuilt automatically when a parameter's
        version counter or storage changes)."""
        self._pack, self._pack_key = None, None
        self._plans.clear()

    def _idx(self, layout: str) -> torch.Tensor:
        return {"seg": self.seg_idx, "nat": self.nat_idx, "segp": self.segp_idx, "natp": self.natp_idx}[layout]

    @staticmethod
    def _perm_mat(mat64: torch.Tensor, out_idx: torch.Tensor, in_idx: torch.Tensor) -> torch.Tensor:
        dev = mat64.device
        oi, ii = out_idx.to(dev), in_idx.to(dev)
        W = torch.zeros(oi.numel(), ii.numel(), dtype=torch.float64, device=dev)
        ro, ci = torch.nonzero(oi >= 0).flatten(), torch.nonzero(ii >= 0).flatten()
        W[ro[:, None], ci[None, :]] = mat64[oi[ro][:, None], ii[ci][None, :]]
        return W.float().contiguous()

    @staticmethod
    def _perm_vec(v64: torch.Tensor, idx: torch.Tensor, pad: float) -> torch.Tensor:
        # (no boolean-mask indexing, no host index tensor: both would synchronise the host on every pack refresh,
        # i.e. on every optimiser step.  What depends on the layout alone -- the gather index, the keep mask, the padding --
        # is built once per index tensor: a refresh is a gather, one fused multiply-add and the rounding, not eight launches)
        idx = idx.to(v64.device)
        key = (idx.data_ptr(), float(pad), str(v64.device))
        ent = _PERM_CONST.get(key)
        if ent is None or ent[0] is not idx:
            valid = idx >= 0
            ent = _PERM_CONST[key] = (idx, idx.clamp(min=0).long(), valid.to(torch.float64),
                                      torch.where(valid, 0.0, float(pad)).to(torch.float64))
            if len(_PERM_CONST) > 256:
                _PERM_CONST.clear()
        _, gidx, keep, padv = ent
        return torch.addcmul(padv, v64[gidx], keep).float()

    def _ptr_key(self, device):
        return (str(device), self.keep_factors) + tuple(p.data_ptr() for p in self._params())

    def pack(self, device) -> dict:
        key = self._version_key(device)
        pk = self._pack
        if pk is not None and key == self._pack_key and (pk["has_factors"] or not self.keep_factors):
            return pk
        if (pk is not None and _ext.TAPES_ENABLED and pk["replayable"] and torch.device(device).type == "cuda"
                and pk["ptr_key"] == self._ptr_key(device) and pk["tape"].stream == _ext.current_stream(device)):
            # same parameter storage, new values (an optimiser step): refill every prepared tensor in place by
            # replaying the recorded launches -- plans and their device pointers stay valid
            with torch.no_grad():
                _ext.replay(pk["tape"])
                pk["ladj_total"] = self._ladj_total(pk)
            self._pack_key = key
            pk["replays"] = pk.get("replays", 0) + 1
            if pk["replays"] % 256 == 0:
                pk.pop("auto_merge", None)          # training moves the parameters: probe the composed plans again
            return pk
        pk = {"affine": {}, "coupling": {}, "scale": {}, "mats": {}, "vecs": {}, "has_factors": self.keep_factors,
              "tape": _ext.Tape(), "ptr_key": self._ptr_key(device), "ladj_terms": []}
        pk["tape"].stream = _ext.current_stream(device) if torch.device(device).type == "cuda" else None
        copies0 = _copied_params[0]
        convnet = False
        with torch.no_grad(), _ext.record(pk["tape"]):
            blocks, seen = [], set()
            for s in self.steps:
                if s.kind == "affine" and id(s.module) not in seen:
                    seen.add(id(s.module))
                    blocks.append(s.module)
            prepared = prepare_affine_blocks(blocks, device, keep_factors=self.keep_factors)
            pk["affine_parts"] = prepared          # also the LU / Householder factors inside Sequential blocks
            for b_ in blocks:
                pk["affine"][id(b_)] = prepared[id(b_)]
            for i, s in enumerate(self.steps):
                if s.kind == "affine":
                    pk["ladj_terms"].append((-1.0 if s.inverted else 1.0, pk["affine"][id(s.module)]["ladj"]))
                elif s.kind == "scale":
                    src = _param(s.module.scale, device)
                    sc = _refreshed(tuple(src.shape), torch.float64, src.device, lambda o, src=src: o.copy_(src))
                    pk["scale"][id(s.module)] = sc
                    la = _refreshed((), torch.float64, src.device,
                                    lambda o, sc=sc: torch.sum(sc.abs().log(), dim=0, out=o))
                    pk["ladj_terms"].append((-1.0 if s.inverted else 1.0, la))
                elif s.kind == "coupling":
                    pk["coupling"][i] = self._pack_coupling(i, s.module, device)
                    # (a PLAIN-MLP ConvNet folds its last two Linears on the torch side: the pack cannot be replayed; the general
                    # form -- GatedMLP / LayerNormVector blocks -- packs from the parameters' own storage like a DenseNN)
                    cond_ = s.module.conditioner
                    convnet = convnet or (isinstance(cond_, ConvNet) and (not cond_.is_vector or cond_.is_plain_mlp()))
            pk["ladj_total"] = self._ladj_total(pk)
        # replay needs every source to be the parameter's own storage (no staging copies) and no torch-side folding
        pk["replayable"] = (_copied_params[0] == copies0) and not convnet
        self._pack, self._pack_key = pk, key
        self._plans.clear()
        return pk

    @staticmethod
    def _ladj_total(pk) -> "LogDet":
        """the flow's parameter-only log|det J| as a LogDet: refreshed in place on the device after every parameter
        change, read back to the host only when somebody asks for the Python float"""
        ld = pk.get("ladj_total")
        terms = pk["ladj_terms"]
        if not terms:
            return ld if ld is not None else LogDet(None)
        if "ladj_signs" not in pk:
            pk["ladj_signs"] = torch.tensor([t[0] for t in terms], dtype=torch.float64).to(terms[0][1].device)
        total = (torch.stack([t[1].reshape(()) for t in terms]) * pk["ladj_signs"]).sum()
        if ld is None:
            ld = LogDet(torch.empty((), dtype=torch.float64, device=total.device))
        ld.refresh(total)
        return ld

    def _pk_record(self, pk):
        """context: launches of a lazily built pack item join the pack's tape (refreshed on replay)"""
        return _ext.record(pk["tape"] if pk.get("replayable") else None)

    def _pack_coupling(self, i: int, layer, device) -> dict:
        """Static description of one coupling layer + handles to its raw conditioner parameters; the weight images
        themselves (unfused: ``_unfused_pack``, fused: ``_fused_pack`` / ``_fused_split``) are built on first use
        by ``usf_pack_weight_f32`` launches (mask-aware column / row selection, zero padding, bf16x3 planes)."""
        cond = layer.conditioner
        flip = self._flip[i]
        # mask==1 features condition, mask==0 features are transformed (transforms.py:285-290)
        if not flip:      # layer mask == base mask: pass = segment 1, transformed = segment 0
            pass_off, pass_n, pass_idx = self.n0a, self.n1a, self.seg_idx[self.n0a: self.n0a + self.n1a]
            tr_off, tr_n, tr_idx = 0, self.n0, self.seg_idx[: self.n0]
        else:
            pass_off, pass_n, pass_idx = 0, self.n0a, self.seg_idx[: self.n0a]
            tr_off, tr_n, tr_idx = self.n0a, self.n1, self.seg_idx[self.n0a: self.n0a + self.n1]
        act, slope = _activation_of(cond.f)
        has_ctx = isinstance(cond, ConditionalDenseNN)
        ctx_l = None
        if isinstance(cond, ConvNet) and not cond.is_plain_mlp():
            # Linear, [GatedMLP | (f, Linear)] [+ LayerNormVector] x n, Linear (networks.py:287-308)
            first, blocks, final = cond.block_view()
            P = lambda l: (_param(l.weight, device), _param(l.bias, device))     # noqa: E731
            raw = dict(first=P(first), last=P(final), device=device, ctx=None, h=[int(first.out_features)],
                       pass_idx=pass_idx.to(torch.int32), tr_idx=tr_idx.to(torch.int32),
                       blocks=[dict(w_in=int(b["w_in"]), w_out=int(b["w_out"]), eps=(float(b["ln"].eps) if b["ln"] is not None else 0.0),
                                    **{k_: P(b[k_]) for k_ in ("lin", "l1", "l2", "proj", "ln") if b.get(k_) is not None})
                               for b in blocks])
            return dict(pass_off=pass_off, pass_n=pass_n, tr_off=tr_off, tr_n=tr_n, act=act, slope=slope, general=True,
                        hidden=[_round_up(int(w), 4) for w in [first.out_features] + list(cond.c_hidden)], has_ctx=False, raw=raw)
        if isinstance(cond, ConvNet):
            # Linear, [f, Linear] x n, Linear (networks.py:287-308): the last block's Linear and the final Linear
            # have no activation between them -> one output map, folded in fp64
            first, hidden, (W_last, b_last), widths = cond.mlp_view()
            W_last, b_last = W_last.to(device).contiguous(), b_last.to(device).contiguous()
        else:
            lin = list(cond.layers)
            first = lin[0]
            ctx_l = lin[1] if has_ctx else None
            hidden = lin[2:-1] if has_ctx else lin[1:-1]
            W_last, b_last = _param(lin[-1].weight, device), _param(lin[-1].bias, device)
            widths = cond.hidden_dims
        h = [int(x) for x in widths]
        hp = [_round_up(x, 4) for x in h]
        raw = dict(first=(_param(first.weight, device), _param(first.bias, device)),
                   hidden=[(_param(l.weight, device), _param(l.bias, device)) for l in hidden],
                   last=(W_last, b_last), h=h, device=device,
                   pass_idx=pass_idx.to(torch.int32), tr_idx=tr_idx.to(torch.int32),
                   ctx=(_param(ctx_l.weight, device), _param(ctx_l.bias, device)) if has_ctx else None)
        return dict(pass_off=pass_off, pass_n=pass_n, tr_off=tr_off, tr_n=tr_n, act=act, slope=slope,
                    hidden=hp, has_ctx=has_ctx, raw=raw)

    def _sel(self, idx: torch.Tensor, n_total: int, device) -> torch.Tensor:
        """int32 device selector: idx (feature numbers, -1 = padding) extended with -1 to n_total"""
        t = torch.full((n_total,), -1, dtype=torch.int32)
        t[: idx.numel()] = idx
        return t.to(device)

    def _packed(self, src, out_sel, n_out, in_sel, n_in, planes_sel=None, planes_ld=0, want_w=True, transpose=False):
        """(W [n_out, n_in] fp32 | None, planes [3, n_out, planes_ld] bf16 | None) from a raw parameter; transpose: the
        image of src^T (out_sel picks columns of src, in_sel rows)"""
        dev = src.device
        W = torch.empty(n_out, n_in, dtype=torch.float32, device=dev) if want_w else None
        kw = dict(transpose=True) if transpose else {}
        if W is not None:
            _ext.pack_weight(src, out_sel, n_out, in_sel, n_in, W=W, ldw=n_in, **kw)
        planes = None
        if planes_ld:
            planes = torch.empty(3, n_out, planes_ld, dtype=torch.bfloat16, device=dev)
            _ext.pack_weight(src, out_sel, n_out, in_sel if planes_sel is None else planes_sel,
                             min(n_in, planes_ld) if planes_sel is None else int(planes_sel.numel()), planes=planes, **kw)
        return W, planes

    def _packed_vec(self, src, sel, n) -> torch.Tensor:
        out = torch.empty(n, dtype=torch.float32, device=src.device)
        _ext.pack_weight(src.reshape(1, -1), None, 1, sel, n, W=out, ldw=n, ld_src=src.numel())
        return out

    def _unfused_pack(self, pk, cp) -> dict:
        """per-layer weight images for the chain-of-linears form of the conditioner (any width / depth)"""
        if "unfused" in cp:
            return cp["unfused"]
        with self._pk_record(pk):
            return self._unfused_pack_build(cp)

    def _unfused_pack_build(self, cp) -> dict:
        raw = cp["raw"]
        dev, h, hp = raw["device"], raw["h"], cp["hidden"]
        pass_sel = self._sel(raw["pass_idx"], cp["pass_n"], dev)
        layers = []
        mats = self._pack["mats"]

        def image(src, out_sel, n_out, in_sel, n_in):
            """fp32 image + (when the linear kernel will take the bf16x3 path for it) its planes, both from `src`"""
            pl = _round_up(n_in, 32) if self._wants_planes(n_out, n_in) else 0
            Wp, P = self._packed(src, out_sel, n_out, in_sel, n_in, planes_ld=pl)
            if P is not None:
                mats[("planes", Wp.data_ptr())] = P
            mats[("imgsrc", Wp.data_ptr())] = (src, out_sel, n_out, in_sel, n_in)      # for the transposed image
            return Wp

        if cp.get("general"):
            return self._general_pack_build(cp, image, pass_sel)
        W, b = raw["first"]
        Wp = image(W, self._iarange(h[0], hp[0], dev), hp[0], pass_sel, cp["pass_n"])
        layers.append((Wp, self._packed_vec(b, self._iarange(h[0], hp[0], dev), hp[0])))
        for j, (W, b) in enumerate(raw["hidden"]):
            Wp = image(W, self._iarange(h[j + 1], hp[j + 1], dev), hp[j + 1], self._iarange(h[j], hp[j], dev), hp[j])
            layers.append((Wp, self._packed_vec(b, self._iarange(h[j + 1], hp[j + 1], dev), hp[j + 1])))
        W, b = raw["last"]
        tr_sel = self._sel(raw["tr_idx"], cp["tr_n"], dev)
        W_out = image(W, tr_sel, cp["tr_n"], self._iarange(h[-1], hp[-1], dev), hp[-1])
        u = dict(layers=layers, W_out=W_out, b_out=self._packed_vec(b, tr_sel, cp["tr_n"]))
        if cp["has_ctx"]:
            Wc, bc = raw["ctx"]
            rows = self._iarange(h[0], hp[0], dev)
            # [h0, 4]: context rides in column 0 of a 4-wide K
            u["W_ctx4"], _ = self._packed(Wc, rows, hp[0], self._iarange(1, 4, dev), 4)
            u["b_ctx"] = self._packed_vec(bc, rows, hp[0])
        cp["unfused"] = u
        return u

    def _general_pack_build(self, cp, image, pass_sel) -> dict:
        """weight images of a vector ConvNet conditioner with GatedMLP / LayerNormVector blocks: every Linear padded to
        multiples of 4 with zeros; a GatedMLP's second Linear keeps its value rows in [0, wp) and its gate rows in
        [wp, 2 wp) (wp = padded block width), so ``chunk(2, dim=1)`` is two column offsets"""
        raw = cp["raw"]
        dev = raw["device"]
        r4 = lambda n: _round_up(n, 4)      # noqa: E731
        ar = lambda n: self._iarange(n, r4(n), dev)      # noqa: E731
        tensors = []

        def lin(Wb, out_sel, n_out, in_sel, n_in):
            W, b = Wb
            Wp = image(W, out_sel, n_out, in_sel, n_in)
            bp = self._packed_vec(b, out_sel, n_out)
            tensors.extend([Wp, bp])
            return Wp, bp

        h0 = raw["h"][0]
        u = dict(general=True, first=lin(raw["first"], ar(h0), r4(h0), pass_sel, cp["pass_n"]), blocks=[])
        for b in raw["blocks"]:
            wi, wo = b["w_in"], b["w_out"]
            e = dict(w_in=wi, w_out=wo, eps=b["eps"])
            if "lin" in b:
                e["lin"] = lin(b["lin"], ar(wo), r4(wo), ar(wi), r4(wi))
            else:
                e["l1"] = lin(b["l1"], ar(wo), r4(wo), ar(wi), r4(wi))
                two = torch.full((2 * r4(wo),), -1, dtype=torch.int32)
                two[:wo] = torch.arange(wo, dtype=torch.int32)
                two[r4(wo): r4(wo) + wo] = torch.arange(wo, 2 * wo, dtype=torch.int32)
                e["l2"] = lin(b["l2"], two.to(dev), 2 * r4(wo), ar(wo), r4(wo))
                if "proj" in b:
                    e["proj"] = lin(b["proj"], ar(wo), r4(wo), ar(wi), r4(wi))
            if "ln" in b:
                g_, b_ = b["ln"]
                e["ln"] = (self._packed_vec(g_, ar(wo), r4(wo)), self._packed_vec(b_, ar(wo), r4(wo)))
                tensors.extend(e["ln"])
            u["blocks"].append(e)
        w_last = raw["blocks"][-1]["w_out"]
        tr_sel = self._sel(raw["tr_idx"], cp["tr_n"], dev)
        u["W_out"], u["b_out"] = lin(raw["last"], tr_sel, cp["tr_n"], ar(w_last), r4(w_last))
        u["tensors"] = tensors
        cp["unfused"] = u
        return u

    def _idx_dev(self, layout: str, device) -> torch.Tensor:
        """int32 device copy of the feature index of every column of a layout (-1 = padding column)"""
        cache = self.__dict__.setdefault("_idx_cache", {})
        key = (layout, str(device))
        if key not in cache:
            cache[key] = self._idx(layout).to(device=device, dtype=torch.int32)
        return cache[key]

    def _iarange(self, n_valid: int, n_total: int, device) -> torch.Tensor:
        """int32 [0 .. n_valid-1, -1, -1, ...] of length n_total (row/column selector with zero padding)"""
        cache = self.__dict__.setdefault("_idx_cache", {})
        key = ("arange", n_valid, n_total, str(device))
        if key not in cache:
            t = torch.full((n_total,), -1, dtype=torch.int32)
            t[:n_valid] = torch.arange(n_valid, dtype=torch.int32)
            cache[key] = t.to(device)
        return cache[key]

    def _wants_planes(self, n_out: int, K: int) -> bool:
        return self.gemm_mode in ("bf16x3", "f16x2") and K % 8 == 0 and n_out > 64

    def _mat(self, pk, blk, which: str, out_layout: str, in_layout: str) -> torch.Tensor:
        """permuted / padded fp32 image of an affine block's M or M^-1 (+ its bf16x3 planes), one launch"""
        key = (id(blk), which, out_layout, in_layout)
        if key not in pk["mats"]:
            src = pk["affine"][id(blk)][which]
            dev = src.device
            oi, ii = self._idx_dev(out_layout, dev), self._idx_dev(in_layout, dev)
            n_out, n_in = int(oi.numel()), int(ii.numel())
            W = torch.empty(n_out, n_in, dtype=torch.float32, device=dev)
            planes = None
            if self._wants_planes(n_out, n_in):
                planes = torch.empty(3, n_out, _round_up(n_in, 32), dtype=torch.bfloat16, device=dev)
            with self._pk_record(pk):
                _ext.pack_weight(src, oi, n_out, ii, n_in, W=W, ldw=n_in, planes=planes)
            pk["mats"][key] = W
            if planes is not None:
                pk["mats"][("planes", W.data_ptr())] = planes
        return pk["mats"][key]

    def _split_kw(self, pk, W: torch.Tensor, K: int) -> dict:
        """descriptor fields that hand the bf16x3 planes of W to usf_linear_f32 (empty in f32 mode)"""
        if not self._wants_planes(W.shape[0], K):
            return {}
        planes = self._split_planes(pk, W)
        return dict(W_split=planes.data_ptr(), ldw_split=planes.shape[2],
                    split_plane_stride=planes.shape[1] * planes.shape[2])

    @staticmethod
    def _split_planes(pk, W: torch.Tensor) -> torch.Tensor:
        """[3, N, ceil32(K)] bf16 planes with W1 + W2 + W3 == W (round-to-nearest residual split)"""
        key = ("planes", W.data_ptr())
        if key not in pk["mats"]:
            N, K = W.shape
            planes = torch.empty(3, N, _round_up(K, 32), dtype=torch.bfloat16, device=W.device)
            with _ext.record(pk["tape"] if pk.get("replayable") else None):
                _ext.flush_jobs()              # W itself may be the output of a queued job
                _ext.pack_weight(W, None, N, None, K, planes=planes)
            pk["mats"][key] = planes
        return pk["mats"][key]

    def _vec(self, pk, name, v64, layout: str, pad: float) -> torch.Tensor:
        key = (name, layout, pad)
        if key not in pk["vecs"]:
            with self._pk_record(pk):
                if pad == 0.0:
                    idx = self._idx_dev(layout, v64.device)
                    out = torch.empty(idx.numel(), dtype=torch.float32, device=v64.device)
                    _ext.pack_weight(v64, None, 1, idx, int(idx.numel()), W=out, ldw=int(idx.numel()))
                    pk["vecs"][key] = out
                else:
                    n = int(self._idx(layout).numel())
                    pk["vecs"][key] = _refreshed((n,), torch.float32, v64.device,
                                                 lambda o: o.copy_(self._perm_vec(v64, self._idx_dev(layout, v64.device), pad)))
        return pk["vecs"][key]

    # ---- workspace ----------------------------------------------------------------------------
    def _workspace(self, B: int, device) -> Dict[str, torch.Tensor]:
        key = (B, str(device))
        ws = self._ws.get(key)
        if ws is None:
            if len(self._ws) > 4:
                self._ws.clear()
                self._plans.clear()
            z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=device)
            ws = dict(zA=z(B, self.LD), zB=z(B, self.LD), nat=z(B, self.LDn), H1=z(B, self.hmax),
                      H2=z(B, self.hmax), P=z(B, self.hmax), ctx4=z(B, 4), ctx=z(B),
                      sum=torch.zeros(2, dtype=torch.float64, device=device))
            self._ws[key] = ws
        return ws

    # ---- plan construction --------------------------------------------------------------------
    def _step(self, i: int) -> _Step:
        return self.steps[i] if i < len(self.steps) else self._virtual[i - len(self.steps)]

    def _merge_on(self, direction: str, device=None) -> bool:
        """are runs of consecutive affine maps composed in this direction's inference plans right now?"""
        if self.merge_affine != "auto":
            return bool(self.merge_affine)
        pk = self._pack
        return bool(pk is not None and pk.get("auto_merge", {}).get(direction, False))

    def resolve_merge(self, direction: str, x: torch.Tensor, runner=None) -> bool:
        """merge_affine == "auto": decide, once per pack and direction, whether this flow's runs of consecutive affine
        maps may be composed.  The probe is end to end -- up to 64 rows of the caller's own batch through the plan with
        composed runs and through the layer-by-layer plan: the composite itself is as accurate as its parts (both are
        rounded once from fp64 factors), what differs between flows is how far ANY change of rounding pattern is
        amplified downstream.  Where the two results agree (1e-5 of the largest output element-wise, 1e-6 in the rows'
        L1 norms -- half resp. a tenth of what the parity bars allow; measured: the reference's default-initialised
        golden flows 4e-6 .. 1.4e-5 in the L1 measure, conditioned flows up to the 65-layer cfg2 model 1e-7 .. 4e-7)
        the flow is well conditioned
        and the composed plan (half the D x D GEMMs of a conjugated flow) is indistinguishable from the reference's
        arithmetic at the 1e-5 parity bar; where they do not (default-initialised, exploding flows: the reference's
        own fp32 run is 2 - 7e-6 from its fp64 run there) the layers keep their own launches.  Costs two 64-row passes
        and one scalar read-back at the first inference call after a (re)pack; re-probed every 256 in-place refreshes.
        runner(plan, x, out): how a plan is executed (tests interpret the op list on the CPU)."""
        if self.merge_affine != "auto":
            return bool(self.merge_affine)
        pk = self.pack(x.device)
        state = pk.setdefault("auto_merge", {})
        if direction in state:
            return state[direction]
        if x.is_cuda and torch.cuda.is_current_stream_capturing():
            return False          # (the probe reads a scalar back: never inside a caller's stream capture -- this call keeps
            #                        the layer-by-layer plan, the next call outside a capture decides)
        has_run = any(a.startswith("affine") and b_.startswith("affine")
                      for (a, _), (b_, _) in zip(self._primitive_ops(direction)[:-1], self._primitive_ops(direction)[1:]))
        if not has_run or x.shape[0] == 0:
            state[direction] = False
            return False
        run = runner if runner is not None else (lambda plan, xs, out: self._run(plan, xs, out, None))
        n = min(64, x.shape[0])
        xs = x[:n].contiguous()
        outs = []
        with torch.no_grad():
            for mode in (False, True):
                state[direction] = mode
                out = torch.empty(n, self.D, dtype=torch.float32, device=x.device)
                run(self._plan(direction, n, x.device, False, "user"), xs, out)
                outs.append(out.double())
            # two measures: the largest element-wise deviation against the largest output (the parity tests allow 2e-5 of
            # it), and the deviation of the rows' L1 norms -- what a Laplace / radial base density sums, i.e. the log_prob's
            # sensitivity (bar: 1e-5 relative)
            d = ((outs[1] - outs[0]).abs().max() / outs[0].abs().max().clamp_min(1e-30)).item()
            l1 = outs[0].abs().sum(-1)
            d1 = ((outs[1].abs().sum(-1) - l1).abs() / l1.clamp_min(1e-30)).max().item()
        ok = bool(d <= 1e-5 and d1 <= 1e-6)               # (NaN / inf compare False)
        self.merge_guard_log.append((ok, d, d1))
        state[direction] = ok
        return ok

    def _primitive_ops(self, direction: str, merge: bool = False):
        """[(prim, step_index)] with prim in scale_mul/scale_div/affine_fwd/affine_bwd/coupling_fwd/coupling_bwd;
        merge: runs of consecutive affine steps become one ``affine_fwd`` on a ``_MergedAffine`` (``_step(index)``) when
        merging is on for this direction (``_merge_on``: merge_affine True, or "auto" and the probe accepted)"""
        seq = list(enumerate(self.steps))
        if direction == "backward":
            seq = seq[::-1]
        prims = []
        for i, s in seq:
            fwd = (direction == "forward") != s.inverted
            if s.kind == "scale":
                prims.append(("scale_mul" if fwd else "scale_div", i))
            elif s.kind == "affine":
                prims.append(("affine_fwd" if fwd else "affine_bwd", i))
            else:
                prims.append(("coupling_fwd" if fwd else "coupling_bwd", i))
        if not (merge and self._merge_on(direction)):
            return prims
        out, k = [], 0
        while k < len(prims):
            j = k
            while j < len(prims) and prims[j][0] in ("affine_fwd", "affine_bwd"):
                j += 1
            if j - k >= 2:
                parts = [(p_, self.steps[i_].module) for p_, i_ in prims[k:j]]
                key = tuple((p_, id(b_)) for p_, b_ in parts)
                if key not in self._virtual_ix:
                    self._virtual.append(_Step("affine", _MergedAffine(parts), False))
                    self._virtual_ix[key] = len(self.steps) + len(self._virtual) - 1
                out.append(("affine_fwd", self._virtual_ix[key]))
                k = j
            else:
                out.append(prims[k])
                k += 1
        return out

    def _affine_entry(self, pk, blk) -> dict:
        """pk["affine"] entry of a block; a merged run is composed here on first use (fp64, launches on the pack's tape:
        refreshed in place with the parameters)"""
        if id(blk) in pk["affine"] or not isinstance(blk, _MergedAffine):
            return pk["affine"][id(blk)]
        with self._pk_record(pk):
            _ext.flush_jobs()
            W = c = None
            for prim, b_ in blk.parts:
                a = pk["affine"][id(b_)]
                if prim == "affine_fwd":
                    Wk, ck = a["M"], a["b"]
                else:
                    if "c" not in a:        # (y - b) Minv^T == y Minv^T + c, c = -(Minv b)
                        a["c"] = torch.empty(a["b"].shape, dtype=torch.float64, device=a["b"].device)
                        _ext.matvec_f64(a["Minv"], a["b"].contiguous(), alpha=-1.0, out64=a["c"])
                    Wk, ck = a["Minv"], a["c"]
                if W is None:
                    W, c = Wk, ck
                else:
                    W = _ext.matmul_f64(Wk.contiguous(), W.contiguous())
                    t = torch.empty(ck.shape, dtype=torch.float64, device=ck.device)
                    _ext.matvec_f64(Wk.contiguous(), c.contiguous(), alpha=1.0, out64=t)
                    c = _refreshed(tuple(ck.shape), torch.float64, ck.device,
                                   lambda o, t=t, ck=ck: torch.add(t, ck, out=o))
            pk["affine"][id(blk)] = dict(M=W, b=c)
        return pk["affine"][id(blk)]

    def _build_plan(self, direction: str, B: int, device, has_ctx: bool, final: str, train: bool = False) -> dict:
        # every weight image the plan needs is queued while the op list is laid out and packed by ONE batched launch
        # per size class at the end (nothing reads them before the plan runs); the launch joins the pack's tape
        pk = self.pack(device)
        with self._pk_record(pk), _ext.batch_jobs(device):
            if self._planes_ok(direction, B, has_ctx, train):
                return self._build_plan_planes(direction, B, device, final, train)
            return self._build_plan_body(direction, B, device, has_ctx, final, train)

    def _build_plan_body(self, direction: str, B: int, device, has_ctx: bool, final: str, train: bool = False) -> dict:
        """final: 'user' (last op writes the caller's [B,D] tensor) or 'nat' (workspace buffer, for the tail).

        Returns the ctypes op array plus the few launches that are not usf_run_ops ops
        (layout gathers at the ends, stand-alone scale layers), each tagged with the op index
        before which it runs."""
        pk = self.pack(device)
        ws = self._workspace(B, device)
        prims = self._primitive_ops(direction, merge=not train)     # (the training backward needs every block's own launch)
        ops: List[_ext.Op] = []
        patch_in: List[int] = []       # ops whose A is the caller's input tensor
        patch_out: List[int] = []      # ops whose C is the caller's output tensor
        side: List[tuple] = []         # ("gather", at, src_cur, dst_name, dst_layout) | ("scale", at, buf, ld, vec, divide, ncols)
        cur = ("user_in", "nat", self.D)     # (buffer name, layout, row stride)
        free = ["zA", "zB"]
        meta: List[dict] = []          # per group of ops: what the training backward needs (training.py)
        n_act = [0]

        def take():
            if train:
                # training: every affine output keeps its own buffer (the saved activations of the backward
                # pass; (K+1) x B x LD x 4 bytes -- cfg2 at B = 65536: 6.8 GB of the 288 GB)
                name = f"act{n_act[0]}"
                n_act[0] += 1
                if name not in ws:
                    ws[name] = torch.zeros(B, self.LD, dtype=torch.float32, device=device)
                return name
            return free.pop(0)

        def release(name):
            if not train and name in ("zA", "zB") and name not in free:
                free.append(name)

        def lin_op(**kw) -> _ext.Op:
            op = _ext.Op()
            op.kind = _ext.OP_LINEAR
            for k_, v_ in kw.items():
                setattr(op.u.linear, k_, v_)
            return op

        def nat2():
            if "nat2" not in ws:
                ws["nat2"] = torch.zeros(B, self.LDn, dtype=torch.float32, device=device)
            return ws["nat2"]

        n = len(prims)
        k = 0
        while k < n:
            prim, i = prims[k]
            s = self._step(i)
            nxt = prims[k + 1] if k + 1 < n else None
            # ---- affine (optionally with the scale layer fused on its outer side) ---------------
            if prim in ("affine_fwd", "affine_bwd") or (prim == "scale_div" and nxt and nxt[0] in ("affine_bwd", "affine_fwd")):
                if cur[0] == "user_in" and self.D % 4 != 0:
                    # rows of the caller's tensor are not 16-B aligned: stage through a padded copy
                    side.append(("gather", len(ops), cur, "nat", "nat"))
                    cur = ("nat", "nat", self.LDn)
                in_layout = cur[1]
                kw = {}
                scale_mod = None
                if prim == "scale_div":
                    scale_mod = s.module
                    sc64 = pk["scale"][id(s.module)]
                    kw["pre_div"] = self._vec(pk, ("scale", id(s.module)), sc64, in_layout, 1.0).data_ptr()
                    k += 1
                    prim, i = prims[k]
                    s = self._step(i)
                    nxt = prims[k + 1] if k + 1 < n else None
                blk = s.module
                a = self._affine_entry(pk, blk)
                fuse_post = prim == "affine_fwd" and nxt is not None and nxt[0] == "scale_mul"
                is_last = (k == n - 1) or (fuse_post and k == n - 2)
                out_layout = "nat" if is_last else "seg"
                Kdim = self.LD if in_layout == "seg" else self.LDn
                if prim == "affine_bwd":
                    W = self._mat(pk, blk, "Minv", out_layout, in_layout)
                    if "pre_div" in kw:
                        # first layer of log_prob: (x / s - b) @ Minv^T, prologue in the operand registers
                        kw["pre_sub"] = self._vec(pk, ("b", id(blk)), a["b"], in_layout, 0.0).data_ptr()
                    else:
                        # (y - b) @ Minv^T == y @ Minv^T + c with c = -(Minv b), c formed in fp64 at pack
                        # time: keeps the bias out of the K loop's registers ("bias folding")
                        if "c" not in a:
                            a["c"] = torch.empty(a["b"].shape, dtype=torch.float64, device=a["b"].device)
                            with self._pk_record(pk):
                                _ext.matvec_f64(a["Minv"], a["b"].contiguous(), alpha=-1.0, out64=a["c"])
                        kw["bias"] = self._vec(pk, ("c", id(blk)), a["c"], out_layout, 0.0).data_ptr()
                else:
                    W = self._mat(pk, blk, "M", out_layout, in_layout)
                    kw["bias"] = self._vec(pk, ("b", id(blk)), a["b"], out_layout, 0.0).data_ptr()
                    if fuse_post:
                        s2 = self._step(nxt[1])
                        kw["post_mul"] = self._vec(pk, ("scale", id(s2.module)), pk["scale"][id(s2.module)],
                                                   out_layout, 1.0).data_ptr()
                        k += 1
                assert W.shape[1] == Kdim
                kw.update(self._split_kw(pk, W, Kdim))
                if out_layout == "seg":
                    dst = take()
                    Ndim, ldc, cptr = self.LD, self.LD, ws[dst].data_ptr()
                elif final == "user":
                    dst, Ndim, ldc, cptr = "user_out", self.D, self.D, 0
                else:
                    dst, Ndim, ldc, cptr = "nat2", self.D, self.LDn, nat2().data_ptr()
                if cur[0] == "user_in":
                    patch_in.append(len(ops))
                if dst == "user_out":
                    patch_out.append(len(ops))
                # training at thousands of rows: the GEMM also writes the bf16 planes it makes of its input -- the operand
                # of this layer's weight gradient, already split (usf_wgrad_planes_f32; 3 x B x LD x 2 bytes per layer)
                in_planes = None
                if (train and cur[0] != "user_in" and "pre_div" not in kw and "pre_sub" not in kw
                        and self.wgrad_from_planes(B, Ndim, Kdim)):
                    in_planes = f"apl{len(ops)}"
                    t = ws.get(in_planes)
                    if t is None or t.shape[1] != -(-B // 32) * 32 or t.shape[2] != -(-Kdim // 32) * 32:
                        t = ws[in_planes] = _ext.row_planes(B, Kdim, device)
                    kw.update(A_planes_out=t.data_ptr(), ldp_out=t.shape[2], planes_out_stride=t.shape[1] * t.shape[2])
                meta.append(dict(kind="affine", op=len(ops), prim=prim, blk=blk, in_buf=cur[0], in_layout=in_layout,
                                 in_ld=cur[2], out_buf=dst, out_layout=out_layout, out_ld=ldc, N=Ndim, K=Kdim,
                                 pre_scale=scale_mod, post_scale=(self._step(nxt[1]).module if fuse_post else None),
                                 in_planes=in_planes))
                ops.append(lin_op(A=(0 if cur[0] == "user_in" else ws[cur[0]].data_ptr()), lda=cur[2],
                                  W=W.data_ptr(), ldw=W.shape[1], C=cptr, ldc=ldc, M=B, N=Ndim, K=Kdim,
                                  res_sign=1.0, slope=0.0, act=_ext.ACT_NONE, **kw))
                release(cur[0])
                cur = (dst, out_layout, ldc)
                k += 1
                continue
            # ---- stand-alone scale layer: elementwise kernel, in place on a workspace buffer -----
            if prim in ("scale_mul", "scale_div"):
                if cur[0] == "user_in":
                    side.append(("gather", len(ops), cur, "nat", "nat"))
                    cur = ("nat", "nat", self.LDn)
                sc = self._vec(pk, ("scale", id(s.module)), pk["scale"][id(s.module)], cur[1], 1.0)
                side.append(("scale", len(ops), cur[0], cur[2], sc, prim == "scale_div",
                             self.LD if cur[1] == "seg" else self.LDn))
                k += 1
                continue
            # ---- coupling (in place on a segment-layout buffer) -----------------------------------
            if cur[1] != "seg" or cur[0] in ("user_in", "nat", "nat2"):
                dst = take()
                side.append(("gather", len(ops), cur, dst, "seg"))
                release(cur[0])
                cur = (dst, "seg", self.LD)
            cp = pk["coupling"][i]
            sign = 1.0 if prim == "coupling_fwd" else -1.0
            zptr = ws[cur[0]].data_ptr()
            use_ctx = has_ctx and cp["has_ctx"]
            meta.append(dict(kind="coupling", op=len(ops), step=i, buf=cur[0], sign=sign, use_ctx=use_ctx))
            # the fused kernel keeps a wave on 16 rows for the whole MLP: unbeatable when the chip is full, but
            # its latency is one wave's serial MFMA chain; small batches run the MLP as 3 short linear launches
            if cp.get("general"):
                self._general_coupling_ops(ops, lin_op, pk, cp, ws, zptr, B, sign, device)
            elif self.use_fused_coupling and self._fused_ok(cp) and (B >= self.fused_min_rows or self.tiny_coupling(cp, B)):
                tiny = B < self.fused_min_rows
                op = self._coupling_op(cp, zptr, B, sign, ws if use_ctx else None)
                # training: the fused kernel also stores the hidden activations (buffers of the layer's own; 2 x B x 256 x 4
                # bytes per coupling) -- the backward pass reads them instead of running the conditioner a second time
                # (tiny layers at launch-bound batches: usf_coupling_tiny.hip does the same, and the backward chain in one launch)
                meta[-1]["tiny"] = tiny
                if train and (tiny or (self.save_fused_hidden(cp, B) and op.u.coupling.split_in)):
                    for j in range(len(cp["hidden"])):
                        hname = f"Hs{j}_{i}"
                        if hname not in ws or ws[hname].shape[0] != B or ws[hname].shape[1] < self.hmax:
                            ws[hname] = torch.zeros(B, self.hmax, dtype=torch.float32, device=device)
                        op.u.coupling.hidden_out[j] = ws[hname].data_ptr()
                    op.u.coupling.ld_hidden_out = self.hmax
                    meta[-1]["hidden_saved_fused"] = True
                ops.append(op)
            else:
                hbufs = ["H1", "H2"]
                un = self._unfused_pack(pk, cp)
                # training at small batches: every hidden layer of every coupling keeps a buffer of its own -- the backward
                # pass reads the activations from there instead of running the conditioner a second time (2 launches per
                # coupling of a step that is bound by the number of its launches; training.py names the same buffers)
                # (round 4: at every training batch, not only the small ones -- the activations land in per-layer buffers instead of
                # the shared pair at no cost to the forward; USFLOWS_AMD_SAVE_HIDDEN=0: only up to GRAD_JOB_MAX_ROWS rows as before)
                save_h = train and B > 0 and (B <= _ext.GRAD_JOB_MAX_ROWS or config.save_hidden)
                meta[-1]["hidden_saved"] = save_h
                src_ptr, src_ld, src_K = zptr + 4 * cp["pass_off"], self.LD, cp["pass_n"]
                for j, (W, b) in enumerate(un["layers"]):
                    if save_h:
                        hname = f"Hs{j}_{i}"
                        if hname not in ws or ws[hname].shape[0] != B or ws[hname].shape[1] < self.hmax:
                            ws[hname] = torch.zeros(B, self.hmax, dtype=torch.float32, device=device)
                        hb = ws[hname]
                    else:
                        hb = ws[hbufs[j % 2]]
                    kw = {}
                    if j == 0 and use_ctx:
                        # P = ctx * Wc + bc as a K=4 GEMM (context in column 0 of a zero-padded [B,4] operand);
                        # added to (acc + b_in) before the activation, as networks.py:741-745 does
                        ops.append(lin_op(A=ws["ctx4"].data_ptr(), lda=4, W=un["W_ctx4"].data_ptr(), ldw=4,
                                          bias=un["b_ctx"].data_ptr(), C=ws["P"].data_ptr(), ldc=self.hmax,
                                          M=B, N=cp["hidden"][0], K=4, res_sign=1.0, slope=0.0, act=_ext.ACT_NONE))
                        kw = dict(addend=ws["P"].data_ptr(), ldadd=self.hmax)
                    kw.update(self._split_kw(pk, W, src_K))
                    ops.append(lin_op(A=src_ptr, lda=src_ld, W=W.data_ptr(), ldw=W.shape[1], bias=b.data_ptr(),
                                      C=hb.data_ptr(), ldc=self.hmax, M=B, N=W.shape[0], K=src_K, res_sign=1.0,
                                      slope=cp["slope"], act=cp["act"], **kw))
                    src_ptr, src_ld, src_K = hb.data_ptr(), self.hmax, W.shape[0]
                tptr = zptr + 4 * cp["tr_off"]
                ops.append(lin_op(A=src_ptr, lda=src_ld, W=un["W_out"].data_ptr(), ldw=un["W_out"].shape[1],
                                  bias=un["b_out"].data_ptr(), residual=tptr, ldr=self.LD, C=tptr, ldc=self.LD,
                                  M=B, N=cp["tr_n"], K=src_K, res_sign=sign, slope=0.0, act=_ext.ACT_NONE,
                                  **self._split_kw(pk, un["W_out"], src_K)))
            k += 1

        # ---- final layout fix-up ------------------------------------------------------------------
        final_gather = None
        if final == "user" and cur[0] != "user_out":
            final_gather = (cur, "user_out")
        elif final == "nat" and cur[1] != "nat":
            nat2()
            final_gather = (cur, "nat2")
            cur = ("nat2", "nat", self.LDn)
        arr = (_ext.Op * max(len(ops), 1))(*ops)
        return dict(arr=arr, n=len(ops), patch_in=[(i_, "linear", "A") for i_ in patch_in],
                    patch_out=[(i_, "linear", "C") for i_ in patch_out], side=side,
                    final_gather=final_gather, out_buf=cur, ws=ws, pk=pk, meta=meta)

    def tiny_coupling(self, cp, B: int) -> bool:
        """launch-bound batches with tiny conditioners (the reference's live flat configuration, gaussian_mixture.yaml: D <= 100,
        DenseNN [32, 32], batch 32): the whole coupling layer is one launch of the tiny-layer kernel each way
        (usf_coupling_tiny.hip's eligibility rule, restated for the forward AND the backward descriptor: <= 256 rows, segments
        and hidden widths <= 64, the layer's weight images + rows + side inputs in 64 KB of LDS)"""
        from .config import config
        if not config.tiny_coupling or not (0 < B <= 256) or cp.get("general"):
            return False
        hid = [int(h_) for h_ in cp["hidden"]]
        if len(hid) > 3 or max(hid) > 64 or cp["pass_n"] > 64 or cp["tr_n"] > 64:
            return False

        def lds_floats(n_pass, hidden, n_trans):
            r4 = lambda v: (v + 3) // 4 * 4
            f, k, rows_sum = 0, n_pass, 0
            for rows in list(hidden) + [n_trans]:
                f += (rows + 15) // 16 * 16 * (r4(k) + 4)
                rows_sum += rows
                k = rows
            return f + 32 * (r4(n_pass) + 4) + 2 * 32 * 68 + rows_sum + 32 * n_trans + 32 * sum(hidden) + 32 + 128

        return max(lds_floats(cp["pass_n"], hid, cp["tr_n"]), lds_floats(cp["tr_n"], hid[::-1], cp["pass_n"])) * 4 <= 64 * 1024

    def save_fused_hidden(self, cp, B: int) -> bool:
        """training: the fused bf16x3 coupling kernel stores its hidden activations (usf_coupling_desc::hidden_out) -- where
        that kernel serves the layer (hidden width in (128, 256], >= 1024 rows), unless USFLOWS_AMD_SAVE_HIDDEN=0"""
        hm = max(cp["hidden"])
        return (self.gemm_mode == "bf16x3" and config.save_hidden and 128 < hm <= 256
                and B >= 1024 and self.hmax >= 256 and self.hmax % 4 == 0
                and cp["tr_n"] % 4 == 0 and cp["tr_off"] % 4 == 0)      # (the backward launch reads the transformed half as its input)

    def wgrad_from_planes(self, B: int, N: int, K: int) -> bool:
        """weight gradients of the training step from pre-split operand planes (usf_wgrad_planes_f32): in the bf16x3 mode,
        where the kernel pays (its own cross-over), unless USFLOWS_AMD_WGRAD_PLANES=0"""
        rows, wid = -(-B // 32) * 32, -(-max(N, K, self.LD, self.LDn) // 32) * 32
        return (self.gemm_mode == "bf16x3" and config.wgrad_planes
                and 3 * rows * wid * 2 < (1 << 31)          # the three planes of an operand stay below 2 GiB (32-bit offsets)
                and _ext.wgrad_planes_ok(B, N, K))

    def _general_coupling_ops(self, ops, lin_op, pk, cp, ws, zptr, B, sign, device):
        """the vector ConvNet conditioner with GatedMLP / LayerNormVector blocks (networks.py:206-245, 287-308) as a chain
        of linear launches and one row pass per block (usf_gated_norm_rows_f32: gate, layer norm and the activation in
        front of the next Linear in one kernel), then the masked residual in the last Linear's epilogue"""
        un = self._unfused_pack(pk, cp)
        hm = self.hmax

        def buf(name, width=hm):
            key = f"G_{name}"
            if key not in ws:
                ws[key] = torch.zeros(B, width, dtype=torch.float32, device=device)
            return ws[key]

        def linear(Wb, src, src_ld, dst, dst_ld, act=False, **extra):
            W, b = Wb
            K = W.shape[1]
            ops.append(lin_op(A=src, lda=src_ld, W=W.data_ptr(), ldw=K, bias=b.data_ptr(), C=dst, ldc=dst_ld, M=B,
                              N=W.shape[0], K=K, res_sign=extra.pop("res_sign", 1.0), slope=cp["slope"] if act else 0.0,
                              act=cp["act"] if act else _ext.ACT_NONE, **extra, **self._split_kw(pk, W, K)))

        def rows(skip, C_, out, out_act, vg=None, gate_off=0, ln=None, eps=0.0):
            op = _ext.Op()
            op.kind = _ext.OP_GATED_NORM
            g = op.u.gated_norm
            g.skip, g.ld_skip, g.M, g.C, g.c_pad = skip.data_ptr(), skip.shape[1], B, C_, _round_up(C_, 4)
            if vg is not None:
                g.vg, g.ld_vg, g.gate_off = vg.data_ptr(), vg.shape[1], gate_off
            if ln is not None:
                g.gamma, g.beta, g.eps = ln[0].data_ptr(), ln[1].data_ptr(), eps
            if out is not None:
                g.out, g.ld_out = out.data_ptr(), out.shape[1]
            if out_act is not None:
                g.out_act, g.ld_act, g.act, g.slope = out_act.data_ptr(), out_act.shape[1], cp["act"], cp["slope"]
            ops.append(op)

        hcur, hnext, A, T, S, VG = buf("H1"), buf("H2"), buf("A"), buf("T"), buf("S"), buf("VG", 2 * hm)
        linear(un["first"], zptr + 4 * cp["pass_off"], self.LD, hcur.data_ptr(), hm)
        rows(hcur, cp["raw"]["h"][0], None, A)          # a = f(h): the activation in front of block 0's Linear
        nb = len(un["blocks"])
        for j, e in enumerate(un["blocks"]):
            wo = e["w_out"]
            nxt_act = A if j + 1 < nb else None          # the final Linear has no activation in front of it
            if "lin" in e:
                linear(e["lin"], A.data_ptr(), hm, T.data_ptr(), hm)
                rows(T, wo, hnext, nxt_act, ln=e.get("ln"), eps=e["eps"])
            else:
                linear(e["l1"], A.data_ptr(), hm, T.data_ptr(), hm, act=True)
                linear(e["l2"], T.data_ptr(), hm, VG.data_ptr(), 2 * hm)
                skip = hcur
                if "proj" in e:
                    linear(e["proj"], hcur.data_ptr(), hm, S.data_ptr(), hm)
                    skip = S
                rows(skip, wo, hnext, nxt_act, vg=VG, gate_off=_round_up(wo, 4), ln=e.get("ln"), eps=e["eps"])
            hcur, hnext = hnext, hcur
        tptr = zptr + 4 * cp["tr_off"]
        linear((un["W_out"], un["b_out"]), hcur.data_ptr(), hm, tptr, self.LD, residual=tptr, ldr=self.LD, res_sign=sign)

    # fused coupling kernel availability (filled in when the kernel is present)
    def _fused_ok(self, cp) -> bool:
        lib = _ext.load()
        wmax = lib.usf_coupling_max_width()
        return not cp.get("general") and wmax > 0 and len(cp["hidden"]) <= 3 and max(cp["hidden"]) <= wmax

    def _fused_pack(self, cp) -> dict:
        """weights re-laid out for the fused kernel's padding contract (include/usflows_hip.h)"""
        lib = _ext.load()
        raw = cp["raw"]
        dev, h = raw["device"], raw["h"]
        Hp = lib.usf_coupling_padded_width(max(cp["hidden"]))
        Kp = _round_up(cp["pass_n"], 32)
        Np = _round_up(cp["tr_n"], 32)
        split = self.gemm_mode in ("bf16x3", "f16x2") and Hp == 256
        if "fused" in cp and (not split or "split" in cp["fused"]):
            return cp["fused"]
        with self._pk_record(self._pack):
            return self._fused_pack_build(cp, Hp, Kp, Np, split)

    def _fused_pack_build(self, cp, Hp, Kp, Np, split) -> dict:
        raw = cp["raw"]
        dev, h = raw["device"], raw["h"]
        pass_sel = self._sel(raw["pass_idx"], Kp, dev)
        tr_sel = self._sel(raw["tr_idx"], Np, dev)
        s3 = dict(hid=[]) if split else None
        # hidden (K) axes of the split hidden / output planes in the accumulator order of the kernel
        perm_sel = {}

        def kperm(n_valid):
            if n_valid not in perm_sel:
                g, j = torch.arange(4)[:, None], torch.arange(8)[None, :]
                within = torch.where(j < 4, 4 * g + j, 16 + 4 * g + (j - 4)).reshape(-1)          # [32]
                perm = (torch.arange(0, Hp, 32)[:, None] + within[None, :]).reshape(-1)
                perm = torch.where(perm < n_valid, perm, torch.full_like(perm, -1))
                perm_sel[n_valid] = perm.to(device=dev, dtype=torch.int32)
            return perm_sel[n_valid]

        W, b = raw["first"]
        rows0 = self._iarange(h[0], Hp, dev)
        W_in, P = self._packed(W, rows0, Hp, pass_sel, Kp, planes_ld=Kp if split else 0)
        if split:
            s3["in"] = P
        f = dict(Hp=Hp, W_in=W_in, b_in=self._packed_vec(b, rows0, Hp), hid=[])
        for j, (W, b) in enumerate(raw["hidden"]):
            rows = self._iarange(h[j + 1], Hp, dev)
            Wp, P = self._packed(W, rows, Hp, self._iarange(h[j], Hp, dev), Hp,
                                 planes_sel=kperm(h[j]) if split else None, planes_ld=Hp if split else 0)
            f["hid"].append((Wp, self._packed_vec(b, rows, Hp)))
            if split:
                s3["hid"].append(P)
        W, b = raw["last"]
        f["W_out"], P = self._packed(W, tr_sel, Np, self._iarange(h[-1], Hp, dev), Hp,
                                     planes_sel=kperm(h[-1]) if split else None, planes_ld=Hp if split else 0)
        f["b_out"] = self._packed_vec(b, tr_sel, Np)
        if split:
            s3["out"] = P
            f["split"] = s3
        if cp["has_ctx"]:
            Wc, bc = raw["ctx"]
            f["W_ctx"] = self._packed_vec(Wc, rows0, Hp)        # layers[1].weight is [h0, 1]: one column
            f["b_ctx"] = self._packed_vec(bc, rows0, Hp)
        cp["fused"] = f
        return f

    def _fused_pack_bwd(self, pk, cp) -> dict:
        """the conditioner's weights transposed and re-laid out for the fused kernel run BACKWARDS (usf_coupling_desc::gate):
        W_in = W_last^T [hidden, trans], hidden matrices reversed and transposed, W_out = W_first^T [pass, hidden], zero biases"""
        if "fused_bwd" in cp:
            return cp["fused_bwd"]
        lib = _ext.load()
        raw = cp["raw"]
        dev, h = raw["device"], raw["h"]
        Hp = lib.usf_coupling_padded_width(max(cp["hidden"]))
        Kp, Np = _round_up(cp["tr_n"], 32), _round_up(cp["pass_n"], 32)      # the roles of the column segments swap
        tr_sel, pass_sel = self._sel(raw["tr_idx"], Kp, dev), self._sel(raw["pass_idx"], Np, dev)

        def kperm(n_valid):
            g, j = torch.arange(4)[:, None], torch.arange(8)[None, :]
            within = torch.where(j < 4, 4 * g + j, 16 + 4 * g + (j - 4)).reshape(-1)
            perm = (torch.arange(0, Hp, 32)[:, None] + within[None, :]).reshape(-1)
            return torch.where(perm < n_valid, perm, torch.full_like(perm, -1)).to(device=dev, dtype=torch.int32)

        with self._pk_record(pk):
            zeros = torch.zeros(max(Hp, Np), dtype=torch.float32, device=dev)
            W, _b = raw["last"]                                   # [features, h_last]
            W_in, P_in = self._packed(W, self._iarange(h[-1], Hp, dev), Hp, tr_sel, Kp, planes_ld=Kp, transpose=True)
            f = dict(Hp=Hp, W_in=W_in, zeros=zeros, hid=[], split=dict(hid=[]))
            f["split"]["in"] = P_in
            for j in range(len(raw["hidden"]) - 1, -1, -1):       # hidden matrix j maps layer j -> j + 1: backwards j + 1 -> j
                W, _b = raw["hidden"][j]                          # [h_{j+1}, h_j]
                Wp, P = self._packed(W, self._iarange(h[j], Hp, dev), Hp, self._iarange(h[j + 1], Hp, dev), Hp,
                                     planes_sel=kperm(h[j + 1]), planes_ld=Hp, transpose=True)
                f["hid"].append(Wp)
                f["split"]["hid"].append(P)
            W, _b = raw["first"]                                  # [h_0, features]
            f["W_out"], f["split"]["out"] = self._packed(W, pass_sel, Np, self._iarange(h[0], Hp, dev), Hp,
                                                         planes_sel=kperm(h[0]), planes_ld=Hp, transpose=True)
        cp["fused_bwd"] = f
        return f

    def coupling_backward_op(self, pk, cp, gptr, ld, B, sign, gates, d_out, act=_ext.ACT_GATE) -> _ext.Op:
        """ONE launch for the data-gradient chain of a coupling layer's conditioner: g[:, pass] += sign * MLP^T(g[:, trans]) with
        the (Leaky)ReLU backward from the saved activations `gates` (layer order of the forward); d_out[l] receives the gradient
        at hidden activation l (forward order)"""
        f = self._fused_pack_bwd(pk, cp)
        nl = len(cp["hidden"])
        op = _ext.Op()
        op.kind = _ext.OP_COUPLING
        d = op.u.coupling
        d.z, d.ldz, d.out, d.ldo, d.M = gptr, ld, gptr, ld, B
        d.off_pass, d.n_pass, d.off_trans, d.n_trans = cp["tr_off"], cp["tr_n"], cp["pass_off"], cp["pass_n"]
        d.n_hidden = nl
        for j in range(nl):
            d.hidden[j] = cp["hidden"][nl - 1 - j]
            d.gate[j] = gates[nl - 1 - j].data_ptr()
            d.hidden_out[j] = d_out[nl - 1 - j].data_ptr()
        d.ld_gate = gates[0].shape[1]
        d.ld_hidden_out = d_out[0].shape[1]
        z = f["zeros"].data_ptr()
        d.W_in, d.ldw_in, d.b_in = f["W_in"].data_ptr(), f["W_in"].shape[1], z
        for j, W in enumerate(f["hid"]):
            d.W_hid[j], d.b_hid[j], d.ldw_hid[j] = W.data_ptr(), z, W.shape[1]
        d.W_out, d.ldw_out, d.b_out = f["W_out"].data_ptr(), f["W_out"].shape[1], z
        d.sign, d.slope, d.act = sign, cp["slope"], act
        s3 = f["split"]
        d.split_in, d.split_in_ld, d.split_in_plane = s3["in"].data_ptr(), s3["in"].shape[2], s3["in"].shape[1] * s3["in"].shape[2]
        for j, P in enumerate(s3["hid"]):
            d.split_hid[j] = P.data_ptr()
        if s3["hid"]:
            d.split_hid_ld, d.split_hid_plane = s3["hid"][0].shape[2], s3["hid"][0].shape[1] * s3["hid"][0].shape[2]
        d.split_out, d.split_out_ld, d.split_out_plane = s3["out"].data_ptr(), s3["out"].shape[2], s3["out"].shape[1] * s3["out"].shape[2]
        return op

    def _coupling_op(self, cp, zptr, B, sign, ws_ctx) -> _ext.Op:
        f = self._fused_pack(cp)
        op = _ext.Op()
        op.kind = _ext.OP_COUPLING
        d = op.u.coupling
        d.z, d.ldz, d.out, d.ldo, d.M = zptr, self.LD, zptr, self.LD, B
        d.off_pass, d.n_pass, d.off_trans, d.n_trans = cp["pass_off"], cp["pass_n"], cp["tr_off"], cp["tr_n"]
        d.n_hidden = len(cp["hidden"])
        for j, hh in enumerate(cp["hidden"]):
            d.hidden[j] = hh
        d.W_in, d.ldw_in, d.b_in = f["W_in"].data_ptr(), f["W_in"].shape[1], f["b_in"].data_ptr()
        for j, (W, b) in enumerate(f["hid"]):
            d.W_hid[j], d.b_hid[j], d.ldw_hid[j] = W.data_ptr(), b.data_ptr(), W.shape[1]
        d.W_out, d.ldw_out, d.b_out = f["W_out"].data_ptr(), f["W_out"].shape[1], f["b_out"].data_ptr()
        if ws_ctx is not None:
            d.context = ws_ctx["ctx"].data_ptr()
            d.W_ctx, d.b_ctx = f["W_ctx"].data_ptr(), f["b_ctx"].data_ptr()
        d.sign, d.slope, d.act = sign, cp["slope"], cp["act"]
        if self.gemm_mode in ("bf16x3", "f16x2") and "split" in f:
            s3 = f["split"]
            d.split_in, d.split_in_ld, d.split_in_plane = s3["in"].data_ptr(), s3["in"].shape[2], s3["in"].shape[1] * s3["in"].shape[2]
            for j, P in enumerate(s3["hid"]):
                d.split_hid[j] = P.data_ptr()
            if s3["hid"]:
                d.split_hid_ld, d.split_hid_plane = s3["hid"][0].shape[2], s3["hid"][0].shape[1] * s3["hid"][0].shape[2]
            d.split_out, d.split_out_ld, d.split_out_plane = s3["out"].data_ptr(), s3["out"].shape[2], s3["out"].shape[1] * s3["out"].shape[2]
        return op

    # ---- execution ----------------------------------------------------------------------------
    def _plan(self, direction, B, device, has_ctx, final, train: bool = False):
        pk = self.pack(device)   # may invalidate plans
        key = (direction, B, str(device), has_ctx, final, self.use_fused_coupling, self.gemm_mode, self.fused_min_rows,
               config.tiny_coupling, train, self.use_planes, self.planes_min_rows, self._planes_fmt(), self.planes_min_rows_bf16x3,
               (not train) and self._merge_on(direction), train and self.use_train_planes, train and self.train_planes_min_rows)
        plan = self._plans.get(key)
        if plan is None:
            plan = self._build_plan(direction, B, device, has_ctx, final, train)
            self._plans[key] = plan
        return plan

    def _run_guarded(self, direction, x, out, context, final):
        """plan + run; a planes pass in fp16x2 format whose range guard fired (a NaN or |value| >= 65000 somewhere in
        the flow: fp16 planes cannot carry it) is void and is redone with bf16x3 planes.  The check reads one int32
        back from the device, i.e. it waits for the pass: only plans of >= planes_min_rows rows (milliseconds) do it."""
        B = x.shape[0]
        if self.merge_affine == "auto" and context is None:
            self.resolve_merge(direction, x)
        plan = self._plan(direction, B, x.device, context is not None, final)
        self._run(plan, x, out, context)
        if plan.get("planes_fmt") == _ext.PLANES_F16X2 and int(plan["ws"]["pflag"].item()) != 0:
            plan["ws"]["pflag"].zero_()
            self.f16_fallbacks += 1
            self._f16_overflow = True
            try:
                plan = self._plan(direction, B, x.device, context is not None, final)
                self._run(plan, x, out, context)
            finally:
                self._f16_overflow = False
        return plan

    def _run(self, plan, x: torch.Tensor, out: Optional[torch.Tensor], context):
        """every pass over a workspace -- training forward or no_grad evaluation, they share the (B, device) buffers --
        bumps its generation counter: activations a training forward left there are stale afterwards, and the
        training backward re-runs its forward when it finds the counter moved (training.py)"""
        ws = plan["ws"]
        ws["_gen"] = ws.get("_gen", 0) + 1
        self._execute(plan, x, out, context)

    def _execute(self, plan, x: torch.Tensor, out: Optional[torch.Tensor], context):
        """run the plan's launches on torch's current stream; small batches replay them as ONE hipGraph"""
        if (self.use_graphs and x.shape[0] <= self.graph_max_rows and self.op_timing is None
                and not torch.cuda.is_current_stream_capturing()):
            return self._execute_graph(plan, x, out, context)
        return self._execute_plain(plan, x, out, context)

    def _execute_graph(self, plan, x, out, context):
        """Launch-bound regime (B <= graph_max_rows: ~130 launches of a few microseconds each): the launch list is
        captured once into a hipGraph (torch.cuda.CUDAGraph over the C-ABI launches -- they go to the capturing
        stream) and replayed; the caller's tensors are staged through fixed buffers so the captured pointers stay
        valid.  An in-place pack refresh (new parameter values, same addresses) keeps the graph valid."""
        ws = plan["ws"]
        B = x.shape[0]
        if "x_static" not in ws:
            ws["x_static"] = torch.empty(B, self.D, dtype=torch.float32, device=x.device)
        xs = ws["x_static"]
        xs.copy_(x)
        os_ = None
        if out is not None:
            if "out_static" not in ws:
                ws["out_static"] = torch.empty(B, self.D, dtype=torch.float32, device=x.device)
            os_ = ws["out_static"]
        if context is not None:
            c = context.reshape(B).to(torch.float32)
            ws["ctx4"][:, 0].copy_(c)
            ws["ctx"].copy_(c)
        g = plan.get("graph")
        if g is None:
            if plan.get("graph_warm", 0) < 1:
                # first call: plain launches (builds every lazily packed weight image before anything is captured)
                plan["graph_warm"] = 1
                self._execute_plain(plan, xs, os_, None)
            else:
                torch.cuda.synchronize(x.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._execute_plain(plan, xs, os_, None)
                plan["graph"] = g
                g.replay()
        else:
            g.replay()
            self.launch_count += 1
        if out is not None:
            out.copy_(os_)

    def _execute_plain(self, plan, x: torch.Tensor, out: Optional[torch.Tensor], context):
        ws = plan["ws"]
        B = x.shape[0]
        dev = x.device
        if context is not None:
            c = context.reshape(B).to(torch.float32)
            ws["ctx4"][:, 0].copy_(c)
            ws["ctx"].copy_(c)
        arr = plan["arr"]
        for idx, member, field in plan["patch_in"]:
            setattr(getattr(arr[idx].u, member), field, x.data_ptr())
        for idx, member, field in plan["patch_out"]:
            setattr(getattr(arr[idx].u, member), field, out.data_ptr())
        lib = _ext.load()
        stream = _ext.current_stream(dev)
        pos = 0

        def run_until(end):
            nonlocal pos
            if end > pos and self.op_timing is not None:
                # instrumented mode: one launch per call, bracketed by HIP events on the launch stream
                for j in range(pos, end):
                    op = arr[j]
                    if op.kind == _ext.OP_LINEAR:
                        kind = "linear_bf16x3" if op.u.linear.W_split else "linear"
                        tag = (kind, op.u.linear.M, op.u.linear.N, op.u.linear.K)
                    elif op.kind == _ext.OP_GEMM_PLANES:
                        g_ = op.u.gemm_planes
                        tag = ("gemm_planes", g_.M, g_.N if g_.C_f32 else 32 * g_.c_kbn, 32 * g_.nk)
                    elif op.kind == _ext.OP_PACK_PLANES:
                        tag = ("pack_planes", op.u.pack_planes.M, 32 * op.u.pack_planes.nkb, 0)
                    elif op.kind == _ext.OP_COUPLING_PLANES:
                        c_ = op.u.coupling_planes
                        tag = ("coupling_planes", c_.M, 32 * c_.nk_t, 32 * c_.nk_p)
                    else:
                        tag = ("coupling", op.u.coupling.M, op.u.coupling.n_trans, op.u.coupling.n_pass)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    sub = C.cast(C.byref(arr, j * C.sizeof(_ext.Op)), C.POINTER(_ext.Op))
                    _ext.check(lib.usf_run_ops(sub, 1, stream), "usf_run_ops")
                    e1.record()
                    self.op_timing.append((tag, e0, e1))
                self.launch_count += 1
                pos = end
            elif end > pos:
                sub = C.cast(C.byref(arr, pos * C.sizeof(_ext.Op)), C.POINTER(_ext.Op))
                _ext.check(lib.usf_run_ops(sub, end - pos, stream), "usf_run_ops")
                self.launch_count += 1
                pos = end

        for g in plan["side"]:           # already in op order (stable)
            run_until(g[1])
            if g[0] == "scale":
                _, _, buf, ld, sc, divide, ncols = g
                _ext.scale(ws[buf], ld, ws[buf], ld, B, ncols, sc, divide)
                self.launch_count += 1
            else:
                _, _, src, dst_name, dst_layout = g
                src_t = x if src[0] == "user_in" else ws[src[0]]
                idx = self._gather_index(src[1], dst_layout, dev)
                dst_t = ws[dst_name]
                _ext.gather_cols(src_t, src[2], dst_t, dst_t.shape[1], B, dst_t.shape[1], idx)
                self.launch_count += 1
        run_until(plan["n"])
        fg = plan["final_gather"]
        if fg is not None:
            src, dst_name = fg
            src_t = ws[src[0]]
            if dst_name == "user_out":
                _ext.gather_cols(src_t, src[2], out, self.D, B, self.D, self._gather_index(src[1], "user", dev))
            else:
                _ext.gather_cols(src_t, src[2], ws[dst_name], self.LDn, B, self.LDn,
                                 self._gather_index(src[1], "nat", dev))
            self.launch_count += 1

    def _gather_index(self, src_layout: str, dst_layout: str, device) -> torch.Tensor:
        """int32 index: dst column j <- src column idx[j] (or -1 -> 0)."""
        key = ("gidx", src_layout, dst_layout, str(device))
        cache = self.__dict__.setdefault("_gidx", {})
        if key not in cache:
            src_idx = self._idx(src_layout)
            pos = {int(f): c for c, f in enumerate(src_idx.tolist()) if f >= 0}   # feature -> src column
            if dst_layout == "user":
                feats = list(range(self.D))
            else:
                feats = self._idx(dst_layout).tolist()
            cache[key] = torch.tensor([pos[f] if f >= 0 else -1 for f in feats], dtype=torch.int32, device=device)
        return cache[key]

    # ---- public API ---------------------------------------------------------------------------
    def _check_input(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("FlowEngine runs on ROCm devices only")
        if x.dim() != 2 or x.shape[1] != self.D:
            raise ValueError(f"expected input of shape [B, {self.D}], got {tuple(x.shape)}")
        if x.dtype != torch.float32:
            x = x.float()
        if not x.is_contiguous() or (x.data_ptr() % 16) != 0:
            x = x.contiguous().clone() if (x.data_ptr() % 16) != 0 else x.contiguous()
        return x

    def transform(self, x: torch.Tensor, direction: str, context: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Flow._forward / Flow.backward on the device: natural layout in, natural layout out."""
        x = self._check_input(x)
        B = x.shape[0]
        out = torch.empty(B, self.D, dtype=torch.float32, device=x.device)
        if B == 0:
            return out
        self._run_guarded(direction, x, out, context, "user")
        return out

    def latent(self, x: torch.Tensor, context=None) -> Tuple[torch.Tensor, int, "LogDet"]:
        """backward pass into the workspace: (z buffer [B, ldn], ldn, the flow's LogDet)."""
        x = self._check_input(x)
        plan = self._run_guarded("backward", x, None, context, "nat")
        buf = plan["ws"][plan["out_buf"][0]]
        return buf, plan["out_buf"][2], plan["pk"]["ladj_total"]

    def latent_base_sums(self, x: torch.Tensor, base: int, loc: torch.Tensor, scale: torch.Tensor):
        """backward pass with the Laplace / Normal base density reduced by the last GEMM's epilogue (planes plans, D <= 1024):
        (partial sums [B, 8], how many of the 8 are used, the flow's LogDet) -- ``usf_base_logprob_f32(USF_BASE_ROWSUM)`` finishes
        the rows; None when this batch does not take a planes plan (the caller runs ``latent`` + the density pass)."""
        from .config import config
        x = self._check_input(x)
        B = x.shape[0]
        if not (config.base_in_epilogue and self.D <= 1024 and self._planes_ok("backward", B, False, False)):
            return None
        if self.merge_affine == "auto":
            self.resolve_merge("backward", x)
        final = f"base{int(base)}"
        plan = self._plan("backward", B, x.device, False, final)
        if plan.get("n_part", 0) < 1:
            return None
        ws = plan["ws"]
        _ext.base_tables(base, loc, scale, self.D, ws["btab"], ws["btab"].numel() // 3)
        plan = self._run_guarded("backward", x, None, None, final)
        return plan["ws"]["bpart"], plan["n_part"], plan["pk"]["ladj_total"]

    def ladj_total(self, device) -> float:
        return float(self.pack(device)["ladj_total"])
